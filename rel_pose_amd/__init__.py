from . import _env  # noqa: F401  (MIOpen user-db path for the CNN front-end; see _env.py)
