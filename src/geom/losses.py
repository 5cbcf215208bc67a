"""Drop-in alias for reference src/geom/losses.py."""
from rel_pose_amd.losses import geodesic_loss  # noqa: F401
