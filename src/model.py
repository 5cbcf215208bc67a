"""Drop-in alias: `from src.model import ViTEss` (reference train.py:15, demo.py:18) resolves to the MI355X build."""
from rel_pose_amd.model import ViTEss  # noqa: F401
