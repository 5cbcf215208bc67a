from rel_pose_amd.data_readers.augmentation import *  # noqa: F401,F403  (drop-in alias of reference src/data_readers/augmentation.py)
