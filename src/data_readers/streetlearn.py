from rel_pose_amd.data_readers.streetlearn import *  # noqa: F401,F403  (drop-in alias of reference src/data_readers/streetlearn.py)
