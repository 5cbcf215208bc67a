from rel_pose_amd.data_readers.factory import *  # noqa: F401,F403  (drop-in alias of reference src/data_readers/factory.py)
