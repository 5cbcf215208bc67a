from rel_pose_amd.data_readers.interiornet import *  # noqa: F401,F403  (drop-in alias of reference src/data_readers/interiornet.py)
