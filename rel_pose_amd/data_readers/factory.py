"""dataset_factory (reference src/data_readers/factory.py:6-23)."""
from torch.utils.data import ConcatDataset

from .interiornet import InteriorNet
from .matterport import Matterport
from .streetlearn import StreetLearn

DATASETS = {"matterport": Matterport, "streetlearn": StreetLearn, "interiornet": InteriorNet}


def dataset_factory(dataset_list, **kwargs):
    """create a combined dataset"""
    db_list = []
    for key in dataset_list:
        db = DATASETS[key](**kwargs)
        print("Dataset {} has {} images".format(key, len(db)))
        db_list.append(db)
    return ConcatDataset(db_list)
