"""`dataset_factory(names, **reader_kwargs)` -- the entry point train.py uses (counterpart of reference
src/data_readers/factory.py:6-23): one reader per requested dataset name, concatenated."""
import torch.utils.data as tud

from . import interiornet, matterport, streetlearn

READERS = {
    "matterport": matterport.Matterport,
    "interiornet": interiornet.InteriorNet,
    "streetlearn": streetlearn.StreetLearn,
}


def dataset_factory(dataset_list, **kwargs):
    unknown = [name for name in dataset_list if name not in READERS]
    if unknown:
        raise KeyError("unknown dataset(s) %s; available: %s" % (unknown, sorted(READERS)))
    parts = []
    for name in dataset_list:
        reader = READERS[name](**kwargs)
        print("Dataset %s has %d images" % (name, len(reader)))
        parts.append(reader)
    return tud.ConcatDataset(parts)
