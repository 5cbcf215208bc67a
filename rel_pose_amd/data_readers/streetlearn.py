"""StreetLearn pairs (reference src/data_readers/streetlearn.py; the translation set lives under data/streetlearn_2016)."""
from .panorama import PanoramaPairs


class StreetLearn(PanoramaPairs):
    META, DATA, DATA_T = "streetlearn", "streetlearn", "streetlearn_2016"

    def __init__(self, mode="training", **kwargs):
        self.mode = mode
        super().__init__(name="StreetLearn", **kwargs)
