"""InteriorNet pairs (reference src/data_readers/interiornet.py)."""
from .panorama import PanoramaPairs


class InteriorNet(PanoramaPairs):
    META, DATA, DATA_T = "interiornet", "interiornet", "interiornet"

    def __init__(self, mode="training", **kwargs):
        self.mode = mode
        super().__init__(name="InteriorNet", **kwargs)
