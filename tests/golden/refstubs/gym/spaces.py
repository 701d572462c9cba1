import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low)
        self.high = np.asarray(high)
        self.shape = shape if shape is not None else self.low.shape
        self.dtype = dtype
