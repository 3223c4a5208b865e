import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def arr(x):
    """nested lists of hex strings -> uint64 ndarray"""
    def conv(v):
        return int(v, 16) if isinstance(v, str) else [conv(u) for u in v]
    return np.array(conv(x), dtype=np.uint64)
