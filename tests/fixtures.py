"""Input fixtures: the reference's own data/ images (copied as data under tests/golden/data) and
the synthetic generator of SURVEY.md section 8(d) / BASELINE.json configs[1]."""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


def load_rgb(name):
    """Decoded u8 RGB [H,W,3], the layout DevIL hands SetImageData (GLTexImage.cpp:1136-1142)."""
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(_DATA, name)).convert("RGB")))


def list640():
    with open(os.path.join(_DATA, "list640.txt")) as f:
        return f.read().split()
