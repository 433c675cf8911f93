import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def osd_frame(index):
    """One OSD frame from data/osd (the files load_images.load_image opens)."""
    from PIL import Image
    d = np.asarray(Image.open(os.path.join(ROOT, "data", "osd", "disparity", f"{index}.png")))
    c = np.asarray(Image.open(os.path.join(ROOT, "data", "osd", "RGB", f"{index}.png")))
    return d, c


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
