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


class DeviceArrays:
    """Device memory through the HIP runtime the library itself is linked against (plain hipMalloc / hipMemcpy by
    ctypes), standing in for a caller's own allocations."""

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so.7")
        self.ptrs = []

    def alloc(self, nbytes, fill=0xA5):
        p = self.c.c_void_p()
        assert self.hip.hipMalloc(self.c.byref(p), self.c.c_size_t(max(nbytes, 1))) == 0
        assert self.hip.hipMemset(p, fill, self.c.c_size_t(max(nbytes, 1))) == 0
        self.ptrs.append(p)
        return p.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        assert self.hip.hipMemcpy(self.c.c_void_p(p), a.ctypes.data_as(self.c.c_void_p), self.c.c_size_t(a.nbytes), 1) == 0
        return p

    def download(self, p, like):
        out = np.empty_like(like)
        assert self.hip.hipMemcpy(out.ctypes.data_as(self.c.c_void_p), self.c.c_void_p(p), self.c.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
