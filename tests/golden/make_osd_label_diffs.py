#!/usr/bin/env python3
"""Per-frame count of CRF labels that differ between the HIP path and the oracle's labels of osd_all.npz, for all
111 OSD frames in one batch.  Needs the GPU:

    python tests/golden/make_osd_label_diffs.py [out.json]     # default tests/golden/osd_label_diffs.json

tests/test_gpu_workload.py requires that no frame ever shows more differences than recorded here (the kernels are
deterministic, so the counts are reproducible): every difference sits at a pixel whose two largest probabilities are
numerically tied in the oracle, and a change that moves more labels, even inside those ties, has to be looked at.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import osd_frame
    from imagesegmentation_amd._lib import Context
    g = np.load(os.path.join(HERE, "osd_all.npz"))
    n = len(g["num_surfaces"])
    frames = [osd_frame(i) for i in range(n)]
    d = np.stack([f[0] for f in frames]).astype(np.uint16)
    c = np.stack([f[1] for f in frames]).astype(np.uint8)
    with Context(480, 640, max_batch=n) as ctx:
        got = ctx.find_surfaces_batch(d, c)
    diffs = [int((got["surfaces"][i] != g["surfaces"][i]).sum()) for i in range(n)]
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "osd_label_diffs.json")
    with open(out, "w") as f:
        json.dump({"frames": n, "pixels_per_frame": 480 * 640, "label_differences": diffs, "total": sum(diffs)}, f)
        f.write("\n")
    print("total", sum(diffs), "worst frame", max(diffs), "agreement of the worst frame", 1 - max(diffs) / (480 * 640))


if __name__ == "__main__":
    main()
