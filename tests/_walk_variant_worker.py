"""Worker of tests/test_gpu_walk_variants.py: the labels, and on one small frame the softmax arguments and probabilities of every
update, that the library selected by FS_HIP_LIB produces for a fixed set of frames, written to the .npz named on the command
line.  The test runs it once per build of the library and compares the files byte for byte."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def results():
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.load_images import read_png
    from imagesegmentation_amd.synth import make_dense_scene, make_mosaic_scene, make_scene
    out = {}
    base = os.path.join(ROOT, "data", "osd")
    idx = [0, 13, 29, 57, 64, 81, 97, 110]
    depth = np.stack([read_png(os.path.join(base, "disparity", f"{i}.png")) for i in idx]).astype(np.uint16)
    rgb = np.stack([read_png(os.path.join(base, "RGB", f"{i}.png")) for i in idx]).astype(np.uint8)
    with Context(480, 640, max_batch=len(idx)) as ctx:                      # r = 7: the static-tile kernel
        out["osd_r7"] = ctx.find_surfaces_batch(depth, rgb)["surfaces"]
    d, c = make_mosaic_scene(5, 960, 1280, 62)
    for r in (10, 15):                                                      # rolling kernel, 8 channels per walk
        with Context(960, 1280, max_batch=1, kernel_size=r) as ctx:
            out[f"mosaic_r{r}"] = ctx.find_surfaces_batch(d, c)["surfaces"]
    d, c = make_dense_scene(2, 960, 1280)
    for r in (7, 10, 15):                                                   # the plans of launches for tiles with many labels
        with Context(960, 1280, max_batch=1, kernel_size=r) as ctx:
            out[f"dense_r{r}"] = ctx.find_surfaces_batch(d, c)["surfaces"]
    d, c = make_scene(33, 96, 128)
    for r in (7, 15):                                                       # diagnostic builds: every update's floats
        with Context(96, 128, max_batch=1, kernel_size=r) as ctx:
            t = ctx.find_surfaces_trace(d, c)
            out[f"trace_r{r}_potentials"] = t["potentials"]
            out[f"trace_r{r}_probabilities"] = t["probabilities"]
            out[f"trace_r{r}_surfaces"] = t["surfaces"]
    return out


if __name__ == "__main__":
    np.savez(sys.argv[1], **results())
