#!/usr/bin/env python3
"""Where a tile's time goes in the r = 7 CRF update: cycle counts per phase from an experiment build.
    bash tools/build_variant.sh clocks -DFS_TILE_CLOCKS=1
    FS_HIP_LIB=$PWD/imagesegmentation_amd/libfs_hip_clocks.so python tools/phase_clocks.py [frames]
Runs the OSD frames as one batch (host arrays; the copies do not matter here) and prints, per phase, the cycles the first
lane of a block spent there per tile that had work (2.4 cycles per ns)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from imagesegmentation_amd import _lib  # noqa: E402

NAMES = ["record, lane state, first half of the features", "channel list", "plane table, rest of the features", "Q staged",
         "window walk", "epilogue"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 111
    depth, rgb, _ = bench.load_frames("osd", n, 0, 1, 0)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    out = (ctypes.c_ulonglong * 12)()
    with _lib.Context(depth.shape[1], depth.shape[2], max_batch=len(depth)) as ctx:
        ctx.find_surfaces_batch(depth, rgb)
        lib.fs_debug_phase_clocks(out)
        for _ in range(3):
            ctx.find_surfaces_batch(depth, rgb)
        if lib.fs_debug_phase_clocks(out):
            raise SystemExit("fs_debug_phase_clocks failed")
    v = np.array(list(out), dtype=np.float64)
    work, tiles = v[6], v[7]
    print(f"{int(tiles)} tile visits, {int(work)} with work ({len(depth)} frames x 3 updates x 3 calls)")
    total = v[:6].sum()
    for name, c in zip(NAMES, v[:6]):
        print(f"  {name:48s} {c / work:9.0f} cycles per worked tile  {c / work / 2400:6.2f} us  {100 * c / total:5.1f} %")
    print(f"  {'sum':48s} {total / work:9.0f} cycles  {total / work / 2400:6.2f} us")
    if v[8] or v[9]:
        print(f"  of the first phase: ticket and descriptor {v[8] / work / 2400:5.2f} us, own code up to the barrier {v[9] / work / 2400:5.2f} us, "
              f"waiting at the barrier {v[0] / work / 2400:5.2f} us")
        print(f"  (the first phase's line above is then the barrier wait alone; the three add up to {(v[0] + v[8] + v[9]) / work / 2400:5.2f} us)")


if __name__ == "__main__":
    main()
