#!/usr/bin/env python3
"""Golden vector for the whole OSD workload (all 111 frames of data/osd) from the C oracle.

    python tests/golden/make_golden_osd_all.py     # rewrites tests/golden/osd_all.npz (about 1.5 min on 8 cores)

Like the other fixtures of this directory it freezes the *oracle's* results (the reference ships no
outputs for find_surfaces).  Per frame: num_surfaces, CRC-32 of the int32 patch labels and of the int8
depth edges (both are bit-exact quantities), and the final CRF labels as uint8 together with a bit mask of
the pixels whose two largest label probabilities lie within 1e-5 relative of each other: labels may
differ from the oracle's only there.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

FRAMES = 111


def main():
    from PIL import Image
    from oracle import oracle as O
    O.build()
    params = O.gauss_params()
    ns, crc_p, crc_e, surf, tie = [], [], [], [], []
    for index in range(FRAMES):
        d = np.asarray(Image.open(os.path.join(ROOT, "data", "osd", "disparity", f"{index}.png"))).astype(np.uint16)
        c = np.asarray(Image.open(os.path.join(ROOT, "data", "osd", "RGB", f"{index}.png"))).astype(np.uint8)
        ref = O.find_surfaces(d, c, params, want_q=True)
        assert ref["surfaces"].max() < 256
        top2 = np.sort(ref["q_final"], axis=-1)[..., -2:]
        ns.append(ref["num_surfaces"])
        crc_p.append(zlib.crc32(np.ascontiguousarray(ref["patches"].astype(np.int32)).tobytes()))
        crc_e.append(zlib.crc32(np.ascontiguousarray(ref["depth_edges"].astype(np.int8)).tobytes()))
        surf.append(ref["surfaces"].astype(np.uint8))
        tie.append(np.packbits((top2[..., 1] - top2[..., 0]) <= 1e-5 * top2[..., 1]))
        print(index, ns[-1], flush=True)
    path = os.path.join(HERE, "osd_all.npz")
    np.savez_compressed(path, num_surfaces=np.array(ns, np.int32), crc_patches=np.array(crc_p, np.uint32),
                        crc_edges=np.array(crc_e, np.uint32), surfaces=np.stack(surf), near_tie=np.stack(tie))
    print(os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
