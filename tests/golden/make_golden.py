#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the C oracle (oracle/fs_oracle.c).

The reference ships no outputs, tests or known-answer vectors for find_surfaces and its
TensorFlow graphs cannot be executed in this environment, so these vectors pin the
*oracle* (and through it the HIP path), not the reference: they freeze the restatement's
results so that any later change to the oracle, the numpy mirror or the kernels shows up.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Inputs: OSD frames 0, 57 and 110 (data/osd, a copy of the reference's Data/ PNGs) and
seeded synthetic scenes (imagesegmentation_amd.synth).  Integer label maps are stored for
the full frame; float stages of the OSD frames only for a 96x128 window (top-left corner
WIN_Y, WIN_X) to keep the files small.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

OSD_FRAMES = (0, 57, 110)
SYNTH_SEEDS = (11, 12, 13)
SYNTH_SHAPE = (96, 128)
WIN_Y, WIN_X, WIN_H, WIN_W = 200, 256, 96, 128


def osd(index):
    from PIL import Image
    d = np.asarray(Image.open(os.path.join(ROOT, "data", "osd", "disparity", f"{index}.png")))
    c = np.asarray(Image.open(os.path.join(ROOT, "data", "osd", "RGB", f"{index}.png")))
    return d, c


def main():
    from oracle import oracle as O
    from imagesegmentation_amd.synth import make_scene
    O.build()
    params = O.gauss_params()
    win = (slice(WIN_Y, WIN_Y + WIN_H), slice(WIN_X, WIN_X + WIN_W))
    for index in OSD_FRAMES:
        depth, rgb = osd(index)
        ref = O.find_surfaces(depth, rgb, params, want_q=True)
        q = ref["q_final"]
        top2 = np.sort(q, axis=-1)[..., -2:]
        np.savez_compressed(
            os.path.join(HERE, f"osd_{index}.npz"),
            patches=ref["patches"].astype(np.int32), depth_edges=ref["depth_edges"].astype(np.int8),
            surfaces=ref["surfaces"].astype(np.int32), num_surfaces=np.int32(ref["num_surfaces"]),
            # pixels whose two largest label probabilities lie within 1e-5 relative of each other:
            # labels may differ from `surfaces` only there (numerical ties of a flat distribution)
            near_tie=((top2[..., 1] - top2[..., 0]) <= 1e-5 * top2[..., 1]),
            window=np.array([WIN_Y, WIN_X, WIN_H, WIN_W]),
            smoothed=ref["smoothed"][win], angles=ref["angles"][win], vectors=ref["vectors"][win],
            points_3d=ref["points_3d"][win])
    for seed in SYNTH_SEEDS:
        depth, rgb = make_scene(seed, *SYNTH_SHAPE)
        ref = O.find_surfaces(depth, rgb, params, want_q=True)
        np.savez_compressed(
            os.path.join(HERE, f"synth_{seed}.npz"),
            depth=depth, rgb=rgb, patches=ref["patches"].astype(np.int32), depth_edges=ref["depth_edges"].astype(np.int8),
            surfaces=ref["surfaces"].astype(np.int32), num_surfaces=np.int32(ref["num_surfaces"]),
            smoothed=ref["smoothed"], angles=ref["angles"], vectors=ref["vectors"], points_3d=ref["points_3d"],
            q_final=ref["q_final"])
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
