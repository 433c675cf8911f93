"""The whole headline workload (all 111 OSD frames in one batch, the configuration bench.py times) through
the C ABI, against the committed oracle results of tests/golden/osd_all.npz
(tests/golden/make_golden_osd_all.py): the reference's main.find_surfaces_for_dataset (main.py:25-26)."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import osd_frame

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "osd_all.npz")
# per-frame number of labels that differed from the oracle's when the fixture was made on the GPU
# (tests/golden/make_osd_label_diffs.py): 12 139 of 34 million, at most 285 in a frame, every one at a numerical tie
LABEL_DIFFS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "osd_label_diffs.json")


def test_all_osd_frames_in_one_batch():
    g = np.load(GOLDEN)
    n = len(g["num_surfaces"])
    frames = [osd_frame(i) for i in range(n)]
    d = np.stack([f[0] for f in frames]).astype(np.uint16)
    c = np.stack([f[1] for f in frames]).astype(np.uint8)
    from imagesegmentation_amd._lib import Context
    with Context(480, 640, max_batch=n) as ctx:
        got = ctx.find_surfaces_batch(d, c)
    assert np.array_equal(got["num_surfaces"], g["num_surfaces"])
    with open(LABEL_DIFFS) as f:
        recorded = json.load(f)["label_differences"]
    worst, total = 1.0, 0
    for i in range(n):
        # integer stages: bit-exact
        assert zlib.crc32(np.ascontiguousarray(got["patches"][i].astype(np.int32)).tobytes()) == g["crc_patches"][i], i
        assert zlib.crc32(np.ascontiguousarray(got["depth_edges"][i].astype(np.int8)).tobytes()) == g["crc_edges"][i], i
        # CRF labels: equal except where the oracle's two largest probabilities are within 1e-5 relative
        dis = got["surfaces"][i] != g["surfaces"][i]
        if dis.any():
            tie = np.unpackbits(g["near_tie"][i])[:dis.size].reshape(dis.shape).astype(bool)
            assert not (dis & ~tie).any(), f"frame {i}: {int((dis & ~tie).sum())} label differences that are not ties"
        # ... and no more of them than recorded: the kernels are deterministic, so a frame that shows more has changed
        # its arithmetic, if only inside the ties (stale operands, another summation order)
        assert int(dis.sum()) <= recorded[i], f"frame {i}: {int(dis.sum())} label differences, {recorded[i]} recorded"
        total += int(dis.sum())
        worst = min(worst, 1.0 - dis.mean())
    print(f"label differences: {total} in {n} frames ({sum(recorded)} recorded), worst frame agreement {worst:.5f}")
    assert worst >= 0.999, worst


def test_all_osd_frames_one_per_call():
    """The same 111 frames one per call (what the reference's in-memory callers do, find_surface_model()(x)): every call
    launches its mean-field updates ahead of the label count.  The results are the batch's, bit for bit, and the guess
    fails only on the frames whose label count exceeds what the context had allocated until then."""
    g = np.load(GOLDEN)
    n = len(g["num_surfaces"])
    frames = [osd_frame(i) for i in range(n)]
    d = np.stack([f[0] for f in frames]).astype(np.uint16)
    c = np.stack([f[1] for f in frames]).astype(np.uint8)
    from imagesegmentation_amd._lib import Context
    with Context(480, 640, max_batch=n) as ctx:
        batch = ctx.find_surfaces_batch(d, c)
    with Context(480, 640, max_batch=1) as ctx:
        for i in range(n):
            one = ctx.find_surfaces_batch(d[i], c[i])
            assert one["num_surfaces"] == g["num_surfaces"][i], i
            for k in ("patches", "depth_edges", "surfaces", "angles", "vectors", "points_3d"):
                assert np.array_equal(one[k], batch[k][i]), (i, k)
        held, repeated = ctx.speculation_counts()
    assert held + repeated == n
    # the planes start at 64 channels and grow by a quarter beyond what a frame needs: one growth step (OSD's largest
    # frame has 65 channels) at most
    assert repeated <= 1, (held, repeated)
    print(f"one frame per call: the early launch stood for {held} of {n} frames")
