"""Floating-point parity of the production mean-field kernels (the fused r = 7 kernel and the rolling-window kernel of
the other radii), update by update: fs_find_surfaces_trace runs the stage with the diagnostic builds of those kernels
(same arithmetic plus stores) and returns, for each of the three calls of conv_crf_Gauss(...).predict at
find_surfaces.py:517-529, the softmax argument unary + weight * compatibility (:361) and the probabilities (:363) of
every label at every pixel with a depth.  The C oracle runs the same three updates; the bound is the north star's:
1e-4 relative on the arguments, on every valid pixel and every channel (measured figures are printed).

Labels: the oracle's label is the first maximum of its float32 probabilities (:531).  Wherever its two largest
arguments lie further apart than 2e-5 relative (the measured error of the arguments is 1e-6 to 1.2e-5) the GPU's label must be that one; the remaining differences are
counted and printed."""
import zlib

import numpy as np
import pytest

from conftest import osd_frame

pytestmark = pytest.mark.gpu

REL_BOUND = 1e-4         # arguments, relative, every valid pixel and channel (the north star's bound; measured: 1e-6, 1.2e-5 at r = 15)
LABEL_BOUND = 2e-5       # relative gap of the two largest arguments beyond which the label must be the oracle's
ABS_Q_BOUND = 5e-5       # probabilities, absolute


def oracle_updates(oracle, depth, rgb, r, patches=None, num_surfaces=None):
    """The three updates of find_surfaces.py:515-530 on the oracle -> patches, num_surfaces, [potentials], [probabilities]
    (each [H, W, S]).  patches / num_surfaces given: a crop of a larger frame keeps the frame's labels."""
    params = oracle.gauss_params()
    sm = oracle.smooth_depth(depth.astype(np.float32))
    _, _, raw = oracle.normals(sm)
    ang = oracle.angle_filter(raw, sm)
    if patches is None:
        patches, _ = oracle.patches(sm)
        num_surfaces = int(patches.max()) + 1
    unary, Q, prob = oracle.unary_q0(patches, num_surfaces)
    pots, qs = [], []
    for it in range(3):
        out, pot = oracle.crf_frame(unary, Q, depth, rgb, ang, params, r, want_potentials=True)
        pots.append(pot)
        qs.append(out)
        Q = out.copy()
        Q[depth == 0] = prob if it < 2 else 0.0
    return patches, num_surfaces, pots, qs, ang


def compare_updates(got_pot, got_q, pots, qs, valid, what):
    """got_* [3, S, h, w] label planes, pots / qs lists of [h, w, S]; valid [h, w]: pixels to compare"""
    worst = 0.0
    for it in range(3):
        g = np.moveaxis(got_pot[it], 0, -1)[valid]
        o = pots[it][valid]
        assert np.isfinite(g).all(), (what, it)
        rel = np.abs(g - o) / np.maximum(np.abs(o), 1e-30)
        dq = np.abs(np.moveaxis(got_q[it], 0, -1)[valid] - qs[it][valid]).max()
        print(f"{what}: update {it + 1}: arguments max rel {rel.max():.3e} (mean {rel.mean():.2e}), probabilities max abs {dq:.3e}")
        assert rel.max() <= REL_BOUND, (what, it, rel.max())
        assert dq <= ABS_Q_BOUND, (what, it, dq)
        worst = max(worst, rel.max())
    return worst


def check_labels(got_labels, pot_last, q_last, valid, num_surfaces, what):
    """got_labels [h, w] against the oracle's last update (pot_last, q_last [h, w, S])"""
    ref = np.argmax(q_last, axis=-1)                       # first maximum of the float32 probabilities
    ref[ref == num_surfaces] = 0
    ref[~valid] = 0
    arg = np.sort(-pot_last, axis=-1)
    clear = (arg[..., -1] - arg[..., -2]) > LABEL_BOUND * np.abs(arg[..., -1])
    dis = (got_labels != ref) & valid
    bad = dis & clear
    print(f"{what}: {int(dis.sum())} of {int(valid.sum())} labels differ from the oracle's, all inside the argument bound: {not bad.any()}")
    assert not bad.any(), (what, int(bad.sum()))
    return int(dis.sum())


@pytest.mark.parametrize("H,W,seeds", [(96, 128, (31, 32)), (100, 150, (33,))])
def test_small_frames_r7(oracle, H, W, seeds):
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_scene
    with Context(H, W, max_batch=1) as ctx:
        for seed in seeds:
            d, c = make_scene(seed, H, W)
            pat, ns, pots, qs, _ = oracle_updates(oracle, d, c, 7)
            got = ctx.find_surfaces_trace(d, c)
            assert got["num_surfaces"] == ns and np.array_equal(got["patches"], pat)
            valid = d != 0
            compare_updates(got["potentials"], got["probabilities"], pots, qs, valid, f"{H}x{W} seed {seed}")
            check_labels(got["surfaces"], pots[2], qs[2], valid, ns, f"{H}x{W} seed {seed}")
            # the diagnostic build computes what the production build computes: same labels
            assert np.array_equal(got["surfaces"], ctx.find_surfaces_batch(d, c)["surfaces"])


@pytest.mark.parametrize("index", [0, 57])
def test_osd_frames_r7(oracle, index):
    from imagesegmentation_amd._lib import Context
    d, c = osd_frame(index)
    d, c = d.astype(np.uint16), c.astype(np.uint8)
    pat, ns, pots, qs, _ = oracle_updates(oracle, d, c, 7)
    with Context(480, 640, max_batch=1) as ctx:
        got = ctx.find_surfaces_trace(d, c)
        plain = ctx.find_surfaces_batch(d, c)
    assert got["num_surfaces"] == ns and np.array_equal(got["patches"], pat)
    valid = d != 0
    compare_updates(got["potentials"], got["probabilities"], pots, qs, valid, f"OSD {index}")
    check_labels(got["surfaces"], pots[2], qs[2], valid, ns, f"OSD {index}")
    assert np.array_equal(got["surfaces"], plain["surfaces"])


def _crop_check(oracle, ctx, d, c, r, window, what):
    """A full frame through the GPU, a crop of it through the oracle (the oracle's three updates at r = 15 on 1280 x 960
    would take minutes): the crop keeps the frame's patch labels, and its border acts as a frame border for the oracle,
    so the comparison covers the pixels at least 3 r from the crop's sides that are not the frame's own."""
    H, W = d.shape
    got = ctx.find_surfaces_trace(d, c)
    ns = got["num_surfaces"]
    y0, y1, x0, x1 = window
    dc, cc = d[y0:y1, x0:x1], c[y0:y1, x0:x1]
    # smoothed depth, angles: stencils of the full frame, so the crop's own would differ near its sides -> full frame
    params = oracle.gauss_params()
    sm = oracle.smooth_depth(d.astype(np.float32))
    _, _, raw = oracle.normals(sm)
    ang = oracle.angle_filter(raw, sm)[y0:y1, x0:x1]
    pat = got["patches"][y0:y1, x0:x1]
    unary, Q, prob = oracle.unary_q0(np.ascontiguousarray(pat), ns)
    pots, qs = [], []
    for it in range(3):
        out, pot = oracle.crf_frame(unary, Q, dc, cc, ang, params, r, want_potentials=True)
        pots.append(pot)
        qs.append(out)
        Q = out.copy()
        Q[dc == 0] = prob if it < 2 else 0.0
    m = 3 * r
    inner = np.zeros(dc.shape, bool)
    inner[(m if y0 > 0 else 0):(dc.shape[0] - m if y1 < H else dc.shape[0]), (m if x0 > 0 else 0):(dc.shape[1] - m if x1 < W else dc.shape[1])] = True
    valid = inner & (dc != 0)
    assert valid.sum() > 5000
    gp = got["potentials"][:, :, y0:y1, x0:x1]
    gq = got["probabilities"][:, :, y0:y1, x0:x1]
    compare_updates(gp, gq, pots, qs, valid, what)
    check_labels(got["surfaces"][y0:y1, x0:x1], pots[2], qs[2], valid, ns, what)
    assert np.array_equal(got["surfaces"], ctx.find_surfaces_batch(d, c)["surfaces"])
    return got


@pytest.mark.parametrize("r,n_patches,seed", [(10, 30, 3), (15, 62, 2), (15, 30, 2), (7, 14, 2)])      # the cases of tests/golden/make_golden_1280.py
def test_large_frames(oracle, r, n_patches, seed):
    """1280 x 960 mosaic frames with a prescribed channel count: the rolling-window kernels at r = 10 and 15 (and the
    r = 7 kernel on a large frame), on a window at the frame's top left corner (frame borders included) and one inside."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_mosaic_scene
    H, W = 960, 1280
    d, c = make_mosaic_scene(seed, H, W, n_patches)
    channels = n_patches + 2
    with Context(H, W, max_batch=1, kernel_size=r) as ctx:
        side = 160 + 6 * r
        _crop_check(oracle, ctx, d, c, r, (0, side, 0, side + 64), f"1280x960 r={r} S={channels} corner")
        _crop_check(oracle, ctx, d, c, r, (400, 400 + side, 600, 600 + side + 64), f"1280x960 r={r} S={channels} inside")
