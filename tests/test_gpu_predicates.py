"""The two filtered predicates of the stencil stage on inputs built to sit at their thresholds.

curvature_mask_kernel decides `total < threshold` (find_surfaces.py:232-262) from an approximate score wherever that
score is more than 1e-4 away from the threshold, and the angle filter decides `|(dj - dc) / dc| > 0.3`
(image_processing_models_GPU.py:86-88) by a product test outside a band of 4e-7 around the threshold; inside the bands
both repeat the reference's arithmetic.  Ordinary frames rarely come near either band.  Here surfaces are tuned (by
bisection on the oracle's scores) until their curvature totals straddle the threshold within 2e-4, and depth steps are
placed within a few ulp of 0.3 on both sides; the GPU's curvature mask must equal the oracle's bit for bit and the
filtered angles must agree to the 4e-6 of the device's acos."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, W, CELL = 96, 128, 16


def _bumps(amplitude):
    """A plane at 1500 with one Gaussian bump per 16x16 cell, amplitude[cy, cx] high."""
    yy, xx = np.mgrid[0:H, 0:W]
    cy, cx = yy // CELL, xx // CELL
    dy, dx = yy % CELL - CELL // 2, xx % CELL - CELL // 2
    g = np.exp(-(dy * dy + dx * dx) / (2 * 3.0 ** 2))
    return (1500.0 + amplitude[cy, cx] * g).astype(np.float32)


def test_curvature_totals_at_the_threshold(oracle):
    from imagesegmentation_amd._lib import Context
    thr = float(oracle.THRESHOLD)
    ncy, ncx = H // CELL, W // CELL
    centres = (np.arange(ncy)[:, None] * CELL + CELL // 2, np.arange(ncx)[None, :] * CELL + CELL // 2)
    lo, hi = np.zeros((ncy, ncx)), np.full((ncy, ncx), 60.0)
    assert (oracle.curvature_mask(_bumps(hi))[1][centres] > thr).all()
    for _ in range(60):                                   # per cell: largest amplitude below, smallest above the threshold
        mid = (lo + hi) / 2
        above = oracle.curvature_mask(_bumps(mid))[1][centres] >= thr
        hi = np.where(above, mid, hi)
        lo = np.where(above, lo, mid)
    pick = np.where((np.add.outer(np.arange(ncy), np.arange(ncx)) % 2) == 0, lo, hi)     # checkerboard: just below / just above
    sm = _bumps(pick)
    mask_o, score_o = oracle.curvature_mask(sm)
    near = np.abs(score_o - thr) < 2e-4
    at_centres = score_o[centres]
    assert near[centres].all(), np.abs(at_centres - thr).max()
    assert (at_centres < thr).any() and (at_centres >= thr).any()
    print(f"{int(near.sum())} pixels within 2e-4 of the threshold, closest {np.abs(score_o - thr).min():.2e}")
    with Context(H, W) as ctx:
        _, _, mask_fast = ctx.patches(sm, want_mask=True)            # production path: approximate score + band
        pat, edg, score_g, mask_exact = ctx.patches(sm, want_curvature=True)
    assert np.array_equal(mask_fast, mask_o.astype(np.uint8))
    assert np.array_equal(mask_exact, mask_o.astype(np.uint8))
    assert np.array_equal(score_g, score_o)
    p_o, e_o = oracle.patches(sm)
    assert np.array_equal(pat, p_o) and np.array_equal(edg, e_o)


def test_depth_gate_at_the_threshold(oracle):
    from imagesegmentation_amd._lib import Context
    rng = np.random.default_rng(7)
    ncy, ncx = H // CELL, W // CELL
    base = rng.uniform(600, 3000, (ncy, ncx)).astype(np.float32)
    sm = np.repeat(np.repeat(base, CELL, 0), CELL, 1)
    f03 = np.float32(0.3)
    ratios = []
    for cy in range(ncy):
        for cx in range(ncx):
            dc = base[cy, cx]
            for (py, px), sign in zip(((4, 4), (4, 11), (11, 4), (11, 11)), (1, -1, 1, -1)):
                dj = np.float32(dc + np.float32(sign) * f03 * dc)
                k = int(rng.integers(-3, 4))                          # a few ulp to either side
                for _ in range(abs(k)):
                    dj = np.nextafter(dj, np.float32(np.inf if k > 0 else -np.inf))
                sm[cy * CELL + py, cx * CELL + px] = dj
                ratios.append(np.abs(np.float32(np.float32(dj - dc) / dc)))
    ratios = np.asarray(ratios, np.float32)
    # one ulp of a depth moves the quotient by up to 4e-7 relative, so "a few ulp" of depth is a few 1e-7 of quotient:
    # every spike within 2e-6 relative of the threshold, a good part of them inside the kernel's band of 4e-7 (where it
    # repeats the division) and the rest just outside it on either side (where the product test alone decides)
    dist = np.abs(ratios.astype(np.float64) - float(f03)) / float(f03)
    assert (dist <= 2e-6).all(), dist.max()
    assert (dist <= 4e-7).sum() >= 20
    assert ((dist > 4e-7) & (ratios > f03)).sum() >= 10 and ((dist > 4e-7) & (ratios < f03)).sum() >= 10
    print(f"{len(ratios)} depth steps: {int((dist <= 4e-7).sum())} inside the band, "
          f"{int((ratios > f03).sum())} gated / {int((ratios <= f03).sum())} passed by the reference")
    _, _, raw_o = oracle.normals(sm)
    want = oracle.angle_filter(raw_o, sm)
    with Context(H, W) as ctx:
        ang, _, _, raw = ctx.normals(sm, want_raw=True)
    assert np.abs(raw - raw_o).max() <= 4e-6
    # a tap on the wrong side of the gate would move a window's mean by about |angle| / 49, orders of magnitude more
    assert np.abs(raw_o).max() > 0.5
    assert np.abs(ang - want).max() <= 4e-6, np.abs(ang - want).max()
