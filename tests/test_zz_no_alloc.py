"""fs_connected_components_noalloc (the entry point that allocates nothing) against fs_connected_components,
through the C ABI, bit for bit: two calls in a row on the same GPU, the second with a smaller input, so that what the
first call left in the static work space must not leak into the second."""
import numpy as np
import pytest

from conftest import DeviceArrays

pytestmark = pytest.mark.gpu


def _masks(rng, n, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    m = (rng.random((n, H, W)) < rng.choice([0.45, 0.6, 0.8], n)[:, None, None]).astype(np.uint8)
    m[0] = ((xx % 4 == 0) | (yy == H - 1)).astype(np.uint8)          # a comb: one component through every tile
    return m


def _run(ctx, dev, masks, thr):
    n, H, W = masks.shape
    d_mask = dev.upload(masks)
    d_lab = dev.alloc(n * H * W * 8)
    d_cnt = dev.alloc(n * 4)
    ctx.connected_components_noalloc(n, d_mask, d_lab, d_cnt, thr)
    ctx.synchronize()
    return dev.download(d_lab, np.empty((n, H, W), np.int64)), dev.download(d_cnt, np.empty(n, np.int32))


def test_noalloc_matches_allocating_entry_point_twice():
    from imagesegmentation_amd._lib import Context
    rng = np.random.default_rng(5)
    dev = DeviceArrays()
    try:
        with Context(200, 300, max_batch=5) as big, Context(72, 100, max_batch=2) as small:
            for ctx, n, thr in ((big, 5, 20), (small, 2, 0), (small, 1, 20)):      # second and third call: smaller inputs
                masks = _masks(rng, n, ctx.H, ctx.W)
                lab, cnt = _run(ctx, dev, masks, thr)
                ref_lab, ref_cnt = ctx.connected_components(masks, thr)
                assert np.array_equal(cnt, ref_cnt)
                assert np.array_equal(lab, ref_lab)
                assert lab.max() > 0
    finally:
        dev.free()


def test_noalloc_without_count_output_and_limit():
    from imagesegmentation_amd._lib import Context, load
    rng = np.random.default_rng(6)
    dev = DeviceArrays()
    try:
        with Context(64, 64, max_batch=1) as ctx:
            masks = _masks(rng, 1, 64, 64)
            d_mask, d_lab = dev.upload(masks), dev.alloc(64 * 64 * 8)
            ctx.connected_components_noalloc(1, d_mask, d_lab, None)
            ctx.synchronize()
            lab = dev.download(d_lab, np.empty((1, 64, 64), np.int64))
            assert np.array_equal(lab, ctx.connected_components(masks)[0])
            too_many = int(load().fs_noalloc_max_pixels()) // (64 * 64) + 1
            with pytest.raises(RuntimeError, match="exceeds the static work space"):
                ctx.connected_components_noalloc(too_many, d_mask, d_lab, None)
    finally:
        dev.free()
