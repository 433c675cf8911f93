"""GPU parity of the pairwise reductions: chi-squared histogram distances (float32: 1e-5 relative)
and label-pair overlap counts / IoU (integers: exact)."""
import numpy as np
import pytest

from conftest import DeviceArrays, osd_frame

pytestmark = pytest.mark.gpu


def test_chi_squared_distances(oracle):
    from imagesegmentation_amd import image_processing_models_GPU as M
    from imagesegmentation_amd.utils import rgb_to_Lab
    from oracle import oracle_np as N
    d, c = osd_frame(33)
    ref = oracle.find_surfaces(d, c, oracle.gauss_params())
    st = oracle.surface_stats(ref["surfaces"], ref["patches"], ref["depth_edges"], ref["points_3d"], ref["angles"], rgb_to_Lab(c),
                              None, ref["num_surfaces"])
    S = ref["num_surfaces"]
    cases = [(st["hist_color"], (10, 10), (4, 4)), (st["hist_angle"], (4, 4), (1, 1)), (st["hist_color"], (5, 5), (2, 2))]
    for hist, pool, stride in cases:
        got = M.chi_squared_distances_model_2D(pool, stride)(hist)
        want = N.chi_squared_distances(hist, pool, stride)
        assert got.shape == (S, S) and got.dtype == np.float32
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
        assert np.array_equal(got, got.T) and (np.diag(got) == 0).all()
        assert (got[1:, 1:] > 0).any()
    # the histograms where fs_surface_stats_dev leaves them (device memory): same kernels, same result
    from imagesegmentation_amd._lib import Context
    mem = DeviceArrays()
    try:
        with Context(480, 640) as ctx:
            for hist, pool, stride in cases:
                want = ctx.chi_squared_distances(hist, pool, stride)
                d_out = mem.alloc(want.nbytes)
                ctx.chi_squared_distances_dev(mem.upload(hist), hist.shape, d_out, pool, stride)
                assert mem.hip.hipDeviceSynchronize() == 0
                assert np.array_equal(mem.download(d_out, want), want)
            with pytest.raises(RuntimeError, match="bad shape"):
                ctx.chi_squared_distances_dev(None, (4, 4, 2), mem.alloc(16))
            with pytest.raises(RuntimeError, match="bad shape"):
                ctx.chi_squared_distances_dev(mem.alloc(256), (4, 4, 2), mem.alloc(16), pool_size=(5, 1))
    finally:
        mem.free()
    rng = np.random.default_rng(1)
    h1 = rng.random((256, S))
    h1[h1 < 0.5] = 0
    got = M.chi_squared_distances_model_1D()(h1)
    assert np.allclose(got, N.chi_squared_distances(h1), rtol=1e-5, atol=1e-6)


def test_iou_information():
    from imagesegmentation_amd import evaluate_models as E
    from imagesegmentation_amd import find_surfaces as fs
    from oracle import oracle_np as N
    rng = np.random.default_rng(2)
    _, _, ann = fs.load_image(20)
    ann = np.asarray(ann).astype(np.int64)
    det = np.roll(ann, (5, -9), (0, 1)) * 2 % 7          # a distorted relabelling with a missing id
    det[rng.random(det.shape) < 0.05] = 0
    va, vd, iou = E.extract_IoU_information(ann, det)
    wa, wd, wiou, woverlap = N.iou_information(ann, det)
    assert va == wa and vd == wd
    assert np.array_equal(iou, wiou)
    assert 0 < iou.max() <= 1
    got_overlap = fs._context(480, 640, 7).overlap_counts(ann, det, int(ann.max()) + 1, int(det.max()) + 1)
    assert np.array_equal(got_overlap, woverlap) and got_overlap.sum() == ann.size
    # identical maps: every annotated object is found exactly
    _, _, iou_same = E.extract_IoU_information(ann, ann)
    present = np.bincount(ann.ravel(), minlength=int(ann.max()) + 1)[1:] > 0
    assert np.array_equal(iou_same[present], np.ones(present.sum()))
    assert np.isclose(E.evaluate_Avg_max_IoU([(ann, ann)]), present.mean())
    # a large label alphabet takes the path without LDS privatisation
    big_a = rng.integers(0, 120, ann.shape)
    big_o = rng.integers(0, 130, ann.shape)
    assert np.array_equal(fs._context(480, 640, 7).overlap_counts(big_a, big_o, 120, 130), N.iou_information(big_a, big_o)[3])


def test_nearest_points(oracle):
    from imagesegmentation_amd import process_surfaces as ps
    d, c = osd_frame(33)
    ref = oracle.find_surfaces(d, c, oracle.gauss_params())
    S = ref["num_surfaces"]
    off, pts = oracle.extract_points(ref["patches"], S)
    sets = [pts[off[i]:off[i + 1]] for i in range(S)]
    rng = np.random.default_rng(4)
    targets = [rng.random((int(rng.integers(1, 6)), 2)).astype(np.float32) * [640, 480] for _ in range(S)]
    targets[2] = np.round(targets[2])                  # integer targets produce ties between equidistant pixels
    got = ps.calculate_nearest_points(sets, targets)
    assert len(got) == S
    for i in range(S):
        diffs = np.sum(np.square(sets[i][None, :, :] - targets[i].astype(np.float32)[:, None, :]), axis=-1)   # the reference's expression, :642-645
        want = sets[i][np.argmin(diffs, axis=1)].astype(np.int32)
        assert got[i].dtype == np.int32 and np.array_equal(got[i], want), i
    centres = ps.prepare_border_centers([[1], [], [0, 1]], np.arange(18, dtype=float).reshape(3, 3, 2))
    assert [c.shape for c in centres] == [(1, 2), (1, 2), (2, 2)] and centres[1].sum() == 0
