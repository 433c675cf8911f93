"""GPU tests of the drop-in layer and of the wider parity cases: the mirror of the reference's
find_surfaces / post_processing modules (fused path, operator path, file output), window
radii other than 7, 1280x960 frames, frames with many labels, and size-independent
properties (batch invariance, shard invariance).  Label maps of the patch stage are compared
bit for bit; CRF labels may differ from the oracle only at numerical ties (two largest label
probabilities within 1e-5 relative) and must agree on >= 99.9 % of a full frame."""
import os

import numpy as np
import pytest

from conftest import ROOT, osd_frame

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, "tests", "golden")
KEYS = {"depth": np.uint16, "rgb": np.uint8, "angles": np.float32, "vectors": np.float32, "points_3d": np.float32,
        "patches": np.int64, "depth_edges": np.int32, "surfaces": np.int64}


def _ties_only(got, ref, q_final):
    dis = got != ref
    if not dis.any():
        return True
    q = np.sort(q_final[dis], axis=-1)
    return bool(((q[:, -1] - q[:, -2]) <= 1e-5 * q[:, -1]).all())


def test_find_surface_model_matches_golden_osd_110():
    """find_surface_model()(image_data) on the OSD frame with the most surfaces (65 channels),
    against the committed golden vectors."""
    from imagesegmentation_amd import find_surfaces as fs
    g = np.load(os.path.join(GOLD, "osd_110.npz"))
    depth, rgb, ann = fs.load_image(110)
    data = fs.find_surface_model("Gauss")((depth, rgb, ann))
    for k, dt in KEYS.items():
        assert data[k].dtype == dt, (k, data[k].dtype)
    assert isinstance(data["num_surfaces"], int) and data["num_surfaces"] == int(g["num_surfaces"])
    assert data["angles"].shape == (480, 640, 2) and data["surfaces"].shape == (480, 640)
    assert np.array_equal(data["patches"], g["patches"])
    assert np.array_equal(data["depth_edges"], g["depth_edges"])
    dis = data["surfaces"] != g["surfaces"]
    assert dis.mean() <= 0.001
    assert not (dis & ~g["near_tie"]).any()
    y0, x0, h, w = g["window"]
    win = (slice(y0, y0 + h), slice(x0, x0 + w))
    assert np.array_equal(data["points_3d"][win], g["points_3d"])
    assert np.array_equal(data["vectors"][win], g["vectors"])
    assert np.abs(data["angles"][win] - g["angles"]).max() <= 4e-6


def test_operator_path_equals_fused_path(oracle):
    """The reference's own sequence (four models, get_inputs tiles, predict x3) over the
    operator-level entry points gives the labels of the fused whole-frame path."""
    from imagesegmentation_amd import find_surfaces as fs
    g = np.load(os.path.join(GOLD, "synth_12.npz"))
    depth, rgb = g["depth"], g["rgb"]
    H, W = depth.shape
    fused = fs.find_surfaces(fs.get_GPU_models("Gauss", 7), (depth, rgb, None), "Gauss", 7)
    models = (fs._SmoothingModel(7), fs._SurfaceModel(7), fs._NormalsModel(7),
              fs.conv_crf_Gauss(*fs.load_Gauss_parameters(), 7, H // fs.div_y, W // fs.div_x))
    stepwise = fs.find_surfaces(models, (depth, rgb, None), "Gauss", 7)
    for k in ("patches", "depth_edges", "points_3d", "vectors", "angles"):
        assert np.array_equal(fused[k], stepwise[k]), k
    assert np.array_equal(fused["patches"], g["patches"])
    assert fused["num_surfaces"] == stepwise["num_surfaces"] == int(g["num_surfaces"])
    assert _ties_only(stepwise["surfaces"], g["surfaces"], g["q_final"])
    assert _ties_only(fused["surfaces"], g["surfaces"], g["q_final"])
    print("label disagreement:", float((fused["surfaces"] != stepwise["surfaces"]).mean()))
    assert (fused["surfaces"] != stepwise["surfaces"]).mean() <= 0.0015


def test_find_surfaces_for_indices_writes_the_reference_files(tmp_path, monkeypatch):
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import load_images as li
    monkeypatch.setattr(li, "out_root", str(tmp_path / "out"))
    res = fs.find_surfaces_for_indices([5, 57], save_data=True, return_results=True)
    assert len(res) == 2
    single = fs.find_surface_model()(fs.load_image(57))
    assert np.array_equal(res[1]["surfaces"], single["surfaces"])       # batch of 2 == batch of 1
    assert np.array_equal(res[1]["patches"], single["patches"])
    back = li.load_image_and_surface_information(57)
    for k in ("surfaces", "patches", "depth_edges", "angles", "vectors", "points_3d", "depth"):
        assert np.array_equal(back[k], res[1][k]) and back[k].dtype == KEYS[k], k
    assert sorted(os.listdir(tmp_path / "out")) == ["5", "57"]
    assert fs.find_surfaces_for_indices([5], save_data=False) is None


def test_apply_crf_matches_oracle(oracle):
    """post_processing.apply_CRF: one update seeded with a segmentation, then argmax."""
    from imagesegmentation_amd import post_processing as pp
    g = np.load(os.path.join(GOLD, "synth_13.npz"))
    depth, rgb, angles = g["depth"], g["rgb"], g["angles"]
    final = g["surfaces"].astype(np.int64)
    val = int(final.max()) + 1
    unary, q0, _ = oracle.unary_q0(final, val)
    Q = oracle.crf_frame(unary, q0, depth, rgb, angles, oracle.gauss_params(), 7)
    Q[depth == 0] = 0
    ref = np.argmax(Q, axis=-1)
    ref[ref == val] = 0
    data = {"final_surfaces": final.copy(), "depth": depth, "rgb": rgb, "angles": angles}
    out = pp.get_postprocessing_model()(data)["final_surfaces"]
    assert out.dtype == np.int64 and out.shape == final.shape
    assert _ties_only(out, ref, Q)
    print("label disagreement:", float((out != ref).mean()))
    assert (out != ref).mean() <= 0.0015
    assert pp.get_postprocessing_model(do_post_processing=False)(data) is data


def test_apply_crf_rejects_labels_out_of_range_and_recovers():
    """Labels are narrowed and range-checked on the device: a label beyond the channel limit or a negative one fails
    the call with a message, and the same context then processes a valid segmentation."""
    from imagesegmentation_amd._lib import Context
    g = np.load(os.path.join(GOLD, "synth_13.npz"))
    depth, rgb, angles = g["depth"], g["rgb"], g["angles"]
    final = g["surfaces"].astype(np.int64)
    H, W = final.shape
    with Context(H, W, max_batch=1) as ctx:
        good = ctx.apply_crf(final, depth, rgb, angles)
        for bad_value in (511, -1, 2 ** 40):
            bad = final.copy()
            bad[H // 2, W // 2] = bad_value
            with pytest.raises(RuntimeError, match="outside"):
                ctx.apply_crf(bad, depth, rgb, angles)
        assert np.array_equal(ctx.apply_crf(final, depth, rgb, angles), good)


@pytest.mark.parametrize("r", [3, 10, 15])
def test_other_window_radii(oracle, r):
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    d, c = make_batch([31, 32], 96, 128)
    with Context(96, 128, max_batch=2, kernel_size=r) as ctx:
        got = ctx.find_surfaces_batch(d, c)
    for i in range(2):
        ref = oracle.find_surfaces(d[i], c[i], oracle.gauss_params(), r=r, want_q=True)
        assert np.array_equal(got["patches"][i], ref["patches"])
        assert _ties_only(got["surfaces"][i], ref["surfaces"], ref["q_final"]), r
        print("label disagreement:", float((got["surfaces"][i] != ref["surfaces"]).mean()))
        assert (got["surfaces"][i] != ref["surfaces"]).mean() <= 0.0015


def test_frame_1280x960(oracle):
    from imagesegmentation_amd._lib import Context
    # 2x2 mosaic of four OSD frames: real sensor noise at four times the frame size
    tiles = [osd_frame(i) for i in (3, 1, 50, 7)]
    d = np.block([[tiles[0][0], tiles[1][0]], [tiles[2][0], tiles[3][0]]])
    c = np.concatenate([np.concatenate([tiles[0][1], tiles[1][1]], 1), np.concatenate([tiles[2][1], tiles[3][1]], 1)], 0)
    assert d.shape == (960, 1280) and c.shape == (960, 1280, 3)
    with Context(960, 1280, max_batch=1) as ctx:
        got = ctx.find_surfaces_batch(d, c)
    ref = oracle.find_surfaces(d, c, oracle.gauss_params(), want_q=True)
    assert got["num_surfaces"] == ref["num_surfaces"]
    assert np.array_equal(got["patches"], ref["patches"])
    assert np.array_equal(got["depth_edges"], ref["depth_edges"])
    # ~300 label channels: regions no patch reaches within 3 updates hold ~300 equal probabilities, and
    # which one the argmax picks is rounding noise in both implementations -- ties only, >= 99.7 % equal
    assert _ties_only(got["surfaces"], ref["surfaces"], ref["q_final"])
    assert got["num_surfaces"] > 128
    print("label disagreement:", float((got["surfaces"] != ref["surfaces"]).mean()))
    assert (got["surfaces"] != ref["surfaces"]).mean() <= 0.003        # measured 0.25 %, every one of them a tie


def test_batch_and_shard_invariance():
    """Properties that hold at any size: a frame's result does not depend on which other
    frames share its batch, nor on the order, nor on which shard it lands in."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.sharding import shard
    idx = [0, 3, 17, 41, 57, 66, 80, 99, 110]
    frames = [osd_frame(i) for i in idx]
    d = np.stack([f[0] for f in frames])
    c = np.stack([f[1] for f in frames])
    with Context(480, 640, max_batch=len(idx)) as ctx:
        whole = ctx.find_surfaces_batch(d, c)
        perm = np.arange(len(idx))[::-1]
        rev = ctx.find_surfaces_batch(d[perm], c[perm])
        for k in ("surfaces", "patches", "depth_edges", "angles"):
            assert np.array_equal(rev[k][perm], whole[k]), k
        for rank in range(2):
            mine = shard(list(range(len(idx))), rank, 2)
            part = ctx.find_surfaces_batch(d[mine], c[mine])
            assert np.array_equal(part["surfaces"], whole["surfaces"][mine])
            assert np.array_equal(part["num_surfaces"], whole["num_surfaces"][mine])
        one = ctx.find_surfaces_batch(d[4], c[4])
        assert np.array_equal(one["surfaces"], whole["surfaces"][4])
        again = ctx.find_surfaces_batch(d, c)          # a larger batch after smaller ones: per-frame counters kept between calls
        for k in ("surfaces", "patches", "depth_edges"):
            assert np.array_equal(again[k], whole[k]), k
    # labels are a subset of {0} + patch labels, and patch pixels with valid depth keep... nothing is
    # guaranteed about keeping, but every label in the output must exist as a patch
    for i in range(len(idx)):
        assert set(np.unique(whole["surfaces"][i])) <= set(np.unique(whole["patches"][i])) | {0}
        assert (whole["surfaces"][i][d[i] == 0] == 0).all()        # invalid depth is never assigned


def test_position_dependent_weights(oracle):
    """Parameters whose position thetas do not vanish: the per-offset weight table is used instead of
    the uniform-window constants."""
    from imagesegmentation_amd._lib import Context
    g = np.load(os.path.join(GOLD, "synth_11.npz"))
    p = oracle.gauss_params().copy()
    p[4:7] = [3.0, 5.0, 2.0]           # theta_1_1, theta_2_1, theta_3_1 (squared inside)
    p[3] = 0.05                        # and a non-zero bias
    with Context(96, 128, params=p) as ctx:
        got = ctx.find_surfaces_batch(g["depth"], g["rgb"])
    ref = oracle.find_surfaces(g["depth"], g["rgb"], p, want_q=True)
    assert np.array_equal(got["patches"], ref["patches"])
    assert _ties_only(got["surfaces"], ref["surfaces"], ref["q_final"])
    print("label disagreement:", float((got["surfaces"] != ref["surfaces"]).mean()))
    assert (got["surfaces"] != ref["surfaces"]).mean() <= 0.003         # measured 0.20 %, ties only
    assert (ref["surfaces"] != g["surfaces"]).any()      # the parameters matter


def test_degenerate_frames(oracle):
    """Frames without any valid depth, without any patch, and with sides that are no multiple of the
    tile sizes."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 50, 70
    d, c = make_batch([41, 42, 43], H, W)
    d[0] = 0                                       # nothing valid: no patches, every label 0
    d[1] = 1500                                    # a plane seen head-on: one patch
    with Context(H, W, max_batch=3) as ctx:
        got = ctx.find_surfaces_batch(d, c)
    assert got["num_surfaces"][0] == 1 and not got["patches"][0].any() and not got["surfaces"][0].any()
    for i in range(3):
        ref = oracle.find_surfaces(d[i], c[i], oracle.gauss_params(), want_q=True)
        assert got["num_surfaces"][i] == ref["num_surfaces"]
        assert np.array_equal(got["patches"][i], ref["patches"])
        assert np.array_equal(got["depth_edges"][i], ref["depth_edges"])
        assert np.array_equal(got["points_3d"][i], ref["points_3d"])
        assert _ties_only(got["surfaces"][i], ref["surfaces"], ref["q_final"])
    assert got["num_surfaces"][1] == 2 and (got["surfaces"][1][10:-10, 10:-10] == 1).all()
    with pytest.raises(RuntimeError):
        with Context(H, W, max_batch=1) as ctx:
            ctx.find_surfaces_batch(d[:2], c[:2])     # batch larger than max_batch


def _apply_crf_reference(oracle, labels, depth, rgb, angles, r=7):
    """post_processing.apply_CRF on the oracle -> labels, probabilities of the update"""
    val = int(labels.max()) + 1
    unary, q0, _ = oracle.unary_q0(labels, val)
    Q = oracle.crf_frame(unary, q0, depth, rgb, angles, oracle.gauss_params(), r)
    Q[depth == 0] = 0
    ref = np.argmax(Q, axis=-1)
    ref[ref == val] = 0
    return ref, Q


def test_more_channels_around_a_tile_than_the_kernels_keep_in_registers(oracle):
    """The update kernels carry up to 128 channels per tile in registers.  A frame whose tiles see more labels (here
    500 labels in 2 x 2 cells: every tile sees all of them, 503 channels) takes one pass per 128 channels with the
    messages kept in memory in between: slower, and the same result.  The reference has no such limit
    (find_surfaces.py:294 takes the channel count from the tensor)."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 96, 128
    d, c = make_batch([7], H, W)
    rng = np.random.default_rng(5)
    ang = rng.normal(0, 0.3, (H, W, 2)).astype(np.float32)
    for r, cells in ((7, 500), (10, 200), (7, 130)):
        labels = 1 + (np.arange(H // 2)[:, None] * (W // 2) + np.arange(W // 2)[None, :]) % cells
        labels = np.repeat(np.repeat(labels, 2, 0), 2, 1).astype(np.int64)
        labels[40:60, 30:90] = 0                              # and a stretch that belongs to no surface
        with Context(H, W, max_batch=1, kernel_size=r) as ctx:
            out = ctx.apply_crf(labels, d[0], c[0], ang)
            ok = ctx.find_surfaces_batch(d, c)                # the context's regular path still works afterwards
            assert ok["surfaces"].shape == (1, H, W)
        ref, Q = _apply_crf_reference(oracle, labels, d[0], c[0], ang, r)
        assert _ties_only(out, ref, Q)
        print(f"r = {r}, {cells} labels: label disagreement {float((out != ref).mean()):.5f}")
        assert (out != ref).mean() <= 0.0015


@pytest.mark.parametrize("r,cells", [(7, 20), (7, 100), (10, 60), (10, 100), (15, 40), (15, 100)])
def test_tiles_with_many_labels_take_their_own_launches(oracle, r, cells):
    """A frame whose left part belongs to two large surfaces and whose right part is cut into `cells` labels in 2 x 2
    cells: the tiles on the left list a few labels, those on the right all of them.  With the trained parameters at
    r = 7, 10 and 15 an update of such a frame is several launches (the 8-channels-per-walk kernel for the tiles that list
    up to 5 labels, 16 or 12 channels per walk for up to 61 or 45, the many-chunk variant beyond); the result is the
    oracle's wherever its probabilities do not tie."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 64, 256
    d, c = make_batch([17], H, W)
    rng = np.random.default_rng(6)
    ang = rng.normal(0, 0.3, (H, W, 2)).astype(np.float32)
    labels = np.ones((H, W), np.int64)
    labels[H // 2:, :128] = 2
    grid = 3 + (np.arange(H // 2)[:, None] * 64 + np.arange(64)[None, :]) % cells
    labels[:, 128:] = np.repeat(np.repeat(grid, 2, 0), 2, 1)
    labels[20:30, 150:200] = 0                                # and a stretch that belongs to no surface
    with Context(H, W, max_batch=1, kernel_size=r) as ctx:
        out = ctx.apply_crf(labels, d[0], c[0], ang)
        listed = ctx.tile_labels(1)[0][0]
    assert (listed <= 5).any() and listed.max() >= cells, (listed.min(), listed.max())
    ref, Q = _apply_crf_reference(oracle, labels, d[0], c[0], ang, r)
    assert _ties_only(out, ref, Q)
    print(f"r = {r}, {cells} labels on the right: tiles list {sorted(set(listed.ravel().tolist()))}, label disagreement {float((out != ref).mean()):.5f}")
    assert (out != ref).mean() <= 0.0015


def test_wide_frames_in_a_batch_with_ordinary_ones(oracle):
    """A batch that mixes a frame with more than 128 channels per tile with ordinary frames: every frame's result is
    what the frame gives alone (the wide frame takes launches of its own)."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 96, 128
    d, c = make_batch([8, 9, 10], H, W)
    with Context(H, W, max_batch=3) as ctx:
        base = ctx.find_surfaces_batch(d, c)
        labels = base["surfaces"].copy()
        cells = 1 + (np.arange(H // 2)[:, None] * (W // 2) + np.arange(W // 2)[None, :]) % 300
        labels[1] = np.repeat(np.repeat(cells, 2, 0), 2, 1)
        got = ctx.apply_crf(labels, d, c, base["angles"])
        listed = ctx.tile_labels(3)[0]
        alone = [ctx.apply_crf(labels[i], d[i], c[i], base["angles"][i]) for i in range(3)]
    assert listed[1].max() + 3 > 128, listed[1].max()          # the middle frame is a wide one
    assert listed[0].max() + 3 <= 128 and listed[2].max() + 3 <= 128
    for i in range(3):
        assert np.array_equal(got[i], alone[i]), i
    ref, Q = _apply_crf_reference(oracle, labels[1], d[1], c[1], base["angles"][1])
    assert _ties_only(got[1], ref, Q)
    assert (got[1] != ref).mean() <= 0.0015


def test_label_limits_fail_loudly():
    """More labels than the library carries (512 channels per frame) are an error with a message, never a wrong result;
    the context stays usable afterwards."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 96, 128
    d, c = make_batch([7], H, W)
    ang = np.zeros((H, W, 2), np.float32)
    with Context(H, W, max_batch=1) as ctx:
        labels = np.zeros((H, W), np.int64)
        labels[5, 5] = 600
        with pytest.raises(RuntimeError, match="outside"):
            ctx.apply_crf(labels, d[0], c[0], ang)
        ok = ctx.find_surfaces_batch(d, c)                 # the context still works
        assert ok["surfaces"].shape == (1, H, W)
    # a frame that the front end cuts into more than 510 patches: flat 24x24 cells, each at a depth of its own
    H, W = 960, 1280
    cell = (np.arange(-(-H // 24))[:, None] * 7 + np.arange(-(-W // 24))[None, :] * 13) % 40
    depth = np.repeat(np.repeat(1000 + 60 * cell, 24, 0), 24, 1)[:H, :W].astype(np.uint16)
    rgb = np.full((H, W, 3), 128, np.uint8)
    with Context(H, W, max_batch=1) as ctx:
        pat, _ = ctx.patches(ctx.smooth_depth(depth.astype(np.float32)))[:2]
        assert pat.max() > 510, pat.max()
        with pytest.raises(RuntimeError, match="at most 512"):
            ctx.find_surfaces_batch(depth[None], rgb[None])


@pytest.mark.parametrize("r,shape", [(10, (50, 70)), (15, (50, 70)), (15, (24, 40)), (4, (37, 90))])
def test_rolling_window_on_ragged_and_tiny_frames(oracle, r, shape):
    """Window radii other than 7 on frames whose sides are no multiples of the 32 x 16 tiles, and on a frame smaller than
    the 31 x 31 window itself (every region row of every tile then crosses the frame border)."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = shape
    d, c = make_batch([51, 52], H, W)
    d[1, : H // 2] = 0                                   # the upper half without depth: whole tiles are skipped
    with Context(H, W, max_batch=2, kernel_size=r) as ctx:
        got = ctx.find_surfaces_batch(d, c)
    for i in range(2):
        ref = oracle.find_surfaces(d[i], c[i], oracle.gauss_params(), r=r, want_q=True)
        assert got["num_surfaces"][i] == ref["num_surfaces"]
        assert np.array_equal(got["patches"][i], ref["patches"])
        assert _ties_only(got["surfaces"][i], ref["surfaces"], ref["q_final"]), (r, shape, i)
        print("label disagreement:", float((got["surfaces"][i] != ref["surfaces"]).mean()))
        assert (got["surfaces"][i] != ref["surfaces"]).mean() <= 0.003


def test_position_dependent_weights_with_a_larger_window(oracle):
    """The per-offset weight table in the rolling-window kernel (r = 10)."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    d, c = make_batch([61], 96, 128)
    p = oracle.gauss_params().copy()
    p[4:7] = [3.0, 5.0, 2.0]
    p[3] = 0.05
    with Context(96, 128, kernel_size=10, params=p) as ctx:
        got = ctx.find_surfaces_batch(d[0], c[0])
    ref = oracle.find_surfaces(d[0], c[0], p, r=10, want_q=True)
    assert np.array_equal(got["patches"], ref["patches"])
    assert _ties_only(got["surfaces"], ref["surfaces"], ref["q_final"])
    print("label disagreement:", float((got["surfaces"] != ref["surfaces"]).mean()))
    assert (got["surfaces"] != ref["surfaces"]).mean() <= 0.003


def test_pinned_blocks_are_reused():
    """Page-locked blocks whose arrays are gone serve the next request of a similar size (locking pages costs more than
    the stage takes for a frame); release_pinned_pool hands them back."""
    from imagesegmentation_amd import _lib
    _lib.release_pinned_pool()
    a = _lib.pinned_empty((480, 640, 3), np.float64)
    a[...] = 1.0
    addr = a.ctypes.data
    del a
    b = _lib.pinned_empty((480, 640, 3), np.float64)
    assert b.ctypes.data == addr
    c = _lib.pinned_empty((480, 640, 3), np.float64)           # b is still alive: a block of its own
    assert c.ctypes.data != addr
    c[...] = 2.0
    b[...] = 3.0
    assert float(c[0, 0, 0]) == 2.0
    del b, c
    assert _lib._POOL_BYTES == 2 * _lib._block_size(480 * 640 * 3 * 8)
    _lib.release_pinned_pool()
    assert _lib._POOL_BYTES == 0 and not _lib._POOL
    assert [_lib._block_size(n) >= n for n in (1, 4096, 4097, 10 ** 6, 10 ** 9)] == [True] * 5


def test_model_calls_from_two_threads():
    """find_surface_model()(x) and the statistics mirror from two threads at once: the module keeps a library context per
    thread (a context serves one call at a time), and each thread's results are those of the main thread's calls."""
    import threading
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import process_surfaces as ps
    frames = [fs.load_image(i) for i in (5, 64)]
    model = fs.find_surface_model()

    def run(frame):
        data = model(frame)
        ps.extract_information_from_surface_data_and_preprocess_surfaces(data, None)
        return data

    want = [run(f) for f in frames]
    got, errors = [None, None], []

    def work(k):
        try:
            for _ in range(4):
                got[k] = run(frames[k])
        except Exception as e:                      # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for g, w in zip(got, want):
        for k in ("surfaces", "patches", "depth_edges", "points_3d", "counts", "hist_color", "hist_angle", "depth_extend"):
            assert np.array_equal(g[k], w[k]), k
