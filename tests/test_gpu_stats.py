"""GPU parity of the per-surface statistics stage (fs_surface_stats) against the C oracle
(oracle/fs_oracle_stats.c).  Integer results -- labels, point lists, counts, centroid indices,
histogram counts, neighbour counts -- are compared bit for bit (the float64 arrays derived from
integers by one division are equal too); float64 sums of float32 values (average angles, texture)
within 1e-12 relative, since the order of additions differs."""
import numpy as np
import pytest

from conftest import DeviceArrays, osd_frame

pytestmark = pytest.mark.gpu


def _frame(oracle, index):
    from imagesegmentation_amd.utils import rgb_to_Lab
    d, c = osd_frame(index)
    ref = oracle.find_surfaces(d, c, oracle.gauss_params())
    rng = np.random.default_rng(index)
    tex = rng.random((120, 160, 256), dtype=np.float32)
    return ref, rgb_to_Lab(c), tex


@pytest.mark.parametrize("index", [57, 110])
def test_surface_stats_osd(oracle, index):
    from imagesegmentation_amd._lib import Context
    ref, lab, tex = _frame(oracle, index)
    S = ref["num_surfaces"]
    args = (ref["surfaces"], ref["patches"], ref["depth_edges"], ref["points_3d"], ref["angles"], lab, tex, S)
    want = oracle.surface_stats(*args)
    with Context(480, 640) as ctx:
        got = ctx.surface_stats(*args)
    exact = ("surfaces", "patch_point_offsets", "patch_points", "avg_pos", "counts", "centroids", "centroid_indices",
             "hist_color", "hist_angle", "neighbor_counts", "border_centers", "norm", "depth_extend")
    for k in exact:
        assert np.array_equal(got[k], want[k]), k
    for k in ("avg_angles", "hist_texture"):
        assert np.allclose(got[k], want[k], rtol=1e-12, atol=1e-13), k
    assert got["surfaces"].dtype == np.int64 and got["depth_extend"].dtype == np.float32
    assert (want["neighbor_counts"] > 3).any()          # the frame has real neighbours
    assert (got["surfaces"] != ref["surfaces"]).any() or index == 110


def test_surface_stats_disconnected_parts_and_edge_cases(oracle):
    """Hand-made labelling: a label split in two parts (the part away from the centroid goes), a label
    without patch pixels (dropped entirely), no texture input."""
    from imagesegmentation_amd._lib import Context
    H, W, S = 32, 48, 5
    surfaces = np.zeros((H, W), np.int64)
    patches = np.zeros((H, W), np.int64)
    surfaces[2:12, 2:20] = 1
    patches[4:10, 4:16] = 1
    surfaces[20:24, 30:34] = 1            # disconnected part of label 1
    surfaces[14:30, 4:20] = 2
    patches[16:28, 6:18] = 2
    surfaces[2:10, 30:44] = 3             # label 3 has no patch pixels at all
    edges = np.ones((H, W), np.int32)
    edges[12:14, :] = 0
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(H, W, 3)).astype(np.float32) * 100
    pts[0, :5] = 0                        # zero-norm pixels are skipped by the extent
    ang = (rng.random((H, W, 2), dtype=np.float32) - 0.5) * 6
    lab = np.stack([rng.random((H, W)) * 100, rng.random((H, W)) * 200 - 100, rng.random((H, W)) * 200 - 100], -1)
    args = (surfaces, patches, edges, pts, ang, lab, None, S)
    want = oracle.surface_stats(*args)
    with Context(H, W) as ctx:
        got = ctx.surface_stats(*args)
    assert (want["surfaces"][20:24, 30:34] == 0).all() and (want["surfaces"] != 3).all()
    assert tuple(want["centroid_indices"][3]) == (-1, -1) and tuple(want["centroid_indices"][4]) == (-1, -1)
    for k in ("surfaces", "patch_point_offsets", "patch_points", "avg_pos", "counts", "centroids", "centroid_indices",
              "hist_color", "hist_angle", "neighbor_counts", "border_centers", "norm", "depth_extend", "hist_texture"):
        assert np.array_equal(got[k], want[k]), k
    assert np.allclose(got["avg_angles"], want["avg_angles"], rtol=1e-12, atol=1e-13)
    assert np.isinf(got["depth_extend"][4]).all()
    # labels outside [0, S) are refused
    bad = surfaces.copy()
    bad[0, 0] = S
    with Context(H, W) as ctx:
        with pytest.raises(RuntimeError):
            ctx.surface_stats(bad, patches, edges, pts, ang, lab, None, S)


def test_process_surfaces_mirror_keys(oracle):
    """find_surface_model -> extract_information_from_surface_data_and_preprocess_surfaces, the
    sequence of the reference's prediction models, with the keys downstream code reads."""
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import process_surfaces as ps
    data = fs.find_surface_model("Gauss")(fs.load_image(20))
    before = {k: np.copy(data[k]) for k in ("surfaces", "patches", "depth_edges", "points_3d", "angles")}
    rng = np.random.default_rng(0)
    tex = rng.random((120, 160, 256), dtype=np.float32)
    ps.extract_information_from_surface_data_and_preprocess_surfaces(data, lambda rgb: tex)
    S = data["num_surfaces"]
    for k in ("patch_points", "surfaces", "avg_pos", "counts", "centroids", "texture", "lab", "hist_color", "hist_angle",
              "hist_texture", "avg_normals", "planes", "neighbors", "border_centers", "norm", "depth_extend",
              "depth_extend_distances", "depth_extend_distance_ratio"):
        assert k in data, k
    assert len(data["patch_points"]) == S and data["patch_points"][1].dtype == np.float32
    assert data["counts"].shape == (S, 1) and data["avg_pos"].shape == (S, 2) and data["centroids"].shape == (S, 3)
    assert data["hist_color"].shape == (256, 256, S) and data["hist_angle"].shape == (40, 40, S)
    assert data["avg_normals"].shape == (S, 3) and data["planes"].shape == (S,)
    assert data["depth_extend_distances"].shape == (S, S) and len(data["neighbors"]) == S
    want = oracle.surface_stats(before["surfaces"], before["patches"], before["depth_edges"], before["points_3d"],
                                before["angles"], data["lab"], tex, S)
    assert np.array_equal(data["surfaces"], want["surfaces"])
    assert np.array_equal(data["depth_extend"], want["depth_extend"])
    assert data["neighbors"] == ps.neighbor_lists(want["neighbor_counts"])
    assert np.allclose(np.linalg.norm(data["avg_normals"][1:], axis=-1), 1)


def test_surface_stats_device_pointers(oracle):
    """fs_surface_stats_dev on arrays resident in HBM: the outputs equal the host-pointer entry point's (same
    kernels; bit for bit except the two float64 sums whose order of additions is not fixed); outputs that are not
    asked for are skipped; a bad label is reported the same way."""
    from imagesegmentation_amd._lib import Context
    ref, lab, tex = _frame(oracle, 57)
    S = ref["num_surfaces"]
    host_in = (ref["surfaces"], ref["patches"], ref["depth_edges"], ref["points_3d"], ref["angles"], lab, tex)
    with Context(480, 640) as ctx:
        want = ctx.surface_stats(*host_in, S)
        mem = DeviceArrays()
        try:
            d_in = [mem.upload(a) for a in host_in]
            d_out = {k: mem.alloc(v.nbytes) for k, v in want.items()}
            ctx.surface_stats_dev(S, *d_in, texture_dim=tex.shape[2], **d_out)
            for k, v in want.items():
                got = mem.download(d_out[k], v)
                if k in ("avg_angles", "hist_texture"):      # float64 sums accumulated by atomics: the order varies
                    assert np.allclose(got, v, rtol=1e-12, atol=1e-13), k
                else:
                    assert np.array_equal(got, v), k
            # a subset of the outputs, no texture
            part = {k: mem.alloc(want[k].nbytes) for k in ("surfaces", "neighbor_counts", "depth_extend")}
            ctx.surface_stats_dev(S, *d_in[:6], **part)
            for k, p in part.items():
                assert np.array_equal(mem.download(p, want[k]), want[k]), k
            bad = ref["surfaces"].copy()
            bad[0, 0] = S
            with pytest.raises(RuntimeError):
                ctx.surface_stats_dev(S, mem.upload(bad), *d_in[1:6])
            with pytest.raises(RuntimeError, match="null input"):
                ctx.surface_stats_dev(S, None, *d_in[1:6])
            with pytest.raises(RuntimeError, match="num_surfaces"):
                ctx.surface_stats_dev(0, *d_in[:6])
        finally:
            mem.free()


def test_surface_stats_batch_dev_with_device_colour_conversion(oracle):
    """fs_surface_stats_batch_dev on four frames of different label counts, consuming device-resident arrays (as
    fs_find_surfaces_batch_dev leaves them) and uint8 colours: every frame's part of every output against the oracle
    run frame by frame on utils.rgb_to_Lab's colours.  The device's Lab image agrees with the numpy one to 1e-10;
    a colour bin could differ only for a pixel whose a or b + 127 lies within that distance of a half-integer."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.utils import rgb_to_Lab
    idx = [5, 57, 80, 110]
    frames = [osd_frame(i) for i in idx]
    d = np.stack([f[0] for f in frames]).astype(np.uint16)
    c = np.stack([f[1] for f in frames]).astype(np.uint8)
    n, H, W = d.shape
    T = 8
    rng = np.random.default_rng(1)
    tex = rng.random((n, H // 4, W // 4, T), dtype=np.float32)
    dev = DeviceArrays()
    try:
        with Context(H, W, max_batch=n) as ctx:
            fsr = ctx.find_surfaces_batch(d, c)
            ns = fsr["num_surfaces"]
            assert len(set(ns.tolist())) > 1
            el = [ctx.stats_output_elements(H, W, int(s), T) for s in ns]
            dtypes = {"surfaces": np.int64, "patch_point_offsets": np.int32, "patch_points": np.float32, "avg_pos": np.float64,
                      "counts": np.float64, "centroids": np.float32, "centroid_indices": np.int32, "hist_color": np.float64,
                      "hist_angle": np.float64, "hist_texture": np.float64, "avg_angles": np.float64, "neighbor_counts": np.float64,
                      "border_centers": np.float64, "norm": np.float32, "depth_extend": np.float32}
            total = {k: sum(e[k] for e in el) for k in dtypes}
            ptr = {k: dev.alloc(total[k] * np.dtype(dt).itemsize) for k, dt in dtypes.items()}
            d_lab = dev.alloc(n * H * W * 24)
            ctx.surface_stats_batch_dev(ns, dev.upload(fsr["surfaces"]), dev.upload(fsr["patches"]), dev.upload(fsr["depth_edges"]),
                                        dev.upload(fsr["points_3d"]), dev.upload(fsr["angles"]), d_rgb=dev.upload(c),
                                        d_texture=dev.upload(tex), texture_dim=T, d_lab_out=d_lab, **ptr)
            got = {k: dev.download(ptr[k], np.empty(total[k], dt)) for k, dt in dtypes.items()}
            lab = dev.download(d_lab, np.empty((n, H, W, 3), np.float64))
    finally:
        dev.free()
    exact = ("surfaces", "patch_point_offsets", "patch_points", "avg_pos", "counts", "centroids", "centroid_indices",
             "hist_color", "hist_angle", "neighbor_counts", "border_centers", "norm", "depth_extend")
    start = {k: 0 for k in dtypes}
    for i in range(n):
        lab_ref = rgb_to_Lab(c[i])
        assert np.abs(lab[i] - lab_ref).max() <= 1e-10
        want = oracle.surface_stats(fsr["surfaces"][i], fsr["patches"][i], fsr["depth_edges"][i], fsr["points_3d"][i], fsr["angles"][i],
                                    lab_ref, tex[i], int(ns[i]))
        for k in dtypes:
            part = got[k][start[k]:start[k] + el[i][k]]
            start[k] += el[i][k]
            ref = np.asarray(want[k]).reshape(-1)
            if k in exact:
                assert np.array_equal(part, ref), (i, k)
            else:
                assert np.allclose(part, ref, rtol=1e-12, atol=1e-13), (i, k)


def test_process_surfaces_mirror_converts_colours_on_the_device(oracle):
    """extract_information_from_surface_data_and_preprocess_surfaces without a "lab" entry: the colour conversion runs on
    the GPU and the dict receives data["lab"]; same statistics as with the numpy conversion."""
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import process_surfaces as ps
    from imagesegmentation_amd.utils import rgb_to_Lab
    d, c = osd_frame(20)
    model = fs.find_surface_model("Gauss")
    a = model((d, c, None))
    b = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in a.items()}
    b["lab"] = rgb_to_Lab(c)
    ps.extract_information_from_surface_data_and_preprocess_surfaces(a, None)
    ps.extract_information_from_surface_data_and_preprocess_surfaces(b, None)
    assert np.abs(a["lab"] - b["lab"]).max() <= 1e-10
    for k in ("surfaces", "avg_pos", "counts", "centroids", "hist_color", "hist_angle", "border_centers", "depth_extend", "planes",
              "depth_extend_distances"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert a["neighbors"] == b["neighbors"]


@pytest.mark.parametrize("shape", [(40, 100), (37, 64), (21, 191)])
def test_surface_stats_rows_that_end_inside_a_wave(oracle, shape):
    """Frame widths that are not a multiple of 64: the lanes of a wave then lie in two rows, which the per-label position
    sums of the preparation pass have to take into account.  Random blocky labelling, every output against the oracle."""
    from imagesegmentation_amd._lib import Context
    H, W = shape
    S = 7
    rng = np.random.default_rng(H * 1000 + W)
    cells = rng.integers(0, S, size=(H // 9 + 1, W // 11 + 1))
    i, j = np.mgrid[0:H, 0:W]
    surfaces = cells[i // 9, j // 11].astype(np.int64)
    patches = np.where((i % 9 > 1) & (j % 11 > 2) & (rng.random((H, W)) < 0.8), surfaces, 0).astype(np.int64)
    edges = (rng.random((H, W)) < 0.9).astype(np.int32)
    pts = rng.normal(size=(H, W, 3)).astype(np.float32) * 100
    ang = (rng.random((H, W, 2), dtype=np.float32) - 0.5) * 6
    lab = np.stack([rng.random((H, W)) * 100, rng.random((H, W)) * 200 - 100, rng.random((H, W)) * 200 - 100], -1)
    args = (surfaces, patches, edges, pts, ang, lab, None, S)
    want = oracle.surface_stats(*args)
    with Context(H, W) as ctx:
        got = ctx.surface_stats(*args)
    for k in ("surfaces", "patch_point_offsets", "patch_points", "avg_pos", "counts", "centroids", "centroid_indices",
              "hist_color", "hist_angle", "neighbor_counts", "border_centers", "norm", "depth_extend"):
        assert np.array_equal(got[k], want[k]), k
    assert np.allclose(got["avg_angles"], want["avg_angles"], rtol=1e-12, atol=1e-13)


def test_colour_bins_of_all_sRGB_colours():
    """Every one of the 2^24 colours through the device's colour conversion, as 64 frames of 512 x 512 pixels with one
    label: the colour histogram must hold exactly the bins that utils.rgb_to_Lab's float64 values round to.  Without a
    Lab output the kernel takes the bins from a float32 evaluation unless a value lies close to a rounding boundary;
    with one it evaluates in float64 throughout.  Both ways are enumerated."""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.utils import rgb_to_Lab
    H = W = 512
    n, S = 32, 2                  # (a batch this large also takes the accumulation pass's fat blocks with colour windows in LDS)
    N = H * W
    dev = DeviceArrays()
    try:
        with Context(H, W, max_batch=n) as ctx:
            ones64 = dev.upload(np.ones((n, H, W), np.int64))
            edges = dev.upload(np.zeros((n, H, W), np.int32))
            pts = dev.upload(np.ones((n, H, W, 3), np.float32))
            ang = dev.upload(np.zeros((n, H, W, 2), np.float32))
            d_hist = dev.alloc(n * 65536 * S * 8)
            d_lab = dev.alloc(n * N * 24)
            for call in range(2 ** 24 // (n * N)):
                v = np.arange(call * n * N, (call + 1) * n * N, dtype=np.uint32)
                rgb = np.stack([v >> 16, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(n, H, W, 3)
                lab = rgb_to_Lab(rgb)
                bins = (np.rint(lab[..., 1] + 127.0).astype(np.int64) * 256 + np.rint(lab[..., 2] + 127.0).astype(np.int64)).reshape(n, N)
                want = np.stack([np.bincount(b, minlength=65536) for b in bins]).astype(np.float64)
                d_rgb = dev.upload(rgb)
                for lab_out in (None, d_lab):
                    ctx.surface_stats_batch_dev([S] * n, ones64, ones64, edges, pts, ang, d_rgb=d_rgb, d_lab_out=lab_out, hist_color=d_hist)
                    ctx.synchronize()
                    got = np.rint(dev.download(d_hist, np.empty((n, 65536, S), np.float64)) * (N / 10))      # counts over a tenth of the label's pixels
                    assert not got[:, :, 0].any()
                    assert np.array_equal(got[:, :, 1], want), (call, lab_out is not None)
    finally:
        dev.free()


def test_colour_bins_of_frames_with_ragged_sizes():
    """Frames whose pixel count is not a multiple of four (the bins-only colour pass takes four pixels per lane and the
    second frame's colours do not start on a word boundary): same colour histograms as with a Lab image as an output,
    which takes the one-pixel-per-lane float64 pass."""
    from imagesegmentation_amd._lib import Context
    H, W, n, S = 21, 191, 3, 5
    N = H * W
    rng = np.random.default_rng(7)
    surfaces = rng.integers(1, S, size=(n, H // 7 + 1, W // 16 + 1))[:, np.arange(H)[:, None] // 7, np.arange(W)[None, :] // 16].astype(np.int64)
    rgb = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
    dev = DeviceArrays()
    try:
        with Context(H, W, max_batch=n) as ctx:
            args = (dev.upload(surfaces), dev.upload(surfaces), dev.upload(np.zeros((n, H, W), np.int32)),
                    dev.upload(np.ones((n, H, W, 3), np.float32)), dev.upload(np.zeros((n, H, W, 2), np.float32)))
            d_rgb, d_hist, d_lab = dev.upload(rgb), dev.alloc(n * 65536 * S * 8), dev.alloc(n * N * 24)
            hists = []
            for lab_out in (None, d_lab):
                ctx.surface_stats_batch_dev([S] * n, *args, d_rgb=d_rgb, d_lab_out=lab_out, hist_color=d_hist)
                ctx.synchronize()
                hists.append(dev.download(d_hist, np.empty((n, 65536, S), np.float64)))
    finally:
        dev.free()
    assert hists[0].any() and np.array_equal(hists[0], hists[1])


def test_process_surfaces_mirror_over_a_list_of_frames():
    """extract_information_for_frames: the per-frame function from several threads, a library context per thread; every
    frame's results are those of calling the per-frame function on it alone."""
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import process_surfaces as ps
    model = fs.find_surface_model("Gauss")
    base = [model(fs.load_image(i)) for i in (3, 20, 57, 88, 110)]
    alone = [ps.extract_information_from_surface_data_and_preprocess_surfaces(dict(d), None) for d in base]
    for workers in (1, 3, 3):
        frames = ps.extract_information_for_frames([dict(d) for d in base], None, workers=workers)
        for got, want in zip(frames, alone):
            assert got.keys() == want.keys()
            for k in ("surfaces", "lab", "counts", "avg_pos", "centroids", "hist_color", "hist_angle", "norm", "depth_extend",
                      "planes", "border_centers"):
                assert np.array_equal(got[k], want[k]), (workers, k)
            assert got["neighbors"] == want["neighbors"]
            assert np.allclose(got["avg_normals"], want["avg_normals"], rtol=0, atol=1e-12, equal_nan=True)
            assert all(np.array_equal(a, b) for a, b in zip(got["patch_points"], want["patch_points"]))


def test_large_batches_count_colours_like_small_ones():
    """A batch of 32 frames makes the accumulation pass use blocks of 2 048 pixels that gather their colour counts in
    windows in LDS; a batch of four uses small blocks that add to the frame's histogram directly (the path the
    oracle comparisons above go through).  Same frames, same histograms."""
    from imagesegmentation_amd._lib import Context
    idx = [5, 57, 80, 110]
    frames = [osd_frame(i) for i in idx]
    d = np.stack([f[0] for f in frames]).astype(np.uint16)
    c = np.stack([f[1] for f in frames]).astype(np.uint8)
    H, W = d.shape[1:]
    rep = 8
    dev = DeviceArrays()
    try:
        with Context(H, W, max_batch=len(idx) * rep) as ctx:
            fsr = ctx.find_surfaces_batch(d, c)
            ns = [int(x) for x in fsr["num_surfaces"]]
            hist = {}
            for r in (1, rep):
                tile = lambda a: np.concatenate([a] * r)
                d_hist = dev.alloc(r * sum(ns) * 65536 * 8)
                d_angle = dev.alloc(r * sum(ns) * 1600 * 8)
                ctx.surface_stats_batch_dev(ns * r, dev.upload(tile(fsr["surfaces"])), dev.upload(tile(fsr["patches"])),
                                            dev.upload(tile(fsr["depth_edges"])), dev.upload(tile(fsr["points_3d"])),
                                            dev.upload(tile(fsr["angles"])), d_rgb=dev.upload(tile(c)), hist_color=d_hist, hist_angle=d_angle)
                ctx.synchronize()
                hist[r] = (dev.download(d_hist, np.empty(r * sum(ns) * 65536, np.float64)),
                           dev.download(d_angle, np.empty(r * sum(ns) * 1600, np.float64)))
    finally:
        dev.free()
    for k in range(2):
        assert hist[1][k].any()
        assert np.array_equal(np.concatenate([hist[1][k]] * rep), hist[rep][k])
