"""CPU tests of the host-side mirror (no GPU, no compute through the library): tiling glue,
potentials/initial probabilities, loaders, the out/{index}/*.npy layout, sharding and the
C ABI surface."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT


# ------------------------------------------------------------------- tiling glue
def test_get_inputs_and_assemble_outputs_round_trip():
    from imagesegmentation_amd import find_surfaces as fs
    rng = np.random.default_rng(0)
    H, W, S, r = 24, 32, 5, 3
    feats = [rng.random((H, W, c)) for c in (3, 6, 5)]
    unary, Q = rng.random((H, W, S)), rng.random((H, W, S))
    inputs = fs.get_inputs(feats, unary, Q, r, 4, 4, W // 4, H // 4)
    un, Qin, f1, f2, f3, val = inputs
    assert un.shape == (16, 6, 8, S) and Qin.shape == (16, 6 + 2 * r, 8 + 2 * r, S)
    assert f1.shape[-1] == 3 and f2.shape[-1] == 6 and f3.shape[-1] == 5
    assert val.shape == (16, 1, 1, 1, S) and (val[..., -1] == 1).all() and val.sum() == 16
    # tile 1 is the second tile down the first column (x outer, y inner)
    assert np.array_equal(un[1], unary[6:12, 0:8])
    # halo of the first tile is zero padding
    assert (Qin[0][:r] == 0).all() and (Qin[0][:, :r] == 0).all()
    assert np.array_equal(Qin[0][r:, r:], Q[:6 + r, :8 + r])
    # cutting the unpadded centre of every tile and stitching gives the frame back
    back = fs.assemble_outputs(Qin[:, r:-r, r:-r], 4, 4, W // 4, H // 4, H, W)
    assert np.array_equal(back, Q)


def test_host_glue_agrees_with_oracle_glue():
    from imagesegmentation_amd import find_surfaces as fs
    from oracle import oracle_np as N
    rng = np.random.default_rng(1)
    H, W, r = 16, 24, 2
    patches = rng.integers(0, 4, (H, W))
    ns = int(patches.max()) + 1
    u1, q1, p1 = fs.get_unary_potentials_and_initial_probabilities({"patches": patches, "num_surfaces": ns})
    u2, q2, p2 = N.get_unary_potentials_and_initial_probabilities(patches, ns)
    assert np.array_equal(u1, u2) and np.array_equal(q1, q2) and np.array_equal(p1, p2)
    feats = [rng.random((H, W, c)) for c in (3, 6, 5)]
    a = fs.get_inputs(feats, u1, q1, r, 2, 2, W // 2, H // 2)
    b = N.get_inputs(feats, u2, q2, r, 2, 2, W // 2, H // 2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_unary_known_answer():
    from imagesegmentation_amd import find_surfaces as fs
    d = {"patches": np.array([[0, 2], [1, 0]]), "num_surfaces": 3}
    unary, q0, prob = fs.get_unary_potentials_and_initial_probabilities(d)
    assert unary.dtype == np.float64 and unary.shape == (2, 2, 4) and prob.shape == (1, 1, 4)
    assert np.array_equal(unary[0, 1], [10, 1, 0, 1]) and np.array_equal(unary[0, 0], [10, 1, 1, 1])
    assert np.allclose(q0[1, 0], [0, 0.9, 0.05, 0.05]) and np.allclose(q0[1, 1], [0, 1 / 3, 1 / 3, 1 / 3])


def test_extract_features_layout():
    from imagesegmentation_amd import find_surfaces as fs
    H, W = 4, 6
    grid = fs._grid(H, W)
    assert grid[2, 5, 0] == 2 and grid[2, 5, 1] == 5
    depth = np.arange(H * W).reshape(H, W)
    f1, f2, f3 = fs.extract_features(depth, np.ones((H, W, 3)), np.zeros((H, W, 2)), grid)
    assert f1.shape == (H, W, 3) and f2.shape == (H, W, 6) and f3.shape == (H, W, 5)
    assert f1[3, 4, 2] == depth[3, 4] and f2[0, 0, 3] == 1


def test_load_gauss_parameters_shape():
    from imagesegmentation_amd import find_surfaces as fs
    p = fs.load_Gauss_parameters()
    assert len(p) == 13 and all(v.shape == (1, 1) for v in p)
    assert p[3][0, 0] == 0 and abs(p[12][0, 0] - 0.01) < 1e-9
    crf = fs.conv_crf_Gauss(*p, 7, 120, 160)
    assert crf.params.shape == (13,) and crf.kernel_size == 7
    # logistic-regression variant: clf.coef_ [1,4], clf.intercept_ + 4 and LR_CRF_parameters
    coef, bias = fs.load_pixel_similarity_parameters()
    assert coef.shape == (1, 4) and bias.shape == (1,)
    assert np.allclose(coef, [[-3.4754366, -1.1483647, -4.0971241, 1.3293620]]) and np.isclose(bias[0], 2.8543673 + 4)
    lr = fs.get_GPU_models("LR")[3]
    assert isinstance(lr, fs.conv_crf_LR) and lr.params.shape == (7,) and lr.params[6] == 15000


# ------------------------------------------------------------------- loaders and file layout
def test_sorted_alphanumeric_and_load_image():
    from imagesegmentation_amd import load_images as li
    assert li.sorted_alphanumeric(["10.png", "9.png", "0.png", "100.png"]) == ["0.png", "9.png", "10.png", "100.png"]
    depth, rgb, ann = li.load_image(3)
    assert depth.shape == (480, 640) and depth.dtype == np.uint16
    assert rgb.shape == (480, 640, 3) and rgb.dtype == np.uint8 and ann.shape == (480, 640)


def test_npy_layout_round_trip(tmp_path, monkeypatch):
    from imagesegmentation_amd import load_images as li
    monkeypatch.setattr(li, "out_root", str(tmp_path / "out"))
    depth, rgb, ann = li.load_image(0)
    rng = np.random.default_rng(2)
    data = {"depth": depth, "rgb": rgb, "annotation": ann,
            "angles": rng.random((480, 640, 2), np.float32), "vectors": rng.random((480, 640, 3), np.float32),
            "points_3d": rng.random((480, 640, 3), np.float32), "patches": rng.integers(0, 5, (480, 640)),
            "depth_edges": rng.integers(0, 2, (480, 640)).astype(np.int32), "surfaces": rng.integers(0, 7, (480, 640))}
    li.save_surface_information(0, data)
    files = sorted(os.listdir(tmp_path / "out" / "0"))
    assert files == ["Q.npy", "angles.npy", "depth.npy", "edges.npy", "patches.npy", "points.npy", "vectors.npy"]
    back = li.load_image_and_surface_information(0)
    for k in ("angles", "vectors", "points_3d", "patches", "depth_edges", "surfaces", "depth"):
        assert np.array_equal(back[k], data[k]) and back[k].dtype == data[k].dtype, k
    assert back["num_surfaces"] == int(data["surfaces"].max()) + 1
    assert back["lab"].shape == (480, 640, 3)
    with pytest.raises(SystemExit):
        li.load_image_and_surface_information(1)       # nothing saved for index 1


def test_rgb_to_lab_known_colours():
    from imagesegmentation_amd.utils import rgb_to_Lab
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 0, 255]]], np.uint8)
    lab = rgb_to_Lab(px)[0]
    assert np.allclose(lab[0], [100, 0, 0], atol=0.02) and np.allclose(lab[1], [0, 0, 0], atol=1e-9)
    assert np.allclose(lab[2], [53.24, 80.09, 67.20], atol=0.02)      # CIE L*a*b* of sRGB red under D65
    assert np.allclose(lab[3], [32.30, 79.19, -107.86], atol=0.02)    # ... and of sRGB blue


def test_angles_to_normals():
    from imagesegmentation_amd.utils import angles_to_normals
    n = angles_to_normals(np.array([[0.0, 0.0], [np.pi / 2, 0.0], [0.0, np.pi / 2]]))
    assert np.allclose(n, [[0, 0, -1], [0, 1, 0], [-1, 0, 0]], atol=1e-12)


# ------------------------------------------------------------------- sharding
def test_shard_is_a_partition():
    from imagesegmentation_amd.sharding import shard
    idx = list(range(111))
    for world in (1, 2, 4, 8):
        parts = [shard(idx, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == idx
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
        assert all(i % world == r for r, p in enumerate(parts) for i in p)
    with pytest.raises(ValueError):
        shard(idx, 2, 2)


# ------------------------------------------------------------------- C ABI surface
def _declared_functions():
    src = open(os.path.join(ROOT, "include", "fs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"\b(fs_[a-z_]+)\s*\(", src)


def test_numa_pinning_reads_the_gpu_node_from_sysfs(tmp_path, monkeypatch):
    """sharding.pin_to_gpu_numa_node on a made-up sysfs tree: two CPU nodes and two GPUs in the KFD topology, the second GPU on
    NUMA node 1; device 1 is pinned to the CPUs of node 1 that the process may use, an unknown device changes nothing."""
    from imagesegmentation_amd import sharding
    nodes = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for i, (simd, loc) in enumerate([(0, 0), (0, 0), (1024, 0x0500), (1024, 0x8500)]):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    for bdf, node in (("0000:05:00.0", 0), ("0000:85:00.0", 1)):
        (tmp_path / "bus" / "pci" / "devices" / bdf).mkdir(parents=True)
        (tmp_path / "bus" / "pci" / "devices" / bdf / "numa_node").write_text(f"{node}\n")
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    for node, cpus in ((0, allowed[:half]), (1, allowed[half:] or allowed[:1])):
        (tmp_path / "devices" / "system" / "node" / f"node{node}").mkdir(parents=True)
        (tmp_path / "devices" / "system" / "node" / f"node{node}" / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "FS_NO_NUMA_PINNING"):
        monkeypatch.delenv(var, raising=False)
    assert sharding.gpu_numa_node(0, str(tmp_path)) == 0 and sharding.gpu_numa_node(1, str(tmp_path)) == 1
    assert sharding.gpu_numa_node(2, str(tmp_path)) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert sharding.gpu_numa_node(0, str(tmp_path)) == 1
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    try:
        assert sharding.pin_to_gpu_numa_node(5, str(tmp_path))["pinned"] is False and os.sched_getaffinity(0) == set(allowed)
        res = sharding.pin_to_gpu_numa_node(1, str(tmp_path))
        assert res["pinned"] and res["numa_node"] == 1 and os.sched_getaffinity(0) == set(allowed[half:] or allowed[:1])
    finally:
        os.sched_setaffinity(0, allowed)
    assert sharding._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}


def test_library_exports_every_declared_symbol():
    from imagesegmentation_amd import _lib
    names = _declared_functions()
    assert len(names) >= 18 and len(set(names)) == len(names)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fs_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes prototypes and header disagree"


def test_no_cpu_fallback_without_gpu():
    """Without a usable GPU the library must refuse to create a context (and say why)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from imagesegmentation_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.Context(480, 640)
    from imagesegmentation_amd import find_surfaces as fs
    with pytest.raises(RuntimeError):
        fs.find_surface_model("Gauss")((np.zeros((480, 640), np.uint16), np.zeros((480, 640, 3), np.uint8), None))


def test_product_code_does_not_touch_the_oracle():
    """Only tests, smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "imagesegmentation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{os.path.join(dirpath, f)} mentions the oracle"


def test_mosaic_generator_reproduces_the_golden_frames():
    """The 1280x960 fixtures hold only the oracle's results; the frames come from the seeded generator, whose output
    must therefore never change (CRC of the depth image stored with every case)."""
    import zlib
    from imagesegmentation_amd.synth import make_mosaic_scene
    g = np.load(os.path.join(ROOT, "tests", "golden", "synth1280.npz"))
    for k, (seed, n_patches, r) in enumerate(g["cases"]):
        d, c = make_mosaic_scene(int(seed), 960, 1280, int(n_patches))
        assert d.dtype == np.uint16 and c.dtype == np.uint8 and c.shape == (960, 1280, 3)
        assert zlib.crc32(d.tobytes()) == g[f"crc_depth_{k}"], k
        assert 0.15 < (d == 0).mean() < 0.4


def test_bench_executed_work_model():
    """bench.executed_work prices what the CRF update carried from the per-tile label counts: channels = listed + label 0
    + "unassigned" (+ one representative while unlisted labels remain), chunks of eight, tiles without depth skipped."""
    sys.path.insert(0, ROOT)
    import bench

    class FakeCtx:
        def tile_labels(self, n):
            listed = np.zeros((1, 2, 2), np.int32)
            listed[0, 0, 0], listed[0, 0, 1], listed[0, 1, 0], listed[0, 1, 1] = 1, 6, 14, 3
            return listed, (16, 32)

    depth = np.ones((1, 32, 64), np.uint16)
    depth[0, 16:, 32:] = 0                                   # the fourth tile has no valid depth
    flops, q_bytes, info = bench.executed_work(FakeCtx(), depth, np.array([20]), 7)
    # tiles: CE = 4 (1 walk, 4 slots), 9 and 17; S = 20 leaves unlisted labels.  The fullest tile lists more than 13 labels, so
    # the tiles that list 6 or more take 16 channels per walk: 9 -> 1 walk with 12 slots, 17 -> 2 walks with 16 + 12 slots
    sf = bench.SIM_FLOPS                                     # flops of one similarity (3 exponentials + 30)
    want = 512 * 224 * ((sf * 1 + 2 * 4) + (sf * 1 + 2 * 12) + (sf * 2 + 2 * 28))
    assert flops == want
    # another radius: 8 channels per walk throughout (9 -> 2 walks with 8 + 4 slots, 17 -> 3 walks with 8 + 8 + 4)
    flops5, _, _ = bench.executed_work(FakeCtx(), depth, np.array([20]), 5)
    assert flops5 == 512 * 120 * ((sf * 1 + 2 * 4) + (sf * 2 + 2 * 12) + (sf * 3 + 2 * 20))
    assert q_bytes == 512 * 4 * 2 * (4 + 9 + 17) + 32 * 64 * 28.0
    assert info["tiles_skipped_frac"] == 0.25 and info["channels_per_tile_max"] == 17
    assert info["channels_per_tile_p50_p90_p99"] == [9, 15, 16]


def _write_png(path, img, filters):
    """A PNG with the given filter type per row (0 None, 1 Sub, 2 Up), written by hand, its stream split over two IDAT chunks."""
    import struct
    import zlib
    h, w = img.shape[:2]
    channels = 1 if img.ndim == 2 else img.shape[2]
    bits = 8 * img.dtype.itemsize
    raw = img.astype(">u2").view(np.uint8).reshape(h, -1) if bits == 16 else img.reshape(h, -1)
    bpp = channels * bits // 8
    rows = []
    for r in range(h):
        cur = raw[r].astype(np.int32)
        if filters[r] == 1:
            out = cur - np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        elif filters[r] == 2:
            out = cur - (raw[r - 1].astype(np.int32) if r else 0)
        else:
            out = cur
        rows.append(bytes([filters[r]]) + (out % 256).astype(np.uint8).tobytes())
    stream = zlib.compress(b"".join(rows))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, 2 if channels == 3 else 0, 0, 0, 0)) +
                chunk(b"IDAT", stream[:len(stream) // 2]) + chunk(b"IDAT", stream[len(stream) // 2:]) + chunk(b"IEND", b""))


def test_png_reader_equals_pil(tmp_path):
    """load_image's own PNG decoder (zlib + numpy, for loader threads that scale) gives PIL's arrays: on every file of the
    data set, on hand-written files with None / Sub / Up rows, and through the PIL fall-back on files it does not take."""
    from PIL import Image
    from imagesegmentation_amd import load_images as L
    root = os.path.join(ROOT, "data", "osd")
    if os.path.isdir(os.path.join(root, "disparity")):
        for sub in ("disparity", "RGB", "annotation"):
            for name in sorted(os.listdir(os.path.join(root, sub))):
                path = os.path.join(root, sub, name)
                mine, ref = L._png_plain(path), np.asarray(Image.open(path))
                assert mine is not None and mine.dtype == ref.dtype and np.array_equal(mine, ref), path
    rng = np.random.default_rng(5)
    for k, (shape, dt) in enumerate((((37, 53), np.uint8), ((37, 53), np.uint16), ((20, 31, 3), np.uint8))):
        img = rng.integers(0, np.iinfo(dt).max, shape, endpoint=True).astype(dt)
        path = str(tmp_path / f"hand{k}.png")
        _write_png(path, img, [int(v) for v in rng.integers(0, 3, shape[0])])
        assert np.array_equal(np.asarray(Image.open(path)), img), "the hand-written file is not what was meant"
        mine = L._png_plain(path)
        assert mine is not None and mine.dtype == img.dtype and np.array_equal(mine, img), (shape, dt)
    # PIL's encoder picks Paeth / Average rows for a noisy image, and palette images are another colour type: not taken,
    # read through PIL all the same
    noisy = rng.integers(0, 255, (40, 40, 3), endpoint=True).astype(np.uint8)
    smooth = (np.add.outer(np.arange(40), np.arange(40))[..., None] * np.array([1, 2, 3])).astype(np.uint8)
    for k, arr in enumerate((noisy, smooth)):
        path = str(tmp_path / f"pil{k}.png")
        Image.fromarray(arr).save(path)
        assert np.array_equal(L.read_png(path), arr)
    Image.fromarray(noisy[..., 0]).convert("P").save(str(tmp_path / "pal.png"))
    assert L._png_plain(str(tmp_path / "pal.png")) is None
    assert np.array_equal(L.read_png(str(tmp_path / "pal.png")), np.asarray(Image.open(str(tmp_path / "pal.png"))))
    # a damaged stream and 16-bit colour are left to PIL as well
    good = open(str(tmp_path / "hand0.png"), "rb").read()
    cut = good.index(b"IDAT") + 4
    open(str(tmp_path / "broken.png"), "wb").write(good[:cut] + bytes(len(good) - cut))
    assert L._png_plain(str(tmp_path / "broken.png")) is None
    _write_png(str(tmp_path / "rgb16.png"), rng.integers(0, 65535, (9, 11, 3), endpoint=True).astype(np.uint16), [0] * 9)
    assert L._png_plain(str(tmp_path / "rgb16.png")) is None


def test_main_entry_point_runs_the_whole_data_set(monkeypatch):
    """main.find_surfaces_for_dataset (main.py:25-26): every index of the data set, results saved, nothing plotted."""
    from imagesegmentation_amd import find_surfaces, main
    calls = []
    monkeypatch.setattr(find_surfaces, "find_surfaces_for_indices", lambda indices, **kw: calls.append((list(indices), kw)))
    main.find_surfaces_for_dataset()
    assert calls == [(list(range(111)), {"save_data": True, "plot_result": False})]
