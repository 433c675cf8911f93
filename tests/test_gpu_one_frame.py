"""One frame per call: the mean-field updates are enqueued before the label count has come back to the host, which
receives it through page-locked memory that a kernel writes.  Whatever happens to the guess, the results must be
those of a context that waits for the count first (FS_NO_SPECULATION=1), bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("patches", "depth_edges", "surfaces", "angles", "vectors", "points_3d")


def _plain_context(H, W, **kw):
    from imagesegmentation_amd._lib import Context
    old = {k: os.environ.get(k) for k in ("FS_NO_SPECULATION",)}
    os.environ["FS_NO_SPECULATION"] = "1"
    try:
        return Context(H, W, **kw)          # the switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _checker_scene(H, W, cell=20):
    """10 x 12 flat facets of 20 x 20 pixels side by side at well separated depths, nothing around them: about 120
    surfaces, and a tile of the CRF update (32 x 16 pixels, labels within 21 pixels count) sees 16 of them."""
    rng = np.random.Generator(np.random.PCG64(0))
    i, j = np.mgrid[0:H, 0:W]
    ci, cj = i // cell, j // cell
    d = (900 + 260 * ((3 * ci + 5 * cj) % 7)).astype(np.uint16)
    rgb = rng.integers(30, 226, size=(H // cell + 1, W // cell + 1, 3))[ci, cj].astype(np.uint8)
    keep = np.zeros((H, W), bool)
    keep[100:300, 100:340] = True
    d[~keep] = 0
    return d, rgb


def _same(a, b, what):
    assert np.array_equal(a["num_surfaces"], b["num_surfaces"]), what
    for k in KEYS:
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(np.asarray(a[k]) != np.asarray(b[k]))
            print("DIFF", what, k, "count", len(bad), "min", bad.min(0), "max", bad.max(0), "first", bad[:5].tolist(),
                  "a", np.asarray(a[k])[tuple(bad[0])], "b", np.asarray(b[k])[tuple(bad[0])], flush=True)
        assert np.array_equal(a[k], b[k]), f"{what}: {k}"


def test_guess_holds_fails_and_recovers():
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_mosaic_scene, make_scene
    H, W = 480, 640
    few = make_scene(5, H, W)                              # a couple of dozen surfaces
    many = make_mosaic_scene(3, H, W, n_patches=100)       # more label channels than a first call allows for
    dense = _checker_scene(H, W)                           # small facets: a tile lists more labels than 16 channels carry
    with Context(H, W, max_batch=1) as ctx, _plain_context(H, W, max_batch=1) as plain:
        assert plain.speculation_counts() == (0, 0)
        ref_few, ref_many, ref_dense = (plain.find_surfaces_batch(d, c) for d, c in (few, many, dense))
        assert plain.speculation_counts() == (0, 0)
        ns_few, ns_many, ns_dense = (int(np.ravel(r["num_surfaces"])[0]) for r in (ref_few, ref_many, ref_dense))
        assert ns_few + 1 <= 64 < ns_many + 1 and ns_many < ns_dense < 200, (ns_few, ns_many, ns_dense)
        listed, _ = plain.tile_labels(1)
        assert listed.max() + 3 > 16, "the dense frame is meant to exceed 16 channels in some tile"

        _same(ctx.find_surfaces_batch(*few), ref_few, "first call")
        assert ctx.speculation_counts() == (1, 0)
        _same(ctx.find_surfaces_batch(*many), ref_many, "more channels than the Q planes hold")
        assert ctx.speculation_counts() == (1, 1)
        _same(ctx.find_surfaces_batch(*many), ref_many, "the planes have grown")
        assert ctx.speculation_counts() == (2, 1)
        # (a buffer that grows takes a quarter more than asked: 1.25 x 104 planes hold this frame's 122 channels)
        _same(ctx.find_surfaces_batch(*dense), ref_dense, "too many channels in a tile")
        assert ctx.speculation_counts() == (2, 2)
        _same(ctx.find_surfaces_batch(*few), ref_few, "back to a frame that fits")
        assert ctx.speculation_counts() == (3, 2)


def test_one_frame_calls_on_the_default_stream():
    """Device pointers, HIP's legacy default stream, calls back to back without synchronisation in between."""
    from conftest import DeviceArrays
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W, n = 96, 128, 4
    d, c = make_batch([41, 42, 43, 44], H, W)
    dev = DeviceArrays()
    try:
        with Context(H, W, max_batch=n) as ctx:
            ref = ctx.find_surfaces_batch(d, c)
            ctx.set_stream(None)
            outs = {"patches": (np.int64, ()), "depth_edges": (np.int32, ()), "surfaces": (np.int64, ()),
                    "angles": (np.float32, (2,)), "vectors": (np.float32, (3,)), "points_3d": (np.float32, (3,))}
            size = {k: H * W * int(np.prod(t, dtype=np.int64)) * np.dtype(dt).itemsize for k, (dt, t) in outs.items()}
            d_depth, d_rgb = dev.upload(d), dev.upload(c)
            ptr = {k: dev.alloc(n * size[k]) for k in outs}
            ns = []
            for i in range(n):
                ns.append(ctx.find_surfaces_batch_dev(1, d_depth + i * H * W * 2, d_rgb + i * H * W * 3,
                                                      d_angles=ptr["angles"] + i * size["angles"],
                                                      d_vectors=ptr["vectors"] + i * size["vectors"],
                                                      d_points=ptr["points_3d"] + i * size["points_3d"],
                                                      d_patches=ptr["patches"] + i * size["patches"],
                                                      d_edges=ptr["depth_edges"] + i * size["depth_edges"],
                                                      d_surfaces=ptr["surfaces"] + i * size["surfaces"])[0])
            for k, (dt, t) in outs.items():
                got = dev.download(ptr[k], np.empty((n, H, W) + t, dt))
                assert np.array_equal(got, ref[k]), k
            assert np.array_equal(ns, ref["num_surfaces"])
            assert ctx.speculation_counts()[0] == n
            ctx.use_own_stream()
    finally:
        dev.free()


def test_two_contexts_in_two_threads():
    """One context per thread on the same GPU, both at work at once (ctypes releases the interpreter lock during the
    calls): one-frame calls with their polled label counts and batched calls, each thread's results those of a run
    alone."""
    import threading
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 192, 256
    sets = [make_batch([51, 52, 53, 54], H, W), make_batch([61, 62, 63, 64], H, W)]
    with Context(H, W, max_batch=4) as ctx:
        alone = [ctx.find_surfaces_batch(d, c) for d, c in sets]
    results, errors = [None, None], []

    def work(k):
        try:
            d, c = sets[k]
            with Context(H, W, max_batch=4) as own:
                singles = [own.find_surfaces_batch(d[i], c[i]) for i in range(4) for _ in range(3)]
                batch = own.find_surfaces_batch(d, c)
                results[k] = (singles, batch, own.speculation_counts())
        except Exception as e:                      # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for k in range(2):
        singles, batch, (held, repeated) = results[k]
        assert held == 12 and repeated == 0
        for key in KEYS:
            assert np.array_equal(batch[key], alone[k][key]), key
            for j, one in enumerate(singles):
                assert np.array_equal(one[key], alone[k][key][j // 3]), (key, j)


def test_small_host_calls_with_and_without_early_downloads():
    """A host-buffer call of one chunk (up to 16 frames) sends each group of outputs to the host as soon as its kernels are
    enqueued and stages pageable inputs through page-locked memory; the arrays it fills are those of a context that does
    neither, for pageable and page-locked arrays alike."""
    from imagesegmentation_amd._lib import Context, pinned_empty
    from imagesegmentation_amd.synth import make_batch
    H, W, n = 96, 128, 5
    d, c = make_batch([71, 72, 73, 74, 75], H, W)
    old = os.environ.get("FS_NO_EARLY_DOWNLOADS")
    os.environ["FS_NO_EARLY_DOWNLOADS"] = "1"
    try:
        late = Context(H, W, max_batch=n)
    finally:
        if old is None:
            del os.environ["FS_NO_EARLY_DOWNLOADS"]
        else:
            os.environ["FS_NO_EARLY_DOWNLOADS"] = old
    with late, Context(H, W, max_batch=n) as ctx:
        ref = late.find_surfaces_batch(d, c)
        pd, pc = pinned_empty(d.shape, np.uint16), pinned_empty(c.shape, np.uint8)
        pd[...] = d
        pc[...] = c
        for depth, rgb, what in ((d, c, "pageable inputs"), (pd, pc, "page-locked inputs"), (pd, c, "one of each")):
            got = ctx.find_surfaces_batch(depth, rgb)
            _same(got, ref, what)
            for i in range(n):
                _same(ctx.find_surfaces_batch(depth[i], rgb[i]), {k: ref[k][i] for k in KEYS + ("num_surfaces",)}, f"{what}, frame {i}")


def test_large_host_calls_with_and_without_narrow_label_downloads():
    """A host-buffer call of several chunks (more than 16 frames) sends patches and surfaces as 16-bit and depth edges as 8-bit
    images and widens them into the caller's arrays on worker threads; the arrays it fills are those of a context that copies
    the int64 / int32 images as they are, with a last chunk that is not full, with one worker thread and with many, and when
    the caller wants only some of the label images."""
    import ctypes
    from imagesegmentation_amd import _lib
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W, n = 96, 128, 37                                    # chunks of 16, 16 and 5 frames
    d, c = make_batch(list(range(200, 200 + n)), H, W)

    def context(**env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return Context(H, W, max_batch=n)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    with context(FS_NO_NARROW_DOWNLOADS="1") as plain, context() as narrow, context(FS_WIDEN_THREADS="1") as one, \
            context(FS_WIDEN_THREADS="13") as many:
        ref = plain.find_surfaces_batch(d, c)
        assert ref["surfaces"].max() > 0 and ref["depth_edges"].max() == 1
        for ctx, what in ((narrow, "narrow"), (one, "one worker"), (many, "thirteen workers")):
            _same(ctx.find_surfaces_batch(d, c), ref, what)
            _same(ctx.find_surfaces_batch(d[:17], c[:17]), {k: ref[k][:17] for k in KEYS + ("num_surfaces",)}, what + ", 17 frames")
        # a frame's arrays do not depend on its place in the call (another chunk, another worker's part of the chunk)
        perm = np.arange(n)[::-1].copy()
        rev = narrow.find_surfaces_batch(d[perm], c[perm])
        _same({k: np.asarray(rev[k])[perm] for k in KEYS + ("num_surfaces",)}, ref, "frames in reverse order")
        # only the final labels wanted: the other outputs are null pointers
        lib = _lib.load()
        sur = np.full((n, H, W), -1, np.int64)
        ns = np.zeros(n, np.int32)
        null = ctypes.c_void_p()
        rc = lib.fs_find_surfaces_batch(narrow._h, n, _lib._ptr(d), _lib._ptr(c), null, null, null, null, null, _lib._ptr(sur), _lib._ptr(ns))
        assert rc == 0, lib.fs_last_error(narrow._h)
        assert np.array_equal(sur, ref["surfaces"]) and np.array_equal(ns, ref["num_surfaces"])


def test_results_do_not_depend_on_what_freed_memory_held():
    """Device memory that the process has freed comes back from the allocator with its old contents.  A context's
    buffers start as zeros all the same: fill a gigabyte with all-ones bytes (NaN as floats), free it, create a
    context in its place and compare with a context that was created before."""
    from conftest import DeviceArrays
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch, make_scene
    H, W = 480, 640
    one = make_scene(5, H, W)
    d, c = make_batch([81, 82, 83], H, W)
    with Context(H, W, max_batch=3) as before:
        ref_one, ref_batch = before.find_surfaces_batch(*one), before.find_surfaces_batch(d, c)
    dirty = DeviceArrays()
    for _ in range(8):
        dirty.alloc(128 << 20, fill=0xFF)
    dirty.free()
    with Context(H, W, max_batch=3) as after:
        _same(after.find_surfaces_batch(*one), ref_one, "one frame")
        _same(after.find_surfaces_batch(d, c), ref_batch, "three frames")


def test_results_do_not_depend_on_what_the_work_buffers_held():
    """FS_DEBUG_NAN fills the float work buffers with NaN and the CRF's tile records with -1 in front of every call: a
    kernel that used a value this call has not written would show it.  (A tile's channel list is walked in groups of
    four; the slots that fill up the last group once compared whatever the record held behind its labels with the
    pixel's own label, and a match shifted the softmax's maximum: labels flipped at near-ties, depending on what earlier
    calls had left there.)"""
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W = 480, 640
    d, c = make_batch([91, 92, 93], H, W)
    with Context(H, W, max_batch=3) as ctx:
        ref_batch = ctx.find_surfaces_batch(d, c)
        ref_one = [ctx.find_surfaces_batch(d[i], c[i]) for i in range(3)]
        ctx.set_debug_fill("qabsrt")
        try:
            for _ in range(2):
                _same(ctx.find_surfaces_batch(d, c), ref_batch, "three frames")
                for i in range(3):
                    _same(ctx.find_surfaces_batch(d[i], c[i]), ref_one[i], f"frame {i} alone")
        finally:
            ctx.set_debug_fill("")
