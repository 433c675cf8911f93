"""ctypes binding of libfs_hip.so (C ABI in include/fs_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or
``make -C imagesegmentation_amd/csrc``).  There is no CPU fallback: if the
library is missing or no GPU is usable, every entry point raises.
"""
import ctypes
import json
import os
import struct
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FS_HIP_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("FS_HIP_LIB") or os.path.join(_HERE, "libfs_hip.so")

_c = ctypes
_vp, _i, _d = _c.c_void_p, _c.c_int, _c.c_double

# name -> (restype, argtypes); mirrors include/fs_hip.h one to one
SIGNATURES = {
    "fs_create": (_i, [_c.POINTER(_vp), _i, _i, _i, _i]),
    "fs_destroy": (None, [_vp]),
    "fs_last_error": (_c.c_char_p, [_vp]),
    "fs_set_view_angles": (_i, [_vp, _d, _d]),
    "fs_set_gauss_params": (_i, [_vp, _vp]),
    "fs_set_lr_params": (_i, [_vp, _vp]),
    "fs_set_kernel_size": (_i, [_vp, _i]),
    "fs_set_max_batch": (_i, [_vp, _i]),
    "fs_set_stream": (_i, [_vp, _vp]),
    "fs_use_own_stream": (_i, [_vp]),
    "fs_stream": (_vp, [_vp]),
    "fs_smooth_depth": (_i, [_vp, _i, _vp, _vp]),
    "fs_normals": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "fs_patches": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "fs_connected_components": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "fs_connected_components_noalloc": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "fs_noalloc_max_pixels": (_c.c_size_t, []),
    "fs_crf_step": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_crf_step_lr": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 9),
    "fs_find_surfaces_batch": (_i, [_vp, _i] + [_vp] * 9),
    "fs_find_surfaces_trace": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "fs_synchronize": (_i, [_vp]),
    "fs_host_alloc": (_vp, [_c.c_size_t]),
    "fs_host_free": (None, [_vp]),
    "fs_find_surfaces_batch_dev": (_i, [_vp, _i] + [_vp] * 9),
    "fs_apply_crf": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "fs_surface_stats": (_i, [_vp, _i] + [_vp] * 7 + [_i] + [_vp] * 15),
    "fs_surface_stats_rgb": (_i, [_vp, _i] + [_vp] * 7 + [_i] + [_vp] * 16),
    "fs_surface_stats_dev": (_i, [_vp, _i] + [_vp] * 7 + [_i] + [_vp] * 15),
    "fs_surface_stats_batch_dev": (_i, [_vp, _i] + [_vp] * 9 + [_i] + [_vp] * 16),
    "fs_chi_squared_distances": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fs_chi_squared_distances_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fs_overlap_counts": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "fs_nearest_points": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "fs_set_debug_fill": (_i, [_vp, _c.c_char_p]),
    "fs_set_timing": (_i, [_vp, _i]),
    "fs_get_tile_labels": (_i, [_vp, _i, _vp] + [_c.POINTER(_i)] * 4),
    "fs_get_timing": (_i, [_vp, _i, _c.POINTER(_d), _c.POINTER(_c.c_int64)]),
    "fs_get_speculation_counts": (_i, [_vp, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
}

_LIB = None


def load():
    """Load libfs_hip.so and declare every prototype.  Raises if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C imagesegmentation_amd/csrc` (there is no CPU fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


def gauss_parameters():
    """load_Gauss_parameters (find_surfaces.py:221-226) from the JSON copy of the
    shipped float32 values: bias forced to 0, weight 0.01 appended."""
    with open(os.path.join(_HERE, "params", "gauss.json")) as f:
        doc = json.load(f)
    p = [struct.unpack("<f", bytes.fromhex(h))[0] for h in doc["hex"]]
    p.append(0.01)
    p[3] = 0
    return np.asarray(p, np.float32)


def lr_parameters():
    """(coef[4], intercept + 4, weight, auxiliary weight): load_pixel_similarity_parameters
    (find_surfaces.py:216-218) and LR_CRF_parameters (:22) from the JSON copy of the shipped values."""
    with open(os.path.join(_HERE, "params", "lr.json")) as f:
        doc = json.load(f)
    return np.asarray(list(doc["coef"]) + [doc["intercept"] + 4, 0.01, 15000], np.float32)


# Page-locking memory costs about a millisecond per few megabytes, more than the stage takes for a frame, so blocks
# that their arrays have let go are kept for the next request of a similar size (up to _POOL_LIMIT bytes in all).
_POOL = {}                  # block size -> addresses of free blocks
_POOL_BYTES = 0
_POOL_LIMIT = 1 << 30
_POOL_LOCK = threading.RLock()      # re-entrant: a collection inside a locked section may finalise another block


def _block_size(nbytes):
    """Sizes come in eight steps per power of two, so that a request finds the block an earlier, slightly different one left."""
    n = max(int(nbytes), 4096)
    step = 1 << max(n.bit_length() - 4, 12)
    return (n + step - 1) // step * step


class _PinnedBlock:
    """Page-locked host allocation that lives as long as an array built on it."""

    def __init__(self, nbytes):
        global _POOL_BYTES
        self._lib = load()
        self.size = _block_size(nbytes)
        self.ptr = None
        with _POOL_LOCK:
            free = _POOL.get(self.size)
            if free:
                self.ptr = free.pop()
                _POOL_BYTES -= self.size
        if not self.ptr:
            self.ptr = self._lib.fs_host_alloc(self.size)
        if not self.ptr:
            raise MemoryError(f"fs_host_alloc({self.size}) failed")
        self.nbytes = int(nbytes)

    def __del__(self):
        global _POOL_BYTES
        try:
            ptr, self.ptr = getattr(self, "ptr", None), None
            if not ptr:
                return
            with _POOL_LOCK:
                if _POOL_BYTES + self.size <= _POOL_LIMIT:
                    _POOL.setdefault(self.size, []).append(ptr)
                    _POOL_BYTES += self.size
                    return
            self._lib.fs_host_free(ptr)
        except Exception:      # interpreter shutdown: the library may already be gone
            pass


def release_pinned_pool():
    """Return the kept page-locked blocks to the system."""
    global _POOL_BYTES
    with _POOL_LOCK:
        blocks = [p for free in _POOL.values() for p in free]
        _POOL.clear()
        _POOL_BYTES = 0
    lib = load()
    for p in blocks:
        lib.fs_host_free(p)


def pinned_empty(shape, dtype):
    """numpy array in page-locked memory (freed when the last view of it goes away)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    block = _PinnedBlock(n)
    buf = (_c.c_char * max(n, 1)).from_address(block.ptr)
    buf._block = block                    # the ctypes buffer is the array's base; it keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return a
    return a.ctypes.data


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


class Context:
    """One fs_ctx: a GPU, a stream, frame size (H, W) and a maximum batch."""

    def __init__(self, height, width, max_batch=1, device=0, kernel_size=7, params=None):
        self._lib = load()
        self._h = _vp()
        self.H, self.W, self.max_batch, self.device = int(height), int(width), int(max_batch), int(device)
        if self._lib.fs_create(ctypes.byref(self._h), self.device, self.H, self.W, self.max_batch) != 0:
            raise RuntimeError("fs_create failed: " + self._lib.fs_last_error(None).decode())
        self.kernel_size = 7
        self.set_kernel_size(kernel_size)
        if params is not None and len(np.reshape(params, -1)) == 7:
            self.set_lr_params(params)
        else:
            self.set_gauss_params(gauss_parameters() if params is None else params)

    # -- plumbing -----------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.fs_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.fs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _same_params(self, p, mode):
        """True if the context already holds exactly these parameters (setting them drains the stream and uploads the
        weight tables: callers that pass their parameters with every frame should not pay that per frame)."""
        have = getattr(self, "params", None)
        return getattr(self, "_params_mode", None) == mode and have is not None and have.shape == p.shape and have.tobytes() == p.tobytes()

    def set_gauss_params(self, params):
        p = _arr(np.asarray(params, np.float32).reshape(-1), np.float32, (13,))
        if self._same_params(p, "gauss"):
            return
        self._check(self._lib.fs_set_gauss_params(self._h, _ptr(p)))
        self.params, self._params_mode = p.copy(), "gauss"

    def set_lr_params(self, params):
        p = _arr(np.asarray(params, np.float32).reshape(-1), np.float32, (7,))
        if self._same_params(p, "lr"):
            return
        self._check(self._lib.fs_set_lr_params(self._h, _ptr(p)))
        self.params, self._params_mode = p.copy(), "lr"

    def set_kernel_size(self, r):
        if int(r) == getattr(self, "kernel_size", None):
            return
        self._check(self._lib.fs_set_kernel_size(self._h, int(r)))
        self.kernel_size = int(r)

    def set_max_batch(self, max_batch):
        """Grow the context in place (handles held elsewhere stay valid)."""
        self._check(self._lib.fs_set_max_batch(self._h, int(max_batch)))
        self.max_batch = max(self.max_batch, int(max_batch))

    def set_view_angles(self, view_x, view_y):
        self._check(self._lib.fs_set_view_angles(self._h, float(view_x), float(view_y)))

    def set_stream(self, stream_ptr):
        """Run on the caller's stream; 0 / None is HIP's legacy default stream."""
        self._check(self._lib.fs_set_stream(self._h, stream_ptr or None))

    def use_own_stream(self):
        self._check(self._lib.fs_use_own_stream(self._h))

    def stream(self):
        return self._lib.fs_stream(self._h)

    def set_timing(self, group_mask):
        self._check(self._lib.fs_set_timing(self._h, int(group_mask)))

    def speculation_counts(self):
        """(held, repeated): one-frame calls whose updates went out ahead of the label count and stood, or had to be
        run again because the frame did not fit what they were sized for."""
        held, rep = _c.c_int64(0), _c.c_int64(0)
        self._check(self._lib.fs_get_speculation_counts(self._h, ctypes.byref(held), ctypes.byref(rep)))
        return held.value, rep.value

    def get_timing(self, group):
        ms, cnt = _d(0), _c.c_int64(0)
        self._check(self._lib.fs_get_timing(self._h, group, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def tile_labels(self, n):
        """Patch labels listed per CRF tile in the last fused call of n frames -> (int32 [n, tiles_y, tiles_x],
        (tile_h, tile_w))."""
        tx, ty, tw, th = _i(0), _i(0), _i(0), _i(0)
        self._check(self._lib.fs_get_tile_labels(self._h, int(n), None, ctypes.byref(tx), ctypes.byref(ty), ctypes.byref(tw),
                                                 ctypes.byref(th)))
        out = np.empty((int(n), ty.value, tx.value), np.int32)
        self._check(self._lib.fs_get_tile_labels(self._h, int(n), _ptr(out), None, None, None, None))
        return out, (th.value, tw.value)

    def _batch(self, a, tail=()):
        a = np.asarray(a)
        single = a.ndim == 2 + len(tail)
        if single:
            a = a[None]
        if a.shape[1:] != (self.H, self.W) + tuple(tail):
            raise ValueError(f"expected [n,{self.H},{self.W}{''.join(',%d' % t for t in tail)}], got {a.shape}")
        return a, single

    # -- stage operators -------------------------------------------------------
    def smooth_depth(self, depth):
        d, single = self._batch(depth)
        d = _arr(d, np.float32)
        out = np.empty_like(d)
        self._check(self._lib.fs_smooth_depth(self._h, d.shape[0], _ptr(d), _ptr(out)))
        return out[0] if single else out

    def normals(self, smoothed, want_raw=False):
        d, single = self._batch(smoothed)
        d = _arr(d, np.float32)
        n = d.shape[0]
        ang, vec, pts = (np.empty((n, self.H, self.W, c), np.float32) for c in (2, 3, 3))
        raw = np.empty((n, self.H, self.W, 2), np.float32) if want_raw else None
        self._check(self._lib.fs_normals(self._h, n, _ptr(d), _ptr(ang), _ptr(vec), _ptr(pts), _ptr(raw)))
        res = [ang, vec, pts] + ([raw] if want_raw else [])
        return [a[0] for a in res] if single else res

    def patches(self, smoothed, want_curvature=False, want_mask=False):
        """-> [patches, depth_edges] (+ [curvature scores, curvature mask] with want_curvature, which evaluates every
        score with correctly rounded arithmetic; + [curvature mask] with want_mask, which keeps the production path: the
        approximate score decides wherever it is clear of the threshold)."""
        d, single = self._batch(smoothed)
        d = _arr(d, np.float32)
        n = d.shape[0]
        pat, edg = np.empty((n, self.H, self.W), np.int64), np.empty((n, self.H, self.W), np.int32)
        sc = np.empty((n, self.H, self.W), np.float32) if want_curvature else None
        cm = np.empty((n, self.H, self.W), np.uint8) if (want_curvature or want_mask) else None
        self._check(self._lib.fs_patches(self._h, n, _ptr(d), _ptr(pat), _ptr(edg), _ptr(sc), _ptr(cm)))
        res = [pat, edg] + ([sc, cm] if want_curvature else ([cm] if want_mask else []))
        return [a[0] for a in res] if single else res

    def connected_components(self, mask, min_size_exclusive=0):
        m, single = self._batch(mask)
        m = _arr(m != 0, np.uint8)
        n = m.shape[0]
        lab, cnt = np.empty((n, self.H, self.W), np.int64), np.empty(n, np.int32)
        self._check(self._lib.fs_connected_components(self._h, n, _ptr(m), int(min_size_exclusive), _ptr(lab), _ptr(cnt)))
        return (lab[0], int(cnt[0])) if single else (lab, cnt)

    def connected_components_noalloc(self, n, d_mask, d_labels, d_num_labels=None, min_size_exclusive=0):
        """fs_connected_components_noalloc: device pointers (integers); enqueues on the context's stream,
        allocates nothing, does not synchronise."""
        self._check(self._lib.fs_connected_components_noalloc(self._h, int(n), d_mask, int(min_size_exclusive), d_labels,
                                                              d_num_labels))

    def set_debug_fill(self, letters):
        """fs_set_debug_fill: work buffers to overwrite in front of every call ("" switches it off)."""
        self._check(self._lib.fs_set_debug_fill(self._h, (letters or "").encode()))

    def synchronize(self):
        """Wait for the context's stream (fs_synchronize: the library's own HIP runtime does the waiting)."""
        self._check(self._lib.fs_synchronize(self._h))

    def crf_step(self, unary, Q, f1, f2, f3, val, want_potentials=False):
        unary = _arr(unary, np.float32)
        B, h, w, S = unary.shape
        r = self.kernel_size
        Q = _arr(Q, np.float32, (B, h + 2 * r, w + 2 * r, S))
        f1 = _arr(f1, np.float32, (B, h + 2 * r, w + 2 * r, 3))
        f2 = _arr(f2, np.float32, (B, h + 2 * r, w + 2 * r, 6))
        f3 = _arr(f3, np.float32, (B, h + 2 * r, w + 2 * r, 5))
        val = _arr(np.asarray(val, np.float32).reshape(-1, S)[0], np.float32)
        out = np.empty((B, h, w, S), np.float32)
        pot = np.empty((B, h, w, S), np.float32) if want_potentials else None
        self._check(self._lib.fs_crf_step(self._h, B, h, w, S, _ptr(unary), _ptr(Q), _ptr(f1), _ptr(f2), _ptr(f3),
                                          _ptr(val), _ptr(out), _ptr(pot)))
        return (out, pot) if want_potentials else out

    def crf_step_lr(self, unary, Q, f1, f2, f3, f4, val, want_potentials=False):
        unary = _arr(unary, np.float32)
        B, h, w, S = unary.shape
        r = self.kernel_size
        pad = (B, h + 2 * r, w + 2 * r)
        Q = _arr(Q, np.float32, pad + (S,))
        f1, f2, f3, f4 = (_arr(f, np.float32, pad + (c,)) for f, c in ((f1, 1), (f2, 2), (f3, 3), (f4, 2)))
        val = _arr(np.asarray(val, np.float32).reshape(-1, S)[0], np.float32)
        out = np.empty((B, h, w, S), np.float32)
        pot = np.empty((B, h, w, S), np.float32) if want_potentials else None
        self._check(self._lib.fs_crf_step_lr(self._h, B, h, w, S, _ptr(unary), _ptr(Q), _ptr(f1), _ptr(f2), _ptr(f3), _ptr(f4),
                                             _ptr(val), _ptr(out), _ptr(pot)))
        return (out, pot) if want_potentials else out

    # -- fused path --------------------------------------------------------------
    def find_surfaces_batch(self, depth, rgb, out=None):
        """depth uint16 [n,H,W], rgb uint8 [n,H,W,3] -> dict of batched arrays with the
        keys of the dict find_surfaces() returns (find_surfaces.py:503-532).  `out`: a dict
        returned by an earlier call with the same n, to be overwritten instead of allocating."""
        d, single = self._batch(depth)
        c, _ = self._batch(rgb, (3,))
        d, c = _arr(d, np.uint16), _arr(c, np.uint8)
        n = d.shape[0]
        # outputs live in page-locked memory: the downloads are DMA transfers overlapped with the kernels
        shapes = {"angles": ((2,), np.float32), "vectors": ((3,), np.float32), "points_3d": ((3,), np.float32),
                  "patches": ((), np.int64), "depth_edges": ((), np.int32), "surfaces": ((), np.int64)}
        if out is None:
            out = {k: pinned_empty((n, self.H, self.W) + tail, dt) for k, (tail, dt) in shapes.items()}
        else:
            out = {k: (out[k][None] if single else out[k]) for k in shapes}
            for k, (tail, dt) in shapes.items():
                if out[k].shape != (n, self.H, self.W) + tail or out[k].dtype != dt or not out[k].flags.c_contiguous:
                    raise ValueError(f"out[{k!r}] does not fit this batch")
        ns = np.empty(n, np.int32)
        self._check(self._lib.fs_find_surfaces_batch(
            self._h, n, _ptr(d), _ptr(c), _ptr(out["angles"]), _ptr(out["vectors"]), _ptr(out["points_3d"]),
            _ptr(out["patches"]), _ptr(out["depth_edges"]), _ptr(out["surfaces"]), _ptr(ns)))
        out["num_surfaces"] = ns
        if single:
            out = {k: v[0] for k, v in out.items()}
            out["num_surfaces"] = int(out["num_surfaces"])
        return out

    def find_surfaces_trace(self, depth, rgb):
        """fs_find_surfaces_trace for one frame: the stage with the diagnostic builds of the mean-field updates ->
        dict with potentials and probabilities float32 [3, S, H, W] (zero where the frame has no depth), patches,
        surfaces, num_surfaces."""
        d, c = _arr(depth, np.uint16, (self.H, self.W)), _arr(rgb, np.uint8, (self.H, self.W, 3))
        S = int(self.find_surfaces_batch(d, c)["num_surfaces"]) + 1
        pot, q = np.empty((3, S, self.H, self.W), np.float32), np.empty((3, S, self.H, self.W), np.float32)
        pat, sur = np.empty((self.H, self.W), np.int64), np.empty((self.H, self.W), np.int64)
        ns = np.zeros(1, np.int32)
        self._check(self._lib.fs_find_surfaces_trace(self._h, _ptr(d), _ptr(c), S, _ptr(pot), _ptr(q), _ptr(pat), _ptr(sur), _ptr(ns)))
        return {"potentials": pot, "probabilities": q, "patches": pat, "surfaces": sur, "num_surfaces": int(ns[0])}

    def find_surfaces_batch_dev(self, n, d_depth, d_rgb, d_angles=None, d_vectors=None, d_points=None, d_patches=None,
                                d_edges=None, d_surfaces=None):
        """Device-pointer variant (integers or None); returns num_surfaces[n]."""
        ns = np.empty(n, np.int32)
        self._check(self._lib.fs_find_surfaces_batch_dev(self._h, n, d_depth, d_rgb, d_angles, d_vectors, d_points,
                                                         d_patches, d_edges, d_surfaces, _ptr(ns)))
        return ns

    def surface_stats(self, surfaces, patches, depth_edges, points_3d, angles, lab, texture, num_surfaces, rgb=None):
        """fs_surface_stats for one frame -> dict of arrays (see include/fs_hip.h).  With lab=None and rgb given
        (uint8 [H,W,3]) the colour conversion runs on the device (fs_surface_stats_rgb) and the dict also holds "lab".
        The large outputs live in page-locked memory (their downloads are DMA transfers)."""
        S, H, W = int(num_surfaces), self.H, self.W
        surfaces, patches = _arr(surfaces, np.int64, (H, W)), _arr(patches, np.int64, (H, W))
        depth_edges = _arr(depth_edges, np.int32, (H, W))
        points_3d, angles = _arr(points_3d, np.float32, (H, W, 3)), _arr(angles, np.float32, (H, W, 2))
        T = 256
        if texture is not None:
            texture = _arr(texture, np.float32)
            if texture.shape[:2] != (H // 4, W // 4):
                raise ValueError(f"texture must be [{H // 4},{W // 4},T], got {texture.shape}")
            T = texture.shape[2]
        out = {
            "surfaces": pinned_empty((H, W), np.int64), "patch_point_offsets": np.empty(S + 1, np.int32),
            "patch_points": pinned_empty((H * W, 2), np.float32), "avg_pos": np.empty((S, 2)), "counts": np.empty((S, 1)),
            "centroids": np.empty((S, 3), np.float32), "centroid_indices": np.empty((S, 2), np.int32),
            "hist_color": pinned_empty((256, 256, S), np.float64), "hist_angle": np.empty((40, 40, S)), "hist_texture": np.empty((S, T)),
            "avg_angles": np.empty((S, 2)), "neighbor_counts": np.empty((S, S)), "border_centers": np.empty((S, S, 2)),
            "norm": pinned_empty((H, W), np.float32), "depth_extend": np.empty((S, 4, 2), np.float32),
        }
        if lab is not None:
            lab = _arr(lab, np.float64, (H, W, 3))
            self._check(self._lib.fs_surface_stats(self._h, S, _ptr(surfaces), _ptr(patches), _ptr(depth_edges), _ptr(points_3d),
                                                   _ptr(angles), _ptr(lab), _ptr(texture), T, *(_ptr(out[k]) for k in self.STATS_OUTPUTS)))
        else:
            rgb = _arr(rgb, np.uint8, (H, W, 3))
            lab_out = pinned_empty((H, W, 3), np.float64)
            self._check(self._lib.fs_surface_stats_rgb(self._h, S, _ptr(surfaces), _ptr(patches), _ptr(depth_edges), _ptr(points_3d),
                                                       _ptr(angles), _ptr(rgb), _ptr(texture), T, _ptr(lab_out),
                                                       *(_ptr(out[k]) for k in self.STATS_OUTPUTS)))
            out["lab"] = lab_out
        return out

    STATS_OUTPUTS = ("surfaces", "patch_point_offsets", "patch_points", "avg_pos", "counts", "centroids", "centroid_indices",
                     "hist_color", "hist_angle", "hist_texture", "avg_angles", "neighbor_counts", "border_centers", "norm",
                     "depth_extend")

    def surface_stats_dev(self, num_surfaces, d_surfaces, d_patches, d_depth_edges, d_points_3d, d_angles, d_lab,
                          d_texture=None, texture_dim=256, **d_out):
        """fs_surface_stats_dev: device pointers (integers); outputs by the names of STATS_OUTPUTS, missing ones
        are not produced.  Sizes as in surface_stats()."""
        unknown = set(d_out) - set(self.STATS_OUTPUTS)
        if unknown:
            raise TypeError(f"unknown outputs {sorted(unknown)}")
        self._check(self._lib.fs_surface_stats_dev(self._h, int(num_surfaces), d_surfaces, d_patches, d_depth_edges, d_points_3d,
                                                   d_angles, d_lab, d_texture, int(texture_dim),
                                                   *(d_out.get(k) for k in self.STATS_OUTPUTS)))

    @staticmethod
    def stats_output_elements(H, W, S, T=256):
        """Elements of each output of the statistics stage for one frame with S surfaces, by the names of
        STATS_OUTPUTS (the layout fs_surface_stats_batch_dev concatenates frame after frame)."""
        N = H * W
        return {"surfaces": N, "patch_point_offsets": S + 1, "patch_points": 2 * N, "avg_pos": 2 * S, "counts": S, "centroids": 3 * S,
                "centroid_indices": 2 * S, "hist_color": 65536 * S, "hist_angle": 1600 * S, "hist_texture": S * T, "avg_angles": 2 * S,
                "neighbor_counts": S * S, "border_centers": 2 * S * S, "norm": N, "depth_extend": 8 * S}

    def surface_stats_batch_dev(self, num_surfaces, d_surfaces, d_patches, d_depth_edges, d_points_3d, d_angles, d_lab=None, d_rgb=None,
                                d_texture=None, texture_dim=256, d_lab_out=None, **d_out):
        """fs_surface_stats_batch_dev: device pointers (integers) of arrays that hold the frames back to back;
        num_surfaces: one entry per frame.  Outputs by the names of STATS_OUTPUTS, missing ones are not produced."""
        unknown = set(d_out) - set(self.STATS_OUTPUTS)
        if unknown:
            raise TypeError(f"unknown outputs {sorted(unknown)}")
        ns = np.ascontiguousarray(num_surfaces, np.int32).reshape(-1)
        self._check(self._lib.fs_surface_stats_batch_dev(self._h, len(ns), _ptr(ns), d_surfaces, d_patches, d_depth_edges, d_points_3d,
                                                         d_angles, d_lab, d_rgb, d_texture, int(texture_dim), d_lab_out,
                                                         *(d_out.get(k) for k in self.STATS_OUTPUTS)))

    def chi_squared_distances(self, hist, pool_size=(1, 1), strides=(1, 1)):
        hist = _arr(hist, np.float64)
        if hist.ndim == 2:                       # 1-D model: [K, S]
            hist = hist[None]
        A, B, S = hist.shape
        out = np.empty((S, S), np.float32)
        self._check(self._lib.fs_chi_squared_distances(self._h, _ptr(hist), A, B, S, int(pool_size[0]), int(pool_size[1]),
                                                       int(strides[0]), int(strides[1]), _ptr(out)))
        return out

    def chi_squared_distances_dev(self, d_hist, shape, d_out, pool_size=(1, 1), strides=(1, 1)):
        """fs_chi_squared_distances_dev: d_hist float64 [A][B][S] and d_out float32 [S][S] are device pointers
        (integers), shape = (A, B, S).  Enqueues on the context's stream; does not synchronise."""
        A, B, S = (int(v) for v in shape)
        self._check(self._lib.fs_chi_squared_distances_dev(self._h, d_hist, A, B, S, int(pool_size[0]), int(pool_size[1]),
                                                           int(strides[0]), int(strides[1]), d_out))

    def overlap_counts(self, annotation, detected, num_annotations, num_detections):
        a = _arr(annotation, np.int64, (self.H, self.W))
        o = _arr(detected, np.int64, (self.H, self.W))
        out = np.empty((int(num_detections), int(num_annotations)))
        self._check(self._lib.fs_overlap_counts(self._h, _ptr(a), _ptr(o), int(num_annotations), int(num_detections), _ptr(out)))
        return out

    def nearest_points(self, point_sets, target_points):
        """calculate_nearest_points (process_surfaces.py:649-654): lists of [n_i, 2] point arrays and of
        [m_i, 2] target arrays -> list of int32 [m_i, 2] arrays."""
        if len(point_sets) != len(target_points):
            raise ValueError("need as many target lists as point sets")
        pts = [np.asarray(p, np.float32).reshape(-1, 2) for p in point_sets]
        tgs = [np.asarray(t, np.float32).reshape(-1, 2) for t in target_points]
        po = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.int32)
        to = np.concatenate([[0], np.cumsum([len(t) for t in tgs])]).astype(np.int32)
        P = _arr(np.concatenate(pts) if pts else np.zeros((0, 2)), np.float32)
        T = _arr(np.concatenate(tgs) if tgs else np.zeros((0, 2)), np.float32)
        out = np.empty((len(T), 2), np.int32)
        self._check(self._lib.fs_nearest_points(self._h, len(pts), _ptr(P), _ptr(po), _ptr(T), _ptr(to), _ptr(out)))
        return [out[to[i]:to[i + 1]] for i in range(len(tgs))]

    def apply_crf(self, labels, depth, rgb, angles):
        lab, single = self._batch(labels)
        d, _ = self._batch(depth)
        c, _ = self._batch(rgb, (3,))
        a, _ = self._batch(angles, (2,))
        lab, d, c, a = _arr(lab, np.int64), _arr(d, np.uint16), _arr(c, np.uint8), _arr(a, np.float32)
        out = pinned_empty(lab.shape, np.int64)          # page-locked: the download is a DMA transfer into the result itself
        self._check(self._lib.fs_apply_crf(self._h, lab.shape[0], _ptr(lab), _ptr(d), _ptr(c), _ptr(a), _ptr(out)))
        return out[0] if single else out
