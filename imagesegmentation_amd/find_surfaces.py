"""Host-side mirror of the reference's find_surfaces module over libfs_hip.so.

Same names, argument meaning and returned structures as find_surfaces.py of the
reference (lines cited per function), so that its callers -- main.find_surfaces_for_dataset,
the three get_full_prediction_model factories, post_processing -- can import this module
instead.  All arithmetic runs in the HIP library; there is no CPU path here.

Two ways through the stage exist, with identical results:
  * the fused path: find_surfaces()/find_surfaces_for_indices() hand whole frames to
    fs_find_surfaces_batch (one call, batched over frames);
  * the operator path: the four objects get_GPU_models() returns can be called one by one
    exactly as the reference's find_surfaces() body does (smoothing, normals, patches,
    conv_crf.predict on 4x4 tiles); find_surfaces() takes that route when handed a plain
    4-tuple of callables that does not come from get_GPU_models().  conv_crf_Gauss.predict carries at most 128
    label channels per call (fs_crf_step), so that route stops at 127 surfaces per frame; the fused path carries 512.
"""
import os
import threading

import numpy as np

from . import _lib
from . import load_images as _li
from .config import height, width, viewing_angle_x, viewing_angle_y, test_indices, train_indices  # noqa: F401
from .load_images import load_image, load_image_and_surface_information, sorted_alphanumeric  # noqa: F401

similarity_normalizers = [6, 2, 1 / 50, 1 / 40]     # depth, angle, colour, position (find_surfaces.py:20)
LR_CRF_parameters = [0.01, 15000]                    # weight, auxiliary weight (find_surfaces.py:22)
standard_kernel_size = 7                             # find_surfaces.py:23
div_x, div_y = 4, 4                                  # find_surfaces.py:24
frames_per_call = 37                                 # frames handed to the GPU per call by find_surfaces_for_indices

# GPU of this process: FS_DEVICE if set (e.g. several ranks sharing one GPU), else the launcher's LOCAL_RANK
_device = int(os.environ.get("FS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
_local = threading.local()      # a library context serves one call at a time: every thread keeps its own


def set_device(index):
    """GPU used by the models built afterwards (one process drives one GPU)."""
    global _device
    _device = int(index)


def _context(h, w, kernel_size, max_batch=1, params=None):
    key = (_device, int(h), int(w))
    _contexts = _local.__dict__.setdefault("contexts", {})
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _lib.Context(h, w, max_batch=max_batch, device=_device, kernel_size=kernel_size, params=params)
        ctx.set_view_angles(viewing_angle_x, viewing_angle_y)
        _contexts[key] = ctx
    elif ctx.max_batch < max_batch:
        ctx.set_max_batch(max_batch)            # grows in place: contexts handed out earlier stay valid
    if ctx.kernel_size != kernel_size:
        ctx.set_kernel_size(kernel_size)
    if params is not None:
        if len(params) == 7:
            ctx.set_lr_params(params)         # logistic-regression similarity
        else:
            ctx.set_gauss_params(params)
    return ctx


# ---------------------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------------------
def load_Gauss_parameters():
    """The 13 arguments of conv_crf_Gauss, each of shape (1, 1): the twelve trained float32
    values with the bias forced to 0, and the message weight 0.01 appended
    (find_surfaces.py:221-226)."""
    return [np.reshape(v, (1, 1)) for v in _lib.gauss_parameters()]


def load_pixel_similarity_parameters():
    """(coef_ [1, 4], intercept_ + 4 [1]) of the logistic-regression pixel similarity
    (find_surfaces.py:216-218), from the JSON copy of the shipped classifier's float64 values."""
    p = _lib.lr_parameters()
    return np.asarray([p[:4]], np.float64), np.asarray([p[4]], np.float64)


# ---------------------------------------------------------------------------------------
# the four models of get_GPU_models()
# ---------------------------------------------------------------------------------------
class _SmoothingModel:
    """smoothing_model(depth, grid) -> float32 [H, W]
    (gaussian_filter_with_depth_factor_model_GPU, image_processing_models_GPU.py:110-135)."""

    def __init__(self, kernel_size):
        self.kernel_size = kernel_size

    def __call__(self, depth, grid=None):
        depth = np.asarray(depth)
        return _context(depth.shape[-2], depth.shape[-1], self.kernel_size).smooth_depth(depth.astype(np.float32))


class _SurfaceModel:
    """surface_model(smoothed) -> [patches int64, depth_edges int32]
    (find_surfaces_model_GPU, find_surfaces.py:229-288)."""

    def __init__(self, kernel_size):
        self.kernel_size = kernel_size

    def __call__(self, smoothed):
        smoothed = np.asarray(smoothed, np.float32)
        return _context(smoothed.shape[-2], smoothed.shape[-1], self.kernel_size).patches(smoothed)


class _NormalsModel:
    """normals_model(smoothed) -> [angles [H,W,2], vectors [H,W,3], points_3d [H,W,3]]
    (normals_model_GPU, image_processing_models_GPU.py:151-182)."""

    def __init__(self, kernel_size):
        self.kernel_size = kernel_size

    def __call__(self, smoothed):
        smoothed = np.asarray(smoothed, np.float32)
        return _context(smoothed.shape[-2], smoothed.shape[-1], self.kernel_size).normals(smoothed)


class conv_crf_Gauss:
    """One mean-field update with the Gaussian pixel similarity (find_surfaces.py:291-367).

    conv_crf_Gauss(w_1, w_2, w_3, bias, theta_1_1, ..., theta_3_3, weight, kernel_size, height, width)
    returns an object whose predict([unary, Q, f1, f2, f3, val], batch_size=1) maps
      unary [B,h,w,S], Q [B,h+2r,w+2r,S], f1 [..,3], f2 [..,6], f3 [..,5], val [B,1,1,1,S]
    to the updated Q [B,h,w,S] (float32), like the Keras model's predict."""

    def __init__(self, w_1, w_2, w_3, bias, theta_1_1, theta_2_1, theta_3_1, theta_1_2, theta_2_2, theta_3_2, theta_2_3,
                 theta_3_3, weight, kernel_size, height, width):
        self.params = np.asarray([np.reshape(v, ()) for v in (w_1, w_2, w_3, bias, theta_1_1, theta_2_1, theta_3_1, theta_1_2,
                                                              theta_2_2, theta_3_2, theta_2_3, theta_3_3, weight)], np.float32)
        self.kernel_size, self.height, self.width = int(kernel_size), int(height), int(width)

    def predict(self, inputs, batch_size=1, want_potentials=False):
        unary, Q, f1, f2, f3, val = inputs
        unary = np.asarray(unary, np.float32)
        if unary.shape[1:3] != (self.height, self.width):
            raise ValueError(f"model built for tiles of {self.height}x{self.width}, got {unary.shape[1:3]}")
        ctx = _context(height, width, self.kernel_size, params=self.params)
        return ctx.crf_step(unary, Q, f1, f2, f3, val, want_potentials=want_potentials)

    __call__ = predict


class conv_crf_LR:
    """One mean-field update with the linear (logistic-regression) pixel similarity
    (find_surfaces.py:371-434).  conv_crf_LR(LR_weights, LR_bias, weight, auxiliary_weight, kernel_size,
    height, width).predict([unary, Q, f1 depth [..,1], f2 angles [..,2], f3 colour [..,3], f4 (row, col)
    [..,2], val], batch_size=1) -> updated Q [B,h,w,S]."""

    def __init__(self, LR_weights, LR_bias, weight, auxiliary_weight, kernel_size, height, width):
        self.params = np.asarray(list(np.reshape(LR_weights, -1)) + [np.reshape(LR_bias, -1)[0], weight, auxiliary_weight],
                                 np.float32)
        self.kernel_size, self.height, self.width = int(kernel_size), int(height), int(width)

    def predict(self, inputs, batch_size=1, want_potentials=False):
        unary, Q, f1, f2, f3, f4, val = inputs
        unary = np.asarray(unary, np.float32)
        if unary.shape[1:3] != (self.height, self.width):
            raise ValueError(f"model built for tiles of {self.height}x{self.width}, got {unary.shape[1:3]}")
        ctx = _context(height, width, self.kernel_size, params=self.params)
        return ctx.crf_step_lr(unary, Q, f1, f2, f3, f4, val, want_potentials=want_potentials)

    __call__ = predict


class _Models(tuple):
    """The 4-tuple get_GPU_models() returns; carrying the settings lets find_surfaces()
    take the fused path."""
    mode = "Gauss"
    kernel_size = standard_kernel_size


def get_GPU_models(mode="LR", kernel_size=standard_kernel_size):
    """-> (smoothing_model, surface_model, normals_model, crf_model) (find_surfaces.py:26-32)."""
    size_x, size_y = int(width / div_x), int(height / div_y)
    if mode == "LR":
        crf = conv_crf_LR(*load_pixel_similarity_parameters(), *LR_CRF_parameters, kernel_size, size_y, size_x)
    else:
        crf = conv_crf_Gauss(*load_Gauss_parameters(), kernel_size, size_y, size_x)
    models = _Models((_SmoothingModel(kernel_size), _SurfaceModel(kernel_size), _NormalsModel(kernel_size), crf))
    models.mode, models.kernel_size = mode, kernel_size
    return models


# ---------------------------------------------------------------------------------------
# host glue of the operator path (kept because post_processing and tests call it)
# ---------------------------------------------------------------------------------------
def get_inputs(features, unary_potentials, Q, kernel_size, div_x, div_y, size_x, size_y):
    """Zero-pad features and Q by kernel_size and cut div_x * div_y tiles with halo, x outer
    and y inner; val is the one-hot of the last label per tile (find_surfaces.py:437-454)."""
    k = kernel_size
    pad = [(k, k), (k, k), (0, 0)]
    feats = [np.pad(f, pad) for f in features]
    Qp = np.pad(Q, pad)
    origins = [(ty * size_y, tx * size_x) for tx in range(div_x) for ty in range(div_y)]
    halo = lambda a: np.asarray([a[y0:y0 + size_y + 2 * k, x0:x0 + size_x + 2 * k] for y0, x0 in origins])
    unary = np.asarray([unary_potentials[y0:y0 + size_y, x0:x0 + size_x] for y0, x0 in origins])
    val = np.zeros((len(origins), 1, 1, 1, Qp.shape[-1]))
    val[..., -1] = 1
    return [unary, halo(Qp)] + [halo(f) for f in feats] + [val]


def assemble_outputs(outputs, div_x, div_y, size_x, size_y, height, width):
    """Inverse of the tiling of get_inputs (find_surfaces.py:487-495)."""
    outputs = np.asarray(outputs)
    Q = np.zeros((height, width) + outputs.shape[3:])
    tiles = iter(outputs)
    for tx in range(div_x):
        for ty in range(div_y):
            Q[ty * size_y:(ty + 1) * size_y, tx * size_x:(tx + 1) * size_x] = next(tiles)
    return Q


def get_unary_potentials_and_initial_probabilities(data):
    """unary [H,W,S], Q0 [H,W,S], prob [1,1,S] with S = num_surfaces + 1 (find_surfaces.py:456-477):
    no patch: unary (10,1,...,1), Q0 (0,1/ns,...); patch p: unary[p] = 0, Q0 = 0.1/(ns-1) except
    Q0[p] = 0.9, Q0[0] = 0.  Vectorised; float64 like the reference's arrays."""
    patches, ns = np.asarray(data["patches"]), int(data["num_surfaces"])
    H, W = patches.shape
    S = ns + 1
    unary = np.ones((H, W, S))
    unary[..., 0] = 10
    prob = np.ones((1, 1, S)) / ns
    prob[0, 0, 0] = 0
    Q0 = np.tile(prob, (H, W, 1))
    ys, xs = np.nonzero(patches)
    if len(ys):
        lab = patches[ys, xs]
        unary[ys, xs, lab] = 0
        # row of a patch pixel, built in the reference's order: fill, 0.9 at the label, then 0 at index 0
        # (the last write wins when the label is 0, which nonzero() excludes)
        rows = np.full((len(ys), S), 1.0 / (ns - 1) * 0.1)
        rows[np.arange(len(ys)), lab] = 0.9
        rows[:, 0] = 0
        Q0[ys, xs] = rows
    return unary, Q0, prob


def extract_features(depth_image, lab_image, angles, grid):
    """f1 = (row, col, depth), f2 = f1 + colour, f3 = f1 + angles (find_surfaces.py:479-484)."""
    f1 = np.concatenate([grid, np.expand_dims(depth_image, axis=-1)], axis=-1)
    return f1, np.concatenate([f1, lab_image], axis=-1), np.concatenate([f1, angles], axis=-1)


def _grid(h, w):
    cols, rows = np.meshgrid(np.arange(0, w), np.arange(0, h))
    return np.stack([rows, cols], axis=-1)          # [..., 0] = row, [..., 1] = column (find_surfaces.py:500-501)


# ---------------------------------------------------------------------------------------
# the stage
# ---------------------------------------------------------------------------------------
def _find_surfaces_operators(models, data, kernel_size, mode="Gauss"):
    """The body of the reference's find_surfaces() (find_surfaces.py:499-532) over the four
    model objects, tile by tile."""
    smoothing_model, surface_model, normals_model, crf_model = models
    h, w = data["depth"].shape
    size_x, size_y = int(w / div_x), int(h / div_y)
    grid = _grid(h, w)
    smoothed = smoothing_model(data["depth"], grid)
    data["angles"], data["vectors"], data["points_3d"] = normals_model(smoothed)
    if mode == "LR":
        features = [np.expand_dims(data["depth"], axis=-1), data["angles"], data["rgb"], grid]     # find_surfaces.py:508-509
    else:
        features = extract_features(data["depth"], data["rgb"], data["angles"], grid)
    data["patches"], data["depth_edges"] = surface_model(smoothed)
    data["num_surfaces"] = int(np.max(data["patches"]) + 1)
    unary, Q, prob = get_unary_potentials_and_initial_probabilities(data)
    for it in range(3):
        inputs = get_inputs(features, unary, Q, kernel_size, div_x, div_y, size_x, size_y)
        out = crf_model.predict(inputs, batch_size=1)
        Q = assemble_outputs(out, div_x, div_y, size_x, size_y, h, w)
        Q[data["depth"] == 0] = prob if it < 2 else 0
    data["surfaces"] = np.argmax(Q, axis=-1)
    data["surfaces"][data["surfaces"] == data["num_surfaces"]] = 0
    return data


def find_surfaces(models, image_data, mode, kernel_size, plot_result=False, save_index=-1):
    """One image through the stage (find_surfaces.py:497-546).  image_data = (depth uint16 [H,W],
    rgb uint8 [H,W,3], annotation); returns the dict with keys depth, rgb, annotation, angles,
    vectors, points_3d, patches, depth_edges, num_surfaces, surfaces and, with save_index >= 0,
    writes out/{save_index}/*.npy."""
    data = {}
    data["depth"], data["rgb"], data["annotation"] = image_data
    if isinstance(models, _Models):
        depth = np.asarray(data["depth"])
        ctx = _context(depth.shape[0], depth.shape[1], kernel_size, params=models[3].params)
        res = ctx.find_surfaces_batch(depth, np.asarray(data["rgb"]))
        for k in ("angles", "vectors", "points_3d", "patches", "depth_edges", "num_surfaces", "surfaces"):
            data[k] = res[k]
    else:
        _find_surfaces_operators(models, data, kernel_size, mode)
    if plot_result:
        from .plot_image import plot_surfaces
        plot_surfaces(data["surfaces"])
    if save_index >= 0:
        _li.save_surface_information(save_index, data)
    return data


io_workers = min(16, os.cpu_count() or 1)                 # threads decoding PNGs / writing .npy files
timings = {"decode": 0.0, "gpu": 0.0, "save": 0.0, "images": 0}   # seconds spent by find_surfaces_for_indices, by phase


def find_surfaces_for_indices(image_indices, mode="Gauss", kernel_size=standard_kernel_size, save_data=True, plot_result=False,
                              return_results=False):
    """find_surfaces for the images of the data set with the given indices
    (find_surfaces.py:551-561).  Frames go to the GPU frames_per_call at a time; PNG decoding and the
    .npy writes run on a thread pool (both release the GIL) and are accounted for in `timings`."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    models = get_GPU_models(mode, kernel_size)
    image_indices = list(image_indices)
    results = []
    with ThreadPoolExecutor(max_workers=io_workers) as pool:
        pending_saves = []
        starts = list(range(0, len(image_indices), frames_per_call))
        decoding = {}                                       # chunk start -> futures; the next chunk decodes during this one's GPU call
        for lo in starts[:1]:
            decoding[lo] = [pool.submit(load_image, i) for i in image_indices[lo:lo + frames_per_call]]
        for lo in starts:
            chunk = image_indices[lo:lo + frames_per_call]
            t0 = time.perf_counter()
            nxt = lo + frames_per_call
            if nxt < len(image_indices):
                decoding[nxt] = [pool.submit(load_image, i) for i in image_indices[nxt:nxt + frames_per_call]]
            images = [f.result() for f in decoding.pop(lo)]
            shapes = {im[0].shape for im in images}
            if len(shapes) != 1:
                raise ValueError(f"images of one call must share a size, got {sorted(shapes)}")
            h, w = shapes.pop()
            ctx = _context(h, w, kernel_size, max_batch=min(frames_per_call, len(image_indices)), params=models[3].params)
            # frames are gathered in page-locked memory so that the uploads are DMA transfers
            depth = _lib.pinned_empty((len(images), h, w), np.uint16)
            rgb = _lib.pinned_empty((len(images), h, w, 3), np.uint8)
            for j, im in enumerate(images):
                depth[j], rgb[j] = im[0], im[1]
            t1 = time.perf_counter()
            res = ctx.find_surfaces_batch(depth, rgb)
            t2 = time.perf_counter()
            for j, index in enumerate(chunk):
                print(index)
                data = {"depth": images[j][0], "rgb": images[j][1], "annotation": images[j][2]}
                for k in ("angles", "vectors", "points_3d", "patches", "depth_edges", "surfaces"):
                    data[k] = res[k][j]
                data["num_surfaces"] = int(res["num_surfaces"][j])
                if plot_result:
                    from .plot_image import plot_surfaces
                    plot_surfaces(data["surfaces"])
                if save_data:
                    pending_saves.append(pool.submit(_li.save_surface_information, index, data))
                if return_results:
                    results.append(data)
            timings["decode"] += t1 - t0
            timings["gpu"] += t2 - t1
            timings["images"] += len(chunk)
        t3 = time.perf_counter()
        for f in pending_saves:
            f.result()                          # re-raises a failed write
        timings["save"] += time.perf_counter() - t3
    if return_results:
        return results


def find_surface_model(mode="Gauss", kernel_size=standard_kernel_size):
    """-> callable(image_data) -> dict, models built once (find_surfaces.py:563-565)."""
    models = get_GPU_models(mode, kernel_size)
    return lambda x: find_surfaces(models, x, mode, kernel_size, False, -1)


if __name__ == "__main__":
    find_surfaces_for_indices(list(range(0, 111)), mode="Gauss", plot_result=False, save_data=True)
