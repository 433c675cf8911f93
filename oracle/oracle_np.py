"""numpy restatement of the reference's TensorFlow graphs, op for op.

TEST INFRASTRUCTURE ONLY (see oracle.py).  PARITY UNPINNED (see fs_oracle.c).

Purpose: an independent second restatement that follows the *graph* formulation
of the reference (extract_patches windows, gathers, tiles, broadcast multiplies)
instead of per-pixel loops, so that the C oracle can be cross-checked against
something written from the other direction.  Everything is float32; short
reductions are accumulated left to right with one vectorised add per term so
that the evaluation order matches the C oracle's.  Only usable on small frames
(the CRF materialises [h, w, k, k, S] windows exactly like the reference).
"""
import math

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

f32 = np.float32
VIEW_X = 62.0 / 180 * math.pi
VIEW_Y = 48.6 / 180 * math.pi


def _seq_sum(a, axis=-1):
    """left-to-right reduce_sum along one axis"""
    a = np.moveaxis(a, axis, -1)
    acc = np.zeros(a.shape[:-1], a.dtype)
    for j in range(a.shape[-1]):
        acc = acc + a[..., j]
    return acc


def divide_no_nan(a, b):
    a, b = np.broadcast_arrays(np.asarray(a, f32), np.asarray(b, f32))
    out = np.zeros(a.shape, f32)
    nz = b != 0
    out[nz] = a[nz] / b[nz]
    return out


def extract_patches_same(img, k):
    """tf.image.extract_patches(..., padding='SAME', stride 1) for [H,W,C] ->
    [H,W,k*k,C] in (row, col) window order with symmetric zero padding."""
    r = k // 2
    p = np.pad(img, [[r, r], [r, r], [0, 0]])
    w = sliding_window_view(p, (k, k), (0, 1))            # [H,W,C,k,k]
    return np.moveaxis(w, 2, -1).reshape(img.shape[0], img.shape[1], k * k, img.shape[2])


_TABLES = {}


def set_tan_tables(tanx, tany):
    """np.tan(float32) and libm tanf differ in the last bit on some arguments.
    Tests that want bit-for-bit agreement with the C oracle inject its tables
    (tanx[W], tany[H]) here; pass None to go back to np.tan."""
    _TABLES.clear()
    if tanx is not None:
        _TABLES[(len(tany), len(tanx))] = (np.asarray(tanx, f32), np.asarray(tany, f32))


def get_angle_arrays(H, W, view_x=VIEW_X, view_y=VIEW_Y):
    """image_processing_models_GPU.py:137-149"""
    if (H, W) in _TABLES:
        tx, ty = _TABLES[(H, W)]
        return np.tile(tx[None, :], [H, 1])[..., None], np.tile(ty[:, None], [1, W])[..., None]
    x_val, y_val = f32(view_x / 2), f32(view_y / 2)
    x_list = x_val * (np.arange(0, W, dtype=f32) / f32(W / 2 - 0.5) - f32(1))
    y_list = y_val * (np.arange(H - 1, -1, -1, dtype=f32) / f32(H / 2 - 0.5) - f32(1))
    x_array = np.tile(x_list[None, :], [H, 1])[..., None]
    y_array = np.tile(y_list[:, None], [1, W])[..., None]
    return np.tan(x_array).astype(f32), np.tan(y_array).astype(f32)


def smooth_depth(depth, size=3):
    """image_processing_models_GPU.py:110-135 with sigma_1 = sigma_2 = 0"""
    d = np.asarray(depth, f32)
    s = 2 * size + 1
    middle = (s * s - 1) // 2
    patches = extract_patches_same(d[..., None], s)          # [H,W,49,1]
    centers = patches[:, :, middle:middle + 1, :]
    indicator = np.where(patches < f32(0.001), f32(0), f32(1))
    div = divide_no_nan(patches, np.broadcast_to(centers, patches.shape))
    gauss = np.exp(-(np.square(div - f32(1)) * f32(0)) - f32(0)).astype(f32) * indicator
    sums = _seq_sum(gauss, axis=-2)
    val = _seq_sum(gauss * patches, axis=-2)
    out = divide_no_nan(val, sums)[..., 0] * np.where(d < f32(0.001), f32(0), f32(1))
    return out.astype(f32)


def angle_calculator(n):
    """image_processing_models_GPU.py:63-72, vectorised over leading axes"""
    v1 = n * np.asarray([0.0, 1.0, 1.0], f32)
    v2 = n * np.asarray([1.0, 0.0, 1.0], f32)
    n1 = np.sqrt(_seq_sum(v1 * v1))
    n2 = np.sqrt(_seq_sum(v2 * v2))
    a1 = np.arccos(np.clip(divide_no_nan(-v1[..., 2], n1), -1, 1)).astype(f32) * np.sign(v1[..., 1])
    a2 = np.arccos(np.clip(divide_no_nan(-v2[..., 2], n2), -1, 1)).astype(f32) * -np.sign(v2[..., 0])
    return np.stack([a1, a2], axis=-1).astype(f32)


def normals_model(D, pool_size=3, view_x=VIEW_X, view_y=VIEW_Y):
    """image_processing_models_GPU.py:151-182 -> (smoothed angles, normals, points, raw angles)"""
    D = np.asarray(D, f32)
    H, W = D.shape
    p = 2 * pool_size + 1
    ax, ay = get_angle_arrays(H, W, view_x, view_y)
    de = D[..., None]
    pts = np.concatenate([ax * de, ay * de, de], axis=-1)
    pad_1 = np.pad(pts, [[2, 0], [0, 0], [0, 0]])
    pad_2 = np.pad(pts, [[0, 2], [0, 0], [0, 0]])
    pad_3 = np.pad(pts, [[0, 0], [2, 0], [0, 0]])
    pad_4 = np.pad(pts, [[0, 0], [0, 2], [0, 0]])
    sub_1 = (pad_1 - pad_2)[1:-1, :, :]
    sub_2 = (pad_4 - pad_3)[:, 1:-1, :]
    a, b = sub_1, sub_2
    vec = np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                    a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                    a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)
    cond = -(vec[..., 2:3] > 0).astype(f32)
    vec = f32(2) * (np.broadcast_to(cond, vec.shape) * vec) + vec
    norm = np.sqrt(_seq_sum(vec * vec))[..., None]
    normals = divide_no_nan(vec, np.broadcast_to(norm, vec.shape))
    angles = angle_calculator(normals)
    # uniform_filter_with_depth_cutoff, :102-108
    middle = (p * p - 1) // 2
    patches = extract_patches_same(angles, p)                      # [H,W,49,2]
    dpatch = extract_patches_same(D[..., None], p)[..., 0]         # [H,W,49]
    mid = dpatch[..., middle:middle + 1]
    mult = np.where(np.abs(divide_no_nan(dpatch - mid, np.broadcast_to(mid, dpatch.shape))) > f32(0.3), f32(0), f32(1))
    sums = _seq_sum(mult)[..., None]
    val = _seq_sum(mult[..., None] * patches, axis=-2)
    smoothed = divide_no_nan(val, np.broadcast_to(sums, val.shape))
    return smoothed.astype(f32), normals.astype(f32), pts.astype(f32), angles


def label_components(mask):
    """Stand-in for tfa.image.connected_components: scipy.ndimage.label with the
    default (4-connected) structure numbers components in raster order."""
    from scipy import ndimage
    lab, n = ndimage.label(mask != 0)
    return lab.astype(np.int64), n


def surface_model(D, depth=4, threshold=0.007003343675404672 * 10.5, alpha=3, s1=2, s2=1, n1=11, n2=5,
                  component_threshold=20, view_x=VIEW_X, view_y=VIEW_Y, return_stages=False):
    """find_surfaces.py:229-288"""
    D = np.asarray(D, f32)
    H, W = D.shape
    img = D[..., None]
    ax, ay = get_angle_arrays(H, W, view_x, view_y)
    middle = img[1:-1, 1:-1, :]
    dfx = np.abs(ax[1:-1, :-2, :] - ax[1:-1, 2:, :]) * middle * f32(0.5)
    dfy = np.abs(ay[:-2, 1:-1, :] - ay[2:, 1:-1, :]) * middle * f32(0.5)
    df = np.concatenate([dfx, dfy], axis=-1)[depth - 1:-(depth - 1), depth - 1:-(depth - 1), :]
    shape = (H - 2 * depth, W - 2 * depth)
    curv = np.zeros(shape, f32)
    central_depths = img[depth:-depth, depth:-depth, :]
    central_points = np.concatenate([np.zeros((shape[0], shape[1], 2), f32), central_depths], axis=-1)
    for direction in [[1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1]]:
        cur = np.zeros_like(curv)
        prev_d = np.zeros_like(curv)
        prev_p = central_points
        for k in range(1, depth + 1):
            y0, x0 = depth - k * direction[0], depth - k * direction[1]
            new_p = np.concatenate([df * f32(k), img[y0:y0 + shape[0], x0:x0 + shape[1], :]], axis=-1)
            dif_1 = np.sqrt(_seq_sum(np.square(new_p - prev_p)))
            dif_2 = np.sqrt(_seq_sum(np.square(new_p - central_points)))
            score = divide_no_nan(dif_1 + prev_d, dif_2) - f32(1)
            cur = cur + score
            curv = curv + cur
            prev_d = dif_2
            prev_p = new_p
    pixels = (curv < f32(threshold)).astype(np.int32)
    pixels = np.pad(np.where(central_points[..., 2] > f32(0.001), pixels, 0), [[depth, depth], [depth, depth]])
    curvature_pixels = pixels.copy()
    pixels = np.where(extract_patches_same(pixels[..., None], 2 * s1 + 1)[..., 0].sum(-1) > n1, 1, 0)
    pixels = np.where(extract_patches_same(pixels[..., None], 2 * s2 + 1)[..., 0].sum(-1) > n2, 1, 0).astype(np.int32)
    sub_1, sub_2 = img[:-2, 1:-1, :], img[2:, 1:-1, :]
    sub_3, sub_4 = img[1:-1, 2:, :], img[1:-1, :-2, :]
    diffs = np.abs(np.concatenate([sub_3, sub_4, sub_1, sub_2], axis=-1) - middle)
    fac = np.concatenate([dfx, dfx, dfy, dfy], axis=-1) * f32(alpha)
    edges = np.pad(np.where(diffs > fac, 0, 1).min(-1), [[1, 1], [1, 1]]).astype(np.int32)
    pixels = edges * pixels
    comp, _ = label_components(pixels)
    counts = np.bincount(comp.ravel())
    pixels = pixels * np.where(counts > component_threshold, 1, 0)[comp]
    out, _ = label_components(pixels)
    if return_stages:
        return out, edges, {"curvature_scores": np.pad(curv, depth), "curvature_pixels": curvature_pixels}
    return out, edges


def get_unary_potentials_and_initial_probabilities(patches, num_surfaces):
    """find_surfaces.py:456-477 (float64, as the reference)"""
    H, W = patches.shape
    potential = np.ones((1, 1, num_surfaces + 1))
    potential[0, 0, 0] = 10
    unary = np.tile(potential, [H, W, 1])
    prob = np.ones((1, 1, num_surfaces + 1)) / num_surfaces
    prob[0, 0, 0] = 0
    prob_copy = prob.copy()
    q = np.tile(prob, [H, W, 1])
    for y in range(H):
        for x in range(W):
            if patches[y][x] != 0:
                pot = np.ones(num_surfaces + 1)
                pot[patches[y][x]] = 0
                pot[0] = 10
                unary[y][x] = pot
                pr = np.ones(num_surfaces + 1) / (num_surfaces - 1) * 0.1
                pr[patches[y][x]] = 0.9
                pr[0] = 0
                q[y][x] = pr
    return unary, q, prob_copy


def get_inputs(features, unary, Q, kernel_size, div_x, div_y, size_x, size_y):
    """find_surfaces.py:437-454"""
    fp = [np.pad(f, [[kernel_size] * 2, [kernel_size] * 2, [0, 0]]) for f in features]
    Q = np.pad(Q, [[kernel_size] * 2, [kernel_size] * 2, [0, 0]])
    fo = [list() for _ in fp]
    Q_in, un = [], []
    for x in range(div_x):
        for y in range(div_y):
            for i in range(len(fp)):
                fo[i].append(fp[i][y * size_y:y * size_y + size_y + 2 * kernel_size, x * size_x:x * size_x + size_x + 2 * kernel_size])
            Q_in.append(Q[y * size_y:y * size_y + size_y + 2 * kernel_size, x * size_x:x * size_x + size_x + 2 * kernel_size])
            un.append(unary[y * size_y:y * size_y + size_y, x * size_x:x * size_x + size_x])
    val = np.zeros((len(un), 1, 1, 1, Q.shape[-1]))
    val[:, 0, 0, 0, -1] = 1
    return [np.asarray(a) for a in [un, Q_in, *fo, val]]


def assemble_outputs(outputs, div_x, div_y, size_x, size_y, height, width):
    """find_surfaces.py:487-495"""
    Q = np.zeros((height, width, *np.shape(outputs)[3:]))
    index = 0
    for x in range(div_x):
        for y in range(div_y):
            Q[y * size_y:y * size_y + size_y, x * size_x:x * size_x + size_x, :] = outputs[index]
            index += 1
    return Q


def conv_crf_gauss_predict(inputs, params, kernel_size, want_potentials=False):
    """find_surfaces.py:291-367, batch_size=1 semantics (each tile on its own)."""
    unary_b, Q_b, f1_b, f2_b, f3_b, val_b = inputs
    (w_1, w_2, w_3, bias, t11, t21, t31, t12, t22, t32, t23, t33, weight) = [f32(v) for v in np.asarray(params).reshape(-1)]
    r, k = kernel_size, 2 * kernel_size + 1
    norms = [6, 2, 1 / 50, 1 / 40]
    outs, pots = [], []
    for b in range(len(unary_b)):
        unary, Q = np.asarray(unary_b[b], f32), np.asarray(Q_b[b], f32)
        feats = [np.asarray(f[b], f32) for f in (f1_b, f2_b, f3_b)]
        val = np.asarray(val_b[b], f32).reshape(-1)
        h, w, S = unary.shape

        def windows(a):                                             # [h,w,k,k,C]
            return np.moveaxis(sliding_window_view(a, (k, k), (0, 1)), 2, -1)

        mask = np.ones((k, k, 1), f32)
        mask[r, r, 0] = 0
        wq = windows(Q) * mask
        masks = [np.asarray(m, f32) for m in ([0, 0, 1], [0, 0, 1, 0, 0, 0], [0, 0, 1, 0, 0])]
        consts = [np.asarray(c, f32) for c in ([norms[3], norms[3], norms[0]],
                                               [norms[3], norms[3], norms[0], norms[2], norms[2], norms[2]],
                                               [norms[3], norms[3], norms[0], norms[1], norms[1]])]
        thetas = [np.square(np.asarray(t, f32)) for t in ([t11, t11, t12], [t21, t21, t22, t23, t23, t23],
                                                          [t31, t31, t32, t33, t33])]
        sims = []
        for f, m, c, th in zip(feats, masks, consts, thetas):
            wf = windows(f)
            fc = wf[:, :, r:r + 1, r:r + 1, :]
            feature = np.tile(fc, [1, 1, k, k, 1])
            div = np.power(fc, m)
            diff = divide_no_nan(wf - feature, np.broadcast_to((div + np.power(wf, m)) / f32(2), wf.shape)) * c
            sims.append(np.exp(-_seq_sum(np.square(diff) * th)).astype(f32))
        sim_sum = ((w_1 * sims[0] + w_2 * sims[1]) + w_3 * sims[2]) + bias
        sim_sum = np.concatenate([sim_sum.reshape(h, w, k * k), np.ones((h, w, 1), f32) * f32(0.00005)], axis=-1)
        wq = np.concatenate([wq.reshape(h, w, k * k, S), np.tile(val.reshape(1, 1, 1, S), [h, w, 1, 1])], axis=-2)
        messages = _seq_sum(sim_sum[..., None] * wq, axis=-2)
        matrix = (np.ones((S, S), f32) - np.eye(S, dtype=f32))
        compat = _seq_sum(matrix[None, None] * np.tile(messages[:, :, None, :], [1, 1, S, 1]))
        add = unary + weight * compat
        logits = -add
        e = np.exp(logits - logits.max(-1, keepdims=True)).astype(f32)
        outs.append(e / _seq_sum(e)[..., None])
        pots.append(add)
    return (np.asarray(outs), np.asarray(pots)) if want_potentials else np.asarray(outs)


def conv_crf_lr_predict(inputs, params, kernel_size, want_potentials=False):
    """find_surfaces.py:371-434 (conv_crf_LR), batch_size=1 semantics; params = coef[4], bias, weight, aux."""
    unary_b, Q_b, f1_b, f2_b, f3_b, f4_b, val_b = inputs
    c = [f32(v) for v in np.asarray(params).reshape(-1)]
    coef, bias, weight, aux = np.asarray(c[:4], f32), c[4], c[5], c[6]
    r, k = kernel_size, 2 * kernel_size + 1
    norms = [f32(6), f32(2), f32(1 / 50), f32(1 / 40)]
    outs, pots = [], []
    for b in range(len(unary_b)):
        unary, Q = np.asarray(unary_b[b], f32), np.asarray(Q_b[b], f32)
        feats = [np.asarray(f[b], f32) for f in (f1_b, f2_b, f3_b, f4_b)]
        val = np.asarray(val_b[b], f32).reshape(-1)
        h, w, S = unary.shape

        def windows(a):                                             # [h,w,k,k,C]
            return np.moveaxis(sliding_window_view(a, (k, k), (0, 1)), 2, -1)

        mask = np.ones((k, k, 1), f32)
        mask[r, r, 0] = 0
        wq = windows(Q) * mask
        sims = []
        for i, f in enumerate(feats):
            wf = windows(f)
            feature = np.tile(wf[:, :, r:r + 1, r:r + 1, :], [1, 1, k, k, 1])
            diff = wf - feature
            if i == 0:
                diff = divide_no_nan(diff, feature)
            sims.append(np.sqrt(_seq_sum(np.square(diff))).astype(f32) * norms[i])
        sim = np.stack(sims, axis=-1).reshape(h, w, k * k, 4)
        lr = bias + _seq_sum(coef * sim)
        concat = np.concatenate([lr, np.ones((h, w, 1), f32) * aux], axis=-1)
        wq = np.concatenate([wq.reshape(h, w, k * k, S), np.tile(val.reshape(1, 1, 1, S), [h, w, 1, 1])], axis=-2)
        messages = _seq_sum(concat[..., None] * wq, axis=-2)
        matrix = (np.ones((S, S), f32) - np.eye(S, dtype=f32))
        compat = _seq_sum(matrix[None, None] * np.tile(messages[:, :, None, :], [1, 1, S, 1]))
        add = unary + weight * compat
        logits = -add
        e = np.exp(logits - logits.max(-1, keepdims=True)).astype(f32)
        outs.append(e / _seq_sum(e)[..., None])
        pots.append(add)
    return (np.asarray(outs), np.asarray(pots)) if want_potentials else np.asarray(outs)


def find_surfaces(depth, rgb, params, kernel_size=7, div_x=4, div_y=4, view_x=VIEW_X, view_y=VIEW_Y):
    """find_surfaces.py:497-532, including the reference's 4x4 tiling."""
    depth = np.asarray(depth)
    H, W = depth.shape
    size_x, size_y = int(W / div_x), int(H / div_y)
    grid = np.meshgrid(np.arange(0, W), np.arange(0, H))
    grid = np.stack([grid[1], grid[0]], axis=-1)
    data = {"depth": depth, "rgb": rgb}
    smoothed = smooth_depth(depth)
    data["angles"], data["vectors"], data["points_3d"], _ = normals_model(smoothed, view_x=view_x, view_y=view_y)
    f1 = np.concatenate([grid, depth[..., None]], axis=-1)
    features = (f1, np.concatenate([f1, rgb], axis=-1), np.concatenate([f1, data["angles"]], axis=-1))
    data["patches"], data["depth_edges"] = surface_model(smoothed, view_x=view_x, view_y=view_y)
    data["num_surfaces"] = int(np.max(data["patches"]) + 1)
    unary, Q, prob = get_unary_potentials_and_initial_probabilities(data["patches"], data["num_surfaces"])
    for it in range(3):
        inputs = get_inputs(features, unary, Q, kernel_size, div_x, div_y, size_x, size_y)
        out = conv_crf_gauss_predict(inputs, params, kernel_size)
        Q = assemble_outputs(out, div_x, div_y, size_x, size_y, H, W)
        Q[depth == 0] = prob if it < 2 else 0
    data["surfaces"] = np.argmax(Q, axis=-1)
    data["surfaces"][data["surfaces"] == data["num_surfaces"]] = 0
    data["smoothed"] = smoothed
    return data


def chi_squared_distances(hist, pool_size=(1, 1), strides=(1, 1)):
    """image_processing_models_GPU.py:189-212 in float32: VALID average pooling of hist [A,B,S], then
    sum over bins of divide_no_nan((a-b)^2, a+b) for every pair of surfaces -> [S,S]."""
    h = np.asarray(hist, f32)
    if h.ndim == 2:
        h = h[None]
    A, B, S = h.shape
    ph, pw = pool_size
    sh, sw = strides
    win = sliding_window_view(h, (ph, pw), (0, 1))[::sh, ::sw]            # [PA,PB,S,ph,pw]
    pool = (win.reshape(win.shape[:3] + (-1,)).sum(-1, dtype=f32) / f32(ph * pw)).astype(f32)
    pool = np.moveaxis(pool, 2, 0)                                         # [S,PA,PB]
    e1, e2 = pool[None, :], pool[:, None]
    d = divide_no_nan(np.square(e1 - e2), e1 + e2)
    return d.sum(-1, dtype=f32).sum(-1, dtype=f32)


def iou_information(annotation, detected, threshold=0.2):
    """evaluate_models.py:19-58 with plain loops replaced by bincount."""
    a, o = np.asarray(annotation, np.int64), np.asarray(detected, np.int64)
    nd, na = int(o.max()) + 1, int(a.max()) + 1
    overlap = np.bincount((o * na + a).ravel(), minlength=nd * na).reshape(nd, na).astype(np.float64)
    cd, ca = overlap.sum(1), overlap.sum(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = overlap / (ca[None] + cd[:, None] - overlap)
        dov = overlap.sum(-1) / cd
    iou_max = np.zeros(na - 1)
    for i in range(na - 1):
        for j in range(nd):
            if iou[j, i + 1] > iou_max[i]:
                iou_max[i] = iou[j, i + 1]
    dov[np.isnan(dov)] = 0
    return float((ca != 0).sum()), float((dov > threshold).sum()), iou_max, overlap
