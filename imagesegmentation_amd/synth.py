"""Seeded synthetic RGB-D scenes shaped like the OSD frames the reference runs on
(load_images.py:13-20: uint16 depth with 0 = invalid, uint8 RGB).

A scene is a tilted table plane plus 6-20 boxes / spheres / cylinders rendered
with a z-buffer in image space, Gaussian depth noise (sigma 2), about a quarter
of the pixels invalidated as connected blobs plus 1 % salt holes, and per-object
base colours with N(0, 8) noise.  Used by the tests (small frames) and by
bench.py (640x480 stream, 1280x960 stress frames).
"""
import numpy as np


def make_scene(seed, height=480, width=640, invalid_fraction=0.24):
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W = height, width
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    u, v = xx / W - 0.5, yy / H - 0.5
    # background wall + table plane (depth grows towards the top of the image)
    depth = np.full((H, W), rng.uniform(3500, 4800), np.float32) + 300 * u
    table = rng.uniform(1400, 2200) - v * rng.uniform(1200, 2200) + u * rng.uniform(-300, 300)
    horizon = rng.uniform(-0.25, 0.05)
    obj = np.zeros((H, W), np.int32)
    sel = v > horizon
    depth[sel] = table[sel]
    obj[sel] = 1
    n_obj = int(rng.integers(6, 21))
    colours = [rng.integers(30, 226, 3), rng.integers(30, 226, 3)]
    for k in range(n_obj):
        kind = int(rng.integers(0, 3))
        cx, cy = rng.uniform(-0.4, 0.4), rng.uniform(horizon, 0.42)
        sx, sy = rng.uniform(0.04, 0.16), rng.uniform(0.05, 0.2)
        z0 = float(table[min(H - 1, max(0, int((cy + 0.5) * H))), min(W - 1, max(0, int((cx + 0.5) * W)))]) - rng.uniform(30, 150)
        if kind == 0:      # box: two planar faces split at a vertical edge
            inside = (np.abs(u - cx) < sx) & (np.abs(v - cy) < sy)
            edge = cx + rng.uniform(-0.5, 0.5) * sx
            slope_l, slope_r = rng.uniform(200, 900), rng.uniform(200, 900)
            z = z0 + np.where(u < edge, (edge - u) * slope_l, (u - edge) * slope_r) - (v - cy) * rng.uniform(0, 200)
        elif kind == 1:    # sphere
            rr = ((u - cx) / sx) ** 2 + ((v - cy) / sx * H / W * 1.33) ** 2
            inside = rr < 1
            z = z0 - np.sqrt(np.clip(1 - rr, 0, 1)) * sx * 1500
        else:              # upright cylinder
            inside = (np.abs(u - cx) < sx) & (np.abs(v - cy) < sy)
            z = z0 - np.sqrt(np.clip(1 - ((u - cx) / sx) ** 2, 0, 1)) * sx * 1500
        vis = inside & (z < depth)
        depth[vis] = z[vis]
        obj[vis] = k + 2
        colours.append(rng.integers(20, 236, 3))
    depth += rng.normal(0, 2.0, (H, W)).astype(np.float32)
    depth = np.clip(depth, 500, 5000)
    # invalid pixels: low-frequency blobs + shadow bands at depth discontinuities + salt
    coarse = rng.normal(0, 1, (H // 16 + 2, W // 16 + 2)).astype(np.float32)
    blob = np.kron(coarse, np.ones((16, 16), np.float32))[:H, :W]
    for _ in range(2):  # cheap blur
        blob = (blob + np.roll(blob, 5, 0) + np.roll(blob, -5, 0) + np.roll(blob, 5, 1) + np.roll(blob, -5, 1)) / 5
    jump = np.abs(np.diff(depth, axis=1, prepend=depth[:, :1])) > 60
    for s in range(1, 7):
        jump |= np.roll(jump, 1, 1)
    thr = np.quantile(blob, 1 - max(0.0, invalid_fraction - 0.06))
    invalid = (blob > thr) | jump | (rng.random((H, W)) < 0.01)
    depth_u16 = np.round(depth).astype(np.uint16)
    depth_u16[invalid] = 0
    cols = np.asarray(colours, np.float32)
    rgb = cols[obj] + rng.normal(0, 8, (H, W, 3)).astype(np.float32)
    rgb = np.clip(np.round(rgb), 0, 255).astype(np.uint8)
    return depth_u16, rgb


def make_batch(seeds, height=480, width=640):
    d, c = zip(*(make_scene(s, height, width) for s in seeds))
    return np.stack(d), np.stack(c)
