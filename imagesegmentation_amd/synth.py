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


def make_mosaic_scene(seed, height=960, width=1280, n_patches=30, noise_sigma=0.5):
    """A scene with a prescribed number of smooth surfaces, so that the CRF runs with a chosen channel count
    S = n_patches + 2: a jittered grid of cells separated by gutters without depth; n_patches of the cells hold
    one smooth facet each (tilted planes, every third one a shallow cylinder) at well separated distances, the
    other cells stay empty.  Elliptic holes inside the facets and 0.3 % salt holes bring the invalid fraction to
    about a quarter.  Depth uint16 (0 = invalid) with Gaussian noise of noise_sigma (the sigma 2 of make_scene breaks
    every facet into several patches, and the patch count is then no longer the prescribed one), RGB = a base
    colour per facet + N(0, 8).  Whether a seed really yields n_patches patches is for the user to check (the
    golden fixtures assert it)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W = height, width
    gx = int(np.ceil(np.sqrt(n_patches * W / H)))
    gy = int(np.ceil(n_patches / gx))
    gutter = 6
    # jittered cell borders: columns per grid row, rows shared
    ys = np.linspace(0, H, gy + 1)
    ys[1:-1] += rng.uniform(-0.12, 0.12, gy - 1) * H / gy
    ys = np.round(ys).astype(int)
    depth = np.zeros((H, W), np.float32)
    rgb = rng.normal(90, 8, (H, W, 3)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cells = [(j, i) for j in range(gy) for i in range(gx)]
    keep = {cells[k] for k in rng.permutation(len(cells))[:n_patches]}
    z_levels = rng.permutation(np.linspace(900, 4400, len(cells)))
    for k, (j, i) in enumerate(cells):
        xs = np.linspace(0, W, gx + 1)
        xs[1:-1] += np.random.Generator(np.random.PCG64(seed * 1000 + j)).uniform(-0.12, 0.12, gx - 1) * W / gx
        xs = np.round(xs).astype(int)
        if (j, i) not in keep:
            continue
        y0, y1, x0, x1 = ys[j] + gutter, ys[j + 1] - gutter, xs[i] + gutter, xs[i + 1] - gutter
        yc, xc = (y0 + y1) / 2, (x0 + x1) / 2
        sl = np.s_[y0:y1, x0:x1]
        z = z_levels[k] + rng.uniform(-0.8, 0.8) * (xx[sl] - xc) + rng.uniform(-0.8, 0.8) * (yy[sl] - yc)
        if k % 3 == 2:      # shallow cylinder: sagitta of a few centimetres across the cell
            z = z - rng.uniform(20, 60) * (1 - ((xx[sl] - xc) / ((x1 - x0) / 2)) ** 2)
        depth[sl] = z
        rgb[sl] = rng.integers(20, 236, 3).astype(np.float32) + rng.normal(0, 8, z.shape + (3,)).astype(np.float32)
        # an elliptic hole strictly inside the cell (cannot split the facet)
        hy, hx = rng.uniform(0.1, 0.22) * (y1 - y0), rng.uniform(0.1, 0.22) * (x1 - x0)
        cy, cx = yc + rng.uniform(-0.15, 0.15) * (y1 - y0), xc + rng.uniform(-0.15, 0.15) * (x1 - x0)
        hole = ((yy[sl] - cy) / hy) ** 2 + ((xx[sl] - cx) / hx) ** 2 < 1
        depth[sl][hole] = 0
    valid = depth > 0
    if noise_sigma > 0:
        depth = depth + rng.normal(0, noise_sigma, (H, W)).astype(np.float32)
    depth = np.clip(depth, 500, 5000)
    depth_u16 = np.round(depth).astype(np.uint16)
    depth_u16[~valid | (rng.random((H, W)) < 0.003)] = 0
    return depth_u16, np.clip(np.round(rgb), 0, 255).astype(np.uint8)


def make_dense_scene(seed, height=960, width=1280, small=(15, 20), medium=(10, 12), noise_sigma=0.5):
    """A scene with hundreds of small surfaces close together, so that the CRF's tiles see many labels each (the mosaic
    scenes keep their facets in separate cells, and a tile sees four or five of them however many there are): a
    block of small[0] x small[1] facets at a pitch of 20 pixels (2-pixel borders without depth between them: the
    smallest facets the patch stage still turns into patches one for one) and a block of medium[0] x medium[1] facets
    at a pitch of 48 pixels, both at seeded positions on a tilted background plane with elliptic holes; every facet a
    tilted plane at a distance of its own.  With the defaults about 420 patches: 20 labels around a 32 x 16 tile of the
    dense block at a window radius of 7, 30 at 10, 50 at 15.  Depth uint16 (0 = invalid), RGB = a base colour per
    facet + N(0, 8)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W = height, width
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = 2500 + 0.3 * (xx - W / 2) + 0.2 * (yy - H / 2)
    rgb = rng.normal(90, 8, (H, W, 3)).astype(np.float32)
    blocks = []
    for (ny, nx), pitch, gap in ((small, 20, 2), (medium, 48, 3)):
        bh, bw = ny * pitch, nx * pitch
        for _ in range(1000):                                    # a position clear of the blocks placed so far
            y0, x0 = int(rng.integers(8, H - bh - 8)), int(rng.integers(8, W - bw - 8))
            if all(y0 + bh + 8 <= b[0] or b[1] + 8 <= y0 or x0 + bw + 8 <= b[2] or b[3] + 8 <= x0 for b in blocks):
                break
        else:
            raise ValueError("the blocks of facets do not fit the frame")
        blocks.append((y0, y0 + bh, x0, x0 + bw))
        depth[y0:y0 + bh, x0:x0 + bw] = 0
        levels = rng.permutation(np.linspace(800, 4600, ny * nx))
        for k in range(ny * nx):
            j, i = divmod(k, nx)
            sl = np.s_[y0 + j * pitch + gap:y0 + (j + 1) * pitch - gap, x0 + i * pitch + gap:x0 + (i + 1) * pitch - gap]
            depth[sl] = levels[k] + rng.uniform(-0.5, 0.5) * (xx[sl] - xx[sl].mean()) + rng.uniform(-0.5, 0.5) * (yy[sl] - yy[sl].mean())
            rgb[sl] = rng.integers(20, 236, 3).astype(np.float32) + rng.normal(0, 8, depth[sl].shape + (3,)).astype(np.float32)
    for _ in range(12):                                          # holes in the background plane
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        hy, hx = rng.uniform(20, 70), rng.uniform(20, 70)
        hole = ((yy - cy) / hy) ** 2 + ((xx - cx) / hx) ** 2 < 1
        for b in blocks:
            hole[max(b[0] - 8, 0):b[1] + 8, max(b[2] - 8, 0):b[3] + 8] = False
        depth[hole] = 0
    valid = depth > 0
    if noise_sigma > 0:
        depth = depth + rng.normal(0, noise_sigma, (H, W)).astype(np.float32)
    depth_u16 = np.round(np.clip(depth, 500, 5000)).astype(np.uint16)
    depth_u16[~valid | (rng.random((H, W)) < 0.003)] = 0
    return depth_u16, np.clip(np.round(rgb), 0, 255).astype(np.uint8)
