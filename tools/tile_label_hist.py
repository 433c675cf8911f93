"""Histogram of the number of labels listed around a tile (32 x 16 pixels, reach 3 r) for the dense 1280x960 scenes."""
import sys
import numpy as np
sys.path.insert(0, ".")
from imagesegmentation_amd._lib import Context
from imagesegmentation_amd.synth import make_dense_scene

for r in (7, 10, 15):
    counts = []
    with Context(960, 1280, max_batch=1, kernel_size=r) as ctx:
        for seed in (1, 2, 4, 5):
            d, c = make_dense_scene(seed, 960, 1280)
            ctx.find_surfaces_batch(d, c)
            counts.append(ctx.tile_labels(1)[0][0].ravel())
    a = np.concatenate(counts)
    edges = [0, 6, 10, 14, 30, 62, 126, 1000]
    h, _ = np.histogram(a, bins=edges)
    print(f"r={r}: tiles {a.size}, listed labels: " + ", ".join(f"[{edges[i]},{edges[i+1]-1}]: {h[i]} ({100*h[i]/a.size:.1f}%)" for i in range(len(h))),
          f"mean channels {np.mean(np.minimum(a+3, 500)):.2f}")
    for lo, hi in ((0, 5), (6, 9), (6, 13), (10, 61), (14, 61)):
        m = (a >= lo) & (a <= hi)
        print(f"   A in [{lo},{hi}]: {m.sum()} tiles, mean channels {np.mean(a[m] + 3):.1f}, mean walks of 8 {np.mean(np.ceil((a[m] + 3) / 8)):.2f}")
