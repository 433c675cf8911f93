#!/usr/bin/env python3
"""Runs the per-surface statistics stage, the chi-squared models and the IoU counts a few times on OSD
frames (for rocprofv3 --kernel-trace --stats: tools/profile_round.sh prints nothing for them)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagesegmentation_amd import find_surfaces as fs                      # noqa: E402
from imagesegmentation_amd import process_surfaces as ps                   # noqa: E402
from imagesegmentation_amd import image_processing_models_GPU as M         # noqa: E402
from imagesegmentation_amd import evaluate_models as E                     # noqa: E402

rng = np.random.default_rng(0)
tex = rng.random((120, 160, 256), dtype=np.float32)
model = fs.find_surface_model("Gauss")
color, angle = M.chi_squared_distances_model_2D((10, 10), (4, 4)), M.chi_squared_distances_model_2D((4, 4), (1, 1))
t_stats = t_chi = t_iou = 0.0
indices = [2, 20, 22, 27, 30]
for i in indices:
    image = fs.load_image(i)
    data = model(image)
    t0 = time.perf_counter()
    ps.extract_information_from_surface_data_and_preprocess_surfaces(data, lambda rgb: tex)
    t1 = time.perf_counter()
    color(data["hist_color"]); angle(data["hist_angle"])
    t2 = time.perf_counter()
    E.extract_IoU_information(image[2].astype("int64"), data["surfaces"])
    t3 = time.perf_counter()
    t_stats += t1 - t0; t_chi += t2 - t1; t_iou += t3 - t2
n = len(indices)
print(f"host wall per frame (copies included): statistics {1e3 * t_stats / n:.2f} ms, chi-squared {1e3 * t_chi / n:.2f} ms, IoU {1e3 * t_iou / n:.2f} ms")
