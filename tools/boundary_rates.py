#!/usr/bin/env python3
"""Host wall time per frame of the per-frame entry points a caller of the reference's evaluation path goes through
(evaluate_models.py:70-94): find_surface_model()(x), the statistics mirror, the two chi-squared models, post-processing
apply_CRF and the IoU counts; numpy arrays in and out, a dozen OSD frames, results released between calls.

    python tools/boundary_rates.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from imagesegmentation_amd import evaluate_models as E
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import image_processing_models_GPU as M
    from imagesegmentation_amd import post_processing as P
    from imagesegmentation_amd import process_surfaces as ps
    from imagesegmentation_amd.load_images import load_image
    frames = [load_image(i) for i in range(12)]
    model = fs.find_surface_model()
    color, angle = M.chi_squared_distances_model_2D((10, 10), (4, 4)), M.chi_squared_distances_model_2D((4, 4), (1, 1))
    post = P.get_postprocessing_model()
    t = {"find_surfaces": 0.0, "statistics": 0.0, "chi_squared": 0.0, "apply_CRF": 0.0, "IoU": 0.0}
    reps = 3
    for rep in range(reps + 1):
        if rep == 1:
            t = dict.fromkeys(t, 0.0)          # the first round warms up (page-locked blocks, work space)
        for f in frames:
            t0 = time.perf_counter()
            data = model(f)
            t1 = time.perf_counter()
            ps.extract_information_from_surface_data_and_preprocess_surfaces(data, None)
            t2 = time.perf_counter()
            color(data["hist_color"]); angle(data["hist_angle"])
            t3 = time.perf_counter()
            data["final_surfaces"] = data["surfaces"]
            post(data)
            t4 = time.perf_counter()
            E.extract_IoU_information(f[2].astype("int64"), data["final_surfaces"])
            t5 = time.perf_counter()
            for k, d in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                t[k] += d
            del data
    n = reps * len(frames)
    print(json.dumps({k + "_ms": round(1e3 * v / n, 3) for k, v in t.items()}))


if __name__ == "__main__":
    main()
