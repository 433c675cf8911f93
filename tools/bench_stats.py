#!/usr/bin/env python3
"""`python bench.py --stage stats`: the per-surface statistics stage (process_surfaces.py:672-689 of the reference)
on the device-resident outputs of the find_surfaces stage, batched over the frames.

One step = fs_surface_stats_batch_dev over the batch (colour conversion from uint8 RGB included, no texture map:
the reference's ResNet feature extractor is not part of this build), inputs and outputs in HBM; the call returns after
its single synchronisation, so the rate is what a host caller sees.  The line also carries the same stage through the
per-frame host-pointer mirror (process_surfaces.extract_information_from_surface_data_and_preprocess_surfaces) and
the C oracle's rate on one host core for a few frames.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def measure_batched(ctx, torch, dev, ns, tensors, H, W, steps, warmup):
    """The batched statistics stage on device-resident outputs of find_surfaces (tensors: depth-independent inputs
    d_rgb, d_angles, d_points, d_patches, d_edges, d_surfaces) -> (seconds for `steps` steps, algorithmic bytes per frame)."""
    n, N = len(ns), H * W
    d_rgb, d_angles, d_points, d_patches, d_edges, d_surfaces = tensors
    t = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    el = [ctx.stats_output_elements(H, W, int(s)) for s in ns]
    dts = {"surfaces": torch.int64, "patch_point_offsets": torch.int32, "patch_points": torch.float32, "avg_pos": torch.float64,
           "counts": torch.float64, "centroids": torch.float32, "centroid_indices": torch.int32, "hist_color": torch.float64,
           "hist_angle": torch.float64, "avg_angles": torch.float64, "neighbor_counts": torch.float64, "border_centers": torch.float64,
           "norm": torch.float32, "depth_extend": torch.float32}
    outs = {k: t((sum(e[k] for e in el),), dt) for k, dt in dts.items()}
    ptrs = {k: v.data_ptr() for k, v in outs.items()}

    def step():
        ctx.surface_stats_batch_dev(ns, d_surfaces.data_ptr(), d_patches.data_ptr(), d_edges.data_ptr(), d_points.data_ptr(),
                                    d_angles.data_ptr(), d_rgb=d_rgb.data_ptr(), **ptrs)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    S = ns.astype(np.int64)
    # algorithmic bytes per frame: the seven input images once (int64 labels x2, int32 edges, 3-D points, angles, uint8 colours),
    # the three image outputs (int64 labels, float2 point lists, float norm) and the float64 histograms [256][256][S] + [40][40][S]
    alg = N * (8 + 8 + 4 + 12 + 8 + 3) + N * (8 + 8 + 4) + 8.0 * (65536 + 1600) * S
    del outs
    return dt, alg


def stats_stage_summary(ctx, torch, dev, ns, tensors, H, W, steps=5, warmup=2):
    """Short form for the default bench line: rate and roofline position of the batched statistics stage."""
    dt, alg = measure_batched(ctx, torch, dev, ns, tensors, H, W, steps, warmup)
    n = len(ns)
    return {"metric": "surface statistics frames/sec (process_surfaces.py:672-689, batched, HBM-resident, uint8 colours converted on the device)",
            "value": n * steps / dt, "unit": "frames/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "roofline": {"bound": "hbm", "per": "one frame", "achieved": float(alg.sum()) * steps / dt / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": float(alg.sum()) * steps / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": float(alg.mean()),
                         "note": "host wall time of the batched call (15 kernels, one synchronisation), not a kernel duration"}}


def main(args):
    from bench import load_frames
    want_cpu = not args.no_cpu_baseline
    if want_cpu:
        from oracle import oracle as O
        O.build()
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    frames = args.frames or 111
    depth, rgb, data_desc = load_frames(args.workload, frames, 0, 1, args.labels)
    n, H, W = depth.shape
    N = H * W
    from imagesegmentation_amd._lib import Context
    ctx = Context(H, W, max_batch=n, device=0, kernel_size=args.kernel_size)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    t = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    d_depth, d_rgb = torch.from_numpy(depth.view(np.int16)).to(dev), torch.from_numpy(rgb).to(dev)
    d_angles, d_vectors, d_points = t((n, H, W, 2), torch.float32), t((n, H, W, 3), torch.float32), t((n, H, W, 3), torch.float32)
    d_patches, d_edges, d_surfaces = t((n, H, W), torch.int64), t((n, H, W), torch.int32), t((n, H, W), torch.int64)
    ns = ctx.find_surfaces_batch_dev(n, d_depth.data_ptr(), d_rgb.data_ptr(), d_angles.data_ptr(), d_vectors.data_ptr(), d_points.data_ptr(),
                                     d_patches.data_ptr(), d_edges.data_ptr(), d_surfaces.data_ptr())
    torch.cuda.synchronize(dev)
    dt, alg = measure_batched(ctx, torch, dev, ns, (d_rgb, d_angles, d_points, d_patches, d_edges, d_surfaces), H, W, args.steps, args.warmup)
    S = ns.astype(np.int64)
    out = {"metric": "surface statistics frames/sec @640x480" if (H, W) == (480, 640) else f"surface statistics frames/sec @{W}x{H}",
           "value": n * args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32/f64",
           "data": data_desc,
           "config": {"workload": f"{args.workload}: per-surface statistics of {n} frames/step, {W}x{H}, inputs = device-resident outputs of "
                                  "find_surfaces, uint8 colours converted on the device, no texture map",
                      "num_surfaces_min_median_max": [int(S.min()), int(np.median(S)), int(S.max())]},
           "roofline": {"bound": "hbm", "kernel": "whole statistics stage (15 batched kernels, the largest: histogram accumulation and "
                                                  "the float64 histogram output)",
                        "per": "one frame", "achieved": float(alg.sum()) * args.steps / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": float(alg.sum()) * args.steps / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes": float(alg.mean()), "avg_us": 1e6 * dt / (n * args.steps),
                        "note": "host wall time of the batched call (one synchronisation per call), not a kernel duration"}}
    if args.no_host_leg:            # (a kernel trace of the batched stage alone)
        print(json.dumps(out), flush=True)
        ctx.close()
        return
    # the per-frame host-pointer mirror, as the reference's callers use the function
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd import process_surfaces as ps
    model = fs.find_surface_model("Gauss", args.kernel_size)
    datas = [model((depth[i], rgb[i], None)) for i in range(min(n, 8))]
    for dct in datas:               # buffers (device work space, page-locked blocks) grow to the frames' sizes once
        ps.extract_information_from_surface_data_and_preprocess_surfaces(dict(dct), None)
    t1 = time.perf_counter()
    for dct in datas:
        # one image at a time, its results dropped before the next one (the reference's loops, e.g. main.py:25-26): the
        # page-locked output blocks of a frame then serve the next frame
        ps.extract_information_from_surface_data_and_preprocess_surfaces(dict(dct), None)
    out["value_host_mirror"] = {"value": len(datas) / (time.perf_counter() - t1), "unit": "frames/s",
                                "note": "process_surfaces.extract_information_from_surface_data_and_preprocess_surfaces per frame: numpy "
                                        "arrays in, numpy arrays out (PCIe both ways, S x S tails in numpy), each frame's results "
                                        "released before the next call"}
    # the same function over a list of frames, two calls in flight (results kept until the list is done)
    many = [datas[i % len(datas)] for i in range(24)]
    for _ in range(4):              # (the workers' contexts and the page-locked blocks grow to the frames' sizes in the first rounds)
        frames = [dict(dct) for dct in many]
        t1 = time.perf_counter()
        ps.extract_information_for_frames(frames, None, workers=2)
        dt_list = time.perf_counter() - t1
        del frames
    out["value_host_mirror_list"] = {"value": len(many) / dt_list, "unit": "frames/s",
                                     "note": "process_surfaces.extract_information_for_frames over 24 frames' dicts, two worker "
                                             "threads with a library context each: uploads of one frame beside downloads of another"}
    if want_cpu:
        from imagesegmentation_amd.utils import rgb_to_Lab
        os.environ["OMP_NUM_THREADS"] = "1"
        refs = [O.find_surfaces(depth[i], rgb[i], O.gauss_params()) for i in range(3)]
        t2 = time.perf_counter()
        for i, r in enumerate(refs):
            O.surface_stats(r["surfaces"], r["patches"], r["depth_edges"], r["points_3d"], r["angles"], rgb_to_Lab(rgb[i]), None, r["num_surfaces"])
        out["cpu_baseline"] = {"value": len(refs) / (time.perf_counter() - t2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "first 3 frames, C oracle (fs_oracle_stats.c) + numpy colour conversion, one core"}
    print(json.dumps(out), flush=True)
    ctx.close()
