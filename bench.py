#!/usr/bin/env python3
"""find_surfaces throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the whole find_surfaces stage (depth smoothing, normals,
angle filter, curvature mask, majority filters, depth edges, connected
components, three mean-field CRF updates, argmax) over one batch of frames whose
inputs are already resident in HBM.  The N=1 workload is the reference's OSD set
(111 frames, 640x480: `main.find_surfaces_for_dataset` of the reference, main.py:25-26)
when data/osd is present, otherwise seeded synthetic frames of the same shape.
With N > 1 every rank processes its own batch of the same size (weak scaling, no
data-path collective; frames are independent units).

Prints one JSON line with the whole-job images/s, the HBM roofline position of
the dominant kernel (the CRF mean-field update) measured with HIP events inside
the timed region, and a CPU baseline (the C oracle, OpenMP) timed on a bounded
sample of the same frames on the host cores of this box.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, spec
VALU_PEAK_TFLOPS = 157.3     # fp32 vector (packed FMA), 256 CUs x 128 lanes x 2 x 2.4 GHz


def load_frames(workload, count, rank):
    """-> depth uint16 [n,H,W], rgb uint8 [n,H,W,3], description"""
    if workload == "osd":
        from PIL import Image
        base = os.path.join(ROOT, "data", "osd")
        if os.path.isdir(os.path.join(base, "disparity")):
            idx = [(i + 7 * rank) % 111 for i in range(count)]
            d = np.stack([np.asarray(Image.open(os.path.join(base, "disparity", f"{i}.png"))) for i in idx])
            c = np.stack([np.asarray(Image.open(os.path.join(base, "RGB", f"{i}.png"))) for i in idx])
            return d.astype(np.uint16), c.astype(np.uint8), "OSD RGB-D frames (data/osd), 640x480"
        workload = "synthetic640"
    from imagesegmentation_amd.synth import make_batch
    H, W = (960, 1280) if workload == "synthetic1280" else (480, 640)
    d, c = make_batch([10000 * rank + i for i in range(count)], H, W)
    return d, c, f"synthetic seeded scenes {W}x{H}"


def cpu_baseline(depth, rgb, budget_s=12.0):
    """C oracle (oracle/fs_oracle.c, OpenMP over pixels) on a bounded sample."""
    from oracle import oracle as O
    O.build()
    params = O.gauss_params()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    O.find_surfaces(depth[0], rgb[0], params)   # warm-up (page-in, thread pool)
    t0, done = time.perf_counter(), 0
    while done < len(depth) and (time.perf_counter() - t0 < budget_s or done < 2):
        O.find_surfaces(depth[done], rgb[done], params)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"first {done} frames of the workload, C oracle (fs_oracle.c) with OpenMP, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="osd", choices=["osd", "synthetic640", "synthetic1280"])
    ap.add_argument("--frames", type=int, default=0, help="frames per step and GPU (default 111; 16 for 1280x960)")
    ap.add_argument("--kernel-size", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--host-buffers", action="store_true",
                    help="time the host-pointer entry point instead (inputs and the 6 output arrays cross PCIe); "
                         "a side measurement, not the headline value")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    frames = args.frames or (16 if args.workload == "synthetic1280" else 111)
    depth, rgb, data_desc = load_frames(args.workload, frames, rank)
    n, H, W = depth.shape

    from imagesegmentation_amd._lib import Context
    ctx = Context(H, W, max_batch=n, device=local_rank, kernel_size=args.kernel_size)
    # one stream for torch's copies and the library's kernels: a side stream made current, handed to the library
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_angles = torch.empty((n, H, W, 2), dtype=torch.float32, device=dev)
    d_vectors = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev)
    d_points = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev)
    d_patches = torch.empty((n, H, W), dtype=torch.int64, device=dev)
    d_edges = torch.empty((n, H, W), dtype=torch.int32, device=dev)
    d_surfaces = torch.empty((n, H, W), dtype=torch.int64, device=dev)

    host_out = [None]
    if args.host_buffers:
        # a caller after throughput keeps its frames in page-locked memory (imagesegmentation_amd._lib.pinned_empty)
        from imagesegmentation_amd._lib import pinned_empty
        pd, pc = pinned_empty(depth.shape, np.uint16), pinned_empty(rgb.shape, np.uint8)
        pd[...] = depth
        pc[...] = rgb
        depth, rgb = pd, pc

    def step():
        if args.host_buffers:
            host_out[0] = ctx.find_surfaces_batch(depth, rgb, out=host_out[0])     # page-locked outputs, reused
            return host_out[0]["num_surfaces"]
        return ctx.find_surfaces_batch_dev(n, d_depth.data_ptr(), d_rgb.data_ptr(), d_angles.data_ptr(), d_vectors.data_ptr(),
                                           d_points.data_ptr(), d_patches.data_ptr(), d_edges.data_ptr(), d_surfaces.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        ns = step()
    barrier()
    ITER = 4
    if not args.no_kernel_timing:
        ctx.get_timing(ITER)            # reset
        ctx.set_timing(1 << ITER)       # events only around the dominant kernel
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ns = step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.set_timing(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        S = ns.astype(np.int64) + 1                       # label channels per frame
        N = H * W
        out = {
            "metric": "find_surfaces images/sec @640x480" if (H, W) == (480, 640) else f"find_surfaces images/sec @{W}x{H}",
            "value": world * n * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": data_desc + (" (host buffers: PCIe copies inside the timed region)" if args.host_buffers else ""),
            "config": {"workload": f"find_surfaces, {n} frames/step/GPU, {W}x{H}, kernel_size {args.kernel_size}, 3 mean-field iterations",
                       "label_channels_min_median_max": [int(S.min()), int(np.median(S)), int(S.max())],
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
        }
        if not args.no_kernel_timing:
            ms, launches = ctx.get_timing(ITER)
            # Unit of the roofline figures: one mean-field update of one frame (launches are batched
            # over up to 128 frames and split by channel class, so the per-frame update is the stable unit).
            # Algorithmic bytes of one update of one frame: 24 B features + 4 B patch label + 4S read Q +
            # 4S write Q per pixel; the last update writes an int64 label instead of Q.  Three updates per frame.
            per_frame = N * ((28 + 8 * S) * 2 + (28 + 4 * S + 8))
            updates = 3 * n * args.steps
            total_bytes = float(per_frame.sum()) * args.steps
            achieved = total_bytes / (ms * 1e-3) / 1e9
            # arithmetic the update needs: 224 neighbours x (about 45 flop similarity + 2S flop message) per pixel
            kk = (2 * args.kernel_size + 1) ** 2 - 1
            flops = float((N * kk * (45 + 2 * S)).sum()) * 3 * args.steps
            traffic = None
            # HBM bytes per update from the committed counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
            # collected separately from this run: counters cannot share a run with the timing)
            import glob
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            if tfiles and args.workload == "osd" and args.kernel_size == 7:
                with open(tfiles[-1]) as f:
                    traffic = json.load(f)["hbm_bytes_per_frame_update"]
            out["roofline"] = {"bound": "hbm", "kernel": "crf_iter_fused_kernel (mean-field update, similarity evaluated in the kernel)",
                               "per": "one mean-field update of one frame",
                               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": traffic,
                               "algorithmic_bytes": total_bytes / updates,
                               "avg_us": 1e3 * ms / updates, "timed_intervals": int(launches),
                               "share_of_step_time": ms * 1e-3 / dt,
                               "valu": {"achieved": flops / (ms * 1e-3) / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": flops / (ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                                        "note": "algorithmic fp32 flops; the kernel is bound by vector ALU issue, not HBM"}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(depth, rgb)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
