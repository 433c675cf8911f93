#!/usr/bin/env python3
"""find_surfaces throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the whole find_surfaces stage (depth smoothing, normals, angle filter, curvature mask,
majority filters, depth edges, connected components, three mean-field CRF updates, argmax) over one batch of
frames.  Workloads (--workload):

  osd            the reference's OSD set (111 frames, 640x480: main.find_surfaces_for_dataset, main.py:25-26),
                 inputs and outputs resident in HBM.  The default, and the configuration the metric is quoted on.
  synthetic640   seeded synthetic scenes of the same shape (also the fallback when data/osd is missing)
  synthetic1280  seeded synthetic scenes 1280x960 with sensor-like noise (hundreds of label channels)
  mosaic1280     seeded 1280x960 scenes with a prescribed channel count S (--labels S, 16 / 32 / 64) for the larger
                 CRF windows (--kernel-size 7 / 10 / 15); the facets lie in separate cells, a tile sees few of them
  dense1280      seeded 1280x960 scenes with about 420 surfaces close together (20 to 50 labels around a tile)
  stream640      a stream of seeded synthetic 640x480 frames, frame id % n_gpu == rank, that enters and leaves
                 through page-locked host memory every step: PCIe copies are inside the timed region, so several
                 ranks on one node compete for the host, which an HBM-resident loop cannot show.

  --stage stats  times the per-surface statistics stage (process_surfaces.py:672-689) on the outputs of the
                 find_surfaces stage instead (HBM-resident, batched).

With N > 1 every rank processes its own batch of the same size (weak scaling, no data-path collective; frames are
independent units).  Prints one JSON line: whole-job images/s, the roofline position of the dominant kernel (the CRF
mean-field update) from HIP events recorded on the library's stream inside the timed region, the same stage timed
host buffer to host buffer, and two CPU baselines measured on this box's host cores on a bounded sample.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E
VALU_PEAK_TFLOPS = 157.3     # fp32 vector (packed FMA): 256 CUs x 128 lanes x 2 flop x 2.4 GHz
SIM_FLOPS = 33               # similarity of one pixel pair: 3 exponentials and 30 flops in adds, multiplies and fused multiply-adds


def load_frames(workload, count, rank, world, labels):
    """-> depth uint16 [n,H,W], rgb uint8 [n,H,W,3], description"""
    if workload == "osd":
        from imagesegmentation_amd.load_images import read_png
        base = os.path.join(ROOT, "data", "osd")
        if os.path.isdir(os.path.join(base, "disparity")):
            idx = [(i + 7 * rank) % 111 for i in range(count)]
            d = np.stack([read_png(os.path.join(base, "disparity", f"{i}.png")) for i in idx])
            c = np.stack([read_png(os.path.join(base, "RGB", f"{i}.png")) for i in idx])
            return d.astype(np.uint16), c.astype(np.uint8), "OSD RGB-D frames (data/osd), 640x480"
        workload = "synthetic640"
    from imagesegmentation_amd.synth import make_batch, make_dense_scene, make_mosaic_scene
    if workload == "dense1280":
        good = [1, 2, 4, 5, 6, 7, 8, 9, 11, 12, 13, 14, 15, 16, 17, 18]          # seeds whose blocks of facets find room in the frame
        d, c = zip(*(make_dense_scene(good[(rank + i) % len(good)], 960, 1280) for i in range(count)))
        return np.stack(d), np.stack(c), "synthetic seeded 1280x960 scenes with about 420 surfaces close together (make_dense_scene)"
    if workload == "mosaic1280":
        seeds = [2 + 100 * rank + i for i in range(count)]
        d, c = zip(*(make_mosaic_scene(s, 960, 1280, labels - 2) for s in seeds))
        return np.stack(d), np.stack(c), f"synthetic seeded mosaic scenes 1280x960 with {labels} label channels prescribed"
    if workload == "stream640":
        # frame ids rank, rank + world, ...: a pool of distinct frames that the stream cycles through
        pool = min(count, 32)
        d, c = make_batch([rank + world * j for j in range(pool)], 480, 640)
        sel = np.arange(count) % pool
        return d[sel], c[sel], (f"synthetic seeded 640x480 stream, frame id % {world} == rank, {pool} distinct frames per rank "
                                "cycled; inputs and the 6 output arrays cross PCIe every step")
    H, W = (960, 1280) if workload == "synthetic1280" else (480, 640)
    d, c = make_batch([10000 * rank + i for i in range(count)], H, W)
    return d, c, f"synthetic seeded scenes {W}x{H}"


def cpu_baseline(depth, rgb, r, budget_s=12.0):
    """B1, the strong CPU path: the C oracle (oracle/fs_oracle.c, OpenMP over pixels) on a bounded sample."""
    from oracle import oracle as O
    params = O.gauss_params()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    O.find_surfaces(depth[0], rgb[0], params, r=r)   # warm-up (page-in, thread pool)
    t0, done = time.perf_counter(), 0
    while done < len(depth) and (time.perf_counter() - t0 < budget_s or done < 2):
        O.find_surfaces(depth[done], rgb[done], params, r=r)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"first {done} frames of the workload, C oracle (fs_oracle.c) with OpenMP, {dt:.1f} s"}


def cpu_baseline_b0(depth, rgb, r):
    """B0, the reference-shaped CPU path: the numpy restatement that mirrors the reference's TensorFlow graph op for op
    (materialised extract_patches windows, 4x4 tiles, batch 1: find_surfaces.py:437-530), on a quarter-frame crop
    (a full 640x480 frame takes about two minutes); images/s = crop fraction / seconds."""
    from oracle import oracle as O
    from oracle import oracle_np as ON
    H, W = depth.shape[1:]
    h, w = H // 2 // 4 * 4, W // 2 // 4 * 4
    y0, x0 = (H - h) // 2, (W - w) // 2
    d, c = depth[0, y0:y0 + h, x0:x0 + w], rgb[0, y0:y0 + h, x0:x0 + w]
    t0 = time.perf_counter()
    ON.find_surfaces(d, c, O.gauss_params(), kernel_size=r)
    dt = time.perf_counter() - t0
    return {"value": (h * w) / (H * W) / dt, "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"centre {w}x{h} crop of frame 0 ({h * w / (H * W):.2f} frame), numpy restatement of the reference's graph "
                      f"(oracle_np.py) with its 4x4 tiling, {dt:.1f} s; elementwise numpy runs on one core"}


def executed_work(ctx, depth, S, r):
    """What the CRF update executed in the last call, from the per-tile label lists the library keeps: a tile with A
    listed labels carries CE = min(S, A + 3) channels (A + 2 when no unlisted label is left), in walks of the window whose
    similarity is evaluated once per walk: 8 channels per walk (4 or 8 slots multiplied), or, for the tiles that list 6 or
    more labels in a frame whose fullest tile lists more than 13 (r = 7, 10, 15 with the trained parameters:
    crf_update_plan in csrc/k_crf.hip), 16 channels per walk (12 at r = 15; 12 or 16, 8 or 12 slots multiplied).  Tiles
    without any valid depth are skipped.
    -> executed flops and compact Q bytes (planes a tile reads for its own pixels + planes it writes) per frame-update."""
    n, H, W = depth.shape
    listed, (th, tw) = ctx.tile_labels(n)
    if (listed < 0).any() or (listed > 512).any():
        raise RuntimeError("fs_get_tile_labels returned counts outside [0, 512]")
    ty, tx = listed.shape[1:]
    pad = np.zeros((n, ty * th, tx * tw), bool)
    pad[:, :H, :W] = depth != 0
    work = pad.reshape(n, ty, th, tx, tw).any(axis=(2, 4))                     # tiles with at least one valid depth
    Sf = S.reshape(n, 1, 1).astype(np.int64)
    A = listed.astype(np.int64)
    CE = np.minimum(Sf, A + 2 + ((Sf - 2 - A) > 0))
    chunks = (CE + 7) // 8
    staged = 8 * (chunks - 1) + np.where(CE - 8 * (chunks - 1) > 4, 8, 4)    # channel slots multiplied per neighbour
    if r in (7, 10, 15):
        quads = 3 if r == 15 else 4
        cw, lo = 4 * quads, 4 * (quads * 3 // 4)
        many = (A.max(axis=(1, 2), keepdims=True) > 13) & (A >= 6) & (A <= 4 * cw - 3)
        walks = (CE + cw - 1) // cw
        last = CE - cw * (walks - 1)                                           # channels of the last walk
        chunks = np.where(many, walks, chunks)
        staged = np.where(many, cw * (walks - 1) + np.where(last > lo, cw, lo), staged)
    kk = (2 * r + 1) ** 2 - 1
    px = th * tw
    flops = (work * px * kk * (SIM_FLOPS * chunks + 2 * staged)).sum() / n
    q_bytes = (work * px * 4 * CE * 2).sum() / n + H * W * 28.0            # read + write of the carried planes, features, patch label
    pct = [int(v) for v in np.percentile(CE[work], [50, 90, 99])] if work.any() else [0, 0, 0]
    info = {"tiles_per_frame": int(tx * ty), "tile": [th, tw], "tiles_skipped_frac": float(1 - work.mean()),
            "channels_per_tile_mean": float(CE[work].mean()) if work.any() else 0.0, "channels_per_tile_max": int(CE.max()),
            "channels_per_tile_p50_p90_p99": pct,
            "walks_per_tile_mean": float(chunks[work].mean()) if work.any() else 0.0}
    if r in (7, 10, 15):
        info["tiles_many_labels_frac"] = float((many & work).mean())          # share of the tiles that take the 12 / 16-channel walks
    return float(flops), float(q_bytes), info


def committed_counters(workload, r, labels=None):
    """HBM bytes per frame-update and VALU busy fraction from the counter passes committed under profiles/ (rocprofv3
    --pmc runs cannot share a run with the timing)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        doc = json.load(f)
    entry = doc.get("workloads", {}).get(f"{workload}_s{labels}_r{r}") or doc.get("workloads", {}).get(f"{workload}_r{r}")
    if entry is None and workload == "osd" and r == 7 and "hbm_bytes_per_frame_update" in doc:
        entry = doc
    if entry is None:
        return None, None, os.path.basename(files[-1])
    return entry.get("hbm_bytes_per_frame_update"), entry.get("valu_busy"), os.path.basename(files[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="osd", choices=["osd", "synthetic640", "synthetic1280", "mosaic1280", "dense1280", "stream640"])
    ap.add_argument("--stage", default="find_surfaces", choices=["find_surfaces", "stats"])
    ap.add_argument("--frames", type=int, default=0, help="frames per step and GPU (default 111; 16 for 1280x960; 100 for the stream)")
    ap.add_argument("--kernel-size", type=int, default=7)
    ap.add_argument("--labels", type=int, default=32, help="label channels S of the mosaic1280 frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-buffer to host-buffer measurement after the headline loop")
    ap.add_argument("--no-stats-leg", action="store_true", help="skip the statistics-stage measurement after the headline loop")
    ap.add_argument("--host-buffers", action="store_true",
                    help="time the host-pointer entry point as the main loop (what --workload stream640 implies)")
    args = ap.parse_args()
    if args.stage == "stats":
        from tools.bench_stats import main as stats_main
        return stats_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}")
    want_cpu = not args.no_cpu_baseline and world == 1 and rank == 0
    if want_cpu:
        from oracle import oracle as O
        O.build()                       # make runs as a child process: before anything touches the GPU

    # before anything touches the GPU: this rank's threads (and with them its page-locked buffers) go to the CPUs of the
    # NUMA node its GPU hangs on; N ranks that each move 17.5 MB per frame over PCIe otherwise meet on one socket
    from imagesegmentation_amd.sharding import pin_to_gpu_numa_node
    pinning = pin_to_gpu_numa_node(int(os.environ.get("FS_DEVICE", local_rank)))

    import torch
    import torch.distributed as dist
    # FS_DEVICE / FS_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with one card (all ranks share it and
    # synchronise over gloo; RCCL does not take two ranks on one GPU).  Unset on a real node: rank i -> GPU i over RCCL.
    device_index = int(os.environ.get("FS_DEVICE", local_rank))
    backend = os.environ.get("FS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    big = args.workload in ("synthetic1280", "mosaic1280", "dense1280")
    frames = args.frames or (16 if big else 100 if args.workload == "stream640" else 111)
    host_main = args.host_buffers or args.workload == "stream640"
    depth, rgb, data_desc = load_frames(args.workload, frames, rank, world, args.labels)
    n, H, W = depth.shape

    from imagesegmentation_amd._lib import Context, pinned_empty
    ctx = Context(H, W, max_batch=n, device=device_index, kernel_size=args.kernel_size)
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

    # a caller after throughput keeps its frames in page-locked memory
    pd, pc = pinned_empty(depth.shape, np.uint16), pinned_empty(rgb.shape, np.uint8)
    pd[...] = depth
    pc[...] = rgb
    host_out = [None]

    def step_host():
        host_out[0] = ctx.find_surfaces_batch(pd, pc, out=host_out[0])     # page-locked outputs, reused
        return host_out[0]["num_surfaces"]

    def step_dev():
        return ctx.find_surfaces_batch_dev(n, d_depth.data_ptr(), d_rgb.data_ptr(), d_angles.data_ptr(), d_vectors.data_ptr(),
                                           d_points.data_ptr(), d_patches.data_ptr(), d_edges.data_ptr(), d_surfaces.data_ptr())

    step = step_host if host_main else step_dev

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
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    host_leg = None
    if not host_main and not args.no_host_leg:
        # the same stage host buffer to host buffer (page-locked): inputs 1.5 MB and the six output arrays 15.8 MB per
        # 640x480 frame cross PCIe inside the timed region; with N > 1 every rank does so at the same time, which is
        # where ranks of one node compete (host memory, PCIe root complexes) while the HBM-resident loop cannot show it
        step_host()
        barrier()
        hs = max(2, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(hs):
            step_host()
        barrier()
        hdt_own = time.perf_counter() - t1                # this rank's own loop, before the barrier's wait for the slowest
        per_rank = [n * hs / hdt_own]
        if world > 1:
            t = torch.tensor([hdt_own], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank = [n * hs / float(e.item()) for e in every]
            hdt = max(float(e.item()) for e in every)
        else:
            hdt = hdt_own
        per_frame_bytes = H * W * (2 + 3) + H * W * (8 + 12 + 12 + 8 + 4 + 8)
        # what crosses the link: a call of more than 16 frames sends patches and surfaces as 16-bit and edges as 8-bit images
        # and widens them on the host (FS_NO_NARROW_DOWNLOADS=1: as they are)
        narrow = n > 16 and os.environ.get("FS_NO_NARROW_DOWNLOADS", "0")[:1] != "1" and (H * W) % 4 == 0
        link_bytes = H * W * (2 + 3) + H * W * (8 + 12 + 12 + (2 + 1 + 2 if narrow else 8 + 4 + 8))
        host_leg = {"value": world * n * hs / hdt, "unit": "images/s", "steps": hs, "ms_per_step": 1e3 * hdt / hs,
                    "bytes_per_frame": per_frame_bytes, "bytes_per_frame_over_pcie": link_bytes,
                    "pcie_gbs": hs * world * n * link_bytes / hdt / 1e9,
                    "per_rank_images_per_s": per_rank,
                    "note": "page-locked host arrays in, six page-locked output arrays back; measured after the headline loop on "
                            "all ranks at once (whole-job rate = frames of all ranks / the slowest rank's time; per_rank = each "
                            "rank's own rate while the others ran: where they differ, ranks compete for the host)"}

    stream_leg = None
    if world > 1 and host_leg is not None and (H, W) == (480, 640):
        # N > 1 only: the synthetic frame stream of the multi-GPU configuration (frame id % n_gpu == rank), host buffer to
        # host buffer like the leg above, so that a scaling run shows it beside the HBM-resident headline
        sd, sc, _ = load_frames("stream640", min(n, 32), rank, world, args.labels)
        psd, psc = pinned_empty(sd.shape, np.uint16), pinned_empty(sc.shape, np.uint8)
        psd[...] = sd
        psc[...] = sc
        sout = ctx.find_surfaces_batch(psd, psc)
        barrier()
        t2 = time.perf_counter()
        for _ in range(3):
            sout = ctx.find_surfaces_batch(psd, psc, out=sout)
        own = time.perf_counter() - t2
        t = torch.tensor([own], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        times = [float(e.item()) for e in every]
        stream_leg = {"value": world * len(sd) * 3 / max(times), "unit": "images/s", "frames_per_rank_and_step": int(len(sd)), "steps": 3,
                      "per_rank_images_per_s": [len(sd) * 3 / v for v in times],
                      "note": "seeded synthetic 640x480 stream, frame id % n_gpu == rank, page-locked host buffers in and out"}

    if rank == 0:
        S = ns.astype(np.int64) + 1                       # label channels per frame
        N = H * W
        r = args.kernel_size
        out = {
            "metric": "find_surfaces images/sec @640x480" if (H, W) == (480, 640) else f"find_surfaces images/sec @{W}x{H}",
            "value": world * n * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": data_desc + (" (host buffers: PCIe copies inside the timed region)" if host_main else ""),
            "config": {"workload": f"{args.workload}: find_surfaces, {n} frames/step/GPU, {W}x{H}, kernel_size {r}, 3 mean-field iterations"
                                   + (", page-locked host buffers in and out" if host_main else ", inputs and outputs resident in HBM"),
                       "label_channels_min_median_max": [int(S.min()), int(np.median(S)), int(S.max())],
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
        }
        if not args.no_kernel_timing:
            ms, launches = ctx.get_timing(ITER)
            # Unit of the roofline figures: one mean-field update of one frame (launches are batched over up to 128 frames
            # and split by channel class, so the per-frame update is the stable unit).
            # Algorithmic bytes of one update of one frame: 24 B features + 4 B patch label + 4S read Q + 4S write Q per
            # pixel; the last update writes an int64 label instead of Q.  Three updates per frame.
            per_frame = N * ((28 + 8 * S) * 2 + (28 + 4 * S + 8))
            updates = 3 * n * args.steps
            sec = ms * 1e-3
            total_bytes = float(per_frame.sum()) * args.steps
            achieved = total_bytes / sec / 1e9
            kk = (2 * r + 1) ** 2 - 1
            alg_flops = float((N * kk * (SIM_FLOPS + 2 * S)).sum()) * 3 * args.steps
            # executed work: what the kernel really carried (per-tile channel lists of the last step; of its last chunk
            # when the host-pointer path is the main loop)
            nt = ((n - 1) % 16 + 1) if host_main else n
            ex_flops, ex_bytes, tiles = executed_work(ctx, depth[n - nt:], S[n - nt:], r)
            us = 1e6 * sec / updates
            traffic, busy, src = committed_counters(args.workload, r, args.labels)
            out["roofline"] = {
                "bound": "hbm", "limiter": "valu",
                "kernel": (("crf_iter_fused_kernel<7> (mean-field update, similarity evaluated in the kernel, tile and halo in LDS)" if r == 7 else
                            f"crf_iter_roll_kernel<{r if r in (10, 15) else 0}> (mean-field update, similarity evaluated in the kernel, "
                            "window rolling through a 16-row ring in LDS)")
                           + (f" + crf_iter_roll_kernel<{r}, ..., {3 if r == 15 else 4}> for the tiles that list 6 or more labels "
                              f"({12 if r == 15 else 16} channels per walk)" if tiles.get("tiles_many_labels_frac", 0) > 0 else "")),
                "per": "one mean-field update of one frame",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes": total_bytes / updates,
                "avg_us": us, "timed_intervals": int(launches), "share_of_step_time": sec / dt,
                "hbm_measured": None if traffic is None else {"achieved": traffic / (us * 1e-6) / 1e9, "unit": "GB/s",
                                                               "frac": traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                               "note": f"PMC bytes of profiles/{src} / this run's time"},
                "executed": {"flops_per_frame_update": ex_flops, "compact_bytes_per_frame_update": ex_bytes, **tiles},
                "valu": {"achieved": ex_flops / (us * 1e-6) / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ex_flops / (us * 1e-6) / 1e12 / VALU_PEAK_TFLOPS, "busy": busy,
                         "algorithmic_tflops": alg_flops / sec / 1e12,
                         "note": "executed fp32 flops: per tile, window positions x (33 similarity flops per walk of the window + 2 per "
                                 "staged channel; 8 channels per walk, 16 or 12 for tiles with many labels); algorithmic_tflops prices all S channels of every pixel and may exceed the peak "
                                 "because the kernel carries only the labels near a tile; busy = VALU-issue share of the kernel's "
                                 "duration from the committed SQ counters"}}
        if host_leg is not None:
            out["value_host_buffers"] = host_leg
        if stream_leg is not None:
            out["value_stream640"] = stream_leg
        out["cpu_affinity"] = pinning
        if world == 1 and not args.no_stats_leg and not host_main and args.workload == "osd":
            # the stage that consumes these outputs (process_surfaces.py:672-689), batched on the frames still in HBM
            from tools.bench_stats import stats_stage_summary
            out["stats_stage"] = stats_stage_summary(ctx, torch, dev, ns, (d_rgb, d_angles, d_points, d_patches, d_edges, d_surfaces), H, W)
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(depth, rgb, r)
            out["cpu_baseline_b0"] = cpu_baseline_b0(depth, rgb, r)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
