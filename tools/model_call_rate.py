#!/usr/bin/env python3
"""Rate of the reference's in-memory entry point, find_surface_model()(x) (find_surfaces.py:563-565): one frame per
call, numpy arrays in, the dict of numpy arrays out, each result released before the next call (as a caller that
processes frame after frame does).  Also the same frames through Context.find_surfaces_batch with page-locked inputs,
and the model called from two and three threads at once.

    python tools/model_call_rate.py [--calls 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=300)
    args = ap.parse_args()
    from imagesegmentation_amd import find_surfaces as fs
    from imagesegmentation_amd._lib import pinned_empty
    from imagesegmentation_amd.load_images import load_image
    frames = [load_image(i) for i in range(16)]
    model = fs.find_surface_model()
    for f in frames[:4]:
        model(f)
    t0 = time.perf_counter()
    for k in range(args.calls):
        out = model(frames[k % len(frames)])
        del out
    dt = time.perf_counter() - t0
    res = {"find_surface_model_calls_per_s": args.calls / dt, "ms_per_call": 1e3 * dt / args.calls}
    # page-locked inputs (a caller that keeps its frames in fs_host_alloc memory)
    ctx = fs._context(480, 640, 7)
    pd = [pinned_empty(f[0].shape, np.uint16) for f in frames]
    pc = [pinned_empty(f[1].shape, np.uint8) for f in frames]
    for a, b, f in zip(pd, pc, frames):
        a[...] = f[0]
        b[...] = f[1]
    out = None
    for k in range(4):
        out = ctx.find_surfaces_batch(pd[k], pc[k], out=out)
    t0 = time.perf_counter()
    for k in range(args.calls):
        out = ctx.find_surfaces_batch(pd[k % len(frames)], pc[k % len(frames)], out=out)
    dt = time.perf_counter() - t0
    res["pinned_in_and_out_calls_per_s"] = args.calls / dt
    res["pinned_ms_per_call"] = 1e3 * dt / args.calls
    # the same entry point from several threads at once (a server's worker threads): the module keeps a library
    # context per thread, and one thread's downloads run beside another's kernels
    import threading
    for workers in (2, 3):
        took = [0.0] * workers

        def work(k):
            for f in frames[:4]:
                model(f)                    # the thread's context and its buffers come into being
            t = time.perf_counter()
            for j in range(args.calls):
                out = model(frames[(j + k) % len(frames)])
                del out
            took[k] = time.perf_counter() - t

        threads = [threading.Thread(target=work, args=(k,)) for k in range(workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        res[f"find_surface_model_calls_per_s_{workers}_threads"] = sum(args.calls / t for t in took)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
