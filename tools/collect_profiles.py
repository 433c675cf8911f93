#!/usr/bin/env python3
"""Copies what tools/profile_round.sh left under gpurun_out/<dir> into profiles/ (tracked) under round-prefixed names and
derives profiles/<round>_traffic.json: per workload the HBM bytes per frame-update of the CRF kernel (FETCH_SIZE doubled,
as gfx950 tallies 128-byte read requests at 64 bytes; WRITE_SIZE as reported; both in KiB per dispatch), the waves
resident per SIMD and the VALU-issue share of the kernel's duration (SQ counters are quad-cycles of one XCD = 128 SIMDs,
GRBM_GUI_ACTIVE is the kernel's duration in cycles).

    python tools/collect_profiles.py gpurun_out/prof_r2 r2
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, kernel_prefix):
    """{counter: sum over all dispatches} of every kernel whose name starts with kernel_prefix in a pmc summary (frames of
    different channel classes run different instantiations of one template), and the number of those dispatches"""
    out, on, n, total = {}, False, 0, 0
    for line in open(path):
        if not line.startswith(" "):
            on = line.startswith(kernel_prefix)
            m = re.search(r"\(dispatches (\d+)\)", line)
            n = int(m.group(1)) if m else 0
            if on:
                total += n
        elif on:
            m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
            if m:
                out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(2)) * n
    return out, total


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    for d in sorted(glob.glob(os.path.join(src, "*"))):
        name = os.path.basename(d)
        stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            stats.sort(key=os.path.getmtime)                 # merged runs leave older traces behind: take the newest
            rows = list(csv.reader(open(stats[-1])))
            with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
                csv.writer(f).writerows(rows)
            shutil.copy(os.path.join(d, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_{name}_bench_under_rocprof.json"))
        if name.startswith("pmc_") and os.path.exists(os.path.join(d, "summary.txt")):
            shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, f"{tag}_{name}.txt"))
    workloads = {}
    for key, pm, frames, kernel in (("osd_r7", "osd", 16, "crf_iter_fused_kernel"), ("mosaic1280_s32_r7", "m7", 4, "crf_iter_fused_kernel"),
                                    ("mosaic1280_s32_r10", "m10", 4, "crf_iter_roll_kernel"), ("mosaic1280_s64_r15", "m15", 4, "crf_iter_roll_kernel"),
                                    ("dense1280_r7", "d7", 4, "crf_iter_"), ("dense1280_r10", "d10", 4, "crf_iter_"),
                                    ("dense1280_r15", "d15", 4, "crf_iter_")):
        entry = {"frames_per_dispatch": frames, "kernel": kernel}
        c = {}
        calls = 2                                            # the counter passes run one warm-up and one timed step
        for part in ("fetch", "write", "sq1", "grbm"):
            p = os.path.join(src, f"pmc_{pm}_{part}", "summary.txt")
            if os.path.exists(p):
                got, dispatches = counters(p, kernel)
                c.update(got)
                entry["dispatches_per_pass"] = dispatches    # three updates per call, one dispatch per channel class among the frames
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # sums over the pass -> per mean-field update of one frame (3 updates per call)
            entry["fetch_size_kib_per_frame_update"] = c["FETCH_SIZE"] / (3 * calls * frames)
            entry["write_size_kib_per_frame_update"] = c["WRITE_SIZE"] / (3 * calls * frames)
            entry["hbm_bytes_per_frame_update"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / (3 * calls * frames)
        if "GRBM_GUI_ACTIVE" in c and "SQ_WAVE_CYCLES" in c:
            simd_cycles = c["GRBM_GUI_ACTIVE"] * 128
            entry["waves_per_simd"] = 4 * c["SQ_WAVE_CYCLES"] / simd_cycles
            entry["valu_busy"] = 4 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
            entry["valu_instructions_per_frame_update_per_xcd"] = c["SQ_INSTS_VALU"] / (3 * calls * frames)
        workloads[key] = entry
    doc = {"source": f"profiles/{tag}_pmc_*.txt (rocprofv3 --pmc, one pass per run, tools/profile_round.sh)",
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported; unit KiB; counters summed over every "
                         "instantiation of the update kernel in a pass of two calls (three updates each)",
           "workloads": workloads}
    with open(os.path.join(dst, f"{tag}_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(workloads, indent=1))


if __name__ == "__main__":
    main()
