"""The multi-GPU path with real GPU ranks: two processes, each with its own context and stream, shard the OSD frames
round-robin (frame position % world == rank), run the HIP stage and write their own out/{index}/ files; the union
equals a single-process run bit for bit.  The box has one GPU, so both ranks are pointed at it (FS_DEVICE=0); on a
multi-GPU node the launcher's LOCAL_RANK selects the device instead."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, osd_frame

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_gpu_ranks_equal_one_process(tmp_path):
    lo, hi, world = 10, 17, 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FS_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker_gpu.py"), str(tmp_path), str(lo), str(hi)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    summary = json.load(open(tmp_path / "summary.json"))
    assert summary[0]["indices"] == list(range(lo, hi, 2)) and summary[1]["indices"] == list(range(lo + 1, hi, 2))
    frames = [osd_frame(i) for i in range(lo, hi)]
    from imagesegmentation_amd._lib import Context
    with Context(480, 640, max_batch=hi - lo) as ctx:
        ref = ctx.find_surfaces_batch(np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]))
    for k, i in enumerate(range(lo, hi)):
        d = tmp_path / str(i)
        for fname, key in (("Q", "surfaces"), ("angles", "angles"), ("vectors", "vectors"), ("patches", "patches"),
                           ("points", "points_3d"), ("edges", "depth_edges")):
            assert np.array_equal(np.load(d / (fname + ".npy")), ref[key][k]), (i, fname)
        assert np.array_equal(np.load(d / "depth.npy"), frames[k][0])


@pytest.mark.parametrize("workload,world", [("osd", 2), ("stream640", 2), ("osd", 4)])
def test_bench_with_several_ranks(workload, world):
    """bench.py's N > 1 path end to end: launched by torch.distributed.run as the driver launches it, two or four ranks,
    barriers on both sides of the timed region, the maximum over ranks, one JSON line from rank 0.  One card here, so the
    ranks share it and meet over gloo (FS_DEVICE / FS_BENCH_BACKEND); on a node the same code runs rank i on GPU i over RCCL."""
    env = dict(os.environ, FS_DEVICE="0", FS_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--frames", "8", "--workload", workload]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == world and doc["steps"] == 2 and doc["scaling"] == "weak"
    assert doc["value"] > 0 and abs(doc["value"] - world * 8 * 2 / (doc["ms_per_step"] * 2e-3)) < 1e-6 * doc["value"]
    assert "roofline" in doc and "cpu_baseline" not in doc
    # the host-buffer leg runs on all ranks at once after the headline loop (the stream workload is that leg already)
    assert ("value_host_buffers" in doc) == (workload == "osd")
    if workload == "osd":
        # ... with every rank's own rate beside the whole-job figure, and the synthetic stream of the multi-GPU configuration
        assert doc["value_host_buffers"]["value"] > 0 and len(doc["value_host_buffers"]["per_rank_images_per_s"]) == world
        assert doc["value_stream640"]["value"] > 0 and len(doc["value_stream640"]["per_rank_images_per_s"]) == world
    assert "pinned" in doc["cpu_affinity"]
    print(f"{workload}: {doc['value']:.0f} images/s over {world} ranks on one card")
