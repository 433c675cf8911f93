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
