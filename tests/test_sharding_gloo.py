"""world_size-2 run of the multi-GPU path on the CPU (gloo): frames are sharded round-robin,
every rank writes its own out/{index}/ files, nothing is exchanged on the data path, and the
union of the ranks' outputs equals a single-process run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_cover_the_frames_once(tmp_path, oracle):
    n_frames, world = 5, 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py"), str(tmp_path), str(n_frames)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    summary = json.load(open(tmp_path / "summary.json"))
    assert [s["rank"] for s in summary] == [0, 1]
    assert summary[0]["indices"] == [0, 2, 4] and summary[1]["indices"] == [1, 3]
    assert sum(s["images"] for s in summary) == n_frames
    # every frame was written exactly where a single process would have written it, with the same content
    from imagesegmentation_amd.synth import make_scene
    for i in range(n_frames):
        d = tmp_path / str(i)
        assert sorted(os.listdir(d)) == ["Q.npy", "angles.npy", "depth.npy", "edges.npy", "patches.npy", "points.npy", "vectors.npy"]
    depth, rgb = make_scene(100 + 3, 48, 64)
    ref = oracle.find_surfaces(depth, rgb, oracle.gauss_params())
    assert np.array_equal(np.load(tmp_path / "3" / "Q.npy"), ref["surfaces"])
    assert np.array_equal(np.load(tmp_path / "3" / "patches.npy"), ref["patches"])
