"""The window walks of the CRF update are hand-scheduled assembly (walk_similarity in csrc/k_crf_common.h).  The build system
also makes imagesegmentation_amd/libfs_hip_cwalk.so, the same sources with those walks left to the compiler (-DFS_ASM_STEP=0).
The two must give the same bits: the assembly performs the compiler's operations on the same operands in another order of
issue, so a difference would mean a register read before it was written or after it was overwritten (a wait state or a
clobber the assembly got wrong), not rounding."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CWALK = os.path.join(ROOT, "imagesegmentation_amd", "libfs_hip_cwalk.so")


@pytest.mark.gpu
def test_assembly_walk_equals_compiler_walk(tmp_path):
    assert os.path.exists(CWALK), "libfs_hip_cwalk.so is missing: run `make` in imagesegmentation_amd/csrc (build() does)"
    from imagesegmentation_amd import _lib
    assert os.path.basename(_lib.LIB_PATH) == "libfs_hip.so", "this test compares the product build with the compiler-walk build"
    sys.path.insert(0, HERE)
    import _walk_variant_worker as worker
    mine = worker.results()
    path = str(tmp_path / "cwalk.npz")
    env = dict(os.environ, FS_HIP_LIB=CWALK)
    subprocess.run([sys.executable, os.path.join(HERE, "_walk_variant_worker.py"), path], check=True, env=env, timeout=600)
    other = np.load(path)
    assert sorted(other.files) == sorted(mine)
    for k in sorted(mine):
        a, b = np.ascontiguousarray(mine[k]), other[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), f"{k}: {int((a != b).sum())} elements differ between the assembly walk and the compiler's"
    print("identical:", ", ".join(sorted(mine)))
