"""Worker of tests/test_sharding_gloo.py: one rank of a world_size-N gloo job on the CPU.
The per-frame compute is the oracle (this is a test of the sharding/gather logic, which is
the same code the GPU ranks run; the GPU library is not touched)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, n_frames = sys.argv[1], int(sys.argv[2])
    import torch.distributed as dist
    from imagesegmentation_amd import load_images as li
    from imagesegmentation_amd.sharding import run_sharded
    from imagesegmentation_amd.synth import make_scene
    from oracle import oracle as O
    dist.init_process_group("gloo")
    li.out_root = out_dir
    params = O.gauss_params()

    def process(indices):
        for i in indices:
            depth, rgb = make_scene(100 + i, 48, 64)
            data = O.find_surfaces(depth, rgb, params)
            data["depth"] = depth
            li.save_surface_information(i, data)
        return {"images": len(indices)}

    res = run_sharded(list(range(n_frames)), process)
    if dist.get_rank() == 0:
        with open(os.path.join(out_dir, "summary.json"), "w") as f:
            json.dump([{k: r[k] for k in ("rank", "indices", "images")} for r in res], f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
