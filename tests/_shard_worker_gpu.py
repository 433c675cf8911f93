"""Worker of tests/test_gpu_sharding.py: one rank of a world_size-N job whose ranks all run the HIP path (each
its own process, context and stream) and write their own out/{index}/ files; gloo carries the summaries only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch.distributed as dist
    from imagesegmentation_amd import load_images as li
    from imagesegmentation_amd.sharding import find_surfaces_for_indices_sharded
    dist.init_process_group("gloo")
    li.out_root = out_dir
    res = find_surfaces_for_indices_sharded(list(range(lo, hi)), save_data=True)
    if dist.get_rank() == 0:
        with open(os.path.join(out_dir, "summary.json"), "w") as f:
            json.dump([{k: r[k] for k in ("rank", "indices", "images")} for r in res], f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
