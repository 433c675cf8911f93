"""Multi-GPU execution of the stage: images are independent units, so the index list is
split over one process per GPU and nothing is exchanged on the data path
(find_surfaces.py:554-559 is a plain loop over indices; outputs are per-index
directories out/{index}/).

Launch one process per GPU, e.g.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m imagesegmentation_amd.sharding --indices 0:111
Each rank takes the indices i with position % world == rank, runs them on GPU
LOCAL_RANK and writes its own out/{index}/ files.  The only communication is one
all_gather_object of the per-rank summaries (counts and wall time) for the report.
"""
import argparse
import os
import time


def shard(items, rank, world):
    """Round-robin shard: positions rank, rank + world, ... (frame_id % n_gpu == rank)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(items)[rank::world]


def run_sharded(image_indices, process, rank=None, world=None, gather=True):
    """Runs process(list_of_indices) -> dict on this rank's shard; returns the list of every
    rank's summary dict (on every rank) when torch.distributed is initialised and gather
    is set, else just this rank's.  `process` owns the GPU work and the file output."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if distributed else int(os.environ.get("RANK", "0"))
    if world is None:
        world = dist.get_world_size() if distributed else int(os.environ.get("WORLD_SIZE", "1"))
    mine = shard(image_indices, rank, world)
    t0 = time.perf_counter()
    summary = dict(process(mine) or {})
    summary.update(rank=rank, indices=mine, seconds=time.perf_counter() - t0)
    if distributed and gather and world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, summary)
        return everyone
    return [summary]


def find_surfaces_for_indices_sharded(image_indices, mode="Gauss", kernel_size=7, save_data=True):
    """find_surfaces_for_indices over this rank's shard on GPU LOCAL_RANK."""
    from . import find_surfaces as fs
    fs.set_device(int(os.environ.get("FS_DEVICE", os.environ.get("LOCAL_RANK", "0"))))

    def process(indices):
        fs.find_surfaces_for_indices(indices, mode=mode, kernel_size=kernel_size, save_data=save_data)
        return {"images": len(indices), "decode_s": fs.timings["decode"], "gpu_s": fs.timings["gpu"], "save_wait_s": fs.timings["save"]}
    return run_sharded(image_indices, process)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--indices", default="0:111", help="lo:hi range or comma-separated list")
    ap.add_argument("--kernel-size", type=int, default=7)
    ap.add_argument("--no-save", action="store_true")
    args = ap.parse_args(argv)
    if ":" in args.indices:
        lo, hi = (int(v) for v in args.indices.split(":"))
        indices = list(range(lo, hi))
    else:
        indices = [int(v) for v in args.indices.split(",")]
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # gloo: the summaries are host objects; the GPUs exchange nothing
        dist.init_process_group("gloo")
    res = find_surfaces_for_indices_sharded(indices, kernel_size=args.kernel_size, save_data=not args.no_save)
    if int(os.environ.get("RANK", "0")) == 0:
        n = sum(r["images"] for r in res)
        t = max(r["seconds"] for r in res)
        print(f"{n} images on {len(res)} GPU(s) in {t:.2f} s = {n / t:.1f} images/s (decode and .npy writes included)")
        for r in res:
            print(f"  rank {r['rank']}: {r['images']} images, PNG decode {r['decode_s']:.2f} s, upload + kernels + download "
                  f"{r['gpu_s']:.2f} s, waiting for .npy writes {r['save_wait_s']:.2f} s")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
