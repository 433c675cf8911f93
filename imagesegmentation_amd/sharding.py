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


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of HIP device `device_index` from sysfs alone (no GPU call): the GPUs of the KFD topology in node order
    are the runtime's device order (the entries of ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES select among them, when they
    are plain indices); a node's location_id and domain give its PCI address, whose numa_node file names the node.
    None when anything on the way is missing (no pinning is then the right answer)."""
    try:
        base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = {}
            with open(os.path.join(base, node, "properties")) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    props[k] = v.strip()
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            sel = os.environ.get(var)
            if sel:
                gpus = [gpus[int(v)] for v in sel.split(",")]       # (UUID entries raise: no pinning)
        props = gpus[device_index]
        loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        with open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError, IndexError, KeyError):
        return None


def pin_to_gpu_numa_node(device_index, sysfs="/sys"):
    """Restricts this process (threads started later inherit it) to the CPUs of the NUMA node its GPU hangs on, before the
    first GPU call: a rank's page-locked buffers are then allocated on that node (first touch) and its copies do not
    cross the socket interconnect, which is what bounds several ranks that each move 17 MB per frame over PCIe.
    Returns a description of what was done; changes nothing when the node cannot be determined, when the node's CPUs
    are not among those the process may use, or with FS_NO_NUMA_PINNING=1."""
    if os.environ.get("FS_NO_NUMA_PINNING") == "1" or not hasattr(os, "sched_setaffinity"):
        return {"numa_node": None, "pinned": False, "why": "switched off or unsupported"}
    node = gpu_numa_node(device_index, sysfs)
    if node is None:
        return {"numa_node": None, "pinned": False, "why": "the GPU's NUMA node could not be read from sysfs"}
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        want = cpus & allowed
        if not want:
            return {"numa_node": node, "pinned": False, "why": "none of the node's CPUs is available to this process"}
        os.sched_setaffinity(0, want)
        return {"numa_node": node, "pinned": True, "cpus": len(want)}
    except (OSError, ValueError):
        return {"numa_node": node, "pinned": False, "why": "the node's CPU list could not be applied"}


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
    device = int(os.environ.get("FS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    pin_to_gpu_numa_node(device)        # before the first GPU call: the context's page-locked buffers land on the GPU's node
    fs.set_device(device)

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
