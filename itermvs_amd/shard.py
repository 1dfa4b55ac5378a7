"""Multi-GPU sharding of the hot path: one process per GPU, independent reference views.

The reference drives its GPUs with single-process ``nn.DataParallel`` (eval.py:119) and, at the
``--batch_size 1`` of every eval script, only ever uses GPU 0.  Reference views are independent
units (eval.py:129-151 carries no state between samples), so the MI355X design shards the list
of reference views rank-strided over ``torch.distributed`` ranks with NO data-path collective;
RCCL (backend "nccl") is used only for the start/stop barriers and the max-over-ranks timing.
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Rank-strided partition: rank r owns items r, r+world, ...  (SURVEY.md 8(e))."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_items, world))


def shard_counts(n_items: int, world: int) -> List[int]:
    return [len(range(r, n_items, world)) for r in range(world)]


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str = None, force: bool = False) -> Tuple[int, int, int]:
    """Join the process group when launched by ``torch.distributed.run`` (any world size: a 1-rank launch through the
    launcher exercises the same RCCL initialisation and collectives as an 8-rank one); a plain ``python`` process of
    world 1 stays without a process group unless ``force``."""
    rank, local_rank, world = env_rank_world()
    launched = "TORCHELASTIC_RUN_ID" in os.environ
    if (world > 1 or force or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # ITERMVS_DIST_BACKEND=gloo: several ranks may then share one GPU (RCCL wants a device per rank) -- how the
            # world > 1 path of bench.py / train.py is exercised on a 1-GPU box (tests/test_rccl_gpu.py)
            backend = os.environ.get("ITERMVS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)           # RCCL binds the communicator to the current device
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


# -- host side of a rank: CPUs next to its GPU ----------------------------------------------------------------------------
# Eight ranks on a two-socket node: a rank whose pinned staging buffers (value_with_transfers, eval.py's prefetcher) and
# Python threads live on the other socket pays the inter-socket hop on every H2D / D2H copy.  Each rank therefore binds itself
# to the CPUs of its GPU's NUMA node BEFORE it allocates pinned memory (first touch then places the buffers on that node).
# The reference has no counterpart (one process, nn.DataParallel: eval.py:119, train.py:95).
_ORIGINAL_AFFINITY = None


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus: Sequence[int]) -> str:
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11'"""
    out, run = [], []
    for c in sorted(set(cpus)):
        if run and c == run[-1] + 1:
            run.append(c)
            continue
        if run:
            out.append(f"{run[0]}-{run[-1]}" if len(run) > 1 else str(run[0]))
        run = [c]
    if run:
        out.append(f"{run[0]}-{run[-1]}" if len(run) > 1 else str(run[0]))
    return ",".join(out)


def pci_numa_cpus(bdf: str, sysfs_root: str = "/sys"):
    """(numa node, CPUs local to it) of the PCI function ``bdf`` ('0000:c1:00.0'), read from sysfs: ``numa_node`` +
    ``local_cpulist`` of the device, else the node's own ``cpulist``; (None, []) when the platform does not say (node -1)"""
    base = os.path.join(sysfs_root, "bus", "pci", "devices", bdf.lower())
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
    except (OSError, ValueError):
        return None, []
    if node < 0:
        return None, []
    cpus: List[int] = []
    for path in (os.path.join(base, "local_cpulist"), os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")):
        try:
            cpus = parse_cpulist(open(path).read())
        except (OSError, ValueError):
            cpus = []
        if cpus:
            break
    return node, cpus


def gpu_pci_address(device_index: int) -> str:
    """PCI address 'dddd:bb:dd.0' of HIP device ``device_index`` (after HIP_/ROCR_VISIBLE_DEVICES re-mapping)"""
    p = torch.cuda.get_device_properties(device_index)
    return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"


def bind_to_gpu_numa(device_index: int, sysfs_root: str = "/sys", bdf: str = None, apply: bool = True):
    """Restrict this process to the CPUs of the NUMA node of its GPU (``os.sched_setaffinity``), intersected with the mask it
    already has (a container's cpuset stays in force).  Returns what bench.py prints in ``config.cpu_affinity``:
    {"gpu": bdf, "numa_node": n, "cpus": "0-31,128-159", "n_cpus": k, "bound": bool}, ``bound`` False (mask untouched) when
    sysfs gives no node, the intersection is empty, or ITERMVS_NO_AFFINITY=1."""
    global _ORIGINAL_AFFINITY
    if bdf is None:
        bdf = gpu_pci_address(device_index)
    node, cpus = pci_numa_cpus(bdf, sysfs_root)
    have = sorted(os.sched_getaffinity(0))
    if _ORIGINAL_AFFINITY is None:
        _ORIGINAL_AFFINITY = set(have)
    want = [c for c in cpus if c in set(have)]
    info = {"gpu": bdf, "numa_node": node, "cpus": format_cpulist(want or have), "n_cpus": len(want or have), "bound": False}
    if node is None or not want or os.environ.get("ITERMVS_NO_AFFINITY") == "1":
        return info
    if apply:
        os.sched_setaffinity(0, want)
        # (threads torch already started keep their masks; ranks bind right after init_distributed, before any pool exists)
    info["bound"] = bool(apply)
    return info


def original_affinity():
    """the CPU mask this process had before ``bind_to_gpu_numa`` narrowed it (None if it never did): the CPU-baseline leg of
    bench.py runs on the whole host, not on one NUMA node"""
    return None if _ORIGINAL_AFFINITY is None else set(_ORIGINAL_AFFINITY)


def restore_affinity() -> bool:
    """give EVERY thread of this process (OpenMP / torch pool threads inherit the mask they were created under) the mask the
    process had before ``bind_to_gpu_numa``; True if something was restored"""
    full = original_affinity()
    if not full:
        return False
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), full)
        except (OSError, ValueError):
            pass
    return True


def barrier(device_sync: bool = True) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device_sync and torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_steps(step: Callable[[int], None], steps: int, warmup: int) -> float:
    """Run ``warmup`` untimed then exactly ``steps`` timed calls of ``step(i)``, bracketed by a
    barrier + device synchronise on both sides; returns the MAX elapsed seconds over all ranks."""
    for i in range(warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed)


def timed_regions(step: Callable[[int], None], steps: int, warmup: int, repeats: int) -> List[float]:
    """``warmup`` untimed calls, then ``repeats`` timed regions of exactly ``steps`` calls each, back to back, every region
    bracketed by a barrier + device synchronise on both sides; returns the MAX-over-ranks elapsed seconds of each region
    (``step`` sees a running index).  A single short region is one scheduling hiccup away from a bad number: callers
    report the median region."""
    for i in range(warmup):
        step(i)
    out, base = [], warmup
    _LOCAL_REGIONS.clear()
    for _ in range(max(1, repeats)):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(base + i)
        if torch.cuda.is_available():
            torch.cuda.synchronize()                    # this rank's own work is done ...
        local = time.perf_counter() - t0
        barrier()                                       # ... the region ends when every rank's is
        _LOCAL_REGIONS.append(local)
        out.append(max_over_ranks(time.perf_counter() - t0))
        base += steps
    return out


_LOCAL_REGIONS: List[float] = []


def last_local_regions() -> List[float]:
    """THIS rank's elapsed seconds of the regions of the last ``timed_regions`` call (before the max over ranks)"""
    return list(_LOCAL_REGIONS)


def gather_over_ranks(value: float) -> List[float]:
    """``value`` of every rank, in rank order, on every rank (one all-gather of a double)"""
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]
    return [value]


def median(values: Sequence[float]) -> float:
    v = sorted(values)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def max_over_ranks(value: float) -> float:
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def sum_over_ranks(value: float) -> float:
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())
    return value
