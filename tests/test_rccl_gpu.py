"""RCCL readiness on ONE MI355X: every collective the multi-GPU paths issue (bench.py's barriers and max-over-ranks,
the training all-reduce and parameter broadcast; /root/reference train.py:95 is single-process DataParallel instead) runs
here on a 1-rank ``nccl`` process group on cuda:0, so that the first 8-GPU launch is not the first time
``init_process_group("nccl")`` and a device-side all-reduce execute.  Also: ``bench.py`` launched exactly the way
``python bench.py --gpus N`` launches itself (torch.distributed.run, rendezvous on 127.0.0.1), with N = 1."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, run_ranks

pytestmark = pytest.mark.gpu


def _nccl_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from itermvs_amd import ddp, shard
    r, lr, w = shard.init_distributed(backend="nccl", force=True)
    assert (r, lr, w) == (0, 0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    shard.barrier()
    mx, sm = shard.max_over_ranks(3.25), shard.sum_over_ranks(1.5)
    regions = shard.timed_regions(lambda i: torch.zeros(8, device=dev).add_(1), steps=3, warmup=1, repeats=2)
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 5).to(dev)
    unused = torch.nn.Parameter(torch.zeros(4, device=dev))
    ddp.broadcast_parameters(lin)                                   # device-side broadcast over RCCL
    lin(torch.ones(2, 7, device=dev)).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    skipped = ddp.flat_allreduce_gradients(list(lin.parameters()) + [unused])          # world 1: returns early
    n = ddp.flat_allreduce_gradients(list(lin.parameters()) + [unused], force=True)    # device all-reduce of the flat bucket
    same = all(torch.equal(a, p.grad) for a, p in zip(local, lin.parameters()))        # mean over one rank == itself
    t = torch.arange(1 << 18, device=dev, dtype=torch.float32)       # the size of the real bucket (1.37 MB)
    dist.all_reduce(t)
    ok_big = bool(torch.equal(t, torch.arange(1 << 18, device=dev, dtype=torch.float32)))
    torch.cuda.synchronize()
    q.put((rank, mx, sm, len(regions), skipped, n, same, ok_big, unused.grad is None))
    dist.destroy_process_group()


def test_one_rank_nccl_process_group_runs_every_collective_of_the_path():
    (_, mx, sm, nreg, skipped, n, same, ok_big, unused_none), = run_ranks(_nccl_worker, 1, timeout=300)
    assert mx == 3.25 and sm == 1.5 and nreg == 2
    assert skipped == 0 and n == 7 * 5 + 5 and same and ok_big and unused_none


def test_bench_self_launch_under_torch_distributed_run_with_one_rank():
    """`python bench.py --gpus N` re-executes itself as this command line; with N = 1 the launcher still sets
    TORCHELASTIC_RUN_ID, so shard.init_distributed() creates the nccl group and the timed region's barriers and
    max-over-ranks all-reduce go through RCCL"""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_command(1, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1", "--minimal",
                                        "--height", "128", "--width", "160", "--views", "3", "--iters", "2"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "127.0.0.1" in cmd
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ITERMVS_EXPECT_PROCESS_GROUP="nccl")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["steps"] == 2
    assert line["config"]["process_group"] == "nccl"


def test_bench_with_two_ranks_sharing_one_gpu_over_gloo():
    """bench.py's OWN world > 1 path -- rank-local samples, barrier-bracketed regions, max over ranks, `world * steps * batch /
    elapsed`, ONE JSON line from rank 0, per-rank spread -- executed before the first 8-GPU box does it: two ranks under
    torch.distributed.run share cuda:0, the process group is gloo (ITERMVS_DIST_BACKEND; RCCL wants a device per rank)"""
    sys.path.insert(0, ROOT)
    import bench
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--minimal",
            "--height", "128", "--width", "160", "--views", "3", "--iters", "2"]
    cmd = bench.self_launch_command(2, args)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ITERMVS_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["process_group"] == "gloo" and "x2" in line["config"]["parallelism"]
    # whole-job aggregate: both ranks' depth maps over the slowest rank's clock
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    spread = line["ms_per_step_ranks"]
    assert 0 < spread["min"] <= spread["max"] <= line["ms_per_step_max"] * 1.001
    assert line["roofline"] is not None and line["roofline"]["launches_timed"] > 0
