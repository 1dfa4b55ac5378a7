"""The N>1 path on CPU: two ``gloo`` processes exercise the reference-view sharding, the
barrier-bracketed timing with max-over-ranks and the throughput aggregation bench.py uses."""
import os
import time

import torch

from itermvs_amd import shard


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = shard.init_distributed(backend="gloo")
    mine = shard.shard_indices(n_items, r, w)
    done = []

    def step(i):
        time.sleep(0.01 * (r + 1))            # rank 1 is the slow one
        done.append(i)

    elapsed = shard.timed_steps(step, steps=5, warmup=2)
    total = shard.sum_over_ranks(float(len(mine)))
    q.put((r, mine, elapsed, total, len(done)))
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_timing():
    from conftest import run_ranks
    world, n_items = 2, 11
    res = run_ranks(_worker, world, n_items)
    owned = sorted(i for _, mine, *_ in res for i in mine)
    assert owned == list(range(n_items))                         # disjoint cover of the reference views
    assert res[0][1] == [0, 2, 4, 6, 8, 10] and res[1][1] == [1, 3, 5, 7, 9]
    e0, e1 = res[0][2], res[1][2]
    assert abs(e0 - e1) < 1e-9 and e0 >= 5 * 0.02 * 0.9          # MAX over ranks: the slow rank's time
    assert res[0][3] == res[1][3] == float(n_items)              # whole-job count aggregates over ranks
    assert res[0][4] == res[1][4] == 7                           # warm-up + exactly K timed steps


def test_shard_helpers_single_process():
    assert shard.shard_counts(10, 4) == [3, 3, 2, 2]
    assert shard.shard_indices(3, 2, 4) == [2] and shard.shard_indices(3, 3, 4) == []
    assert shard.env_rank_world()[2] >= 1
    assert shard.max_over_ranks(1.5) == 1.5
