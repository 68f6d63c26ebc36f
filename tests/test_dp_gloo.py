"""N>1 path on CPU: world_size-2 gloo run of the data-parallel plumbing (sharding, the single
weight broadcast, max-over-ranks timing, whole-job aggregation)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from qwen3_tts_rs_amd import dp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, l, w = dp.init(backend="gloo")
    assert (r, w) == (rank, world)
    # sharding: 64 utterances, i -> i mod N, disjoint and complete
    mine = dp.shard_indices(64, rank, world)
    # the one collective: broadcast of the (stand-in) weight arena from rank 0
    arena = torch.full((1 << 16,), 7 if rank == 0 else 0, dtype=torch.uint8)
    if rank == 0:
        arena[::3] = 11
    dp.broadcast_tensor(arena, 0)
    ok_bcast = bool((arena[::3] == 11).all() and arena[1] == 7)
    dp.barrier()
    t = dp.max_over_ranks(1.0 + rank)            # slowest rank defines the step time
    total = dp.sum_over_ranks(float(len(mine)))  # whole-job units
    q.put((rank, mine, ok_bcast, t, total))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60); assert p.exitcode == 0
    all_idx = sorted(res[0][1] + res[1][1])
    assert all_idx == list(range(64)) and not set(res[0][1]) & set(res[1][1])
    assert res[0][1] == list(range(0, 64, 2)) and res[1][1] == list(range(1, 64, 2))
    for rank, mine, ok, t, total in res:
        assert ok and t == 2.0 and total == 64.0


def test_shard_edge_cases():
    assert dp.shard_indices(0, 0, 4) == []
    assert dp.shard_indices(3, 3, 4) == []
    assert dp.shard_indices(5, 1, 4) == [1]
    assert sum(len(dp.shard_indices(64, r, 8)) for r in range(8)) == 64
