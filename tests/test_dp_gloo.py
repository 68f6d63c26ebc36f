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


def test_file_rendezvous_ignores_stale_files(tmp_path):
    """NativeComm.from_file's handshake (a stale RCCL id under the same file name — same MASTER_PORT, crashed job —
    must never be taken for this run's): plant a complete set of files from a 'previous job', then run three ranks in
    threads with rank 0 starting LAST; every rank must come out with the id rank 0 made for THIS run."""
    import threading, time
    path = str(tmp_path / "rccl_id")
    world = 3
    stale = bytes([0xEE]) * 128
    stale_tok = [bytes([r]) * 16 for r in range(1, world)]
    with open(path, "wb") as f:
        f.write(stale + b"".join(stale_tok))
    for r in range(1, world):
        for kind in ("req", "ack"):
            with open(f"{path}.{kind}.{r}", "wb") as f:
                f.write(stale_tok[r - 1])
    fresh = bytes(range(128))
    out, errs = {}, []

    def run(rank, delay):
        try:
            time.sleep(delay)
            out[rank] = dp.file_rendezvous(path, rank, world, lambda: fresh, timeout_s=20.0)
        except Exception as e:          # noqa: BLE001
            errs.append((rank, repr(e)))
    ts = [threading.Thread(target=run, args=(1, 0.0)), threading.Thread(target=run, args=(2, 0.15)), threading.Thread(target=run, args=(0, 0.3))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert not errs, errs
    assert out == {0: fresh, 1: fresh, 2: fresh}
    dp.rendezvous_cleanup(path, world)
    assert not [p for p in os.listdir(tmp_path) if p.startswith("rccl_id")]


def test_file_rendezvous_times_out_without_root(tmp_path):
    import pytest
    with open(str(tmp_path / "id"), "wb") as f:          # a stale id alone must not satisfy a non-root rank
        f.write(bytes(128 + 16))
    with pytest.raises(TimeoutError):
        dp.file_rendezvous(str(tmp_path / "id"), 1, 2, lambda: bytes(128), timeout_s=0.3)
