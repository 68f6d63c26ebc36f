"""Data-parallel plumbing: one process per GPU, utterances sharded round-robin, exactly ONE
collective on the data path — the broadcast of the weight arena from rank 0 at start-up (RCCL over
xGMI when the backend is "nccl"; "gloo" in the CPU tests). Steady state has no inter-GPU traffic:
utterances are independent (reference lib.rs:744-756 — per-call KV, RNG, masks).
torch.distributed is plumbing here (rendezvous + the broadcast), not part of the compute path.
"""
import os
from typing import List, Tuple


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_total: int, rank: int, world: int) -> List[int]:
    """utterance i → rank i mod N."""
    return [i for i in range(n_total) if i % world == rank]


def init(backend: str = "nccl"):
    import torch.distributed as dist
    rank, local, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class _DevMem:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it (no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_arena(model, device: int, chunk_bytes: int = 1 << 28) -> int:
    """Broadcast rank 0's weight arena into every other rank's arena (same layout by construction:
    the arena layout is a pure function of the config). Returns the number of bytes broadcast."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    ptr, nbytes = model.arena()
    t = torch.as_tensor(_DevMem(ptr, nbytes), device=f"cuda:{device}")
    for off in range(0, nbytes, chunk_bytes):
        dist.broadcast(t[off:off + chunk_bytes], src=0)
    torch.cuda.synchronize(device)
    return nbytes


def broadcast_tensor(t, src: int = 0):
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
