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


class NativeComm:
    """RCCL communicator through the C ABI (q3_dp_*, q3_dp.cpp) — what a host without torch.distributed (the reference's
    Rust host) uses for the one weight broadcast. Rendezvous: rank 0 creates the 128-byte id and the host ships it to
    the other ranks; `from_file` does that through a shared path (every rank of the job must call it)."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int):
        import ctypes
        from . import _lib
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        _lib.check(_lib.lib.q3_dp_init(rank, world, buf, device, ctypes.byref(h)))
        self._h = h; self.rank = rank; self.world = world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib.q3_dp_unique_id(buf))
        return buf.raw

    @classmethod
    def from_file(cls, path: str, rank: int, world: int, device: int, timeout_s: float = 120.0, nonce: str = "") -> "NativeComm":
        """Rendezvous through a file: rank 0 publishes the RCCL id, the others read it. The file name carries a job nonce
        (pass the same `nonce` on every rank: the launcher's job id / master port; default: MASTER_PORT or TORCHELASTIC_RUN_ID
        from the environment) so that an id left behind by an earlier job, or read before rank 0 has rewritten it, can
        never be mistaken for this job's — a mismatched id makes ncclCommInitRank hang until its timeout."""
        import time
        nonce = nonce or os.environ.get("TORCHELASTIC_RUN_ID", "") + os.environ.get("MASTER_PORT", "")
        path = f"{path}.{nonce}" if nonce else path
        if rank == 0:
            if os.path.exists(path):
                os.unlink(path)          # never leave a stale id visible while the new one is being written
            uid = cls.unique_id()
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(path + ".tmp", path)
        else:
            t0 = time.time()
            started = t0 - 5.0           # an id older than this rank's start (minus clock slack) belongs to an earlier job
            while not (os.path.exists(path) and os.path.getsize(path) == 128 and (nonce or os.path.getmtime(path) >= started)):
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout_s}s")
                time.sleep(0.01)
            with open(path, "rb") as f:
                uid = f.read()
        return cls(rank, world, uid, device)

    def broadcast_weights(self, model, root: int = 0):
        from . import _lib
        _lib.check(_lib.lib.q3_dp_broadcast_weights(self._h, model._h, root))

    def allgather(self, values) -> "list":
        import ctypes
        import numpy as np
        from . import _lib
        v = np.ascontiguousarray(values, dtype=np.float64)
        out = np.empty((self.world, v.size), np.float64)
        _lib.check(_lib.lib.q3_dp_allgather_f64(self._h, v.ctypes.data_as(ctypes.c_void_p), v.size, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def close(self):
        from . import _lib
        if getattr(self, "_h", None):
            _lib.lib.q3_dp_free(self._h); self._h = None
