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


def _rdv_paths(path: str, world: int, nonce: str):
    base = f"{path}.{nonce}" if nonce else path
    return base, [f"{base}.req.{r}" for r in range(1, world)], [f"{base}.ack.{r}" for r in range(1, world)]


def _rdv_put(p: str, data: bytes):
    tmp = f"{p}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, p)


def _rdv_get(p: str, n: int):
    try:
        with open(p, "rb") as f:
            d = f.read()
        return d if len(d) == n else None
    except OSError:
        return None


def file_rendezvous(path: str, rank: int, world: int, make_uid, timeout_s: float = 120.0, nonce: str = "") -> bytes:
    """File handshake that cannot return a stale id, whatever an earlier job (same path, same MASTER_PORT, a crash that
    skipped the clean-up) left behind. Every non-root rank draws a fresh 16-byte token and posts it (`<path>.req.<rank>`);
    rank 0 publishes `id || tokens of ranks 1..world-1` and re-publishes whenever the posted tokens change; a non-root rank
    accepts an id file only if it carries ITS token at its slot, then acknowledges with the token (`<path>.ack.<rank>`);
    rank 0 leaves once every acknowledgement equals the token it published for that rank. Rank 0 starts by deleting the id
    and acknowledgement files (no valid one can exist before its first publication), so a stale `req` can only delay the
    handshake until the live rank overwrites it. Plain files + atomic renames: works on any shared directory."""
    import time
    base, reqs, acks = _rdv_paths(path, world, nonce)
    n_blob = 128 + 16 * (world - 1)
    deadline = time.time() + timeout_s
    if rank == 0:
        for p in [base] + acks:
            try:
                os.unlink(p)
            except OSError:
                pass
        uid = make_uid()
        assert len(uid) == 128
        published = None
        while True:
            toks = [_rdv_get(p, 16) for p in reqs]
            if all(t is not None for t in toks):
                if toks != published:
                    _rdv_put(base, uid + b"".join(toks)); published = toks
                if [_rdv_get(p, 16) for p in acks] == toks:
                    return uid
            if time.time() > deadline:
                raise TimeoutError(f"rank 0: ranks 1..{world - 1} did not complete the rendezvous at {base} within {timeout_s}s")
            time.sleep(0.005)
    token = os.urandom(16)
    _rdv_put(reqs[rank - 1], token)
    while True:
        blob = _rdv_get(base, n_blob)
        if blob is not None and blob[128 + 16 * (rank - 1):128 + 16 * rank] == token:
            _rdv_put(acks[rank - 1], token)
            return blob[:128]
        if time.time() > deadline:
            raise TimeoutError(f"rank {rank}: no RCCL id for this run at {base} after {timeout_s}s")
        time.sleep(0.005)


def rendezvous_cleanup(path: str, world: int, nonce: str = ""):
    base, reqs, acks = _rdv_paths(path, world, nonce)
    for p in [base] + reqs + acks:
        try:
            os.unlink(p)
        except OSError:
            pass


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
        """Rendezvous through a shared path (every rank of the job calls this): `file_rendezvous` below hands every rank
        the 128-byte RCCL id rank 0 created FOR THIS RUN — never one left behind by an earlier job, which would make
        ncclCommInitRank hang until its timeout — then the communicator is built, one all-gather proves that every rank
        is past the rendezvous, and rank 0 removes the files. `nonce` is only a namespace (several jobs sharing a directory)."""
        uid = file_rendezvous(path, rank, world, cls.unique_id, timeout_s, nonce)
        comm = cls(rank, world, uid, device)
        comm.allgather([float(rank)])
        if rank == 0:
            rendezvous_cleanup(path, world, nonce)
        return comm

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
