"""Two data-parallel ranks on ONE GPU over gloo (test_dp_same_gpu launches this under torch.distributed.run): rank 0 builds
the synthetic checkpoint, the arena is broadcast, rank 1 finalizes from the received bytes; both ranks then synthesize the
same utterance and rank 1's codes must equal rank 0's. RCCL itself refuses two ranks per device, so this is the closest a
single-GPU box gets to the 8-GPU flow of bench.py (same dp.* calls, same order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import dp, synth
from common import synthetic_prompt

rank, local, world = dp.env_rank()
torch.cuda.set_device(0)
dp.init("gloo")
cfg = q.tiny()
model = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=synth.DEFAULT_SEED) if rank == 0 else q.Qwen3TTS(cfg, device=0)
n = dp.broadcast_arena(model, 0)
assert n > 0
if rank != 0:
    model.mark_loaded(); model.finalize()
opts = q.SynthesisOptions(max_length=6, seed=5, eos_token_id=None)
s = model.session([q.Utterance(synthetic_prompt(9, 1), seed=5)], opts); s.prefill(); s.generate(6)
codes = torch.from_numpy(s.codes(0).astype(np.int64)); pcm = torch.from_numpy(s.decode(0).copy())
s.close()
ref_codes, ref_pcm = codes.clone(), pcm.clone()
dist.broadcast(ref_codes, src=0); dist.broadcast(ref_pcm, src=0)
ok = bool((codes == ref_codes).all()) and bool(torch.equal(pcm, ref_pcm))
# sharding helper: utterance i -> rank i mod N
assert dp.shard_indices(5, rank, world) == [i for i in range(5) if i % world == rank]
t = dp.max_over_ranks(float(rank + 1)); assert t == float(world)
dp.barrier()
print(f"rank {rank}: arena {n} bytes, codes {'match' if ok else 'DIFFER'}", flush=True)
sys.exit(0 if ok else 1)
