"""One rank of a world-2 q3_dp_* communicator with BOTH ranks on cuda:0 (test_native_rccl_world2_same_gpu starts two of
these). RCCL is expected to refuse two ranks on one device; whatever happens is printed as one line
`rank R: <ok|refused|error> <detail>` — the test records it and only requires that nothing hangs or crashes.
usage: dp_native_world2_check.py <rank> <rendezvous path>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import dp, synth, _lib

rank = int(sys.argv[1]); path = sys.argv[2]
try:
    comm = dp.NativeComm.from_file(path, rank, 2, 0, timeout_s=60.0, nonce="w2")
except _lib.Q3Error as e:
    print(f"rank {rank}: refused status {e.status}: {e}", flush=True); sys.exit(0)
except Exception as e:      # rendezvous timeout etc.
    print(f"rank {rank}: error {type(e).__name__}: {e}", flush=True); sys.exit(0)
cfg = q.tiny()
model = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=synth.DEFAULT_SEED) if rank == 0 else q.Qwen3TTS(cfg, device=0)
comm.broadcast_weights(model, 0)
if rank != 0:
    model.mark_loaded(); model.finalize()
s = model.session([q.Utterance([5, 6, 7], seed=5)], q.SynthesisOptions(max_length=4, seed=5, eos_token_id=None)); s.prefill(); s.generate(4)
digest = float(np.asarray(s.codes(0), dtype=np.float64).sum()); s.close()
got = comm.allgather([digest])
comm.close(); model.close()
print(f"rank {rank}: ok codes digest {got[:, 0].tolist()} {'match' if got[0, 0] == got[1, 0] else 'DIFFER'}", flush=True)
