import os
import sys

import pytest

# Load torch BEFORE the HIP library, as bench.py does: torch's libtorch_hip.so asks for "libamdhip64.so" (its bundled
# copy), libq3tts.so for "libamdhip64.so.7"; with torch first the loader satisfies both with ONE runtime (same soname),
# in the other order the process ends up with two HIP runtimes and torch.cuda finds no GPU. Only the RCCL plumbing test
# and bench.py use torch at all.
try:
    import torch  # noqa: F401
except Exception:      # pragma: no cover - torch is plumbing, the suite runs without it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# exercise the opt-in overlapped segment decode of q3_session_run in the GPU suite (the library reads this once,
# at its first q3_session_run); runs of <= 128 frames still take the default whole-utterance path
os.environ.setdefault("Q3_DECODE_OVERLAP", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
