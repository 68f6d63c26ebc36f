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



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
