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


def pytest_runtest_teardown(item, nextitem):
    """Q3_RES_TRACE=<file> (development aid): process resources after every test — open fds, resident set, threads, memory
    maps, free device memory — to find what a long suite run exhausts."""
    path = os.environ.get("Q3_RES_TRACE")
    if not path:
        return
    try:
        fds = len(os.listdir("/proc/self/fd"))
        st = open("/proc/self/status").read()
        rss = [l for l in st.splitlines() if l.startswith("VmRSS")][0].split()[1]
        thr = [l for l in st.splitlines() if l.startswith("Threads")][0].split()[1]
        maps = sum(1 for _ in open("/proc/self/maps"))
        free = -1
        if torch is not None and torch.cuda.is_available():
            free = torch.cuda.mem_get_info()[0] >> 20
        with open(path, "a") as f:
            f.write(f"{item.name[:60]:60s} fds {fds} rss_kb {rss} threads {thr} maps {maps} gpu_free_mb {free}\n")
    except Exception as e:      # pragma: no cover
        with open(path, "a") as f:
            f.write(f"{item.name}: trace failed: {e}\n")
