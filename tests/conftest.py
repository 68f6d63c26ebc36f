import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# exercise the opt-in overlapped segment decode of q3_session_run in the GPU suite (the library reads this once,
# at its first q3_session_run); runs of <= 128 frames still take the default whole-utterance path
os.environ.setdefault("Q3_DECODE_OVERLAP", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
