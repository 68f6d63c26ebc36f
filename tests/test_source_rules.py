"""Source rules of the HIP code that a compiler will not enforce.

Separately rounded products: ROCm's `__fmul_rn` / `__fadd_rn` / `__fsub_rn` are plain `x * y` / `x + y` / `x - y`
(`__clang_hip_math.h` without OCML_BASIC_ROUNDED_OPERATIONS), so hipcc may contract them into an FMA with a neighbour — it did in one
template instance of k_attn_cp's RoPE and not in the other. The kernels use `mul_rn` / `add_rn` / `sub_rn` of q3_kernels.h, which are
compiled with `#pragma clang fp contract(off)`."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qwen3_tts_rs_amd", "csrc")


def _code(path):
    """file text without // and /* */ comments"""
    s = open(path).read()
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def test_no_contractible_rounding_intrinsics():
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    assert len(files) >= 10
    for f in files:
        m = re.search(r"\b__f(mul|add|sub)_rn\s*\(", _code(f))
        assert m is None, (os.path.basename(f), m.group(0))


def test_rounding_helpers_switch_contraction_off():
    h = open(os.path.join(CSRC, "q3_kernels.h")).read()
    for name in ("mul_rn", "add_rn", "sub_rn"):
        m = re.search(r"float\s+%s\s*\(float a, float b\)\s*\{\s*#pragma clang fp contract\(off\)" % name, h)
        assert m, name
