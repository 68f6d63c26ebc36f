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


def test_split_k_gemv_requests_before_it_waits(tmp_path):
    """k_gemv_sk2 asks for the k = 0 half's bias / residual up front so that they land under the weight stream. Summing the two at the
    top (`pre = resid + bias`) made hipcc wait for both (`s_waitcnt vmcnt(0)`) BEFORE the first x / weight request went out — a round
    trip at the head of every k = 0 workgroup, 0.6 % of the B = 8 frame (profiles/r5_ab_log.txt). The kernel's prologue is straight-line
    code, so the rule can be read off the ISA: no full vector-memory wait ahead of the first (nontemporal) weight-tile request."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "qwen3_tts_rs_amd", "csrc", "q3_kernels_gemv.hip")
    out = tmp_path / "gemv.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=14",
                        "--cuda-device-only", "-S", "-o", str(out), src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2q310k_gemv_sk2\w+:", l)]
    assert len(starts) >= 8, len(starts)
    for k, i in enumerate(starts):
        j = i + 1
        while j < len(lines) and not ("global_load_dwordx4" in lines[j] and lines[j].rstrip().endswith(" nt")):
            assert not re.search(r"s_waitcnt\s+vmcnt\(0\)", lines[j]), (lines[i][:80], j - i, lines[j])
            assert ".Lfunc_end" not in lines[j], lines[i][:80]
            j += 1
