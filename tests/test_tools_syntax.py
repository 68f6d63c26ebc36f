"""tools/ is development aid, not product — but a driver that no longer parses is worse than none ("tools/dev/*
A/B drivers are untested"). Every Python file must compile, every shell script must pass `bash -n`, every HIP probe must name
its build line, and the job scripts of the current round must only call tools that exist."""
import glob
import os
import py_compile
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "**", "*.py"), recursive=True))
    assert len(files) >= 15
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / "x.pyc"), doraise=True)


def test_shell_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "**", "*.sh"), recursive=True))
    assert len(files) >= 15
    for f in files:
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)


def test_job_scripts_reference_existing_tools():
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "jobs", "**", "*.sh"), recursive=True)):
        for m in re.finditer(r"\b(tools/[\w/.-]+\.(?:py|sh|hip))\b", open(f).read()):
            assert os.path.exists(os.path.join(ROOT, m.group(1))), (f, m.group(1))


def test_hw_probes_carry_their_build_line():
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "hw", "*.hip"))):
        head = open(f).read(6000)
        assert "Build:" in head and "--offload-arch=gfx950" in head, f
