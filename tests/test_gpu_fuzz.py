"""The differential fuzzers (tests/gpu_fuzz_*.py: HIP path vs the oracle, per-row / per-pair status words bit for bit) as bounded
legs of the driver-run GPU suite (round 5 ran them by hand: profiles/r05_fuzz.txt).  Each leg is the script itself in a child
process — its exit code is the number of mismatching cases != 0 — with a case count that keeps all three under a minute."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, clean="0 mismatching cases", timeout=240):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), *map(str, args)], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = "\n".join(p.stdout.strip().splitlines()[-6:])
    assert p.returncode == 0, tail
    assert clean in tail, tail
    return tail


@pytest.mark.gpu
def test_fuzz_state_100_cases():
    tail = _run("gpu_fuzz_state.py", 100, 61)
    assert "100 cases" in tail


@pytest.mark.gpu
def test_fuzz_copy_50_cases():
    tail = _run("gpu_fuzz_copy.py", 50, 17)
    assert "50 cases" in tail


@pytest.mark.gpu
def test_fuzz_evm_5_rounds():
    tail = _run("gpu_fuzz_evm.py", 5, 43, clean="mismatching cases: 0")
    assert "fuzzed step pairs" in tail
