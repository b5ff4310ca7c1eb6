"""The reference's OWN test files, run with their drivers rebound to this package's mirrors (tools/run_reference_suite.py),
through the CPU backend behind the C ABI (libzkevm_cpu.so) — a sample of files here so the CPU suite stays within minutes; the
full 3,000-test run is `python tools/run_reference_suite.py --backend cpu|hip` (summaries under profiles/).  Build container
only: needs /root/reference."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="the reference is not on this machine")

SAMPLE = ["test_state_circuit.py", "test_bytecode_circuit.py", "test_tx_circuit.py", "test_sig_circuit.py", "test_public_inputs.py",
          "evm/test_add_sub.py", "evm/test_mul_div_mod.py", "evm/test_sdiv_smod.py", "evm/test_sha3.py", "evm/test_exp.py",
          "evm/test_callop.py", "evm/test_begin_tx.py", "evm/test_end_block.py", "evm/precompiles/test_ecRecover.py"]


def test_reference_test_files_pass_through_the_cpu_backend(tmp_path):
    out = tmp_path / "summary.json"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("ZK_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_suite.py"), "--backend", "cpu", "--out", str(out),
                        "--select"] + SAMPLE, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    s = json.loads(out.read_text())
    assert s["not_passed"] == 0, s["not_passed_detail"]
    assert s["tests_run"] >= 300
    calls = s["calls_through_the_boundary"]
    for entry in ("evm_circuit.verify_steps", "state_circuit.check_state_row", "bytecode_circuit.check_bytecode_row",
                  "copy_circuit.verify_copy_table", "exp_circuit.verify_exp_circuit", "tx_circuit.verify_circuit",
                  "sig_circuit.verify_circuit", "pi_circuit.verify_circuit"):
        assert calls.get("zkevm_specs_amd." + entry, 0) > 0, (entry, calls)
    assert not s["modules_without_a_rebound_driver"]
