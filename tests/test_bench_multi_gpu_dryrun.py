"""bench.py's N > 1 paths on ONE GPU: two ranks pinned to device 0 (ZK_BENCH_DEVICE=0), gloo instead of RCCL (which refuses two
ranks on one device) — the same code path the driver runs over 8 GPUs with backend nccl: process-group setup, table broadcast
and row sharding of ONE global witness (--scaling strong), the tally all-gather, max-over-ranks timing, rank-0 JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, port):
    env = dict(os.environ, ZK_BENCH_DEVICE="0", ZK_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-cold-leg", "--no-fresh-leg"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1, p.stdout.decode()[-2000:]
    return json.loads(line[0])


@pytest.mark.parametrize("workload,log_rows", [("evm", 15), ("state", 14)])
def test_strong_scaling_two_ranks_one_gpu(workload, log_rows):
    d = _run(["--workload", workload, "--log-rows", str(log_rows), "--scaling", "strong"], 29731 + log_rows)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    total = (1 << log_rows) - (1 if workload == "evm" else 0)
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - total) < 1e-6 * total  # value = units all ranks processed / time


def test_weak_scaling_two_ranks_one_gpu():
    d = _run(["--workload", "evm", "--log-rows", "14"], 29761)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2 * ((1 << 14) - 1)) < 1e-3
