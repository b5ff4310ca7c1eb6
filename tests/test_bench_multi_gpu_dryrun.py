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


def _run(extra, port, env_extra=None):
    env = dict(os.environ, ZK_BENCH_DEVICE="0", ZK_BENCH_BACKEND="gloo", **(env_extra or {}))
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


def _run_plain(extra):
    """the way the driver may start it: plain `python bench.py --gpus 2 ...`, no torchrun environment — bench.py spawns its ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZK_BENCH_DEVICE="0", ZK_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-cold-leg",
           "--no-fresh-leg"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    err = p.stderr.decode()
    assert p.returncode == 0, "\n".join(ln for ln in err.splitlines() if "[rank" in ln or "Error" in ln)[-6000:] + err[-1500:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1, p.stdout.decode()[-2000:]
    return json.loads(line[0])


@pytest.mark.parametrize("workload,log_rows,scaling", [("evm", 14, "strong"), ("state", 13, "strong"), ("tx", 8, "strong"), ("super", 14, "strong"),
                                                      ("tx", 7, "weak")])
def test_self_spawned_ranks(workload, log_rows, scaling):
    d = _run_plain(["--workload", workload, "--log-rows", str(log_rows), "--scaling", scaling])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    units = d["value"] * d["ms_per_step"] / 1e3  # units all ranks processed per pass
    if workload == "super":
        total = sum(d["config"]["rows_total"].values())
        # every circuit's rows are cut in two: the ranks' shares add up to the block
        assert sum(d["config"]["rows_per_gpu"].values()) in (total // 2, (total + 1) // 2, total // 2 + 2, total // 2 + 3)
    else:
        total = ((1 << log_rows) - (1 if workload == "evm" else 0)) * (2 if scaling == "weak" else 1)
    assert abs(units - total) < 1e-6 * total


def test_weak_scaling_two_ranks_one_gpu():
    d = _run(["--workload", "evm", "--log-rows", "14"], 29761)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2 * ((1 << 14) - 1)) < 1e-3


def _fake_rccl_hip():
    d = os.path.join(ROOT, "tests", "fakerccl")
    so, src = os.path.join(d, "libfakerccl_hip.so"), os.path.join(d, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([os.path.join(d, "build.sh")])
    return so


@pytest.mark.parametrize("workload,log_rows,scaling", [("evm", 14, "strong"), ("state", 13, "weak")])
def test_abi_tally_two_ranks_one_gpu(workload, log_rows, scaling):
    """`--tally abi` with N = 2: the HIP library's own zk_dist_init / zk_dist_tally with world 2 — ncclCommInitRank and the
    all-gather of device buffers on the communicator's stream — through tests/fakerccl's stand-in (RCCL refuses two ranks on one
    device); bench.py asserts that the tally equals torch.distributed's"""
    d = _run(["--workload", workload, "--log-rows", str(log_rows), "--scaling", scaling, "--tally", "abi"], 29791 + log_rows,
             {"ZK_RCCL_LIB": _fake_rccl_hip()})
    assert d["n_gpus"] == 2 and d["value"] > 0
