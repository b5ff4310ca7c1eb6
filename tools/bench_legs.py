#!/usr/bin/env python3
"""Side measurements of bench.py — everything that is NOT the timed headline loop and its roofline block: the CPU baseline legs
(`cpu_baseline`: pure-Python port, libzkevm_cpu.so on one core and on the cgroup's effective cores, the unmodified reference in a
child process), the other BASELINE configurations at N = 1 (`other_configs`), the batch entry, the flushed-cache open / pass
split, the live `rocprofv3 --pmc` traffic measurement and the marshalling sample.  bench.py imports this module and hands itself
over as `core` where a leg needs its builders / timing loops; their results go to bench_full.json and, as a few scalars, into
the compact line."""
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0


def timed_batch(ctx, w, steps, warmup):
    """the same K fresh-witness verifications through ONE call of the batch entry (zk_evm_verify_batch: two witnesses in flight on
    two streams); wall clock of the call, bracketed like the headline"""
    from zkevm_specs_amd import engine

    n_copies = len(w.shots)
    mk = lambda n, first: engine.EvmBatch(w.copies, [(first + i) % n_copies for i in range(n)], device=ctx.local_rank)  # noqa: E731
    mk(max(warmup, 2), 0)()
    b = mk(steps, warmup)
    ctx.barrier()
    t0 = time.perf_counter()
    b()
    ctx.barrier()
    dt = time.perf_counter() - t0
    rs = b.results()
    assert all(r.ok for r in rs), "synthetic witness must satisfy every constraint"
    return {"ms_per_witness": dt / steps * 1e3, "rows_per_s": w.units * steps / dt, "witnesses": steps,
            "mean_pass_kernel_ms": sum(r.kernel_ms for r in rs) / len(rs)}


def fresh_leg(ctx, w, flush):
    """A verifier sees each witness once.  (a) open (device-resident inputs: packed key records, density check, bytecode
    directory, small-table indices — device kernels, no host synchronisation inside) + one pass over cold caches, wall clock
    around both with ONE synchronisation at the end (the collect); (b) the same split into open / pass with a flush and a
    synchronisation between them (round-2 definition, comparable); (c) the one-shot C entry (open + launch + collect + close).
    Best of 5 each."""
    np, torch = ctx.np, ctx.torch
    both, opens, passes, open_dev, shots = [], [], [], [], []
    for _ in range(5):
        flush.sum()
        torch.cuda.synchronize()
        t = time.perf_counter()
        s2 = w.fresh()
        r2 = s2.run()
        both.append(time.perf_counter() - t)
        assert r2.ok
        s2.close()
    for _ in range(5):
        flush.sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record()
        s2 = w.fresh()
        e1.record()
        torch.cuda.synchronize()
        t_open = time.perf_counter() - t
        flush.sum()
        torch.cuda.synchronize()
        t = time.perf_counter()
        r2 = s2.run()
        t_pass = time.perf_counter() - t
        assert r2.ok
        s2.close()
        opens.append(t_open)
        passes.append(t_pass)
        open_dev.append(e0.elapsed_time(e1))
    if w.oneshot is not None:
        for _ in range(5):
            flush.sum()
            torch.cuda.synchronize()
            t = time.perf_counter()
            r3 = w.oneshot()
            shots.append(time.perf_counter() - t)
            assert r3.ok
    k = int(np.argmin([a + b for a, b in zip(opens, passes)]))
    best = min(both)
    return {"open_plus_pass_ms": best * 1e3, "rows_per_s": w.units / best,
            "split": {"open_ms": opens[k] * 1e3, "open_device_span_ms": open_dev[k], "cold_pass_ms": passes[k] * 1e3,
                      "rows_per_s": w.units / (opens[k] + passes[k])},
            "one_shot_c_entry_ms": min(shots) * 1e3 if shots else None,
            "one_shot_rows_per_s": w.units / min(shots) if shots else None,
            "note": "inputs resident in HBM; caches flushed (2 GiB sweep) before every repetition; `open_plus_pass_ms` = session open (device-side "
                    "index / packed-key / directory builds from a per-device buffer arena, no host synchronisation) + launch + collect, one wall-clock "
                    "interval; `split` re-flushes and synchronises between open and pass (the round-2 definition); best of 5"}


def live_pmc_traffic(core, log_rows):
    """HBM-side bytes of ONE one-shot step, measured now: two short child runs of this file's own one-shot step under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE — separate passes, counters only, as the MI355X guide prescribes), the same
    corrections as the committed profile's (oneshot_profile_numbers).  Steps are counted by the fill kernel's dispatches (one per
    zk_evm_verify).  Returns (bytes per step or None, how it was obtained)."""
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    kernels = {}
    steps = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="zk_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child",
               "--log-rows", str(log_rows), "--steps", "4", "--warmup", "2"]
        try:
            child = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)  # its own process group: a pass that overruns is ended with everything it started
            try:
                rc = child.wait(timeout=120)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(child.pid, signal.SIGKILL)
                child.wait()
                raise
            if rc:
                raise subprocess.CalledProcessError(rc, cmd)
            agg = {}
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                import csv

                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter:
                        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
                        a[0] += 1
                        a[1] += float(r["Counter_Value"])
            for name, (n, tot) in agg.items():
                kernels.setdefault(name, {}).setdefault("pmc", {})[counter] = {"dispatches": n, "avg_per_dispatch": tot / n}
                if "evm_open_fill_kernel" in name:
                    steps = n
        except Exception as e:  # noqa: BLE001 — the committed profile stays the source then, and the line says so
            shutil.rmtree(out, ignore_errors=True)
            return None, f"live rocprofv3 --pmc {counter} pass failed ({type(e).__name__})"
        shutil.rmtree(out, ignore_errors=True)
    if not steps:
        return None, "live rocprofv3 passes saw no one-shot steps"
    traffic, _ = core.oneshot_profile_numbers({"bench_line": {"steps": steps, "warmup": 0}, "kernels": kernels})
    return traffic, f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over {steps} one-shot steps of a child process (bench.py live_pmc_traffic)"


def config0_bytecode(args):
    """BASELINE configs[0]: the Bytecode circuit over ONE 256-byte contract, k = 9 (512 rows, bytecode_circuit.py:37,104) — the
    reference's own CPU-runnable case, "pure CPU path (plumbing, no GPU)": timed through the CPU backend behind the same C ABI
    (libzkevm_cpu.so, one core), beside the reference's own figure from the build container."""
    import numpy as np

    from zkevm_specs_amd import _lib as zlib, oneshot as zoneshot
    from zkevm_specs_amd.synth import synth_bytecode_witness

    code = bytes(np.random.default_rng(1).integers(0, 256, 256, dtype=np.uint8))
    r = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % (1 << 253)
    cols, keccak = synth_bytecode_witness([code], 9, r)
    zlib.set_cpu_threads(1)
    reps = 200
    res, _ = zoneshot.bytecode_verify(cols, keccak, r, device="cpu")
    assert res.ok and res.rows_evaluated == 512
    t = time.perf_counter()
    for _ in range(reps):
        zoneshot.bytecode_verify(cols, keccak, r, device="cpu")
    dt = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(20):
        res_g, _ = zoneshot.bytecode_verify(cols, keccak, r)
    dt_g = (time.perf_counter() - t) / 20
    assert res_g.ok
    blk = {"workload": "Bytecode circuit, one 256-byte contract, k = 9: 512 rows (BASELINE configs[0]: the CPU-runnable plumbing case)",
           "value": 512 / dt, "unit": "rows/s", "steps": reps, "warmup": 1, "ms_per_step": dt * 1e3, "units_per_pass": 512,
           "backend": "libzkevm_cpu.so, 1 core (one-shot zk_bytecode_verify incl. its keccak-table index build and the ctypes call)",
           "hip_one_shot": {"rows_per_s": 512 / dt_g, "ms": dt_g * 1e3,
                            "note": "the same one-shot call on the MI355X incl. H2D staging of the 200 KB witness: launch-latency bound at 512 rows"}}
    if not args.no_cpu_baseline:
        ref_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_reference.json")), reverse=True)
        ref = json.load(open(ref_file[0])) if ref_file else None
        if ref and "bytecode" in ref:
            blk["cpu_baseline"] = {"value": ref["bytecode"]["rows_per_s"], "unit": "rows/s", "cores": 1, "kind": "reference", "measured_on": "build container",
                                   "sample": f"check_bytecode_row of the unmodified reference over the 512 rows of this configuration, build container ({os.path.basename(ref_file[0])}): a cross-box figure"}
    return blk


def own_process_config(name, log_rows, steps, warmup, args, with_cpu_baseline, extra=()):
    """One other configuration measured in a process of its own (`python bench.py --workload <name> ...` as a child, its full record
    read back): the block pass and the Tx / Sig pass are several concurrent streams, and inside the default line's process — where
    the batch entry's pipeline streams, side streams and the earlier configurations' streams exist by then — they share hardware
    queues (round 5: block pass 0.352 ms in-process against 0.331 ms alone, Tx + Sig at 2^11 1.48 against 1.18 ms).  None on any
    failure: the caller then measures in-process."""
    import subprocess
    import tempfile

    fd, path = tempfile.mkstemp(prefix="zk_bench_child_", suffix=".json", dir="/tmp")
    os.close(fd)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", name, "--log-rows", str(log_rows), "--steps", str(steps), "--warmup", str(warmup),
           "--no-other-configs", "--no-cold-leg", "--no-fresh-leg"] + ([] if with_cpu_baseline else ["--no-cpu-baseline"]) + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        t0 = time.perf_counter()
        subprocess.run(cmd, env=dict(env, ZK_BENCH_FULL=path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
        wall = time.perf_counter() - t0
        d = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    cfg = dict(d.get("config") or {})
    blk = {"workload": cfg.pop("workload", name), "value": d["value"], "unit": "txs/s" if name == "tx" else "rows/s", "steps": d["steps"], "warmup": d["warmup"],
           "pre_ramp_steps": d.get("pre_ramp_steps"), "ms_per_step": d["ms_per_step"], "units_per_pass": d["value"] * d["ms_per_step"] / 1e3,
           "process": "own process: " + " ".join(cmd[1:]), "process_wall_s": wall, "roofline": d["roofline"], "config": cfg}
    if d.get("cpu_baseline"):
        blk["cpu_baseline"] = d["cpu_baseline"]
    if d.get("block_oneshot"):
        blk["block_oneshot"] = d["block_oneshot"]
    return blk


def other_configs(core, ctx, args):
    """BASELINE configs[0], [1], [3], [4] on the driver's clock, after the headline: each one timed like the headline (barrier +
    synchronize around K passes), a reduced K so that they add well under a minute of GPU time (witness synthesis is
    host work outside the timed regions).  State also at 2^20 rows (past the 256 MiB Infinity Cache: the HBM figure) and, at
    2^16, with a cold-cache leg (its 120 MB witness otherwise never leaves the Infinity Cache between passes)."""
    out = {"bytecode_256B": config0_bytecode(args)}
    torch = ctx.torch
    # tx 2^11: ONE GPU's shard of BASELINE configs[3] at 8 GPUs (2^14 txs over 8 ranks) — the ECDSA launch is a dependent chain per
    # signature, so this is what every rank of the 8-GPU run takes per pass, and why that configuration's strong scaling is flat
    for name, log_rows, steps, warmup in (("state", 16, 50, 5), ("state", 20, 20, 3), ("tx", 14, 10, 2), ("tx", 11, 10, 2), ("super", 20, 20, 3)):
        if name in ("tx", "super") and os.environ.get("ZK_BENCH_INPROCESS") != "1":  # the multi-stream passes: a process of their own
            want_cpu = not args.no_cpu_baseline and not (name == "tx" and log_rows == 11)
            blk = own_process_config(name, log_rows, steps, warmup, args, want_cpu)
            if blk is not None:
                out[f"{name}_2p{log_rows}"] = blk
                if name == "super":  # the same block with the State rows in their compact form (ZK_OPT_STATE_COMPACT): a labelled second figure
                    blk_c = own_process_config(name, log_rows, steps, warmup, args, False, extra=("--state-compact",))
                    if blk_c is not None:
                        out[f"{name}_2p{log_rows}_compact"] = blk_c
                    # ... and with no State witness at all: the rows evaluated from the RW table where they are computed (a labelled third figure)
                    blk_f = own_process_config(name, log_rows, steps, warmup, args, False, extra=("--state-fused",))
                    if blk_f is not None:
                        blk_f.pop("block_oneshot", None)  # (the block one-shot is the same call whatever the resident form)
                        out[f"{name}_2p{log_rows}_fused"] = blk_f
                continue
        t_build = time.perf_counter()
        w = core.BUILDERS[name](ctx, log_rows, False)
        t_build = time.perf_counter() - t_build
        dt, res = core.timed_passes(ctx, w.sess, steps, warmup)
        per_circuit = None
        if name == "super":
            res, per_circuit = core.resolve_super(w, res)
        assert res.fail_count == 0, f"{name}: synthetic witness must satisfy every constraint"
        cold_ms = None
        if name == "state" and not args.no_cold_leg:
            flush = torch.zeros(1 << 29, dtype=torch.int32, device="cuda")
            for _ in range(8):
                flush.sum()
                w.sess.launch()
            cold_ms = w.sess.collect().kernel_ms
            del flush
        roof, profile, profile_src = core.roofline_block(w, res, 1, False, cold_ms)
        if name == "tx":
            core.tx_extras(roof, res, profile, profile_src)
        if per_circuit is not None:
            roof["per_circuit"] = per_circuit
        blk = {"workload": w.workload, "value": w.total_units * steps / dt, "unit": "txs/s" if name == "tx" else "rows/s", "steps": steps, "warmup": warmup,
               "ms_per_step": dt / steps * 1e3, "units_per_pass": w.total_units, "witness_build_s": t_build,
               "roofline": roof, "config": w.extra_cfg}
        blk["pre_ramp_steps"] = core.timed_passes.last_ramp
        if not args.no_cpu_baseline and not (name == "state" and log_rows == 20) and not (name == "tx" and log_rows == 11):
            blk["cpu_baseline"] = cpu_baseline(name, w)
        w.sess.close()
        out[f"{name}_2p{log_rows}"] = blk
        del w
        torch.cuda.empty_cache()
    return out


def marshalling_sample(wire_h, n_steps=1 << 14):
    """flatten_evm (reference-shaped Python objects -> wire arrays, zkevm_specs_amd/flatten.py) timed on a bounded prefix of this run's
    trace: the objects are rebuilt from the wire first (zkevm_specs_amd/objects.py), which is not part of the figure.  The prefix carries
    the WHOLE bytecode table (43,913 rows whatever the prefix): rounds 4-5 sampled 2^10 steps, where that table is most of the work."""
    import numpy as np

    from zkevm_specs_amd import flatten, objects

    w = {k: v for k, v in wire_h.items()}
    w["steps"] = np.ascontiguousarray(w["steps"][: n_steps + 1])
    hi = int(w["steps"][-1, 1, 0]) + 64  # rw_counter of the last sampled step: the RW rows the prefix can look up
    base = int(w["rw"][0, 0, 0])
    w["rw"] = np.ascontiguousarray(w["rw"][: max(hi - base, 1)])
    w["rw_flags"] = np.ascontiguousarray(w["rw_flags"][: len(w["rw"])])
    tables, steps = objects.evm_from_wire(w)
    was, legs = flatten.USE_EXT, {}
    try:
        for name, use in (("extension", True), ("python_loops", False)):
            if use and flatten._ext is None:
                continue
            flatten.USE_EXT = use
            t = time.perf_counter()
            out = flatten.flatten_evm(tables, steps)
            dt = time.perf_counter() - t
            cells = sum(int(v.size) // 4 for v in out.values() if hasattr(v, "dtype") and v.dtype == np.uint64)
            legs[name] = {"seconds": dt, "steps_per_s": n_steps / dt, "cells_per_s": cells / dt}
    finally:
        flatten.USE_EXT = was
    best = legs.get("extension") or legs["python_loops"]
    return {"steps": n_steps, "cells": cells, "seconds": best["seconds"], "steps_per_s": best["steps_per_s"], "cells_per_s": best["cells_per_s"], "cores": 1,
            "legs": legs,
            "note": "the walk over the objects' attributes in C (zkevm_specs_amd/_flatten_ext, csrc/flatten_ext.c) + de-duplication / ordering on the "
                    "packed rows in numpy; `python_loops` = the per-cell Python that defines the result (`x.expr().n` -> 4 x u64).  A caller that keeps "
                    "its witness in wire arrays (device-side assignment, zk_state_assign / zk_bytecode_assign / zk_copy_assign) pays neither"}


def block_oneshot(parts, to_dev, device=0, reps=8, copies=3, state_compact=False):
    """BASELINE config 5 as a ONE-SHOT: zk_block_verify on a block not touched before (rotating over `copies` device-resident
    copies of the raw inputs) — the keccak table, the Bytecode / Copy / State assignments (State = the RW table re-keyed and radix-sorted
    on the device), the six opens, one pass of every circuit, collects, closes.  Wall clock per block, median of `reps` (3 untimed first)."""
    import torch

    from zkevm_specs_amd.block import stage_block, verify_block_native

    blocks = [stage_block(parts, to_dev) for _ in range(copies)]
    times = []
    warm = int(os.environ.get("ZK_ONESHOT_WARM", "3"))
    for r in range(reps + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        results, total, ends = verify_block_native(blocks[r % copies], device, state_compact)
        t1 = time.perf_counter()
        assert total == 0, {k: (v.fail_count, v.first_fail_row, v.first_fail_code) for k, v in results.items()}
        if r >= warm:
            times.append((t1 - t0) * 1e3)
    times.sort()
    rows = sum(v.rows_evaluated for v in results.values())
    ms = times[len(times) // 2]
    return {"ms": ms, "min_ms": times[0], "max_ms": times[-1], "rows": rows, "rows_per_s": rows / (ms / 1e3), "reps": reps, "copies": copies,
            "chain_end_ms": dict(zip(("state", "keccak", "copy", "rest"), ends[:4])),
            "note": "wall clock of zk_block_verify (block.verify_block_native): four persistent host threads / HIP streams inside the library (State chain: class scan + radix sort + mock MPT + "
                    "the State circuit on rows evaluated where op2row computes them (zk_state_verify_from_rw; with state_compact the 15-cell witness is written and read back); "
                    "contracts' keccak -> Bytecode circuit; copy assignment + SHA3 keccak -> EVM open + pass; Bytecode assignment, Exp, Tx, Copy circuit); "
                    "every derived table and witness is rebuilt on the device for every block"}


def effective_cores():
    """Cores this process may actually burn: the smaller of its affinity mask and the cgroup's CFS quota.  On the round-4 GPU box
    os.cpu_count() said 256 while all-core passes after the first took 99.8 / 139.7 / 200.1 ms — multiples of the 100 ms CFS
    period: 256 OpenMP threads spend the quota in a burst (pass 0: 10.9 ms) and are then throttled.  Returns (cores, how)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "affinity mask"
    try:
        quota = period = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):  # cgroup v2: "<quota|max> <period>"
            q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota, period = (None if q == "max" else float(q)), float(p_)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota, period = (None if q <= 0 else q), float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota and period and quota / period < n:
            n, how = max(1, int(quota / period)), f"cgroup CFS quota {quota / period:.1f} cores"
    except (OSError, ValueError):
        pass
    return n, how


def cpu_baseline(workload, w):
    """CPU legs on rank 0's host cores, bounded samples.  `value` is the oracle port's (kind "port"), timed on this box: the
    reference is Python and does not travel to the GPU box, and bench.py never reads /root/reference.  The committed
    build-container measurement of the reference itself (tools/time_reference.py -> profiles/r*_cpu_reference.json) rides along as
    the cross-box leg `reference_build_container`."""
    import numpy as np

    cores_total, cores_how = effective_cores()
    ref_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_reference.json")), reverse=True)
    ref = json.load(open(ref_file[0])) if ref_file else None
    legs = {}
    if workload in ("evm", "super"):
        from oracle import evm_oracle, wire

        from tests.evm_cases import to_witness

        ev = w.wire_h if workload == "evm" else dict(w.env["parts"]["evm"])
        if workload == "super" and "copy_events" in w.env["parts"]:
            # the block's copy table on the host (on the device it is zk_copy_assign's output): the oracle's restatement of the same
            # assignment — test infrastructure, used here only to hand the CPU legs the witness the device evaluates
            from oracle import copy_assign_oracle
            from zkevm_specs_amd.wire import rows_to_rowmajor

            ce = w.env["parts"]["copy_events"]
            ev["copy"] = rows_to_rowmajor(copy_assign_oracle.assign(wire.rowmajor_to_rows(ce["events"]), ce["flags"].tolist(), ce["data"],
                                                                     ce["offsets"], ce["r"])[2], 14)
        sample = min(int(ev["steps"].shape[0]) - 1, 1 << 15)
        W = to_witness(dict({k: v for k, v in ev.items() if k != "meta"}, steps=ev["steps"][: sample + 1]))
        tc = time.perf_counter()
        st = evm_oracle.verify_steps(W)
        tc = time.perf_counter() - tc
        assert not any(st)
        legs["port"] = {"value": sample / tc, "unit": "rows/s", "cores": 1, "units": sample,
                        "sample": f"first {sample} step pairs of the same trace, pure-Python oracle with dict-indexed lookups (oracle/evm_oracle.py)"}
        from zkevm_specs_amd import _lib as zlib, engine as zengine

        hs = min(int(ev["steps"].shape[0]) - 1, 1 << 18)
        sub = {k: v for k, v in ev.items() if k != "meta"}
        sub["steps"] = ev["steps"][: hs + 1]
        for threads, key in ((1, "cpu_backend_1core"), (cores_total, "cpu_backend_allcores")):
            zlib.set_cpu_threads(threads)
            tc = time.perf_counter()
            with zengine.open_evm(sub, device="cpu") as cs:
                t_open = time.perf_counter() - tc
                runs = [cs.run() for _ in range(4 if threads == 1 else 8)]  # pass 0 (thread-pool start, and a CFS burst) is not counted
            assert all(r.ok for r in runs)
            ms_sorted = sorted(r.kernel_ms for r in runs[1:])
            med = ms_sorted[len(ms_sorted) // 2]
            legs[key] = {"value": hs / (med / 1e3), "unit": "rows/s", "cores": threads, "pass_ms": med, "pass0_ms": runs[0].kernel_ms, "open_s": t_open,
                         "pass_ms_spread": {"min": ms_sorted[0], "median": med, "max": ms_sorted[-1], "passes": len(ms_sorted)},
                         "sample": f"{hs} step pairs through libzkevm_cpu.so (ZK_BACKEND=cpu: the kernels' own per-step functions compiled for the "
                                   f"host, OpenMP over the pairs; csrc/cpu_backend.cpp), MEDIAN of passes 1..{len(ms_sorted)} with tables and indices "
                                   f"resident, {threads} thread(s) = {cores_how if threads > 1 else 'one core'} — the optimised-CPU line"}
        if ref and "evm" in ref:  # the committed build-container measurement of the reference itself (tools/time_reference.py): a cross-box leg
            e = ref["evm"]
            legs["reference"] = {"value": e["extrapolated_2p18"]["pairs_per_s"], "unit": "rows/s", "cores": 1,
                                 "measured": [{"step_pairs": p["step_pairs"], "table_rows": p["rw_rows"] + p["bytecode_rows"],
                                               "rows_per_s": p["pairs_per_s"]} for p in e["measured"]],
                                 "fit": e["fit"], "extrapolated": True,
                                 "sample_short": "unmodified reference verify_steps, 2^4/2^6/2^8-pair prefixes, build container; 2^18 EXTRAPOLATED",
                                 "sample": "verify_steps of the unmodified reference on 2^4 / 2^6 / 2^8-pair prefixes of this trace, build container "
                                           f"({os.path.basename(ref_file[0])}); the 2^18 figure is EXTRAPOLATED from the fit (linear-scan lookups, table.py:864-884)"}
    elif workload == "tx":
        from oracle import sign_oracle

        tx, sg = w.env["tx"], w.env["sig"]
        from zkevm_specs_amd import _lib as zlib, oneshot as zoneshot

        n = int(tx["bytes"].shape[0])
        R_TX = 0x1F2E3D4C5B6A79881726354433221100FFEEDDCCBBAA99887766554433221
        r4 = np.frombuffer(int(R_TX).to_bytes(32, "little"), dtype="<u8").copy()
        def both_circuits(sample):
            """Tx circuit + Sig circuit over the first `sample` txs through the one-shot entries of the CPU backend; wall clock"""
            tc = time.perf_counter()
            for wire_, layout, is_sig in ((tx, 1, False), (sg, 2, True)):
                wslice = {k: (np.ascontiguousarray(v[:sample]) if k in ("bytes", "meta") else (np.ascontiguousarray(v[:, :sample]) if k == "cells" else v))
                          for k, v in wire_.items()}
                if not is_sig:
                    wslice["tx_rows"], wslice["tx_flags"] = wire_["tx_rows"][: 12 * sample], wire_["tx_flags"][: 12 * sample]
                _, est = zoneshot.ecdsa_verify(wslice["bytes"], np.ascontiguousarray(wslice["meta"][:, 3]) if is_sig else None, layout=layout, device="cpu")
                wslice["meta"] = wslice["meta"].copy()
                wslice["meta"][:, 0] = est
                r_cpu, _ = zoneshot.sign_verify(wslice, R_TX, is_sig, device="cpu")
                assert r_cpu.ok, (is_sig, r_cpu)
            return time.perf_counter() - tc

        for threads, key, sample, reps in ((1, "cpu_backend_1core", min(n, 1 << 9), 1), (cores_total, "cpu_backend_allcores", n, 3)):
            zlib.set_cpu_threads(threads)
            tc = sorted(both_circuits(sample) for _ in range(reps))[reps // 2]  # median (all cores: repetition 0 pays for the OpenMP thread pool)
            legs[key] = {"value": sample / tc, "unit": "txs/s", "cores": threads,
                         "sample": f"first {sample} txs: Tx circuit + Sig circuit incl. both ECDSA verifications through libzkevm_cpu.so (ZK_BACKEND=cpu: "
                                   "the kernels' own per-unit functions compiled for the host, OpenMP; csrc/cpu_backend.cpp), wall clock of the one-shot "
                                   f"entries, median of {reps}"}
        from oracle import ecdsa_oracle, wire

        sample_p = min(n, 1 << 5)
        tc = time.perf_counter()
        b = tx["bytes"][:sample_p]
        packed = np.stack([b[:, 0], b[:, 1], b[:, 4, ::-1], b[:, 7], b[:, 8]], axis=1)  # Tx units carry msg_hash little-endian
        meta = tx["meta"][:sample_p].copy()
        meta[:, 0] = ecdsa_oracle.verify_packed(packed)
        st = sign_oracle.verify_units(b, tx["cells"][:, :sample_p], meta, wire.rowmajor_to_rows(tx["keccak"]), int.from_bytes(r4.tobytes(), "little"), False,
                                      wire.rowmajor_to_rows(tx["tx_rows"][: 12 * sample_p]), tx["tx_flags"][: 12 * sample_p])
        tc = time.perf_counter() - tc
        assert not any(st)
        legs["port"] = {"value": sample_p / tc, "unit": "txs/s", "cores": 1, "units": sample_p,
                        "sample": f"first {sample_p} txs, Tx circuit with ECDSA verification, pure-Python oracle (oracle/sign_oracle.py + oracle/ecdsa_oracle.py)"}
        if ref and "tx" in ref:
            t = ref["tx"]
            legs["reference"] = {"value": t["txs_per_s"], "unit": "txs/s", "cores": 1, "extrapolated": False, "measured": t.get("measured"),
                                 "sample": f"tx_circuit.verify_circuit + sig_circuit.verify_circuit of the unmodified reference over {t['txs']} of these txs "
                                           f"(eth-keys stand-in: oracle/refshim), build container ({os.path.basename(ref_file[0])})"}
    else:
        from oracle import state_oracle, wire

        cols, flags, mpt = w.env["cols"], w.env["flags"], w.env["mpt"]
        sample = min(int(cols.shape[1]), 1 << 16)
        rows_i = wire.colmajor_to_rows(cols[:, :sample])
        mpt_i = wire.rowmajor_to_rows(mpt)
        tc = time.perf_counter()
        state_oracle.verify_rows(rows_i, flags[:sample], mpt_i)
        tc = time.perf_counter() - tc
        legs["port"] = {"value": sample / tc, "unit": "rows/s", "cores": 1, "units": sample,
                        "sample": f"first {sample} rows of the same witness, pure-Python oracle (oracle/state_oracle.py)"}
        from zkevm_specs_amd import _lib as zlib, engine as zengine

        for threads, key in ((1, "cpu_backend_1core"), (cores_total, "cpu_backend_allcores")):
            zlib.set_cpu_threads(threads)
            with zengine.open_state(cols, flags, mpt, device="cpu") as cs:
                runs = [cs.run() for _ in range(6)]  # pass 0 (thread-pool start, and a CFS burst) is not counted
            assert all(r.ok for r in runs)
            ms_sorted = sorted(r.kernel_ms for r in runs[1:])
            med = ms_sorted[len(ms_sorted) // 2]
            legs[key] = {"value": int(cols.shape[1]) / (med / 1e3), "unit": "rows/s", "cores": threads, "pass_ms": med, "pass0_ms": runs[0].kernel_ms,
                         "sample": f"all {int(cols.shape[1])} rows through libzkevm_cpu.so (ZK_BACKEND=cpu: the kernel's own per-row function compiled for "
                                   f"the host, OpenMP over the rows; csrc/cpu_backend.cpp), MEDIAN of passes 1..5 with the witness and the MPT index resident, "
                                   f"{threads} thread(s)"}
        if ref and "state" in ref:
            legs["reference"] = {"value": ref["state"]["rows_per_s"], "unit": "rows/s", "cores": 1, "extrapolated": False,
                                 "sample": f"check_state_row of the unmodified reference over all {ref['state']['rows']} rows of this witness, build container "
                                           f"({os.path.basename(ref_file[0])})"}
    # the headline baseline is timed on THIS box (the oracle port); the committed build-container figure of the reference stays beside it
    if "reference" in legs:
        legs["reference_build_container"] = legs.pop("reference")
    head = legs.get("reference") or legs.get("port") or legs["cpu_backend_1core"]
    return {"value": head["value"], "unit": head["unit"], "cores": 1,
            "kind": "reference" if "reference" in legs else "port",
            "sample": head["sample"], "sample_short": head.get("sample_short"), "sample_pairs": head.get("units"), "legs": legs,
            "this_box_cores_total": cores_total, "this_box_cores_source": cores_how, "this_box_cpu_count": os.cpu_count(),
            "reference_measured_on": (None if not ref else dict(ref.get("host", {}), note="the BUILD CONTAINER, not this GPU box: a Python reference does not "
                                                                "travel, so `reference_build_container` is a cross-box figure; `port` and the `cpu_backend_*` legs are timed on this box")),
            }
