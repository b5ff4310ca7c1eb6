#!/usr/bin/env python3
"""bench.py — constraint-rows/s of the hot path on N MI355X GPUs (one process per GPU).

A "step" is one evaluation pass of the circuit kernel(s) over the rank's witness shard, with all inputs already resident
in HBM.  `--scaling weak` (default): every rank holds its own 2^log_rows-row witness.  `--scaling strong`: ONE global
witness — rank 0 builds it, the lookup tables are replicated with a broadcast (RCCL), the rows are sharded with a halo
(zkevm_specs_amd/distributed.py).  The only collective inside the timed region's result path is the tally exchange
(one all-gather of three words per rank).

The JSON line carries, next to the contract's fields:
  roofline        `achieved` / `frac` are PHYSICAL: HBM-side bytes per launch from the committed rocprofv3 counter passes
                  (profiles/r02_*_profile.json, tools/profile_bench.sh) over this run's live kernel time; the algorithmic
                  figure (SURVEY.md §8d bytes / kernel time) is reported beside it as `algorithmic`; `valu` is the
                  VALU-issue roofline of the same kernel (SQ counters), the resource that actually binds this path.
  fresh_witness   what a verifier pays for a witness it sees once: session open (index / packed-key / directory builds on
                  the device, inputs resident) + one pass over cold caches.
  cpu_baseline    `port` (oracle/, pure Python, dict-indexed lookups), `hostsim` (the kernels' own sources built for the CPU,
                  tests/hostsim: the "optimised CPU" line) — both timed here on a bounded sample — and `reference`: the
                  unmodified reference timed in the build container (tools/time_reference.py -> profiles/r02_cpu_reference.json;
                  /root/reference does not exist on the GPU box).
  host_path       marshalling (Python objects -> wire arrays, flatten.py) and H2D staging, reported separately.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (≈ 6.3 TB/s achievable)
N_SIMD = 1024                # 256 CUs x 4 SIMDs
SHADER_CLOCK_HZ = 2.4e9      # nominal; profiled passes run lower (guide: 1.9 - 2.3 GHz effective)
FETCH_GATHER_CORRECTION = 1.0 / 0.95  # profiles/r01_fetch_size_calibration.txt: per-lane 416 / 448-byte record gathers
FETCH_STREAM_CORRECTION = 2.0         # guide: wide coalesced streaming reads report half the bytes


def load_profile(workload, log_rows):
    """committed rocprofv3 summary of this workload / size (tools/profile_bench.sh), newest round first"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_2p{log_rows}_profile.json")), reverse=True)
    return (json.load(open(files[0])), "profiles/" + os.path.basename(files[0])) if files else (None, None)


def kernel_counters(profile, needle):
    if not profile:
        return None
    for name, k in profile["kernels"].items():
        if all(n in name for n in needle):
            return dict(k, name=name)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="evm", choices=["evm", "state", "super", "tx"])
    ap.add_argument("--log-rows", type=int, default=None, help="log2 rows per GPU (weak) or in total (strong); default 18 evm, 16 state, 20 super, 14 tx")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold-leg", action="store_true", help="skip the cold-cache kernel timing after the timed region")
    ap.add_argument("--no-fresh-leg", action="store_true", help="skip the fresh-witness (open + one cold pass) timing")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    # test hooks (single-GPU dry run of the N > 1 path): ZK_BENCH_DEVICE pins every rank to one GPU, ZK_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device
    if os.environ.get("ZK_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["ZK_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from zkevm_specs_amd import _lib, distributed, engine

    log_rows = args.log_rows if args.log_rows is not None else {"evm": 18, "state": 16, "super": 20, "tx": 14}[args.workload]
    n = 1 << log_rows
    strong = args.scaling == "strong" and world > 1
    assert not (args.scaling == "strong" and args.workload not in ("evm", "state")), "--scaling strong: evm / state workloads"
    _lib.init(local_rank)
    # one real (non-default) stream shared by torch and the engine: uploads, the passes and the cold leg's flush kernel are
    # ordered on it (torch's default stream is handle 0, which the engine reads as "use your own stream")
    bench_stream = torch.cuda.Stream()
    torch.cuda.set_stream(bench_stream)
    _lib.check(_lib.load().zk_set_stream(bench_stream.cuda_stream), "zk_set_stream")

    def to_dev(x):
        if x.dtype == np.uint8:
            return torch.from_numpy(x).cuda()
        if x.dtype.itemsize == 2:
            return torch.from_numpy(x.view(np.int16)).cuda()
        return torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()

    h2d = {"bytes": 0, "seconds": 0.0}

    def upload(arrays):
        """numpy dict -> device tensors, timed (pageable host memory, the path a ctypes caller takes)"""
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = {k: to_dev(v) for k, v in arrays.items()}
        torch.cuda.synchronize()
        h2d["seconds"] += time.perf_counter() - t
        h2d["bytes"] += sum(int(v.nbytes) for v in arrays.values())
        return out

    def replicate(arrays, src=0):
        """strong scaling: rank `src`'s arrays on every rank (shapes first, then one broadcast per array: RCCL over xGMI)"""
        names = sorted(arrays) if rank == src else None
        meta = [[(k, tuple(arrays[k].shape), str(arrays[k].dtype)) for k in names]] if rank == src else [None]
        dist.broadcast_object_list(meta, src=src)
        out = {}
        for k, shape, dt in meta[0]:
            if rank == src:
                t = to_dev(arrays[k])
            else:
                tdt = torch.uint8 if dt == "uint8" else (torch.int64 if dt == "uint64" else torch.int32)
                t = torch.empty(shape, dtype=tdt, device="cuda")
            if t.numel():
                dist.broadcast(t, src=src)
            out[k] = t
        return out

    fresh = None  # (open_fn, close_fn) of the fresh-witness leg
    row_offset = 0
    wire_h = None
    if args.workload == "evm":
        from zkevm_specs_amd.synth_evm import synth_evm_trace

        if strong:
            wire_h = synth_evm_trace(n, seed=3) if rank == 0 else None
            meta_box = [wire_h.pop("meta") if rank == 0 else None]
            dist.broadcast_object_list(meta_box, src=0)
            meta = meta_box[0]
            full = replicate(wire_h if rank == 0 else {})
            lo, hi = distributed.shard_bounds(n - 1, rank, world)
            wire_d = dict(full, steps=full["steps"][lo: hi + 1].contiguous())
            units, row_offset = hi - lo, lo
            total_units = n - 1
            algo_bytes = meta["algorithmic_bytes"] * units / (n - 1)
        else:
            wire_h = synth_evm_trace(n, seed=3 + rank)
            meta = wire_h.pop("meta")
            wire_d = upload(wire_h)
            units, row_offset = n - 1, rank * (n - 1)
            total_units = units * world
            algo_bytes = meta["algorithmic_bytes"]
        open_fn = lambda: engine.open_evm(wire_d, device=local_rank)  # noqa: E731
        sess = open_fn()
        fresh = open_fn
        kernel_name, kernel_needle = "evm_steps_kernel", ("evm_steps_kernel", "-1")
        workload = (f"EVM circuit, 2^{log_rows} execution steps {'in total' if strong else 'per GPU'}, mixed-opcode synthetic trace "
                    f"(BASELINE configs[2]); RW table {meta['n_rw']} rows, bytecode table {meta['n_bytecode']} rows")
        extra_cfg = {"steps_per_gpu": units + 1, "rw_rows": meta["n_rw"], "bytecode_rows": meta["n_bytecode"]}
    elif args.workload == "tx":
        # BASELINE configs[3]: Tx circuit over 2^log_rows signed synthetic txs per GPU; a pass = secp256k1 ECDSA verification of
        # every signature (fills the units' ecdsa_status column in HBM) + the SignVerify / copy-constraint kernel.  The public-key
        # hashes are keccak-256 digests built by the device table builder once per witness.
        from zkevm_specs_amd.synth import device_keccak_digests, synth_tx_witness

        r_tx = 0x1F2E3D4C5B6A79881726354433221100FFEEDDCCBBAA99887766554433221 + rank
        w_tx = synth_tx_witness(n, r_tx, seed=4 + rank, signed=True, digests_of=device_keccak_digests(r_tx))
        d_tx = upload(w_tx)

        class _TxPass:
            def __init__(self):
                self.ecdsa = engine.open_ecdsa(d_tx["bytes"], layout=engine.ECDSA_LAYOUT_TX_UNITS, out_dev=d_tx["meta"], out_stride=4,
                                               device=local_rank)
                self.sign = engine.open_sign(d_tx, r_tx, False, device=local_rank)

            def launch(self):
                self.ecdsa.launch()
                self.sign.launch()

            def collect(self):
                re_, rs_ = self.ecdsa.collect(), self.sign.collect()
                rs_.ecdsa_ms = re_.kernel_ms  # a signature that does not verify fails its unit in the Tx kernel already
                return rs_

            def close(self):
                self.ecdsa.close()
                self.sign.close()

        sess = _TxPass()
        units, row_offset = n, rank * n
        total_units = units * world
        algo_bytes = n * (8 * 32 + 288 + 2 * 5 * 32)
        kernel_name, kernel_needle = "sign_units_kernel", ("sign_units_kernel",)
        workload = f"Tx circuit, 2^{log_rows} signed synthetic txs per GPU (BASELINE configs[3]): ECDSA verification + SignVerify kernel per pass"
        extra_cfg = {"txs_per_gpu": n}
    elif args.workload == "super":
        # BASELINE configs[4]: EVM + State + Bytecode + Tx kernels over one witness set of 2^log_rows rows per GPU
        from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block

        # ONE consistent witness: the State rows are the EVM trace's RW table (synth_block.py), Copy / Exp circuits included
        parts = synth_super_block(log_rows, seed=5 + rank)
        super_meta = parts["meta"]
        sess = SuperCircuit(parts, device=local_rank, to_device=to_dev)
        units, row_offset = sum(sess.rows.values()), rank * sum(sess.rows.values())
        total_units = units * world
        tx_bytes = 8 * 32 + 288 + 2 * 5 * 32
        super_bytes = {"evm": super_meta["algorithmic_bytes"], "state": sess.rows["state"] * 57 * 32,
                       "bytecode": sess.rows["bytecode"] * 12 * 32, "tx": sess.rows["tx"] * tx_bytes,
                       "copy": sess.rows.get("copy", 0) * (20 + 14) * 32, "exp": sess.rows.get("exp", 0) * 21 * 32}
        algo_bytes = None  # per-circuit, resolved after the run (dominant kernel)
        kernel_name, kernel_needle = None, None
        workload = (f"Super circuit, ~2^{log_rows} rows per GPU over ONE consistent witness (BASELINE configs[4]; State rows = the EVM trace's RW "
                    "table re-keyed and re-sorted): " + ", ".join(f"{k} {v}" for k, v in sess.rows.items()) + " rows")
        extra_cfg = {"rows_per_gpu": dict(sess.rows), "state_assign_ms": sess.assign_ms}
    else:
        from zkevm_specs_amd.synth import synth_state_witness

        if strong:
            host = None
            if rank == 0:
                cols, flags, mpt = synth_state_witness(n, seed=2)
                host = {"cols": cols, "flags": flags, "mpt": mpt}
            full = replicate(host if rank == 0 else {})
            lo, hi = distributed.shard_bounds(n, rank, world)
            idx = torch.arange(lo - 1, hi + 1, device="cuda") % n  # the rank's rows + one halo row on each side
            d_cols, d_flags, d_mpt = full["cols"][:, idx].contiguous(), full["flags"][idx].contiguous(), full["mpt"]
            units, row_offset, total_units = hi - lo, lo, n

            def open_fn():
                s = engine.open_state(d_cols, d_flags, d_mpt, device=local_rank)
                s.set_range(1, 1 + units)
                return s
        else:
            cols, flags, mpt = synth_state_witness(n, seed=2 + rank)
            d = upload({"cols": cols, "flags": flags, "mpt": mpt})
            d_cols, d_flags, d_mpt = d["cols"], d["flags"], d["mpt"]
            units, row_offset = n, rank * n
            total_units = units * world
            open_fn = lambda: engine.open_state(d_cols, d_flags, d_mpt, device=local_rank)  # noqa: E731
        sess = open_fn()
        fresh = open_fn
        algo_bytes = units * 57 * 32  # every witness cell counted once
        kernel_name, kernel_needle = "state_rows_kernel", ("state_rows_kernel",)
        workload = f"State circuit, 2^{log_rows} RW rows {'in total' if strong else 'per GPU'} (BASELINE configs[1])"
        extra_cfg = {"rows_per_gpu": units, "mpt_rows": int(d_mpt.shape[0])}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sess.launch()
    sess.collect()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sess.launch()
    res = sess.collect()
    barrier()
    dt = time.perf_counter() - t0
    per_circuit = None
    if args.workload == "super":
        results, total_fail_local, first_local = res
        per_circuit = {k: {"rows": sess.rows[k], "kernel_ms": r.kernel_ms,
                           "algorithmic_GBps": super_bytes[k] / (r.kernel_ms / 1e3) / 1e9} for k, r in results.items()}
        dom = max(results, key=lambda k: results[k].kernel_ms)
        kernel_name = {"evm": "evm_steps_kernel", "state": "state_rows_kernel", "bytecode": "bytecode_rows_kernel",
                       "tx": "sign_units_kernel", "copy": "copy_rows_kernel", "exp": "exp_rows_kernel"}[dom]
        kernel_needle = (kernel_name,) + (("-1",) if dom == "evm" else ())
        algo_bytes = super_bytes[dom]

        class _Tally:
            fail_count = total_fail_local
            first_fail_row = None if first_local is None else first_local[1]
            first_fail_code = 0 if first_local is None else first_local[2]
            kernel_ms = results[dom].kernel_ms

        res = _Tally

    # Cold-cache leg (outside the timed region): the timed passes re-read the same witness, so page-table lines and part of
    # the rows are still in L2 / Infinity Cache from the previous pass.  Here every pass is preceded by a read-only sweep over
    # 2 GiB of unrelated data ON THE SAME STREAM (the flush is ordered before the pass).
    cold_ms = None
    flush = None
    if (not args.no_cold_leg or not args.no_fresh_leg) and args.workload not in ("super", "tx"):
        flush = torch.zeros(1 << 29, dtype=torch.int32, device="cuda")
    if not args.no_cold_leg and flush is not None:
        for _ in range(8):
            flush.sum()
            sess.launch()
        cold_ms = sess.collect().kernel_ms

    # Fresh-witness leg: a verifier sees each witness once.  open (device-resident inputs: index builds, packed key records,
    # density check, bytecode directory) + one pass over cold caches, wall clock, 3 repetitions.
    fresh_block = None
    if not args.no_fresh_leg and fresh is not None and flush is not None:
        opens, passes, open_dev = [], [], []
        for _ in range(3):
            flush.sum()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t = time.perf_counter()
            e0.record()
            s2 = fresh()
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
        k = int(np.argmin([a + b for a, b in zip(opens, passes)]))
        fresh_block = {"open_ms": opens[k] * 1e3, "open_device_span_ms": open_dev[k], "cold_pass_ms": passes[k] * 1e3,
                       "rows_per_s": units / (opens[k] + passes[k]),
                       "note": "inputs resident in HBM; open = hipMalloc + index / packed-key / directory builds (device kernels, two "
                               "host syncs); cold pass = launch + collect after a 2 GiB flush; best of 3 (wall clock)"}
    del flush

    total_fail, first_row, first_code = distributed.reduce_tally(res.fail_count, res.first_fail_row, res.first_fail_code,
                                                                 row_offset, device="cuda")
    t_max = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    assert total_fail == 0 and first_row is None, "synthetic witness must satisfy every constraint"

    if rank == 0:
        rows_total = total_units * args.steps
        kernel_s = res.kernel_ms / 1e3
        algo_gbps = algo_bytes / kernel_s / 1e9
        # HBM traffic and SQ counters come from separate rocprofv3 --pmc passes over this same command
        # (tools/profile_bench.sh); the committed per-dispatch summary is attached when there is one for this size
        profile, profile_src = load_profile(args.workload, log_rows)
        kc = kernel_counters(profile, kernel_needle) if world == 1 or not strong else None
        traffic = valu = None
        if kc and "pmc" in kc:
            pmc = kc["pmc"]
            corr = FETCH_GATHER_CORRECTION if kernel_name == "evm_steps_kernel" else FETCH_STREAM_CORRECTION
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = pmc["FETCH_SIZE"]["avg_per_dispatch"] * 1024.0 * corr + pmc["WRITE_SIZE"]["avg_per_dispatch"] * 1024.0
            if "SQ_ACTIVE_INST_VALU" in pmc:
                # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles summed over waves (guide, PMC table): VALU-active shader
                # cycles = 4 x counter; the chip offers N_SIMD x kernel cycles of VALU issue
                act = pmc["SQ_ACTIVE_INST_VALU"]["avg_per_dispatch"] * 4.0
                valu = {"insts_valu_per_launch": pmc.get("SQ_INSTS_VALU", {}).get("avg_per_dispatch"),
                        "active_valu_cycles_per_launch": act,
                        "wave_cycles_per_launch": pmc.get("SQ_WAVE_CYCLES", {}).get("avg_per_dispatch", 0) * 4.0,
                        "frac_of_issue_peak": act / (kernel_s * SHADER_CLOCK_HZ * N_SIMD),
                        "note": f"VALU-active cycles / ({N_SIMD} SIMDs x kernel time x {SHADER_CLOCK_HZ / 1e9:.1f} GHz nominal); counters from {profile_src}"}
        physical = traffic / kernel_s / 1e9 if traffic else None
        roofline = {
            "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            # physical HBM-side bytes moved per launch (PMC) over the live kernel time; falls back to the algorithmic figure,
            # labelled, when no counter pass is committed for this configuration
            "achieved": physical if physical is not None else algo_gbps,
            "frac": (physical if physical is not None else algo_gbps) / HBM_PEAK_GBPS,
            "frac_source": "pmc traffic / live kernel time" if physical is not None else "ALGORITHMIC bytes (no counter pass committed for this configuration)",
            "traffic": traffic, "traffic_source": profile_src if traffic else None,
            "algorithmic": {"bytes_per_launch": algo_bytes, "GBps": algo_gbps, "frac": algo_gbps / HBM_PEAK_GBPS,
                            "note": "SURVEY.md §8d bytes (every looked-up row at its wire size) / kernel time: work per byte budget, "
                                    "not HBM utilisation — lookups read packed key records, so it may exceed what the memory system moves"},
            "binding_resource": "VALU issue + dependent-lookup latency (integer-modular path); see `valu`",
            "valu": valu,
            "kernel": kernel_name, "kernel_ms": res.kernel_ms,
            "rocprof_avg_kernel_ms": None if not kc or "trace" not in kc else kc["trace"]["avg_ns"] / 1e6,
            "cold_cache": None if cold_ms is None else {
                "kernel_ms": cold_ms, "algorithmic_GBps": algo_bytes / (cold_ms / 1e3) / 1e9,
                "traffic_GBps": None if not traffic else traffic / (cold_ms / 1e3) / 1e9,
                "note": "same kernel, each pass preceded (same stream) by a 2 GiB read-only sweep: cold L2 / Infinity Cache / page-table lines"},
        }
        out = {
            "metric": "BN254 constraint-rows/sec",
            "value": rows_total / dt,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u256 (BN254 Fr, 4xu64 canonical cells; u32-limb Montgomery multiply)",
            "data": "synthetic",
            "config": dict({"workload": workload, "sharding": f"rows x{world} ({'one global witness, tables broadcast' if strong else 'independent witnesses'}), tally all-gather"},
                           **extra_cfg),
            "roofline": roofline,
        }
        if fresh_block is not None:
            out["fresh_witness"] = fresh_block
        if h2d["bytes"]:
            out["host_path"] = {"h2d_bytes": h2d["bytes"], "h2d_seconds": h2d["seconds"], "h2d_GBps": h2d["bytes"] / h2d["seconds"] / 1e9,
                                "rows_per_s_including_h2d": units / (h2d["seconds"] + dt / args.steps),
                                "note": "pageable host arrays -> HBM (torch .cuda()); never part of `value`"}
        if args.workload == "evm" and wire_h is not None and "host_path" in out and not args.no_cpu_baseline:
            out["host_path"]["marshalling"] = marshalling_sample(wire_h)
        if per_circuit is not None:
            # per-circuit HBM-side traffic from the committed counter passes of this workload (when there are any)
            names = {"evm": ("evm_steps_kernel", "-1"), "state": ("state_rows",), "bytecode": ("bytecode_rows_kernel",), "tx": ("sign_units_kernel",),
                     "copy": ("copy_rows_kernel",), "exp": ("exp_rows_kernel",)}
            for k, v in per_circuit.items():
                kc2 = kernel_counters(profile, names[k])
                if kc2 and "pmc" in kc2 and "FETCH_SIZE" in kc2["pmc"]:
                    corr = FETCH_STREAM_CORRECTION if k != "evm" else FETCH_GATHER_CORRECTION
                    v["traffic_bytes"] = kc2["pmc"]["FETCH_SIZE"]["avg_per_dispatch"] * 1024.0 * corr + kc2["pmc"].get("WRITE_SIZE", {}).get("avg_per_dispatch", 0) * 1024.0
                    v["traffic_GBps"] = v["traffic_bytes"] / (v["kernel_ms"] / 1e3) / 1e9
            out["roofline"]["per_circuit"] = per_circuit
        if args.workload == "tx":
            out["roofline"]["ecdsa_verify_kernel_ms"] = res.ecdsa_ms
            ke = kernel_counters(profile, ("ecdsa_verify_kernel",))
            if ke and "pmc" in ke and "SQ_ACTIVE_INST_VALU" in ke["pmc"]:
                act = ke["pmc"]["SQ_ACTIVE_INST_VALU"]["avg_per_dispatch"] * 4.0
                out["roofline"]["ecdsa_valu"] = {
                    "insts_valu_per_launch": ke["pmc"]["SQ_INSTS_VALU"]["avg_per_dispatch"], "active_valu_cycles_per_launch": act,
                    "wavefronts": ke["pmc"].get("SQ_WAVES", {}).get("avg_per_dispatch"),
                    "frac_of_issue_peak": act / ((res.ecdsa_ms / 1e3) * SHADER_CLOCK_HZ * N_SIMD),
                    "note": "the pass is VALU-issue bound: one wavefront per SIMD issues one VALU instruction per ~4 cycles (half the "
                            f"2-cycle SIMD rate, profiles/r02_valu_issue_rates.txt); counters from {profile_src}"}
            out["roofline"]["note"] = ("the pass is dominated by ecdsa_verify_kernel (integer-ALU bound, no HBM roofline); the roofline block "
                                       "describes the SignVerify kernel")
        if not args.no_cpu_baseline and args.workload != "tx":
            out["cpu_baseline"] = cpu_baseline(args.workload, units, wire_h, locals())
        print(json.dumps(out))
    sess.close()
    if world > 1:
        dist.destroy_process_group()


def marshalling_sample(wire_h, n_steps=1 << 10):
    """flatten_evm (reference-shaped Python objects -> wire arrays, zkevm_specs_amd/flatten.py) timed on a bounded prefix of this run's
    trace: the objects are rebuilt from the wire first (zkevm_specs_amd/objects.py), which is not part of the figure."""
    import numpy as np

    from zkevm_specs_amd import flatten, objects

    w = {k: v for k, v in wire_h.items()}
    w["steps"] = np.ascontiguousarray(w["steps"][: n_steps + 1])
    hi = int(w["steps"][-1, 1, 0]) + 64  # rw_counter of the last sampled step: the RW rows the prefix can look up
    base = int(w["rw"][0, 0, 0])
    w["rw"] = np.ascontiguousarray(w["rw"][: max(hi - base, 1)])
    w["rw_flags"] = np.ascontiguousarray(w["rw_flags"][: len(w["rw"])])
    tables, steps = objects.evm_from_wire(w)
    t = time.perf_counter()
    out = flatten.flatten_evm(tables, steps)
    dt = time.perf_counter() - t
    cells = sum(int(v.size) // 4 for v in out.values() if hasattr(v, "dtype") and v.dtype == np.uint64)
    return {"steps": n_steps, "cells": cells, "seconds": dt, "steps_per_s": n_steps / dt, "cells_per_s": cells / dt, "cores": 1,
            "note": "per-cell Python (`x.expr().n` -> 4 x u64); a caller that keeps its witness in wire arrays (device-side assignment, "
                    "zk_state_assign / zk_bytecode_assign / zk_copy_assign) never pays it"}


def cpu_baseline(workload, units, wire_h, env):
    """CPU legs on rank 0's host cores, bounded samples.  `value` is the reference's own figure when the committed
    build-container measurement exists (kind "reference"), else the oracle port's."""
    import ctypes
    import subprocess

    import numpy as np

    cores_total = os.cpu_count()
    ref_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_reference.json")), reverse=True)
    ref = json.load(open(ref_file[0])) if ref_file else None
    so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    if not os.path.exists(so):
        subprocess.check_call([os.path.join(ROOT, "tests", "hostsim", "build.sh")])
    sim = ctypes.CDLL(so)
    legs = {}
    if workload in ("evm", "super"):
        from oracle import evm_oracle, wire
        from tests.evm_cases import hostsim_status

        ev = wire_h if workload == "evm" else env["parts"]["evm"]
        sample = min(int(ev["steps"].shape[0]) - 1, 1 << 15)
        W = evm_oracle.EvmWitness(wire.rowmajor_to_rows(ev["steps"][: sample + 1]), wire.rowmajor_to_rows(ev["rw"]),
                                  ev["rw_flags"], wire.rowmajor_to_rows(ev["bytecode"]))
        tc = time.perf_counter()
        st = evm_oracle.verify_steps(W)
        tc = time.perf_counter() - tc
        assert not any(st)
        legs["port"] = {"value": sample / tc, "unit": "rows/s", "cores": 1,
                        "sample": f"first {sample} step pairs of the same trace, pure-Python oracle with dict-indexed lookups (oracle/evm_oracle.py)"}
        hs = min(int(ev["steps"].shape[0]) - 1, 1 << 18)
        sub = {k: v for k, v in ev.items() if k != "meta"}
        sub["steps"] = ev["steps"][: hs + 1]
        tc = time.perf_counter()
        st = hostsim_status(sim, sub)
        tc = time.perf_counter() - tc
        assert not any(st)
        legs["hostsim"] = {"value": hs / tc, "unit": "rows/s", "cores": 1,
                           "sample": f"{hs} step pairs, the kernels' own device functions compiled for the host (tests/hostsim, g++ -O2), "
                                     "index build included: the optimised-CPU line"}
        if ref and "evm" in ref:
            e = ref["evm"]
            legs["reference"] = {"value": e["extrapolated_2p18"]["pairs_per_s"], "unit": "rows/s", "cores": 1,
                                 "measured": [{"step_pairs": p["step_pairs"], "table_rows": p["rw_rows"] + p["bytecode_rows"],
                                               "rows_per_s": p["pairs_per_s"]} for p in e["measured"]],
                                 "fit": e["fit"], "extrapolated": True,
                                 "sample": "verify_steps of the unmodified reference on 2^4 / 2^6 / 2^8-pair prefixes of this trace, build container "
                                           f"({os.path.basename(ref_file[0])}); the 2^18 figure is EXTRAPOLATED from the fit (linear-scan lookups, table.py:864-884)"}
    else:
        from oracle import state_oracle, wire

        cols, flags, mpt = env["cols"], env["flags"], env["mpt"]
        sample = min(units, 1 << 16)
        rows_i = wire.colmajor_to_rows(cols[:, :sample])
        mpt_i = wire.rowmajor_to_rows(mpt)
        tc = time.perf_counter()
        state_oracle.verify_rows(rows_i, flags[:sample], mpt_i)
        tc = time.perf_counter() - tc
        legs["port"] = {"value": sample / tc, "unit": "rows/s", "cores": 1,
                        "sample": f"first {sample} rows of the same witness, pure-Python oracle (oracle/state_oracle.py)"}
        vp = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
        c, f, m = np.ascontiguousarray(cols), np.ascontiguousarray(flags), np.ascontiguousarray(mpt)
        st = np.zeros(c.shape[1], dtype=np.uint32)
        tc = time.perf_counter()
        sim.sim_state_verify(vp(c), vp(f), ctypes.c_uint64(c.shape[1]), vp(m), ctypes.c_uint64(m.shape[0]), vp(st))
        tc = time.perf_counter() - tc
        assert not st.any()
        legs["hostsim"] = {"value": c.shape[1] / tc, "unit": "rows/s", "cores": 1,
                           "sample": f"all {c.shape[1]} rows, the kernel's own device functions compiled for the host (tests/hostsim, g++ -O2)"}
        if ref and "state" in ref:
            legs["reference"] = {"value": ref["state"]["rows_per_s"], "unit": "rows/s", "cores": 1, "extrapolated": False,
                                 "sample": f"check_state_row of the unmodified reference over all {ref['state']['rows']} rows of this witness, build container "
                                           f"({os.path.basename(ref_file[0])})"}
    head = legs.get("reference") or legs["port"]
    return {"value": head["value"], "unit": "rows/s", "cores": 1, "cores_total": cores_total,
            "kind": "reference" if "reference" in legs else "port",
            "sample": head["sample"], "legs": legs,
            "reference_host": None if not ref else ref.get("host")}


if __name__ == "__main__":
    main()
