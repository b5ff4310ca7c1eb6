#!/usr/bin/env python3
"""bench.py — constraint-rows/s of the hot path on N MI355X GPUs (one process per GPU).

EVM workload (the default, BASELINE configs[2]): a "step" is ONE one-shot verification of a witness the device has not seen —
the C entry `zk_evm_verify` (session open: lookup indices, packed key records, density verdict, bytecode directory, the
counting sort; one evaluation pass; tally collect; close) over inputs already resident in HBM.  The witness is resident in
THREE copies (2.1 GB, 8x the 256 MiB Infinity Cache) and the steps rotate over them, so every step reads a witness whose
lines were evicted since it was last touched — no flush kernel sits in the timed region.  `value` = rows / wall time of K such
steps (each step ends in the host synchronisation a verifier needs for its verdict).  `roofline.achieved` =
§8d's algorithmic bytes / the DEVICE span of one step (first kernel of the open to the end of the last evaluation kernel, HIP
events riding on the dispatches): every byte is read by the kernels inside that span, so the fraction cannot exceed 1.
The round-1..3 definition — K passes of an OPEN session over packed records built once — is still measured and reported as a
side value (`roofline.resident_*`, `resident_session`); `--session-pass` makes it the headline again (not §8d-conforming).

Other workloads (`--workload state|tx|super`): a "step" is one evaluation pass over the rank's resident witness (State / Tx /
Sig sessions derive nothing from the witness that a pass then reads instead of it, except the State circuit's MPT index).

`--scaling weak` (default): every rank holds its own 2^log_rows-row witness.  `--scaling strong`: ONE global
witness — rank 0 builds it, its arrays are replicated with broadcasts (RCCL), every circuit's rows are cut into
contiguous ranges with that circuit's halo (zkevm_specs_amd/distributed.py), lookup tables stay whole on every rank.  The
only collective in the result path is the tally exchange (one all-gather of three words per rank).

`python bench.py --gpus N` with no torchrun environment starts its N ranks itself (re-exec under torch.distributed.run on
127.0.0.1); under torchrun (RANK / WORLD_SIZE set) it is one rank.

The JSON line carries, next to the contract's fields (flat scalars inside `roofline` / `config` are the digest of everything
below them: the driver's record keeps scalars of those two objects):
  roofline        EVM: see above; `traffic` = HBM-side bytes per step from the committed rocprofv3 counter passes of this same
                  command (profiles/r*_evm_2p18_profile.json, tools/profile_bench.sh; FETCH_SIZE corrected as the guide
                  prescribes), null when no counter pass of this command is committed.  `resident_*`: the open-session pass.
                  `state_2p16_*`, `state_2p20_*`, `tx_2p14_*`, `super_2p20_*`: the other BASELINE configurations' kernels.
  resident_session  the open-session pass in full (physical traffic / VALU-issue blocks of the hot kernel)
  fresh_witness   open + pass split with explicit cache flushes (the round-2/3 definition of the same thing, for comparison)
  cpu_baseline    `port` (oracle/, pure Python, dict-indexed lookups), `cpu_backend_1core` / `cpu_backend_allcores` (libzkevm_cpu.so:
                  the kernels' own sources built for the host behind the same C ABI, OpenMP: the "optimised CPU" line) — timed
                  here on a bounded sample; the headline `value` is the `port` leg's — and `reference_build_container`: the
                  unmodified reference timed in the build container (tools/time_reference.py -> profiles/r*_cpu_reference.json;
                  the reference is Python and does not travel to the GPU box, so this leg is a CROSS-BOX figure and says so).
  host_path       marshalling (Python objects -> wire arrays, flatten.py) and H2D staging, reported separately.
  other_configs   (default N = 1 run only) BASELINE configs[0], [1], [3], [4] measured after the headline on the same clock:
                  Bytecode 256 B (CPU backend: configs[0] is the CPU-runnable case), State 2^16 (+ cold-cache leg) and 2^20,
                  Tx + Sig 2^14, Super 2^20 — ms / pass, dominant kernel, physical and algorithmic fractions, CPU legs.
"""
import argparse
import ctypes
import glob
import json
import os
import socket
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_legs as legs  # noqa: E402  (side measurements: CPU baselines, the other configurations, batch / fresh / PMC legs)
import zkevm_specs_amd  # noqa: E402,F401
from zkevm_specs_amd import _lib as _zk_lib  # noqa: E402

# before torch makes the process's first HIP call: loading the library sets the runtime's hardware-queue default (one queue per
# concurrent circuit session: with the default 4 the super circuit's six streams share queues and its sessions run back to back)
_zk_lib.load()

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (≈ 6.3 TB/s achievable)
N_SIMD = 1024                # 256 CUs x 4 SIMDs
SHADER_CLOCK_HZ = 2.4e9      # nominal; profiled passes run lower (guide: 1.9 - 2.3 GHz effective)
FETCH_GATHER_CORRECTION = 1.0 / 0.95  # profiles/r01_fetch_size_calibration.txt: per-lane 416 / 448-byte record gathers
FETCH_STREAM_CORRECTION = 2.0         # guide: wide coalesced streaming reads report half the bytes
DEFAULT_LOG_ROWS = {"evm": 18, "state": 16, "super": 20, "tx": 14}
TX_UNIT_BYTES = 8 * 32 + 288 + 2 * 5 * 32   # 8 cells + 288 B of byte rows + 2 tx-table rows
SIG_UNIT_BYTES = 8 * 32 + 288


def load_profile(workload, log_rows):
    """committed rocprofv3 summary of this workload / size (tools/profile_bench.sh), newest round first"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_2p{log_rows}_profile.json")), reverse=True)
    return (json.load(open(files[0])), "profiles/" + os.path.basename(files[0])) if files else (None, None)


def kernel_counters(profile, needle):
    if not profile or not needle:
        return None
    for name, k in profile["kernels"].items():
        if all(n in name for n in needle):
            return dict(k, name=name)
    return None


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` outside torchrun: become `python -m torch.distributed.run ... bench.py <same args>` (one rank
    per GPU, rendezvous on 127.0.0.1 — the container hostname may not resolve).  Rank 0's JSON line goes to the same stdout."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class Ctx:
    """per-process plumbing: rank, device, process group, upload / replicate helpers"""

    def __init__(self, args):
        import numpy as np
        import torch

        self.np, self.torch, self.args = np, torch, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        assert self.world == args.gpus, f"WORLD_SIZE={self.world} but --gpus {args.gpus}"
        # test hooks (single-GPU dry run of the N > 1 path): ZK_BENCH_DEVICE pins every rank to one GPU, ZK_BENCH_BACKEND=gloo
        # replaces RCCL, which refuses two ranks on one device
        if os.environ.get("ZK_BENCH_DEVICE") is not None:
            self.local_rank = int(os.environ["ZK_BENCH_DEVICE"])
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            self.dist = dist
            backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)
        self.h2d = {"bytes": 0, "seconds": 0.0}

    def to_dev(self, x):
        np, torch = self.np, self.torch
        if x.dtype == np.uint8:
            return torch.from_numpy(x).cuda()
        if x.dtype.itemsize == 2:
            return torch.from_numpy(x.view(np.int16)).cuda()
        return torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()

    def upload(self, arrays):
        """numpy dict -> device tensors, timed (pageable host memory, the path a ctypes caller takes)"""
        self.torch.cuda.synchronize()
        t = time.perf_counter()
        out = {k: self.to_dev(v) for k, v in arrays.items()}
        self.torch.cuda.synchronize()
        self.h2d["seconds"] += time.perf_counter() - t
        self.h2d["bytes"] += sum(int(v.nbytes) for v in arrays.values())
        return out

    def replicate(self, arrays, src=0):
        """strong scaling: rank `src`'s arrays on every rank (shapes first, then one broadcast per array: RCCL over xGMI)"""
        torch, dist = self.torch, self.dist
        names = sorted(arrays) if self.rank == src else None
        meta = [[(k, tuple(arrays[k].shape), str(arrays[k].dtype)) for k in names]] if self.rank == src else [None]
        dist.broadcast_object_list(meta, src=src)
        out = {}
        for k, shape, dt in meta[0]:
            if self.rank == src:
                t = self.to_dev(arrays[k])
            else:
                tdt = torch.uint8 if dt == "uint8" else (torch.int16 if dt in ("uint16", "int16") else (torch.int64 if dt == "uint64" else torch.int32))
                t = torch.empty(shape, dtype=tdt, device="cuda")
            if t.numel():
                dist.broadcast(t, src=src)
            out[k] = t
        return out

    def replicate_tree(self, obj, src=0, big=1 << 16):
        """a nested dict / tuple / list of numpy arrays and plain values from rank `src` on every rank: arrays of `big` bytes
        or more travel as device broadcasts (and stay device tensors), the rest as one pickled object (host values)"""
        np = self.np
        flat = {}

        def strip(x, path):
            if isinstance(x, np.ndarray) and x.nbytes >= big:
                flat[path] = x
                return ("__dev__", path)
            if isinstance(x, dict):
                return {k: strip(v, f"{path}/{k}") for k, v in x.items()}
            if isinstance(x, (tuple, list)):
                return type(x)(strip(v, f"{path}/{i}") for i, v in enumerate(x))
            return x

        box = [strip(obj, "") if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        dev = self.replicate(flat if self.rank == src else {}, src=src)

        def fill(x):
            if isinstance(x, tuple) and len(x) == 2 and x[0] == "__dev__":
                return dev[x[1]]
            if isinstance(x, dict):
                return {k: fill(v) for k, v in x.items()}
            if isinstance(x, (tuple, list)):
                return type(x)(fill(v) for v in x)
            return x

        return fill(box[0])

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


class Workload:
    """what the timed loop and the report need to know about one configuration"""
    sess = None
    fresh = None          # open_fn of the fresh-witness leg (None: no such leg)
    oneshot = None        # callable running the one-shot C entry over the resident witness -> Result
    shots = None          # EVM: prepared one-shot calls (engine.EvmOneShot) over the resident copies of the witness
    units = total_units = row_offset = 0
    algo_bytes = None
    kernel_name = kernel_needle = None
    workload = ""
    extra_cfg = None
    wire_h = None         # host wire of the EVM trace (CPU legs / marshalling sample)
    env = None            # host arrays the CPU legs read
    profile_key = None


def build_evm(ctx, log_rows, strong):
    from zkevm_specs_amd import distributed, engine
    from zkevm_specs_amd.synth_evm import synth_evm_trace

    n, w = 1 << log_rows, Workload()
    if strong:
        w.wire_h = synth_evm_trace(n, seed=3) if ctx.rank == 0 else None
        meta_box = [w.wire_h.pop("meta") if ctx.rank == 0 else None]
        ctx.dist.broadcast_object_list(meta_box, src=0)
        meta = meta_box[0]
        full = ctx.replicate(w.wire_h if ctx.rank == 0 else {})
        lo, hi = distributed.shard_bounds(n - 1, ctx.rank, ctx.world)
        wire_d = dict(full, steps=full["steps"][lo: hi + 1].contiguous())
        w.units, w.row_offset, w.total_units = hi - lo, lo, n - 1
        w.algo_bytes = meta["algorithmic_bytes"] * w.units / (n - 1)
    else:
        w.wire_h = synth_evm_trace(n, seed=3 + ctx.rank)
        meta = w.wire_h.pop("meta")
        wire_d = ctx.upload(w.wire_h)
        w.units, w.row_offset = n - 1, ctx.rank * (n - 1)
        w.total_units = w.units * ctx.world
        w.algo_bytes = meta["algorithmic_bytes"]
    w.fresh = lambda: engine.open_evm(wire_d, device=ctx.local_rank)  # noqa: E731
    w.oneshot = lambda status=None: engine.evm_verify(wire_d, status_dev=status, device=ctx.local_rank)  # noqa: E731
    # the witness resident N_COPIES times (device-side clones): the one-shot steps rotate over them
    copies = [wire_d] + [{k: v.clone() for k, v in wire_d.items()} for _ in range(ctx.args.witness_copies - 1)]
    w.shots = [engine.EvmOneShot(c, device=ctx.local_rank) for c in copies]
    w.copies = copies
    w.witness_bytes = sum(int(v.numel()) * v.element_size() for v in wire_d.values())
    w.sess = None if ctx.args.no_session_leg and not ctx.args.session_pass else w.fresh()
    w.kernel_name, w.kernel_needle = "evm_steps_kernel", ("evm_steps_kernel", "-1")
    w.workload = (f"EVM circuit, 2^{log_rows} execution steps {'in total' if strong else 'per GPU'}, mixed-opcode synthetic trace "
                  f"(BASELINE configs[2]); RW table {meta['n_rw']} rows, bytecode table {meta['n_bytecode']} rows")
    w.extra_cfg = {"steps_per_gpu": w.units + 1, "rw_rows": meta["n_rw"], "bytecode_rows": meta["n_bytecode"]}
    w.profile_key = ("evm", log_rows)
    return w


def build_state(ctx, log_rows, strong):
    from zkevm_specs_amd import distributed, engine
    from zkevm_specs_amd.synth import synth_state_witness

    n, w = 1 << log_rows, Workload()
    if strong:
        host = None
        if ctx.rank == 0:
            cols, flags, mpt = synth_state_witness(n, seed=2)
            host = {"cols": cols, "flags": flags, "mpt": mpt}
            w.env = host
        full = ctx.replicate(host if ctx.rank == 0 else {})
        d_cols, d_flags, elo, ehi, lo = distributed.shard_rows(full["cols"], full["flags"], ctx.rank, ctx.world, "state")
        d_mpt = full["mpt"]
        w.units, w.row_offset, w.total_units = ehi - elo, lo, n

        def open_fn():
            s = engine.open_state(d_cols, d_flags, d_mpt, device=ctx.local_rank)
            s.set_range(elo, ehi)
            return s
    else:
        cols, flags, mpt = synth_state_witness(n, seed=2 + ctx.rank)
        w.env = {"cols": cols, "flags": flags, "mpt": mpt}
        d = ctx.upload(w.env)
        d_cols, d_flags, d_mpt = d["cols"], d["flags"], d["mpt"]
        w.units, w.row_offset = n, ctx.rank * n
        w.total_units = w.units * ctx.world
        open_fn = lambda: engine.open_state(d_cols, d_flags, d_mpt, device=ctx.local_rank)  # noqa: E731
    w.sess = open_fn()
    w.fresh = open_fn
    w.algo_bytes = w.units * 57 * 32  # every witness cell counted once
    w.kernel_name, w.kernel_needle = "state_rows_dma_kernel", ("state_rows",)
    w.workload = f"State circuit, 2^{log_rows} RW rows {'in total' if strong else 'per GPU'} (BASELINE configs[1])"
    w.extra_cfg = {"rows_per_gpu": w.units, "mpt_rows": int(d_mpt.shape[0])}
    w.profile_key = ("state", log_rows)
    return w


class TxSigPass:
    """BASELINE configs[3]: one pass = the Tx circuit (tx_circuit.py:253-291) AND the Sig circuit (sig_circuit.py:113-122) over
    the same 2^k signed transactions, each circuit verifying its own chips' signatures as the reference does: secp256k1 ECDSA
    verification (fills the units' ecdsa_status column in HBM) + the SignVerify / Row.verify kernel.  The signatures of both
    circuits go through ONE ECDSA launch; the two row kernels then run on two streams."""

    def __init__(self, ctx, d_tx, d_sig, r):
        from zkevm_specs_amd import engine

        torch, dev = ctx.torch, ctx.local_rank
        self.n = int(d_tx["bytes"].shape[0])
        # ONE ECDSA launch for the chips of both circuits (zk_ecdsa_open_batches: 2 x 2^14 signatures in one dispatch take 1.4 ms,
        # two concurrent launches 1.9 ms each), each circuit's verdict column filled in its own units
        tx_b = dict(sig_bytes=d_tx["bytes"], layout=engine.ECDSA_LAYOUT_TX_UNITS, out_dev=d_tx["meta"], out_stride=4)
        if d_sig is not None:
            # the chips' v: column 3 of the units' meta (a compact copy: the engine reads v[i * v_stride])
            sig_b = dict(sig_bytes=d_sig["bytes"], v=d_sig["meta"][:, 3].contiguous(), layout=engine.ECDSA_LAYOUT_SIG_UNITS, out_dev=d_sig["meta"],
                         out_stride=4, v_stride=1)
            self.ecdsa = engine.open_ecdsa_batches([tx_b, sig_b], device=dev)
        else:
            self.ecdsa = engine.open_ecdsa_batches([tx_b], device=dev)
        self.tx = engine.open_sign(d_tx, r, False, device=dev)
        self.sig = engine.open_sign(d_sig, r, True, device=dev) if d_sig is not None else None
        torch.cuda.synchronize()
        # the two row kernels are independent: the Sig circuit's runs on a side stream that waits for the ECDSA launch
        self._side = torch.cuda.Stream() if d_sig is not None else None
        self._main = torch.cuda.current_stream()
        self._ev = torch.cuda.Event() if d_sig is not None else None
        self._ev_side = None
        self.torch = torch
        if self._side is not None:
            self.sig.set_stream(self._side)

    def launch(self):
        if self.sig is not None and self._ev_side is not None:
            self._main.wait_event(self._ev_side)  # the previous pass's Sig kernel has read its units' verdict column
        self.ecdsa.launch()
        if self.sig is not None:
            self._ev.record(self._main)
            self._side.wait_event(self._ev)
            self.sig.launch()
            self._ev_side = self._ev_side or self.torch.cuda.Event()
            self._ev_side.record(self._side)
        self.tx.launch()

    def collect(self):
        re_, rs_ = self.ecdsa.collect(), self.tx.collect()
        rs_.ecdsa_ms = re_.kernel_ms  # a signature that does not verify fails its unit in the row kernel already
        rs_.sig_ms = rs_.sig_ecdsa_ms = None
        if self.sig is not None:
            r2 = self.sig.collect()
            rs_.sig_ms, rs_.sig_ecdsa_ms = r2.kernel_ms, re_.kernel_ms
            if not r2.ok and rs_.ok:
                rs_.first_fail_row, rs_.first_fail_code = r2.first_fail_row, r2.first_fail_code
            rs_.fail_count += r2.fail_count
        return rs_

    def close(self):
        for s in (self.ecdsa, self.tx, self.sig):
            if s is not None:
                s.close()


def build_tx(ctx, log_rows, strong):
    from zkevm_specs_amd import distributed
    from zkevm_specs_amd.synth import device_keccak_digests, synth_sig_witness, synth_tx_witness

    n, w = 1 << log_rows, Workload()
    seed = 4 if strong else 4 + ctx.rank
    r_tx = 0x1F2E3D4C5B6A79881726354433221100FFEEDDCCBBAA99887766554433221 + (0 if strong else ctx.rank)
    if strong:
        host = None
        if ctx.rank == 0:
            host = {"tx": synth_tx_witness(n, r_tx, seed=seed, signed=True, digests_of=device_keccak_digests(r_tx)),
                    "sig": synth_sig_witness(n, r_tx, seed=seed, digests_of=device_keccak_digests(r_tx))}
            w.env = host
        full = ctx.replicate_tree(host, big=0)
        d_tx, lo = distributed.shard_units(full["tx"], ctx.rank, ctx.world)
        d_sig, _ = distributed.shard_units(full["sig"], ctx.rank, ctx.world)
        w.units, w.row_offset, w.total_units = int(d_tx["bytes"].shape[0]), lo, n
    else:
        w.env = {"tx": synth_tx_witness(n, r_tx, seed=seed, signed=True, digests_of=device_keccak_digests(r_tx)),
                 "sig": synth_sig_witness(n, r_tx, seed=seed, digests_of=device_keccak_digests(r_tx))}
        d_tx, d_sig = ctx.upload(w.env["tx"]), ctx.upload(w.env["sig"])
        w.units, w.row_offset = n, ctx.rank * n
        w.total_units = w.units * ctx.world
    w.sess = TxSigPass(ctx, d_tx, d_sig, r_tx)
    w.algo_bytes = w.units * TX_UNIT_BYTES
    w.kernel_name, w.kernel_needle = "sign_units_kernel", ("sign_units_kernel",)
    w.workload = (f"Tx + Sig circuits, 2^{log_rows} signed synthetic txs {'in total' if strong else 'per GPU'} (BASELINE configs[3]): per pass, ECDSA "
                  "verification + SignVerify kernel of the Tx circuit and ECDSA verification + Row.verify kernel of the Sig circuit")
    w.extra_cfg = {"txs_per_gpu": w.units, "circuits": ["tx", "sig"], "signatures_verified_per_pass": 2 * w.units}
    w.profile_key = ("tx", log_rows)
    return w


STATE_FUSED = False  # --state-fused: the super workload's State rows evaluated where they are computed, from the block's RW table (no State witness in HBM)
STATE_COMPACT = False  # --state-compact: the super workload's State rows without their limb / byte columns (ZK_OPT_STATE_COMPACT)


def build_super(ctx, log_rows, strong):
    from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block

    w = Workload()
    # ONE consistent witness: the State rows are the EVM trace's RW table (synth_block.py), Copy / Exp circuits included
    if strong:
        parts = synth_super_block(log_rows, seed=5) if ctx.rank == 0 else None
        if ctx.rank == 0:
            w.env = {"parts": parts}
        parts = ctx.replicate_tree(parts, big=1 << 20)
        w.sess = SuperCircuit(parts, device=ctx.local_rank, to_device=ctx.to_dev, shard=(ctx.rank, ctx.world), state_compact=STATE_COMPACT)
        w.total_units = sum(w.sess.global_rows.values())
    else:
        parts = synth_super_block(log_rows, seed=5 + ctx.rank)
        w.env = {"parts": parts}
        w.sess = SuperCircuit(parts, device=ctx.local_rank, to_device=ctx.to_dev, state_compact=STATE_COMPACT, state_fused=STATE_FUSED)
    sess = w.sess
    w.units = sum(sess.rows.values())
    if not strong:
        w.row_offset, w.total_units = ctx.rank * w.units, w.units * ctx.world
    frac = {k: sess.rows[k] / max(sess.global_rows[k], 1) for k in sess.rows}
    fused = bool(getattr(sess, "state_fused", False))
    # (fused: what the State pass reads is the op's 14-cell RW row, through the sorted order)
    w.super_bytes = {"evm": parts["meta"]["algorithmic_bytes"] * frac["evm"], "state": sess.rows["state"] * (14 if fused else 15 if STATE_COMPACT else 57) * 32,
                     "bytecode": sess.rows["bytecode"] * 12 * 32, "tx": sess.rows["tx"] * TX_UNIT_BYTES,
                     "copy": sess.rows.get("copy", 0) * (20 + 14) * 32, "exp": sess.rows.get("exp", 0) * 21 * 32}
    w.workload = (f"Super circuit, ~2^{log_rows} rows {'in total' if strong else 'per GPU'} over ONE consistent witness (BASELINE configs[4]; State rows = "
                  "the EVM trace's RW table re-keyed and re-sorted; Copy / Exp rows from the trace's own SHA3 / CODECOPY / EXP steps where "
                  "the generator emits them): " + ", ".join(f"{k} {v}" for k, v in (sess.global_rows if strong else sess.rows).items()) + " rows")
    w.extra_cfg = {"rows_per_gpu": dict(sess.rows), "state_assign_ms": sess.assign_ms}
    if strong:
        w.extra_cfg["rows_total"] = dict(sess.global_rows)
    w.profile_key = ("super", log_rows)
    if fused:
        w.workload += ("; STATE ROWS FUSED (zk_state_verify_from_rw_open): no State witness in HBM — the State session evaluates op2row's rows in the "
                       "registers it computes them in, from the block's RW table through the sorted order it keeps after its first pass")
        w.extra_cfg["state_fused"] = 1
        w.profile_key = ("super_fused", log_rows)
    if STATE_COMPACT:
        w.workload += ("; STATE ROWS COMPACT (ZK_OPT_STATE_COMPACT): 15 of the State row's 57 cells are stored — the ten address limbs and 32 key "
                       "bytes are derived from the address / key cells where the checks use them, not assigned and read back")
        w.extra_cfg["state_compact"] = 1
        w.profile_key = ("super_compact", log_rows)
    return w


BUILDERS = {"evm": build_evm, "state": build_state, "tx": build_tx, "super": build_super}


PRE_RAMP_STEPS = 200  # untimed, before the W warm-up steps: lets the GPU clocks reach their sustained state (see timed_oneshots)


def timed_passes(ctx, sess, steps, warmup):
    """W untimed passes, then exactly K passes bracketed by barrier + synchronize on both sides; returns (seconds, last result)"""
    t_ramp, n_ramp = time.perf_counter(), 0
    while time.perf_counter() - t_ramp < 0.03 and n_ramp < 4000:  # ~30 ms of untimed passes (clock ramp), collected in small groups
        sess.launch()
        n_ramp += 1
        if n_ramp % 16 == 0:
            sess.collect()
    sess.collect()
    timed_passes.last_ramp = n_ramp  # reported as `pre_ramp_steps`
    for _ in range(warmup):
        sess.launch()
    sess.collect()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sess.launch()
    res = sess.collect()
    ctx.barrier()
    return time.perf_counter() - t0, res


def timed_oneshots(ctx, w, steps, warmup):
    """W untimed one-shot verifications, then exactly K, bracketed by barrier + synchronize on both sides; the steps rotate
    over the resident witness copies.  Returns (seconds, tally of the last step, mean device spans of the timed steps)."""
    from zkevm_specs_amd import engine

    shots, lib = w.shots, w.shots[0]._lib
    # clock ramp: a fresh box's first milliseconds of GPU work run below the sustained clocks (one default run in round 4 measured
    # 0.24 ms per step as the first command on its box, 0.178 ms on every repeat); ~40 ms of the same work before the W warm-ups
    for i in range(PRE_RAMP_STEPS):
        shots[i % len(shots)]()
    for i in range(warmup):
        shots[i % len(shots)]()
    spans = [0.0, 0.0, 0.0]
    fails = 0
    ctx.barrier()
    t0 = time.perf_counter()
    # the device spans of the K timed steps (HIP events riding on each step's first and last dispatch) are summed inside the library and
    # read ONCE behind the loop (zk_timing_sums): the timed region holds one foreign call per step and nothing of the measurement itself
    sums, cnt = (ctypes.c_double * 3)(), ctypes.c_uint64()
    lib.zk_timing_sums(sums, ctypes.byref(cnt), 1)  # reset (the warm-up steps' spans)
    order = [shots[(warmup + i) % len(shots)] for i in range(steps)]
    for shot in order:
        fails += shot().fail_count
    ctx.barrier()
    dt = time.perf_counter() - t0
    lib.zk_timing_sums(sums, ctypes.byref(cnt), 1)
    assert cnt.value == steps, (cnt.value, steps)
    spans = [sums[0], sums[1], sums[2]]
    res = shots[(warmup + steps - 1) % len(shots)].result()
    res.fail_count = fails
    hp = (ctypes.c_double * 4)()
    lib.zk_last_host_phases(hp)
    w.host_phases_us = {"open": hp[0], "launch": hp[1], "collect": hp[2], "close": hp[3]}  # of the last timed step
    return dt, res, [x / steps for x in spans]


def resolve_super(w, res):
    """the super circuit's collect() -> (tally-like object, per-circuit block); picks the dominant kernel"""
    sess = w.sess
    results, total_fail_local, first_local = res
    per_circuit = {k: {"rows": sess.rows[k], "kernel_ms": r.kernel_ms,
                       "algorithmic_GBps": w.super_bytes[k] / (r.kernel_ms / 1e3) / 1e9} for k, r in results.items()}
    dom = max(results, key=lambda k: results[k].kernel_ms)
    # the EVM figure is a span over two dispatches (hot start -> warm end); when the State kernel — one dispatch — is within a quarter
    # of it, the State kernel is the block's dominant kernel
    if dom == "evm" and results["state"].kernel_ms * 1.25 >= results["evm"].kernel_ms:
        dom = "state"
    w.kernel_name = {"evm": "evm_steps_kernel", "state": "state_rows_dma_kernel", "bytecode": "bytecode_rows_kernel",
                     "tx": "sign_units_kernel", "copy": "copy_rows_kernel", "exp": "exp_rows_kernel"}[dom]
    w.kernel_needle = (w.kernel_name,) + (("-1",) if dom == "evm" else ())
    w.algo_bytes = w.super_bytes[dom]

    class _Tally:
        fail_count = total_fail_local
        # a failing row of circuit c is reported as its row in the GLOBAL block of that circuit (SuperCircuit.collect)
        first_fail_row = None if first_local is None else first_local[1]
        first_fail_code = 0 if first_local is None else first_local[2]
        kernel_ms = results[dom].kernel_ms

    return _Tally, per_circuit


def roofline_block(w, res, world, strong, cold_ms=None):
    kernel_s = res.kernel_ms / 1e3
    algo_gbps = w.algo_bytes / kernel_s / 1e9
    # HBM traffic and SQ counters come from separate rocprofv3 --pmc passes over this same command
    # (tools/profile_bench.sh); the committed per-dispatch summary is attached when there is one for this size
    profile, profile_src = load_profile(*w.profile_key)
    kc = kernel_counters(profile, w.kernel_needle) if world == 1 or not strong else None
    traffic = valu = None
    wait_frac = None
    if kc and "pmc" in kc:
        pmc = kc["pmc"]
        corr = FETCH_GATHER_CORRECTION if w.kernel_name == "evm_steps_kernel" else FETCH_STREAM_CORRECTION
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            traffic = pmc["FETCH_SIZE"]["avg_per_dispatch"] * 1024.0 * corr + pmc["WRITE_SIZE"]["avg_per_dispatch"] * 1024.0
        if "SQ_ACTIVE_INST_VALU" in pmc:
            # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles summed over waves (guide, PMC table): VALU-active shader
            # cycles = 4 x counter; the chip offers N_SIMD x kernel cycles of VALU issue
            act = pmc["SQ_ACTIVE_INST_VALU"]["avg_per_dispatch"] * 4.0
            wave_cycles = pmc.get("SQ_WAVE_CYCLES", {}).get("avg_per_dispatch", 0) * 4.0
            if wave_cycles and "SQ_WAIT_ANY" in pmc:
                wait_frac = pmc["SQ_WAIT_ANY"]["avg_per_dispatch"] * 4.0 / wave_cycles
            valu = {"insts_valu_per_launch": pmc.get("SQ_INSTS_VALU", {}).get("avg_per_dispatch"),
                    "active_valu_cycles_per_launch": act, "wave_cycles_per_launch": wave_cycles,
                    "wait_any_frac_of_wave_cycles": wait_frac,
                    "frac_of_issue_peak": act / (kernel_s * SHADER_CLOCK_HZ * N_SIMD),
                    "note": f"VALU-active cycles / ({N_SIMD} SIMDs x kernel time x {SHADER_CLOCK_HZ / 1e9:.1f} GHz nominal); counters from {profile_src}"}
    physical = traffic / kernel_s / 1e9 if traffic else None
    streaming = w.kernel_name in ("state_rows_dma_kernel", "bytecode_rows_kernel", "exp_rows_kernel")
    return {
        # what binds the dominant kernel; the HBM figures below are the roofline north_star names, reported either way
        "bound": "hbm" if streaming else "latency+valu-issue (not hbm: see binding_resource)",
        "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        # physical HBM-side bytes moved per launch (PMC) over the live kernel time; falls back to the algorithmic figure,
        # labelled, when no counter pass is committed for this configuration
        "achieved": physical if physical is not None else algo_gbps,
        "frac": (physical if physical is not None else algo_gbps) / HBM_PEAK_GBPS,
        "frac_source": "pmc traffic / live kernel time" if physical is not None else "ALGORITHMIC bytes (no counter pass committed for this configuration)",
        "traffic": traffic, "traffic_source": profile_src if traffic else None,
        "algorithmic": {"bytes_per_launch": w.algo_bytes, "GBps": algo_gbps, "frac": algo_gbps / HBM_PEAK_GBPS,
                        "note": "SURVEY.md §8d bytes (every looked-up row at its wire size) / kernel time: work per byte budget, "
                                "not HBM utilisation — lookups read packed key records, so it may exceed what the memory system moves"},
        "binding_resource": ("HBM streaming (one coalesced pass over the witness columns)" if streaming else
                             "VALU issue + dependent-lookup latency (integer-modular path); see `valu`"),
        "valu": valu,
        "kernel": w.kernel_name, "kernel_ms": res.kernel_ms,
        "rocprof_avg_kernel_ms": None if not kc or "trace" not in kc else kc["trace"]["avg_ns"] / 1e6,
        "cold_cache": None if cold_ms is None else {
            "kernel_ms": cold_ms, "algorithmic_GBps": w.algo_bytes / (cold_ms / 1e3) / 1e9,
            "traffic_GBps": None if not traffic else traffic / (cold_ms / 1e3) / 1e9,
            "note": "same kernel, each pass preceded (same stream) by a 2 GiB read-only sweep: cold L2 / Infinity Cache / page-table lines"},
    }, profile, profile_src


def oneshot_profile_numbers(profile):
    """per-STEP figures from the committed rocprofv3 passes of the one-shot command (tools/profile_bench.sh): HBM-side traffic
    (FETCH_SIZE + WRITE_SIZE of every kernel of a step; FETCH_SIZE x2 for the streaming kernels as the guide prescribes, x 1/0.95 —
    calibrated on this access pattern — for the gathering evaluation kernels) and the sum of the kernels' average durations"""
    if not profile or not profile.get("bench_line"):
        return None, None
    n = profile["bench_line"]["steps"] + profile["bench_line"]["warmup"] + (profile["bench_line"].get("pre_ramp_steps") or 0)
    traffic, kernel_ns, seen_pmc = 0.0, 0.0, False
    for name, k in profile["kernels"].items():
        if name.startswith("__amd_rocclr"):
            continue  # the runtime's copy kernels: the witness uploads of that run (outside the steps) and the 128-byte result read-back
        pmc = k.get("pmc", {})
        corr = FETCH_GATHER_CORRECTION if "evm_steps_kernel" in name or "evm_deferred" in name else FETCH_STREAM_CORRECTION
        if "FETCH_SIZE" in pmc:
            traffic += pmc["FETCH_SIZE"]["avg_per_dispatch"] * pmc["FETCH_SIZE"]["dispatches"] * 1024.0 * corr
            seen_pmc = True
        if "WRITE_SIZE" in pmc:
            traffic += pmc["WRITE_SIZE"]["avg_per_dispatch"] * pmc["WRITE_SIZE"]["dispatches"] * 1024.0
        if "trace" in k:
            kernel_ns += k["trace"]["avg_ns"] * k["trace"]["calls"]
    return (traffic / n if seen_pmc else None), (kernel_ns / n / 1e6 if kernel_ns else None)


def oneshot_roofline(w, spans, log_rows, live=None):
    """§8(d): algorithmic bytes of one step / the device span of one step (every kernel that reads them lies inside it)"""
    open_ms, pass_ms, span_ms = spans
    profile, profile_src = load_profile("evm_oneshot", log_rows)
    traffic, rocprof_ms = oneshot_profile_numbers(profile)
    committed = traffic
    if live and live[0]:
        traffic, profile_src_traffic = live
    else:
        profile_src_traffic = (profile_src + (f" ({live[1]})" if live and live[1] else "")) if traffic else None
    achieved = w.algo_bytes / (span_ms / 1e3) / 1e9
    return {
        "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS,
        "frac_source": "SURVEY 8(d) algorithmic bytes per step / device span of one zk_evm_verify (HIP events on its first and last dispatch)",
        "traffic": traffic, "traffic_source": profile_src_traffic, "traffic_committed_profile": committed,
        "traffic_over_algorithmic": None if not traffic else traffic / w.algo_bytes,
        "algorithmic_bytes": w.algo_bytes, "witness_bytes_resident": getattr(w, "witness_bytes", None),
        "kernel": "zk_evm_verify = evm_open_fill + evm_open_phase1 + evm_open_phase2 + evm_steps_kernel<hot> (+ warm / cold)",
        "kernel_ms": span_ms, "open_ms": open_ms, "pass_kernel_ms": pass_ms,
        "rocprof_sum_kernel_ms_per_step": rocprof_ms,
        "host_us_in_open": w.host_phases_us["open"], "host_us_in_launch": w.host_phases_us["launch"],
        "host_us_in_collect": w.host_phases_us["collect"], "host_us_in_close": w.host_phases_us["close"],
    }, profile, profile_src


def digest_other(roof, cfg, other):
    """flat scalars of the other configurations inside `roofline` / `config` (the driver's record keeps those)"""
    for key, b in other.items():
        unit = "txs" if key.startswith("tx") else "rows"
        cfg[f"{key}_{unit}_per_s"] = b["value"]
        cfg[f"{key}_ms_per_step"] = b["ms_per_step"]
        r = b.get("roofline") or {}
        if r.get("kernel"):
            roof[f"{key}_kernel"] = r["kernel"]
            roof[f"{key}_kernel_ms"] = r.get("kernel_ms")
            roof[f"{key}_frac"] = r.get("frac")
            alg = r.get("algorithmic") or {}
            if alg.get("frac") is not None:
                roof[f"{key}_algorithmic_frac"] = alg["frac"]
            cold = r.get("cold_cache") or {}
            if cold.get("kernel_ms"):
                roof[f"{key}_cold_kernel_ms"] = cold["kernel_ms"]
                roof[f"{key}_cold_algorithmic_frac"] = cold["algorithmic_GBps"] / HBM_PEAK_GBPS
        if b.get("backend"):
            cfg[f"{key}_backend"] = b["backend"]


LINE_BUDGET = 3800  # bytes: the driver keeps a bounded tail of stdout; round 4's 25 KB line outgrew it and was recorded as unparsed


def _sig(x, digits=6):
    """floats at `digits` significant digits (the full-precision record is bench_full.json)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{digits}g}")


def compact_line(out, full_path=None):
    """The ONE line the driver parses: the contract's fields, a `config` naming the workload, `roofline` and `cpu_baseline` as flat
    numbers — never prose.  Everything else the run measured (other_configs, the legs' sample texts, resident_session, batch,
    fresh_witness, host_path) is in `out`, which main() writes to bench_full.json.  Keeps the line under LINE_BUDGET bytes by
    construction: fixed key lists, short strings, 6 significant digits; a last-resort trim drops the other configurations' digests."""
    cfg_full, roof_full = out.get("config") or {}, out.get("roofline") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "pre_ramp_steps", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = {"workload": str(cfg_full.get("workload", ""))[:150]}
    for k in ("step", "sharding"):
        if cfg_full.get(k):
            cfg[k] = str(cfg_full[k])[:100]
    others = tuple(out.get("other_configs") or ())
    for k, v in cfg_full.items():
        if k.startswith(others) and others:
            continue  # digest_other's flat copies: the line carries them once, under roofline.other_configs
        if k not in cfg and isinstance(v, (int, float)) and not isinstance(v, bool):
            cfg[k] = v
        elif k in ("rows_per_gpu", "rows_total") and isinstance(v, dict):
            cfg[k] = v
    roof = {}
    for k in ("bound", "kernel"):
        if roof_full.get(k) is not None:
            roof[k] = str(roof_full[k])[:110]
    for k in ("peak", "unit", "achieved", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes", "witness_bytes_resident", "kernel_ms",
              "open_ms", "pass_kernel_ms", "rocprof_sum_kernel_ms_per_step", "rocprof_avg_kernel_ms", "host_us_in_open", "host_us_in_launch",
              "host_us_in_collect", "host_us_in_close", "batch_ms_per_witness", "batch_rows_per_s", "resident_ms_per_pass", "resident_rows_per_s",
              "resident_hot_kernel_ms", "oneshot_ms"):
        if k in roof_full:
            roof[k] = roof_full[k]
    alg = roof_full.get("algorithmic")
    if isinstance(alg, dict):  # the pass workloads: frac = physical traffic; the §8(d) algorithmic figure beside it
        roof["algorithmic_bytes"], roof["algorithmic_frac"] = alg.get("bytes_per_launch"), alg.get("frac")
    cold = roof_full.get("cold_cache")
    if isinstance(cold, dict):
        roof["cold_kernel_ms"] = cold.get("kernel_ms")
    if roof_full.get("traffic_source"):
        roof["traffic_live"] = str(roof_full["traffic_source"]).startswith("measured in this run")
    pc = roof_full.get("per_circuit")
    if isinstance(pc, dict):
        roof["per_circuit_kernel_ms"] = {k: v.get("kernel_ms") for k, v in pc.items()}
    other = {}  # <= 3 scalars per other configuration: throughput, ms per step, fraction of the roof of its dominant kernel
    for key in (out.get("other_configs") or {}):
        unit = "txs" if key.startswith("tx") else "rows"
        b = out["other_configs"][key]
        other[key] = {f"{unit}_per_s": b.get("value"), "ms_per_step": b.get("ms_per_step")}
        r = b.get("roofline") or {}
        if (b.get("block_oneshot") or {}).get("ms") is not None:
            other[key]["oneshot_ms"] = b["block_oneshot"]["ms"]  # the block verified from raw inputs, everything derived on the device (block.py)
        if r.get("frac") is not None:
            other[key]["frac"] = r["frac"]
            # what the fraction is OF (the enclosing object's bound / unit describe the headline only): "hbm-physical" = PMC traffic of the
            # dominant kernel / its time / 8 TB/s, "hbm-algorithmic" = SURVEY 8(d) bytes, "valu" = VALU-active cycles / issue peak
            src, bnd = str(r.get("frac_source", "")), str(r.get("bound", ""))
            other[key]["bound"] = "valu" if bnd.startswith("valu") else ("hbm-physical" if src.startswith("pmc traffic") else "hbm-algorithmic")
    if other:
        roof["other_configs"] = other
    # the headline `frac` is ALGORITHMIC (SURVEY 8(d) bytes / device span); the physical figure beside it: measured HBM-side traffic / span
    roof["frac_bound"] = "hbm-algorithmic"
    if roof_full.get("traffic") and roof_full.get("kernel_ms") and roof_full.get("peak"):
        roof["physical_frac"] = _sig(roof_full["traffic"] / (roof_full["kernel_ms"] * 1e-3) / 1e9 / roof_full["peak"])
    line["config"], line["roofline"] = cfg, roof
    cb = out.get("cpu_baseline")
    if cb:
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = str(cb.get("sample_short") or cb.get("sample") or "")[:120]
        legs = cb.get("legs") or {}
        c["extrapolated"] = False  # the headline leg is measured
        c["measured_on"] = "this box"  # cpu_baseline() only ever heads the line with a leg timed here (bench_legs.py)
        c["this_box_cores_total"] = cb.get("this_box_cores_total")
        if cb.get("sample_pairs") is not None:
            c["sample_pairs"] = cb["sample_pairs"]  # units of the timed sample, as a number (the `sample` text says what they were)
        c["legs"] = {k: {"value": v.get("value"), "cores": v.get("cores")} for k, v in legs.items()}
        line["cpu_baseline"] = c
    if full_path:
        line["full_record"] = full_path

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return _sig(o)

    line = {k: (v if k in ("value", "ms_per_step") else rnd(v)) for k, v in line.items()}  # the two contract scalars at full precision
    for drop in ("other_configs", "per_circuit_kernel_ms"):  # last resort; never reached with the key lists above
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_BUDGET:
            break
        line["roofline"].pop(drop, None)
    return line


def write_full(out):
    """the full record next to the script (and into gpurun_out/ when that exists, so that it comes back from a lease)"""
    name = "bench_full.json"
    if os.environ.get("ZK_BENCH_FULL"):  # a child run of another configuration (tools/bench_legs.py own_process_config): its record goes where the parent reads it
        with open(os.environ["ZK_BENCH_FULL"], "w") as f:
            json.dump(out, f)
        return os.environ["ZK_BENCH_FULL"]
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, name), "w") as f:
                    json.dump(out, f, indent=1)
        except OSError:
            pass
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="evm", choices=["evm", "state", "super", "tx"])
    ap.add_argument("--log-rows", type=int, default=None, help="log2 rows per GPU (weak) or in total (strong); default 18 evm, 16 state, 20 super, 14 tx")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--session-pass", action="store_true", help="EVM: a step = one pass of an OPEN session (the round-1..3 headline; not the SURVEY 8(d) metric)")
    ap.add_argument("--witness-copies", type=int, default=3, help="EVM one-shot steps rotate over this many resident copies of the witness")
    ap.add_argument("--no-session-leg", action="store_true", help="EVM: skip the open-session side measurement")
    ap.add_argument("--no-batch-leg", action="store_true", help="EVM: skip the batch-entry (two witnesses in flight) side measurement")
    ap.add_argument("--tally", default="torch", choices=["torch", "abi"],
                    help="the tally exchange: torch.distributed all-gather (default) or the C ABI's own RCCL communicator (zk_dist_*; checked against the other)")
    ap.add_argument("--no-live-pmc", action="store_true", help="EVM one-shot line: take roofline.traffic from the committed profile instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the child of live_pmc_traffic: one-shot steps only
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--state-fused", action="store_true", help="super: State rows evaluated from the RW table where they are computed (zk_state_verify_from_rw_open), no 57-cell witness")
    ap.add_argument("--state-compact", action="store_true", help="super: State rows without their limb / byte columns (ZK_OPT_STATE_COMPACT)")
    ap.add_argument("--no-oneshot-leg", action="store_true", help="super: skip the block one-shot side measurement (block.BlockVerifier)")
    ap.add_argument("--no-cold-leg", action="store_true", help="skip the cold-cache kernel timing after the timed region")
    ap.add_argument("--no-fresh-leg", action="store_true", help="skip the open / pass split with explicit cache flushes")
    ap.add_argument("--full-line", action="store_true", help="print the full nested record on stdout instead of the compact line (it is always written to bench_full.json)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs[0], [1], [3], [4] after the headline (default: run them at N = 1, evm workload)")
    args = ap.parse_args()
    global STATE_COMPACT
    STATE_COMPACT = bool(args.state_compact)
    global STATE_FUSED
    STATE_FUSED = bool(args.state_fused)
    # what the side legs (tools/bench_legs.py) need from this file: the builders, the timing loop and the roofline blocks
    CORE = types.SimpleNamespace(BUILDERS=BUILDERS, timed_passes=timed_passes, resolve_super=resolve_super, roofline_block=roofline_block,
                                 tx_extras=tx_extras, oneshot_profile_numbers=oneshot_profile_numbers)
    if args.pmc_child:  # counters are collected over the one-shot steps alone
        args.no_session_leg = args.no_batch_leg = args.no_cold_leg = args.no_fresh_leg = args.no_other_configs = args.no_cpu_baseline = True
        args.no_live_pmc = args.no_oneshot_leg = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)

    ctx = Ctx(args)
    np, torch, dist, rank, world = ctx.np, ctx.torch, ctx.dist, ctx.rank, ctx.world
    from zkevm_specs_amd import _lib, distributed

    log_rows = args.log_rows if args.log_rows is not None else DEFAULT_LOG_ROWS[args.workload]
    strong = args.scaling == "strong" and world > 1
    oneshot_mode = args.workload == "evm" and not args.session_pass
    _lib.init(ctx.local_rank)
    # one real (non-default) stream shared by torch and the engine: uploads, the passes and the cold leg's flush kernel are
    # ordered on it (torch's default stream is handle 0, which the engine reads as "use your own stream")
    bench_stream = torch.cuda.Stream()
    torch.cuda.set_stream(bench_stream)
    _lib.check(_lib.load().zk_set_stream(bench_stream.cuda_stream), "zk_set_stream")

    w = BUILDERS[args.workload](ctx, log_rows, strong)
    sess = w.sess
    per_circuit = None
    spans = None
    session_side = None
    batch_side = None
    if oneshot_mode:
        dt, res, spans = timed_oneshots(ctx, w, args.steps, args.warmup)
        if not args.no_batch_leg:
            batch_side = legs.timed_batch(ctx, w, args.steps, args.warmup)
        if sess is not None:
            dt_s, res_s = timed_passes(ctx, sess, args.steps, args.warmup)
            session_side = (dt_s, res_s)
    else:
        dt, res = timed_passes(ctx, sess, args.steps, args.warmup)
        pre_ramp = timed_passes.last_ramp
        if args.workload == "super":
            res, per_circuit = resolve_super(w, res)

    # Cold-cache leg (outside the timed region): the timed passes re-read the same witness, so page-table lines and part of
    # the rows are still in L2 / Infinity Cache from the previous pass.  Here every pass is preceded by a read-only sweep over
    # 2 GiB of unrelated data ON THE SAME STREAM (the flush is ordered before the pass).
    cold_ms = None
    flush = None
    if (not args.no_cold_leg or not args.no_fresh_leg) and args.workload not in ("super", "tx") and sess is not None:
        flush = torch.zeros(1 << 29, dtype=torch.int32, device="cuda")
    if not args.no_cold_leg and flush is not None:
        for _ in range(8):
            flush.sum()
            sess.launch()
        cold_ms = sess.collect().kernel_ms
    fresh_block = None
    if not args.no_fresh_leg and w.fresh is not None and flush is not None:
        fresh_block = legs.fresh_leg(ctx, w, flush)
    del flush

    total_fail, first_row, first_code = distributed.reduce_tally(res.fail_count, res.first_fail_row if res.fail_count else None, res.first_fail_code,
                                                                 0 if args.workload == "super" else w.row_offset, device="cuda")
    if args.tally == "abi":  # the same exchange through zk_dist_* (the engine's own RCCL communicator): must agree
        with distributed.RcclTally(rank, world, device=ctx.local_rank) as rt:
            via_abi = rt.reduce(res if res.fail_count else type("Clean", (), {"fail_count": 0, "first_fail_row": None, "first_fail_code": 0}),
                                0 if args.workload == "super" else w.row_offset)
        assert via_abi == (total_fail, first_row, first_code), (via_abi, total_fail, first_row, first_code)
    t_max = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    assert total_fail == 0 and first_row is None, "synthetic witness must satisfy every constraint"

    if rank == 0:
        rows_total = w.total_units * args.steps
        if oneshot_mode:
            live = None
            if world == 1 and not args.no_live_pmc and not args.pmc_child:
                try:
                    live = legs.live_pmc_traffic(CORE, log_rows)
                except Exception as e:  # noqa: BLE001 — never let the side measurement take the line down
                    live = (None, f"live PMC measurement failed ({type(e).__name__})")
            roofline, profile, profile_src = oneshot_roofline(w, spans, log_rows, live)
        else:
            roofline, profile, profile_src = roofline_block(w, res, world, strong, cold_ms)
        out = {
            "metric": "BN254 constraint-rows/sec",
            "value": rows_total / dt,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            # untimed steps BEFORE the W warm-ups (clock ramp of a fresh box): the untimed work is pre_ramp_steps + warmup steps
            "pre_ramp_steps": PRE_RAMP_STEPS if oneshot_mode else pre_ramp,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u256 (BN254 Fr: 4xu64 canonical cells, u32-limb Montgomery)",
            "data": "synthetic",
            "config": dict({"workload": w.workload, "sharding": f"rows x{world} ({'one global witness, tables broadcast' if strong else 'independent witnesses'}), tally all-gather"},
                           **w.extra_cfg),
            "roofline": roofline,
        }
        if oneshot_mode:
            out["config"]["step"] = "one-shot zk_evm_verify (open + pass + collect + close) of a witness not touched for the previous steps"
            out["config"]["witness_copies"] = len(w.shots)
            if batch_side is not None:
                traffic = roofline.get("traffic")
                batch_side["note"] = ("zk_evm_verify_batch: the same fresh-witness verifications, two in flight on two streams (the HBM-bound open of one under "
                                      "the latency-bound evaluation of the other); every witness is opened, evaluated and collected once")
                batch_side["algorithmic_GBps"] = w.algo_bytes / (batch_side["ms_per_witness"] / 1e3) / 1e9
                batch_side["traffic_GBps"] = None if not traffic else traffic / (batch_side["ms_per_witness"] / 1e3) / 1e9
                out["batch"] = batch_side
                roofline["batch_ms_per_witness"] = batch_side["ms_per_witness"]
                roofline["batch_rows_per_s"] = batch_side["rows_per_s"]
                roofline["batch_traffic_frac"] = None if not traffic else batch_side["traffic_GBps"] / HBM_PEAK_GBPS
            if session_side is not None:
                dt_s, res_s = session_side
                blk, _, _ = roofline_block(w, res_s, world, strong, cold_ms)
                out["resident_session"] = {"ms_per_pass": dt_s / args.steps * 1e3, "rows_per_s": w.units * args.steps / dt_s, "roofline": blk,
                                           "note": "K passes of ONE open session: every pass re-sorts and re-evaluates, but reads the packed step / key records "
                                                   "built once by zk_evm_open — the round-1..3 headline, NOT the SURVEY 8(d) metric"}
                roofline["resident_ms_per_pass"] = dt_s / args.steps * 1e3
                roofline["resident_rows_per_s"] = w.units * args.steps / dt_s
                roofline["resident_hot_kernel_ms"] = res_s.kernel_ms
                roofline["resident_traffic_frac"] = blk["frac"] if blk.get("traffic") else None
        if fresh_block is not None:
            out["fresh_witness"] = fresh_block
        h2d = ctx.h2d
        if h2d["bytes"]:
            out["host_path"] = {"h2d_bytes": h2d["bytes"], "h2d_seconds": h2d["seconds"], "h2d_GBps": h2d["bytes"] / h2d["seconds"] / 1e9,
                                "rows_per_s_including_h2d": w.units / (h2d["seconds"] + dt / args.steps),
                                "note": "pageable host arrays -> HBM (torch .cuda()); never part of `value`"}
        if args.workload == "evm" and w.wire_h is not None and "host_path" in out and not args.no_cpu_baseline:
            out["host_path"]["marshalling"] = legs.marshalling_sample(w.wire_h)
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
            tx_extras(out["roofline"], res, profile, profile_src)
        if not args.no_cpu_baseline and world == 1 and (w.env is not None or w.wire_h is not None):
            out["cpu_baseline"] = legs.cpu_baseline(args.workload, w)
        elif world > 1:
            out["cpu_baseline"] = {"value": None, "unit": "rows/s", "cores": 0, "kind": "port", "sample": "timed at N = 1 only (run without --gpus)"}
    if sess is not None:
        sess.close()
    if rank == 0 and args.workload == "super" and world == 1 and w.env is not None and not args.no_oneshot_leg and not STATE_FUSED:
        # (after the resident sessions are closed: the block's four chains want hardware queues of their own, and the six resident
        # sessions' streams would share them — measured 1.12 ms with them open against 0.94 ms alone)
        try:
            del sess
            sess = None
            torch.cuda.synchronize()
            out["block_oneshot"] = legs.block_oneshot(w.env["parts"], ctx.to_dev, device=ctx.local_rank, state_compact=STATE_COMPACT)
            out["roofline"]["oneshot_ms"] = out["block_oneshot"]["ms"]
        except Exception as e:  # noqa: BLE001 — a side leg: the line says so instead of losing the resident figures
            out["block_oneshot"] = {"ms": None, "error": f"{type(e).__name__}: {e}"[:300]}
    del w, sess
    if rank == 0 and world == 1 and args.workload == "evm" and args.log_rows is None and not args.no_other_configs:
        torch.cuda.empty_cache()
        out["other_configs"] = legs.other_configs(CORE, ctx, args)
        digest_other(out["roofline"], out["config"], out["other_configs"])
    if rank == 0:
        full_path = write_full(out)
        if args.full_line:
            print(json.dumps(out))  # everything on one line (tools that want the nested blocks on stdout); NOT what the driver should parse
        else:
            line = json.dumps(compact_line(out, full_path), separators=(",", ":"))
            assert len(line) <= LINE_BUDGET, len(line)
            print(line, flush=True)
    if world > 1:
        dist.destroy_process_group()


def tx_extras(roof, res, profile, profile_src):
    """The Tx pass is two ecdsa_verify_kernel launches (Tx circuit's and Sig circuit's signatures, two streams) next to two ~40 us
    row kernels: the roofline that binds is VALU issue of the 256-bit modular arithmetic, not HBM.  Top level = that kernel priced
    against the chip's VALU-issue cycles; the SignVerify row kernel's HBM figures stay in `sign_units_kernel`."""
    row = {k: roof[k] for k in ("bound", "peak", "unit", "achieved", "frac", "frac_source", "traffic", "traffic_source", "algorithmic",
                                "binding_resource", "valu", "kernel", "kernel_ms", "rocprof_avg_kernel_ms")}
    row["bound"] = "latency (16k units = a quarter wavefront per SIMD)"
    ke = kernel_counters(profile, ("ecdsa_verify_kernel",))
    kernel_s = res.ecdsa_ms / 1e3
    peak = N_SIMD * SHADER_CLOCK_HZ / 1e9  # G VALU-issue cycles / s the chip offers
    act = insts = waves = None
    if ke and "pmc" in ke and "SQ_ACTIVE_INST_VALU" in ke["pmc"]:
        act = ke["pmc"]["SQ_ACTIVE_INST_VALU"]["avg_per_dispatch"] * 4.0
        insts = ke["pmc"].get("SQ_INSTS_VALU", {}).get("avg_per_dispatch")
        waves = ke["pmc"].get("SQ_WAVES", {}).get("avg_per_dispatch")
    for k in list(roof):
        del roof[k]
    roof.update({
        "bound": "valu-issue (integer ALU: 256-bit modular arithmetic; neither hbm nor mfma applies)",
        "kernel": "ecdsa_verify_kernel", "kernel_ms": res.ecdsa_ms,
        "rocprof_avg_kernel_ms": None if not ke or "trace" not in ke else ke["trace"]["avg_ns"] / 1e6,
        "unit": "G VALU-active cycles/s", "peak": peak,
        "achieved": None if act is None else act / kernel_s / 1e9, "frac": None if act is None else act / kernel_s / 1e9 / peak,
        "frac_source": f"SQ_ACTIVE_INST_VALU x 4 per launch ({profile_src}) / live kernel time / ({N_SIMD} SIMDs x {SHADER_CLOCK_HZ / 1e9:.1f} GHz)",
        "traffic": None, "insts_valu_per_launch": insts, "wavefronts_per_launch": waves,
        "note": "one wavefront per SIMD issues a VALU instruction every ~4 cycles (half the 2-cycle SIMD rate, profiles/r02_valu_issue_rates.txt), "
                "so 0.5 of this peak is what one resident wavefront per SIMD can reach; a pass = two launches (Tx and Sig circuits) on two streams",
        "sig_circuit": {"row_kernel_ms": res.sig_ms, "ecdsa_verify_kernel_ms": res.sig_ecdsa_ms},
        "sign_units_kernel": row,
    })


if __name__ == "__main__":
    main()
