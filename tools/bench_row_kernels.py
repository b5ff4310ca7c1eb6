#!/usr/bin/env python3
"""Throughput of the row-stencil kernels (Bytecode / Exp / Tx-Sig circuits) on synthetic witnesses:
rows/s and algorithmic GB/s (SURVEY.md §8d bytes per unit), device-resident inputs, HIP-event kernel
time.  bench.py carries the headline EVM / State workloads; this is the side table in DESIGN.md §3."""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth import synth_bytecode_witness, synth_exp_witness, synth_state_ops, synth_tx_witness

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
_lib.init(0)
out = {}


def run(name, sess, units, bytes_per_unit):
    with sess as s:
        for _ in range(3):
            s.launch()
        s.collect()
        for _ in range(20):
            s.launch()
        r = s.collect()
        assert r.ok, (name, r)
    out[name] = {"units": units, "kernel_ms": round(r.kernel_ms, 4), "units_per_s": round(units / r.kernel_ms * 1e3),
                 "algorithmic_GBps": round(units * bytes_per_unit / r.kernel_ms / 1e6, 1),
                 "frac_of_8TBps": round(units * bytes_per_unit / r.kernel_ms / 1e6 / 8000, 4)}
    print(name, out[name], flush=True)


rng = random.Random(1)
r = rng.randrange(P)
k = int(os.environ.get("LOGN", "20"))
codes = [bytes(rng.getrandbits(8) for _ in range(24000)) for _ in range((1 << k) // 24576)]
cols, keccak = synth_bytecode_witness(codes, k, r)
run("bytecode", engine.open_bytecode(cols, keccak, r), 1 << k, 12 * 32)
cols = synth_exp_witness(1 << k, seed=8)
run("exp", engine.open_exp(cols), 1 << k, 21 * 32)
n_tx = 1 << min(k, 17)
w = synth_tx_witness(n_tx, r, seed=4)
run("tx_sign", engine.open_sign(w, r, False), n_tx,
    8 * 32 + 288 + 2 * 5 * 32)
# Copy circuit: events expanded on the device (zk_copy_assign), then evaluated from the same HBM buffers (zk_copy_open)
from zkevm_specs_amd.synth import synth_copy_events
for kk in sorted({15, min(k, 19)}):
    ce = synth_copy_events(1 << kk, seed=6)
    ev, fl, da, of = to_dev(ce["events"]), to_dev(ce["flags"]), torch.from_numpy(ce["data"].view(np.int16)).cuda(), to_dev(ce["offsets"])
    n_rows, n_table, n_rw = engine.copy_assign_sizes(ce["events"], ce["flags"], ce["data"], ce["offsets"])
    c_rows = torch.empty((20, n_rows, 4), dtype=torch.int64, device="cuda")
    c_rf = torch.empty(n_rows, dtype=torch.int32, device="cuda")
    c_rw = torch.empty((n_rw, 14, 4), dtype=torch.int64, device="cuda")
    c_rwf = torch.empty(n_rw, dtype=torch.int32, device="cuda")
    # per output row: 20 circuit cells + the RW row of every second row, written; the events and bytes read are negligible
    run(f"copy_assign_2p{kk}", engine.open_copy_assign(ev, fl, da, of, ce["r"], c_rows, c_rf, None, c_rw, c_rwf), n_rows, 20 * 32 + 4 + 7 * 32)
    run(f"copy_rows_2p{kk}", engine.open_copy(c_rows, c_rf, ce["r"], c_rw, c_rwf, to_dev(ce["bytecode"]), to_dev(ce["tx"]), to_dev(ce["tx_flags"])),
        n_rows, 20 * 32 + 14 * 32)
# keccak table generation (integer-ALU bound: ~3.6 k VALU ops per 136-byte block + ~25 per RLC byte)
n_keys = 1 << k
nrng = np.random.default_rng(6)
keys = torch.from_numpy(nrng.integers(0, 256, size=n_keys * 64, dtype=np.uint8)).cuda()
offs = torch.arange(0, (n_keys + 1) * 64, 64, dtype=torch.int64, device="cuda")
rows_dev = torch.zeros((n_keys, 5, 4), dtype=torch.int64, device="cuda")
run("keccak_table_64B", engine.open_keccak(keys, offs, r, engine.KECCAK_MODE_TABLE, rows_dev=rows_dev), n_keys, 64 + 160)
n_codes = 1 << max(k - 8, 4)
code_bytes = torch.from_numpy(nrng.integers(0, 256, size=n_codes * 24576, dtype=np.uint8)).cuda()
offs = torch.arange(0, (n_codes + 1) * 24576, 24576, dtype=torch.int64, device="cuda")
rows_dev = torch.zeros((n_codes, 5, 4), dtype=torch.int64, device="cuda")
run("keccak_table_24KiB", engine.open_keccak(code_bytes, offs, r, engine.KECCAK_MODE_CIRCUIT, rows_dev=rows_dev), n_codes, 24576 + 160)
# State witness assignment (HBM-bound: 12 slots read + 57 cells written per op; the mock-MPT index, scans and MPT rows ride along)
for kk in sorted({16, k}):
    ops, oflags, *_ = synth_state_ops(1 << kk, seed=2)
    d_ops, d_oflags = to_dev(ops), to_dev(oflags)
    d_rows = torch.empty((57, 1 << kk, 4), dtype=torch.int64, device="cuda")
    d_rflags = torch.empty(1 << kk, dtype=torch.int32, device="cuda")
    d_mpt = torch.empty((1 << kk, 12, 4), dtype=torch.int64, device="cuda")
    run(f"state_assign_2p{kk}", engine.open_state_assign(d_ops, d_oflags, d_rows, d_rflags, d_mpt), 1 << kk, (12 + 57) * 32 + 8)
# Bytecode witness assignment: the unrolled bytecode table of 2^17 / 24,576-byte contracts -> the circuit's 2^17 rows (HBM: 6 cells read, 12 written)
from zkevm_specs_amd.wire import rows_to_rowmajor  # noqa: E402


def unrolled_bytecode_table(codes_):
    """BytecodeTableRows of Bytecode.table_assignments() in wire form (Header row, then a Byte row per byte with is_code)"""
    rows_, offs_, lens_ = [], [0], []
    for ci, code in enumerate(codes_):
        lo, hi = 0x1000 + ci, 0x77
        rows_.append([lo, hi, 1, 0, 0, len(code)])
        left = 0
        for idx, b in enumerate(code):
            is_code = left == 0
            rows_.append([lo, hi, 2, idx, int(is_code), b])
            left = (b - 0x5F if 0x60 <= b <= 0x7F else 0) if is_code else left - 1
        offs_.append(len(rows_))
        lens_.append(len(code))
    return rows_to_rowmajor(rows_, 6), np.array(offs_, dtype=np.uint64), np.array(lens_, dtype=np.uint64)


kb = 17
b_codes = [bytes(rng.getrandbits(8) for _ in range(24000)) for _ in range(((1 << kb) // 24576))]
ub_rows, ub_off, ub_len = unrolled_bytecode_table(b_codes)
d_bc = torch.empty((12, 1 << kb, 4), dtype=torch.int64, device="cuda")
run(f"bytecode_assign_2p{kb}", engine.open_bytecode_assign(to_dev(ub_rows), to_dev(ub_off), to_dev(ub_len), kb, r, rows_dev=d_bc), 1 << kb, (6 + 12) * 32)
# RW table -> State witness in one session (re-keying + radix sort + assignment): the 2^18-step block trace's RW table
from zkevm_specs_amd.synth_block import synth_block_trace  # noqa: E402
wb = synth_block_trace(1 << 18, seed=5)
n_rw = int(wb["rw"].shape[0])
f_rows, f_fl = torch.empty(57 * 4 * (n_rw + 1), dtype=torch.int64, device="cuda"), torch.empty(n_rw + 1, dtype=torch.int32, device="cuda")
f_mpt = torch.empty(48 * (n_rw + 1), dtype=torch.int64, device="cuda")
d_rw, d_rwf = to_dev(wb["rw"]), to_dev(wb["rw_flags"])
run("state_assign_from_rw_2p18", engine.open_state_assign_from_rw(d_rw, d_rwf, f_rows, f_fl, f_mpt), n_rw, (14 + 57) * 32 + 8)
d_ops_b, d_of_b = torch.empty(48 * (n_rw + 1), dtype=torch.int64, device="cuda"), torch.empty(n_rw + 1, dtype=torch.int32, device="cuda")
run("state_ops_from_rw_2p18", engine.open_state_ops_from_rw(d_rw, d_rwf, d_ops_b, d_of_b), n_rw, (14 + 12) * 32 + 8)
# secp256k1 ECDSA verification (integer-ALU bound: ~8.6 k 256-bit Montgomery products per signature)
from zkevm_specs_amd.synth import synth_signatures
n_sig = 1 << 14
sigs = synth_signatures(n_sig, 9)
packed = np.frombuffer(b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") + z.to_bytes(32, "big") + rr.to_bytes(32, "little") +
                                ss.to_bytes(32, "little") for x, y, z, rr, ss in sigs), dtype=np.uint8).reshape(n_sig, 5, 32).copy()
for reps in (1, 2, 4, 8):
    d = torch.from_numpy(np.tile(packed, (reps, 1, 1))).cuda()
    run(f"ecdsa_verify_{n_sig * reps}", engine.open_ecdsa(d), n_sig * reps, 160 + 4)
print(json.dumps(out))
