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
