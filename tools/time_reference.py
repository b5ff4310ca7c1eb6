#!/usr/bin/env python3
"""Time the REFERENCE ITSELF on the workloads bench.py measures (build container only: needs /root/reference and the
dependency stand-ins under oracle/refshim; the GPU box has neither, so the result is committed under profiles/ and
bench.py attaches it to `cpu_baseline.reference`).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=oracle/refshim:/root/reference/src python3 tools/time_reference.py \
        > profiles/r02_cpu_reference.json

What is timed, single process, one core (the reference has no parallelism):
  * EVM circuit: `verify_steps(tables, steps)` (evm_circuit/main.py:14) on prefixes of bench.py's own config-3 trace
    (synth_evm_trace(2^18, seed=3)) with N = 2^4, 2^6, 2^8 step pairs; each prefix carries the RW / bytecode rows its steps
    touch (the reference's lookups scan the whole table, table.py:864-884, so the table size is part of the cost).  The
    cost model t = steps * (a + b * table_rows) is fitted by least squares and extrapolated to the full 2^18-step trace
    with its full tables — EXTRAPOLATED, NOT MEASURED (the full run would take months).
  * State circuit: `check_state_row` over all 2^16 rows of bench.py's config-2 witness (state_circuit.py:492).
  * Bytecode circuit: `check_bytecode_row` over config 1 (256-byte contract, k = 9).
  * Exp circuit: `verify_exp_circuit` over 2^12 rows.
  * Tx + Sig circuits (round 3): `tx_circuit.verify_circuit` + `sig_circuit.verify_circuit` over 2^6 / 2^8 / 2^10 signed legacy txs.
ZK_TIME_ONLY=tx,bytecode,... selects sections; ZK_TIME_MERGE=<earlier json> carries the other sections over (marked).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def evm_prefix(w, n_pairs):
    """the first n_pairs step pairs of the trace with only the table rows those steps can look up"""
    steps = w["steps"][: n_pairs + 1]
    rwc_lo = int(steps[0, 1, 0])
    rwc_hi = int(steps[-1, 1, 0])
    rwc = w["rw"][:, 0, 0].astype(np.int64)
    keep = (rwc >= rwc_lo) & (rwc < max(rwc_hi, rwc_lo + 1) + 64)  # + the pair's own lookups past next.rw_counter
    # bytecode rows: the contracts those steps execute (whole contracts: opcode / push-data lookups index into them)
    hashes = {tuple(steps[i, 5:7].reshape(-1).tolist()) for i in range(steps.shape[0])}
    bk = np.array([tuple(r[0:2].reshape(-1).tolist()) in hashes for r in w["bytecode"]])
    out = {k: v for k, v in w.items()}
    out["aux"], out["aux_kind"] = w["aux"][: n_pairs + 1], w["aux_kind"][: n_pairs + 1]
    out["steps"], out["rw"], out["rw_flags"], out["bytecode"] = steps, w["rw"][keep], w["rw_flags"][keep], w["bytecode"][bk]
    return out


def time_evm():
    from oracle.gen_golden_evm import unflatten
    from zkevm_specs.evm_circuit.main import verify_steps
    from zkevm_specs_amd.synth_evm import synth_evm_trace

    from tests.evm_cases import with_defaults

    w = synth_evm_trace(1 << 18, seed=3)
    meta = w.pop("meta")
    w = with_defaults(w)
    pts = []
    for log_n in [int(x) for x in os.environ.get("ZK_TIME_EVM_LOGS", "4,6,8").split(",")]:  # bench.py's live leg uses 4,5 (~30 s)
        n = 1 << log_n
        pw = evm_prefix(w, n)
        tables, steps = unflatten(pw)
        t0 = time.perf_counter()
        verify_steps(tables, steps)  # raises on any failing pair: the trace must be valid for the reference too
        dt = time.perf_counter() - t0
        pts.append({"step_pairs": n, "rw_rows": int(pw["rw"].shape[0]), "bytecode_rows": int(pw["bytecode"].shape[0]),
                    "seconds": dt, "pairs_per_s": n / dt})
        print(f"evm 2^{log_n}: {dt:.2f} s", file=sys.stderr, flush=True)
    # t / steps = a + b * (rw_rows + bytecode_rows): two-parameter least squares over the three sizes
    x = np.array([p["rw_rows"] + p["bytecode_rows"] for p in pts], dtype=np.float64)
    y = np.array([p["seconds"] / p["step_pairs"] for p in pts])
    A = np.stack([np.ones_like(x), x], axis=1)
    (a, b), *_ = np.linalg.lstsq(A, y, rcond=None) if len(pts) > 1 else ((0.0, float(y[0] / x[0])), None)
    if a < 0:  # a per-step cost cannot be negative: refit through the origin
        a, b = 0.0, float((x * y).sum() / (x * x).sum())
    full_tables = meta["n_rw"] + meta["n_bytecode"]
    per_step = a + b * full_tables
    return {"measured": pts,
            "fit": {"model": "seconds_per_step_pair = a + b * (rw_rows + bytecode_rows), a >= 0", "a": float(a), "b": float(b)},
            "extrapolated_2p18": {"step_pairs": (1 << 18) - 1, "table_rows": int(full_tables), "seconds": float(per_step * ((1 << 18) - 1)),
                                  "pairs_per_s": float(1.0 / per_step), "note": "EXTRAPOLATED from the fit, not measured"}}


def time_evm_prefixes(npz_path):
    """bench.py's live leg: prefixes of the trace cut by the caller (evm_prefix above, saved as '<log_n>/<table>' arrays), timed
    here through the reference's own verify_steps — this process sees the reference (PYTHONPATH), bench.py's does not"""
    from oracle.gen_golden_evm import unflatten
    from zkevm_specs.evm_circuit.main import verify_steps

    z = np.load(npz_path)
    logs = sorted({int(k.split("/")[0]) for k in z.files})
    pts = []
    for log_n in logs:
        pw = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(f"{log_n}/")}
        tables, steps = unflatten(pw)
        t0 = time.perf_counter()
        verify_steps(tables, steps)
        dt = time.perf_counter() - t0
        n = int(pw["steps"].shape[0]) - 1
        pts.append({"step_pairs": n, "rw_rows": int(pw["rw"].shape[0]), "bytecode_rows": int(pw["bytecode"].shape[0]), "seconds": dt, "pairs_per_s": n / dt})
    return {"measured": pts}


def time_state():
    from zkevm_specs.evm_circuit.table import MPTTableRow
    from zkevm_specs.state_circuit import Row, Tables, check_state_row
    from zkevm_specs.util import FQ, Word, WordOrValue

    from oracle import wire
    from zkevm_specs_amd.synth import synth_state_witness

    n = 1 << 16
    cols, flags, mpt = synth_state_witness(n, seed=2)
    W = lambda lo, hi: Word((FQ(lo), FQ(hi)), check=False)  # noqa: E731

    def wov(lo, hi, is_word):
        if is_word:
            return WordOrValue(W(lo, hi))
        v = WordOrValue(FQ(lo))
        v.hi = FQ(hi)
        return v

    rows = []
    for c, f in zip(wire.colmajor_to_rows(cols), flags):
        rows.append(Row(FQ(c[0]), FQ(c[1]), (FQ(c[2]), FQ(c[3]), FQ(c[4]), FQ(c[5]), W(c[6], c[7])), tuple(FQ(x) for x in c[8:18]),
                        tuple(FQ(x) for x in c[18:50]), wov(c[50], c[51], f & 1), wov(c[52], c[53], f & 2), W(c[54], c[55]), FQ(c[56])))
    tables = Tables(set(MPTTableRow(FQ(m[0]), FQ(m[1]), W(m[2], m[3]), W(m[4], m[5]), W(m[6], m[7]), W(m[8], m[9]), W(m[10], m[11]))
                        for m in wire.rowmajor_to_rows(mpt)))
    t0 = time.perf_counter()
    for i, row in enumerate(rows):
        check_state_row(row, rows[(i - 1) % n], rows[(i + 1) % n], tables)
    dt = time.perf_counter() - t0
    return {"rows": n, "mpt_rows": int(mpt.shape[0]), "seconds": dt, "rows_per_s": n / dt}


def time_bytecode():
    from zkevm_specs.bytecode_circuit import assign_push_table, check_bytecode_row
    from zkevm_specs.util import FQ

    from oracle.gen_golden_rows import unflatten_bytecode
    from zkevm_specs_amd.synth import synth_bytecode_witness

    code = bytes(np.random.default_rng(1).integers(0, 256, 256, dtype=np.uint8))
    r = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % (1 << 253)
    cols, keccak = synth_bytecode_witness([code], 9, r)
    rows, keccak_table = unflatten_bytecode(cols, keccak)
    push_table = assign_push_table()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        for i, row in enumerate(rows):
            check_bytecode_row(row, rows[(i + 1) % len(rows)], push_table, keccak_table, FQ(r))
    dt = (time.perf_counter() - t0) / reps
    return {"rows": len(rows), "seconds": dt, "rows_per_s": len(rows) / dt, "config": "256-byte contract, k = 9 (BASELINE configs[0])"}


def time_exp():
    from zkevm_specs.exp_circuit import verify_exp_circuit

    from oracle.gen_golden_rows import unflatten_exp
    from zkevm_specs_amd.synth import synth_exp_witness

    cols = synth_exp_witness(1 << 12, seed=5)
    rows = unflatten_exp(cols)

    class Circuit:
        def table(self):
            return rows

    t0 = time.perf_counter()
    verify_exp_circuit(Circuit())
    dt = time.perf_counter() - t0
    return {"rows": len(rows), "seconds": dt, "rows_per_s": len(rows) / dt}


def time_tx_sig():
    """BASELINE configs[3] at the sizes the pure-Python reference finishes: N signed legacy transactions built like the reference's
    own tests (tests/test_tx_circuit.py:103-114 gen_tx / sign_tx; keys sk_i = SHA-256(i) mod n), then
    `tx_circuit.verify_circuit` (tx_circuit.py:253-291) and `sig_circuit.verify_circuit` (sig_circuit.py:113-122) over them.
    secp256k1 / keccak / rlp are the oracle/refshim stand-ins (eth-keys 0.4.0's native backend restated), so the ECDSA share
    is pure Python exactly as with the real eth-keys native backend."""
    import hashlib

    import rlp
    from eth_keys import KeyAPI, keys
    from eth_utils import keccak
    from zkevm_specs import sig_circuit, tx_circuit
    from zkevm_specs.tx_circuit import Transaction, txs2witness
    from zkevm_specs.util import FQ, KeccakTable, Word
    from zkevm_specs.util.ec import ECDSAVerifyChip

    SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    chain_id = 1337
    r = FQ(0x1F2E3D4C5B6A79881726354433221100FFEEDDCCBBAA99887766554433221)
    pts = []
    for log_n in (6, 8, 10):
        n = 1 << log_n
        sks = [keys.PrivateKey((int.from_bytes(hashlib.sha256(i.to_bytes(4, "big")).digest(), "big") % (SECP_N - 1) + 1).to_bytes(32, "big"))
               for i in range(n)]
        txs, signed = [], []
        for i, sk in enumerate(sks):
            to = int.from_bytes(sks[(i + 1) % n].public_key.to_canonical_address(), "big")
            data = bytes([i % 256]) * (i % 64)
            tx = Transaction(300 + i, 1000 + 2 * i, 20000 + 3 * i, to, 0x30000 + 4 * i, data, 0, 0, 0)
            sign_data = rlp.encode([tx.nonce, tx.gas_price, tx.gas, tx.encode_to(), tx.value, tx.data, chain_id, 0, 0])
            h = keccak(sign_data)
            sig = sk.sign_msg_hash(h)
            txs.append(Transaction(tx.nonce, tx.gas_price, tx.gas, tx.to, tx.value, tx.data, sig.v + chain_id * 2 + 35, sig.r, sig.s))
            signed.append((h, sig, sk.public_key))
        max_calldata = sum(len(t.data) for t in txs) + 16
        witness = txs2witness(txs, chain_id, n, max_calldata, r)
        t0 = time.perf_counter()
        tx_circuit.verify_circuit(witness, n, max_calldata, r)
        t_tx = time.perf_counter() - t0
        rows, kt = [], KeccakTable()
        for h, sig, pk in signed:
            chip = ECDSAVerifyChip.assign(KeyAPI.Signature(vrs=(sig.v, sig.r, sig.s)), pk, h)
            kt.add(pk.to_bytes(), r)
            ph = keccak(pk.to_bytes())
            rows.append(sig_circuit.Row(ph, FQ(int.from_bytes(ph[-20:], "big")), Word(h), chip))
        t0 = time.perf_counter()
        sig_circuit.verify_circuit(sig_circuit.Witness(rows, kt), r)
        t_sig = time.perf_counter() - t0
        pts.append({"txs": n, "tx_circuit_seconds": t_tx, "sig_circuit_seconds": t_sig, "txs_per_s": n / (t_tx + t_sig),
                    "tx_rows": len(witness.rows), "calldata_bytes": max_calldata})
        print(f"tx+sig 2^{log_n}: {t_tx:.2f} + {t_sig:.2f} s", file=sys.stderr, flush=True)
    last = pts[-1]
    return {"txs": last["txs"], "txs_per_s": last["txs_per_s"], "measured": pts,
            "note": "Tx circuit + Sig circuit over the same signed txs; per-tx cost is flat in N (no table scans), so the largest measured size "
                    "is the figure (not extrapolated)"}


def main():
    import platform

    out = {"what": "the unmodified reference (/root/reference, tag 2024_08_07) on oracle/refshim dependency stand-ins, pure Python, 1 process / 1 core",
           "host": {"cpu": platform.processor() or open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
                    "cores_total": os.cpu_count(), "cores_used": 1, "python": platform.python_version()}}
    if os.environ.get("ZK_TIME_EVM_NPZ"):  # bench.py's live reference leg
        out["evm_live"] = time_evm_prefixes(os.environ["ZK_TIME_EVM_NPZ"])
        print(json.dumps(out))
        return
    only = set(os.environ.get("ZK_TIME_ONLY", "bytecode,exp,tx,state,evm").split(","))
    prev = os.environ.get("ZK_TIME_MERGE")  # an earlier result file: sections not re-timed are carried over, marked with their origin
    if prev:
        old = json.load(open(prev))
        for k in ("bytecode", "exp", "state", "evm", "tx"):
            if k in old and k not in only:
                out[k] = dict(old[k], carried_over_from=os.path.basename(prev))
    for k, fn in (("bytecode", time_bytecode), ("exp", time_exp), ("tx", time_tx_sig), ("state", time_state), ("evm", time_evm)):
        if k in only:
            out[k] = fn()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
