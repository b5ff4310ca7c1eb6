#!/usr/bin/env python3
"""Tx- and Sig-circuit golden vectors from the UNMODIFIED reference (build container only; needs the
secp256k1 shim in oracle/refshim/eth_keys).  Replays reference tests/test_tx_circuit.py and
tests/test_sig_circuit.py with their `verify` drivers intercepted, adds attribute-level tampering,
and records the per-unit outcome of the reference (tx: `verify_circuit` on the single tx slot;
sig: `Row.verify`)."""
import copy
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle.gen_golden import kind_of_exception  # noqa: E402


def ref_tx_outcomes(witness, max_txs, max_calldata, r):
    from zkevm_specs.tx_circuit import Witness, verify_circuit

    out = []
    for i in range(max_txs):
        try:
            verify_circuit(Witness(witness.rows[i * 12:(i + 1) * 12], witness.keccak_table, witness.sign_verifications[i:i + 1]),
                           1, max_calldata, r)
            out.append(0)
        except Exception as e:  # noqa: BLE001
            out.append(kind_of_exception(e))
    return out


def ref_sig_outcomes(witness, r):
    out = []
    for i, row in enumerate(witness.rows):
        try:
            row.verify(witness.keccak_table, r, "")
            out.append(0)
        except Exception as e:  # noqa: BLE001
            out.append(kind_of_exception(e))
    return out


def tamper_tx(w, rng):
    from zkevm_specs.tx_circuit import KeccakTable, Secp256k1ScalarField, Witness
    from zkevm_specs.util import FQ, Word, WordOrValue

    w = copy.deepcopy(w)
    i = rng.randrange(len(w.sign_verifications))
    sv = w.sign_verifications[i]
    c = rng.randrange(9)
    if c == 0:
        sv.address = FQ(rng.getrandbits(160))
    elif c == 1:
        sv.msg_hash = Word(rng.getrandbits(256))
    elif c == 2:
        sv.pub_key_x_bytes = bytes(rng.getrandbits(8) for _ in range(32))
    elif c == 3:
        sv.ecdsa_chip.signature = (Secp256k1ScalarField(rng.getrandbits(250) + 1), Secp256k1ScalarField(rng.getrandbits(250) + 1))
    elif c == 4:
        sv.pub_key_hash = bytes(rng.getrandbits(8) for _ in range(32))
    elif c == 5:
        w.rows[i * 12 + 3].value = WordOrValue(FQ(rng.getrandbits(160)))
    elif c == 6:
        w.rows[i * 12 + 11].value = WordOrValue(Word(rng.getrandbits(256)))
    elif c == 7:
        w.rows[i * 12 + 3].value = WordOrValue(Word(sv.address.n))
    else:
        return Witness(w.rows, KeccakTable(), w.sign_verifications)
    return w


def tamper_sig(w, rng):
    from zkevm_specs.util import FQ, KeccakTable, Word
    from zkevm_specs.sig_circuit import Witness

    w = copy.deepcopy(w)
    row = w.rows[rng.randrange(len(w.rows))]
    c = rng.randrange(8)
    if c == 0:
        row.recovered_addr = FQ(rng.getrandbits(160))
    elif c == 1:
        row.msg_hash = Word(rng.getrandbits(256))
    elif c == 2:
        row.sig_v = FQ(rng.randrange(0, 4))
    elif c == 3:
        row.sig_r = Word(rng.getrandbits(256))
    elif c == 4:
        row.is_valid = not row.is_valid
    elif c == 5:
        row.pub_key_hash = bytes(rng.getrandbits(8) for _ in range(32))
    elif c == 6:
        row.pub_key_y_bytes = bytes(rng.getrandbits(8) for _ in range(32))
    else:
        return Witness(w.rows, KeccakTable())
    return w


def main():
    import test_sig_circuit as TS
    import test_tx_circuit as TT
    from zkevm_specs_amd.flatten import flatten_sig_witness, flatten_tx_witness

    rng = random.Random(404)
    out, names = {}, []

    def store(name, kind, wire, ref, r):
        key = f"c{len(names):04d}"
        names.append(f"{kind}:{name}")
        for k, v in wire.items():
            out[f"{key}_{k}"] = v
        out[f"{key}_r"] = np.frombuffer(int(r.n).to_bytes(32, "little"), dtype="<u8").copy()
        out[f"{key}_is_sig"] = np.array([1 if kind == "sig" else 0], dtype=np.uint8)
        out[f"{key}_ref_kind"] = np.array(ref, dtype=np.uint8)

    # ---- tx circuit
    tx_cases = []
    current = [None]

    def cap_tx(witness, MAX_TXS, MAX_CALLDATA_BYTES, chain_id, r, success=True):
        tx_cases.append((current[0], witness, MAX_TXS, MAX_CALLDATA_BYTES, r, success))

    TT.verify = cap_tx
    for nm in sorted(dir(TT)):
        if nm.startswith("test_") and nm not in ("test_ecdsa_verify_chip", "test_tx2witness"):
            current[0] = nm
            getattr(TT, nm)()
    for nm, w, mt, mc, r, success in list(tx_cases):
        ref = ref_tx_outcomes(w, mt, mc, r)
        assert (not any(ref)) == success, (nm, ref)
        store(nm, "tx", flatten_tx_witness(w, mt), ref, r)
        if success:
            for k in range(10):
                tw = tamper_tx(w, rng)
                store(f"{nm}#tamper{k}", "tx", flatten_tx_witness(tw, mt), ref_tx_outcomes(tw, mt, mc, r), r)
    # ---- sig circuit
    sig_cases = []

    def cap_sig(witness, keccak_randomness, success=True):
        sig_cases.append((current[0], witness, keccak_randomness, success))

    TS.verify = cap_sig
    for nm in sorted(dir(TS)):
        if nm.startswith("test_") and nm != "test_ecdsa_verify_chip":
            current[0] = nm
            try:
                getattr(TS, nm)()
            except Exception as e:  # some reference tests expect exceptions outside `verify`
                print("skip", nm, type(e).__name__)
    for nm, w, r, success in list(sig_cases):
        ref = ref_sig_outcomes(w, r)
        store(nm, "sig", flatten_sig_witness(w), ref, r)
        if success:
            for k in range(10):
                tw = tamper_sig(w, rng)
                store(f"{nm}#tamper{k}", "sig", flatten_sig_witness(tw), ref_sig_outcomes(tw, r), r)
    out["names"] = np.array(names)
    fn = os.path.join(GOLDEN, "sign_cases.npz")
    np.savez_compressed(fn, **out)
    nf = sum(int(out[f"c{i:04d}_ref_kind"].any()) for i in range(len(names)))
    print(f"sign: {len(names)} cases ({nf} with failing units) -> {os.path.getsize(fn)//1024} KiB")


if __name__ == "__main__":
    main()
