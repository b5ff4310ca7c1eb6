"""Shared helpers for the EVM-circuit tests: golden-case loading and oracle / hostsim drivers."""
import ctypes
import glob
import os

import numpy as np

from oracle import evm_oracle as eo, wire

FIELDS = ("steps", "rw", "rw_flags", "bytecode", "tx", "tx_flags", "block", "block_flags", "copy", "keccak", "exp", "aux",
          "aux_kind", "withdrawals", "sig", "ecc")
_OPTIONAL = {"copy": 14, "keccak": 5, "exp": 11, "withdrawals": 4, "sig": 9, "ecc": 13}  # tables only some gadgets need; absent = empty


def with_defaults(w):
    w = dict(w)
    for k, nc in _OPTIONAL.items():
        if k not in w:
            w[k] = np.zeros((0, nc, 4), dtype=np.uint64)
    n = w["steps"].shape[0]
    if "aux" not in w:  # StepState.aux_data: absent for every step
        w["aux"] = np.zeros((n, 2, 4), dtype=np.uint64)
        w["aux_kind"] = np.zeros(n, dtype=np.uint32)
    return w


def golden_files(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "evm_*.npz")))


def load_cases(fn):
    g = np.load(fn)
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        w = with_defaults({f: g[f"{k}_{f}"] for f in FIELDS if f"{k}_{f}" in g.files})
        yield str(nm), w, g[k + "_opts"], g[k + "_ref_kind"]


def to_witness(w):
    w = with_defaults(w)
    return eo.EvmWitness(wire.rowmajor_to_rows(w["steps"]), wire.rowmajor_to_rows(w["rw"]), w["rw_flags"],
                         wire.rowmajor_to_rows(w["bytecode"]), wire.rowmajor_to_rows(w["tx"]), w["tx_flags"],
                         wire.rowmajor_to_rows(w["block"]), w["block_flags"], wire.rowmajor_to_rows(w["copy"]),
                         wire.rowmajor_to_rows(w["keccak"]), wire.rowmajor_to_rows(w["exp"]),
                         wire.rowmajor_to_rows(w["aux"]), w["aux_kind"], wire.rowmajor_to_rows(w["withdrawals"]),
                         wire.rowmajor_to_rows(w["sig"]), wire.rowmajor_to_rows(w["ecc"]))


def oracle_status(w, opts=(0, 0)):
    return eo.verify_steps(to_witness(w), bool(opts[0]), bool(opts[1]))


def hostsim_status(lib, w, opts=(0, 0), generic_index=False):
    w = with_defaults(w)
    a = {k: np.ascontiguousarray(w[k]) for k in FIELDS}
    n = a["steps"].shape[0]
    st = np.zeros(max(n - 1, 1), dtype=np.uint32)
    vp = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
    u64 = ctypes.c_uint64
    lib.sim_evm_verify(vp(a["steps"]), u64(n), vp(a["rw"]), vp(a["rw_flags"]), u64(a["rw"].shape[0]),
                       vp(a["bytecode"]), u64(a["bytecode"].shape[0]), vp(a["tx"]), vp(a["tx_flags"]),
                       u64(a["tx"].shape[0]), vp(a["block"]), vp(a["block_flags"]), u64(a["block"].shape[0]),
                       vp(a["copy"]), u64(a["copy"].shape[0]), vp(a["keccak"]), u64(a["keccak"].shape[0]),
                       vp(a["exp"]), u64(a["exp"].shape[0]), vp(a["aux"]), vp(a["aux_kind"]),
                       vp(a["withdrawals"]), u64(a["withdrawals"].shape[0]), vp(a["sig"]), u64(a["sig"].shape[0]),
                       vp(a["ecc"]), u64(a["ecc"].shape[0]), ctypes.c_uint32(a["aux"].shape[1]),
                       ctypes.c_uint32(int(opts[0]) | (int(opts[1]) << 1) | (4 if generic_index else 0)), vp(st))
    return st[: n - 1].tolist()


def fuzz_wire(w, rng, copy=True):
    """Overwrite 1..3 random cells of the steps / rw / bytecode tables (canonical values); copy=False: in place (large traces)."""
    P = wire.P
    w = {k: (v.copy() if copy else v) for k, v in w.items() if k in FIELDS}

    def put(arr, idx, val):
        arr[idx] = np.frombuffer(int(val % P).to_bytes(32, "little"), dtype="<u8")

    def cur(arr, idx):
        return int.from_bytes(arr[idx].tobytes(), "little")

    for _ in range(rng.choice([1, 1, 2, 3])):
        which = rng.choice(["steps", "steps", "rw", "rw", "rw", "bytecode", "flags"])
        aux = [k for k in ("copy", "keccak", "exp", "sig", "ecc") if k in w and w[k].shape[0]]
        if "aux" in w and w["aux"].shape[1] > 2 and rng.random() < 0.3:  # precompile gadgets read their inputs from aux_data
            i = rng.choice([j for j in range(w["aux"].shape[0]) if w["aux_kind"][j]] or [0])
            c = rng.randrange(w["aux"].shape[1])
            old = cur(w["aux"], (i, c))
            put(w["aux"], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(P), old ^ (1 << rng.randrange(128)), 27, 28]))
        elif aux and rng.random() < 0.25:
            k = rng.choice(aux)
            i, c = rng.randrange(w[k].shape[0]), rng.randrange(w[k].shape[1])
            old = cur(w[k], (i, c))
            put(w[k], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(P), old ^ (1 << rng.randrange(64))]))
        elif which == "steps":
            c, i = rng.randrange(1, 13), rng.randrange(w["steps"].shape[0])
            if c in (3, 4):
                put(w["steps"], (i, c), rng.randrange(2))
            else:
                old = cur(w["steps"], (i, c))
                put(w["steps"], (i, c), rng.choice([old + 1, old - 1, 0, rng.randrange(P), old ^ 1, 2**64, 2**128 + old]))
        elif which == "rw" and w["rw"].shape[0]:
            i, c = rng.randrange(w["rw"].shape[0]), rng.randrange(14)
            old = cur(w["rw"], (i, c))
            put(w["rw"], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(P), old ^ (1 << rng.randrange(128)),
                                              2**128, 2**255 % P, old + 2**128, rng.randrange(2**128)]))
        elif which == "bytecode" and w["bytecode"].shape[0]:
            i, c = rng.randrange(w["bytecode"].shape[0]), rng.randrange(2, 6)
            old = cur(w["bytecode"], (i, c))
            put(w["bytecode"], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(256), rng.randrange(P)]))
        elif w["rw_flags"].shape[0]:
            i = rng.randrange(w["rw_flags"].shape[0])
            w["rw_flags"][i] ^= np.uint32(rng.choice([1, 2]))
    return w
