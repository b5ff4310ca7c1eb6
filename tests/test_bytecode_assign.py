"""Bytecode-circuit witness assignment: oracle vs the rows the unmodified reference's
`assign_bytecode_circuit` returned, the device functions' logic (hostsim) vs the same rows, and the HIP path (gpu) incl.
the chain bytes -> keccak table + unrolled rows -> assigned rows -> Bytecode circuit, all on the device."""
import ctypes
import os
import random

import numpy as np
import pytest

from oracle import bytecode_assign_oracle as B, keccak_table, wire

vp = lambda x: ctypes.c_void_p(np.ascontiguousarray(x).ctypes.data)  # noqa: E731


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "bytecode_assign_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield (str(nm), int(g[k + "_k"]), np.ascontiguousarray(g[k + "_in_rows"]), np.ascontiguousarray(g[k + "_offsets"]),
               np.ascontiguousarray(g[k + "_lengths"]), wire.cells_to_ints(g[k + "_r"])[0], g[k + "_rows"])


def _hostsim(lib, k, rows_in, offsets, lengths, r):
    out = np.zeros((12, 1 << k, 4), dtype=np.uint64)
    rc = np.frombuffer(int(r).to_bytes(32, "little"), dtype="<u8").copy()
    lib.sim_bytecode_assign(vp(rows_in), ctypes.c_uint64(rows_in.shape[0]), vp(offsets), vp(lengths), ctypes.c_uint64(len(lengths)),
                            ctypes.c_uint32(k), vp(rc), vp(out))
    return out


def _unroll(codes, digest):
    """BytecodeTableRows of Bytecode.table_assignments() (evm_circuit/typing.py) in wire form: Header row (value = length),
    then one Byte row per byte with is_code from the push-data walk"""
    rows, offsets, lengths = [], [0], []
    for code in codes:
        h = digest(code)
        lo, hi = h & ((1 << 128) - 1), h >> 128
        rows.append([lo, hi, 1, 0, 0, len(code)])
        left = 0
        for idx, b in enumerate(code):
            is_code = left == 0
            rows.append([lo, hi, 2, idx, int(is_code), b])
            left = (b - 0x5F if 0x60 <= b <= 0x7F else 0) if is_code else left - 1
        offsets.append(len(rows))
        lengths.append(len(code))
    return wire.rows_to_rowmajor(rows, 6), np.array(offsets, dtype=np.uint64), np.array(lengths, dtype=np.uint64)


def test_oracle_and_kernel_logic_match_the_reference(golden_dir, hostsim):
    n = 0
    for name, k, rows_in, offsets, lengths, r, ref_rows in _cases(golden_dir):
        assert B.assign(k, wire.rowmajor_to_rows(rows_in), offsets, lengths, r) == wire.colmajor_to_rows(ref_rows), name
        assert np.array_equal(_hostsim(hostsim, k, rows_in, offsets, lengths, r), ref_rows), name
        n += 1
    assert n >= 90


def test_kernel_logic_on_large_random_codes(hostsim):
    """chunk boundaries (63 / 64 / 65 / 24,576 rows), PUSH32 runs across chunks, truncation and padding"""
    rng = random.Random(12)
    r = rng.randrange(wire.P)
    codes = [bytes(rng.choice([rng.randrange(256), 0x7F, 0x60]) for _ in range(n)) for n in (62, 63, 64, 0, 129, 24576, 1)]
    rows_in, offsets, lengths = _unroll(codes, lambda c: rng.getrandbits(256))
    for k in (6, 15):
        got = _hostsim(hostsim, k, rows_in, offsets, lengths, r)
        assert wire.colmajor_to_rows(got) == B.assign(k, wire.rowmajor_to_rows(rows_in), offsets, lengths, r), k


def _malformed_inputs():
    """value cells that are not bytes — 2^32 - 1 (still on the lazy path), 2^32, 2^64 + 5, a full field element — on both sides of
    chunk boundaries and in a bytecode's first chunk (whose Header row is skipped): value_rlc must follow the reference's
    recurrence `value_rlc * r + value` with the value as it is (bytecode_circuit.py:117-130 never range-checks it)"""
    rng = random.Random(77)
    r = rng.randrange(wire.P)
    codes = [bytes(rng.randrange(256) for _ in range(n)) for n in (200, 64, 130, 5)]
    rows_in, offsets, lengths = _unroll(codes, lambda c: rng.getrandbits(256))
    rows_in = rows_in.copy()
    put = lambda row, val: rows_in.__setitem__((row, 5), np.frombuffer(int(val % wire.P).to_bytes(32, "little"), dtype="<u8"))  # noqa: E731
    for row, val in ((3, (1 << 32) - 1), (40, 1 << 32), (63, (1 << 64) + 5), (64, wire.P - 1), (65, 0x7F), (130, rng.randrange(wire.P)),
                     (int(offsets[1]) + 1, 1 << 200), (int(offsets[2]) + 64, (1 << 32) + 0x60), (int(offsets[3]) + 2, 1 << 40)):
        put(row, val)
    return rows_in, offsets, lengths, r


def test_kernel_logic_on_values_that_are_not_bytes(hostsim):
    rows_in, offsets, lengths, r = _malformed_inputs()
    for k in (9, 10):
        got = _hostsim(hostsim, k, rows_in, offsets, lengths, r)
        assert wire.colmajor_to_rows(got) == B.assign(k, wire.rowmajor_to_rows(rows_in), offsets, lengths, r), k


@pytest.mark.gpu
def test_hip_on_values_that_are_not_bytes():
    from zkevm_specs_amd import engine

    rows_in, offsets, lengths, r = _malformed_inputs()
    for k in (9, 10):
        with engine.open_bytecode_assign(rows_in, offsets, lengths, k, r) as s:
            assert s.run().ok
            assert wire.colmajor_to_rows(s.rows()) == B.assign(k, wire.rowmajor_to_rows(rows_in), offsets, lengths, r), k


@pytest.mark.gpu
def test_hip_matches_the_reference_rows(golden_dir):
    from zkevm_specs_amd import engine

    for name, k, rows_in, offsets, lengths, r, ref_rows in _cases(golden_dir):
        with engine.open_bytecode_assign(rows_in, offsets, lengths, k, r) as s:
            res = s.run()
            assert res.ok and res.rows_evaluated == 1 << k, name
            assert np.array_equal(s.rows(), ref_rows), name


@pytest.mark.gpu
def test_bytes_to_verified_bytecode_circuit_on_device():
    """contracts -> keccak table (zk_keccak_table) + unrolled rows -> circuit rows assigned in HBM (zk_bytecode_assign)
    -> Bytecode circuit evaluated on the same buffers: equals the oracle's rows and satisfies every constraint"""
    import torch

    from zkevm_specs_amd import engine

    rng = random.Random(3)
    r = rng.randrange(wire.P)
    codes = [bytes(rng.choice([rng.randrange(256), rng.randrange(0x60, 0x80)]) for _ in range(n)) for n in (24576, 24000, 700, 64, 0, 1, 9000)]
    krows = engine.keccak_table(codes, r, engine.KECCAK_MODE_CIRCUIT)
    assert np.array_equal(krows, keccak_table.table_rows(codes, r, keccak_table.MODE_CIRCUIT)[0])
    digest = {c: int.from_bytes(krows[i, 3].tobytes(), "little") | (int.from_bytes(krows[i, 4].tobytes(), "little") << 128) for i, c in enumerate(codes)}
    rows_in, offsets, lengths = _unroll(codes, lambda c: digest[c])
    k = 16
    dev = lambda x: torch.from_numpy(x.view(np.int64)).cuda()  # noqa: E731
    d_rows = torch.empty((12, 1 << k, 4), dtype=torch.int64, device="cuda")
    with engine.open_bytecode_assign(dev(rows_in), dev(offsets), dev(lengths), k, r, rows_dev=d_rows) as s:
        assert s.run().ok
    got = d_rows.cpu().numpy().view(np.uint64)
    assert wire.colmajor_to_rows(got) == B.assign(k, wire.rowmajor_to_rows(rows_in), offsets, lengths, r)
    with engine.open_bytecode(d_rows, dev(krows), r) as s:
        res = s.run()
    assert res.ok and res.rows_evaluated == 1 << k
