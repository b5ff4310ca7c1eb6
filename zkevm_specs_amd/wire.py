"""Wire format of the C-ABI: one Fr cell = 4 x u64 little-endian canonical (== `FQ.n`).

Witness rows are column-major `uint64[n_cells, n_rows, 4]` (coalesced per-row loads on the
device); lookup tables are row-major `uint64[n_rows, n_cells, 4]`; `uint32[n_rows]` carries
the per-row type bits (is_word of WordOrValue columns, reference util/arithmetic.py:171-195).
"""
import numpy as np

FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def ints_to_cells(vals):
    buf = b"".join([int(v).to_bytes(32, "little") for v in vals])
    return np.frombuffer(buf, dtype="<u8").reshape(-1, 4).copy()


def cells_to_ints(arr):
    raw = np.ascontiguousarray(arr, dtype="<u8").tobytes()
    return [int.from_bytes(raw[i : i + 32], "little") for i in range(0, len(raw), 32)]


def rows_to_colmajor(rows, ncells):
    n = len(rows)
    if n == 0:
        return np.zeros((ncells, 0, 4), dtype=np.uint64)
    flat = ints_to_cells([v for r in rows for v in r]).reshape(n, ncells, 4)
    return np.ascontiguousarray(flat.transpose(1, 0, 2))


def rows_to_rowmajor(rows, ncells):
    n = len(rows)
    if n == 0:
        return np.zeros((0, ncells, 4), dtype=np.uint64)
    return ints_to_cells([v for r in rows for v in r]).reshape(n, ncells, 4)
