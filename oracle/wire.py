"""Wire format helpers (test infrastructure): one Fr cell = 4 x u64 little-endian canonical.

Column-major witnesses are `uint64[n_cells, n_rows, 4]`; row-major lookup tables are
`uint64[n_rows, n_cells, 4]`; per-row type bits are `uint32[n_rows]`.
"""
import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def ints_to_cells(vals):
    """list of python ints (0 <= v < 2^256) -> uint64[len, 4]"""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, 4).copy()


def cells_to_ints(arr):
    """uint64[..., 4] -> flat list of python ints"""
    a = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[i : i + 32], "little") for i in range(0, len(raw), 32)]


def rows_to_colmajor(rows):
    """rows: list of lists of ints [n][ncells] -> uint64[ncells, n, 4]"""
    n = len(rows)
    nc = len(rows[0]) if n else 0
    flat = ints_to_cells([v for r in rows for v in r]).reshape(n, nc, 4)
    return np.ascontiguousarray(flat.transpose(1, 0, 2))


def rows_to_rowmajor(rows, ncells):
    n = len(rows)
    if n == 0:
        return np.zeros((0, ncells, 4), dtype=np.uint64)
    return ints_to_cells([v for r in rows for v in r]).reshape(n, ncells, 4)


def colmajor_to_rows(arr):
    nc, n, _ = arr.shape
    flat = cells_to_ints(np.ascontiguousarray(arr.transpose(1, 0, 2)))
    return [flat[i * nc : (i + 1) * nc] for i in range(n)]


def rowmajor_to_rows(arr):
    n, nc, _ = arr.shape
    flat = cells_to_ints(arr)
    return [flat[i * nc : (i + 1) * nc] for i in range(n)]
