"""Row sharding across the GPUs of one node (one process per GPU) and the tally reduction.

The path shards embarrassingly: contiguous row ranges per rank with a read-only
halo (State: one row before and after, modulo n; EVM: the step after the last pair), lookup
tables replicated.  The only collective is one all-gather of the tally — (fail count, first failing global row,
its status code) per rank, 24 bytes over RCCL/xGMI (`nccl` backend) or gloo in the CPU tests — reduced locally
(SUM of the counts, MIN of the rows).
"""
import numpy as np


def shard_bounds(n, rank, world):
    """[lo, hi) of the units (rows / step pairs) rank `rank` evaluates."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_state(cols, flags, rank, world):
    """State witness shard: rows [lo-1, hi] (mod n) of the column-major table, i.e. the rank's rows
    plus one halo row on each side.  Returns (cols_local, flags_local, eval_lo, eval_hi, lo)."""
    return shard_rows(cols, flags, rank, world, "state")


HALO = {"state": (1, 1), "bytecode": (0, 1), "exp": (0, 1), "pi": (0, 1), "copy": (0, 2), "tx": (0, 0), "sig": (0, 0)}
"""rows before / after a rank's range that its boundary rows read (SURVEY.md §8e; the neighbours wrap modulo n:
state_circuit.py:492 prev / next, bytecode_circuit.py:37 next, exp_circuit.py:88-97 next, copy_circuit.py:92-130
rows i + 1 and i + 2, tx_circuit.py:253-291 none)"""


def _take(a, idx, axis):
    """rows `idx` along `axis` of a numpy array or a torch tensor (device tensors are gathered on the device)"""
    if hasattr(a, "is_cuda"):
        import torch

        return a.index_select(axis, torch.as_tensor(idx, device=a.device)).contiguous()
    return np.ascontiguousarray(np.take(a, idx, axis=axis))


def shard_rows(cols, flags, rank, world, circuit):
    """Row shard of a column-major circuit witness cols[c, n, 4] (+ optional flags[n]): the rank's rows [lo, hi) with the
    circuit's halo (HALO) on either side, modulo n.  Returns (cols_local, flags_local, eval_lo, eval_hi, lo): the
    session over the local rows evaluates [eval_lo, eval_hi) (Session.set_range) and its row i is global row
    lo + i - eval_lo."""
    n = int(cols.shape[1])
    before, after = HALO[circuit]
    lo, hi = shard_bounds(n, rank, world)
    idx = np.arange(lo - before, hi + after) % n
    return _take(cols, idx, 1), (None if flags is None else _take(flags, idx, 0)), before, before + (hi - lo), lo


def shard_units(wire, rank, world):
    """Tx / Sig units shard (no halo: units are independent, tx_circuit.py:253-291): bytes[n, 9, 32], cells[8, n, 4],
    meta[n, 4] and the units' twelve fixed tx-table rows are cut to the rank's units, the keccak table stays whole (replicated).  Returns (wire_local, lo)."""
    n = int(wire["bytes"].shape[0])
    lo, hi = shard_bounds(n, rank, world)
    idx = np.arange(lo, hi)
    local = dict(wire)
    local["bytes"], local["cells"], local["meta"] = _take(wire["bytes"], idx, 0), _take(wire["cells"], idx, 1), _take(wire["meta"], idx, 0)
    # the Tx circuit reads unit i's CallerAddress / TxSignHash rows at tx_rows[12 i + 3], [12 i + 11] (tx_circuit.py:270-289): the
    # twelve fixed rows of a tx belong to its unit and travel with it
    if wire.get("tx_rows") is not None and int(wire["tx_rows"].shape[0]) > 0:
        assert int(wire["tx_rows"].shape[0]) >= 12 * n, "tx_rows: twelve fixed rows per unit expected"
        ridx = np.arange(12 * lo, 12 * hi)
        local["tx_rows"], local["tx_flags"] = _take(wire["tx_rows"], ridx, 0), _take(wire["tx_flags"], ridx, 0)
    return local, lo


def shard_evm(wire, rank, world, begin_with_first_step=False, end_with_last_step=False):
    """EVM witness shard: step pairs [lo, hi) need steps [lo, hi]; tables are replicated.
    Returns (wire_local, begin_flag, end_flag, lo)."""
    n_pairs = wire["steps"].shape[0] - 1
    lo, hi = shard_bounds(n_pairs, rank, world)
    local = dict(wire)
    local["steps"] = _take(wire["steps"], np.arange(lo, hi + 1), 0)
    for k in ("aux", "aux_kind"):  # per-step side data travels with the step rows
        if wire.get(k) is not None:
            local[k] = _take(wire[k], np.arange(lo, hi + 1), 0)
    return local, bool(begin_with_first_step and rank == 0), bool(end_with_last_step and rank == world - 1), lo


def reduce_tally(fail_count, first_fail_row, first_fail_code, row_offset, device=None, group=None):
    """Exchange the per-rank tallies: ONE collective (an all-gather of three words per rank: fail count, first failing
    GLOBAL row, its status code — 24 bytes, latency-bound), no host synchronisation before the single read-back at the end.
    SUM and lexicographic MIN are then taken locally, identically on every rank.  `first_fail_row` is local (None = no
    failure).  Returns (total_fail_count, first_fail_global_row or None, its status code)."""
    import torch
    import torch.distributed as dist

    none = 1 << 62
    mine = torch.tensor([int(fail_count), none if first_fail_row is None else int(first_fail_row) + int(row_offset),
                         0 if first_fail_row is None else int(first_fail_code)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        every = torch.empty(world * 3, dtype=torch.int64, device=device)  # flat: gloo's allgather wants a 1-d output
        dist.all_gather_into_tensor(every, mine, group=group)
    else:
        every = mine
    every = every.reshape(-1, 3).cpu().tolist()  # the only synchronisation
    total = sum(r[0] for r in every)
    row, code = min((r[1], r[2]) for r in every)
    if row == none:
        return total, None, 0
    return total, row, code


class RcclTally:
    """The same exchange through the C ABI (`zk_dist_*`, include/zkevm_hip.h): the engine's own RCCL communicator, for hosts without
    torch.distributed.  Here the communicator id still travels over torch.distributed (any backend) when it is initialised; a
    single process (world 1) needs nothing.  `reduce(result, row_offset)` is collective and returns what `reduce_tally` returns."""

    def __init__(self, rank=0, world=1, device=None, group=None):
        import ctypes

        from . import _lib
        from .engine import check

        self._lib = _lib.init(device)
        self._check = check
        ident = (ctypes.c_uint8 * 128)()
        if rank == 0:
            check(self._lib.zk_dist_unique_id(ident), "zk_dist_unique_id", self._lib)
        if world > 1:
            import torch.distributed as dist

            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        self._h = ctypes.c_void_p()
        check(self._lib.zk_dist_init(ident, int(rank), int(world), ctypes.byref(self._h)), "zk_dist_init", self._lib)

    def reduce(self, result, row_offset=0):
        """result: an engine.Result (or anything with fail_count / first_fail_row / first_fail_code) of this rank's shard"""
        import ctypes

        from ._lib import ZkResult

        raw = ZkResult()
        raw.fail_count = int(result.fail_count)
        raw.first_fail_row = 0xFFFFFFFFFFFFFFFF if result.first_fail_row is None else int(result.first_fail_row)
        raw.first_fail_code = int(getattr(result, "first_fail_code", 0) or 0)
        raw.rows_evaluated = int(getattr(result, "rows_evaluated", 0) or 0)
        raw.kernel_ms = float(getattr(result, "kernel_ms", 0.0) or 0.0)
        out = ZkResult()
        self._check(self._lib.zk_dist_tally(self._h, ctypes.byref(raw), int(row_offset), ctypes.byref(out)), "zk_dist_tally", self._lib)
        if out.first_fail_row == 0xFFFFFFFFFFFFFFFF:
            return int(out.fail_count), None, 0
        return int(out.fail_count), int(out.first_fail_row), int(out.first_fail_code)

    def close(self):
        if self._h:
            self._lib.zk_dist_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
