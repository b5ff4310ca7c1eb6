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
    n = cols.shape[1]
    lo, hi = shard_bounds(n, rank, world)
    idx = np.arange(lo - 1, hi + 1) % n
    return np.ascontiguousarray(cols[:, idx]), np.ascontiguousarray(flags[idx]), 1, 1 + (hi - lo), lo


def shard_evm(wire, rank, world, begin_with_first_step=False, end_with_last_step=False):
    """EVM witness shard: step pairs [lo, hi) need steps [lo, hi]; tables are replicated.
    Returns (wire_local, begin_flag, end_flag, lo)."""
    n_pairs = wire["steps"].shape[0] - 1
    lo, hi = shard_bounds(n_pairs, rank, world)
    local = dict(wire)
    local["steps"] = np.ascontiguousarray(wire["steps"][lo : hi + 1])
    for k in ("aux", "aux_kind"):  # per-step side data travels with the step rows
        if wire.get(k) is not None:
            local[k] = np.ascontiguousarray(wire[k][lo : hi + 1])
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
