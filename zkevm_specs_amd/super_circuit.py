"""Super-circuit driver (BASELINE config 5): the EVM, State, Bytecode and Tx circuit kernels evaluated over one
witness set on one GPU; ranks hold independent shards and all-reduce the tally (distributed.py).

The reference has no super-circuit driver: each circuit has its own `verify_*` entry and
they are only related through the tables they share.  This module launches the same C-ABI sessions the per-circuit
mirrors use, each session bound to its own HIP stream (zk_session_set_stream), and sums their tallies.  What is shared in the synthetic witness:
  * the contracts the EVM trace executes ARE the byte strings of the Bytecode circuit's rows: the circuit rows are assigned
    on the device (`zk_bytecode_assign`) from the very bytecode table the EVM circuit looks up (its rows, sorted by
    hash / tag / index, are the unrolled bytecodes), and the code hashes are real keccak-256 digests taken from the
    keccak table the device builds from those byte strings (`zk_keccak_table`, mode 0) — the same table the Bytecode
    circuit looks up;
  * the State circuit's rows come out of the device-side witness assignment (`zk_state_assign`) of a synthetic op
    list sized like the trace's RW table.  They are NOT derived from the EVM trace's RW table: the synthetic trace
    does not model cross-step stack / memory consistency (synth_evm.py), which is exactly what the State circuit
    checks, so such rows would not satisfy it;
  * the Tx circuit's units are independent synthetic transactions (the trace has a single root call).
The Copy and Exp circuits have no rows here: BASELINE config 3's opcode mix contains no copy / EXP steps.

Concurrency note: the sessions overlap on the device only when their streams land on different hardware queues.  The HIP runtime
reads GPU_MAX_HW_QUEUES (default 4) at its first call; `zkevm_specs_amd._lib.load()` defaults it to 16 — call it (or `init()`)
before the process's first HIP call (e.g. before `torch.cuda.set_device`), or export the variable yourself.
"""
import os

import numpy as np

from . import engine
from .errors import exception_for_code
from .synth import synth_bytecode_witness, synth_state_ops, synth_tx_witness
from .synth_evm import synth_evm_codes, synth_evm_trace

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
CIRCUITS = ("evm", "state", "bytecode", "tx")
BLOCK_CIRCUITS = CIRCUITS + ("copy", "exp")


def _digest_of_row(row):
    lo = int.from_bytes(row[3].tobytes(), "little")
    hi = int.from_bytes(row[4].tobytes(), "little")
    return lo | (hi << 128)


def synth_super(log_total=20, seed=5, keccak_rows_of=None):
    """Witness parts of a 2^log_total-row super circuit: EVM 2^(log_total-2) steps, Bytecode rows of its contracts
    (padded to a power of two), 2^(log_total-8) Tx units, and State rows for the remainder.
    keccak_rows_of(codes, r) -> uint64[n, 5, 4] builds the keccak table of the contracts; default: on the GPU
    (engine.keccak_table).  Returns dict(evm=wire, state_ops=(ops, flags), bytecode=(rows, keccak, r), tx=(wire, r),
    rows={circuit: evaluated rows}, codes=[bytes], meta=...)."""
    assert log_total >= 12
    r = (0x1234567 * (seed + 1) ** 7 + 0x9E3779B97F4A7C15) % P
    # contracts sized so that their Bytecode-circuit rows take at most 1/16 of the total
    n_contracts = min(16, 1 << (log_total - 16)) if log_total >= 16 else 2
    seg_len = 640 if log_total >= 16 else 96
    codes = synth_evm_codes(seed, seg_len=seg_len, n_contracts=n_contracts)
    if keccak_rows_of is None:
        keccak_rows_of = lambda c, rr: engine.keccak_table(c, rr, engine.KECCAK_MODE_CIRCUIT)  # noqa: E731
    keccak = keccak_rows_of(codes, r)
    hashes = [_digest_of_row(keccak[i]) for i in range(len(codes))]
    n_steps = 1 << (log_total - 2)
    evm = synth_evm_trace(n_steps, seed=seed, seg_len=seg_len, n_contracts=n_contracts, code_hashes=hashes)
    meta = evm.pop("meta")
    n_code_rows = sum(len(c) + 1 for c in codes)
    k = max(6, int(np.ceil(np.log2(n_code_rows + 1))))
    digest = dict(zip(codes, hashes))
    bc_rows, bc_keccak = synth_bytecode_witness(codes, k, r, digest=lambda c: digest[c])
    # the generator's own keccak rows (python RLC + the digests above) must be the device-built table rows
    dev_sorted = sorted(tuple(int.from_bytes(keccak[i, c].tobytes(), "little") for c in range(5)) for i in range(len(codes)))
    gen_sorted = sorted(tuple(int.from_bytes(bc_keccak[i, c].tobytes(), "little") for c in range(5)) for i in range(bc_keccak.shape[0]))
    assert dev_sorted == gen_sorted, "keccak table of the contracts: generator and table builder disagree"
    n_tx = 1 << (log_total - 8)
    tx = synth_tx_witness(n_tx, r, seed=seed + 1)
    n_state = (1 << log_total) - n_steps - (1 << k) - n_tx
    ops, op_flags, *_ = synth_state_ops(n_state, seed=seed + 2)
    # the EVM circuit's bytecode table (sorted by hash, tag, index) read as unrolled bytecodes: one group of rows per hash
    bt = evm["bytecode"].copy()
    n_bt = bt.shape[0]
    starts = [0] + [i for i in range(1, n_bt) if not np.array_equal(bt[i, 0:2], bt[i - 1, 0:2])] + [n_bt]
    offsets = np.array(starts, dtype=np.uint64)
    lengths = (offsets[1:] - offsets[:-1] - np.uint64(1)).astype(np.uint64)
    rows = {"evm": n_steps - 1, "state": n_state, "bytecode": 1 << k, "tx": n_tx}
    assert n_state >= 64
    return {"codes": codes, "evm": evm, "state_ops": (ops, op_flags), "bytecode": (bc_rows, keccak, r),
            "bytecode_unrolled": (bt, offsets, lengths, k), "tx": (tx, r), "rows": rows,
            "meta": dict(meta, n_contracts=n_contracts, code_rows=n_code_rows)}


def synth_super_block(log_total=20, seed=5, keccak_rows_of=None, block_ops=None, n_steps=None):
    """BASELINE config 5 over ONE consistent witness (synth_block.py): the State circuit's rows ARE the EVM trace's RW table
    (re-keyed with the explicit Target -> Tag / key-slot mapping of csrc/state_rekey.hpp and re-sorted ON THE DEVICE by SuperCircuit:
    `state_ops` is None here), the Bytecode
    circuit's rows are the contracts the trace executes (their code hashes the keccak digests of the device-built keccak
    table), plus a Copy circuit (copy events expanded on the device, zk_copy_assign) and an Exp circuit — about 2^log_total
    rows in total: the EVM share is sized so that its RW table (about 2.9 rows per step) fits.  Returns the dict
    SuperCircuit takes; `rows` has the six circuits."""
    from .synth_block import synth_block_codes, synth_block_trace

    assert log_total >= 12
    total = 1 << log_total
    r = (0x1234567 * (seed + 1) ** 7 + 0x9E3779B97F4A7C15) % P
    n_contracts = min(16, 1 << (log_total - 16)) if log_total >= 16 else 2
    seg_len = 640 if log_total >= 16 else 96
    from .synth_block import synth_block_sha3_inputs

    # weight of the SHA3 / CODECOPY / EXP kinds (synth_block._BLOCK_MIX); `block_ops` / `n_steps` override the block's shape (tests of the
    # warm gadgets use a trace where those kinds dominate)
    bw = block_ops if block_ops is not None else (1 if log_total >= 18 else (4 if log_total >= 16 else 12))
    codes = synth_block_codes(seed, seg_len=seg_len, n_contracts=n_contracts, block_ops=bw)
    if keccak_rows_of is None:
        keccak_rows_of = lambda c, rr: engine.keccak_table(c, rr, engine.KECCAK_MODE_CIRCUIT)  # noqa: E731
    # ONE keccak pass over everything the block hashes: the contracts (their code hashes) and the SHA3 steps' inputs
    sha3_inputs = synth_block_sha3_inputs(seed, seg_len=seg_len, n_contracts=n_contracts, block_ops=bw)
    sha3_unique = sorted(set(sha3_inputs))
    all_rows = keccak_rows_of(list(codes) + sha3_unique, r)
    k_data, k_offsets = engine.pack_messages(list(codes) + sha3_unique)  # the same messages in wire form (block.verify_block re-derives the table)
    keccak = np.ascontiguousarray(all_rows[: len(codes)])
    sha3_rows = np.ascontiguousarray(all_rows[len(codes):])
    hashes = [_digest_of_row(keccak[i]) for i in range(len(codes))]
    sha3_digest = {m: _digest_of_row(sha3_rows[i]).to_bytes(32, "big") for i, m in enumerate(sha3_unique)}
    n_code_rows = sum(len(c) + 1 for c in codes)
    k = max(6, int(np.ceil(np.log2(n_code_rows + 1))))
    n_tx = 1 << max(log_total - 8, 2)
    # EVM share: BASELINE configs[4] has the 2^18-step trace inside a ~2^20-row block; smaller blocks scale it (a step brings
    # ~2.9 State rows and, through its SHA3 / CODECOPY / EXP steps, ~0.2 Copy / Exp rows)
    if n_steps is None:
        n_steps = (1 << (log_total - 2)) if log_total >= 16 else int((total - (1 << k) - n_tx) / 4.3)
    evm = synth_block_trace(n_steps, seed=seed, seg_len=seg_len, n_contracts=n_contracts, code_hashes=hashes, block_ops=bw,
                            digest_of=lambda msgs: [sha3_digest[m] for m in msgs], randomness=r)
    meta = evm.pop("meta")
    copy_ev = evm.pop("copy_events")
    exp_rows = evm.pop("exp_rows")
    evm.pop("sha3_inputs")
    evm["keccak"] = sha3_rows  # the EVM circuit's keccak table: the SHA3 steps' rows (execution/sha3.py:31)
    # evm["exp"] (the exp table of the EXP steps) came with the trace; evm["copy"] is produced by the copy assignment (SuperCircuit)
    copy_ev["from_trace"] = True  # the Copy circuit looks up the BLOCK's RW / bytecode tables, not tables of its own
    # State rows: 1 StartOp + one per RW row the State circuit can carry (CallContext field tags above its MAX_FIELD_TAG are left out,
    # csrc/state_rekey.hpp); the ops themselves are derived on the device from evm["rw"] (zk_state_assign_from_rw), not here
    n_state = 1 + int(evm["rw"].shape[0]) - int(np.count_nonzero((evm["rw"][:, 2, 0] == 7) & (evm["rw"][:, 4, 0] > 24)))
    digest = dict(zip(codes, hashes))
    bc_rows, bc_keccak = synth_bytecode_witness(codes, k, r, digest=lambda c: digest[c])
    tx = synth_tx_witness(n_tx, r, seed=seed + 1)
    bt = evm["bytecode"].copy()
    n_bt = bt.shape[0]
    starts = [0] + [i for i in range(1, n_bt) if not np.array_equal(bt[i, 0:2], bt[i - 1, 0:2])] + [n_bt]
    offsets = np.array(starts, dtype=np.uint64)
    lengths = (offsets[1:] - offsets[:-1] - np.uint64(1)).astype(np.uint64)
    n_exp = int(exp_rows.shape[1])
    rows = {"evm": n_steps - 1, "state": n_state, "bytecode": 1 << k, "tx": n_tx, "copy": int(copy_ev["n_rows"]), "exp": n_exp}
    return {"codes": codes, "evm": evm, "state_ops": None, "bytecode": (bc_rows, keccak, r), "bytecode_unrolled": (bt, offsets, lengths, k),
            "tx": (tx, r), "copy_events": copy_ev, "exp_rows": exp_rows, "rows": rows, "keccak_messages": (k_data, k_offsets, len(codes)),
            "meta": dict(meta, n_contracts=n_contracts, code_rows=n_code_rows, state_rows_from_rw_table=True, copy_exp_rows_from_trace=True)}


def _on_device(x):
    return hasattr(x, "is_cuda") and x.is_cuda


class SuperCircuit:
    """Sessions of the four circuits over one witness set; launch() enqueues one pass of each, collect() returns
    ({circuit: Result}, total fail_count, first failing (circuit, row, code))."""

    def __init__(self, parts, device=None, to_device=None, shard=None, state_compact=False, state_fused=False):
        """shard = (rank, world): ONE global block, every circuit's rows cut into `world` contiguous ranges with that
        circuit's halo (distributed.HALO), all tables whole on every rank (BASELINE configs[4]).  The
        witness assignment (State / Bytecode / Copy rows from ops / unrolled codes / copy events) runs over the whole
        block on every rank — it is open-time work, tables and op lists being replicated anyway — and the rank keeps its
        slice.  self.rows then holds the rank's evaluated rows, self.global_rows the block's, self.row_lo each circuit's
        first global row."""
        from . import distributed

        to_dev_fn = to_device if to_device is not None else (lambda x: x)
        dev = lambda x: x if (_on_device(x) or x is None) else to_dev_fn(x)  # noqa: E731  (replicated tensors arrive on the device)
        rank, world = shard if shard is not None else (0, 1)
        self._keep = []
        evm_w = {k: dev(v) for k, v in parts["evm"].items()}
        on_dev = hasattr(evm_w["steps"], "is_cuda")
        odev = evm_w["steps"].device if on_dev else None
        self.state_from_rw = parts.get("state_ops") is None
        # state_compact: the State rows are assigned and evaluated without their limb / byte columns (ZK_OPT_STATE_COMPACT: 15 of the 57
        # cells; the decompositions are derived where the checks use them).  Only for rows derived on the device from the RW table.
        self.state_compact = bool(state_compact) and self.state_from_rw
        nc = 15 if self.state_compact else 57
        # state_fused: no State witness in HBM at all — the State session evaluates op2row's rows where it computes them, from the block's RW
        # table through the sorted order (zk_state_verify_from_rw_open; what zk_block_verify does for a block seen once).  The session keeps
        # the order, the first-access links and the MPT root ranks after its first pass, so a resident pass is the evaluation kernel alone.
        # Whole blocks only (a sharded State range needs the 57-cell rows' halo handling).
        self.state_fused = bool(state_fused) and self.state_from_rw and world == 1 and not self.state_compact
        st_fused = None
        if self.state_fused:
            st_fused = engine.open_state_verify_from_rw(evm_w["rw"], evm_w["rw_flags"], device=device)
            self._keep.append(st_fused)
            res0 = st_fused.run()  # re-keying + sort + MPT roots + first evaluation; raises if the assignment itself fails
            self.assign_ms = res0.kernel_ms
            rows = flags = mpt = None
            res = None
        elif self.state_from_rw:
            # State rows: the block's own RW table re-keyed, sorted and assigned on the device in one session (zk_state_assign_from_rw:
            # Target -> Tag / key slots, LSD radix sort on (tag, id, address, field_tag, storage_key, rw_counter), op2row) — no host step
            ops = op_flags = None
            if on_dev:
                import torch

                n_rw = int(evm_w["rw"].shape[0])
                rows_b = torch.empty(nc * 4 * (n_rw + 1), dtype=torch.int64, device=evm_w["rw"].device)
                flags_b = torch.empty(n_rw + 1, dtype=torch.int32, device=evm_w["rw"].device)
                mpt_b = torch.empty(48 * (n_rw + 1), dtype=torch.int64, device=evm_w["rw"].device)
                with engine.open_state_assign_from_rw(evm_w["rw"], evm_w["rw_flags"], rows_b, flags_b, mpt_b, device=device, compact=self.state_compact) as a:
                    res = a.run()
                    n, m = a.n, a.n_mpt()
                rows, flags, mpt = rows_b[: nc * 4 * n].view(nc, n, 4), flags_b[:n], mpt_b[: 48 * m].view(m, 12, 4)
            else:
                with engine.open_state_assign_from_rw(evm_w["rw"], evm_w["rw_flags"], device=device, compact=self.state_compact) as a:
                    res = a.run()
                    rows, flags, mpt = a.read()
        else:
            ops, op_flags = (dev(a) for a in parts["state_ops"])
        # State rows: assigned on the device from the op list, then evaluated from the same HBM buffers
        if self.state_from_rw or self.state_fused:
            pass
        elif on_dev:
            import torch

            n = int(ops.shape[1])
            rows = torch.empty((57, n, 4), dtype=torch.int64, device=odev)
            flags = torch.empty(n, dtype=torch.int32, device=odev)
            mpt = torch.empty((n, 12, 4), dtype=torch.int64, device=odev)
            with engine.open_state_assign(ops, op_flags, rows, flags, mpt, device=device) as a:
                res = a.run()
                m = a.n_mpt()
            mpt = mpt[:m]
        else:
            with engine.open_state_assign(ops, op_flags, device=device) as a:
                res = a.run()
                rows, flags, mpt = a.read()
        if res is not None:
            if not res.ok:
                raise exception_for_code(res.first_fail_code, f"state witness assignment: op {res.first_fail_row}")
            self.assign_ms = res.kernel_ms
        _, bc_keccak, r = parts["bytecode"]
        # Bytecode circuit rows: assigned on the device from the EVM circuit's own bytecode table
        ub_rows, ub_off, ub_len, k = parts["bytecode_unrolled"]
        d_ub = dev(ub_rows)
        if on_dev:
            import torch

            bc_rows = torch.empty((12, 1 << k, 4), dtype=torch.int64, device=d_ub.device)
            with engine.open_bytecode_assign(d_ub, dev(ub_off), dev(ub_len), k, r, rows_dev=bc_rows, device=device) as a:
                res = a.run()
        else:
            with engine.open_bytecode_assign(d_ub, ub_off, ub_len, k, r, device=device) as a:
                res = a.run()
                bc_rows = a.rows()
        if not res.ok:
            raise exception_for_code(res.first_fail_code, f"bytecode witness assignment: row {res.first_fail_row}")
        dev_rows = bc_rows
        # Copy circuit: the block's copy events expanded on the device (zk_copy_assign): the Copy circuit's rows, the EVM circuit's
        # copy table and (for events that do not come from the trace) the RW rows they imply
        copy_open = None
        if "copy_events" in parts and int(parts["copy_events"]["events"].shape[0]) > 0:  # (a block without copy events has no Copy circuit rows)
            ce = parts["copy_events"]
            ev, fl, da, of = dev(ce["events"]), dev(ce["flags"]), (dev(ce["data"].view(np.int16)) if on_dev else ce["data"]), dev(ce["offsets"])
            if on_dev:
                import torch

                h_ev, h_fl, h_da, h_of = (x.cpu().numpy() if hasattr(x, "is_cuda") else x for x in (ce["events"], ce["flags"], ce["data"], ce["offsets"]))
                n_rows, n_table, n_rw = engine.copy_assign_sizes(h_ev.view(np.uint64), h_fl.view(np.uint32), h_da.view(np.uint16), h_of.view(np.uint64), device)
                c_rows = torch.empty((20, n_rows, 4), dtype=torch.int64, device=odev)
                c_rf = torch.empty(n_rows, dtype=torch.int32, device=odev)
                c_table = torch.empty((n_table, 14, 4), dtype=torch.int64, device=odev)
                c_rw = torch.empty((n_rw, 14, 4), dtype=torch.int64, device=odev)
                c_rwf = torch.empty(n_rw, dtype=torch.int32, device=odev)
                with engine.open_copy_assign(ev, fl, da, of, ce["r"], c_rows, c_rf, c_table, c_rw, c_rwf, device=device) as a:
                    res = a.run()
                if not res.ok:
                    raise exception_for_code(res.first_fail_code, f"copy witness assignment: event {res.first_fail_row}")
            else:
                with engine.open_copy_assign(ce["events"], ce["flags"], ce["data"], ce["offsets"], ce["r"], device=device) as a:
                    res = a.run()
                    if not res.ok:
                        raise exception_for_code(res.first_fail_code, f"copy witness assignment: event {res.first_fail_row}")
                    c_rows, c_rf, c_table, c_rw, c_rwf = a.read()
            copy_open = (c_rows, c_rf, c_table, c_rw, c_rwf, ce)
        tx, r_tx = parts["tx"]
        if copy_open is not None and copy_open[5].get("from_trace"):
            evm_w["copy"] = copy_open[2]  # the copy table the SHA3 / CODECOPY steps look up = the table of the very events the Copy circuit checks

        tx_w = {k: dev(v) for k, v in tx.items()}
        self.global_rows = {"evm": int(evm_w["steps"].shape[0]) - 1, "state": st_fused.n if st_fused is not None else int(rows.shape[1]), "bytecode": int(dev_rows.shape[1]),
                            "tx": int(tx_w["bytes"].shape[0])}
        self.row_lo = {k: 0 for k in self.global_rows}
        ranges = {}
        if world > 1:
            evm_w, _, _, self.row_lo["evm"] = distributed.shard_evm(evm_w, rank, world)
            rows, flags, lo_, hi_, self.row_lo["state"] = distributed.shard_rows(rows, flags, rank, world, "state")
            ranges["state"] = (lo_, hi_)
            dev_rows, _, lo_, hi_, self.row_lo["bytecode"] = distributed.shard_rows(dev_rows, None, rank, world, "bytecode")
            ranges["bytecode"] = (lo_, hi_)
            tx_w, self.row_lo["tx"] = distributed.shard_units(tx_w, rank, world)
        self.sessions = {
            "evm": engine.open_evm(evm_w, device=device, side_stream=True),
            "state": st_fused if st_fused is not None else engine.open_state(rows, flags, mpt, device=device, compact=self.state_compact),
            "bytecode": engine.open_bytecode(dev_rows, dev(bc_keccak), r, device=device),
            "tx": engine.open_sign(tx_w, r_tx, False, device=device),
        }
        if copy_open is not None:
            c_rows, c_rf, c_table, c_rw, c_rwf, ce = copy_open
            self.global_rows["copy"], self.row_lo["copy"] = int(c_rows.shape[1]), 0
            if world > 1:
                c_rows, c_rf, lo_, hi_, self.row_lo["copy"] = distributed.shard_rows(c_rows, c_rf, rank, world, "copy")
                ranges["copy"] = (lo_, hi_)
            if ce.get("from_trace"):  # the events are the trace's: the Copy circuit looks up the block's own RW / bytecode / tx tables
                self.sessions["copy"] = engine.open_copy(c_rows, c_rf, ce["r"], evm_w["rw"], evm_w["rw_flags"], evm_w["bytecode"], evm_w["tx"],
                                                         evm_w["tx_flags"], device=device)
            else:
                self.sessions["copy"] = engine.open_copy(c_rows, c_rf, ce["r"], c_rw, c_rwf, dev(ce["bytecode"]), dev(ce["tx"]), dev(ce["tx_flags"]),
                                                         device=device)
        if "exp_rows" in parts and int(parts["exp_rows"].shape[1]) > 0:
            e_rows = dev(parts["exp_rows"])
            self.global_rows["exp"], self.row_lo["exp"] = int(e_rows.shape[1]), 0
            if world > 1:
                e_rows, _, lo_, hi_, self.row_lo["exp"] = distributed.shard_rows(e_rows, None, rank, world, "exp")
                ranges["exp"] = (lo_, hi_)
            self.sessions["exp"] = engine.open_exp(e_rows, device=device)
        for k, (lo_, hi_) in ranges.items():
            self.sessions[k].set_range(lo_, hi_)
        self.rows = {k: (ranges[k][1] - ranges[k][0] if k in ranges else s.n) for k, s in self.sessions.items()}
        self.eval_lo = {k: ranges.get(k, (0, 0))[0] for k in self.sessions}
        # one HIP stream per circuit: the kernels are independent and bound by different things (the State kernel streams
        # HBM, the EVM kernel is latency / issue bound), so their passes overlap on the device
        self._streams = None
        self._launch_order = None  # results / first-failure reporting keep the sessions' own order whatever the launch order is
        if on_dev:
            import torch

            torch.cuda.synchronize()  # witness uploads / open-time packing ran on the stream the sessions were opened on
            # The four small circuits (Exp, Tx, Copy, Bytecode: 1 k - 131 k rows, 12 - 50 us alone) run on high-priority streams and are
            # launched first, then the EVM chain, the State launch last: parked behind the State kernel's wavefronts they used to take
            # 100 - 240 us each and stretch the pass (round 5, four alternating runs: 0.354 -> 0.331 ms; priority alone 0.341, order alone
            # no gain).  ZK_SUPER_PRIO=0 / ZK_SUPER_ORDER=<names> restore / change it for A/B runs.
            prio = os.environ.get("ZK_SUPER_PRIO", "1") == "1"
            hi = tuple(os.environ.get("ZK_SUPER_PRIO_SET", "exp,tx,copy,bytecode").split(","))  # which sessions get the high-priority streams
            self._streams = {k: (torch.cuda.Stream(priority=-1) if prio and k in hi else torch.cuda.Stream()) for k in self.sessions}
            order = os.environ.get("ZK_SUPER_ORDER", "exp,tx,copy,bytecode,evm,state" if prio else "").split(",")
            self._launch_order = [k for k in order if k in self.sessions] + [k for k in self.sessions if k not in order]
            for k, s in self.sessions.items():
                s.set_stream(self._streams[k])

    def launch(self):
        for k in (self._launch_order or self.sessions):
            self.sessions[k].launch()

    def collect(self):
        results = {k: s.collect() for k, s in self.sessions.items()}
        total = sum(r.fail_count for r in results.values())
        # first failure: (circuit, row of the GLOBAL block, code) — a sharded session reports rows of its local slice
        first = next(((k, r.first_fail_row - self.eval_lo[k] + self.row_lo[k], r.first_fail_code) for k, r in results.items() if not r.ok), None)
        return results, total, first

    def close(self):
        for s in self.sessions.values():
            s.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
