"""One-shot verification of a block (BASELINE config 5) from its raw, device-resident inputs: everything the Super circuit derives
on the device — the keccak table, the Bytecode / Copy / State witness assignments, the six sessions' opens — plus one evaluation
pass of each circuit, its collect and the closes, as ONE call.  `SuperCircuit` (super_circuit.py) is the resident form of the
same thing (open once, many passes); this is what a verifier pays for a block it has not seen before.

The reference has no super-circuit driver.  The inputs are what the reference's own constructors take:
  * the EVM circuit's tables (`Tables`, evm_circuit/table.py:583-625) and step rows, as `zk_evm_tables` wires;
  * the byte strings the block hashes — the contracts (`KeccakCircuit.add`, evm_circuit/typing.py:854-865) and the SHA3 inputs;
  * the copy events (`CopyCircuit.copy`, typing.py:1010-1091), the Exp circuit's rows, the Tx units.
Derived here, on the device: keccak rows (zk_keccak_*), Bytecode rows (zk_bytecode_assign_*), Copy rows + copy table
(zk_copy_assign_*), State rows from the RW table (zk_state_assign_from_rw_open: re-keying + lexicographic sort + op2row).

Four host threads drive four independent chains, each on its own HIP stream (the C ABI is re-entrant per session; ctypes releases
the GIL inside every call): the State chain (the longest), keccak -> Bytecode, copy assignment -> EVM + Copy, and Exp + Tx.
"""
import ctypes
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, engine
from .errors import exception_for_code


def stage_block(parts, to_device):
    """parts (super_circuit.synth_super_block) -> the device-resident inputs verify_block takes"""
    dev = to_device

    def cell(r):  # the randomness as a device-resident cell: an int would be uploaded per open, on the default stream (a device-wide wait)
        return dev(np.frombuffer(int(r).to_bytes(32, "little"), dtype="<u8").copy())

    b = {"evm": {k: dev(v) for k, v in parts["evm"].items() if k not in ("keccak", "copy")}}
    data, offsets, n_codes = parts["keccak_messages"]
    b["keccak"] = (dev(data), dev(offsets), int(n_codes), int(offsets.shape[0]) - 1)
    ub_rows, ub_off, ub_len, k = parts["bytecode_unrolled"]
    b["bytecode"] = (dev(ub_rows), dev(ub_off), dev(ub_len), int(k))
    b["r"] = cell(parts["bytecode"][2])
    ce = parts["copy_events"]
    b["copy_events"] = (dev(ce["events"]), dev(ce["flags"]), dev(ce["data"].view(np.int16)), dev(ce["offsets"]), cell(ce["r"]))
    b["copy_sizes"] = engine.copy_assign_sizes(ce["events"], ce["flags"], ce["data"], ce["offsets"])
    b["exp_rows"] = dev(parts["exp_rows"])
    tx, r_tx = parts["tx"]
    b["tx"] = ({k: dev(v) for k, v in tx.items()}, cell(r_tx))
    return b


def native_block(b):
    """the zk_block struct of a staged block (stage_block): built once, every pointer is a device pointer of a tensor `b` keeps alive"""
    def p(x, n=1):
        return ctypes.c_void_p(x.data_ptr()).value if (x is not None and n) else None

    def rows(x):
        return 0 if x is None else int(x.shape[0])

    e = b["evm"]
    evm = _lib.ZkEvmTables(
        p(e["steps"]), rows(e["steps"]), p(e.get("rw"), rows(e.get("rw"))), p(e.get("rw_flags"), rows(e.get("rw"))), rows(e.get("rw")),
        p(e.get("bytecode"), rows(e.get("bytecode"))), rows(e.get("bytecode")),
        p(e.get("tx"), rows(e.get("tx"))), p(e.get("tx_flags"), rows(e.get("tx"))), rows(e.get("tx")),
        p(e.get("block"), rows(e.get("block"))), p(e.get("block_flags"), rows(e.get("block"))), rows(e.get("block")),
        0, 0, None, 0, None, 0, p(e.get("exp"), rows(e.get("exp"))), rows(e.get("exp")),
        None, None, None, 0, None, 0, None, 0, 0, 0)
    data, offsets, n_codes, n_msgs = b["keccak"]
    ub_rows, ub_off, ub_len, k = b["bytecode"]
    ev, fl, da, of, r_copy = b["copy_events"]
    ce = _lib.ZkCopyEvents(p(ev), p(fl), rows(ev), p(da, rows(da)), p(of), p(r_copy))
    tx_w, r_tx = b["tx"]
    n_tx = rows(tx_w.get("bytes"))
    tx = _lib.ZkSignUnits(p(tx_w.get("bytes"), n_tx), p(tx_w.get("cells"), n_tx), p(tx_w.get("meta"), n_tx), n_tx, p(r_tx),
                          p(tx_w.get("keccak"), rows(tx_w.get("keccak"))), rows(tx_w.get("keccak")),
                          p(tx_w.get("tx_rows"), rows(tx_w.get("tx_rows"))), p(tx_w.get("tx_flags"), rows(tx_w.get("tx_rows"))), rows(tx_w.get("tx_rows")), 0)
    ex = b["exp_rows"]
    n_exp = 0 if ex is None else int(ex.shape[1])
    return _lib.ZkBlock(evm, p(data, int(data.shape[0])), int(data.shape[0]), p(offsets), int(n_codes), int(n_msgs), p(b["r"]),
                        p(ub_off), p(ub_len), int(ub_len.shape[0]), int(k), 0, ce, p(ex, n_exp), n_exp, tx)


def verify_block_native(b, device=0, state_compact=False, state_rows=False):
    """zk_block_verify (include/zkevm_hip.h): the same chains as BlockVerifier.verify, driven by four threads inside the library —
    -> ({circuit: Result}, total fail count, chain_ms[10]: ends, starts, all ended, return).  The State rows are evaluated where they are computed
    (zk_state_verify_from_rw_open) unless state_rows (57-cell witness written and read back) or state_compact (15-cell) asks for them."""
    lib = _lib.init(device)
    blk = b.get("_native")
    if blk is None:
        blk = b["_native"] = native_block(b)
    res = (_lib.ZkResult * 6)()
    ends = (ctypes.c_double * 10)()  # chain ends [4], chain starts [4], all chains ended, return (host ms from the call)
    opts = _lib.OPT_DEVICE_PTRS | (_lib.OPT_STATE_COMPACT if state_compact else 0) | (_lib.OPT_BLOCK_STATE_ROWS if state_rows else 0)
    _lib.check(lib.zk_block_verify(ctypes.byref(blk), opts, res, ends), "zk_block_verify", lib)
    results = {name: engine.Result(res[i]) for i, name in enumerate(_lib.BLOCK_CIRCUITS) if res[i].rows_evaluated or res[i].launches}
    return results, sum(r.fail_count for r in results.values()), list(ends)


class BlockVerifier:
    """Persistent worker threads + streams + output buffers for verify(): nothing is allocated or created per block."""

    CHAINS = ("state", "keccak", "copy", "rest")

    def __init__(self, device=0, state_compact=False):
        import torch

        self.state_compact = bool(state_compact)  # ZK_OPT_STATE_COMPACT for the State chain (include/zkevm_hip.h)
        self.device = device
        self.lib = _lib.init(device)
        self.pool = ThreadPoolExecutor(max_workers=len(self.CHAINS))
        # the State chain is the longest and floods the device with HBM-bound kernels; the chains the EVM circuit waits for (keccak
        # table, copy table) are short, latency-bound kernels: they run on high-priority streams
        self.streams = {c: torch.cuda.Stream(device=device, priority=0 if c == "state" else -1) for c in self.CHAINS}
        self.keccak_ready = torch.cuda.Event()
        import os

        self.gate_state = os.environ.get("ZK_BLOCK_GATE_STATE", "0") == "1"
        self._bound = threading.local()
        self._bufs = {}

    def _bind(self, chain):
        """first use of a worker thread: select the device and make this chain's stream the one its sessions open on"""
        if getattr(self._bound, "chain", None) != chain:
            _lib.init(self.device)
            _lib.check(self.lib.zk_set_stream(ctypes.c_void_p(self.streams[chain].cuda_stream)), "zk_set_stream", self.lib)
            self._bound.chain = chain

    def _buf(self, name, n, dtype):
        import torch

        t = self._bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=f"cuda:{self.device}")
            self._bufs[name] = t
        return t

    def close(self):
        self.pool.shutdown()

    def verify(self, b):
        """-> ({circuit: Result}, total fail count).  Every call opens, evaluates and closes everything."""
        import torch

        i64, i32 = torch.int64, torch.int32
        evm_in = b["evm"]
        done_keccak, done_copy, keccak_enqueued = threading.Event(), threading.Event(), threading.Event()
        shared, results = {}, {}
        import time

        t_start = time.perf_counter()
        trace = self.trace = []

        def mark(chain, what):
            trace.append((chain, what, (time.perf_counter() - t_start) * 1e3))

        def chain_state():
            self._bind("state")
            mark("state", "start")
            n_rw = int(evm_in["rw"].shape[0])
            nc = 15 if self.state_compact else 57
            rows_b, flags_b, mpt_b = self._buf("st_rows", nc * 4 * (n_rw + 1), i64), self._buf("st_flags", n_rw + 1, i32), self._buf("st_mpt", 48 * (n_rw + 1), i64)
            with engine.open_state_assign_from_rw(evm_in["rw"], evm_in["rw_flags"], rows_b, flags_b, mpt_b, device=self.device, compact=self.state_compact) as a:
                mark("state", "assign opened (class scan + plan)")
                if self.gate_state:  # the keccak pass (latency-bound, a handful of wavefronts) runs 4x slower beside the State chain's
                    done_keccak.wait()  # HBM-bound kernels: let it finish first (the EVM chain waits for its table)
                    mark("state", "keccak done: go")
                res = a.run()
                mark("state", "assign done (sort + rows)")
                n, m = a.n, a.n_mpt()
            if not res.ok:
                raise exception_for_code(res.first_fail_code, f"state witness assignment: {res.first_fail_row}")
            mark("state", "assign closed")
            with engine.open_state(rows_b[: nc * 4 * n].view(nc, n, 4), flags_b[:n], mpt_b[: 48 * m].view(m, 12, 4), device=self.device, compact=self.state_compact) as s:
                mark("state", "state opened")
                results["state"] = s.run()
                mark("state", "state pass done")
            mark("state", "end")

        def chain_keccak():
            self._bind("keccak")
            mark("keccak", "start")
            data, offsets, n_codes, n_msgs = b["keccak"]
            rows = self._buf("keccak_rows", n_msgs * 20, i64)[: n_msgs * 20].view(n_msgs, 5, 4)
            ub_rows, ub_off, ub_len, k = b["bytecode"]
            bc_rows = self._buf("bc_rows", 12 * (1 << k) * 4, i64)[: 12 * (1 << k) * 4].view(12, 1 << k, 4)
            # the Bytecode assignment does not depend on the digests: opened first (its open reads the offsets back: a stream
            # synchronisation that must not find the keccak pass in front of it), enqueued behind the keccak pass, collected after it
            ba = engine.open_bytecode_assign(ub_rows, ub_off, ub_len, k, b["r"], rows_dev=bc_rows, device=self.device)
            mark("keccak", "bytecode assign opened")
            with engine.open_keccak(data, offsets, b["r"], engine.KECCAK_MODE_CIRCUIT, rows_dev=rows, device=self.device) as ks:
                mark("keccak", "keccak opened")
                ks.launch()
                # the EVM chain orders itself behind the keccak pass on the DEVICE (event), not behind this thread's collect
                shared["keccak_codes"], shared["keccak_sha3"] = rows[:n_codes], rows[n_codes:]
                self.keccak_ready.record(self.streams["keccak"])
                keccak_enqueued.set()
                ba.launch()
                mark("keccak", "keccak + bytecode assign launched")
                res = ks.collect()
                mark("keccak", "keccak collected")
            if not res.ok:
                raise exception_for_code(res.first_fail_code, f"keccak table: message {res.first_fail_row}")
            done_keccak.set()
            with ba:
                res = ba.collect()
            if not res.ok:
                raise exception_for_code(res.first_fail_code, f"bytecode witness assignment: row {res.first_fail_row}")
            mark("keccak", "bytecode assign collected")
            with engine.open_bytecode(bc_rows, shared["keccak_codes"], b["r"], device=self.device) as s:
                results["bytecode"] = s.run()
            mark("keccak", "end")

        def chain_copy():
            self._bind("copy")
            mark("copy", "start")
            ev, fl, da, of, r = b["copy_events"]
            n_rows, n_table, n_rw = b["copy_sizes"]
            c_rows = self._buf("c_rows", 20 * n_rows * 4, i64)[: 20 * n_rows * 4].view(20, n_rows, 4)
            c_rf = self._buf("c_rf", n_rows, i32)[:n_rows]
            c_table = self._buf("c_table", n_table * 56, i64)[: n_table * 56].view(n_table, 14, 4)
            c_rw = self._buf("c_rw", n_rw * 56, i64)[: n_rw * 56].view(n_rw, 14, 4)
            c_rwf = self._buf("c_rwf", n_rw, i32)[:n_rw]
            with engine.open_copy_assign(ev, fl, da, of, r, c_rows, c_rf, c_table, c_rw, c_rwf, device=self.device) as a:
                mark("copy", "copy assign opened")
                res = a.run()
                mark("copy", "copy assign done")
            if not res.ok:
                raise exception_for_code(res.first_fail_code, f"copy witness assignment: event {res.first_fail_row}")
            copy_s = engine.open_copy(c_rows, c_rf, r, evm_in["rw"], evm_in["rw_flags"], evm_in["bytecode"], evm_in["tx"], evm_in["tx_flags"], device=self.device)
            mark("copy", "copy circuit opened")
            copy_s.launch()
            mark("copy", "copy circuit launched")
            keccak_enqueued.wait()
            self.streams["copy"].wait_event(self.keccak_ready)
            mark("copy", "keccak enqueued")
            w = dict(evm_in, copy=c_table, keccak=shared["keccak_sha3"])
            with engine.open_evm(w, device=self.device, single_pass=True) as s:
                mark("copy", "evm opened")
                results["evm"] = s.run()
                mark("copy", "evm pass done")
            with copy_s:
                results["copy"] = copy_s.collect()
            done_copy.set()
            mark("copy", "end")

        def chain_rest():
            self._bind("rest")
            mark("rest", "start")
            tx_w, r_tx = b["tx"]
            ex = engine.open_exp(b["exp_rows"], device=self.device)
            ex.launch()
            mark("rest", "exp launched")
            with engine.open_sign(tx_w, r_tx, False, device=self.device) as s:
                results["tx"] = s.run()
            with ex:
                results["exp"] = ex.collect()
            mark("rest", "end")

        futs = [self.pool.submit(f) for f in (chain_state, chain_keccak, chain_copy, chain_rest)]
        err = None
        for f in futs:
            try:
                f.result()
            except Exception as e:  # noqa: BLE001 — the first failure is re-raised once every chain has ended (no thread is left waiting)
                err = err or e
                keccak_enqueued.set()
                done_keccak.set()
        if err is not None:
            raise err
        return results, sum(r.fail_count for r in results.values())
