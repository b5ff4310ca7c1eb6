"""Synthetic EVM-circuit witness (BASELINE config 3): a mixed-opcode execution trace.

The trace is a chain of call contexts.  Each segment executes one of `n_contracts` straight-line
contracts (<= `seg_len` opcodes, code size < 24,576 B) and ends with a non-root STOP that
restores the next segment's context (instruction.py:292-363), so program counters restart and the
stack pointer stays inside [0, 1024] without any state the reference does not implement
(JUMPDEST/DUP/SWAP/PC are unimplemented there).

Every step satisfies its gadget's constraints (the EVM circuit checks each step against the RW /
bytecode rows it looks up and the step-state transition; cross-step stack consistency is the
State circuit's job and is not modelled).  Operands are uniform 256-bit words with ~10 % drawn
from the reference tests' NASTY values (tests/common.py:23-45).
"""
import random

import numpy as np

from . import evm_tables as T
from .wire import rows_to_colmajor, rows_to_rowmajor

ES, OP, TG, CC = T.ExecutionState, T.Opcode, T.Target, T.CallContextFieldTag
M256 = (1 << 256) - 1
M128 = (1 << 128) - 1
FR_P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
NASTY = [0, 1, 255, 256, 260, 513, 65535, 65536, M256, M256 - 1, 1 << 255, (1 << 255) - 1, 1 << 128, M128]

# (weight, kind) — BASELINE config-3 mix, restricted to states the engine evaluates
_MIX = [
    (30, "PUSH"), (12, "ADDSUB"), (10, "MULDIVMOD"), (5, "CMP"), (3, "SCMP"), (4, "BITWISE"), (2, "NOT"),
    (5, "ISZERO"), (3, "BYTE"), (2, "SIGNEXTEND"), (5, "SHIFT"), (1.5, "ADDMOD"), (1.5, "MULMOD"), (8, "POP"),
    (4, "MEMORY"), (1, "SLOAD"), (1, "SSTORE"), (2, "READER"),
]
_PUSH_SIZES = [1, 2, 4, 8, 20, 32]
_READERS = ["ADDRESS", "CALLER", "CALLVALUE", "CALLDATASIZE", "GAS", "MSIZE", "CODESIZE"]


def _weq(a, b):
    """Instruction.is_equal_word (instruction.py:409-414): zero test on the *sum* of the lo and hi
    differences in the field — (1, 0) and (0, 1) compare equal there; a valid trace must follow it."""
    return ((a & M128) - (b & M128) + (a >> 128) - (b >> 128)) % FR_P == 0


def _signed(x):
    return x - (1 << 256) if x >> 255 else x


class _Contract:
    def __init__(self, rng, n_ops, mix=None):
        self.ops = []  # (opcode name, pc, push bytes)
        code = bytearray()
        mix = mix or _MIX
        weights = [w for w, _ in mix]
        kinds = [k for _, k in mix]
        for _ in range(n_ops):
            kind = rng.choices(kinds, weights)[0]
            data = b""
            if kind.startswith("PUSH") and kind != "PUSH":
                n = int(kind[4:])
                name, data = kind, bytes(rng.getrandbits(8) for _ in range(n))
            elif kind == "PUSH":
                n = rng.choice(_PUSH_SIZES)
                name, data = f"PUSH{n}", bytes(rng.getrandbits(8) for _ in range(n))
            elif kind == "ADDSUB":
                name = rng.choice(["ADD", "SUB"])
            elif kind == "MULDIVMOD":
                name = rng.choice(["MUL", "DIV", "MOD"])
            elif kind == "CMP":
                name = rng.choice(["LT", "GT", "EQ"])
            elif kind == "SCMP":
                name = rng.choice(["SLT", "SGT"])
            elif kind == "BITWISE":
                name = rng.choice(["AND", "OR", "XOR"])
            elif kind == "SHIFT":
                name = rng.choice(["SHL", "SHR", "SAR"])
            elif kind == "SDIVSMOD":
                name = rng.choice(["SDIV", "SMOD"])
            elif kind == "MEMORY":
                name = rng.choice(["MLOAD", "MSTORE", "MSTORE8"])
            elif kind == "READER":
                name = rng.choice(_READERS)
            else:
                name = kind
            self.ops.append((name, len(code), data))
            code.append(int(OP[name]))
            code += data
        self.ops.append(("STOP", len(code), b""))
        code.append(0)
        self.code = bytes(code)
        h = rng.getrandbits(256)
        self.hash = (h & M128, h >> 128)

    def table_rows(self):
        lo, hi = self.hash
        rows = [[lo, hi, 1, 0, 0, len(self.code)]]
        is_data = bytearray(len(self.code))
        for name, pc, data in self.ops:
            for k in range(len(data)):
                is_data[pc + 1 + k] = 1
        for idx, b in enumerate(self.code):
            rows.append([lo, hi, 2, idx, 0 if is_data[idx] else 1, b])
        return rows


def _word(rng):
    if rng.random() < 0.1:
        return rng.choice(NASTY)
    return rng.getrandbits(256)


_STATE_OF = {}
for _st, _ops in T.RESPONSIBLE.items():
    for _o in _ops:
        _STATE_OF[_o] = int(ES[_st])


def synth_evm_codes(seed=3, seg_len=640, n_contracts=16, mix=None):
    """The contract byte strings synth_evm_trace(seed, ...) executes (they depend on the seed and the mix only)."""
    rng = random.Random(seed)
    return [_Contract(rng, seg_len - 1, mix).code for _ in range(n_contracts)]


def synth_evm_trace(n_steps, seed=3, seg_len=640, n_contracts=16, as_wire=True, mix=None, code_hashes=None):
    """Build an n_steps-step trace (n_steps - 1 evaluated pairs).  Returns a dict with the wire
    arrays `steps, rw, rw_flags, bytecode, tx, tx_flags, block, block_flags` plus `meta`.
    code_hashes: optional list of 256-bit ints, the code hash of each contract (synth_evm_codes order) — e.g. their
    keccak-256 digests when the bytecode / keccak tables of the same contracts are evaluated next to the trace
    (super_circuit.py).  Default: a random 256-bit value (the EVM circuit only matches hashes, it never hashes code)."""
    rng = random.Random(seed)
    contracts = [_Contract(rng, seg_len - 1, mix) for _ in range(n_contracts)]
    if code_hashes is not None:
        assert len(code_hashes) == n_contracts
        for c, h in zip(contracts, code_hashes):
            c.hash = (h & M128, h >> 128)
    steps, rw, rw_flags = [], [], []
    looked_up_cells = 0  # algorithmic-bytes accounting: cells of rows the step pairs look up
    tx_id = 1
    callee = rng.getrandbits(160)
    rwc = 1
    seg = 0
    gas_left, rev_wc = 10**9, 0
    SP0, GAS_REFILL = (400 if mix is None else 512), 10**7

    def add_rw(rw_, tag, id_=0, addr=0, ft=0, key=0, value=0, prev=0, aux=0, vw=True, pw=True):
        nonlocal rwc
        rw.append((rwc, rw_, tag, id_, addr, ft, key & M128, key >> 128, value & M128, value >> 128,
                   prev & M128, prev >> 128, aux & M128, aux >> 128))
        rw_flags.append((1 if vw else 0) | (2 if pw else 0))
        rwc += 1

    while len(steps) < n_steps:
        C = contracts[seg % n_contracts]
        call_id = 1 + seg
        sp, mws = SP0, 0
        for name, pc, data in C.ops:
            if len(steps) >= n_steps:
                break
            op = int(OP[name])
            state = int(ES.STOP) if name == "STOP" else _STATE_OF[name]
            steps.append([state, rwc, call_id, 0, 0, C.hash[0], C.hash[1], pc, sp, gas_left, mws, rev_wc, 0])
            gas = T.OPCODES[name][1]
            n_bc = 1  # bytecode rows looked up
            rw0 = rwc

            def pop(v, off):
                add_rw(0, TG.Stack, call_id, sp + off, value=v)

            def push(v, off):
                add_rw(1, TG.Stack, call_id, sp + off, value=v)

            def cc(tag, v, word=False, w=0, cid=None):
                add_rw(w, TG.CallContext, call_id if cid is None else cid, int(tag), value=v, vw=word)

            if name.startswith("PUSH"):
                v = int.from_bytes(data, "big")
                push(v, -1)
                sp -= 1
                n_bc += 1 + len(data)
            elif name in ("ADD", "SUB"):
                a, b = _word(rng), _word(rng)
                c = (a + b) & M256 if name == "ADD" else (a - b) & M256
                pop(a, 0); pop(b, 1); push(c, 1)
                sp += 1
            elif name in ("MUL", "DIV", "MOD"):
                a, b = _word(rng), _word(rng)
                if name != "MUL" and rng.random() < 0.3:
                    b >>= rng.randrange(0, 200)
                c = (a * b) & M256 if name == "MUL" else (0 if b == 0 else (a // b if name == "DIV" else a % b))
                pop(a, 0); pop(b, 1); push(c, 1)
                sp += 1
            elif name in ("LT", "GT", "EQ", "SLT", "SGT"):
                a, b = _word(rng), _word(rng)
                if rng.random() < 0.15:
                    b = a
                c = {"LT": a < b, "GT": a > b, "EQ": a == b, "SLT": _signed(a) < _signed(b),
                     "SGT": _signed(a) > _signed(b)}[name]
                pop(a, 0); pop(b, 1); push(int(c), 1)
                sp += 1
            elif name in ("AND", "OR", "XOR"):
                a, b = _word(rng), _word(rng)
                c = a & b if name == "AND" else (a | b if name == "OR" else a ^ b)
                pop(a, 0); pop(b, 1); push(c, 1)
                sp += 1
            elif name == "NOT":
                a = _word(rng)
                pop(a, 0); push(a ^ M256, 0)
            elif name == "ISZERO":
                a = 0 if rng.random() < 0.3 else _word(rng)
                pop(a, 0); push(int(a == 0), 0)
            elif name == "BYTE":
                i_ = rng.randrange(0, 40) if rng.random() < 0.9 else _word(rng)
                x = _word(rng)
                c = (x >> (8 * (31 - i_))) & 0xFF if i_ < 32 else 0
                pop(i_, 0); pop(x, 1); push(c, 1)
                sp += 1
            elif name == "SIGNEXTEND":
                # index >= 256 with a low byte < 31 makes the reference look up (0, sign_byte) with the
                # sign byte of an unselected byte (signextend.py:17-52) — unsatisfiable when that byte
                # is negative; valid traces keep the low byte of such indices >= 31.
                i_ = rng.randrange(0, 36) if rng.random() < 0.9 else ((rng.getrandbits(248) << 8) | rng.randrange(31, 256))
                x = _word(rng)
                if i_ < 31:
                    bit = 8 * i_ + 7
                    m = (1 << (bit + 1)) - 1
                    c = (x | (M256 ^ m)) if (x >> bit) & 1 else (x & m)
                else:
                    c = x
                pop(i_, 0); pop(x, 1); push(c, 1)
                sp += 1
            elif name in ("SHL", "SHR"):
                s_ = rng.randrange(0, 256) if rng.random() < 0.85 else rng.choice([256, 257, 1 << 64, M256])
                x = _word(rng)
                c = 0 if s_ >= 256 else ((x << s_) & M256 if name == "SHL" else x >> s_)
                pop(s_, 0); pop(x, 1); push(c, 1)
                sp += 1
            elif name == "SAR":
                s_ = rng.randrange(0, 256) if rng.random() < 0.85 else rng.choice([256, 257, 1 << 64, M256])
                x = _word(rng)
                c = (_signed(x) >> min(s_, 256)) & M256
                pop(s_, 0); pop(x, 1); push(c, 1)
                sp += 1
            elif name in ("SDIV", "SMOD"):
                a, b = _word(rng), _word(rng)
                if rng.random() < 0.3:
                    b = (_signed(b) >> rng.randrange(0, 250)) & M256
                sa, sb_ = _signed(a), _signed(b)
                if b == 0:
                    c = 0
                else:
                    q_ = abs(sa) // abs(sb_)
                    r_ = abs(sa) % abs(sb_)
                    c = (q_ if (sa < 0) == (sb_ < 0) else -q_) if name == "SDIV" else (-r_ if sa < 0 else r_)
                    c &= M256
                pop(a, 0); pop(b, 1); push(c, 1)
                sp += 1
            elif name in ("ADDMOD", "MULMOD"):
                a, b = _word(rng), _word(rng)
                n_ = 0 if rng.random() < 0.1 else _word(rng)
                if name == "ADDMOD":
                    n_ %= FR_P  # addmod.py:61 compares the pushed value with its residue mod p
                c = 0 if n_ == 0 else ((a + b) % n_ if name == "ADDMOD" else (a * b) % n_)
                pop(a, 0); pop(b, 1); pop(n_, 2); push(c, 2)
                sp += 2
            elif name == "POP":
                pop(_word(rng), 0)
                sp += 1
            elif name in ("MLOAD", "MSTORE", "MSTORE8"):
                addr = rng.randrange(0, 4096)
                v = _word(rng)
                pop(addr, 0)
                if name == "MLOAD":
                    push(v, 0)
                else:
                    pop(v, 1)
                    sp += 2
                vb = v.to_bytes(32, "little")
                if name == "MSTORE8":
                    add_rw(1, TG.Memory, call_id, addr, value=vb[0], vw=False)
                    length = addr + 1
                else:
                    w_ = 0 if name == "MLOAD" else 1
                    for k in range(32):
                        add_rw(w_, TG.Memory, call_id, addr + k, value=vb[31 - k], vw=False)
                    length = addr + 32
                # reference's memory_expansion(offset=curr.memory_word_size, length) (memory.py:23-26)
                mem_size = (length + mws + 31) // 32
                nxt = max(mws, mem_size)
                gas += (nxt * nxt // 512 + 3 * nxt) - (mws * mws // 512 + 3 * mws)
                mws = nxt
            elif name == "SLOAD":
                key, val = _word(rng), _word(rng)
                warm = rng.randrange(2)
                cc(CC.TxId, tx_id); cc(CC.RwCounterEndOfReversion, 0); cc(CC.IsPersistent, 1)
                cc(CC.CalleeAddress, callee, word=True)
                pop(key, 0)
                add_rw(0, TG.AccountStorage, tx_id, callee, key=key, value=val, prev=val, aux=_word(rng))
                push(val, 0)
                add_rw(1, TG.TxAccessListAccountStorage, tx_id, callee, key=key, value=1, prev=warm, vw=False, pw=False)
                gas += 100 if warm else 2100
                rev_wc += 1
            elif name == "SSTORE":
                key = _word(rng)
                orig = rng.choice([0, _word(rng)])
                prev = rng.choice([orig, 0, _word(rng)])
                val = rng.choice([prev, orig, 0, _word(rng)])
                warm = rng.randrange(2)
                refund_prev = rng.randrange(4800, 10**6)
                cc(CC.TxId, tx_id); cc(CC.IsStatic, 0); cc(CC.RwCounterEndOfReversion, 0); cc(CC.IsPersistent, 1)
                cc(CC.CalleeAddress, callee, word=True)
                pop(key, 0); pop(val, 1)
                add_rw(1, TG.AccountStorage, tx_id, callee, key=key, value=val, prev=prev, aux=orig)
                add_rw(1, TG.TxAccessListAccountStorage, tx_id, callee, key=key, value=1, prev=warm, vw=False, pw=False)
                # EIP-2200/3529 refund + gas (storage.py:84-135)
                CLR, SET, RST, SLD = 4800, 20000, 2900, 100
                if _weq(prev, val):
                    refund = refund_prev
                elif _weq(orig, prev):
                    refund = refund_prev + CLR if (orig != 0 and val == 0) else refund_prev
                else:
                    if orig != 0:
                        r_ = refund_prev - CLR if prev == 0 else (refund_prev + CLR if val == 0 else refund_prev)
                        refund = r_ + RST - SLD if _weq(orig, val) else r_
                    else:
                        refund = refund_prev + SET - SLD if _weq(orig, val) else refund_prev
                add_rw(1, TG.TxRefund, tx_id, value=refund, prev=refund_prev, vw=False, pw=False)
                warm_case = SLD if (_weq(prev, val) or not _weq(prev, orig)) else (SET if orig == 0 else RST)
                gas += warm_case if warm else warm_case + 2100
                rev_wc += 3
                sp += 2
            elif name in ("ADDRESS", "CALLER", "CALLVALUE"):
                v = rng.getrandbits(160) if name != "CALLVALUE" else _word(rng)
                tag = {"ADDRESS": CC.CalleeAddress, "CALLER": CC.CallerAddress, "CALLVALUE": CC.Value}[name]
                cc(tag, v, word=True)
                push(v, -1)
                sp -= 1
            elif name == "CALLDATASIZE":
                v = rng.randrange(0, 1 << 20)
                cc(CC.CallDataLength, v)
                push(v, -1)
                sp -= 1
            elif name in ("GAS", "MSIZE", "CODESIZE"):
                v = gas_left - 2 if name == "GAS" else (mws * 32 if name == "MSIZE" else len(C.code))
                push(v, -1)
                sp -= 1
                if name == "CODESIZE":
                    n_bc += 1
            elif name == "STOP":
                n_bc += 1  # header row (code length) + the STOP byte
                nxt_seg = seg + 1
                caller_id = 1 + nxt_seg
                NC = contracts[nxt_seg % n_contracts]
                cc(CC.IsSuccess, 1)
                cc(CC.CallerId, caller_id)
                saved = [(CC.IsRoot, 0, False), (CC.IsCreate, 0, False), (CC.CodeHash, NC.hash[0] | (NC.hash[1] << 128), True),
                         (CC.ProgramCounter, 0, False), (CC.StackPointer, SP0, False), (CC.GasLeft, GAS_REFILL, False),
                         (CC.MemorySize, 0, False), (CC.ReversibleWriteCounter, 0, False)]
                for tag, v, word in saved:
                    cc(tag, v, word=word, cid=caller_id)
                cc(CC.LastCalleeId, call_id, w=1, cid=caller_id)
                cc(CC.LastCalleeReturnDataOffset, 0, w=1, cid=caller_id)
                cc(CC.LastCalleeReturnDataLength, 0, w=1, cid=caller_id)
                gas_left += GAS_REFILL
            else:
                raise AssertionError(name)
            gas_left -= gas
            assert 0 <= sp <= 1024 and gas_left > 0
            looked_up_cells += 14 * (rwc - rw0) + 6 * n_bc
        seg += 1

    bytecode_rows = [r for c in contracts for r in c.table_rows()]
    meta = {
        "n_steps": n_steps, "n_pairs": n_steps - 1, "n_rw": len(rw), "n_bytecode": len(bytecode_rows),
        "segments": seg, "looked_up_cells": looked_up_cells,
        # 32 B x (step cells of the pair's current step + every looked-up row's cells)
        "algorithmic_bytes": 32 * (13 * (n_steps - 1) + looked_up_cells),
    }
    if not as_wire:
        return steps, rw, rw_flags, bytecode_rows, meta
    return {
        "steps": rows_to_rowmajor(steps, 13),
        "rw": rows_to_rowmajor(rw, 14), "rw_flags": np.array(rw_flags, dtype=np.uint32),
        "bytecode": rows_to_rowmajor(bytecode_rows, 6),
        "tx": np.zeros((0, 5, 4), dtype=np.uint64), "tx_flags": np.zeros(0, dtype=np.uint32),
        "block": np.zeros((0, 4, 4), dtype=np.uint64), "block_flags": np.zeros(0, dtype=np.uint32),
        "meta": meta,
    }
