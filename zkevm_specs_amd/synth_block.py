"""ONE consistent synthetic block witness (BASELINE config 5): an execution trace whose RW table is ALSO a valid State-circuit
witness, so that the EVM circuit and the State circuit (and the Bytecode circuit over the executed contracts) run over the
same data — the State rows are the trace's RW rows, re-keyed and re-sorted (`rw_to_state_ops`), not an independent op list.

What "consistent" adds over synth_evm.synth_evm_trace (whose steps are individually valid but draw operands at random):
  * a simulated machine: operands are POPPED from a real stack, results are pushed back, memory is byte-addressed state per
    call, storage / access-list / refund state is per tx — every read returns what the last write left (State circuit 0.5);
  * stack positions are first written, then read, and a call touches a contiguous range of them (state_circuit.py:270-301);
  * every call-context field a step reads with a non-zero value was written before (check_call_context :328-345 wants the
    first access of a key to be a write or to read 0): a prelude of CallContext writes at rw_counters 1..K, before the first
    step, plays the role BeginTx / the CALL gadget play in a real block;
  * storage keeps one value per slot (SLOAD, and SSTORE of the value already there): the reference's own mock MPT table
    (`_mock_mpt_updates`, state_circuit.py:904-934) is built from the FIRST access of a slot while the circuit looks the LAST
    one up, so value-changing stores cannot satisfy it (the same restriction synth.synth_state_ops documents);
  * operand shapes the reference's gadgets cannot satisfy are avoided by construction: memory offsets, ADDMOD moduli and
    SIGNEXTEND indices come from a PUSH placed right in front of the instruction.
The opcode mix therefore has more PUSHes than config 3's (a straight-line program that only consumes its stack would
underflow): about 45 % instead of 30 %.
"""
import random

import numpy as np

from . import evm_tables as T
from .synth_evm import FR_P, M128, M256, NASTY, _STATE_OF, _signed, _weq
from .wire import rows_to_colmajor, rows_to_rowmajor

ES, OP, TG, CC = T.ExecutionState, T.Opcode, T.Target, T.CallContextFieldTag

# (weight, kind): config 3's mix without the kinds a consistent straight-line program cannot carry (GAS / MSIZE readers push
# values that depend on the call's history; they are kept, followed by what consumes them)
_MIX = [(12, "ADDSUB"), (10, "MULDIVMOD"), (5, "CMP"), (3, "SCMP"), (4, "BITWISE"), (2, "NOT"), (5, "ISZERO"), (3, "BYTE"),
        (2, "SIGNEXTEND"), (5, "SHIFT"), (1.5, "ADDMOD"), (1.5, "MULMOD"), (6, "POP"), (4, "MEMORY"), (1, "SLOAD"), (1, "SSTORE"),
        (2, "READER"), (14, "PUSH")]
# config 5's copy / keccak / exp traffic comes from the trace itself (block_ops=True): SHA3 over a word the program has just stored
# (execution/sha3.py:20-34: copy lookup Memory -> RlcAcc + keccak lookup), CODECOPY of a piece of the running contract
# (codecopy.py:26: copy lookup Bytecode -> Memory) and EXP with a small exponent (exp.py:31-33: two exp-table lookups)
_BLOCK_MIX = [(0.16, "SHA3"), (0.14, "CODECOPY"), (0.12, "EXP")]
_POPS = {"ADDSUB": 2, "MULDIVMOD": 2, "CMP": 2, "SCMP": 2, "BITWISE": 2, "NOT": 1, "ISZERO": 1, "BYTE": 2, "SHIFT": 2, "MULMOD": 3, "POP": 1,
         "SLOAD": 1, "SSTORE": 2, "READER": 0, "PUSH": 0}
_READERS = ["ADDRESS", "CALLER", "CALLVALUE", "CALLDATASIZE", "CODESIZE"]
# State-circuit tags (state_circuit.py:42-60) of the RW-table targets (evm_circuit/table.py:184-216)


class _Program:
    """straight-line contract generated against a stack-depth counter: no instruction ever finds too few operands"""

    def __init__(self, rng, n_ops, seed_tag=0, block_ops=False):
        self.ops, code, depth = [], bytearray(), 0
        # block_ops: False, or the weight multiplier of the SHA3 / CODECOPY / EXP kinds (True = 1: BASELINE configs[4]'s shares at 2^18 steps;
        # small test blocks raise it so that a few of each appear)
        mix = _MIX + ([(w * float(block_ops), k) for w, k in _BLOCK_MIX] if block_ops else [])
        kinds, weights = [k for _, k in mix], [w for w, _ in mix]
        self.sha3_inputs = []

        def emit(name, data=b""):
            self.ops.append((name, len(code), data))
            code.append(int(OP[name]))
            code.extend(data)

        def push(n=None, value=None):
            nonlocal depth
            n = n or rng.choice([1, 2, 4, 8, 20, 32])
            v = value if value is not None else (rng.choice(NASTY) & ((1 << (8 * n)) - 1) if rng.random() < 0.1 else rng.getrandbits(8 * n))
            emit(f"PUSH{n}", int(v).to_bytes(n, "big"))
            depth += 1

        while len(self.ops) < n_ops:
            kind = rng.choices(kinds, weights)[0]
            if kind == "MEMORY":
                name = rng.choice(["MLOAD", "MSTORE", "MSTORE8"])
                if name != "MLOAD":
                    if depth < 1:
                        push()
                push(2, rng.randrange(0, 4096))  # the offset, right in front of the instruction
                emit(name)
                depth -= 1 if name == "MLOAD" else 2
                depth += 1 if name == "MLOAD" else 0
                continue
            if kind == "SHA3":  # store a constant word, then hash exactly those 32 bytes: the input is known when the program is written
                c = rng.getrandbits(256)
                off = rng.randrange(0, 4096)
                push(32, c)
                push(2, off)
                emit("MSTORE")
                depth -= 2
                push(1, 32)
                push(2, off)
                emit("SHA3")
                depth -= 1
                self.sha3_inputs.append(c.to_bytes(32, "big"))
                continue
            if kind == "CODECOPY":  # size, code offset, memory offset from pushes right in front (reads past the code's end pad with zeros)
                push(1, rng.randrange(1, 64))
                push(2, rng.randrange(0, 700))
                push(2, rng.randrange(0, 4096))
                emit("CODECOPY")
                depth -= 3
                continue
            if kind == "EXP":  # exponent >= 2 from a PUSH1 / PUSH2 (the exp circuit's trace is ~1.5 rows per exponent bit), any base
                e_ = rng.randrange(2, 1 << rng.choice([3, 8, 12]))
                push(1 if e_ < 256 else 2, e_)
                push(rng.choice([1, 8, 32]))
                emit("EXP")
                depth -= 1
                continue
            if kind == "ADDMOD":  # the modulus must be below the field modulus (addmod.py:61): a 31-byte push, then a, b
                push(31)
                push(32)
                push(32)
                emit("ADDMOD")
                depth -= 2
                continue
            if kind in ("SSTORE", "SLOAD"):
                # slot keys come from a PUSH4 (the State circuit packs the 256-bit storage key over the address / field-tag
                # limbs, state_circuit.py:552-565: keys of 32 bits keep its lexicographic order meaningful), unique per program
                # location, odd for SSTORE and even for SLOAD; SSTORE's value is a non-zero PUSH32 (a slot that is 0 -> 0 needs the
                # NonExisting proof type, which the reference's mock MPT never produces, :904-934)
                self.n_slots = getattr(self, "n_slots", 0) + 1
                if kind == "SSTORE":
                    push(32, rng.getrandbits(256) | 1)
                push(4, ((seed_tag * 4096 + self.n_slots) << 1) | (1 if kind == "SSTORE" else 0))
                emit(kind)
                depth -= 2 if kind == "SSTORE" else 0
                continue
            if kind == "SIGNEXTEND":  # index from a PUSH1 (signextend.py:17-52: indices >= 256 are a corner the gadget cannot satisfy)
                if depth < 1:
                    push()
                push(1, rng.randrange(0, 40))
                emit("SIGNEXTEND")
                depth -= 1
                continue
            while depth < _POPS[kind]:
                push()
            if kind == "PUSH":
                push()
                continue
            name = {"ADDSUB": ["ADD", "SUB"], "MULDIVMOD": ["MUL", "DIV", "MOD"], "CMP": ["LT", "GT", "EQ"], "SCMP": ["SLT", "SGT"],
                    "BITWISE": ["AND", "OR", "XOR"], "SHIFT": ["SHL", "SHR", "SAR"], "READER": _READERS}.get(kind, [kind])
            name = rng.choice(name)
            emit(name)
            depth += {"NOT": 0, "ISZERO": 0, "SLOAD": 0, "POP": -1, "SSTORE": -2, "MULMOD": -2}.get(name, 1 if kind == "READER" else -1)
            assert depth >= 0
        emit("STOP")
        self.code = bytes(code)
        self.is_code = bytearray(b"\x01" * len(self.code))  # 0 on push data (bytecode table's is_code column)
        for _, pc_, data_ in self.ops:
            for k in range(len(data_)):
                self.is_code[pc_ + 1 + k] = 0
        h = rng.getrandbits(256)
        self.hash = (h & M128, h >> 128)

    def table_rows(self):
        lo, hi = self.hash
        rows = [[lo, hi, 1, 0, 0, len(self.code)]]
        is_data = bytearray(len(self.code))
        for _, pc, data in self.ops:
            for k in range(len(data)):
                is_data[pc + 1 + k] = 1
        rows.extend([lo, hi, 2, idx, 0 if is_data[idx] else 1, b] for idx, b in enumerate(self.code))
        return rows


def synth_block_codes(seed=5, seg_len=640, n_contracts=16, block_ops=False):
    rng = random.Random(seed)
    return [_Program(rng, seg_len - 1, k, block_ops).code for k in range(n_contracts)]


def synth_block_sha3_inputs(seed=5, seg_len=640, n_contracts=16, block_ops=True):
    """the byte strings the SHA3 steps of synth_block_trace(seed, ..., block_ops) hash (known from the programs alone)"""
    rng = random.Random(seed)
    return [m for k in range(n_contracts) for m in _Program(rng, seg_len - 1, k, block_ops).sha3_inputs]


def exp_event_rows(base, exponent, identifier):
    """ExpCircuit.add_event(base, exponent, identifier) (evm_circuit/typing.py:880-939): the square-and-multiply steps, most
    significant first -> (circuit rows of 21 cells, table rows of 11 cells: ExpTableRow, table.py:654-671)"""
    M = M256
    lo_hi = lambda v: [v & M128, v >> 128]  # noqa: E731
    steps = []

    def rec(e):
        if e == 0:
            return 1
        if e == 1:
            return base
        e1 = rec(e // 2)
        e2 = (e1 * e1) & M
        steps.append((e1, e1, e2))
        if e % 2 == 0:
            return e2
        ex = (base * e2) & M
        steps.append((e2, base, ex))
        return ex

    rec(exponent)
    steps.reverse()
    rows, table, e = [], [], exponent
    limbs = [(base >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
    for i, (a, b, d) in enumerate(steps):
        q, odd = divmod(e, 2)
        last = int(i == len(steps) - 1)
        rows.append([1, 1, identifier, last] + lo_hi(base) + lo_hi(e) + lo_hi(d) + lo_hi(a) + lo_hi(b) + [0, 0] + lo_hi(d) + lo_hi(q) + [odd])
        table.append([1, identifier, last] + limbs + lo_hi(e) + lo_hi(d))
        e = e // 2 if odd == 0 else e - 1
    return rows, table


def synth_block_trace(n_steps, seed=5, seg_len=640, n_contracts=16, code_hashes=None, block_ops=False, digest_of=None, randomness=None):
    """-> EVM wire dict (steps, rw, rw_flags, bytecode, tx, tx_flags, block, block_flags, meta) of a consistent n_steps-step trace.
    block_ops=True: the programs also hash, copy code and exponentiate (SHA3 / CODECOPY / EXP); the dict then carries what the
    other circuits and tables of the block are derived from — `copy_events` (the events of those steps in zk_copy_events form,
    rw_counters absolute: zk_copy_assign expands them to the Copy circuit's rows and the EVM circuit's copy table; their Memory
    rows are ALREADY in `rw`), `exp_rows` / `exp` (Exp circuit rows, column-major, and the EVM circuit's exp table), `sha3_inputs`
    (the keccak table's messages).  digest_of(list of bytes) -> list of 32-byte keccak-256 digests (the pushed hashes must be the
    real ones: the keccak table is built from the inputs); randomness: the block's keccak randomness (copy-table RLCs)."""
    rng = random.Random(seed)
    contracts = [_Program(rng, seg_len - 1, k, block_ops) for k in range(n_contracts)]
    digest = {}
    if block_ops:
        assert digest_of is not None and randomness is not None, "block_ops needs keccak digests and the keccak randomness"
        msgs = [m for c in contracts for m in c.sha3_inputs]
        digest = dict(zip(msgs, digest_of(msgs)))
    copy_events, copy_flags, copy_data, copy_offsets = [], [], [], [0]
    exp_rows, exp_table = [], []
    fixups = []  # (list, index, column): rw_counter-valued cells recorded relative to the prelude, shifted with the rows at the end
    if code_hashes is not None:
        for c, h in zip(contracts, code_hashes):
            c.hash = (h & M128, h >> 128)
    steps, rw, rw_flags = [], [], []
    looked_up_cells = 0
    tx_id, callee = 1, rng.getrandbits(160)
    rwc, seg = 0, 0  # rw_counters relative to the end of the prelude; shifted at the end
    gas_left, rev_wc = 10**9, 0
    SP0, GAS_REFILL = 1024, 10**7
    storage, warm_slots = {}, set()   # slot -> its one value; warmed slots
    ctx_writes = []                   # prelude: (call_id, field tag, value, is_word)
    ctx_vals = {}

    def add_rw(rw_, tag, id_=0, addr=0, ft=0, key=0, value=0, prev=0, aux=0, vw=True, pw=True):
        nonlocal rwc
        rw.append([rwc, rw_, int(tag), id_, addr, ft, key & M128, key >> 128, value & M128, value >> 128, prev & M128, prev >> 128,
                   aux & M128, aux >> 128])
        rw_flags.append((1 if vw else 0) | (2 if pw else 0))
        rwc += 1

    def ctx_value(call_id, tag, make, word=False):
        """the value of a call-context field: fixed per call, written once in the prelude when it is not zero"""
        k = (call_id, int(tag))
        if k not in ctx_vals:
            ctx_vals[k] = make()
            if ctx_vals[k] != 0:
                ctx_writes.append((call_id, int(tag), ctx_vals[k], word))
        return ctx_vals[k]

    while len(steps) < n_steps:
        C = contracts[seg % n_contracts]
        call_id = 1 + seg
        sp, mws = SP0, 0
        stack, memory = {}, {}
        for name, pc, data in C.ops:
            if len(steps) >= n_steps:
                break
            state = int(ES.STOP) if name == "STOP" else _STATE_OF[name]
            steps.append([state, rwc, call_id, 0, 0, C.hash[0], C.hash[1], pc, sp, gas_left, mws, rev_wc, 0])
            gas = T.OPCODES[name][1]
            n_bc = 1
            rw0 = rwc

            def pop(off):
                v = stack[sp + off]
                add_rw(0, TG.Stack, call_id, sp + off, value=v)
                return v

            def push(v, off):
                stack[sp + off] = v
                add_rw(1, TG.Stack, call_id, sp + off, value=v)

            def cc(tag, v, word=False, w=0, cid=None):
                add_rw(w, TG.CallContext, call_id if cid is None else cid, int(tag), value=v, vw=word)

            if name.startswith("PUSH"):
                push(int.from_bytes(data, "big"), -1)
                sp -= 1
                n_bc += 1 + len(data)
            elif name in ("ADD", "SUB", "MUL", "DIV", "MOD", "LT", "GT", "EQ", "SLT", "SGT", "AND", "OR", "XOR", "BYTE", "SIGNEXTEND", "SHL",
                          "SHR", "SAR"):
                a, b = pop(0), pop(1)
                if name == "ADD":
                    c = (a + b) & M256
                elif name == "SUB":
                    c = (a - b) & M256
                elif name == "MUL":
                    c = (a * b) & M256
                elif name in ("DIV", "MOD"):
                    c = 0 if b == 0 else (a // b if name == "DIV" else a % b)
                elif name in ("LT", "GT", "EQ", "SLT", "SGT"):
                    c = int({"LT": a < b, "GT": a > b, "EQ": a == b, "SLT": _signed(a) < _signed(b), "SGT": _signed(a) > _signed(b)}[name])
                elif name in ("AND", "OR", "XOR"):
                    c = a & b if name == "AND" else (a | b if name == "OR" else a ^ b)
                elif name == "BYTE":
                    c = (b >> (8 * (31 - a))) & 0xFF if a < 32 else 0
                elif name == "SIGNEXTEND":
                    if a < 31:
                        bit = 8 * a + 7
                        m = (1 << (bit + 1)) - 1
                        c = (b | (M256 ^ m)) if (b >> bit) & 1 else (b & m)
                    else:
                        c = b
                elif name in ("SHL", "SHR"):
                    c = 0 if a >= 256 else ((b << a) & M256 if name == "SHL" else b >> a)
                else:
                    c = (_signed(b) >> min(a, 256)) & M256
                push(c, 1)
                sp += 1
            elif name == "NOT":
                push(pop(0) ^ M256, 0)
            elif name == "ISZERO":
                push(int(pop(0) == 0), 0)
            elif name in ("ADDMOD", "MULMOD"):
                a, b, n_ = pop(0), pop(1), pop(2)
                c = 0 if n_ == 0 else ((a + b) % n_ if name == "ADDMOD" else (a * b) % n_)
                push(c, 2)
                sp += 2
            elif name == "POP":
                pop(0)
                sp += 1
            elif name in ("MLOAD", "MSTORE", "MSTORE8"):
                addr = pop(0)
                if name == "MLOAD":
                    v = int.from_bytes(bytes(memory.get(addr + k, 0) for k in range(32)), "big")
                    push(v, 0)
                else:
                    v = pop(1)
                    sp += 2
                vb = v.to_bytes(32, "little")
                if name == "MSTORE8":
                    memory[addr] = vb[0]
                    add_rw(1, TG.Memory, call_id, addr, value=vb[0], vw=False)
                    length = addr + 1
                else:
                    w_ = 0 if name == "MLOAD" else 1
                    for k in range(32):
                        if w_:
                            memory[addr + k] = vb[31 - k]
                        add_rw(w_, TG.Memory, call_id, addr + k, value=vb[31 - k], vw=False)
                    length = addr + 32
                mem_size = (length + mws + 31) // 32
                nxt = max(mws, mem_size)
                gas += (nxt * nxt // 512 + 3 * nxt) - (mws * mws // 512 + 3 * mws)
                mws = nxt
            elif name in ("SLOAD", "SSTORE"):
                cc(CC.TxId, ctx_value(call_id, CC.TxId, lambda: tx_id))
                if name == "SSTORE":
                    cc(CC.IsStatic, 0)
                cc(CC.RwCounterEndOfReversion, 0)
                cc(CC.IsPersistent, ctx_value(call_id, CC.IsPersistent, lambda: 1))
                cc(CC.CalleeAddress, ctx_value(call_id, CC.CalleeAddress, lambda: callee, True), word=True)
                key = pop(0)
                if key not in storage:
                    storage[key] = stack[sp + 1] if name == "SSTORE" else (rng.getrandbits(256) | 1)
                val = storage[key]
                warm = int(key in warm_slots)
                warm_slots.add(key)
                if name == "SLOAD":
                    add_rw(0, TG.AccountStorage, tx_id, callee, key=key, value=val, prev=val, aux=val)
                    push(val, 0)
                    add_rw(1, TG.TxAccessListAccountStorage, tx_id, callee, key=key, value=1, prev=warm, vw=False, pw=False)
                    gas += 100 if warm else 2100
                    rev_wc += 1
                else:
                    assert pop(1) == val  # the program stores the slot's own value (module docstring)
                    add_rw(1, TG.AccountStorage, tx_id, callee, key=key, value=val, prev=val, aux=val)
                    add_rw(1, TG.TxAccessListAccountStorage, tx_id, callee, key=key, value=1, prev=warm, vw=False, pw=False)
                    refund_prev = ctx_vals.get(("refund", tx_id), 0)
                    add_rw(1, TG.TxRefund, tx_id, value=refund_prev, prev=refund_prev, vw=False, pw=False)  # prev == value: unchanged
                    gas += 100 if warm else 2200
                    rev_wc += 3
                    sp += 2
            elif name in ("ADDRESS", "CALLER", "CALLVALUE"):
                tag = {"ADDRESS": CC.CalleeAddress, "CALLER": CC.CallerAddress, "CALLVALUE": CC.Value}[name]
                v = ctx_value(call_id, tag, (lambda: callee) if name == "ADDRESS" else (lambda: rng.getrandbits(160 if name == "CALLER" else 256)), True)
                cc(tag, v, word=True)
                push(v, -1)
                sp -= 1
            elif name == "CALLDATASIZE":
                v = ctx_value(call_id, CC.CallDataLength, lambda: rng.randrange(1, 1 << 20))
                cc(CC.CallDataLength, v)
                push(v, -1)
                sp -= 1
            elif name == "CODESIZE":
                push(len(C.code), -1)
                sp -= 1
                n_bc += 1
            elif name == "SHA3":
                off, size = pop(0), pop(1)
                data = bytes(memory.get(off + k, 0) for k in range(size))
                push(int.from_bytes(digest[data], "big"), 1)
                sp += 1
                copy_events.append([call_id, 0, 2, call_id, 0, 5, off, off + size, 0, size, 0, rwc])  # Memory -> RlcAcc
                fixups.append((copy_events, len(copy_events) - 1, 11))
                copy_flags.append(0)
                copy_data.extend(data)
                copy_offsets.append(len(copy_data))
                for k in range(size):
                    add_rw(0, TG.Memory, call_id, off + k, value=data[k], vw=False)
                nxt = max(mws, (off + size + 31) // 32)
                gas += 6 * ((size + 31) // 32) + (nxt * nxt // 512 + 3 * nxt) - (mws * mws // 512 + 3 * mws)
                mws = nxt
            elif name == "CODECOPY":
                moff, coff, size = pop(0), pop(1), pop(2)
                sp += 3
                n_real = max(0, min(size, len(C.code) - coff))
                copy_events.append([C.hash[0], C.hash[1], 1, call_id, 0, 2, coff, len(C.code), moff, size, 0, rwc])  # Bytecode -> Memory
                fixups.append((copy_events, len(copy_events) - 1, 11))
                copy_flags.append(1)  # the source id is a Word (the code hash)
                copy_data.extend(C.code[coff + k] | (C.is_code[coff + k] << 8) for k in range(n_real))
                copy_offsets.append(len(copy_data))
                for k in range(size):
                    b = C.code[coff + k] if k < n_real else 0
                    memory[moff + k] = b
                    add_rw(1, TG.Memory, call_id, moff + k, value=b, vw=False)
                n_bc += n_real
                nxt = max(mws, (moff + size + 31) // 32)
                gas += 3 * ((size + 31) // 32) + (nxt * nxt // 512 + 3 * nxt) - (mws * mws // 512 + 3 * mws)
                mws = nxt
            elif name == "EXP":
                base, exponent = pop(0), pop(1)
                push(pow(base, exponent, 1 << 256), 1)
                sp += 1
                rows_, table_ = exp_event_rows(base, exponent, rwc)  # identifier = the rw_counter after the three stack rows (exp.py:31)
                for r_ in rows_:
                    fixups.append((exp_rows, len(exp_rows), 2))
                    exp_rows.append(r_)
                for t_ in table_:
                    fixups.append((exp_table, len(exp_table), 1))
                    exp_table.append(t_)
                gas += 50 * ((exponent.bit_length() + 7) // 8)
            elif name == "STOP":
                n_bc += 1
                nxt_seg = seg + 1
                caller_id = 1 + nxt_seg
                NC = contracts[nxt_seg % n_contracts]
                cc(CC.IsSuccess, ctx_value(call_id, CC.IsSuccess, lambda: 1))
                cc(CC.CallerId, ctx_value(call_id, CC.CallerId, lambda: caller_id))
                saved = [(CC.IsRoot, 0, False), (CC.IsCreate, 0, False), (CC.CodeHash, NC.hash[0] | (NC.hash[1] << 128), True),
                         (CC.ProgramCounter, 0, False), (CC.StackPointer, SP0, False), (CC.GasLeft, GAS_REFILL, False),
                         (CC.MemorySize, 0, False), (CC.ReversibleWriteCounter, 0, False)]
                for tag, v, word in saved:
                    cc(tag, ctx_value(caller_id, tag, lambda v=v: v, word), word=word, cid=caller_id)
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

    # prelude: the call-context writes, at rw_counters 1..K; everything else moves up by K + 1
    K = len(ctx_writes)
    pre = [[1 + j, 1, int(TG.CallContext), cid, tag, 0, 0, 0, v & M128, v >> 128, 0, 0, 0, 0] for j, (cid, tag, v, _) in enumerate(ctx_writes)]
    pre_flags = [(1 if word else 0) | 2 for (_, _, _, word) in ctx_writes]
    for row in rw:
        row[0] += K + 1
    for s in steps:
        s[1] += K + 1
    for lst, j, col in fixups:
        lst[j][col] += K + 1
    rw, rw_flags = pre + rw, pre_flags + rw_flags
    bytecode_rows = [r for c in contracts for r in c.table_rows()]
    meta = {"n_steps": n_steps, "n_pairs": n_steps - 1, "n_rw": len(rw), "n_bytecode": len(bytecode_rows), "segments": seg,
            "prelude_rows": K, "looked_up_cells": looked_up_cells, "algorithmic_bytes": 32 * (13 * (n_steps - 1) + looked_up_cells)}
    out = {"steps": rows_to_rowmajor(steps, 13), "rw": rows_to_rowmajor(rw, 14), "rw_flags": np.array(rw_flags, dtype=np.uint32),
           "bytecode": rows_to_rowmajor(bytecode_rows, 6), "tx": np.zeros((0, 5, 4), dtype=np.uint64), "tx_flags": np.zeros(0, dtype=np.uint32),
           "block": np.zeros((0, 4, 4), dtype=np.uint64), "block_flags": np.zeros(0, dtype=np.uint32), "meta": meta}
    if block_ops:
        meta.update(copy_events=len(copy_events), copy_rows=2 * sum(e[9] for e in copy_events), exp_rows=len(exp_rows), sha3_steps=len(digest))
        out["copy_events"] = {"events": rows_to_rowmajor(copy_events, 12), "flags": np.array(copy_flags, dtype=np.uint32),
                              "data": np.array(copy_data, dtype=np.uint16), "offsets": np.array(copy_offsets, dtype=np.uint64), "r": int(randomness),
                              "n_rows": 2 * sum(e[9] for e in copy_events)}
        out["exp_rows"] = rows_to_colmajor(exp_rows, 21)
        out["exp"] = rows_to_rowmajor(exp_table, 11)
        out["sha3_inputs"] = [m for c in contracts for m in c.sha3_inputs]
    return out
