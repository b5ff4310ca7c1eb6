"""CPU restatement (TEST INFRASTRUCTURE) of the reference's Bytecode-circuit witness assignment
`assign_bytecode_circuit(k, bytecodes, keccak_randomness)` (src/zkevm_specs/bytecode_circuit.py:104-167):
push-data tracking (`push_data_left`, `push_data_size`, get_push_size evm_circuit/opcode.py:427-433), the running
`value_rlc = value_rlc * r + value` over the byte rows, `length`, q_first / q_last, truncation at 2^k rows and the
EMPTY_HASH padding rows.  Pinned against the unmodified reference by oracle/gen_golden_bytecode_assign.py ->
tests/golden/bytecode_assign_cases.npz.  Only tests/ may import this module.

rows: list of 6-int lists (hash lo, hi, tag, index, is_code, value) of all bytecodes back to back; offsets / lengths per
bytecode.  Returns the 2^k circuit rows as 12-int lists (bytecode_circuit.Row order: q_first, q_last, hash lo, hi, tag,
index, value, is_code, push_data_left, value_rlc, length, push_data_size)."""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
EMPTY_HASH = 0xC5D2460186F7233C927E7DB2DCC703C0E500B653CA82273B7BFAD8045D85A470


def push_size(v):
    return v - 0x5F if 0x60 <= v <= 0x7F else 0


def assign(k, rows, offsets, lengths, r):
    n = 1 << k
    out = []
    for j in range(len(lengths)):
        nxt, rlc = 0, 0
        for idx, row in enumerate(rows[int(offsets[j]):int(offsets[j + 1])]):
            left = nxt
            is_code = left == 0
            size = 0
            if idx > 0:
                size = push_size(row[5])
                nxt = size if is_code else left - 1
                rlc = (rlc * r + row[5]) % P
            off = len(out)
            out.append([int(off == 0), int(off == n - 1), row[0], row[1], row[2], row[3], row[5], row[4], left, rlc, int(lengths[j]) % P, size])
            if len(out) == n:
                return out
    lo, hi = EMPTY_HASH & ((1 << 128) - 1), EMPTY_HASH >> 128
    for off in range(len(out), n):
        out.append([int(off == 0), int(off == n - 1), lo, hi, 1, 0, 0, 0, 0, 0, 0, 0])
    return out
