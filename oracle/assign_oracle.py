"""CPU restatement (TEST INFRASTRUCTURE) of the reference's State-circuit witness assignment.

Follows src/zkevm_specs/state_circuit.py:
  * `_mpt_key` :897-901, `_mock_mpt_updates` :904-934 (first op of every distinct MPT key makes one
    MPTTableRow; the mock state root starts at 3 and grows by 5 per update),
  * `assign_state_circuit` :855-884 (root back-fill: row i carries the root_prev of the first MPT
    update *after* it, the final root 3 + 5 * n_updates when there is none),
  * `op2row` :827-852 (address -> ten 16-bit limbs, storage key -> 32 bytes, is_write),
  * `mpt_table_from_ops` :887-888.
Pinned against the unmodified reference by oracle/gen_golden_assign.py -> tests/golden/assign_cases.npz.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Wire format of one op (the reference's `Operation` NamedTuple :616-630), 12 slots of 256 bits:
  0 rw_counter, 1 rw, 2 tag, 3 id, 4 address, 5 field_tag, 6 storage_key   -- Python ints (U256), NOT reduced
  7 value.lo, 8 value.hi, 9 initial_value.lo, 10 initial_value.hi, 11 lexicographic_ordering_selector -- field cells
flags: bit0 value.is_word, bit1 initial_value.is_word, bit2 isinstance(field_tag, AccountFieldTag).
"""
from .codes import code

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
TAG_STORAGE, TAG_ACCOUNT = 4, 6
OP_NSLOTS = 12
KIND_ASSERT, KIND_OVERFLOW, KIND_UNSUPPORTED = 1, 8, 15
SITE_MPT_VALUE, SITE_MPT_INITIAL, SITE_ADDRESS, SITE_FIELD_TAG = 1, 2, 3, 4


def _has_key(op):
    return op[2] == TAG_ACCOUNT or op[2] == TAG_STORAGE  # raw ints, :898


def _mpt_key(op):
    return (op[4] % P, op[5] % P, op[6] & ((1 << 128) - 1), op[6] >> 128)  # :901


def _int_value(lo, hi):
    return lo + (hi << 128)  # Word.int_value, util/arithmetic.py:130-132


def _word(v):
    return v & ((1 << 128) - 1), v >> 128


def assign(ops, flags):
    """ops: list of 12-int lists, flags: list of ints ->
    (rows: list of 57-int lists, row_flags, mpt: list of 12-int lists in first-occurrence order,
     status: per-op code)."""
    n = len(ops)
    status = [0] * n
    # ---- _mock_mpt_updates :904-934
    first_of = {}
    order = []
    for i, op in enumerate(ops):
        if not _has_key(op):
            continue
        k = _mpt_key(op)
        if k not in first_of:
            first_of[k] = i
            order.append(i)
    rank = {i: r for r, i in enumerate(order)}
    mpt = []
    for i in order:
        op, f = ops[i], flags[i]
        root = 3 + 5 * rank[i]
        if f & 4:
            if not 1 <= op[5] <= 4:
                status[i] = code(KIND_UNSUPPORTED, SITE_FIELD_TAG)  # not a member of AccountFieldTag: wire misuse
            proof = op[5]  # MPTProofType.from_account_field_tag, table.py:341-350
        else:
            proof = 6  # StorageMod :914
        v = _int_value(op[7], op[8])
        iv = _int_value(op[9], op[10])
        if v >= 1 << 256:  # Word(int) sanity assert, util/arithmetic.py:116
            status[i] = status[i] or code(KIND_ASSERT, SITE_MPT_VALUE)
        elif iv >= 1 << 256:
            status[i] = status[i] or code(KIND_ASSERT, SITE_MPT_INITIAL)
        mpt.append([op[4] % P, proof, *_word(op[6]), *_word(root + 5), *_word(root), *_word(v % (1 << 256)),
                    *_word(iv % (1 << 256))])
    # ---- assign_state_circuit :855-884: root of row i = root_prev of the next MPT-keyed op after i
    final_root = 3 + 5 * len(order)
    nxt = final_root
    roots = [0] * n
    for i in reversed(range(n)):
        roots[i] = nxt
        if _has_key(ops[i]):
            nxt = 3 + 5 * rank[first_of[_mpt_key(ops[i])]]
    # ---- op2row :827-852
    rows, row_flags = [], []
    for i, op in enumerate(ops):
        addr = op[4]
        if addr >= 1 << 160 and not status[i]:
            status[i] = code(KIND_OVERFLOW, SITE_ADDRESS)  # int.to_bytes(20, "little") :834
        cells = [op[0] % P, 0 if op[1] == 0 else 1, op[2] % P, op[3] % P, addr % P, op[5] % P, *_word(op[6])]
        cells += [(addr >> (16 * k)) & 0xFFFF for k in range(10)]
        cells += [(op[6] >> (8 * k)) & 0xFF for k in range(32)]
        cells += [op[7], op[8], op[9], op[10], roots[i], 0, op[11]]
        rows.append(cells)
        row_flags.append(flags[i] & 3)
    return rows, row_flags, mpt, status


def first_error(status):
    """The exception `assign_state_circuit` raises: every `_mock_mpt_updates` failure precedes every
    `op2row` failure (:856 runs before :880-883); returns (op index, code) or None."""
    for sites in ((SITE_MPT_VALUE, SITE_MPT_INITIAL, SITE_FIELD_TAG), (SITE_ADDRESS,)):
        for i, c in enumerate(status):
            if c and (c & 0xFFFFFF) in sites:
                return i, c
    return None
