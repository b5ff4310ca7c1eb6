"""Checker of the RW table -> State-circuit operations mapping (zk_state_ops_from_rw, csrc/state_rekey.hpp).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

The reference has no such function (nothing links the EVM circuit's rw_table to State-circuit rows).
What it does pin, and what this restates in plain Python:
  * the two numberings — EVM-side `Target` (src/zkevm_specs/evm_circuit/table.py:184-204) and State-side `Tag`
    (src/zkevm_specs/state_circuit.py:42-60);
  * the key slots every RWDictionary method writes (evm_circuit/typing.py:464-845) against the fields of the State circuit's
    `Operation` subclasses (state_circuit.py:616-825) — e.g. the CallContext field tag travels in the RW row's address cell
    (typing.py:510-531) but in `CallContextOp.field_tag` (state_circuit.py:717-722); `TxLogOp` unpacks log_id / field_tag / index;
  * the order the State circuit demands: (tag, id, address, field_tag, storage_key, rw_counter) strictly increasing
    (state_circuit.py:552-570), a StartOp in front (:634-645).
Pinning: tests/test_state_rekey.py builds RW rows with the reference's own RWDictionary where /root/reference exists and checks
that the ops this function derives are the reference's `Operation`s of the same accesses (parity of the mapping itself is
otherwise the State circuit accepting the derived witness: tests/test_super_circuit.py).
"""
M128 = (1 << 128) - 1
MAX_STATE_FIELD_TAG = 24  # state_circuit.py:34
# Target (table.py:184-204): Start 1, TxAccessListAccount 2, TxAccessListAccountStorage 3, TxRefund 4, Account 5, AccountStorage 6,
# CallContext 7, Stack 8, Memory 9, TxLog 10, TxReceipt 11 -> Tag (state_circuit.py:42-60)
STATE_TAG_OF_TARGET = {1: 1, 9: 2, 8: 3, 6: 4, 7: 5, 5: 6, 4: 7, 2: 8, 3: 9, 10: 10, 11: 11}
T_ACCOUNT, T_STORAGE, T_CALL_CONTEXT, T_TX_LOG = 5, 6, 7, 10


def rekey_row(c, flags):
    """One RW row (14 ints) -> (op 12 ints, op flags) or None for a dropped row.  Raises ValueError for a target cell that is not a
    Target, OverflowError when storage_key hi >= 2^128 (the 256-bit slot cannot hold lo | hi << 128)."""
    target = c[2]
    if target not in STATE_TAG_OF_TARGET:
        raise ValueError(f"not a Target: {target}")
    if c[7] >> 128:
        raise OverflowError("storage_key does not fit 256 bits")
    tag = STATE_TAG_OF_TARGET[target]
    id_, address, ft, key = c[3], c[4], c[5], c[6] | (c[7] << 128)
    vlo, vhi, ilo, ihi = c[8], c[9], 0, 0
    vw, iw, acc = flags & 1, 0, 0
    if target == T_CALL_CONTEXT:
        address, ft = 0, c[4]
        if ft > MAX_STATE_FIELD_TAG:
            return None
    elif target in (T_STORAGE, T_ACCOUNT):
        ilo, ihi, iw = c[12], c[13], 1
        if target == T_ACCOUNT:
            acc, id_ = 4, 0
    elif target == T_TX_LOG:
        address, ft, key = (c[4] >> 48), (c[4] >> 32) & 0xFFFF, c[4] & 0xFFFFFFFF
    return [c[0], c[1], tag, id_, address, ft, key, vlo, vhi, ilo, ihi, 1], vw | (iw << 1) | acc


def rw_to_state_ops(rows, rw_flags, strict=True):
    """rows: list of n RW rows (14 ints each), rw_flags: n ints -> (ops: list of 12-int lists, StartOp first, flags, status):
    status[i] = 0, or the reject code of RW row i (kind << 24 | site: ValueError 9 | 1, OverflowError 8 | 2) — with strict=True
    the first reject raises instead."""
    ops, flags, status = [], [], []
    for c, f in zip(rows, rw_flags):
        try:
            r = rekey_row(c, int(f))
            status.append(0)
        except ValueError:
            if strict:
                raise
            r = None
            status.append((9 << 24) | 1)
        except OverflowError:
            if strict:
                raise
            r = None
            status.append((8 << 24) | 2)
        if r is not None:
            ops.append(r[0])
            flags.append(r[1])
    order = sorted(range(len(ops)), key=lambda j: (ops[j][2], ops[j][3], ops[j][4], ops[j][5], ops[j][6], ops[j][0]))
    return [[0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]] + [ops[j] for j in order], [0] + [flags[j] for j in order], status
