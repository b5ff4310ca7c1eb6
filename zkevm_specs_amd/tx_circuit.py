"""Host-side mirror of `zkevm_specs.tx_circuit.verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES,
keccak_randomness)` (tx_circuit.py:253-291), evaluated on the MI355X.  The reference verifies every tx's signature
with a third-party secp256k1 call inside `ECDSAVerifyChip.verify` (:147-158); here that verdict is computed on the device
too (`zk_ecdsa_verify` over the chips' limbs) and enters the Tx kernel (`zk_sign_verify`) as the `ecdsa_status` column.
The first failing unit's exception propagates, as in the reference."""
import numpy as np

from . import oneshot
from .errors import raise_for_code
from .flatten import _n, flatten_tx_witness


def fill_ecdsa_column(wire, device=None):
    """Verdicts of the deferred chips (flatten_*_witness(..., ecdsa_on_device=True)) -> wire["meta"][:, 0]"""
    idx = np.nonzero(wire["ecdsa_deferred"])[0]
    if idx.size:
        v = None if wire["ecdsa_v"] is None else np.ascontiguousarray(wire["ecdsa_v"][idx])
        _, status = oneshot.ecdsa_verify(np.ascontiguousarray(wire["ecdsa_packed"][idx]), v, layout=0, device=device)
        wire["meta"][idx, 0] = status
    return wire


def verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, keccak_randomness):
    wire = fill_ecdsa_column(flatten_tx_witness(witness, MAX_TXS, ecdsa_on_device=True))
    if wire["bytes"].shape[0] == 0:
        return None
    res, _ = oneshot.sign_verify(wire, _n(keccak_randomness), is_sig=False)
    raise_for_code(res.first_fail_code, f"Tx circuit tx_index {res.first_fail_row}")
    return res


verify_tx_circuit = verify_circuit  # round-1 name


def verify_sig_circuit(witness, keccak_randomness):
    from .sig_circuit import verify_circuit as sig_verify_circuit

    return sig_verify_circuit(witness, keccak_randomness)
