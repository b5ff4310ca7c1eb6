"""Host-side mirrors of `zkevm_specs.tx_circuit.verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES,
keccak_randomness)` (tx_circuit.py:253-291) and `zkevm_specs.sig_circuit.verify_circuit(witness,
keccak_randomness)` (sig_circuit.py:113-122), evaluated on the MI355X.  The secp256k1 verification is a
third-party call in the reference; its outcome enters as the pre-computed `ecdsa_status` column
(flatten.py).  The first failing unit's exception propagates, as in the reference."""
from . import engine
from .errors import raise_for_code
from .flatten import _n, flatten_sig_witness, flatten_tx_witness


def verify_tx_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, keccak_randomness):
    wire = flatten_tx_witness(witness, MAX_TXS)
    with engine.open_sign(wire, _n(keccak_randomness), is_sig=False) as s:
        res = s.run()
    raise_for_code(res.first_fail_code, f"Tx circuit tx_index {res.first_fail_row}")
    return res


def verify_sig_circuit(witness, keccak_randomness):
    wire = flatten_sig_witness(witness)
    with engine.open_sign(wire, _n(keccak_randomness), is_sig=True) as s:
        res = s.run()
    raise_for_code(res.first_fail_code, f"Sig circuit row {res.first_fail_row}")
    return res
