"""Host-side mirror of `zkevm_specs.sig_circuit.verify_circuit(witness, keccak_randomness)` (sig_circuit.py:113-122:
`Row.verify` :64-104 per signature row), evaluated on the MI355X (`zk_ecdsa_verify` for the chips' verdicts, then
`zk_sign_verify` with Sig-circuit semantics)."""
from . import oneshot
from .errors import raise_for_code
from .flatten import _n, flatten_sig_witness
from .tx_circuit import fill_ecdsa_column


def verify_circuit(witness, keccak_randomness):
    wire = fill_ecdsa_column(flatten_sig_witness(witness, ecdsa_on_device=True))
    if wire["bytes"].shape[0] == 0:
        return None
    res, _ = oneshot.sign_verify(wire, _n(keccak_randomness), is_sig=True)
    raise_for_code(res.first_fail_code, f"Sig circuit row {res.first_fail_row}")
    return res
