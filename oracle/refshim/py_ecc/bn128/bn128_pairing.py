"""Pairing placeholder — the pairing precompile/ECC circuit is out of scope."""


def pairing(q, p):
    raise NotImplementedError("BN254 pairing is out of scope for the oracle shims")
