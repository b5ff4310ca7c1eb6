"""Stand-in for eth-utils 2.0.0: only `keccak(bytes) -> bytes`."""
from Crypto.Hash.keccak import keccak256 as _k


def keccak(primitive=None, hexstr=None, text=None):
    if primitive is None:
        if hexstr is not None:
            primitive = bytes.fromhex(hexstr[2:] if hexstr.startswith("0x") else hexstr)
        elif text is not None:
            primitive = text.encode()
    if isinstance(primitive, int):
        primitive = primitive.to_bytes((primitive.bit_length() + 7) // 8, "big")
    return _k(bytes(primitive))
