"""Stand-in for rlp 3.0.0: `encode` of ints / bytes / nested lists (Ethereum RLP)."""


def _enc_len(n, offset):
    if n < 56:
        return bytes([offset + n])
    bl = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([offset + 55 + len(bl)]) + bl


def encode(obj):
    if isinstance(obj, bool):
        raise TypeError("cannot RLP-encode bool")
    if isinstance(obj, int):
        if obj < 0:
            raise ValueError("negative int")
        obj = obj.to_bytes((obj.bit_length() + 7) // 8, "big")
    if isinstance(obj, (bytes, bytearray)):
        obj = bytes(obj)
        if len(obj) == 1 and obj[0] < 0x80:
            return obj
        return _enc_len(len(obj), 0x80) + obj
    if isinstance(obj, (list, tuple)):
        payload = b"".join(encode(x) for x in obj)
        return _enc_len(len(payload), 0xC0) + payload
    raise TypeError(f"cannot RLP-encode {type(obj)}")
