"""Keccak-256 (the pre-NIST padding Ethereum uses) and the two address derivations the CREATE gadgets
compute inside the reference (instruction.py:1338-1352: rlp + eth_utils.keccak, third-party there).
Test infrastructure, like the rest of oracle/.  Pinned by tests/test_keccak.py (known answers and the
dependency shim the unmodified reference runs on)."""

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(x, n):
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def keccak_f(a):
    """a[x][y], 5x5 lanes of 64 bits"""
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y] & _M) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    p = bytearray(data) + b"\x01"
    p += b"\x00" * (-len(p) % rate)
    p[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for k in range(rate // 8):
            a[k % 5][k // 5] ^= int.from_bytes(p[off + 8 * k: off + 8 * k + 8], "little")
        a = keccak_f(a)
    return b"".join(a[k % 5][k // 5].to_bytes(8, "little") for k in range(4))


def rlp_addr_nonce(address: int, nonce: int) -> bytes:
    """rlp.encode([address as 20 big-endian bytes, nonce as an integer])"""
    nb = nonce.to_bytes((nonce.bit_length() + 7) // 8, "big")
    item = nb if (len(nb) == 1 and nb[0] < 0x80) else bytes([0x80 + len(nb)]) + nb  # len(nb) <= 32 < 56
    payload = bytes([0x94]) + address.to_bytes(20, "big") + item
    return bytes([0xC0 + len(payload)]) + payload  # len(payload) <= 54


def create_address(address: int, nonce: int) -> int:  # generate_contract_address (instruction.py:1338-1340)
    return int.from_bytes(keccak256(rlp_addr_nonce(address, nonce))[-20:], "big")


def create2_address(address: int, salt: int, code_hash: int) -> int:  # generate_CREAET2_contract_address (:1342-1352)
    pre = b"\xff" + address.to_bytes(20, "big") + salt.to_bytes(32, "little") + code_hash.to_bytes(32, "little")
    return int.from_bytes(keccak256(pre)[-20:], "big")
