"""secp256k1 ECDSA verification — CPU restatement (TEST INFRASTRUCTURE, like everything under oracle/).

The reference verifies signatures through third-party eth-keys 0.4.0 (setup.cfg:24, absent from
/root/reference): `KeyAPI.Signature(vrs=[v, r, s])` + `KeyAPI.PublicKey(x_be + y_be)` +
`KeyAPI().ecdsa_verify(msg_hash, signature, public_key)` at src/zkevm_specs/tx_circuit.py:147-158 and
util/ec.py:109-117.  This restates that package's published native algorithm (`ecdsa_raw_verify`: w = s^-1 mod N,
u1 = z w, u2 = r w, R = u1 G + u2 Q with the affine chord / tangent formulas and LSB-first double-and-add) on
Python ints; `Signature` rejects v outside {0, 1} and r, s outside [0, N).
Pinned by tests/golden/ecdsa_cases.npz: verdicts of the UNMODIFIED reference chips (`ECDSAVerifyChip.verify` of
util/ec.py and of tx_circuit.py) run in the build container on the dependency stand-in oracle/refshim/eth_keys
(oracle/gen_golden_ecdsa.py).

status: 0 verified, 1 not verified, else (kind << 24) | site of the exception (codes.py).
"""
from .codes import UNSUPPORTED, VALUE_ERROR, code

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
BAD_SIGNATURE = code(UNSUPPORTED, 1)  # eth_keys BadSignature has no class of its own on the wire
KEY_RANGE = code(UNSUPPORTED, 2)      # coordinate >= P: outside the engine's declared domain
POW_ZERO = code(VALUE_ERROR, 3)       # pow(0, -1, P): "base is not invertible"


class _PowZero(Exception):
    pass


def _inv(a, m):
    a %= m
    if a == 0:
        raise _PowZero()
    return pow(a, -1, m)


def add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        m = 3 * x1 * x1 * _inv(2 * y1, P) % P
    else:
        m = (y2 - y1) * _inv(x2 - x1, P) % P
    x3 = (m * m - x1 - x2) % P
    return x3, (m * (x1 - x3) - y1) % P


def mul(pt, k):
    k %= N
    acc = None
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def verify(pk_x, pk_y, z, r, s, v=None):
    """ints -> status"""
    if v is not None and v not in (0, 1):
        return BAD_SIGNATURE
    if not (0 <= r < N and 0 <= s < N):
        return BAD_SIGNATURE
    if pk_x >= P or pk_y >= P:
        return KEY_RANGE
    if r == 0 or s == 0:
        return 1
    w = pow(s, -1, N)
    try:
        pt = add(mul(G, z * w % N), mul((pk_x, pk_y), r * w % N))
    except _PowZero:
        return POW_ZERO
    return 0 if pt is not None and pt[0] % N == r else 1


def verify_packed(sigs, v=None):
    """sigs uint8[n, 5, 32] (pk_x LE, pk_y LE, msg_hash BE, r LE, s LE), v optional uint32[n] -> list of status"""
    out = []
    for i in range(sigs.shape[0]):
        f = [sigs[i, k].tobytes() for k in range(5)]
        out.append(verify(int.from_bytes(f[0], "little"), int.from_bytes(f[1], "little"), int.from_bytes(f[2], "big"),
                          int.from_bytes(f[3], "little"), int.from_bytes(f[4], "little"), None if v is None else int(v[i])))
    return out


def sign_batch(n, seed):
    """n valid (pk_x, pk_y, z, r, s, v) tuples, cheap: keys d_i = d_0 + i and nonces k_i = k_0 + i advance by one
    affine addition of G each (test vectors only — never sign anything real this way)."""
    import random

    rng = random.Random(seed)
    d, k = rng.randrange(1, N - n), rng.randrange(1, N - n)
    Q, R = mul(G, d), mul(G, k)
    out = []
    for _ in range(n):
        z = rng.getrandbits(256)
        r = R[0] % N
        s = pow(k, -1, N) * (z + r * d) % N
        v = (R[1] & 1) ^ (1 if R[0] >= N else 0)
        if s > N // 2:
            s, v = N - s, v ^ 1
        out.append((Q[0], Q[1], z, r, s, v))
        d, k, Q, R = d + 1, k + 1, add(Q, G), add(R, G)
    return out


def pack(cases):
    """list of (pk_x, pk_y, z, r, s[, v]) -> uint8[n, 5, 32]"""
    import numpy as np

    buf = b"".join(c[0].to_bytes(32, "little") + c[1].to_bytes(32, "little") + c[2].to_bytes(32, "big") +
                   c[3].to_bytes(32, "little") + c[4].to_bytes(32, "little") for c in cases)
    return np.frombuffer(buf, dtype=np.uint8).reshape(-1, 5, 32).copy()
