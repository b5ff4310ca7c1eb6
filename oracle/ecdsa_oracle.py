"""secp256k1 ECDSA verification — CPU restatement (TEST INFRASTRUCTURE, like everything under oracle/).

The reference verifies signatures through third-party eth-keys 0.4.0 (setup.cfg:24, absent from
/root/reference): `KeyAPI.Signature(vrs=[v, r, s])` + `KeyAPI.PublicKey(x_be + y_be)` +
`KeyAPI().ecdsa_verify(msg_hash, signature, public_key)` at src/zkevm_specs/tx_circuit.py:147-158 and
util/ec.py:109-117.  This restates that package's published native backend (ecdsa_raw_verify + jacobian.py; see
oracle/refshim/eth_keys/__init__.py for the statement of what is restated) on Python ints:
  * Signature: v in {0, 1}, 0 < r < N, 0 < s < N, else BadSignature;
  * w = inv(s, N); u1 = z w; u2 = r w; R = fast_add(fast_multiply(G, u1), fast_multiply(Q, u2)); verified iff r == R.x;
  * Jacobian coordinates, MSB-first double-and-add, a point with Y == 0 is the point at infinity, inv(0) == 0.
Pins: tests/golden/ecdsa_openssl.npz (signatures made and labelled by the image's OpenSSL 3.0.2 — independent of this
repo's arithmetic) and tests/golden/ecdsa_cases.npz (verdicts of the UNMODIFIED reference chips run on the stand-in).
What neither can pin is eth-keys' behaviour on public keys that are NOT on the curve beyond what is restated here.

status: 0 verified, 1 not verified, else (kind << 24) | site of the exception (codes.py).
"""
from .codes import UNSUPPORTED, code

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
BAD_SIGNATURE = code(UNSUPPORTED, 1)  # eth_keys BadSignature has no class of its own on the wire
KEY_RANGE = code(UNSUPPORTED, 2)      # coordinate >= P: outside the engine's declared domain


def jac_double(p):
    if not p[1]:
        return (0, 0, 0)
    ysq = p[1] * p[1] % P
    S = 4 * p[0] * ysq % P
    M = 3 * p[0] * p[0] % P
    nx = (M * M - 2 * S) % P
    ny = (M * (S - nx) - 8 * ysq * ysq) % P
    return (nx, ny, 2 * p[1] * p[2] % P)


def jac_add(p, q):
    if not p[1]:
        return q
    if not q[1]:
        return p
    U1, U2 = p[0] * q[2] ** 2 % P, q[0] * p[2] ** 2 % P
    S1, S2 = p[1] * q[2] ** 3 % P, q[1] * p[2] ** 3 % P
    if U1 == U2:
        return (0, 0, 1) if S1 != S2 else jac_double(p)
    H, R = U2 - U1, S2 - S1
    H2 = H * H % P
    H3 = H * H2 % P
    U1H2 = U1 * H2 % P
    nx = (R * R - H3 - 2 * U1H2) % P
    return (nx, (R * (U1H2 - nx) - S1 * H3) % P, H * p[2] * q[2] % P)


def jac_mul(a, n):
    """MSB-first double-and-add, iteratively (the recursion of jacobian_multiply unrolled)"""
    if a[1] == 0 or n == 0:
        return (0, 0, 1)
    acc = a
    for bit in bin(n)[3:]:
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, a)
    return acc


def from_jac(p):
    z = pow(p[2], -1, P) if p[2] % P else 0
    return (p[0] * z * z % P, p[1] * z * z * z % P)


def add(p1, p2):
    """affine group law on curve points (None = infinity): used to build test vectors only"""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    r = from_jac(jac_add((p1[0], p1[1], 1), (p2[0], p2[1], 1)))
    return None if r == (0, 0) else r


def mul(pt, k):
    r = from_jac(jac_mul((pt[0], pt[1], 1), k % N))
    return None if r == (0, 0) else r


def verify(pk_x, pk_y, z, r, s, v=None):
    """ints -> status"""
    if v is not None and v not in (0, 1):
        return BAD_SIGNATURE
    if not (0 < r < N and 0 < s < N):
        return BAD_SIGNATURE
    # coordinates >= P: eth-keys' formulas reduce mod P as they go, so the verdict is that of (x mod P, y mod P) — except
    # for y == P, which its `if not p[1]` infinity test sees as non-zero (csrc/secp256k1.hpp ecdsa_prepare): out of domain
    if pk_y == P:
        return KEY_RANGE
    pk_x, pk_y = pk_x % P, pk_y % P
    w = pow(s, -1, N)
    u1, u2 = z * w % N, r * w % N
    a = from_jac(jac_mul((G[0], G[1], 1), u1))
    b = from_jac(jac_mul((pk_x, pk_y, 1), u2))
    x, _ = from_jac(jac_add((a[0], a[1], 1), (b[0], b[1], 1)))
    return 0 if r == x else 1


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
