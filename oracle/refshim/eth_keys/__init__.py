"""Stand-in for eth-keys 0.4.0 (pure-Python secp256k1; API surface of SURVEY.md Appendix B).

Verification restates the package's published native backend (eth_keys/backends/native/ecdsa.py + jacobian.py, the
backend a plain install of the reference uses: coincurve is not a dependency) as faithfully as it can be written down
without the source at hand:
  * `Signature(vrs=...)`: v in {0, 1}; r and s must satisfy 0 < value < N (`validate_signature_r_or_s`: `validate_gt(value, 0)`
    + `validate_lt_secpk1n`), violations raise BadSignature;
  * `ecdsa_raw_verify`: w = inv(s, N); u1 = z w mod N; u2 = r w mod N; (x, y) = fast_add(fast_multiply(G, u1),
    fast_multiply(Q, u2)); return r == x (NOT x mod N) and r mod N != 0 and s mod N != 0;
  * `inv(0, n) == 0` (extended Euclid with an early return), so the point at infinity (0, 0, 1) / (0, 0, 0) maps to (0, 0);
  * Jacobian arithmetic with the case analysis of jacobian.py: a point whose Y is 0 IS the point at infinity for
    `jacobian_add`; doubling it gives (0, 0, 0); equal X with different Y gives (0, 0, 1); MSB-first double-and-add.
  For keys on the curve every correct group law gives the same verdict; the case analysis only matters for public keys
  that are not on the curve (the reference never checks), e.g. a key with y = 0 is ignored altogether.
tests/golden/ecdsa_openssl.npz pins the verdicts on curve points independently (signatures made by the image's OpenSSL).

`KeyAPI.Signature(vrs=...)`, `.v/.r/.s`, `.recover_public_key_from_msg_hash`,
`KeyAPI.PublicKey(bytes64)` with `.to_bytes/.to_canonical_address/.to_address`,
`KeyAPI().ecdsa_verify`, `keys.PrivateKey(b32).sign_msg_hash` (RFC-6979) and `.public_key`.
"""
import hashlib
import hmac

from Crypto.Hash.keccak import keccak256 as _keccak

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (
    0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)


def _inv(a, m):
    return pow(a, -1, m)


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        m = 3 * x1 * x1 * _inv(2 * y1, P) % P
    else:
        m = (y2 - y1) * _inv(x2 - x1, P) % P
    x3 = (m * m - x1 - x2) % P
    return (x3, (m * (x1 - x3) - y1) % P)


def _mul(pt, k):
    k %= N
    acc = None
    while k:
        if k & 1:
            acc = _add(acc, pt)
        pt = _add(pt, pt)
        k >>= 1
    return acc


class BadSignature(Exception):
    pass


class ValidationError(Exception):
    pass


class PublicKey:
    def __init__(self, public_key_bytes):
        if len(public_key_bytes) != 64:
            raise ValidationError("Unexpected public key length")
        self._raw = bytes(public_key_bytes)

    def to_bytes(self):
        return self._raw

    def to_canonical_address(self):
        return _keccak(self._raw)[-20:]

    def to_address(self):
        return "0x" + self.to_canonical_address().hex()

    def _point(self):
        return (int.from_bytes(self._raw[:32], "big"), int.from_bytes(self._raw[32:], "big"))

    def __eq__(self, other):
        return isinstance(other, PublicKey) and self._raw == other._raw

    def __hash__(self):
        return hash(self._raw)


class Signature:
    def __init__(self, signature_bytes=None, vrs=None):
        if vrs is not None:
            v, r, s = vrs
        else:
            r = int.from_bytes(signature_bytes[:32], "big")
            s = int.from_bytes(signature_bytes[32:64], "big")
            v = signature_bytes[64]
        if v not in (0, 1):
            raise BadSignature("v must be 0 or 1")
        if not (0 < r < N and 0 < s < N):  # validate_signature_r_or_s: validate_gt(value, 0), validate_lt_secpk1n(value)
            raise BadSignature("r/s out of range")
        self._v, self._r, self._s = v, r, s

    v = property(lambda self: self._v)
    r = property(lambda self: self._r)
    s = property(lambda self: self._s)
    vrs = property(lambda self: (self._v, self._r, self._s))

    def recover_public_key_from_msg_hash(self, msg_hash):
        return _recover(msg_hash, self)


def _recover(msg_hash, sig):
    r, s, v = sig.r, sig.s, sig.v
    if r == 0 or s == 0:
        raise BadSignature("invalid signature")
    x = r
    y2 = (pow(x, 3, P) + 7) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        raise BadSignature("invalid signature")
    if (y & 1) != v:
        y = P - y
    z = int.from_bytes(msg_hash, "big")
    rinv = _inv(r, N)
    q = _add(_mul((x, y), s * rinv % N), _mul(G, (-z * rinv) % N))
    if q is None:
        raise BadSignature("invalid signature")
    return PublicKey(q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big"))


def _nat_inv(a, n):
    if a == 0:
        return 0
    lm, hm = 1, 0
    low, high = a % n, n
    while low > 1:
        r = high // low
        nm, new = hm - lm * r, high - low * r
        lm, low, hm, high = nm, new, lm, low
    return lm % n


def _jac_double(p):
    if not p[1]:
        return (0, 0, 0)
    ysq = (p[1] ** 2) % P
    S = (4 * p[0] * ysq) % P
    M = (3 * p[0] ** 2) % P  # curve parameter A = 0
    nx = (M ** 2 - 2 * S) % P
    ny = (M * (S - nx) - 8 * ysq ** 2) % P
    nz = (2 * p[1] * p[2]) % P
    return (nx, ny, nz)


def _jac_add(p, q):
    if not p[1]:
        return q
    if not q[1]:
        return p
    U1 = (p[0] * q[2] ** 2) % P
    U2 = (q[0] * p[2] ** 2) % P
    S1 = (p[1] * q[2] ** 3) % P
    S2 = (q[1] * p[2] ** 3) % P
    if U1 == U2:
        if S1 != S2:
            return (0, 0, 1)
        return _jac_double(p)
    H = U2 - U1
    R = S2 - S1
    H2 = (H * H) % P
    H3 = (H * H2) % P
    U1H2 = (U1 * H2) % P
    nx = (R ** 2 - H3 - 2 * U1H2) % P
    ny = (R * (U1H2 - nx) - S1 * H3) % P
    nz = (H * p[2] * q[2]) % P
    return (nx, ny, nz)


def _from_jac(p):
    z = _nat_inv(p[2], P)
    return ((p[0] * z ** 2) % P, (p[1] * z ** 3) % P)


def _jac_mul(a, n):
    if a[1] == 0 or n == 0:
        return (0, 0, 1)
    if n == 1:
        return a
    if n < 0 or n >= N:
        return _jac_mul(a, n % N)
    if (n % 2) == 0:
        return _jac_double(_jac_mul(a, n // 2))
    return _jac_add(_jac_double(_jac_mul(a, n // 2)), a)


def _fast_multiply(a, n):
    return _from_jac(_jac_mul((a[0], a[1], 1), n))


def _fast_add(a, b):
    return _from_jac(_jac_add((a[0], a[1], 1), (b[0], b[1], 1)))


def _verify(msg_hash, sig, pk):
    r, s = sig.r, sig.s
    w = _nat_inv(s, N)
    z = int.from_bytes(msg_hash, "big")
    u1, u2 = z * w % N, r * w % N
    x, _y = _fast_add(_fast_multiply(G, u1), _fast_multiply(pk._point(), u2))
    return bool(r == x and (r % N) and (s % N))


def _rfc6979(msg_hash, sk):
    x = sk.to_bytes(32, "big")
    v = b"\x01" * 32
    k = b"\x00" * 32
    k = hmac.new(k, v + b"\x00" + x + msg_hash, hashlib.sha256).digest()
    v = hmac.new(k, v, hashlib.sha256).digest()
    k = hmac.new(k, v + b"\x01" + x + msg_hash, hashlib.sha256).digest()
    v = hmac.new(k, v, hashlib.sha256).digest()
    while True:
        v = hmac.new(k, v, hashlib.sha256).digest()
        cand = int.from_bytes(v, "big")
        if 1 <= cand < N:
            return cand
        k = hmac.new(k, v + b"\x00", hashlib.sha256).digest()
        v = hmac.new(k, v, hashlib.sha256).digest()


class PrivateKey:
    def __init__(self, private_key_bytes):
        if len(private_key_bytes) != 32:
            raise ValidationError("Unexpected private key length")
        self._raw = bytes(private_key_bytes)
        self._d = int.from_bytes(self._raw, "big")
        q = _mul(G, self._d)
        self.public_key = PublicKey(q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big"))

    def to_bytes(self):
        return self._raw

    def sign_msg_hash(self, msg_hash):
        z = int.from_bytes(msg_hash, "big")
        k = _rfc6979(msg_hash, self._d)
        pt = _mul(G, k)
        r = pt[0] % N
        s = _inv(k, N) * (z + r * self._d) % N
        v = (pt[1] & 1) ^ (1 if pt[0] >= N else 0)
        if s > N // 2:  # low-s normalisation (eth-keys native backend)
            s = N - s
            v ^= 1
        return Signature(vrs=(v, r, s))


class KeyAPI:
    Signature = Signature
    PublicKey = PublicKey
    PrivateKey = PrivateKey

    def __init__(self, backend=None):
        pass

    def ecdsa_verify(self, msg_hash, signature, public_key):
        return _verify(msg_hash, signature, public_key)

    def ecdsa_recover(self, msg_hash, signature):
        return _recover(msg_hash, signature)

    def ecdsa_sign(self, msg_hash, private_key):
        return private_key.sign_msg_hash(msg_hash)


keys = KeyAPI()
