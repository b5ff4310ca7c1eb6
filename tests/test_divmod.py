"""Long division used by the MOD / ADDMOD / MULMOD / SHR witness values (csrc/fr.hpp: Knuth D with a reciprocal-based
trial quotient) against Python integers: 256 / 256 and 512 / 256, random and the cases the trial-quotient step is
delicate on (top limb of the partial remainder equal to the divisor's, all-ones limbs, single-limb divisors)."""
import ctypes
import random

import numpy as np


def _pack(x, words):
    return [(x >> (64 * k)) & (2**64 - 1) for k in range(words)]


def _run(lib, wide, pairs):
    nw = 8 if wide else 4
    n = np.array([_pack(a, nw) for a, _ in pairs], dtype=np.uint64)
    d = np.array([_pack(b, 4) for _, b in pairs], dtype=np.uint64)
    q = np.zeros((len(pairs), nw), dtype=np.uint64)
    r = np.zeros((len(pairs), 4), dtype=np.uint64)
    vp = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
    lib.sim_divmod(ctypes.c_int(wide), vp(n), vp(d), vp(q), vp(r), ctypes.c_uint64(len(pairs)))
    unpack = lambda row: sum(int(v) << (64 * k) for k, v in enumerate(row))  # noqa: E731
    return [(unpack(q[i]), unpack(r[i])) for i in range(len(pairs))]


def _cases(bits, rng):
    M = (1 << bits) - 1
    out = []
    for _ in range(3000):
        db = rng.choice([1, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 160, 200, 255, 256])
        d = rng.getrandbits(db) | 1 << (db - 1) if rng.random() < 0.7 else max(1, rng.getrandbits(db))
        nb = rng.choice([0, 1, 32, 64, db, min(bits, db + 32), bits - 1, bits])
        out.append((rng.getrandbits(nb) if nb else 0, d))
    for db in (1, 32, 33, 64, 96, 128, 200, 224, 255, 256):
        d = (1 << db) - 1
        out += [(M, d), (M - 1, d), (d, d), (d - 1 if d > 1 else 0, d), (d * ((M // d) or 1) & M, d), (d << (bits - db) & M, d)]
        top = 1 << (db - 1)
        out += [(M, top), (M, top + 1), ((top << 32) - 1 & M, top), (((top + 1) << 64) - 1 & M, top + 1)]
        # partial remainders whose top limb equals the divisor's top limb: n = d * 2^k - 1 and neighbours
        for k in (32, 64, 96, bits - db):
            if 0 <= k and db + k <= bits:
                out += [((d << k) - 1, d), ((d << k) + 1 & M, d), (((d << k) - 1) ^ (1 << (k // 2)), d)]
    return out


def test_u256_and_u512_long_division(hostsim):
    rng = random.Random(5)
    for wide, bits in ((0, 256), (1, 512)):
        pairs = _cases(bits, rng)
        got = _run(hostsim, wide, pairs)
        for (n, d), (q, r) in zip(pairs, got):
            assert (q, r) == (n // d, n % d), (wide, hex(n), hex(d), hex(q), hex(r))


def test_fr_inv_and_div_match_python(hostsim):
    """FQ.inv / FQ division (util/arithmetic.py:59-60; py_ecc's prime_field_inv(0) == 0): the field inverse used by `zk_fr_op`
    ops 5 / 6, through the CPU build of the same source, against Python's pow(x, -1, p)."""
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = random.Random(7)
    edge = [0, 1, 2, P - 1, P - 2, 8, 4, 2**128, 2**253, pow(8, -1, P)]
    A = edge + [rng.randrange(P) for _ in range(300)]
    B = list(reversed(edge)) + [rng.randrange(P) for _ in range(300)]
    a = np.array([_pack(x, 4) for x in A], dtype=np.uint64)
    b = np.array([_pack(x, 4) for x in B], dtype=np.uint64)
    vp = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
    unpack = lambda row: sum(int(v) << (64 * k) for k, v in enumerate(row))  # noqa: E731
    inv = lambda x: pow(x, -1, P) if x else 0  # noqa: E731
    for op, f in ((5, lambda x, y: inv(x)), (6, lambda x, y: x * inv(y) % P)):
        out = np.zeros_like(a)
        hostsim.sim_fr_op(ctypes.c_int(op), vp(a), vp(b), vp(out), ctypes.c_uint64(len(A)))
        assert [unpack(r) for r in out] == [f(x, y) for x, y in zip(A, B)], op
    # the reference's own use: is_mul / is_div / is_mod come out as exactly 0 / 1 after FQ(8).inv() (execution/mul_div_mod.py:14-16)
    for opcode, want in ((2, (1, 0, 0)), (4, (0, 1, 0)), (6, (0, 0, 1))):
        got = ((4 - opcode) * (6 - opcode) * inv(8) % P, (opcode - 2) * (6 - opcode) * inv(4) % P, (opcode - 2) * (opcode - 4) * inv(8) % P)
        assert got == want
