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


def test_wide_witness_matches_python_ints(hostsim):
    """csrc/bigz.hpp `wide_witness`: the witness values the reference computes on unbounded Python ints when word cells are
    >= 2^128 (mul_div_mod.py:23-41, shl_shr.py:121, instruction.py:545, sdiv_smod.py:85-104, addmod.py:32-41,61,
    mulmod.py:10,41-50), restated here with Python ints: low 256 bits, the two ways `Word(int)` raises, the side booleans."""
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = random.Random(11)

    def cell():
        return rng.choice([0, 1, rng.getrandbits(64), rng.getrandbits(128), 1 << 128, (1 << 128) + rng.getrandbits(20), P - 1,
                           rng.randrange(P), 1 << 253, 1 << 127, (1 << 128) - 1, rng.getrandbits(130) % P, 1 << 255 - 128])

    def word_flags(v):
        return 1 if v < 0 else (2 if v >= 1 << 256 else 0)

    def int_neg(x):
        return 0 if x == 0 else (1 << 256) - x

    def int_abs(x):
        return int_neg(x) if x >> 255 else x

    def expect(op, c):
        x = [c[2 * k] + (c[2 * k + 1] << 128) for k in range(4)]
        outs, b0, b1 = [None] * 4, 0, 0
        if op == 0:
            outs[0] = x[0] - x[1] * x[2]
        elif op == 1:
            if x[1] == 0:
                return None
            outs[0] = (x[0] - x[2]) // x[1]
        elif op == 2:
            outs[0] = (1 << 256) - x[0]
        elif op == 3:
            a1, a2, ap = int_abs(x[0]), int_abs(x[1]), int_abs(x[2])
            rem = a1 - ap * a2
            outs[0] = rem if x[0] >> 255 == 0 else int_neg(rem)
        elif op == 4:
            b0 = int(x[1] == 0)
            if not b0:
                a1, a2 = int_abs(x[0]), int_abs(x[1])
                if a2 == 0:
                    b1 = 1
                else:
                    outs[0] = a1 // a2 if x[0] >> 255 == x[1] >> 255 else int_neg(a1 // a2)
        elif op == 5:
            a, b, n, pr = x
            b0 = int(n == 0)
            if b0:
                ared, k, d, r = a, 0, 0, (a + b) % (1 << 256)
            else:
                ared, k, d, r = a % n, a // n, (a % n + b) // n, pr
            outs = [k, ared, d, r]
            n_is_zero = int((c[4] + c[5]) % P == 0)
            r_int = (r & ((1 << 256) - 1)) if b0 else pr
            b1 = int(pr == r_int * (1 - n_is_zero) % P)
        elif op == 6:
            a, b, n, r = x
            b0 = int(n == 0)
            ared, k = (0, 0) if b0 else (a % n, (a % n * b) // n)
            prod = ared * b
            outs = [ared, k, prod % (1 << 256), prod >> 256]
            b1 = int(prod == k * n + r)
        elif op == 7:
            outs[0] = x[0] // x[1] if x[1] else 0
        elif op == 8:
            outs[0] = x[0]
        return outs, b0, b1

    vp = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    n_checked = 0
    for op in range(9):
        tuples = [[cell() for _ in range(8)] for _ in range(1500)]
        exp = [expect(op, c) for c in tuples]
        keep = [i for i, e in enumerate(exp) if e is not None]
        x = np.array([[_pack(v, 4) for v in tuples[i]] for i in keep], dtype=np.uint64)
        out = np.zeros((len(keep), 4, 4), dtype=np.uint64)
        fl = np.zeros((len(keep), 6), dtype=np.uint32)
        hostsim.sim_wide_witness(ctypes.c_uint32(op), vp(x), vp(out), vp(fl), ctypes.c_uint64(len(keep)))
        for row, i in enumerate(keep):
            outs, b0, b1 = exp[i]
            assert (int(fl[row, 4]), int(fl[row, 5])) == (b0, b1), (op, tuples[i])
            for o, v in enumerate(outs):
                if v is None:
                    continue
                assert int(fl[row, o]) == word_flags(v), (op, o, tuples[i])
                if word_flags(v) == 0:
                    got = sum(int(w) << (64 * k) for k, w in enumerate(out[row, o]))
                    assert got == v, (op, o, tuples[i])
                n_checked += 1
    assert n_checked > 15000
