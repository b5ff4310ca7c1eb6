// Signed multi-limb integers for the witness values the reference computes with *unbounded* Python ints on malformed word
// cells.  `Word.int_value()` (util/arithmetic.py:127-129) is `lo.n + (hi.n << 128)` with both cells anywhere in [0, p): up to
// 2^382, products up to 2^766, and `get_int_abs` / `get_int_neg` (util/arithmetic.py:279-287) turn such values negative
// (`(1 << 256) - x`), after which Python's `//` is a *floor* division with a negative operand.  The gadgets that do this —
// MUL/DIV/MOD (mul_div_mod.py:23-41), SHL/SHR (shl_shr.py:103-127), SDIV/SMOD (sdiv_smod.py:79-119, through abs_word
// instruction.py:539-571), ADDMOD (addmod.py:32-41,61), MULMOD (mulmod.py:6-29,41-50) — take this path only when a stack
// cell is >= 2^128 (never on a well-formed witness): the general build of the gadgets calls `wide_witness`, the fast build
// defers the pair to the general build.  Plain loops over limb arrays: correctness over speed, one out-of-line function.
#pragma once
#include "fr.hpp"

#define BZ_N 26  // 832 bits: |values| stay below 2^770 (see the bounds at each use)

struct BigZ {
    u32 m[BZ_N];  // magnitude, little-endian
    u32 neg;      // 1 = negative (never set for zero)
};

ZK_HD void bz_zero(BigZ& r) {
    for (int i = 0; i < BZ_N; i++) r.m[i] = 0;
    r.neg = 0;
}
ZK_HD bool bz_is_zero(const BigZ& a) {
    u32 o = 0;
    for (int i = 0; i < BZ_N; i++) o |= a.m[i];
    return o == 0;
}
ZK_HD int bz_cmp_mag(const BigZ& a, const BigZ& b) {
    for (int i = BZ_N - 1; i >= 0; i--)
        if (a.m[i] != b.m[i]) return a.m[i] < b.m[i] ? -1 : 1;
    return 0;
}
ZK_HD void bz_add_mag(BigZ& r, const BigZ& a, const BigZ& b) {
    u64 c = 0;
    for (int i = 0; i < BZ_N; i++) {
        c += (u64)a.m[i] + b.m[i];
        r.m[i] = (u32)c;
        c >>= 32;
    }
}
ZK_HD void bz_sub_mag(BigZ& r, const BigZ& a, const BigZ& b) {  // |a| >= |b|
    u64 bw = 0;
    for (int i = 0; i < BZ_N; i++) {
        const u64 t = (u64)a.m[i] - b.m[i] - bw;
        r.m[i] = (u32)t;
        bw = (t >> 32) & 1u;
    }
}
ZK_HD void bz_add_signed(BigZ& r, const BigZ& a, const BigZ& b, u32 b_neg) {
    BigZ t;
    if (a.neg == b_neg) {
        bz_add_mag(t, a, b);
        t.neg = a.neg;
    } else {
        const int c = bz_cmp_mag(a, b);
        if (c >= 0) { bz_sub_mag(t, a, b); t.neg = a.neg; }
        else { bz_sub_mag(t, b, a); t.neg = b_neg; }
    }
    if (bz_is_zero(t)) t.neg = 0;
    r = t;
}
ZK_HD void bz_add(BigZ& r, const BigZ& a, const BigZ& b) { bz_add_signed(r, a, b, b.neg); }
ZK_HD void bz_sub(BigZ& r, const BigZ& a, const BigZ& b) { bz_add_signed(r, a, b, bz_is_zero(b) ? 0u : (b.neg ^ 1u)); }
ZK_HD void bz_mul(BigZ& r, const BigZ& a, const BigZ& b) {  // the product fits BZ_N limbs (callers' bounds)
    BigZ t;
    bz_zero(t);
    for (int i = 0; i < BZ_N; i++) {
        if (a.m[i] == 0) continue;
        u64 c = 0;
        for (int j = 0; i + j < BZ_N; j++) {
            c += (u64)a.m[i] * b.m[j] + t.m[i + j];
            t.m[i + j] = (u32)c;
            c >>= 32;
        }
    }
    t.neg = bz_is_zero(t) ? 0u : (a.neg ^ b.neg);
    r = t;
}
ZK_HD int bz_bit_len(const BigZ& a) {
    for (int i = BZ_N - 1; i >= 0; i--)
        if (a.m[i]) return 32 * i + 32 - (int)zk_clz32(a.m[i]);
    return 0;
}
// Python's divmod(a, b) for b != 0: q = floor(a / b), r = a - q*b (the remainder takes the divisor's sign).  Restoring
// shift-and-subtract over the numerator's bits.
ZK_HD void bz_divmod_floor(BigZ& q, BigZ& r, const BigZ& a, const BigZ& b) {
    BigZ qq, rr;
    bz_zero(qq);
    bz_zero(rr);
    for (int bit = bz_bit_len(a) - 1; bit >= 0; bit--) {
        u32 c = (a.m[bit >> 5] >> (bit & 31)) & 1u;  // rr = (rr << 1) | bit
        for (int i = 0; i < BZ_N; i++) {
            const u32 n = rr.m[i] >> 31;
            rr.m[i] = (rr.m[i] << 1) | c;
            c = n;
        }
        if (bz_cmp_mag(rr, b) >= 0) {
            bz_sub_mag(rr, rr, b);
            qq.m[bit >> 5] |= 1u << (bit & 31);
        }
    }
    // qq = |a| div |b|, rr = |a| mod |b|
    if (a.neg != b.neg && !bz_is_zero(rr)) {  // floor: one further step towards minus infinity
        BigZ one;
        bz_zero(one);
        one.m[0] = 1;
        bz_add_mag(qq, qq, one);
        bz_sub_mag(rr, b, rr);
    }
    qq.neg = bz_is_zero(qq) ? 0u : (a.neg ^ b.neg);
    rr.neg = bz_is_zero(rr) ? 0u : b.neg;
    q = qq;
    r = rr;
}
ZK_HD void bz_shr_nonneg(BigZ& r, const BigZ& a, int k) {  // a >= 0
    BigZ t;
    bz_zero(t);
    const int ws = k >> 5, bs = k & 31;
    for (int i = 0; i + ws < BZ_N; i++) {
        const u32 lo = a.m[i + ws], hi = i + ws + 1 < BZ_N ? a.m[i + ws + 1] : 0u;
        t.m[i] = bs ? ((lo >> bs) | (hi << (32 - bs))) : lo;
    }
    r = t;
}
ZK_HD void bz_from_u256(BigZ& r, const U256& v) {
    bz_zero(r);
    for (int i = 0; i < 8; i++) r.m[i] = v.v[i];
}
ZK_HD void bz_pow2(BigZ& r, int k) {
    bz_zero(r);
    r.m[k >> 5] = 1u << (k & 31);
}
// Word.int_value(): lo + (hi << 128), cells anywhere below p
ZK_HD void bz_from_cells(BigZ& r, const Fr& lo, const Fr& hi) {
    BigZ h;
    bz_zero(r);
    bz_zero(h);
    for (int i = 0; i < 8; i++) { r.m[i] = lo.v[i]; h.m[4 + i] = hi.v[i]; }
    bz_add_mag(r, r, h);
}
// Word(int) (util/arithmetic.py:115-122): the low 256 bits and the two ways the constructor raises; flag bit 0 = negative
// (int.to_bytes -> OverflowError), bit 1 = not below 2^256 (the sanity assert, checked first)
ZK_HD u32 bz_to_word_int(U256& out, const BigZ& v) {
    for (int i = 0; i < 8; i++) out.v[i] = v.m[i];
    u32 hi = 0;
    for (int i = 8; i < BZ_N; i++) hi |= v.m[i];
    return v.neg ? 1u : (hi ? 2u : 0u);
}
ZK_HD void bz_int_neg(BigZ& r, const BigZ& x) {  // get_int_neg (util/arithmetic.py:283-284)
    if (bz_is_zero(x)) { bz_zero(r); return; }
    BigZ t;
    bz_pow2(t, 256);
    bz_sub(r, t, x);
}
ZK_HD void bz_int_abs(BigZ& r, const BigZ& x) {  // get_int_abs: `x >> 255` is truthy for every x >= 2^255 (x >= 0 here)
    BigZ s;
    bz_shr_nonneg(s, x, 255);
    if (!bz_is_zero(s)) bz_int_neg(r, x);
    else r = x;
}
ZK_HD bool bz_eq(const BigZ& a, const BigZ& b) { return a.neg == b.neg && bz_cmp_mag(a, b) == 0; }

// ---- the witness stages -----------------------------------------------------------------------------------------------
enum WideOp : u32 {
    WIDE_SUB_MUL = 0,   // o0 = Word(x0 - x1 * x2)                       DIV's c (mul_div_mod.py:29), SHR's remainder (shl_shr.py:121)
    WIDE_MOD_QUOT = 1,  // o0 = Word((x0 - x2) // x1), x1 != 0            MOD's a (mul_div_mod.py:38)
    WIDE_NEG256 = 2,    // o0 = Word((1 << 256) - x0)                     abs_word (instruction.py:545)
    WIDE_SDIV = 3,      // x0 = pop1, x1 = pop2, x2 = push: o0 = remainder sdiv_smod.py:88-94
    WIDE_SMOD = 4,      // x0 = pop1, x1 = pop2:           o0 = quotient  sdiv_smod.py:95-104; B0 = (pop2 == 0), B1 = ZeroDivisionError
    WIDE_ADDMOD = 5,    // x0 = a, x1 = b, x2 = n, x3 = pushed_r: o0 = k, o1 = a_reduced, o2 = d, o3 = r (n == 0) | B0 = (n == 0),
                        // B1 = pushed_r.int_value() == r.int_value() * (1 - n_is_zero) % p   (addmod.py:32-41,61)
    WIDE_MULMOD = 6,    // x0 = a, x1 = b, x2 = n, x3 = r: o0 = a_reduced, o1 = k, o2 = e, o3 = d | B0 = (n == 0),
                        // B1 = (prod == k * n + r)                                           (mulmod.py:41-50)
    WIDE_DIV = 7,       // o0 = x0 // x1 (x1 != 0; 0 when x1 == 0)                            mulmod.py:10 (a // n of the reduced a)
    WIDE_INT256 = 8     // o0 = x0.int_value() when it fits 256 bits; flag bit 1 when int.to_bytes(32) overflows
};
struct WideRes {
    U256 o[4];
    u32 fl[4];  // Word(int) flags of o[k] (bz_to_word_int)
    u32 b0, b1;
};

ZK_NOINLINE WideRes wide_witness(u32 op, Fr x0lo, Fr x0hi, Fr x1lo, Fr x1hi, Fr x2lo, Fr x2hi, Fr x3lo, Fr x3hi) {
    WideRes R;
    for (int k = 0; k < 4; k++) { R.o[k] = fr_zero(); R.fl[k] = 0; }
    R.b0 = R.b1 = 0;
    BigZ x0, x1, x2, x3, t, u, q, r;
    bz_from_cells(x0, x0lo, x0hi);
    bz_from_cells(x1, x1lo, x1hi);
    bz_from_cells(x2, x2lo, x2hi);
    bz_from_cells(x3, x3lo, x3hi);
    switch (op) {
    case WIDE_SUB_MUL:
        bz_mul(t, x1, x2);  // < 2^766
        bz_sub(t, x0, t);
        R.fl[0] = bz_to_word_int(R.o[0], t);
        break;
    case WIDE_MOD_QUOT:
        bz_sub(t, x0, x2);
        bz_divmod_floor(q, r, t, x1);
        R.fl[0] = bz_to_word_int(R.o[0], q);
        break;
    case WIDE_NEG256:
        bz_pow2(t, 256);
        bz_sub(t, t, x0);
        R.fl[0] = bz_to_word_int(R.o[0], t);
        break;
    case WIDE_SDIV: {
        BigZ a1, a2, ap, n1;
        bz_int_abs(a1, x0);
        bz_int_abs(a2, x1);
        bz_int_abs(ap, x2);
        bz_shr_nonneg(n1, x0, 255);
        bz_mul(t, ap, a2);  // |ap|, |a2| < 2^383
        bz_sub(t, a1, t);   // rem
        if (!bz_is_zero(n1)) bz_int_neg(t, t);
        R.fl[0] = bz_to_word_int(R.o[0], t);
        break;
    }
    case WIDE_SMOD: {
        R.b0 = bz_is_zero(x1);
        if (!R.b0) {
            BigZ a1, a2, n1, n2;
            bz_int_abs(a1, x0);
            bz_int_abs(a2, x1);
            bz_shr_nonneg(n1, x0, 255);
            bz_shr_nonneg(n2, x1, 255);
            if (bz_is_zero(a2)) { R.b1 = 1; break; }  // pop2 == 2^256 exactly (hi cell 2^128): a1 // 0
            bz_divmod_floor(q, r, a1, a2);
            if (!bz_eq(n1, n2)) bz_int_neg(q, q);
            R.fl[0] = bz_to_word_int(R.o[0], q);
        }
        break;
    }
    case WIDE_ADDMOD: {
        BigZ k, ared, d, rr;
        bz_zero(k);
        bz_zero(d);
        R.b0 = bz_is_zero(x2);
        if (R.b0) {
            ared = x0;
            bz_add(t, ared, x1);
            bz_zero(rr);
            for (int i = 0; i < 8; i++) rr.m[i] = t.m[i];  // % 2^256
        } else {
            bz_divmod_floor(k, ared, x0, x2);
            bz_add(t, ared, x1);
            bz_divmod_floor(d, u, t, x2);
            rr = x3;  // r = pushed_r
        }
        R.fl[0] = bz_to_word_int(R.o[0], k);
        R.fl[1] = bz_to_word_int(R.o[1], ared);
        R.fl[2] = bz_to_word_int(R.o[2], d);
        R.fl[3] = bz_to_word_int(R.o[3], rr);
        // pushed_r.int_value() == FQ(r.int_value() * (1 - n_is_zero)).n, where r is the Word just made (its cells: the low 256
        // bits when n == 0, pushed_r's own cells otherwise) and n_is_zero = is_zero(n.lo + n.hi) in the field
        {
            BigZ rv, pm;
            if (R.b0) { bz_zero(rv); for (int i = 0; i < 8; i++) rv.m[i] = rr.m[i]; }
            else rv = x3;
            const Fr nsum = fr_add(x2lo, x2hi);
            if (fr_is_zero(nsum)) bz_zero(rv);  // * (1 - 1)
            const Fr p = fr_modulus();
            bz_from_u256(pm, p);
            bz_divmod_floor(q, r, rv, pm);
            R.b1 = bz_eq(x3, r);
        }
        break;
    }
    case WIDE_MULMOD: {
        BigZ ared, k, prod;
        bz_zero(ared);
        bz_zero(k);
        R.b0 = bz_is_zero(x2);
        if (!R.b0) {
            bz_divmod_floor(q, ared, x0, x2);
            bz_mul(t, ared, x1);  // ared < n < 2^383
            bz_divmod_floor(k, r, t, x2);
        }
        bz_mul(prod, ared, x1);
        R.fl[0] = bz_to_word_int(R.o[0], ared);
        R.fl[1] = bz_to_word_int(R.o[1], k);
        bz_zero(t);
        for (int i = 0; i < 8; i++) t.m[i] = prod.m[i];  // prod % 2^256
        R.fl[2] = bz_to_word_int(R.o[2], t);
        bz_shr_nonneg(t, prod, 256);  // prod // 2^256
        R.fl[3] = bz_to_word_int(R.o[3], t);
        bz_mul(t, k, x2);  // k <= prod / n: k * n <= prod < 2^766
        bz_add(t, t, x3);
        R.b1 = bz_eq(prod, t);
        break;
    }
    case WIDE_DIV:
        if (!bz_is_zero(x1)) {
            bz_divmod_floor(q, r, x0, x1);
            R.fl[0] = bz_to_word_int(R.o[0], q);
        }
        break;
    case WIDE_INT256:
        R.fl[0] = bz_to_word_int(R.o[0], x0);
        break;
    default:
        break;
    }
    return R;
}
