// BN254 scalar field (Fr) + 256-bit integer helpers for the constraint kernels.
//
// Semantics follow the reference's field type `FQ` (src/zkevm_specs/util/arithmetic.py:41-63,
// a py_ecc.bn128.FQ with field_modulus = curve_order): every cell is a canonical integer in
// [0, p).  Wire/HBM format is 4 x u64 little-endian canonical (bit-comparable with `FQ.n`);
// in registers a cell is 8 x u32 (gfx950 VGPRs are 32-bit and the multiplier is
// v_mad_u64_u32, so the Montgomery CIOS inner product runs on 32-bit limbs).
//
// Multiplication is Montgomery (R = 2^256): mont(a, b) = a*b*R^-1.  Cells stay canonical in
// registers (range checks / comparisons read `.n` directly, like the reference does), so
//   fr_mul(a, b)   = mont(mont(a, b), R^2)      (canonical x canonical -> canonical)
//   fr_mulc(a, cM) = mont(a, cM)                 (cM = constant pre-converted to Montgomery form)
//
// The same source is compiled by hipcc for gfx950 (product) and by g++ with -DZK_HOSTSIM for
// the CPU logic tests under tests/hostsim (test infrastructure, never loaded by the package).
#pragma once
#include <stdint.h>
#include "fr_constants.h"

#if defined(ZK_HOSTSIM)
#define ZK_HD static inline
#define ZK_NOINLINE static
#define ZK_CONST static const
struct uint4 {
    uint32_t x, y, z, w;
};
#else
#include <hip/hip_runtime.h>
#define ZK_HD __device__ __forceinline__
// Out-of-line device functions (arguments and results by value = in VGPRs): keeps the EVM kernel's
// instruction footprint inside the instruction cache instead of inlining ~300-instruction bodies
// at hundreds of call sites.
#define ZK_NOINLINE __device__ __noinline__
#define ZK_CONST __device__ static const
#endif

typedef uint32_t u32;
typedef uint64_t u64;

struct Fr {
    u32 v[8];
};

#define FR_CONST_ARR(name, limbs) ZK_HD Fr name() { Fr r = {limbs}; return r; }
FR_CONST_ARR(fr_modulus, FR_P_LIMBS)
FR_CONST_ARR(frm_one, FR_R_LIMBS)
FR_CONST_ARR(frm_r2, FR_R2_LIMBS)
FR_CONST_ARR(frm_2p16, FRM_2P16_LIMBS)
FR_CONST_ARR(frm_2p64, FRM_2P64_LIMBS)
FR_CONST_ARR(frm_2p128, FRM_2P128_LIMBS)
FR_CONST_ARR(frm_256, FRM_256_LIMBS)
FR_CONST_ARR(frm_inv_2p128, FRM_INV_2P128_LIMBS)
FR_CONST_ARR(frm_inv2, FRM_INV2_LIMBS)
FR_CONST_ARR(frm_inv4, FRM_INV4_LIMBS)
FR_CONST_ARR(frm_inv8, FRM_INV8_LIMBS)
FR_CONST_ARR(frm_inv192, FRM_INV192_LIMBS)

ZK_HD Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
ZK_HD Fr fr_from_u64(u64 x) {
    Fr r = fr_zero();
    r.v[0] = (u32)x;
    r.v[1] = (u32)(x >> 32);
    return r;
}
ZK_HD Fr fr_from_u128(u64 lo, u64 hi) {
    Fr r = fr_zero();
    r.v[0] = (u32)lo;
    r.v[1] = (u32)(lo >> 32);
    r.v[2] = (u32)hi;
    r.v[3] = (u32)(hi >> 32);
    return r;
}
// Load one canonical cell (4 x u64 LE = 8 x u32 LE on a little-endian machine).
ZK_HD Fr fr_load(const u64* p) {
    Fr r;
    const uint4* q = (const uint4*)p;
    uint4 a = q[0], b = q[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
ZK_HD bool fr_is_zero(const Fr& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
ZK_HD bool fr_eq(const Fr& a, const Fr& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// value fits in 64 / 128 bits (upper limbs zero)
ZK_HD bool fr_fits64(const Fr& a) { return (a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0; }
ZK_HD bool fr_fits128(const Fr& a) { return (a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0; }
ZK_HD bool fr_fits32(const Fr& a) { return (a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0; }
ZK_HD u64 fr_lo64(const Fr& a) { return (u64)a.v[0] | ((u64)a.v[1] << 32); }
ZK_HD u64 fr_hi64of128(const Fr& a) { return (u64)a.v[2] | ((u64)a.v[3] << 32); }
ZK_HD bool fr_eq_u64(const Fr& a, u64 x) { return fr_fits64(a) && fr_lo64(a) == x; }
// a <= x as integers (x: u64)
ZK_HD bool fr_le_u64(const Fr& a, u64 x) { return fr_fits64(a) && fr_lo64(a) <= x; }
// number of significant bytes of the canonical integer (0 for zero)
ZK_HD int fr_byte_len(const Fr& a) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 w = a.v[i];
        if (w) n = 4 * i + (w >> 24 ? 4 : (w >> 16 ? 3 : (w >> 8 ? 2 : 1)));
    }
    return n;
}
ZK_HD u32 fr_byte(const Fr& a, int i) { return (a.v[i >> 2] >> (8 * (i & 3))) & 0xff; }

// integer compare of canonical values: a < b
ZK_HD bool fr_lt(const Fr& a, const Fr& b) {
    bool lt = false;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        lt = (a.v[i] < b.v[i]) || (a.v[i] == b.v[i] && lt);
    }
    return lt;
}
// raw 256-bit add/sub with carry/borrow out
ZK_HD u32 u256_add(Fr& r, const Fr& a, const Fr& b) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    return (u32)c;
}
ZK_HD u32 u256_sub(Fr& r, const Fr& a, const Fr& b) {
    u64 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 t = (u64)a.v[i] - b.v[i] - bw;
        r.v[i] = (u32)t;
        bw = (t >> 32) & 1;
    }
    return (u32)bw;
}
ZK_HD bool fr_geq_p(const Fr& a) {
    Fr p = fr_modulus();
    return !fr_lt(a, p);
}
ZK_HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr s, t;
    u256_add(s, a, b);  // a,b < p < 2^254: no carry out
    Fr p = fr_modulus();
    u32 bw = u256_sub(t, s, p);
    return bw ? s : t;
}
ZK_HD Fr fr_sub(const Fr& a, const Fr& b) {
    Fr d, t;
    u32 bw = u256_sub(d, a, b);
    Fr p = fr_modulus();
    u256_add(t, d, p);
    return bw ? t : d;
}
ZK_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
ZK_HD Fr fr_add_u64(const Fr& a, u64 x) { return fr_add(a, fr_from_u64(x)); }
ZK_HD Fr fr_sub_u64(const Fr& a, u64 x) { return fr_sub(a, fr_from_u64(x)); }

// Montgomery product a*b*R^-1 mod p (CIOS on 8 x 32-bit limbs); inputs < p.
ZK_NOINLINE Fr fr_mont(Fr a, Fr b) {
    const Fr p = fr_modulus();
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
        const u32 bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)a.v[j] * bi + t[j];
            t[j] = (u32)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (u32)c;
        t[9] = (u32)(c >> 32);
        const u32 m = t[0] * FR_INV32;
        c = (u64)m * p.v[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (u64)m * p.v[j] + t[j];
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (u32)c;
        t[8] = t[9] + (u32)(c >> 32);
    }
    Fr r, s;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    u32 bw = u256_sub(s, r, p);
    return (t[8] || !bw) ? s : r;
}
ZK_HD Fr fr_mul(const Fr& a, const Fr& b) { return fr_mont(fr_mont(a, b), frm_r2()); }
ZK_HD Fr fr_mulc(const Fr& a, const Fr& cM) { return fr_mont(a, cM); }
ZK_HD Fr fr_to_mont(const Fr& a) { return fr_mont(a, frm_r2()); }
// Mont(r^k) for k < 128 from rM = Mont(r), square-and-multiply: the lanes of a power-table kernel compute their own entries
// (14 products each) instead of one lane walking the table (64 dependent products: 80-180 us per table at one wavefront's pace)
ZK_HD Fr fr_pow_small_mont(const Fr& rM, u32 k) {
    Fr acc = frm_one(), sq = rM;
    for (int b = 0; b < 7; b++) {
        if ((k >> b) & 1u) acc = fr_mont(acc, sq);
        sq = fr_mont(sq, sq);
    }
    return acc;
}
// FQ.inv (util/arithmetic.py:59-60: py_ecc's prime_field_inv, which returns 0 for 0): Fermat, a^(p-2), a square-and-multiply
// walk over the bits of p - 2 in Montgomery form (254 squarings + the products of its 109 one-bits below the top; a vector
// op for callers and tests — the circuits' own inverses are all of constants, folded at build time)
ZK_HD Fr fr_inv(const Fr& a) {
    const Fr p = fr_modulus();
    Fr e = p;
    e.v[0] -= 2u;  // p - 2 (p ends in ...0001: no borrow past the low limb)
    const Fr aM = fr_to_mont(a);
    Fr acc = aM;   // bit 253 of p - 2 is its top bit
    for (int bit = 252; bit >= 0; bit--) {
        acc = fr_mont(acc, acc);
        if ((e.v[bit >> 5] >> (bit & 31)) & 1u) acc = fr_mont(acc, aM);
    }
    Fr one = fr_zero();
    one.v[0] = 1u;
    return fr_mont(acc, one);  // out of Montgomery form; a == 0 stays 0
}
ZK_HD Fr fr_div(const Fr& a, const Fr& b) { return fr_mul(a, fr_inv(b)); }  // FQ.__truediv__: a * b.inv()
// small-constant multiply by repeated doubling is avoided: use integer path when it fits.
ZK_HD Fr fr_mul_u64(const Fr& a, u64 k) { return fr_mulc(a, fr_to_mont(fr_from_u64(k))); }

// ---------------------------------------------------------------------------------------
// 256-bit unsigned integer helpers (EVM words). U256 reuses the Fr limb container but the
// value is a plain integer in [0, 2^256).
// ---------------------------------------------------------------------------------------
typedef Fr U256;

ZK_HD U256 u256_from_lo_hi(const Fr& lo, const Fr& hi) {  // lo, hi < 2^128 (caller-checked)
    U256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.v[i] = lo.v[i];
        r.v[4 + i] = hi.v[i];
    }
    return r;
}
ZK_HD Fr u256_lo(const U256& a) {
    Fr r = fr_zero();
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = a.v[i];
    return r;
}
ZK_HD Fr u256_hi(const U256& a) {
    Fr r = fr_zero();
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = a.v[4 + i];
    return r;
}
ZK_HD u64 u256_limb64(const U256& a, int i) { return (u64)a.v[2 * i] | ((u64)a.v[2 * i + 1] << 32); }

struct U512 {
    u32 v[16];
};
ZK_HD U512 u256_mul_full(const U256& a, const U256& b) {
    U512 r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.v[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)a.v[j] * b.v[i] + r.v[i + j];
            r.v[i + j] = (u32)c;
            c >>= 32;
        }
        r.v[i + 8] = (u32)c;
    }
    return r;
}
ZK_HD U256 u512_lo(const U512& a) {
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a.v[i];
    return r;
}
ZK_HD U256 u512_hi(const U512& a) {
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a.v[8 + i];
    return r;
}
ZK_HD U512 u512_from(const U256& lo, const U256& hi) {
    U512 r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = lo.v[i];
        r.v[8 + i] = hi.v[i];
    }
    return r;
}
ZK_HD int u256_bit(const U256& a, int i) { return (a.v[i >> 5] >> (i & 31)) & 1; }
ZK_HD int u512_bit(const U512& a, int i) { return (a.v[i >> 5] >> (i & 31)) & 1; }

// Long division (Knuth, TAOCP vol.2 4.3.1 algorithm D) of an NL-limb numerator by a non-zero
// 256-bit divisor on 32-bit limbs.  The divisor is normalised by a left shift of s = clz256(d)
// bits so that it always occupies 8 limbs with the top bit set: every array index below is then a
// compile-time constant after unrolling (no private-memory arrays on the GPU).  Used for the
// witness values the reference computes with Python big-int // and % (mul_div_mod.py:23-41,
// addmod.py:32-41, mulmod.py:41-50).
struct DivRes {
    U512 q;
    U256 r;
};
ZK_HD u32 zk_clz32(u32 x) {
#if defined(ZK_HOSTSIM)
    return x ? (u32)__builtin_clz(x) : 32u;
#else
    return (u32)__clz((int)x);
#endif
}
template <int NL>  // NL = numerator limbs (8 or 16)
ZK_HD void zk_divmod_limbs(const u32* n, const U256& d, u32* q /*NL limbs*/, U256& rem) {
    // s = leading zero bits of d
    u32 s = 0;
    bool seen = false;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        if (!seen) {
            if (d.v[k]) { s += zk_clz32(d.v[k]); seen = true; }
            else s += 32;
        }
    }
    const u32 bs = s & 31u, ws = s >> 5;
    // v = d << s (8 limbs), u = n << s (NL + 8 limbs, plus a zero guard limb)
    u32 v[8], u[NL + 9];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = bs ? ((d.v[k] << bs) | (k ? (d.v[k - 1] >> (32 - bs)) : 0u)) : d.v[k];
#pragma unroll
    for (int k = 0; k < NL + 9; k++) {
        const u32 lo = k < NL ? n[k] : 0u, prev = (k >= 1 && k - 1 < NL) ? n[k - 1] : 0u;
        u[k] = bs ? ((lo << bs) | (prev >> (32 - bs))) : lo;
    }
#pragma unroll
    for (int st = 2; st >= 0; st--) {  // word shift by ws in {0..7}: 4, 2, 1
        const int w = 1 << st;
        const bool on = (ws >> st) & 1u;
#pragma unroll
        for (int k = 7; k >= 0; k--) v[k] = on ? (k >= w ? v[k - w] : 0u) : v[k];
#pragma unroll
        for (int k = NL + 8; k >= 0; k--) u[k] = on ? (k >= w ? u[k - w] : 0u) : u[k];
    }
    const u64 vtop = v[7], vsec = v[6];
    // The NL trial quotients all divide by the same normalised top limb (2^31 <= vtop < 2^32): one real division for its
    // reciprocal dinv = floor((2^64 - 1) / vtop) - 2^32, then every 64-by-32 quotient is two multiplications and two
    // corrections (Moeller & Granlund, "Improved division by invariant integers", algorithm 4) instead of a 64-bit
    // division of ~100 instructions each.
    const u32 dtop = v[7];
    const u32 dinv = (u32)(~0ull / (u64)dtop - (1ull << 32));
#pragma unroll
    for (int j = NL - 1; j >= 0; j--) {
        const u32 u1 = u[j + 8], u0 = u[j + 7];
        // branch-free: the reciprocal form needs u1 < vtop; u1 == vtop (the partial remainder is below v * 2^32, so u1 cannot
        // exceed it) caps the trial quotient at 2^32 - 1 with remainder u0 + vtop
        const bool cap = u1 >= dtop;
        const u32 u1s = cap ? 0u : u1;
        const u64 qq = (u64)dinv * u1s + (((u64)u1s << 32) | u0);
        u32 q1 = (u32)(qq >> 32) + 1u;
        u32 r = u0 - q1 * dtop;
        const bool fix1 = r > (u32)qq;
        q1 -= fix1 ? 1u : 0u;
        r += fix1 ? dtop : 0u;
        const bool fix2 = r >= dtop;
        q1 += fix2 ? 1u : 0u;
        r -= fix2 ? dtop : 0u;
        u64 qhat = cap ? 0xffffffffull : (u64)q1;
        u64 rhat = cap ? (u64)u0 + vtop : (u64)r;
        // at most two corrections (Knuth D3)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (rhat >> 32 == 0 && qhat * vsec > ((rhat << 32) | u[j + 6])) {
                qhat--;
                rhat += vtop;
            }
        }
        // multiply and subtract: u[j..j+8] -= qhat * v
        u64 borrow = 0, carry = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 p = (u64)(u32)qhat * v[k] + carry;
            carry = p >> 32;
            const u64 t = (u64)u[j + k] - (u32)p - borrow;
            u[j + k] = (u32)t;
            borrow = (t >> 32) & 1u;
        }
        const u64 t = (u64)u[j + 8] - carry - borrow;
        u[j + 8] = (u32)t;
        if ((t >> 32) & 1u) {  // qhat was one too large: add the divisor back (D6)
            qhat--;
            u64 c2 = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                c2 += (u64)u[j + k] + v[k];
                u[j + k] = (u32)c2;
                c2 >>= 32;
            }
            u[j + 8] += (u32)c2;
        }
        q[j] = (u32)qhat;
    }
    // remainder = u[0..7] >> s  (word shift then bit shift)
    u32 r8[9];
#pragma unroll
    for (int k = 0; k < 8; k++) r8[k] = u[k];
    r8[8] = 0;
    // undo: the remainder of n<<s by d<<s is (n mod d) << s, still below d << s: it lives in the
    // low 8 + ws limbs... after the word shift it occupies limbs ws..7, so shift back down.
#pragma unroll
    for (int st = 2; st >= 0; st--) {
        const int w = 1 << st;
        const bool on = (ws >> st) & 1u;
#pragma unroll
        for (int k = 0; k < 8; k++) r8[k] = on ? (k + w < 8 ? r8[k + w] : 0u) : r8[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) rem.v[k] = bs ? ((r8[k] >> bs) | (k + 1 < 8 ? (r8[k + 1] << (32 - bs)) : 0u)) : r8[k];
}
ZK_NOINLINE DivRes u512_divmod_v(U512 n, U256 d, int nbits) {
    DivRes o;
    (void)nbits;
    zk_divmod_limbs<16>(n.v, d, o.q.v, o.r);
    return o;
}
struct DivRes256 {
    U256 q, r;
};
ZK_NOINLINE DivRes256 u256_divmod_v(U256 n, U256 d) {
    DivRes256 o;
    zk_divmod_limbs<8>(n.v, d, o.q.v, o.r);
    return o;
}
ZK_HD void u512_divmod(const U512& n, const U256& d, U512& q, U256& r, int nbits) {
    DivRes o = u512_divmod_v(n, d, nbits);
    q = o.q;
    r = o.r;
}
ZK_HD void u256_divmod(const U256& n, const U256& d, U256& q, U256& r) {
    DivRes256 o = u256_divmod_v(n, d);
    q = o.q;
    r = o.r;
}
