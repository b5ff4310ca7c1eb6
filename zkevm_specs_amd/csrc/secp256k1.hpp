// secp256k1 ECDSA verification on the device: produces the `ecdsa_status` column the
// Tx / Sig kernels consume (sign_circuit.hpp) instead of taking it pre-computed from the host.
//
// Reference call sites: `ECDSAVerifyChip.verify` src/zkevm_specs/tx_circuit.py:147-158 and util/ec.py:109-117 —
// `KeyAPI.Signature(vrs=[v, r, s])`, `KeyAPI.PublicKey(x_be + y_be)`, `KeyAPI().ecdsa_verify(msg_hash, sig, pk)`.
// The arithmetic lives in third-party eth-keys 0.4.0 (setup.cfg:24; not under /root/reference): its native backend's
// `ecdsa_raw_verify` — w = inv(s, N), u1 = z w, u2 = r w, (x, y) = fast_add(fast_multiply(G, u1), fast_multiply(Q, u2)),
// accept iff r == x — with `Signature` rejecting v outside {0, 1} and r, s outside (0, N) (BadSignature), restated by
// oracle/ecdsa_oracle.py and pinned on curve points by OpenSSL-made vectors (tests/golden/ecdsa_openssl.npz).
// u2 * Q is the same MSB-first double-and-add with the same case analysis (Y == 0 is the point at infinity, equal X
// with different Y gives (0, 0, 1), inv(0) == 0), so that the verdict also agrees for public keys that are not on the
// curve (the formulas never use b); u1 * G (G is on the curve: any correct method yields the same point) uses a
// fixed-base table of 64 four-bit windows.
//
// Field elements: 8 x u32 limbs; the scalar field N in Montgomery form (R = 2^256), the base field P as plain residues
// (its special form makes the folded product cheaper than a Montgomery one);
// points: Jacobian (X, Y, Z) over P, infinity = "Y == 0".
#pragma once
#include "common.hpp"
#include "secp_constants.h"
#include "secp_g_table.h"

#if defined(ZK_HOSTSIM)
#define SP_MEMBER static inline
#else
#define SP_MEMBER __device__ __forceinline__ static
#endif
#define SP_CONST(name, limbs) SP_MEMBER Fr name() { Fr r = {limbs}; return r; }
struct SecpP {
    static constexpr bool plain = true;  // P = 2^256 - 2^32 - 977: plain residues, products folded with 2^32 + 977
    static constexpr u32 inv32 = SECP_P_INV32;
    SP_CONST(mod, SECP_P_LIMBS) SP_CONST(one, SECP_P_ONE_LIMBS) SP_CONST(r2, SECP_P_R2_LIMBS) SP_CONST(m2, SECP_P_M2_LIMBS)
};
struct SecpN {
    static constexpr bool plain = false;  // Montgomery form, R = 2^256
    static constexpr u32 inv32 = SECP_N_INV32;
    SP_CONST(mod, SECP_N_LIMBS) SP_CONST(one, SECP_N_ONE_LIMBS) SP_CONST(r2, SECP_N_R2_LIMBS) SP_CONST(m2, SECP_N_M2_LIMBS)
};
FR_CONST_ARR(secp_gx_m, SECP_GX_M_LIMBS)
FR_CONST_ARR(secp_gy_m, SECP_GY_M_LIMBS)

// a * b mod P for P = 2^256 - 2^32 - 977 (plain residues, a, b < P, result < P): 8 x 8 schoolbook product, then the high
// half is folded twice with 2^256 = 2^32 + 977 (mod P) — 72 multiply-adds instead of the 128 of a Montgomery product
ZK_NOINLINE Fr sp_mul_p(Fr a, Fr b) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
        const u32 bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)a.v[j] * bi + t[i + j];
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
    // r = lo + hi * 977 + (hi << 32): ten limbs, r[9] <= 1
    u32 r[10];
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c += (u64)t[k] + (u64)t[8 + k] * 977ull + (k > 0 ? (u64)t[8 + k - 1] : 0ull);
        r[k] = (u32)c;
        c >>= 32;
    }
    c += t[15];
    r[8] = (u32)c;
    r[9] = (u32)(c >> 32);
    // second fold: v = r[8..9] < 2^33, v * (2^32 + 977) < 2^67
    const u64 v = (u64)r[8] | ((u64)r[9] << 32);
    const u64 m977 = v * 977ull;
    Fr s;
    c = (u64)r[0] + (u32)m977;
    s.v[0] = (u32)c; c >>= 32;
    c += (u64)r[1] + (m977 >> 32) + (u32)v;
    s.v[1] = (u32)c; c >>= 32;
    c += (u64)r[2] + (v >> 32);
    s.v[2] = (u32)c; c >>= 32;
#pragma unroll
    for (int k = 3; k < 8; k++) {
        c += r[k];
        s.v[k] = (u32)c;
        c >>= 32;
    }
    if (c) {  // wrapped past 2^256 (s is tiny then): once more + (2^32 + 977), no further carry
        u64 d = (u64)s.v[0] + 977ull;
        s.v[0] = (u32)d; d >>= 32;
        d += (u64)s.v[1] + 1ull;
        s.v[1] = (u32)d; d >>= 32;
#pragma unroll
        for (int k = 2; k < 8; k++) { d += s.v[k]; s.v[k] = (u32)d; d >>= 32; }
    }
    Fr q;
    const u32 bw = u256_sub(q, s, SecpP::mod());
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = bw ? s.v[i] : q.v[i];
    return s;
}
// Montgomery product a*b*R^-1 mod m (CIOS); a, b < m; result < m.  (Base field: the plain product above.)
template <class M>
ZK_NOINLINE Fr sp_mont(Fr a, Fr b) {
    if constexpr (M::plain) return sp_mul_p(a, b);
    const Fr m = M::mod();
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
        const u32 q = t[0] * M::inv32;
        c = (u64)q * m.v[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (u64)q * m.v[j] + t[j];
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
    const u32 bw = u256_sub(s, r, m);
    const bool take = t[8] || !bw;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = take ? s.v[i] : r.v[i];
    return r;
}
template <class M>
ZK_HD Fr sp_add(const Fr& a, const Fr& b) {
    Fr s, t;
    const u32 carry = u256_add(s, a, b);
    const u32 bw = u256_sub(t, s, M::mod());
    const bool take = carry || !bw;
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = take ? t.v[i] : s.v[i];
    return s;
}
template <class M>
ZK_HD Fr sp_sub(const Fr& a, const Fr& b) {
    Fr d, t;
    const u32 bw = u256_sub(d, a, b);
    u256_add(t, d, M::mod());
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = bw ? t.v[i] : d.v[i];
    return d;
}
template <class M>
ZK_HD Fr sp_reduce_once(const Fr& x) {  // x < 2^256 < 2m
    Fr t, r = x;
    const u32 bw = u256_sub(t, x, M::mod());
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = bw ? x.v[i] : t.v[i];
    return r;
}
template <class M>
ZK_HD Fr sp_to_mont(const Fr& a) { return sp_mont<M>(a, M::r2()); }
template <class M>
ZK_HD Fr sp_from_mont(const Fr& aM) { return sp_mont<M>(aM, fr_from_u64(1)); }
// aM^(m-2) in Montgomery form (Fermat inverse; m prime, a != 0)
template <class M>
ZK_NOINLINE Fr sp_inv(Fr aM) {
    Fr e = M::m2();
    Fr acc = M::one();
    for (int i = 0; i < 256; i++) {
        acc = sp_mont<M>(acc, acc);
        if (e.v[7] >> 31) acc = sp_mont<M>(acc, aM);
#pragma unroll
        for (int j = 7; j > 0; j--) e.v[j] = (e.v[j] << 1) | (e.v[j - 1] >> 31);
        e.v[0] <<= 1;
    }
    return acc;
}

// Jacobian points (X, Y, Z) over P with eth-keys' conventions (jacobian.py): a point whose Y is 0 IS the point at
// infinity for every operation — (0, 0, 1) is what a multiplication by zero returns, (0, 0, 0) what doubling a Y == 0
// point returns — and Z == 0 only ever occurs in (0, 0, 0).  For points of the curve no finite point has Y == 0 (the group
// order is odd), so these conventions are an ordinary group law there; for public keys that are not on the curve they are
// what makes the verdict reproducible.
struct SpPoint {
    Fr X, Y, Z;  // residues mod P
};
ZK_HD SpPoint sp_infinity() {  // (0, 0, 1)
    SpPoint p;
    p.X = fr_zero(); p.Y = fr_zero(); p.Z = SecpP::one();
    return p;
}
// jacobian_double: Y == 0 -> (0, 0, 0); else "dbl-2009-l" with a = 0 (any Jacobian doubling formula yields a representative
// of the same point; only the Y == 0 / X-equality tests and the final X / Z^2 are observable)
ZK_NOINLINE SpPoint sp_dbl(SpPoint p) {
    if (fr_is_zero(p.Y)) {
        SpPoint o;
        o.X = fr_zero(); o.Y = fr_zero(); o.Z = fr_zero();
        return o;
    }
    typedef SecpP F;
    const Fr A = sp_mont<F>(p.X, p.X), B = sp_mont<F>(p.Y, p.Y), C = sp_mont<F>(B, B);
    Fr t = sp_add<F>(p.X, B);
    t = sp_sub<F>(sp_sub<F>(sp_mont<F>(t, t), A), C);
    const Fr D = sp_add<F>(t, t), E = sp_add<F>(sp_add<F>(A, A), A), Fq = sp_mont<F>(E, E);
    SpPoint r;
    r.X = sp_sub<F>(Fq, sp_add<F>(D, D));
    Fr c8 = sp_add<F>(C, C);
    c8 = sp_add<F>(c8, c8);
    c8 = sp_add<F>(c8, c8);
    r.Y = sp_sub<F>(sp_mont<F>(E, sp_sub<F>(D, r.X)), c8);
    const Fr yz = sp_mont<F>(p.Y, p.Z);
    r.Z = sp_add<F>(yz, yz);
    return r;
}
// jacobian_add: p.Y == 0 -> q; q.Y == 0 -> p; U1 == U2: S1 != S2 -> (0, 0, 1), else double(p); otherwise the chord
ZK_NOINLINE SpPoint sp_add_points(SpPoint p, SpPoint q) {
    if (fr_is_zero(p.Y)) return q;
    if (fr_is_zero(q.Y)) return p;
    typedef SecpP F;
    const Fr z1z1 = sp_mont<F>(p.Z, p.Z), z2z2 = sp_mont<F>(q.Z, q.Z);
    const Fr u1 = sp_mont<F>(p.X, z2z2), u2 = sp_mont<F>(q.X, z1z1);
    const Fr s1 = sp_mont<F>(sp_mont<F>(p.Y, q.Z), z2z2), s2 = sp_mont<F>(sp_mont<F>(q.Y, p.Z), z1z1);
    if (fr_eq(u1, u2)) {
        if (!fr_eq(s1, s2)) return sp_infinity();
        return sp_dbl(p);
    }
    const Fr h = sp_sub<F>(u2, u1), rr = sp_sub<F>(s2, s1);
    const Fr hh = sp_mont<F>(h, h), hhh = sp_mont<F>(h, hh), v = sp_mont<F>(u1, hh);
    SpPoint r;
    r.X = sp_sub<F>(sp_sub<F>(sp_mont<F>(rr, rr), hhh), sp_add<F>(v, v));
    r.Y = sp_sub<F>(sp_mont<F>(rr, sp_sub<F>(v, r.X)), sp_mont<F>(s1, hhh));
    r.Z = sp_mont<F>(sp_mont<F>(p.Z, q.Z), h);
    return r;
}
// p + (x2, y2, 1): the fixed-base table's addition (8M + 3S); same case analysis (table entries are curve points: y2 != 0)
ZK_NOINLINE SpPoint sp_add_affine(SpPoint p, Fr x2, Fr y2) {
    typedef SecpP F;
    if (fr_is_zero(p.Y)) {
        SpPoint r;
        r.X = x2; r.Y = y2; r.Z = F::one();
        return r;
    }
    const Fr z1z1 = sp_mont<F>(p.Z, p.Z);
    const Fr u2 = sp_mont<F>(x2, z1z1), s2 = sp_mont<F>(sp_mont<F>(y2, p.Z), z1z1);
    if (fr_eq(p.X, u2)) {
        if (!fr_eq(p.Y, s2)) return sp_infinity();
        return sp_dbl(p);
    }
    const Fr h = sp_sub<F>(u2, p.X), rr = sp_sub<F>(s2, p.Y);
    const Fr hh = sp_mont<F>(h, h), hhh = sp_mont<F>(h, hh), v = sp_mont<F>(p.X, hh);
    SpPoint r;
    r.X = sp_sub<F>(sp_sub<F>(sp_mont<F>(rr, rr), hhh), sp_add<F>(v, v));
    r.Y = sp_sub<F>(sp_mont<F>(rr, sp_sub<F>(v, r.X)), sp_mont<F>(p.Y, hhh));
    r.Z = sp_mont<F>(p.Z, h);
    return r;
}
// k * G, k < N: sixty-four 4-bit windows over the precomputed multiples d * 16^j * G (secp_g_table.h) — 64 mixed additions.
// G is on the curve, so this is the group element eth-keys' double-and-add produces (k == 0: a Y == 0 point).
ZK_NOINLINE SpPoint sp_scalar_mul_g(Fr k) {
    SpPoint acc = sp_infinity();
    for (int j = 0; j < 64; j++) {
        const u32 d = k.v[0] & 15u;
        if (d) {
            const uint32_t* e = secp_g_table[15 * j + (int)d - 1];
            Fr x, y;
#pragma unroll
            for (int q = 0; q < 8; q++) { x.v[q] = e[q]; y.v[q] = e[8 + q]; }
            acc = sp_add_affine(acc, x, y);
        }
#pragma unroll
        for (int q = 0; q < 7; q++) k.v[q] = (k.v[q] >> 4) | (k.v[q + 1] << 28);
        k.v[7] >>= 4;
    }
    return acc;
}
// jacobian_multiply(pt, k), k < N: pt.Y == 0 or k == 0 -> (0, 0, 1); else MSB-first double-and-add (the recursion
// n -> n // 2 unrolled: start from pt at the top bit, then double, and add pt where the bit is set)
ZK_NOINLINE SpPoint sp_scalar_mul(SpPoint pt, Fr k) {
    if (fr_is_zero(pt.Y) || fr_is_zero(k)) return sp_infinity();
    int top = 255;
    while (!((k.v[top >> 5] >> (top & 31)) & 1u)) top--;
    SpPoint acc = pt;
    for (int b = top - 1; b >= 0; b--) {
        acc = sp_dbl(acc);
        if ((k.v[b >> 5] >> (b & 31)) & 1u) acc = sp_add_points(acc, pt);
    }
    return acc;
}

ZK_HD Fr sp_load_le(const uint8_t* p) {  // 32 little-endian bytes, 4-byte aligned
    Fr r;
    const u32* w = (const u32*)p;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = w[j];
    return r;
}
ZK_HD Fr sp_load_be(const uint8_t* p) {
    Fr r;
    const u32* w = (const u32*)p;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32 x = w[7 - j];
        r.v[j] = (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24);
    }
    return r;
}

enum { ECDSA_OK = 0, ECDSA_NOT_VERIFIED = 1 };
#define ECDSA_BAD_SIGNATURE ZK_CODE(ZK_UNSUPPORTED, 1)   // eth_keys BadSignature: no class of its own on the wire
#define ECDSA_KEY_RANGE ZK_CODE(ZK_UNSUPPORTED, 2)       // public-key coordinate >= P: outside the engine's domain

struct EcdsaArgs {
    const uint8_t* bytes;  // per signature: pk_x LE, pk_y LE, msg_hash (BE or LE), sig_r LE, sig_s LE (32 bytes each)
    u64 stride;            // bytes between signatures
    u32 off[5];            // byte offsets of the five fields
    u32 msg_be;            // 1: msg_hash bytes are big-endian (util/ec.py:93), 0: little-endian (tx_circuit.py:131)
    const u32* v;          // optional recovery ids, v[i * v_stride] (Sig circuit): outside {0, 1} -> BadSignature
    u32 v_stride;
    u64 n;
    u32* out;              // optional: out[i * out_stride] = status (e.g. the sign units' meta column)
    u32 out_stride;
};

// `ecdsa_status` of signature i: 0 verified, 1 not verified, else the status code of the exception
ZK_HD u32 ecdsa_verify_one(const EcdsaArgs& a, u64 i) {
    const uint8_t* base = a.bytes + i * a.stride;
    const Fr pkx = sp_load_le(base + a.off[0]), pky = sp_load_le(base + a.off[1]);
    const Fr z = a.msg_be ? sp_load_be(base + a.off[2]) : sp_load_le(base + a.off[2]);
    const Fr r = sp_load_le(base + a.off[3]), s = sp_load_le(base + a.off[4]);
    const Fr n = SecpN::mod(), p = SecpP::mod();
    if (a.v && a.v[i * a.v_stride] > 1u) return ECDSA_BAD_SIGNATURE;
    // validate_signature_r_or_s: 0 < value < N
    if (!fr_lt(r, n) || !fr_lt(s, n) || fr_is_zero(r) || fr_is_zero(s)) return ECDSA_BAD_SIGNATURE;
    if (!fr_lt(pkx, p) || !fr_lt(pky, p)) return ECDSA_KEY_RANGE;
    const Fr wM = sp_inv<SecpN>(sp_to_mont<SecpN>(s));
    const Fr u1 = sp_mont<SecpN>(sp_reduce_once<SecpN>(z), wM);  // z * w mod N (canonical: one operand in Montgomery form)
    const Fr u2 = sp_mont<SecpN>(r, wM);
    SpPoint q;
    q.X = pkx; q.Y = pky; q.Z = SecpP::one();
    const SpPoint A = sp_scalar_mul_g(u1);
    const SpPoint B = sp_scalar_mul(q, u2);
    // fast_add(from_jacobian(A), from_jacobian(B)): the case analysis is projective, so A and B are added as they are
    const SpPoint C = sp_add_points(A, B);
    // r == from_jacobian(C).x with inv(0) == 0: Z == 0 gives x = 0 != r; else r == X / Z^2  <=>  r Z^2 == X (r < N < P)
    if (fr_is_zero(C.Z)) return ECDSA_NOT_VERIFIED;
    return fr_eq(sp_mont<SecpP>(r, sp_mont<SecpP>(C.Z, C.Z)), C.X) ? ECDSA_OK : ECDSA_NOT_VERIFIED;
}
