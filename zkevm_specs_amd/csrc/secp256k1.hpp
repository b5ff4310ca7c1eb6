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

// 512-bit t -> t mod P for P = 2^256 - 2^32 - 977 (result < P): the high half is folded twice with 2^256 = 2^32 + 977 (mod P)
ZK_HD Fr sp_fold_p(const u32* t) {
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
#ifndef ZK_HOSTSIM
// ---- 8 x 8 limb product by product scanning (round 3).  Column k of the 512-bit product is the sum of the a[i] * b[j] with
// i + j = k, accumulated in 96 bits (acc: 64, carry: the overflow count).  The compiler's code for the row-wise form above was
// 397 instructions per product, half of them v_mov (register-pair shuffling around v_mad_u64_u32) plus a 64-bit add per
// multiply-add; here a multiply-add is v_mad_u64_u32 with the carry-out in an SGPR pair + one v_addc on that pair — the
// carry-outs of a column all land before the first v_addc reads one (gfx950 wants two wait states between a VALU write of an
// SGPR and a VALU read of it as carry-in; the one- and two-product columns pad with s_nop).
__device__ __forceinline__ void sp_col1(u64& acc, u32& cy, u32 a0, u32 b0) {
    u64 c0;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0)
        : [a0] "v"(a0), [b0] "v"(b0));
}
__device__ __forceinline__ void sp_col2(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1) {
    u64 c0, c1;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1));
}
__device__ __forceinline__ void sp_col3(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2) {
    u64 c0, c1, c2;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c2], %[a2], %[b2], %[acc]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c2], 0, %[cy], %[c2]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2));
}
__device__ __forceinline__ void sp_col4(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3) {
    u64 c0, c1, c2, c3;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c2], %[a2], %[b2], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c3], %[a3], %[b3], %[acc]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c2], 0, %[cy], %[c2]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c3], 0, %[cy], %[c3]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [a3] "v"(a3), [b3] "v"(b3));
}
__device__ __forceinline__ void sp_col5(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3, u32 a4, u32 b4) {
    u64 c0, c1, c2, c3, c4;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c2], %[a2], %[b2], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c3], %[a3], %[b3], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c4], %[a4], %[b4], %[acc]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c2], 0, %[cy], %[c2]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c3], 0, %[cy], %[c3]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c4], 0, %[cy], %[c4]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [a3] "v"(a3), [b3] "v"(b3), [a4] "v"(a4), [b4] "v"(b4));
}
__device__ __forceinline__ void sp_col6(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3, u32 a4, u32 b4, u32 a5, u32 b5) {
    u64 c0, c1, c2, c3, c4, c5;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c2], %[a2], %[b2], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c3], %[a3], %[b3], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c4], %[a4], %[b4], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c5], %[a5], %[b5], %[acc]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c2], 0, %[cy], %[c2]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c3], 0, %[cy], %[c3]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c4], 0, %[cy], %[c4]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c5], 0, %[cy], %[c5]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4), [c5] "=&s"(c5)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [a3] "v"(a3), [b3] "v"(b3), [a4] "v"(a4), [b4] "v"(b4), [a5] "v"(a5), [b5] "v"(b5));
}
__device__ __forceinline__ void sp_col7(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3, u32 a4, u32 b4, u32 a5, u32 b5, u32 a6, u32 b6) {
    u64 c0, c1, c2, c3, c4, c5, c6;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c2], %[a2], %[b2], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c3], %[a3], %[b3], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c4], %[a4], %[b4], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c5], %[a5], %[b5], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c6], %[a6], %[b6], %[acc]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c2], 0, %[cy], %[c2]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c3], 0, %[cy], %[c3]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c4], 0, %[cy], %[c4]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c5], 0, %[cy], %[c5]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c6], 0, %[cy], %[c6]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4), [c5] "=&s"(c5), [c6] "=&s"(c6)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [a3] "v"(a3), [b3] "v"(b3), [a4] "v"(a4), [b4] "v"(b4), [a5] "v"(a5), [b5] "v"(b5), [a6] "v"(a6), [b6] "v"(b6));
}
__device__ __forceinline__ void sp_col8(u64& acc, u32& cy, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3, u32 a4, u32 b4, u32 a5, u32 b5, u32 a6, u32 b6, u32 a7, u32 b7) {
    u64 c0, c1, c2, c3, c4, c5, c6, c7;
    asm("v_mad_u64_u32 %[acc], %[c0], %[a0], %[b0], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c1], %[a1], %[b1], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c2], %[a2], %[b2], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c3], %[a3], %[b3], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c4], %[a4], %[b4], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c5], %[a5], %[b5], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c6], %[a6], %[b6], %[acc]\n\t"
        "v_mad_u64_u32 %[acc], %[c7], %[a7], %[b7], %[acc]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c0], 0, %[cy], %[c0]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c1], 0, %[cy], %[c1]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c2], 0, %[cy], %[c2]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c3], 0, %[cy], %[c3]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c4], 0, %[cy], %[c4]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c5], 0, %[cy], %[c5]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c6], 0, %[cy], %[c6]\n\t"
        "v_addc_co_u32_e64 %[cy], %[c7], 0, %[cy], %[c7]"
        : [acc] "+v"(acc), [cy] "+v"(cy), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4), [c5] "=&s"(c5), [c6] "=&s"(c6), [c7] "=&s"(c7)
        : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [a3] "v"(a3), [b3] "v"(b3), [a4] "v"(a4), [b4] "v"(b4), [a5] "v"(a5), [b5] "v"(b5), [a6] "v"(a6), [b6] "v"(b6), [a7] "v"(a7), [b7] "v"(b7));
}
// t[0..15] = a * b (8 x 8 limbs)
__device__ __forceinline__ void sp_mul_wide(const Fr& a, const Fr& b, u32* t) {
    u64 acc = 0;
    u32 cy = 0;
    sp_col1(acc, cy, a.v[0], b.v[0]);
    t[0] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col2(acc, cy, a.v[0], b.v[1], a.v[1], b.v[0]);
    t[1] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col3(acc, cy, a.v[0], b.v[2], a.v[1], b.v[1], a.v[2], b.v[0]);
    t[2] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col4(acc, cy, a.v[0], b.v[3], a.v[1], b.v[2], a.v[2], b.v[1], a.v[3], b.v[0]);
    t[3] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col5(acc, cy, a.v[0], b.v[4], a.v[1], b.v[3], a.v[2], b.v[2], a.v[3], b.v[1], a.v[4], b.v[0]);
    t[4] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col6(acc, cy, a.v[0], b.v[5], a.v[1], b.v[4], a.v[2], b.v[3], a.v[3], b.v[2], a.v[4], b.v[1], a.v[5], b.v[0]);
    t[5] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col7(acc, cy, a.v[0], b.v[6], a.v[1], b.v[5], a.v[2], b.v[4], a.v[3], b.v[3], a.v[4], b.v[2], a.v[5], b.v[1], a.v[6], b.v[0]);
    t[6] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col8(acc, cy, a.v[0], b.v[7], a.v[1], b.v[6], a.v[2], b.v[5], a.v[3], b.v[4], a.v[4], b.v[3], a.v[5], b.v[2], a.v[6], b.v[1], a.v[7], b.v[0]);
    t[7] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col7(acc, cy, a.v[1], b.v[7], a.v[2], b.v[6], a.v[3], b.v[5], a.v[4], b.v[4], a.v[5], b.v[3], a.v[6], b.v[2], a.v[7], b.v[1]);
    t[8] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col6(acc, cy, a.v[2], b.v[7], a.v[3], b.v[6], a.v[4], b.v[5], a.v[5], b.v[4], a.v[6], b.v[3], a.v[7], b.v[2]);
    t[9] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col5(acc, cy, a.v[3], b.v[7], a.v[4], b.v[6], a.v[5], b.v[5], a.v[6], b.v[4], a.v[7], b.v[3]);
    t[10] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col4(acc, cy, a.v[4], b.v[7], a.v[5], b.v[6], a.v[6], b.v[5], a.v[7], b.v[4]);
    t[11] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col3(acc, cy, a.v[5], b.v[7], a.v[6], b.v[6], a.v[7], b.v[5]);
    t[12] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col2(acc, cy, a.v[6], b.v[7], a.v[7], b.v[6]);
    t[13] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    sp_col1(acc, cy, a.v[7], b.v[7]);
    t[14] = (u32)acc; acc = (acc >> 32) | ((u64)cy << 32); cy = 0;
    t[15] = (u32)acc;
}
#endif
#ifndef ZK_HOSTSIM
// ---- the fold as explicit carry chains (round 4).  The C form above compiled to ~160 instructions per product (64-bit adds in
// register pairs around every 977-multiply, a branch for the rare wrap, a compare + subtract + select at the end).  Here:
//   m = hi * 977                      eight v_mad_u64_u32 chained through their high halves (m: nine limbs)
//   r = lo + m + (hi << 32)           two add-with-carry chains over ten limbs
//   s = r[0..7] + v * 977 + (v << 32) v = r[8..9] < 2^34: one 64-bit multiply, one carry chain
//   a carry out of s (s is tiny then) and the final "s >= P" both add 2^32 + 977: s >= P  <=>  s + 2^32 + 977 carries out of 2^256
// ~70 instructions.  Carries live in VCC; every chain is one asm statement.
__device__ __forceinline__ Fr sp_fold_p_chain(const u32* t) {
    u32 m[9];
    {
        u64 c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c = (u64)t[8 + k] * 977ull + (c >> 32);
            m[k] = (u32)c;
        }
        m[8] = (u32)(c >> 32);
    }
    u32 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9;
    // r = lo + m (nine limbs), then += hi << 32 (limbs 1..8, carry into limb 9).  (VOP2: src1 is a VGPR, constants go first)
    asm("v_add_co_u32 %0, vcc, %10, %18\n\t"
        "v_addc_co_u32 %1, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %15, %23, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %16, %24, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %17, %25, vcc\n\t"
        "v_addc_co_u32 %8, vcc, 0, %26, vcc\n\t"
        "v_add_co_u32 %1, vcc, %1, %27\n\t"
        "v_addc_co_u32 %2, vcc, %2, %28, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %29, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %30, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %31, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %32, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %33, vcc\n\t"
        "v_addc_co_u32 %8, vcc, %8, %34, vcc\n\t"
        "v_addc_co_u32_e64 %9, vcc, 0, 0, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9)
        : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
          "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]),
          "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15])
        : "vcc");
    // second fold: v = r8 + r9 * 2^32 (< 2^34): s = r[0..7] + v * 977 + (v << 32)
    const u64 v977 = ((u64)r8 | ((u64)r9 << 32)) * 977ull;  // < 2^44
    const u32 a0 = (u32)v977, a1 = (u32)(v977 >> 32);
    u32 s0, s1, s2, s3, s4, s5, s6, s7, cy, cy2;
    asm("v_add_co_u32 %0, vcc, %10, %18\n\t"
        "v_addc_co_u32 %1, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %12, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %13, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %14, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %15, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %16, vcc\n\t"
        "v_addc_co_u32 %7, vcc, 0, %17, vcc\n\t"
        "v_addc_co_u32_e64 %8, vcc, 0, 0, vcc\n\t"
        "v_add_co_u32 %1, vcc, %1, %20\n\t"      // + (v << 32): limb 1 += r8, limb 2 += r9
        "v_addc_co_u32 %2, vcc, %2, %21, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, 0, %7, vcc\n\t"
        "v_addc_co_u32_e64 %9, vcc, 0, 0, vcc"
        : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5), "=&v"(s6), "=&v"(s7), "=&v"(cy), "=&v"(cy2)
        : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(a0), "v"(a1), "v"(r8), "v"(r9)
        : "vcc");
    // wrapped past 2^256 (at most once over the two chains; s is tiny then): + (2^32 + 977), no further carry.  Then
    // s >= P  <=>  s + 2^32 + 977 >= 2^256: the sum's low 256 bits are s - P
    const u32 w = cy + cy2;  // 0 or 1
    const u32 w977 = w * 977u;
    u32 q0, q1, q2, q3, q4, q5, q6, q7;
    Fr out;
    asm("v_add_co_u32 %16, vcc, %16, %24\n\t"
        "v_addc_co_u32 %17, vcc, %17, %25, vcc\n\t"
        "v_addc_co_u32 %18, vcc, 0, %18, vcc\n\t"
        "v_addc_co_u32 %19, vcc, 0, %19, vcc\n\t"
        "v_addc_co_u32 %20, vcc, 0, %20, vcc\n\t"
        "v_addc_co_u32 %21, vcc, 0, %21, vcc\n\t"
        "v_addc_co_u32 %22, vcc, 0, %22, vcc\n\t"
        "v_addc_co_u32 %23, vcc, 0, %23, vcc\n\t"
        "v_add_co_u32 %0, vcc, 0x3d1, %16\n\t"
        "v_addc_co_u32 %1, vcc, 1, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %18, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %20, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %22, vcc\n\t"
        "v_addc_co_u32 %7, vcc, 0, %23, vcc\n\t"
        "v_cndmask_b32 %8, %16, %0, vcc\n\t"
        "v_cndmask_b32 %9, %17, %1, vcc\n\t"
        "v_cndmask_b32 %10, %18, %2, vcc\n\t"
        "v_cndmask_b32 %11, %19, %3, vcc\n\t"
        "v_cndmask_b32 %12, %20, %4, vcc\n\t"
        "v_cndmask_b32 %13, %21, %5, vcc\n\t"
        "v_cndmask_b32 %14, %22, %6, vcc\n\t"
        "v_cndmask_b32 %15, %23, %7, vcc"
        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7),
          "=&v"(out.v[0]), "=&v"(out.v[1]), "=&v"(out.v[2]), "=&v"(out.v[3]), "=&v"(out.v[4]), "=&v"(out.v[5]), "=&v"(out.v[6]), "=&v"(out.v[7]),
          "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7)
        : "v"(w977), "v"(w)
        : "vcc");
    return out;
}
#endif
// a * b mod P for P = 2^256 - 2^32 - 977 (plain residues, a, b < P, result < P): 8 x 8 schoolbook product, then the high
// half is folded twice with 2^256 = 2^32 + 977 (mod P) — 72 multiply-adds instead of the 128 of a Montgomery product
ZK_HD Fr sp_mul_p(Fr a, Fr b) {  // inlined (round 3): the call's argument / result moves were ~6 % of a product; the kernel grows to 240 KB, measured +3 %
    u32 t[16];
#ifndef ZK_HOSTSIM
    sp_mul_wide(a, b, t);
#ifdef ZK_SECP_FOLD_C
    return sp_fold_p(t);
#else
    return sp_fold_p_chain(t);
#endif
#endif
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
    return sp_fold_p(t);
}
// a^2 mod P: the 28 cross products once, doubled, plus the 8 squares (36 multiply-adds instead of 64), then the same fold
ZK_HD Fr sp_sqr_p(Fr a) {
    u32 t[16];
#ifndef ZK_HOSTSIM
    // (on the device the product-scanning multiplier is shorter than this specialised form: 164 instructions for the product
    // against ~190; a squaring that doubles the cross products pays more per column than it saves)
    sp_mul_wide(a, a, t);
#ifdef ZK_SECP_FOLD_C
    return sp_fold_p(t);
#else
    return sp_fold_p_chain(t);
#endif
#endif
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u64 c = 0;
        const u32 ai = a.v[i];
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            c += (u64)a.v[j] * ai + t[i + j];
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
#pragma unroll
    for (int k = 15; k > 0; k--) t[k] = (t[k] << 1) | (t[k - 1] >> 31);
    t[0] = 0;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 sq = (u64)a.v[i] * a.v[i];
        c += (u64)t[2 * i] + (u32)sq;
        t[2 * i] = (u32)c;
        c >>= 32;
        c += (u64)t[2 * i + 1] + (sq >> 32);
        t[2 * i + 1] = (u32)c;
        c >>= 32;
    }
    return sp_fold_p(t);
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
ZK_HD Fr spf_mul(const Fr& a, const Fr& b) { return sp_mul_p(a, b); }
ZK_HD Fr spf_sqr(const Fr& a) { return sp_sqr_p(a); }
ZK_HD Fr spf_add(const Fr& a, const Fr& b) { return sp_add<SecpP>(a, b); }
ZK_HD Fr spf_sub(const Fr& a, const Fr& b) { return sp_sub<SecpP>(a, b); }
ZK_HD SpPoint sp_infinity() {  // (0, 0, 1)
    SpPoint p;
    p.X = fr_zero(); p.Y = fr_zero(); p.Z = SecpP::one();
    return p;
}
// jacobian_double: Y == 0 -> (0, 0, 0); else "dbl-2009-l" with a = 0, 2M + 5S (any Jacobian doubling formula yields a
// representative of the same point; only the Y == 0 / X-equality tests and the final X / Z^2 are observable)
ZK_HD void sp_dbl_ip(SpPoint& p) {
    if (fr_is_zero(p.Y)) {
        p.X = fr_zero(); p.Z = fr_zero();
        return;
    }
    const Fr A = spf_sqr(p.X), B = spf_sqr(p.Y), C = spf_sqr(B);
    Fr t = spf_add(p.X, B);
    t = spf_sub(spf_sub(spf_sqr(t), A), C);
    const Fr D = spf_add(t, t), E = spf_add(spf_add(A, A), A), Fq = spf_sqr(E);
    const Fr yz = spf_mul(p.Y, p.Z);
    p.X = spf_sub(Fq, spf_add(D, D));
    Fr c8 = spf_add(C, C);
    c8 = spf_add(c8, c8);
    c8 = spf_add(c8, c8);
    p.Y = spf_sub(spf_mul(E, spf_sub(D, p.X)), c8);
    p.Z = spf_add(yz, yz);
}
// jacobian_add: p.Y == 0 -> q; q.Y == 0 -> p; U1 == U2: S1 != S2 -> (0, 0, 1), else double(p); otherwise the chord (12M + 4S)
ZK_HD void sp_add_ip(SpPoint& p, const SpPoint& q) {
    if (fr_is_zero(p.Y)) { p = q; return; }
    if (fr_is_zero(q.Y)) return;
    const Fr z1z1 = spf_sqr(p.Z), z2z2 = spf_sqr(q.Z);
    const Fr u1 = spf_mul(p.X, z2z2), u2 = spf_mul(q.X, z1z1);
    const Fr s1 = spf_mul(spf_mul(p.Y, q.Z), z2z2), s2 = spf_mul(spf_mul(q.Y, p.Z), z1z1);
    if (fr_eq(u1, u2)) {
        if (!fr_eq(s1, s2)) p = sp_infinity();
        else sp_dbl_ip(p);
        return;
    }
    const Fr h = spf_sub(u2, u1), rr = spf_sub(s2, s1);
    const Fr hh = spf_sqr(h), hhh = spf_mul(h, hh), v = spf_mul(u1, hh);
    const Fr zz = spf_mul(p.Z, q.Z);
    p.X = spf_sub(spf_sub(spf_sqr(rr), hhh), spf_add(v, v));
    p.Y = spf_sub(spf_mul(rr, spf_sub(v, p.X)), spf_mul(s1, hhh));
    p.Z = spf_mul(zz, h);
}
// p + (x2, y2, 1), y2 != 0: the fixed-base tables' addition (8M + 3S); same case analysis
ZK_HD void sp_add_affine_ip(SpPoint& p, const Fr& x2, const Fr& y2) {
    if (fr_is_zero(p.Y)) {
        p.X = x2; p.Y = y2; p.Z = SecpP::one();
        return;
    }
    const Fr z1z1 = spf_sqr(p.Z);
    const Fr u2 = spf_mul(x2, z1z1), s2 = spf_mul(spf_mul(y2, p.Z), z1z1);
    if (fr_eq(p.X, u2)) {
        if (!fr_eq(p.Y, s2)) p = sp_infinity();
        else sp_dbl_ip(p);
        return;
    }
    const Fr h = spf_sub(u2, p.X), rr = spf_sub(s2, p.Y);
    const Fr hh = spf_sqr(h), hhh = spf_mul(h, hh), v = spf_mul(p.X, hh);
    const Fr y1hhh = spf_mul(p.Y, hhh);
    p.X = spf_sub(spf_sub(spf_sqr(rr), hhh), spf_add(v, v));
    p.Y = spf_sub(spf_mul(rr, spf_sub(v, p.X)), y1hhh);
    p.Z = spf_mul(p.Z, h);
}
ZK_HD void sp_load_affine(const uint32_t* e, Fr& x, Fr& y) {
#pragma unroll
    for (int q = 0; q < 8; q++) { x.v[q] = e[q]; y.v[q] = e[8 + q]; }
}
// k * G, k < N: sixty-four 4-bit windows over the precomputed multiples d * 16^j * G (secp_g_table.h) — 64 mixed additions.
// G is on the curve, so this is the group element eth-keys' double-and-add produces (k == 0: a Y == 0 point).
// (Only the case-exact path of off-curve keys uses this; curve keys go through ecdsa_partial below.)
ZK_NOINLINE SpPoint sp_scalar_mul_g(Fr k) {
    SpPoint acc = sp_infinity();
    for (int j = 0; j < 64; j++) {
        const u32 d = k.v[0] & 15u;
        if (d) {
            Fr x, y;
            sp_load_affine(secp_g_table[15 * j + (int)d - 1], x, y);
            sp_add_affine_ip(acc, x, y);
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
        sp_dbl_ip(acc);
        if ((k.v[b >> 5] >> (b & 31)) & 1u) sp_add_ip(acc, pt);
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
#define ECDSA_KEY_RANGE ZK_CODE(ZK_UNSUPPORTED, 2)       // public key with y == P exactly: outside the engine's domain (see ecdsa_prepare)
#define ECDSA_PENDING 0xfffffffeu                        // internal: the verdict comes out of the joint multiplication

struct EcdsaArgs {
    const uint8_t* bytes;  // per signature: pk_x LE, pk_y LE, msg_hash (BE or LE), sig_r LE, sig_s LE (32 bytes each)
    u64 stride;            // bytes between signatures
    u32 off[5];            // byte offsets of the five fields
    u32 msg_be;            // 1: msg_hash bytes are big-endian (util/ec.py:93), 0: little-endian (tx_circuit.py:131)
    const u32* v;          // optional recovery ids, v[i * v_stride] (Sig circuit): outside {0, 1} -> BadSignature
    u32 v_stride;
    u64 n;
    u64 first;             // first signature of this launch (large batches run in chunks that share one key table, zk_launch_ecdsa)
    u32* out;              // optional: out[i * out_stride] = status (e.g. the sign units' meta column)
    u32 out_stride;
    u32* qtab;             // per-lane tables of the key's multiples, word w of entry e of lane l at qtab[(e * 24 + w) * qtab_lanes + l]
    u64 qtab_lanes;
    u32 lanes_per_sig;     // 1: one lane runs both halves of the GLV split; 2: a lane pair, one half each; 4: two more lanes for the halves of u1 G (ecdsa_partial4)
    const u32* gcomb;      // optional: the 8-bit fixed-base table of G built on the device (ecdsa_comb_entry), nullptr: 4-bit constant table
    // A second batch in the same launch (zk_ecdsa_open_batches): signatures [n0, n) come from these arrays (the Tx circuit's and
    // the Sig circuit's chips of one block: two launches of 2^14 signatures each ran 1.9 ms side by side where one launch of
    // 2^15 takes 1.4 ms — 1,024 wavefronts placed one per SIMD by ONE dispatch).  n0 == n: a single batch.
    u64 n0;
    const uint8_t* bytes1;
    u64 stride1;
    u32 off1[5];
    u32 msg_be1;
    const u32* v1;
    u32 v_stride1;
    u32* out1;
    u32 out_stride1;
};
#if defined(ZK_HOSTSIM)
static inline
#else
__host__ __device__ inline
#endif
void ecdsa_single_batch(EcdsaArgs& a) {  // callers that fill the first batch only (host code)
    a.n0 = a.n;
    a.gcomb = nullptr;
    a.bytes1 = nullptr; a.stride1 = 0; a.msg_be1 = 0; a.v1 = nullptr; a.v_stride1 = 0; a.out1 = nullptr; a.out_stride1 = 0;
    for (int k = 0; k < 5; k++) a.off1[k] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Joint multiplication u1 G + u2 Q for keys ON the curve (there every correct group law gives eth-keys' point).
//   * GLV: u2 = k1 + k2 lambda with |k1|, |k2| < 2^128 and lambda (x, y) = (beta x, y), so u2 Q = k1 Q + k2 Q' with
//     Q' = (beta Qx, Qy): 128 doublings instead of 256;  u1 = lo + 2^128 hi goes with the fixed points G and 2^128 G.
//   * 4-bit windows, most significant first: 32 steps of 4 doublings + one addition per (scalar, base) pair: the key's
//     multiples 1..15 from a per-lane table in HBM (laid out so that a wavefront's loads coalesce; each entry is requested
//     before the four doublings that precede its use), G's from two 15-entry constant tables (mixed additions).
//   * "role" h = 0: (k1, Q) and (lo, G); h = 1: (k2, Q') and (hi, 2^128 G).  One lane runs both roles in one loop, or a
//     lane pair runs one role each and the two partial sums are added at the end (half the dependent chain per lane: the
//     small batches of the Tx circuit are latency-bound, not throughput-bound).
//   * no inversion at the end: r == X / Z^2  <=>  r Z^2 == X.
// ---------------------------------------------------------------------------------------------------------------------
FR_CONST_ARR(secp_beta, SECP_BETA_LIMBS)
FR_CONST_ARR(secp_glv_a1, SECP_GLV_A1_LIMBS)
FR_CONST_ARR(secp_glv_mb1, SECP_GLV_MB1_LIMBS)
FR_CONST_ARR(secp_glv_a2, SECP_GLV_A2_LIMBS)
FR_CONST_ARR(secp_glv_b2, SECP_GLV_B2_LIMBS)
FR_CONST_ARR(secp_glv_g1, SECP_GLV_G1_LIMBS)
FR_CONST_ARR(secp_glv_g2, SECP_GLV_G2_LIMBS)

// round(a * g / 2^384) for 256-bit a, g (the result has at most 128 bits)
ZK_NOINLINE Fr sp_mul_shift384(Fr a, Fr g) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
        const u32 gi = g.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)a.v[j] * gi + t[i + j];
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
    Fr r = fr_zero();
    u64 c = t[11] >> 31;  // rounding bit 383
#pragma unroll
    for (int k = 0; k < 4; k++) {
        c += t[12 + k];
        r.v[k] = (u32)c;
        c >>= 32;
    }
    r.v[4] = (u32)c;
    return r;
}
// low 256 bits of a * b for a < 2^160, b < 2^160 (five limbs each; two's-complement arithmetic does the rest)
ZK_HD Fr sp_mul_lo_5x5(const Fr& a, const Fr& b) {
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            c += (u64)a.v[j] * b.v[i] + t[i + j];
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 5] = (u32)c;
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}
// k (< N) -> |k1|, |k2| (128 bits each) and their signs; false if a magnitude does not fit 128 bits (never for k < N, but the
// caller then takes the plain path instead of trusting a bound)
ZK_HD bool sp_glv_split(const Fr& k, Fr& k1, u32& neg1, Fr& k2, u32& neg2) {
    const Fr c1 = sp_mul_shift384(k, secp_glv_g1()), c2 = sp_mul_shift384(k, secp_glv_g2());
    Fr t, u;
    u256_sub(t, k, sp_mul_lo_5x5(c1, secp_glv_a1()));       // mod 2^256, two's complement
    u256_sub(k1, t, sp_mul_lo_5x5(c2, secp_glv_a2()));
    u = sp_mul_lo_5x5(c1, secp_glv_mb1());
    u256_sub(k2, u, sp_mul_lo_5x5(c2, secp_glv_b2()));
    neg1 = k1.v[7] >> 31;
    neg2 = k2.v[7] >> 31;
    const Fr z = fr_zero();
    if (neg1) { Fr m; u256_sub(m, z, k1); k1 = m; }
    if (neg2) { Fr m; u256_sub(m, z, k2); k2 = m; }
    return (k1.v[4] | k1.v[5] | k1.v[6] | k1.v[7] | k2.v[4] | k2.v[5] | k2.v[6] | k2.v[7]) == 0u;
}

// a^-1 mod N for 0 < a < N, canonical in and out: the right-shift binary algorithm with v kept odd.  Invariants x1 a = u and
// x2 a = v (mod N); every round makes u even (if it is odd: swap so that u >= v, subtract) and halves it, so bitlen(u) +
// bitlen(v) drops by at least one per round: at most 512 rounds of ~150 instructions where the Fermat ladder of rounds 1-2 was ~390
// Montgomery products of ~600 instructions (a quarter of the whole verification).  Branch-free inside a round; the lanes of a wavefront leave
// the loop together (the slowest one's round count).
ZK_HD Fr sp_inv_n_binary(const Fr& a) {
    const Fr n = SecpN::mod();
    Fr u = a, v = n, x1 = fr_zero(), x2 = fr_zero();
    x1.v[0] = 1u;
    for (int round = 0; round < 512 && !fr_is_zero(u); round++) {
        const u32 odd = u.v[0] & 1u;
        const bool sw = odd && fr_lt(u, v);
#pragma unroll
        for (int k = 0; k < 8; k++) {  // conditional swap (u, x1) <-> (v, x2)
            const u32 tu = u.v[k], tv = v.v[k], t1 = x1.v[k], t2 = x2.v[k];
            u.v[k] = sw ? tv : tu; v.v[k] = sw ? tu : tv;
            x1.v[k] = sw ? t2 : t1; x2.v[k] = sw ? t1 : t2;
        }
        // u odd: u -= v (>= 0, even), x1 -= x2 (mod N)
        Fr vm, xm;
        const u32 m = 0u - odd;
#pragma unroll
        for (int k = 0; k < 8; k++) { vm.v[k] = v.v[k] & m; xm.v[k] = x2.v[k] & m; }
        Fr d;
        u256_sub(d, u, vm);
        u = d;
        if (u256_sub(d, x1, xm)) {  // went below zero: + N
            Fr e;
            u256_add(e, d, n);
            d = e;
        }
        x1 = d;
        // halve u; halve x1 mod N (x1 odd: (x1 + N) / 2, a 257-bit sum)
#pragma unroll
        for (int k = 0; k < 7; k++) u.v[k] = (u.v[k] >> 1) | (u.v[k + 1] << 31);
        u.v[7] >>= 1;
        Fr nm;
        const u32 mx = 0u - (x1.v[0] & 1u);
#pragma unroll
        for (int k = 0; k < 8; k++) nm.v[k] = n.v[k] & mx;
        Fr h;
        const u32 carry = u256_add(h, x1, nm);
#pragma unroll
        for (int k = 0; k < 7; k++) x1.v[k] = (h.v[k] >> 1) | (h.v[k + 1] << 31);
        x1.v[7] = (h.v[7] >> 1) | (carry << 31);
    }
    return x2;  // u == 0: v == gcd == 1 (N is prime), x2 a == 1
}

// a^-1 mod N by Bernstein-Yang division steps ("Fast constant-time gcd computation and modular inversion", TCHES 2019), in the
// batched form with 30-bit signed limbs: 20 x (30 divsteps on the low words of f, g -> a 2 x 2 transition matrix scaled by 2^30;
// the matrix applied to (f, g) and, modulo N, to (d, e)) = 600 divsteps, enough for any 256-bit input (590 are).  zeta = -(delta +
// 1/2) starts at -1; afterwards g == 0, f == +-1 and d == +-a^-1.  Branch-free: the lanes of a wavefront do identical work, about
// 22k instructions against ~60-77k for the binary algorithm above (its round count is the slowest lane's).
// oracle/ecdsa_oracle.py uses pow(x, -1, N); tests/test_ecdsa.py pins both forms to it.
ZK_HD Fr sp_inv_n_safegcd(const Fr& a) {
    const int32_t M30 = (int32_t)0x3fffffff;
    const int32_t NL[9] = {0x10364141, 0x3f497a33, 0x348a03bb, 0x2bb739ab, 0x3ffffeba, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff};  // N in 30-bit limbs
    const u32 NINV30 = 0x2a774ec1u;  // N^-1 mod 2^30
    int32_t d[9], e[9], f[9], g[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d[i] = 0; e[i] = i == 0 ? 1 : 0; f[i] = NL[i];
        // bits [30 i, 30 i + 30) of a
        const int lo = 30 * i, w = lo >> 5, sh = lo & 31;
        u32 x = a.v[w] >> sh;
        if (sh > 2 && w + 1 < 8) x |= a.v[w + 1] << (32 - sh);
        g[i] = (int32_t)(x & (u32)M30);
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        // 30 division steps on the low words: (u, v; q, r) with u f0 + v g0 = f << 30 etc.
        u32 u = 1, v = 0, q = 0, r = 0 + 1, ff = (u32)f[0], gg = (u32)g[0];
#pragma unroll
        for (int k = 0; k < 30; k++) {
            u32 c1 = (u32)(zeta >> 31);
            const u32 c2 = 0u - (gg & 1u);
            const u32 x = (ff ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
            gg += x & c2; q += y & c2; r += z & c2;
            c1 &= c2;
            zeta = (int32_t)(((u32)zeta ^ c1) - 1u);
            ff += gg & c1; u += q & c1; v += r & c1;
            gg >>= 1; u <<= 1; v <<= 1;
        }
        const int64_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        {   // (d, e) <- (t / 2^30) (d, e) mod N: multiples of N make the low 30 bits vanish
            const int32_t sd = d[8] >> 31, se = e[8] >> 31;
            int32_t md = ((int32_t)tu & sd) + ((int32_t)tv & se), me = ((int32_t)tq & sd) + ((int32_t)tr & se);
            int64_t cd = tu * d[0] + tv * e[0], ce = tq * d[0] + tr * e[0];
            md -= (int32_t)((NINV30 * (u32)cd + (u32)md) & (u32)M30);
            me -= (int32_t)((NINV30 * (u32)ce + (u32)me) & (u32)M30);
            cd += (int64_t)NL[0] * md; ce += (int64_t)NL[0] * me;
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                cd += tu * d[i] + tv * e[i] + (int64_t)NL[i] * md;
                ce += tq * d[i] + tr * e[i] + (int64_t)NL[i] * me;
                d[i - 1] = (int32_t)cd & M30; cd >>= 30;
                e[i - 1] = (int32_t)ce & M30; ce >>= 30;
            }
            d[8] = (int32_t)cd; e[8] = (int32_t)ce;
        }
        {   // (f, g) <- (t / 2^30) (f, g): exact
            int64_t cf = tu * f[0] + tv * g[0], cg = tq * f[0] + tr * g[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                cf += tu * f[i] + tv * g[i];
                cg += tq * f[i] + tr * g[i];
                f[i - 1] = (int32_t)cf & M30; cf >>= 30;
                g[i - 1] = (int32_t)cg & M30; cg >>= 30;
            }
            f[8] = (int32_t)cf; g[8] = (int32_t)cg;
        }
    }
    // d in (-2N, N), the inverse up to the sign of f: add N if negative, negate if f < 0, carry, add N once more if still negative
    {
        const int32_t ca = d[8] >> 31, cn = f[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = ((d[i] + (NL[i] & ca)) ^ cn) - cn;
#pragma unroll
        for (int i = 0; i < 8; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
        const int32_t cb = d[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] += NL[i] & cb;
#pragma unroll
        for (int i = 0; i < 8; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    }
    Fr out;
#pragma unroll
    for (int w = 0; w < 8; w++) {  // bits [32 w, 32 w + 32) from the 30-bit limbs
        const int lo = 32 * w, i = lo / 30, sh = lo - 30 * i;
        u32 x = (u32)d[i] >> sh;
        x |= (u32)d[i + 1] << (30 - sh);
        if (30 - sh + 30 < 32 && i + 2 < 9) x |= (u32)d[i + 2] << (60 - sh);
        out.v[w] = x;
    }
    return out;
}

struct EcdsaPrep {
    Fr r;           // signature r (canonical)
    Fr qx, qy;      // public key
    Fr kq[2];       // |k1|, |k2|
    Fr kg[2];       // u1 mod 2^128, u1 >> 128
    u32 neg[2];
};
ZK_HD u32 sp_digit4(const Fr& k, int w) { return (k.v[w >> 3] >> ((w & 7) * 4)) & 15u; }

// Validation + scalars.  ECDSA_PENDING: the key is on the curve and `pr` is filled for ecdsa_partial; any other value is
// the final status (the case-exact path ran here).
ZK_HD u32 ecdsa_prepare(const EcdsaArgs& a, u64 i, EcdsaPrep& pr, bool run_exact_path) {
    const bool second = i >= a.n0;
    const u64 li = second ? i - a.n0 : i;
    const uint8_t* base = second ? a.bytes1 + li * a.stride1 : a.bytes + li * a.stride;
    const u32* off = second ? a.off1 : a.off;
    Fr pkx = sp_load_le(base + off[0]), pky = sp_load_le(base + off[1]);
    const Fr z = (second ? a.msg_be1 : a.msg_be) ? sp_load_be(base + off[2]) : sp_load_le(base + off[2]);
    const Fr r = sp_load_le(base + off[3]), s = sp_load_le(base + off[4]);
    const Fr n = SecpN::mod(), p = SecpP::mod();
    const u32* vv = second ? a.v1 : a.v;
    if (vv && vv[li * (second ? a.v_stride1 : a.v_stride)] > 1u) return ECDSA_BAD_SIGNATURE;
    // validate_signature_r_or_s: 0 < value < N
    if (!fr_lt(r, n) || !fr_lt(s, n) || fr_is_zero(r) || fr_is_zero(s)) return ECDSA_BAD_SIGNATURE;
    // A coordinate >= P (the native backend does not range-check public-key bytes): every formula of eth-keys' Jacobian chain
    // reduces mod P as it goes, so the verdict is that of the coordinates mod P — with ONE exception, `if not p[1]` (the
    // point-at-infinity test of jacobian_double / jacobian_add), which sees y == P as non-zero: that single value stays outside
    // the domain.  (2^256 < 2 P: one conditional subtraction reduces.)
    if (fr_eq(pky, p)) return ECDSA_KEY_RANGE;
    pkx = sp_reduce_once<SecpP>(pkx);
    pky = sp_reduce_once<SecpP>(pky);
#ifdef ZK_SECP_INV_BINARY
    const Fr wM = sp_to_mont<SecpN>(sp_inv_n_binary(s));
#else
    const Fr wM = sp_to_mont<SecpN>(sp_inv_n_safegcd(s));
#endif
    const Fr u1 = sp_mont<SecpN>(sp_reduce_once<SecpN>(z), wM);  // z * w mod N (canonical: one operand in Montgomery form)
    const Fr u2 = sp_mont<SecpN>(r, wM);
    // y^2 == x^3 + 7 ?
    Fr seven = fr_zero();
    seven.v[0] = 7u;
    const bool on_curve = fr_eq(spf_sqr(pky), spf_add(spf_mul(spf_sqr(pkx), pkx), seven));
    pr.r = r; pr.qx = pkx; pr.qy = pky;
    if (on_curve && sp_glv_split(u2, pr.kq[0], pr.neg[0], pr.kq[1], pr.neg[1])) {
        pr.kg[0] = fr_zero(); pr.kg[1] = fr_zero();
#pragma unroll
        for (int q = 0; q < 4; q++) { pr.kg[0].v[q] = u1.v[q]; pr.kg[1].v[q] = u1.v[4 + q]; }
        return ECDSA_PENDING;
    }
    if (!run_exact_path) return ECDSA_NOT_VERIFIED;  // the partner lane of a pair: lane 0 carries the verdict
    // public key not on the curve: eth-keys' own chain, step by step
    SpPoint q;
    q.X = pkx; q.Y = pky; q.Z = SecpP::one();
    SpPoint C = sp_scalar_mul_g(u1);
    const SpPoint B = sp_scalar_mul(q, u2);
    // fast_add(from_jacobian(A), from_jacobian(B)): the case analysis is projective, so A and B are added as they are
    sp_add_ip(C, B);
    // r == from_jacobian(C).x with inv(0) == 0: Z == 0 gives x = 0 != r; else r == X / Z^2  <=>  r Z^2 == X (r < N < P)
    if (fr_is_zero(C.Z)) return ECDSA_NOT_VERIFIED;
    return fr_eq(spf_mul(r, spf_sqr(C.Z)), C.X) ? ECDSA_OK : ECDSA_NOT_VERIFIED;
}

ZK_HD void sp_tab_store(u32* tab, u64 stride, int e, const SpPoint& p) {
    u32* t = tab + (u64)e * 24 * stride;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        t[(u64)w * stride] = p.X.v[w];
        t[(u64)(8 + w) * stride] = p.Y.v[w];
        t[(u64)(16 + w) * stride] = p.Z.v[w];
    }
}
ZK_HD SpPoint sp_tab_load(const u32* tab, u64 stride, int e) {
    const u32* t = tab + (u64)e * 24 * stride;
    SpPoint p;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        p.X.v[w] = t[(u64)w * stride];
        p.Y.v[w] = t[(u64)(8 + w) * stride];
        p.Z.v[w] = t[(u64)(16 + w) * stride];
    }
    return p;
}
// Fixed-base comb for the u1 G half (round 4).  u1 = sum over 32 byte windows W of d_W * 2^(8 W): with the table
// T[W][d - 1] = d * 2^(8 W) * G (affine, 32 x 255 entries of 64 bytes = 522 KB, L2-resident) u1 G is 32 mixed additions and NO
// doublings — it no longer rides in the key's doubling chain, and a role (16 windows) adds 16 points where the 4-bit interleaved
// form added ~30.  The table is built once per device by ecdsa_comb_entry (one lane per entry: d * 2^(8 W) * G through the
// 4-bit constant table, then to affine with one Fermat inversion); without it (CPU builds) the same comb runs over the 4-bit
// constant table secp_g_table (64 windows).
#define ECDSA_COMB_WINDOWS 32
#define ECDSA_COMB_ENTRIES (ECDSA_COMB_WINDOWS * 255)
ZK_HD void ecdsa_comb_entry(u32 e, u32* table) {  // e = W * 255 + (d - 1)
    const u32 W = e / 255u, d = e % 255u + 1u;
    Fr k = fr_zero();
    k.v[W >> 2] = d << (8u * (W & 3u));  // d * 2^(8 W) < 2^256, below N for every W < 32 (255 * 2^248 < N)
    const SpPoint pt = sp_scalar_mul_g(k);
    const Fr zi = sp_inv<SecpP>(pt.Z), zi2 = spf_sqr(zi);
    const Fr x = spf_mul(pt.X, zi2), y = spf_mul(pt.Y, spf_mul(zi2, zi));
    u32* out = table + (u64)e * 16;
#pragma unroll
    for (int q = 0; q < 8; q++) { out[q] = x.v[q]; out[8 + q] = y.v[q]; }
}
// acc += kg * base_h for the role h (kg < 2^128: windows 16 h .. 16 h + 15 of u1)
ZK_HD void ecdsa_add_g_half(SpPoint& acc, const Fr& kg, int h, const u32* gcomb) {
    if (gcomb) {
        for (int w = 0; w < 16; w++) {
            const u32 d = (kg.v[w >> 2] >> (8 * (w & 3))) & 0xffu;
            if (d) {
                Fr x, y;
                sp_load_affine(gcomb + ((u64)(16 * h + w) * 255u + (d - 1u)) * 16u, x, y);
                sp_add_affine_ip(acc, x, y);
            }
        }
        return;
    }
    for (int j = 0; j < 32; j++) {
        const u32 d = (kg.v[j >> 3] >> (4 * (j & 7))) & 15u;
        if (d) {
            Fr x, y;
            sp_load_affine(secp_g_table[15 * (32 * h + j) + (int)d - 1], x, y);
            sp_add_affine_ip(acc, x, y);
        }
    }
}
// Partial sum of the roles [h_lo, h_hi] (0-0, 1-1 or 0-1).  `tab`: this lane's table (15 entries x 24 words, stride apart).
// The table holds the multiples of the FIRST role's base with that role's sign; the second role of a one-lane run derives
// its entries from it (X times beta, Y negated when the two signs differ).
ZK_HD SpPoint ecdsa_partial(const EcdsaPrep& pr, int h_lo, int h_hi, u32* tab, u64 stride, const u32* gcomb = nullptr) {
    const Fr beta = secp_beta();
    {   // entry e - 1 = e * base: 2k = double(k), 2k + 1 = 2k + base (mixed)
        Fr bx = h_lo == 1 ? spf_mul(pr.qx, beta) : pr.qx;
        Fr by = pr.neg[h_lo] ? spf_sub(fr_zero(), pr.qy) : pr.qy;
        SpPoint b;
        b.X = bx; b.Y = by; b.Z = SecpP::one();
        sp_tab_store(tab, stride, 0, b);
        for (int k = 1; k <= 7; k++) {
            SpPoint d = sp_tab_load(tab, stride, k - 1);
            sp_dbl_ip(d);
            sp_tab_store(tab, stride, 2 * k - 1, d);
            sp_add_affine_ip(d, bx, by);
            sp_tab_store(tab, stride, 2 * k, d);
        }
    }
    const bool flip = h_hi != h_lo && pr.neg[0] != pr.neg[1];
    SpPoint acc = sp_infinity();
    for (int w = 31; w >= 0; w--) {
        if (w != 31) {
            for (int k = 0; k < 4; k++) sp_dbl_ip(acc);
        }
        for (int h = h_lo; h <= h_hi; h++) {
            const u32 d = sp_digit4(pr.kq[h], w);
            if (d) {
                SpPoint t = sp_tab_load(tab, stride, (int)d - 1);
                if (h != h_lo) {
                    t.X = spf_mul(t.X, beta);
                    if (flip) t.Y = spf_sub(fr_zero(), t.Y);
                }
                sp_add_ip(acc, t);
            }
        }
    }
    for (int h = h_lo; h <= h_hi; h++) ecdsa_add_g_half(acc, pr.kg[h], h, gcomb);
    return acc;
}
// Four lanes per signature (small batches: the chip is far from full, the chain is what counts): roles 0 / 1 ride the two GLV halves of
// u2 Q through the windowed ladder as above WITHOUT their half of u1 G; roles 2 / 3 carry that half — sixteen comb additions each —
// and make them in the ladder's own addition slots (one sp_add_ip call site for the whole wavefront: the comb lanes sit out the
// doublings and hand their point to the add the ladder lanes execute anyway).  The sixteen mixed additions a lane pair made behind its
// ladder (176 of ~1,700 dependent products) disappear from the chain; the price is one more addition when the four partial sums meet.
ZK_HD SpPoint ecdsa_partial4(const EcdsaPrep& pr, int role, u32* tab, u64 stride, const u32* gcomb) {
    if (role < 2) {  // the table of this half's base, as in ecdsa_partial
        const Fr beta = secp_beta();
        Fr bx = role == 1 ? spf_mul(pr.qx, beta) : pr.qx;
        Fr by = pr.neg[role] ? spf_sub(fr_zero(), pr.qy) : pr.qy;
        SpPoint b;
        b.X = bx; b.Y = by; b.Z = SecpP::one();
        sp_tab_store(tab, stride, 0, b);
        for (int k = 1; k <= 7; k++) {
            SpPoint d = sp_tab_load(tab, stride, k - 1);
            sp_dbl_ip(d);
            sp_tab_store(tab, stride, 2 * k - 1, d);
            sp_add_affine_ip(d, bx, by);
            sp_tab_store(tab, stride, 2 * k, d);
        }
    }
    SpPoint acc = sp_infinity();
    for (int w = 31; w >= 0; w--) {
        if (role < 2 && w != 31) {
            for (int k = 0; k < 4; k++) sp_dbl_ip(acc);
        }
        SpPoint t = sp_infinity();
        bool have = false;
        if (role < 2) {
            const u32 d = sp_digit4(pr.kq[role], w);
            if (d) { t = sp_tab_load(tab, stride, (int)d - 1); have = true; }
        } else if (w < 16) {
            const Fr& kg = pr.kg[role - 2];
            const u32 d = (kg.v[w >> 2] >> (8 * (w & 3))) & 0xffu;
            if (d) {
                sp_load_affine(gcomb + ((u64)(16 * (role - 2) + w) * 255u + (d - 1u)) * 16u, t.X, t.Y);
                t.Z = SecpP::one();
                have = true;
            }
        }
        if (have) sp_add_ip(acc, t);
    }
    return acc;
}
ZK_HD u32 ecdsa_verdict(const EcdsaPrep& pr, const SpPoint& C) {
    if (fr_is_zero(C.Z)) return ECDSA_NOT_VERIFIED;
    return fr_eq(spf_mul(pr.r, spf_sqr(C.Z)), C.X) ? ECDSA_OK : ECDSA_NOT_VERIFIED;
}

// `ecdsa_status` of signature i with one lane: 0 verified, 1 not verified, else the status code of the exception
ZK_HD u32 ecdsa_verify_one(const EcdsaArgs& a, u64 i, u32* tab, u64 stride) {
    EcdsaPrep pr;
    const u32 st = ecdsa_prepare(a, i, pr, true);
    if (st != ECDSA_PENDING) return st;
    return ecdsa_verdict(pr, ecdsa_partial(pr, 0, 1, tab, stride, a.gcomb));
}
