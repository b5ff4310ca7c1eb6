// State (RW-consistency) circuit: per-row constraint evaluation.
//
// Reference: src/zkevm_specs/state_circuit.py — `check_state_row` :492-613, per-tag checks
// :216-488, `LowerThanGadget` :195-207, `all_keys_eq` :188, `Tables.mpt_lookup` :165-184;
// the row loop with wrap-around neighbours lives in tests/test_state_circuit.py:26-30.
//
// Witness layout (column-major, 57 cells per row, 32 B per cell):
//   0 rw_counter | 1 is_write | 2 tag | 3 id | 4 address | 5 field_tag | 6,7 storage_key lo,hi
//   8..17 address limbs (16-bit, LE) | 18..49 storage-key bytes (LE)
//   50,51 value lo,hi | 52,53 initial_value lo,hi | 54,55 root lo,hi | 56 lexicographic selector
// flags bit0 = value.is_word, bit1 = initial_value.is_word (WordOrValue, util/arithmetic.py:171).
// MPT table row (row-major, 12 cells): address, proof_type, storage_key lo,hi, root lo,hi,
//   root_prev lo,hi, value lo,hi, value_prev lo,hi  (evm_circuit/table.py:461-468).
//
// Site numbers (low 24 bits of the status code) are listed beside each check; they follow the
// reference's evaluation order, so the first failing site is the one Python would raise at.
#pragma once
#include "common.hpp"

#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 1
__shared__ u64 st_prof_stamp[4][8];
#define ST_STAMP_ANY(k) do { st_prof_stamp[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
#define ST_STAMP(k) do { if ((threadIdx.x & 63u) == 0) st_prof_stamp[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ST_STAMP(k) do { } while (0)
#define ST_STAMP_ANY(k) do { } while (0)
#endif
enum {
    ST_RWC = 0, ST_IS_WRITE = 1, ST_TAG = 2, ST_ID = 3, ST_ADDR = 4, ST_FIELD_TAG = 5,
    ST_KEY_LO = 6, ST_KEY_HI = 7, ST_LIMB0 = 8, ST_BYTE0 = 18, ST_VAL_LO = 50, ST_VAL_HI = 51,
    ST_INIT_LO = 52, ST_INIT_HI = 53, ST_ROOT_LO = 54, ST_ROOT_HI = 55, ST_LEX = 56,
    ST_NCELLS = 57,
};
enum { MPT_NCELLS = 12 };

// Column c of a State row.  With ZK_OPT_STATE_COMPACT (w.skip == 42) the witness is the 15 cells that are not limb / byte
// decompositions — rw_counter .. storage_key hi, then value .. lexicographic selector — and the decompositions are DERIVED from the
// address and storage-key cells where the checks need them (state_load_row): what assign_state_circuit's op2row computes
// (state_circuit.py:834-842) is not stored and read back.  For witnesses assigned on the device (zk_state_assign*): 480 B per row
// instead of 1,824.
ZK_HD Fr st_col(const ZkCols& w, u32 c, u64 i) { return fr_load(w.cells + ((u64)(c >= 50u ? c - w.skip : c) * w.n + i) * 4); }

struct StateArgs {
    ZkCols rows;
    ZkTable mpt;
    u64 eval_lo, eval_hi;  // rows [eval_lo, eval_hi) are evaluated; the others are read-only halo
};

// 576-bit accumulator for the lexicographic key packing of state_circuit.py:552-565.
struct Big18 {
    u32 v[18];
};
ZK_HD void big_shl_add(Big18& a, int shift_words, int shift_bits, const Fr& x) {
    // a = (a << (32*shift_words + shift_bits)) + x, truncated to 576 bits (the reference
    // keeps only the low 31 16-bit limbs = 496 bits afterwards).
    if (shift_bits) {
        for (int k = 17; k > 0; k--) a.v[k] = (a.v[k] << shift_bits) | (a.v[k - 1] >> (32 - shift_bits));
        a.v[0] <<= shift_bits;
    }
    if (shift_words) {
        for (int k = 17; k >= 0; k--) a.v[k] = (k >= shift_words) ? a.v[k - shift_words] : 0;
    }
    u64 c = 0;
    for (int k = 0; k < 18; k++) {
        c += (u64)a.v[k] + (k < 8 ? x.v[k] : 0u);
        a.v[k] = (u32)c;
        c >>= 32;
    }
}

// Packs (tag, id, address, field_tag, storage_key_bytes, rw_counter) exactly like
// keys_rwc_to_limbs_in_order (state_circuit.py:552-565) using the raw `.n` of every cell.
// Returns false if a storage-key byte is >= 256 (Python: bytes() raises ValueError).
ZK_HD bool state_pack_keys(const ZkCols& w, u64 i, Big18& out) {
    for (int k = 0; k < 18; k++) out.v[k] = 0;
    U256 key = fr_zero();
    bool ok = true;
    for (int b = 0; b < 32; b++) {
        Fr c = st_col(w, ST_BYTE0 + b, i);
        if (!fr_le_u64(c, 255)) ok = false;
        key.v[b >> 2] |= (c.v[0] & 0xff) << (8 * (b & 3));
    }
    if (!ok) return false;
    big_shl_add(out, 0, 0, st_col(w, ST_TAG, i));
    big_shl_add(out, 0, 28, st_col(w, ST_ID, i));         // v * 2^ID_BITS + id
    big_shl_add(out, 5, 0, st_col(w, ST_ADDR, i));        // v * 2^160 + address
    big_shl_add(out, 0, 16, st_col(w, ST_FIELD_TAG, i));  // v * 2^16 + field_tag
    big_shl_add(out, 1, 0, key);                          // v * 2^32 + storage key (256-bit)
    big_shl_add(out, 1, 0, st_col(w, ST_RWC, i));         // v * 2^32 + rw_counter
    // keep 31 limbs of 16 bits = 496 bits
    out.v[15] &= 0xffffu;
    out.v[16] = 0;
    out.v[17] = 0;
    return true;
}
ZK_HD bool big_lt(const Big18& a, const Big18& b) {
    bool lt = false;
    for (int k = 0; k < 16; k++) lt = (a.v[k] < b.v[k]) || (a.v[k] == b.v[k] && lt);
    return lt;
}

ZK_HD bool state_keys_eq(const ZkCols& w, u64 i, u64 j) {
    bool eq = true;
    for (int c = ST_TAG; c <= ST_KEY_HI; c++) eq = eq && fr_eq(st_col(w, c, i), st_col(w, c, j));
    return eq;
}
ZK_HD bool state_pair_eq(const ZkCols& w, int c, u64 i, u64 j) {
    return fr_eq(st_col(w, c, i), st_col(w, c, j)) && fr_eq(st_col(w, c + 1, i), st_col(w, c + 1, j));
}
ZK_HD bool state_pair_zero(const ZkCols& w, int c, u64 i) {
    return fr_is_zero(st_col(w, c, i)) && fr_is_zero(st_col(w, c + 1, i));
}

// MPT lookup with every field given (state_circuit.py:165-184 -> table.py:864-884): since the query covers all 12 cells and the
// table is a set, 0 or 1 distinct rows can match.  The index is keyed on the WHOLE row, so a probe chain holds only genuine hash
// neighbours (expected length ~1 at load factor <= 1/2), and a candidate row is compared with all of its 24 loads in flight at
// once (no short-circuit): a lookup is two dependent memory latencies, not one per cell.
ZK_HD u64 state_mpt_hash_cells(const Fr q[MPT_NCELLS]) {
    // address, proof type, storage key and the new root's low half: the trie leaf and which of its updates (the other seven cells
    // follow from these in any consistent table; the hash only has to spread the rows, the compare below is on all twelve)
    u64 h = 0x5bd1e995u;
#pragma unroll
    for (int c = 0; c < 5; c++) h = zk_hash_cell(h, q[c]);
    return h;
}
// Index slot of an MPT row: the row number with eight more bits of its hash on top (tables below 2^24 - 1 rows), so that a probe
// chain's other residents are told apart without reading their 384 bytes.
ZK_HD u32 state_mpt_fp_bits(const ZkTable& t) { return t.n < 0xffffffu ? 8u : 0u; }
ZK_HD u32 state_mpt_slot_value(const ZkTable& t, u32 r, u64 h) { return state_mpt_fp_bits(t) ? (r | ((u32)(h >> 56) << 24)) : r; }
#ifndef ZK_HOSTSIM
typedef u32 st_ld_u32x4 __attribute__((ext_vector_type(4)));
// The 24 16-byte loads of one table row issued together and waited for together.  Written as asm because under this kernel's
// register pressure the compiler emits such a compare as load, wait, compare, 24 times over (one register tuple re-used: a
// candidate row cost 24 dependent memory latencies, ~50k clocks per Storage / Account wavefront); here a row is one.
__device__ __forceinline__ void st_load_mpt_row(const u64* p, st_ld_u32x4 (&x)[24]) {
    asm volatile(
        "global_load_dwordx4 %0, %24, off\n\tglobal_load_dwordx4 %1, %24, off offset:16\n\t"
        "global_load_dwordx4 %2, %24, off offset:32\n\tglobal_load_dwordx4 %3, %24, off offset:48\n\t"
        "global_load_dwordx4 %4, %24, off offset:64\n\tglobal_load_dwordx4 %5, %24, off offset:80\n\t"
        "global_load_dwordx4 %6, %24, off offset:96\n\tglobal_load_dwordx4 %7, %24, off offset:112\n\t"
        "global_load_dwordx4 %8, %24, off offset:128\n\tglobal_load_dwordx4 %9, %24, off offset:144\n\t"
        "global_load_dwordx4 %10, %24, off offset:160\n\tglobal_load_dwordx4 %11, %24, off offset:176\n\t"
        "global_load_dwordx4 %12, %24, off offset:192\n\tglobal_load_dwordx4 %13, %24, off offset:208\n\t"
        "global_load_dwordx4 %14, %24, off offset:224\n\tglobal_load_dwordx4 %15, %24, off offset:240\n\t"
        "global_load_dwordx4 %16, %24, off offset:256\n\tglobal_load_dwordx4 %17, %24, off offset:272\n\t"
        "global_load_dwordx4 %18, %24, off offset:288\n\tglobal_load_dwordx4 %19, %24, off offset:304\n\t"
        "global_load_dwordx4 %20, %24, off offset:320\n\tglobal_load_dwordx4 %21, %24, off offset:336\n\t"
        "global_load_dwordx4 %22, %24, off offset:352\n\tglobal_load_dwordx4 %23, %24, off offset:368\n\ts_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
          "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17]), "=&v"(x[18]),
          "=&v"(x[19]), "=&v"(x[20]), "=&v"(x[21]), "=&v"(x[22]), "=&v"(x[23])
        : "v"(p)
        : "memory");
}
// key cells (tag .. storage_key hi: columns 2 .. 7) of row j against `mine`: twelve loads in flight, one wait
__device__ __forceinline__ u32 st_next_keys_diff(const ZkCols& w, u64 j, const Fr* const mine[6]) {
    st_ld_u32x4 x[12];
    const u64* p0 = w.cells + ((u64)ST_TAG * w.n + j) * 4;
    const u64 *p1 = p0 + w.n * 4, *p2 = p1 + w.n * 4, *p3 = p2 + w.n * 4, *p4 = p3 + w.n * 4, *p5 = p4 + w.n * 4;
    asm volatile(
        "global_load_dwordx4 %0, %12, off\n\tglobal_load_dwordx4 %1, %12, off offset:16\n\t"
        "global_load_dwordx4 %2, %13, off\n\tglobal_load_dwordx4 %3, %13, off offset:16\n\t"
        "global_load_dwordx4 %4, %14, off\n\tglobal_load_dwordx4 %5, %14, off offset:16\n\t"
        "global_load_dwordx4 %6, %15, off\n\tglobal_load_dwordx4 %7, %15, off offset:16\n\t"
        "global_load_dwordx4 %8, %16, off\n\tglobal_load_dwordx4 %9, %16, off offset:16\n\t"
        "global_load_dwordx4 %10, %17, off\n\tglobal_load_dwordx4 %11, %17, off offset:16\n\ts_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
          "=&v"(x[10]), "=&v"(x[11])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5)
        : "memory");
    u32 d = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const Fr& c = *mine[k >> 1];
        const int o = (k & 1) * 4;
        d |= (x[k].x ^ c.v[o]) | (x[k].y ^ c.v[o + 1]) | (x[k].z ^ c.v[o + 2]) | (x[k].w ^ c.v[o + 3]);
    }
    return d;
}
#endif
ZK_HD bool state_mpt_row_equals(const ZkTable& t, u32 r, const Fr q[MPT_NCELLS]) {
    u32 diff = 0;
#ifndef ZK_HOSTSIM
    st_ld_u32x4 x[24];
    st_load_mpt_row(t.cells + (u64)r * (MPT_NCELLS * 4), x);
#pragma unroll
    for (int k = 0; k < 24; k++) {
        const Fr& c = q[k >> 1];
        const int o = (k & 1) * 4;
        diff |= (x[k].x ^ c.v[o]) | (x[k].y ^ c.v[o + 1]) | (x[k].z ^ c.v[o + 2]) | (x[k].w ^ c.v[o + 3]);
    }
#else
    for (int c = 0; c < MPT_NCELLS; c++) {
        const Fr x = zk_table_cell(t, r, c);
        for (int k = 0; k < 8; k++) diff |= x.v[k] ^ q[c].v[k];
    }
#endif
    return diff == 0;
}
ZK_HD bool state_mpt_lookup(const ZkTable& t, const Fr q[MPT_NCELLS]) {
    if (t.n == 0) return false;
    const u64 h = state_mpt_hash_cells(q);
    const u32 fp_bits = state_mpt_fp_bits(t), want = fp_bits ? (u32)(h >> 56) : 0u, row_mask = fp_bits ? 0xffffffu : 0xffffffffu;
    u32 slot = (u32)h & t.mask;
    ST_STAMP_ANY(6);
    // Four slots of the probe chain per round trip (the index has at least 16 slots and always an empty one).  Each lane scans its
    // window in registers for the next slot whose fingerprint matches; the row compare sits after the scan, so all lanes of a
    // wavefront that have a candidate compare it in the SAME round trip wherever in their windows it was found.
    bool found = false, done = false;
    u32 k = 4, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    for (u32 scanned = 0; !done && scanned <= t.mask; ) {  // (bounded by the slot count: a damaged index cannot spin the wavefront)
        if (k == 4) {
            w0 = t.slots[slot], w1 = t.slots[(slot + 1u) & t.mask], w2 = t.slots[(slot + 2u) & t.mask], w3 = t.slots[(slot + 3u) & t.mask];
            slot = (slot + 4) & t.mask;
            k = 0;
        }
        u32 cand = ZK_EMPTY_SLOT;
        while (k < 4) {
            const u32 sv = w0;
            w0 = w1, w1 = w2, w2 = w3;
            k++;
            scanned++;
            if (sv == ZK_EMPTY_SLOT) {
                done = true;
                break;
            }
            if (!fp_bits || (sv >> 24) == want) {
                cand = sv;
                break;
            }
        }
        if (!done && cand != ZK_EMPTY_SLOT && state_mpt_row_equals(t, cand & row_mask, q)) found = done = true;
    }
    return found;
}
ZK_HD u64 state_mpt_key_hash(const ZkTable& t, u32 r) {
    Fr q[MPT_NCELLS];
#pragma unroll
    for (int c = 0; c < MPT_NCELLS; c++) q[c] = zk_table_cell(t, r, c);
    return state_mpt_hash_cells(q);
}

// Checks are accumulated branch-free ("first failure wins") instead of returning early, so the
// witness loads of later checks are not control-dependent on earlier ones and the compiler can
// keep many 32-byte cell loads in flight per lane (the kernel is HBM-latency/bandwidth bound).
#define ST_FAIL(kind, site) code = (code == 0u) ? ZK_CODE(kind, site) : code
#define ST_ASSERT(cond, site) code = (code == 0u && !(cond)) ? ZK_CODE(ZK_ASSERT, site) : code

// Everything the checks need from ONE row.  A lane loads only its own row (57 coalesced cells);
// the previous row's values come from the neighbouring lane (wave_shr DPP moves on the device, the
// separately loaded previous row in the host build): a wavefront evaluates 63 rows, lane 0 is the
// read-only halo row in front of them.
struct StRow {
    Fr rwc, tag, id, addr, ftag, key_lo, key_hi, val_lo, val_hi, init_lo, init_hi, root_lo, root_hi;
    u32 pack[16];  // keys_rwc_to_limbs_in_order packing of this row (state_circuit.py:552-565), 496 bits
    u32 pack_ok;   // 0 when a storage-key byte cell is >= 256 (Python: bytes() raises ValueError)
    u32 flags;
    u32 is_write01;  // is_write cell: 0, 1, or 2 = anything else
};

// Loads row i and evaluates the checks that involve this row only (sites 1..8, evaluation order
// of check_state_row :498-520); `code` is the row's running first-failure code.
ZK_HD void state_load_row(const ZkCols& w, u64 i, StRow& R, u32& code) {
    R.flags = w.flags ? w.flags[i] : 0u;
    R.rwc = st_col(w, ST_RWC, i);
    const Fr is_write = st_col(w, ST_IS_WRITE, i);
    R.tag = st_col(w, ST_TAG, i);
    R.id = st_col(w, ST_ID, i);
    R.addr = st_col(w, ST_ADDR, i);
    R.ftag = st_col(w, ST_FIELD_TAG, i);
    R.key_lo = st_col(w, ST_KEY_LO, i);
    R.key_hi = st_col(w, ST_KEY_HI, i);
    // 0.0 tag, id, field_tag ranges (:498-502)
    ST_ASSERT(fr_fits64(R.tag) && fr_lo64(R.tag) >= 1 && fr_lo64(R.tag) <= 12, 1);
    ST_ASSERT(fr_le_u64(R.id, (1ull << 28) - 1), 2);
    ST_ASSERT(fr_le_u64(R.ftag, 24), 3);
    // 0.1 address limbs are 16-bit and recompose to address in Fr (:505-509)
    U256 key = fr_zero();
    if (w.skip) {
        // derived decompositions (ZK_OPT_STATE_COMPACT): the ten limbs are the address's low 160 bits, the 32 bytes the key halves' low
        // 128 bits each — in range by construction; what remains of 0.1 / 0.2 is that they recompose, i.e. that the cells are that narrow
        U256 lc = fr_zero();
#pragma unroll
        for (int k = 0; k < 5; k++) lc.v[k] = R.addr.v[k];
        ST_ASSERT(fr_eq(R.addr, lc), 5);
#pragma unroll
        for (int k = 0; k < 4; k++) { key.v[k] = R.key_lo.v[k]; key.v[4 + k] = R.key_hi.v[k]; }
        R.pack_ok = 1u;
        ST_ASSERT(fr_eq(R.key_lo, u256_lo(key)) && fr_eq(R.key_hi, u256_hi(key)), 7);
    } else {
    {
        U256 lc = fr_zero();
        bool limbs_ok = true;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const Fr limb = st_col(w, ST_LIMB0 + k, i);
            limbs_ok = limbs_ok && fr_le_u64(limb, 65535);
            lc.v[k >> 1] |= (limb.v[0] & 0xffffu) << (16 * (k & 1));
        }
        ST_ASSERT(limbs_ok, 4);
        ST_ASSERT(fr_eq(R.addr, lc), 5);  // sum < 2^160 < p: integer value == field value
    }
    // 0.2 storage-key bytes are bytes and recompose to (lo, hi) (:512-517)
    {
        bool bytes_ok = true;
#pragma unroll
        for (int b = 0; b < 32; b++) {
            const Fr c = st_col(w, ST_BYTE0 + b, i);
            bytes_ok = bytes_ok && fr_le_u64(c, 255);
            key.v[b >> 2] |= (c.v[0] & 0xffu) << (8 * (b & 3));
        }
        R.pack_ok = bytes_ok ? 1u : 0u;
        ST_ASSERT(bytes_ok, 6);
        ST_ASSERT(fr_eq(R.key_lo, u256_lo(key)) && fr_eq(R.key_hi, u256_hi(key)), 7);
    }
    }
    // 0.3 is_write boolean (:520)
    ST_ASSERT(fr_le_u64(is_write, 1), 8);
    R.is_write01 = fr_is_zero(is_write) ? 0u : (fr_eq_u64(is_write, 1) ? 1u : 2u);
    // key packing (tag, id, address, field_tag, storage_key_bytes, rw_counter), :552-565
    {
        Big18 out;
        for (int k = 0; k < 18; k++) out.v[k] = 0;
        big_shl_add(out, 0, 0, R.tag);
        big_shl_add(out, 0, 28, R.id);     // v * 2^ID_BITS + id
        big_shl_add(out, 5, 0, R.addr);    // v * 2^160 + address
        big_shl_add(out, 0, 16, R.ftag);   // v * 2^16 + field_tag
        big_shl_add(out, 1, 0, key);       // v * 2^32 + storage key (256-bit)
        big_shl_add(out, 1, 0, R.rwc);     // v * 2^32 + rw_counter
        out.v[15] &= 0xffffu;              // keep 31 limbs of 16 bits = 496 bits
#pragma unroll
        for (int k = 0; k < 16; k++) R.pack[k] = out.v[k];
    }
    R.val_lo = st_col(w, ST_VAL_LO, i);
    R.val_hi = st_col(w, ST_VAL_HI, i);
    R.init_lo = st_col(w, ST_INIT_LO, i);
    R.init_hi = st_col(w, ST_INIT_HI, i);
    R.root_lo = st_col(w, ST_ROOT_LO, i);
    R.root_hi = st_col(w, ST_ROOT_HI, i);
}

// state_load_row for the 15-cell rows alone (ZK_OPT_STATE_COMPACT): the device kernel's loader — none of the 42-column code.
ZK_HD void state_finish_row_compact(StRow& R, const Fr& is_write, u32& code);
ZK_HD void state_load_row_compact(const ZkCols& w, u64 i, StRow& R, u32& code) {
    R.flags = w.flags ? w.flags[i] : 0u;
    R.rwc = st_col(w, ST_RWC, i);
    const Fr is_write = st_col(w, ST_IS_WRITE, i);
    R.tag = st_col(w, ST_TAG, i);
    R.id = st_col(w, ST_ID, i);
    R.addr = st_col(w, ST_ADDR, i);
    R.ftag = st_col(w, ST_FIELD_TAG, i);
    R.key_lo = st_col(w, ST_KEY_LO, i);
    R.key_hi = st_col(w, ST_KEY_HI, i);
    R.val_lo = st_col(w, ST_VAL_LO, i);
    R.val_hi = st_col(w, ST_VAL_HI, i);
    R.init_lo = st_col(w, ST_INIT_LO, i);
    R.init_hi = st_col(w, ST_INIT_HI, i);
    R.root_lo = st_col(w, ST_ROOT_LO, i);
    R.root_hi = st_col(w, ST_ROOT_HI, i);
    state_finish_row_compact(R, is_write, code);
}
// The row's own checks (sites 1..8) and its key packing from the fifteen cells alone; also the tail of the loader that computes a row
// from its State op instead of reading it (state_fused.hpp).
ZK_HD void state_finish_row_compact(StRow& R, const Fr& is_write, u32& code) {
    ST_ASSERT(fr_fits64(R.tag) && fr_lo64(R.tag) >= 1 && fr_lo64(R.tag) <= 12, 1);
    ST_ASSERT(fr_le_u64(R.id, (1ull << 28) - 1), 2);
    ST_ASSERT(fr_le_u64(R.ftag, 24), 3);
    ST_ASSERT((R.addr.v[5] | R.addr.v[6] | R.addr.v[7]) == 0u, 5);   // the derived limbs recompose iff address < 2^160
    R.pack_ok = 1u;
    ST_ASSERT(fr_fits128(R.key_lo) && fr_fits128(R.key_hi), 7);       // the derived bytes recompose iff both halves < 2^128
    U256 keyv;
#pragma unroll
    for (int k = 0; k < 4; k++) { keyv.v[k] = R.key_lo.v[k]; keyv.v[4 + k] = R.key_hi.v[k]; }
    ST_ASSERT(fr_le_u64(is_write, 1), 8);
    R.is_write01 = fr_is_zero(is_write) ? 0u : (fr_eq_u64(is_write, 1) ? 1u : 2u);
    {
        Big18 out;
        for (int k = 0; k < 18; k++) out.v[k] = 0;
        big_shl_add(out, 0, 0, R.tag);
        big_shl_add(out, 0, 28, R.id);
        big_shl_add(out, 5, 0, R.addr);
        big_shl_add(out, 0, 16, R.ftag);
        big_shl_add(out, 1, 0, keyv);
        big_shl_add(out, 1, 0, R.rwc);
        out.v[15] &= 0xffffu;
#pragma unroll
        for (int k = 0; k < 16; k++) R.pack[k] = out.v[k];
    }
}

#ifdef ZK_HOSTSIM
#define ST_PREV_U32(x) (P.x)
#define ST_PREV_FR(f) (P.f)
#else
// value of the same expression in the lane that holds the previous row: lane - 1 with one lane per row (gfx9 DPP wave_shr:1),
// lane - 4 with a lane quad per row (ds_bpermute through __shfl_up; no LDS memory involved).  The first row's lanes are the
// halo and unused.
template <int SHIFT>
ZK_HD u32 st_shr_u32(u32 v) {
    if constexpr (SHIFT == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
    else return (u32)__shfl_up((int)v, SHIFT);
}
template <int SHIFT>
ZK_HD Fr st_shr_fr(const Fr& x) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = st_shr_u32<SHIFT>(x.v[k]);
    return r;
}
#define ST_PREV_U32(x) st_shr_u32<SHIFT>(C.x)
#define ST_PREV_FR(f) st_shr_fr<SHIFT>(C.f)
#endif

// Checks of row i that involve the previous row (sites 9..13) and the per-tag rules; C = this
// row, P = previous row (host build only; the device takes it from the neighbouring lane).
// Must be called by every lane of the wavefront (the DPP moves read the neighbour's registers).
// Where the two reads of state_check_loaded that are not in a StRow come from — the Start row's lexicographic-ordering cell and the key
// cells of row i + 1 for the last row of a wavefront: the witness (StateArgs), or the State ops themselves (StateFusedArgs,
// state_fused.hpp: overloads on the argument type).
ZK_HD Fr st_src_lex(const StateArgs& a, u64 i) { return st_col(a.rows, ST_LEX, i); }
#ifndef ZK_HOSTSIM
__device__ __forceinline__ u32 st_src_next_keys_diff(const StateArgs& a, u64 j, const Fr* const mine[6]) { return st_next_keys_diff(a.rows, j, mine); }
#endif
template <int SHIFT = 1, class ARGS = StateArgs>
ZK_HD u32 state_check_loaded(const ARGS& a, u64 i, const StRow& C, const StRow& P, u32 code) {
    (void)P;
    ST_STAMP(0);
#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 2
    u32 prof_lane = 0;  // tuning build: the return value is (clocks / 16) of the next-row compare | of the MPT lookup << 16
#endif
    const ZkCols& w = a.rows;
    const u64 n = w.n;
    const u64 in = i + 1 == n ? 0 : i + 1;
    const bool val_is_word = C.flags & 1u, init_is_word = C.flags & 2u;
    const u32 tagv = C.tag.v[0];
    const Fr& rwc = C.rwc; const Fr& tag = C.tag; const Fr& id = C.id; const Fr& addr = C.addr; const Fr& ftag = C.ftag;
    const bool is_read = C.is_write01 == 0u;

    // 0.4 lexicographic ordering (:552-570)
    {
        const u32 p_ok = ST_PREV_U32(pack_ok);
        u32 kp[16];
#pragma unroll
        for (int k = 0; k < 16; k++) kp[k] = ST_PREV_U32(pack[k]);
        if (!p_ok) ST_FAIL(ZK_VALUE_ERROR, 9);
        bool lt = false;
#pragma unroll
        for (int k = 0; k < 16; k++) lt = (kp[k] < C.pack[k]) | ((kp[k] == C.pack[k]) & lt);
        ST_ASSERT(tagv == 1 || lt, 10);
    }
    // every neighbour read happens unconditionally, with all lanes active (no short-circuit around a DPP move)
    const Fr p_tag = ST_PREV_FR(tag), p_id = ST_PREV_FR(id), p_addr = ST_PREV_FR(addr), p_ftag = ST_PREV_FR(ftag);
    const Fr p_key_lo = ST_PREV_FR(key_lo), p_key_hi = ST_PREV_FR(key_hi);
    const Fr p_val_lo = ST_PREV_FR(val_lo), p_val_hi = ST_PREV_FR(val_hi);
    const Fr p_init_lo = ST_PREV_FR(init_lo), p_init_hi = ST_PREV_FR(init_hi);
    const Fr p_root_lo = ST_PREV_FR(root_lo), p_root_hi = ST_PREV_FR(root_hi);
    const bool keys_eq_prev = fr_eq(tag, p_tag) & fr_eq(id, p_id) & fr_eq(addr, p_addr) & fr_eq(ftag, p_ftag) &
                              fr_eq(C.key_lo, p_key_lo) & fr_eq(C.key_hi, p_key_hi);
#ifndef ZK_HOSTSIM
    const u64 eq_prev_mask = __ballot(keys_eq_prev);  // every lane is active here
#endif
    const bool val_same = fr_eq(C.val_lo, p_val_lo) & fr_eq(C.val_hi, p_val_hi);
    const bool init_same = fr_eq(C.init_lo, p_init_lo) & fr_eq(C.init_hi, p_init_hi);
    const bool root_same = fr_eq(C.root_lo, p_root_lo) & fr_eq(C.root_hi, p_root_hi);
    const Fr p_rwc = ST_PREV_FR(rwc);
    const u32 p_flags = ST_PREV_U32(flags);
    // 0.5 read consistency (:577-581)
    ST_ASSERT(!(is_read && keys_eq_prev) || val_same, 11);
    ST_ASSERT(!keys_eq_prev || init_same, 12);
    // 8. rw_counter != 0 except Start (:584-585)
    ST_ASSERT(tagv == 1 || !fr_is_zero(rwc), 13);

    const Fr& val_lo = C.val_lo; const Fr& val_hi = C.val_hi; const Fr& init_lo = C.init_lo; const Fr& init_hi = C.init_hi;
    const bool key_zero = fr_is_zero(C.key_lo) && fr_is_zero(C.key_hi);
    const bool val_zero = fr_is_zero(val_lo) && fr_is_zero(val_hi);
    const bool init_zero = fr_is_zero(init_lo) && fr_is_zero(init_hi);
    const bool is_write_one = C.is_write01 == 1u;

    ST_STAMP(1);
    switch (tagv) {
    case 1:  // Start (:216-236)
        ST_ASSERT(fr_is_zero(ftag), 20);
        ST_ASSERT(fr_is_zero(addr), 21);
        ST_ASSERT(fr_is_zero(id), 22);
        ST_ASSERT(key_zero, 23);
        ST_ASSERT(fr_is_zero(val_hi), 24);
        ST_ASSERT(fr_is_zero(init_hi), 25);
        {
            Fr lex = st_src_lex(a, i);
            Fr d = fr_sub_u64(fr_sub(rwc, p_rwc), 1);
            ST_ASSERT(fr_is_zero(lex) || fr_is_zero(d), 26);  // p prime: product zero iff a factor is
            ST_ASSERT(!val_is_word, 27);
            ST_ASSERT(fr_is_zero(val_lo), 28);
            ST_ASSERT(!init_is_word, 29);
            ST_ASSERT(fr_is_zero(init_lo), 30);
            if (!fr_is_zero(lex)) ST_ASSERT(root_same, 31);
        }
        break;
    case 2:  // Memory (:240-266)
        ST_ASSERT(fr_is_zero(ftag), 40);
        ST_ASSERT(key_zero, 41);
        ST_ASSERT(fr_is_zero(val_hi), 42);
        ST_ASSERT(fr_is_zero(init_hi), 43);
        if (!keys_eq_prev && is_read) {
            ST_ASSERT(!val_is_word, 44);
            ST_ASSERT(fr_is_zero(val_lo), 45);
        }
        ST_ASSERT(fr_le_u64(addr, 0xffffffffull), 46);
        ST_ASSERT(!val_is_word, 47);
        ST_ASSERT(fr_le_u64(val_lo, 255), 48);
        ST_ASSERT(!init_is_word, 49);
        ST_ASSERT(fr_is_zero(init_lo), 50);
        ST_ASSERT(root_same, 51);
        break;
    case 3:  // Stack (:270-301)
        ST_ASSERT(fr_is_zero(ftag), 60);
        ST_ASSERT(key_zero, 61);
        if (!keys_eq_prev) ST_ASSERT(is_write_one, 62);
        ST_ASSERT(fr_le_u64(addr, 1023), 63);
        if (fr_eq(tag, p_tag) && fr_eq(id, p_id)) {
            Fr d = fr_sub(addr, p_addr);
            ST_ASSERT(fr_le_u64(d, 1), 64);
        }
        ST_ASSERT(init_zero, 65);
        ST_ASSERT(root_same, 66);
        break;
    case 4:    // Storage (:305-324)
    case 6: {  // Account (:349-380)
        u64 proof_type;
        if (tagv == 4) {
            ST_ASSERT(fr_is_zero(ftag), 70);
            proof_type = (val_zero && init_zero) ? 4 : 6;  // NonExistingAccountProof : StorageMod
        } else {
            // AccountFieldTag(field_tag.n) raises ValueError outside 1..4 (:350)
            if (!(fr_lo64(ftag) >= 1 && fr_lo64(ftag) <= 4)) ST_FAIL(ZK_VALUE_ERROR, 90);
            ST_ASSERT(fr_is_zero(id), 91);
            ST_ASSERT(key_zero, 92);
            if (fr_eq_u64(ftag, 1)) {
                ST_ASSERT(fr_is_zero(val_hi), 93);
                ST_ASSERT(fr_is_zero(init_hi), 94);
            }
            bool non_exist = val_zero && init_zero && fr_eq_u64(ftag, 3);
            proof_type = non_exist ? 4 : fr_lo64(ftag);  // from_account_field_tag: tag k -> proof k
        }
        // last access to this key = the next row's keys differ (six loads in flight; this row's own keys are in registers)
#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 2
        const u64 pt0 = __builtin_amdgcn_s_memtime();
#endif
        u32 next_diff;
        {
            const Fr* const mine[6] = {&tag, &id, &addr, &ftag, &C.key_lo, &C.key_hi};
#ifndef ZK_HOSTSIM
            // row i + 1 sits SHIFT lanes up and has compared its keys with this row's already (its keys_eq_prev): one ballot.
            // The last rows of the wavefront, and the row in front of the wrap-around, read the next row instead.
            const u32 lane = threadIdx.x & 63u;
            if (lane + SHIFT < 64u && i + 1 < n) next_diff = ((eq_prev_mask >> (lane + SHIFT)) & 1ull) ? 0u : 1u;
            else next_diff = st_src_next_keys_diff(a, in, mine);
#else
            next_diff = 0;
            for (int c = 0; c < 6; c++) {
                const Fr x = st_col(w, ST_TAG + c, in);
                for (int k = 0; k < 8; k++) next_diff |= x.v[k] ^ mine[c]->v[k];
            }
#endif
        }
#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 2
        const u64 pt1 = __builtin_amdgcn_s_memtime();
        prof_lane = (u32)((pt1 - pt0) >> 4) & 0xffffu;
#endif
        ST_STAMP_ANY(3);
        if (next_diff != 0) {
            Fr q[MPT_NCELLS];
            q[0] = addr;
            q[1] = fr_from_u64(proof_type);
            q[2] = C.key_lo;
            q[3] = C.key_hi;
            q[4] = C.root_lo;
            q[5] = C.root_hi;
            q[6] = p_root_lo;  // root of row i - 1: already here from the neighbour exchange
            q[7] = p_root_hi;
            q[8] = val_lo;
            q[9] = val_hi;
            q[10] = init_lo;
            q[11] = init_hi;
            ST_STAMP_ANY(4);
            if (code == 0u && !state_mpt_lookup(a.mpt, q)) ST_FAIL(ZK_LOOKUP_UNSAT, tagv == 4 ? 71 : 95);
            ST_STAMP_ANY(5);
#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 2
            prof_lane |= ((u32)((__builtin_amdgcn_s_memtime() - pt1) >> 4) & 0xffffu) << 16;
#endif
        } else {
            ST_ASSERT(root_same, tagv == 4 ? 73 : 97);
        }
        break;
    }
    case 5:  // CallContext (:328-345)
        ST_ASSERT(fr_is_zero(addr), 80);
        ST_ASSERT(key_zero, 81);
        ST_ASSERT(fr_le_u64(ftag, 24), 82);
        if (!keys_eq_prev && is_read) {
            ST_ASSERT(!val_is_word, 83);
            ST_ASSERT(fr_is_zero(val_lo), 84);
        }
        ST_ASSERT(init_zero, 85);
        ST_ASSERT(root_same, 86);
        break;
    case 7:  // TxRefund (:387-402)
        ST_ASSERT(fr_is_zero(addr), 100);
        ST_ASSERT(fr_is_zero(ftag), 101);
        ST_ASSERT(key_zero, 102);
        ST_ASSERT(root_same, 103);
        ST_ASSERT(init_zero, 104);
        if (!keys_eq_prev && is_read) ST_ASSERT(val_zero, 105);
        break;
    case 8:  // TxAccessListAccount (:406-419)
        ST_ASSERT(fr_is_zero(ftag), 110);
        ST_ASSERT(key_zero, 111);
        ST_ASSERT(fr_is_zero(val_hi), 112);
        ST_ASSERT(fr_is_zero(init_hi), 113);
        ST_ASSERT(root_same, 114);
        if (!keys_eq_prev && is_read) {
            ST_ASSERT(!val_is_word, 115);
            ST_ASSERT(fr_is_zero(val_lo), 116);
        }
        break;
    case 9:  // TxAccessListAccountStorage (:423-435)
        ST_ASSERT(fr_is_zero(ftag), 120);
        ST_ASSERT(fr_is_zero(val_hi), 121);
        ST_ASSERT(fr_is_zero(init_hi), 122);
        ST_ASSERT(root_same, 123);
        if (!keys_eq_prev && is_read) {
            ST_ASSERT(!val_is_word, 124);
            ST_ASSERT(fr_is_zero(val_lo), 125);
        }
        break;
    case 10:  // TxLog (:439-453)
        if (!fr_eq_u64(ftag, 2)) {  // TxLogFieldTag.Topic
            ST_ASSERT(fr_is_zero(val_hi), 130);
            ST_ASSERT(fr_is_zero(init_hi), 131);
        }
        ST_ASSERT(is_write_one, 132);
        ST_ASSERT(root_same, 133);
        break;
    case 11: {  // TxReceipt (:460-488)
        ST_ASSERT(fr_is_zero(addr), 140);
        ST_ASSERT(key_zero, 141);
        ST_ASSERT(fr_is_zero(val_hi), 142);
        ST_ASSERT(fr_is_zero(init_hi), 143);
        if (fr_eq_u64(ftag, 1)) {  // PostStateOrStatus
            ST_ASSERT(!val_is_word, 144);
            ST_ASSERT(fr_le_u64(val_lo, 1), 145);
        }
        const bool same_tag = fr_eq(tag, p_tag);
        if (!fr_eq(id, p_id) && same_tag) {
            ST_ASSERT(fr_eq(id, fr_add_u64(p_id, 1)), 146);
            if (fr_eq_u64(ftag, 2)) {  // CumulativeGasUsed
                ST_ASSERT(!val_is_word, 147);
                ST_ASSERT(!(p_flags & 1u), 148);
                ST_ASSERT(fr_lt(p_val_lo, val_lo), 149);
            }
        }
        if (!same_tag) ST_ASSERT(fr_eq_u64(id, 1), 150);
        ST_ASSERT(fr_lo64(id) >= 1 && fr_le_u64(id, 1ull << 11), 151);
        ST_ASSERT(root_same, 152);
        break;
    }
    default:  // tag 12: passes 0.0 but is no Tag variant -> ValueError("Unreachable") (:613)
        ST_FAIL(ZK_VALUE_ERROR, 160);
    }
    ST_STAMP(2);
#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 2
    return prof_lane;
#else
    return code;
#endif
}

#ifndef ZK_HOSTSIM
// ---- A lane QUAD per row.  The 56 cells a row's own checks read are dealt round-robin to the four lanes (cell c goes to
// lane c & 3: fourteen 32-byte loads per lane instead of 56, all of them in flight at once), each lane range-checks and packs
// the limb / byte cells it holds, the partial packings are OR-reduced across the quad with two DPP quad_perm steps and the
// wide cells are broadcast from their owner lane — after which every lane of the quad holds the full StRow and lane 0 of the
// quad carries the verdict.  A wavefront covers 16 rows (15 evaluated + the halo row in front of them), so a launch has four
// times the wavefronts of the one-lane-per-row form and a quarter of its per-lane dependent work: that is what the
// latency-bound small batches (BASELINE config 2: 2^16 rows) and the load-issue-bound large ones both want.
// L = 4 (a quad per row) or 2 (a lane pair per row: cell c goes to lane c & 1, 28 loads per lane, 32 rows per wavefront — at
// 2^16 rows that is one resident round of wavefronts instead of two).
template <int L, int K>
ZK_HD u32 st_group_bcast_u32(u32 v) {  // value of lane K of this lane's group of L (groups are aligned inside quads)
    if constexpr (L == 4) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | ((2 + K) << 4) | ((2 + K) << 6), 0xf, 0xf, false);
}
template <int L, int K>
ZK_HD Fr st_group_bcast(const Fr& x) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = st_group_bcast_u32<L, K>(x.v[k]);
    return r;
}
template <int L>
ZK_HD u32 st_group_or(u32 v) {
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
    if constexpr (L == 4) v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
    return v;
}
// cell C of the row, from the lane of the group that loaded it (cell c sits in slot c / L of lane c % L)
#define ST_GROUP_CELL(C) st_group_bcast<L, (C) % L>(mine[(C) / L])
template <int L>
ZK_HD void state_load_row_group(const ZkCols& w, u64 i, u32 q, StRow& R, u32& code) {
    R.flags = w.flags ? w.flags[i] : 0u;
    constexpr int NS = 56 / L;
    Fr mine[NS];  // cell q + L k
#pragma unroll
    for (int k = 0; k < NS; k++) mine[k] = st_col(w, q + (u32)L * (u32)k, i);
    // limb cells 8..17 and key-byte cells 18..49 held by this lane: range flags + their bits of the packed values
    u32 lc[5] = {0, 0, 0, 0, 0}, key[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bad_limb = 0, bad_byte = 0;
#pragma unroll
    for (int k = 8 / L; k < 52 / L; k++) {
        const u32 c = q + (u32)L * (u32)k;  // 8 .. 51
        const Fr& x = mine[k];
        if (c < (u32)ST_BYTE0) {
            const u32 l = c - (u32)ST_LIMB0;
            bad_limb |= fr_le_u64(x, 65535) ? 0u : 1u;
            lc[l >> 1] |= (x.v[0] & 0xffffu) << (16u * (l & 1u));
        } else if (c < (u32)ST_VAL_LO) {
            const u32 b = c - (u32)ST_BYTE0;
            bad_byte |= fr_le_u64(x, 255) ? 0u : 1u;
            key[b >> 2] |= (x.v[0] & 0xffu) << (8u * (b & 3u));
        }
    }
    bad_limb = st_group_or<L>(bad_limb);
    bad_byte = st_group_or<L>(bad_byte);
    U256 lcv = fr_zero(), keyv;
#pragma unroll
    for (int k = 0; k < 5; k++) lcv.v[k] = st_group_or<L>(lc[k]);
#pragma unroll
    for (int k = 0; k < 8; k++) keyv.v[k] = st_group_or<L>(key[k]);
    // the wide cells, from their owner lanes
    R.rwc = ST_GROUP_CELL(0);
    const Fr is_write = ST_GROUP_CELL(1);
    R.tag = ST_GROUP_CELL(2);
    R.id = ST_GROUP_CELL(3);
    R.addr = ST_GROUP_CELL(4);
    R.ftag = ST_GROUP_CELL(5);
    R.key_lo = ST_GROUP_CELL(6);
    R.key_hi = ST_GROUP_CELL(7);
    R.val_lo = ST_GROUP_CELL(50);
    R.val_hi = ST_GROUP_CELL(51);
    R.init_lo = ST_GROUP_CELL(52);
    R.init_hi = ST_GROUP_CELL(53);
    R.root_lo = ST_GROUP_CELL(54);
    R.root_hi = ST_GROUP_CELL(55);
    // sites 1..8 exactly as state_load_row
    ST_ASSERT(fr_fits64(R.tag) && fr_lo64(R.tag) >= 1 && fr_lo64(R.tag) <= 12, 1);
    ST_ASSERT(fr_le_u64(R.id, (1ull << 28) - 1), 2);
    ST_ASSERT(fr_le_u64(R.ftag, 24), 3);
    ST_ASSERT(!bad_limb, 4);
    ST_ASSERT(fr_eq(R.addr, lcv), 5);
    R.pack_ok = bad_byte ? 0u : 1u;
    ST_ASSERT(!bad_byte, 6);
    ST_ASSERT(fr_eq(R.key_lo, u256_lo(keyv)) && fr_eq(R.key_hi, u256_hi(keyv)), 7);
    ST_ASSERT(fr_le_u64(is_write, 1), 8);
    R.is_write01 = fr_is_zero(is_write) ? 0u : (fr_eq_u64(is_write, 1) ? 1u : 2u);
    {
        Big18 out;
        for (int k = 0; k < 18; k++) out.v[k] = 0;
        big_shl_add(out, 0, 0, R.tag);
        big_shl_add(out, 0, 28, R.id);
        big_shl_add(out, 5, 0, R.addr);
        big_shl_add(out, 0, 16, R.ftag);
        big_shl_add(out, 1, 0, keyv);
        big_shl_add(out, 1, 0, R.rwc);
        out.v[15] &= 0xffffu;
#pragma unroll
        for (int k = 0; k < 16; k++) R.pack[k] = out.v[k];
    }
}
#endif

#ifndef ZK_HOSTSIM
// ---- The 42 range-check cells through LDS (round 3).  Three quarters of a row are cells whose whole content is "a 16-bit
// limb" or "a byte": 10 address limbs + 32 storage-key bytes, 32 B each on the wire.  Held in registers they are what caps the
// one-lane-per-row kernel at the loads a lane can keep in flight (174 VGPRs, 4.7 TB/s at 2^20 rows, one dependent batch per
// wavefront at 2^16).  Here a wavefront streams them through a private LDS ring with global_load_lds_dwordx4 (1 KiB per wave
// instruction = 32 rows of one column, no registers, asynchronous): ST_DMA_COLS columns x 64 rows per chunk, ST_DMA_RING chunks
// in flight, each lane then reads ITS row's cell back (two ds_read_b128), range-checks it and ORs its bits into the packed
// address / key words.  The 14 wide cells are ordinary loads issued first, in flight the whole time; the cross-row part is the
// unchanged wave_shr exchange of state_check_loaded<1>.  vmcnt retires in issue order, so "chunk k has landed" is
// vmcnt(pieces issued after chunk k); the compiler's own loads only make that wait more conservative.
#ifndef ST_DMA_COLS
#define ST_DMA_COLS 3
#endif
#ifndef ST_DMA_RING
#define ST_DMA_RING 3
#endif
#define ST_DMA_SMALL 42  // cells 8 .. 49
#define ST_DMA_CHUNKS (ST_DMA_SMALL / ST_DMA_COLS)
#define ST_DMA_CHUNK_U4 (ST_DMA_COLS * 64 * 2)               // uint4s per chunk (64 rows x 32 B per column)
#define ST_DMA_WAVE_BYTES (ST_DMA_RING * ST_DMA_CHUNK_U4 * 16)
static_assert(ST_DMA_SMALL % ST_DMA_COLS == 0, "chunks cover the 42 small cells exactly");
static_assert(2 * ST_DMA_COLS * (ST_DMA_RING - 1) < 64, "vmcnt is a 6-bit counter");

template <int N>
__device__ __forceinline__ void st_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// chunk -> its ring slot: ST_DMA_COLS columns, two 1 KiB pieces each (rows 0..31, 32..63 of the wavefront)
__device__ __forceinline__ void st_dma_issue(const ZkCols& w, int chunk, uint4* ring, const u64 off[2]) {
    uint4* slot = ring + (chunk % ST_DMA_RING) * ST_DMA_CHUNK_U4;
    const char* base = (const char*)w.cells;
#pragma unroll
    for (int k = 0; k < ST_DMA_COLS; k++) {
        const u64 col = (u64)(ST_LIMB0 + chunk * ST_DMA_COLS + k) * w.n * 32u;
#pragma unroll
        for (int h = 0; h < 2; h++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + col + off[h]),
                                             (__attribute__((address_space(3))) void*)(slot + (k * 64 + h * 32) * 2), 16, 0, 0);
    }
}
typedef u32 st_u32x4 __attribute__((ext_vector_type(4)));
// One column's cell of this lane out of a landed chunk.  Written as asm because the compiler orders every LDS read it emits itself
// behind vmcnt(0) once an LDS-DMA is in flight (it cannot tell the ring slots apart), which would serialise the ring.
template <int OFF>
__device__ __forceinline__ void st_lds_cell(u32 addr, st_u32x4& lo, st_u32x4& hi) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF), "n"(OFF + 16) : "memory");
}
template <int K>
__device__ __forceinline__ void st_dma_steps(const ZkCols& w, uint4* ring, u32 lds_lane, const u64 off[2], u32 lc[5], u32 key[8], u32& bad_limb,
                                             u32& bad_byte) {
    if constexpr (K < ST_DMA_CHUNKS) {
        // chunks K+1 .. K+RING-1 were issued after chunk K (those that exist)
        constexpr int later = (ST_DMA_CHUNKS - 1 - K) < (ST_DMA_RING - 1) ? (ST_DMA_CHUNKS - 1 - K) : (ST_DMA_RING - 1);
#ifdef ST_DMA_WAIT_ALL
        st_wait_vmcnt<0>();
#else
        st_wait_vmcnt<later * ST_DMA_COLS * 2>();
#endif
        constexpr int slot_off = (K % ST_DMA_RING) * ST_DMA_CHUNK_U4 * 16;
        st_u32x4 lo[ST_DMA_COLS], hi[ST_DMA_COLS];
        if constexpr (ST_DMA_COLS > 0) st_lds_cell<slot_off + 0 * 2048>(lds_lane, lo[0], hi[0]);
        if constexpr (ST_DMA_COLS > 1) st_lds_cell<slot_off + 1 * 2048>(lds_lane, lo[1], hi[1]);
        if constexpr (ST_DMA_COLS > 2) st_lds_cell<slot_off + 2 * 2048>(lds_lane, lo[2], hi[2]);
        if constexpr (ST_DMA_COLS > 3) st_lds_cell<slot_off + 3 * 2048>(lds_lane, lo[3], hi[3]);
        if constexpr (ST_DMA_COLS > 4) st_lds_cell<slot_off + 4 * 2048>(lds_lane, lo[4], hi[4]);
        if constexpr (ST_DMA_COLS > 5) st_lds_cell<slot_off + 5 * 2048>(lds_lane, lo[5], hi[5]);
        static_assert(ST_DMA_COLS <= 6, "add st_lds_cell lines");
        // the cells are in registers after this (the wait names them: their consumers must not be scheduled in front of it); the
        // slot may then be refilled
        if constexpr (ST_DMA_COLS == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1])::"memory");
        else if constexpr (ST_DMA_COLS == 3)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2])::"memory");
        else {
            static_assert(ST_DMA_COLS == 6, "add a wait for this chunk width");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]), "+v"(lo[4]), "+v"(hi[4]),
                           "+v"(lo[5]), "+v"(hi[5])::"memory");
        }
        if constexpr (K + ST_DMA_RING < ST_DMA_CHUNKS) st_dma_issue(w, K + ST_DMA_RING, ring, off);
#pragma unroll
        for (int k = 0; k < ST_DMA_COLS; k++) {
            const u32 upper = lo[k].y | lo[k].z | lo[k].w | hi[k].x | hi[k].y | hi[k].z | hi[k].w;
            const int c = ST_LIMB0 + K * ST_DMA_COLS + k;
            if (c < ST_BYTE0) {
                const int l = c - ST_LIMB0;
                bad_limb |= upper | (lo[k].x >> 16);
                lc[l >> 1] |= (lo[k].x & 0xffffu) << (16 * (l & 1));
            } else {
                const int b = c - ST_BYTE0;
                bad_byte |= upper | (lo[k].x >> 8);
                key[b >> 2] |= (lo[k].x & 0xffu) << (8 * (b & 3));
            }
        }
        // pin the chunk's arithmetic in front of the next chunk's reads: asm statements keep their order, plain VALU work does not,
        // and the scheduler otherwise parks all 42 cells in registers (then scratch) and reduces them at the end
        asm volatile("" : "+v"(bad_limb), "+v"(bad_byte), "+v"(lc[0]), "+v"(lc[1]), "+v"(lc[2]), "+v"(lc[3]), "+v"(lc[4]), "+v"(key[0]), "+v"(key[1]),
                     "+v"(key[2]), "+v"(key[3]), "+v"(key[4]), "+v"(key[5]), "+v"(key[6]), "+v"(key[7]));
        st_dma_steps<K + 1>(w, ring, lds_lane, off, lc, key, bad_limb, bad_byte);
    }
}
// state_load_row for lane `lane` of a wavefront whose lane j holds row rowof(j); off[h] = byte offset inside a column of what
// this lane fetches for piece h (row rowof(32 h + lane / 2), half lane & 1).  Sites 1..8 in state_load_row's order.
__device__ __forceinline__ void state_load_row_dma(const ZkCols& w, u64 i, const u64 off[2], uint4* ring, u32 lane, StRow& R, u32& code) {
    R.flags = w.flags ? w.flags[i] : 0u;
#ifndef ST_DMA_BIG_LATE
    R.rwc = st_col(w, ST_RWC, i);
    const Fr is_write = st_col(w, ST_IS_WRITE, i);
    R.tag = st_col(w, ST_TAG, i);
    R.id = st_col(w, ST_ID, i);
    R.addr = st_col(w, ST_ADDR, i);
    R.ftag = st_col(w, ST_FIELD_TAG, i);
    R.key_lo = st_col(w, ST_KEY_LO, i);
    R.key_hi = st_col(w, ST_KEY_HI, i);
    R.val_lo = st_col(w, ST_VAL_LO, i);
    R.val_hi = st_col(w, ST_VAL_HI, i);
    R.init_lo = st_col(w, ST_INIT_LO, i);
    R.init_hi = st_col(w, ST_INIT_HI, i);
    R.root_lo = st_col(w, ST_ROOT_LO, i);
    R.root_hi = st_col(w, ST_ROOT_HI, i);
#endif
#pragma unroll
    for (int k = 0; k < ST_DMA_RING; k++) st_dma_issue(w, k, ring, off);
    u32 lc[5] = {0, 0, 0, 0, 0}, key[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bad_limb = 0, bad_byte = 0;
    // LDS byte address of this lane's cell in column 0 of slot 0 (64 rows x 32 B per column)
    const u32 lds_lane = (u32)(size_t)(__attribute__((address_space(3))) void*)ring + lane * 32u;
    st_dma_steps<0>(w, ring, lds_lane, off, lc, key, bad_limb, bad_byte);
#ifdef ST_DMA_BIG_LATE
    R.rwc = st_col(w, ST_RWC, i);
    const Fr is_write = st_col(w, ST_IS_WRITE, i);
    R.tag = st_col(w, ST_TAG, i);
    R.id = st_col(w, ST_ID, i);
    R.addr = st_col(w, ST_ADDR, i);
    R.ftag = st_col(w, ST_FIELD_TAG, i);
    R.key_lo = st_col(w, ST_KEY_LO, i);
    R.key_hi = st_col(w, ST_KEY_HI, i);
    R.val_lo = st_col(w, ST_VAL_LO, i);
    R.val_hi = st_col(w, ST_VAL_HI, i);
    R.init_lo = st_col(w, ST_INIT_LO, i);
    R.init_hi = st_col(w, ST_INIT_HI, i);
    R.root_lo = st_col(w, ST_ROOT_LO, i);
    R.root_hi = st_col(w, ST_ROOT_HI, i);
#endif
    U256 lcv = fr_zero(), keyv;
#pragma unroll
    for (int k = 0; k < 5; k++) lcv.v[k] = lc[k];
#pragma unroll
    for (int k = 0; k < 8; k++) keyv.v[k] = key[k];
    ST_ASSERT(fr_fits64(R.tag) && fr_lo64(R.tag) >= 1 && fr_lo64(R.tag) <= 12, 1);
    ST_ASSERT(fr_le_u64(R.id, (1ull << 28) - 1), 2);
    ST_ASSERT(fr_le_u64(R.ftag, 24), 3);
    ST_ASSERT(!bad_limb, 4);
    ST_ASSERT(fr_eq(R.addr, lcv), 5);
    R.pack_ok = bad_byte ? 0u : 1u;
    ST_ASSERT(!bad_byte, 6);
    ST_ASSERT(fr_eq(R.key_lo, u256_lo(keyv)) && fr_eq(R.key_hi, u256_hi(keyv)), 7);
    ST_ASSERT(fr_le_u64(is_write, 1), 8);
    R.is_write01 = fr_is_zero(is_write) ? 0u : (fr_eq_u64(is_write, 1) ? 1u : 2u);
    {
        Big18 out;
        for (int k = 0; k < 18; k++) out.v[k] = 0;
        big_shl_add(out, 0, 0, R.tag);
        big_shl_add(out, 0, 28, R.id);
        big_shl_add(out, 5, 0, R.addr);
        big_shl_add(out, 0, 16, R.ftag);
        big_shl_add(out, 1, 0, keyv);
        big_shl_add(out, 1, 0, R.rwc);
        out.v[15] &= 0xffffu;
#pragma unroll
        for (int k = 0; k < 16; k++) R.pack[k] = out.v[k];
    }
}
#endif

#ifdef ZK_HOSTSIM
// Evaluate row i against prev = (i-1) mod n and next = (i+1) mod n (host build: both rows loaded).
ZK_HD u32 state_check_row(const StateArgs& a, u64 i) {
    const u64 n = a.rows.n;
    StRow C, P;
    u32 code = 0, pcode = 0;
    state_load_row(a.rows, i == 0 ? n - 1 : i - 1, P, pcode);
    state_load_row(a.rows, i, C, code);
    return state_check_loaded(a, i, C, P, code);
}
#endif
