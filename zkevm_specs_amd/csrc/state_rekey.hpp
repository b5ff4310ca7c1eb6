// RW table -> State-circuit operations on the device (the "RW-table lexicographic sort" half).
//
// The reference never links the EVM circuit's rw_table to the State circuit in code: its State
// witnesses are built from `Operation` objects (src/zkevm_specs/state_circuit.py:616-825) handed to assign_state_circuit
// (:855-884) in the order the circuit then checks, (tag, id, address, field_tag, storage_key, rw_counter) strictly
// increasing (:552-570).  A block's RW rows come out of `RWDictionary` (evm_circuit/typing.py:464-845) keyed by the EVM
// side's `Target` numbering (evm_circuit/table.py:184-204) with the key slots of each target:
//   Stack / Memory                        id = call_id, address = stack pointer / memory address        (:482-509)
//   CallContext                           id = call_id, the CallContextFieldTag travels in the ADDRESS cell (:510-531; instruction.py:895)
//   AccountStorage                        id = tx_id, address, storage_key; aux0 = the committed value   (:741-775)
//   Account                               address, field_tag; aux0 unused by the State side               (:700-740)
//   TxAccessListAccount(Storage), TxRefund   id = tx_id (+ address, storage_key)                          (:588-699)
//   TxLog                                 id = tx_id, address cell = log_id << 48 | field_tag << 32 | index (:532-587)
//   TxReceipt                             id = tx_id, field_tag                                            (:776-845)
// This header is the per-row half of that mapping (RW row -> one `Operation` in the wire form of zk_state_assign: 12 slots of
// 256 bits) and the order-preserving compact sort key; k_rekey.hip sorts with it (LSD radix over the compact key),
// cpu_backend.cpp with std::stable_sort.  Both follow `sorted(..., key=(tag, id, address, field_tag, storage_key, rw_counter))`
// of the checker (oracle/rw_state_oracle.py): stable, so rows with equal keys keep their table order.
//
// Domain: every cell is taken as the 256-bit integer on the wire.  Rejected with a status code (never guessed):
//   site 1  the target cell is not one of Target's eleven values (KeyError in the checker)            -> ZK_VALUE_ERROR
//   site 2  storage_key hi cell >= 2^128: lo | hi << 128 does not fit the 256-bit slot               -> ZK_OVERFLOW_ERROR
// A rejected row is left out of the output like a dropped one.
// Dropped (not an error): CallContext rows whose field tag exceeds the State circuit's MAX_FIELD_TAG (24,
// state_circuit.py:34,334): CallContextFieldTag.ReversibleWriteCounter is 25, a row the reference's own State circuit cannot carry.
#pragma once
#include "common.hpp"

enum { RWK_RW_NCELLS = 14, RWK_NSLOTS = 12, RWK_NFIELDS = 5, RWK_NCLASSES = 16, RWK_CLASS_DROPPED = 15 };
enum { RWK_F_ID = 0, RWK_F_ADDR = 1, RWK_F_FT = 2, RWK_F_KEY = 3, RWK_F_RWC = 4, RWK_F_RANK0 = 5 /* + field: its rank column */ };
enum { RWK_SITE_TARGET = 1, RWK_SITE_KEY = 2 };
#define RWK_MAX_FIELD_TAG 24u
#define RWK_MAX_RUNS 44  // per class: at most 8 word runs per field (5 fields) + slack

// State-circuit Tag of an EVM-side Target (table.py:184-204 -> state_circuit.py:42-60); 0 = not a Target
ZK_HD u32 rwk_tag_of_target(u32 target) {
    // Target:  Start 1, TxAccessListAccount 2, TxAccessListAccountStorage 3, TxRefund 4, Account 5, AccountStorage 6, CallContext 7,
    //          Stack 8, Memory 9, TxLog 10, TxReceipt 11
    // Tag:     Start 1, Memory 2, Stack 3, Storage 4, CallContext 5, Account 6, TxRefund 7, TxAccessListAccount 8,
    //          TxAccessListAccountStorage 9, TxLog 10, TxReceipt 11
    const u64 packed = 0xBA2354679810ull;  // nibble t = tag of target t
    return target < 12u ? (u32)((packed >> (4u * target)) & 0xfu) : 0u;
}

struct RwkKey {      // what the sort looks at
    u32 cls;         // state tag 1..11, RWK_CLASS_DROPPED for rows left out
    u32 status;      // 0, or the reject code (then cls == RWK_CLASS_DROPPED)
    Fr f[RWK_NFIELDS];
};
struct RwkOp {       // the other slots of the operation
    Fr rw, vlo, vhi, ilo, ihi;
    u32 flags;       // bit0 value.is_word, bit1 initial_value.is_word, bit2 field_tag is an AccountFieldTag
};

ZK_HD Fr rwk_cell(const u64* row, int c) { return fr_load(row + 4 * c); }
ZK_HD Fr rwk_shr(const Fr& x, int bits) {  // bits in {32, 48}
    Fr r = fr_zero();
    if (bits == 32) {
#pragma unroll
        for (int i = 0; i < 7; i++) r.v[i] = x.v[i + 1];
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++) r.v[i] = (x.v[i + 1] >> 16) | ((i + 2 < 8 ? x.v[i + 2] : 0u) << 16);
    }
    return r;
}

// The sort key of one RW row (uint64[14][4]).
ZK_HD RwkKey rwk_key(const u64* row) {
    RwkKey k;
    k.status = 0;
    const Fr target = rwk_cell(row, 2);
    const u32 tag = fr_fits32(target) ? rwk_tag_of_target(target.v[0]) : 0u;
    k.cls = tag;
    k.f[RWK_F_RWC] = rwk_cell(row, 0);
    k.f[RWK_F_ID] = rwk_cell(row, 3);
    const Fr c4 = rwk_cell(row, 4);
    k.f[RWK_F_ADDR] = c4;
    k.f[RWK_F_FT] = rwk_cell(row, 5);
    const Fr klo = rwk_cell(row, 6), khi = rwk_cell(row, 7);
    // storage_key = lo | hi << 128 as Python ints
    Fr key = klo;
    key.v[4] |= khi.v[0]; key.v[5] |= khi.v[1]; key.v[6] |= khi.v[2]; key.v[7] |= khi.v[3];
    k.f[RWK_F_KEY] = key;
    if (tag == 0u) {
        k.cls = RWK_CLASS_DROPPED;
        k.status = ZK_CODE(ZK_VALUE_ERROR, RWK_SITE_TARGET);
        return k;
    }
    if (!fr_fits128(khi)) {
        k.cls = RWK_CLASS_DROPPED;
        k.status = ZK_CODE(ZK_OVERFLOW_ERROR, RWK_SITE_KEY);
        return k;
    }
    if (tag == 5u) {  // CallContext: the field tag travels in the address cell
        k.f[RWK_F_ADDR] = fr_zero();
        k.f[RWK_F_FT] = c4;
        if (!fr_le_u64(c4, RWK_MAX_FIELD_TAG)) k.cls = RWK_CLASS_DROPPED;
    } else if (tag == 6u) {  // Account: no id
        k.f[RWK_F_ID] = fr_zero();
    } else if (tag == 10u) {  // TxLog: log_id << 48 | field_tag << 32 | index
        k.f[RWK_F_ADDR] = rwk_shr(c4, 48);
        k.f[RWK_F_FT] = fr_from_u64((u64)(c4.v[1] & 0xffffu));
        k.f[RWK_F_KEY] = fr_from_u64((u64)c4.v[0]);
    }
    return k;
}
// The remaining slots (value, initial value, rw, flags) of a row whose key is `k`.
ZK_HD RwkOp rwk_op(const u64* row, u32 rw_flags, const RwkKey& k) {
    RwkOp o;
    o.rw = rwk_cell(row, 1);
    o.vlo = rwk_cell(row, 8);
    o.vhi = rwk_cell(row, 9);
    o.ilo = fr_zero();
    o.ihi = fr_zero();
    o.flags = rw_flags & 1u;
    if (k.cls == 4u || k.cls == 6u) {  // AccountStorage / Account: initial_value = aux0 (the committed value)
        o.ilo = rwk_cell(row, 12);
        o.ihi = rwk_cell(row, 13);
        o.flags |= 2u | (k.cls == 6u ? 4u : 0u);
    }
    return o;
}

// ---- the compact, order-preserving key ------------------------------------------------------------------------------------
// Within one class (state tag) a key bit that is the same in every row of the class cannot decide a comparison, and rows of
// different classes are ordered by the tag alone.  The plan lists, per class and most significant first, the bit runs of the five
// key fields that vary inside the class; a row's compact key is   tag (4 bits) | its runs | zero padding   right-aligned in
// key_words 32-bit words.  A field whose varying bits are many but whose class is small is replaced by its RANK among the class's
// values (k_rekey.hip rwk_rank_kernel; runs with field >= RWK_F_RANK0 read the rank column), which keeps config 5's storage rows
// (160-bit addresses, 256-bit keys: a few thousand rows) from costing every row 50 more radix passes.
struct RwkRun {
    uint8_t field, word, shift, nbits;  // bits [shift, shift + nbits) of 32-bit word `word` of `field`
};
struct RwkClassPlan {
    u32 n_runs;
    u32 width;                  // payload bits of this class
    RwkRun runs[RWK_MAX_RUNS];
};
struct RwkPlan {
    RwkClassPlan cls[RWK_NCLASSES];
    u32 key_bits;               // 4 + max class width
    u32 key_words;              // ceil(key_bits / 32)
    u32 n_passes;               // ceil(key_bits / 8) radix passes of 8 bits
    u32 pad;
};

struct RwkRankJob {
    u32 cls, field, base, count;  // members of class `cls` get the rank of their `field`; they sit at [base, base + count) of job_rows / job_vals
};
#define RWK_MAX_JOBS 8
struct RekeyArgs {
    const u64* rw;        // [n][14][4]
    const u32* rw_flags;  // [n] or nullptr
    u64 n;
    u32* masks;           // [2][16][5][8] OR / OR-of-complements per class and field word, then [16] rows per class
    const RwkPlan* plan;  // device copy of the host-built plan
    u32* keys;            // [key_words][n] compact keys, word 0 most significant
    u32 key_words, n_passes;
    u32* ranks[RWK_NFIELDS];  // rank column of a field (nullptr: the plan does not rank it)
    RwkRankJob jobs[RWK_MAX_JOBS];
    u32 n_jobs;
    u32 ntiles;
    u32* job_cursor;      // [n_jobs]
    u32* job_rows;        // members of the jobs' classes
    u64* job_vals;        // their field values
    u32* idx_a; u32* idx_b;  // the two index buffers of the radix passes
    u32* hist;            // [ntiles][256]
    u64* ops;             // out [12][n_ops][4]
    u32* op_flags;        // out [n_ops]
    u64 n_ops;            // 1 (StartOp) + kept rows
    // fast path (keys of at most 64 bits, at most 8 passes): the keys travel with the indices, one kernel per pass (chained scan)
    u32 fast;
    u32 ntiles_fast;
    u64* key64_a; u64* key64_b;  // [n] each
    u32* sweep;           // zeroed per launch: ghist [8][256], ticket [8], error flag [8], then desc [n_passes][ntiles_fast][256]
};
#define RWK_SWEEP_HEAD (8 * 256 + 16)
// one sweep tile = one 1024-thread block over RWK_SW_ITEMS keys per thread
#define RWK_SW_BLOCK 1024
#ifndef RWK_SW_ITEMS
#define RWK_SW_ITEMS 8
#endif
#define RWK_SW_TILE (RWK_SW_BLOCK * RWK_SW_ITEMS)

// Append `nbits` (<= 32) to the MSB-first bit stream (acc: pending bits right-aligned, cnt of them); full 32-bit words go to
// out[w * stride] (w counts up).
ZK_HD void rwk_put(u64& acc, u32& cnt, u32& w, u32* out, u64 stride, u32 bits, u32 nbits) {
    acc = (acc << nbits) | (u64)bits;
    cnt += nbits;
    if (cnt >= 32u) {
        cnt -= 32u;
        out[(u64)w * stride] = (u32)(acc >> cnt);
        w++;
        acc &= (cnt ? ((1ull << cnt) - 1ull) : 0ull);
    }
}
// Compact key of a row: words out[w * stride], w = 0 (most significant) .. key_words - 1.  ranks[f] = the row's rank in field f
// where the plan asks for it (else ignored).
ZK_HD void rwk_pack(const RwkPlan& pl, const RwkClassPlan& cp, const RwkKey& k, const u32* ranks, u32* out, u64 stride) {
    u64 acc = 0;
    u32 cnt = 0, w = 0;
    const u32 total = pl.key_words * 32u;
    u32 lead = total - pl.key_bits;
    while (lead) {  // (at most 31 bits)
        const u32 t = lead > 16u ? 16u : lead;
        rwk_put(acc, cnt, w, out, stride, 0u, t);
        lead -= t;
    }
    rwk_put(acc, cnt, w, out, stride, k.cls & 0xfu, 4u);
    u32 used = 0;
    if (k.cls != RWK_CLASS_DROPPED) {
        for (u32 r = 0; r < cp.n_runs; r++) {
            const RwkRun run = cp.runs[r];
            u32 word;
            if (run.field >= RWK_F_RANK0) {
                word = 0;
#pragma unroll
                for (int fi = 0; fi < RWK_NFIELDS; fi++)
                    if (run.field == RWK_F_RANK0 + fi) word = ranks[fi];
            } else {
                // (run.word is a runtime index: pick the word with selects, not a dynamically indexed register array)
                word = 0;
#pragma unroll
                for (int fi = 0; fi < RWK_NFIELDS; fi++)
#pragma unroll
                    for (int wi = 0; wi < 8; wi++)
                        if (run.field == fi && run.word == wi) word = k.f[fi].v[wi];
            }
            const u32 bits = run.nbits == 32u ? word : ((word >> run.shift) & ((1u << run.nbits) - 1u));
            rwk_put(acc, cnt, w, out, stride, bits, run.nbits);
            used += run.nbits;
        }
    }
    u32 rest = pl.key_bits - 4u - used;
    while (rest) {
        const u32 t = rest > 16u ? 16u : rest;
        rwk_put(acc, cnt, w, out, stride, 0u, t);
        rest -= t;
    }
}

// One `Operation` in the wire form of zk_state_assign (ops uint64[12][n_ops][4] column-major, slot s of op j at ops + (s * n_ops + j) * 4).
ZK_HD void rwk_store(u64* out, const Fr& x) {
    uint4* q = (uint4*)out;
    uint4 lo, hi;
    lo.x = x.v[0]; lo.y = x.v[1]; lo.z = x.v[2]; lo.w = x.v[3];
    hi.x = x.v[4]; hi.y = x.v[5]; hi.z = x.v[6]; hi.w = x.v[7];
    q[0] = lo;
    q[1] = hi;
}
ZK_HD void rwk_emit(u64* ops, u32* op_flags, u64 n_ops, u64 j, const RwkKey& k, const RwkOp& o) {
#define RWK_OUT(s) (ops + ((u64)(s) * n_ops + j) * 4)
    rwk_store(RWK_OUT(0), k.f[RWK_F_RWC]);
    rwk_store(RWK_OUT(1), o.rw);
    rwk_store(RWK_OUT(2), fr_from_u64(k.cls));
    rwk_store(RWK_OUT(3), k.f[RWK_F_ID]);
    rwk_store(RWK_OUT(4), k.f[RWK_F_ADDR]);
    rwk_store(RWK_OUT(5), k.f[RWK_F_FT]);
    rwk_store(RWK_OUT(6), k.f[RWK_F_KEY]);
    rwk_store(RWK_OUT(7), o.vlo);
    rwk_store(RWK_OUT(8), o.vhi);
    rwk_store(RWK_OUT(9), o.ilo);
    rwk_store(RWK_OUT(10), o.ihi);
    rwk_store(RWK_OUT(11), fr_from_u64(1));  // lexicographic_ordering_selector
#undef RWK_OUT
    op_flags[j] = o.flags;
}
// StartOp(rw_counter 0, lexicographic_ordering_selector 0) in front (state_circuit.py:634-645)
ZK_HD void rwk_emit_start(u64* ops, u32* op_flags, u64 n_ops) {
    for (u32 s = 0; s < RWK_NSLOTS; s++) rwk_store(ops + ((u64)s * n_ops) * 4, s == 2 ? fr_from_u64(1) : fr_zero());
    op_flags[0] = 0;
}
