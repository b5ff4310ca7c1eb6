// State-circuit witness assignment on the device.
//
// Replaces the reference's `assign_state_circuit` (src/zkevm_specs/state_circuit.py:855-884: `op2row`
// :827-852 per op, root back-fill :866-878) and `mpt_table_from_ops` (:887-888, `_mock_mpt_updates`
// :904-934).  Input: the list of `Operation`s (:616-630) column-major, 12 slots of 256 bits per op;
// output: the 57-cell State-circuit rows in the column-major layout `zk_state_open` takes, their
// type bits, and the mock MPT table rows.
//
// Per-op pieces (shared with the g++ -DZK_HOSTSIM logic harness):
//   asg_insert      first op of every distinct MPT key (address, field_tag, storage_key) -> open-addressing
//                   slot holding the SMALLEST op index with that key (atomicMin)
//   asg_find_first  op -> that smallest index
//   asg_write_mpt   MPTTableRow of a first-occurrence op (rank r: root_prev = 3 + 5 r, root = root_prev + 5)
//   asg_write_row   op2row: reductions mod p, 16-bit address limbs, storage-key bytes, root, type bits
//   asg_status      exception class of the reference for this op (see include/zkevm_hip.h)
// The ranks (exclusive prefix count of first occurrences) and the "next MPT-keyed op" suffix-min are
// computed with wave ballots + one small single-block scan over per-block partials (zkevm_hip.hip).
#pragma once
#include "common.hpp"
#include "state_rekey.hpp"

enum { ASG_NSLOTS = 12, ASG_ROW_NCELLS = 57, ASG_MPT_NCELLS = 12 };
enum { ASG_RWC = 0, ASG_RW, ASG_TAG, ASG_ID, ASG_ADDR, ASG_FT, ASG_KEY, ASG_VLO, ASG_VHI, ASG_ILO, ASG_IHI, ASG_LEX };
enum { ASG_SITE_MPT_VALUE = 1, ASG_SITE_MPT_INITIAL = 2, ASG_SITE_ADDRESS = 3, ASG_SITE_FIELD_TAG = 4 };
#define ASG_NONE 0xffffffffu
#define ASG_BLOCK 256

struct AssignArgs {
    const u64* ops;      // [12][n][4] column-major
    const u32* op_flags; // [n] bit0 value.is_word, bit1 initial_value.is_word, bit2 field_tag is an AccountFieldTag
    u64 n;
    // The ops may also be read straight from an EVM-circuit RW table through the sorted order of its rows (state_rekey.hpp:
    // zk_state_assign_from_rw): op 0 = StartOp, op i = the re-keyed RW row order[i - 1]; `ops` / `op_flags` are then unused.
    u32 compact;         // 1: rows out are the 15 cells of ZK_OPT_STATE_COMPACT ([15][n][4]: no limb / byte columns)
    u32 pad_;
    const u64* rw;       // [n_rw][14][4] or nullptr
    const u32* rw_flags; // [n_rw] or nullptr
    const u32* order;    // [n - 1]
    u64* rows;           // out [57][n][4]
    u32* row_flags;      // out [n]
    u64* mpt;            // out [n_mpt][12][4] (capacity n rows), first-occurrence order
    u32* slots;          // [mask + 1] MPT key -> smallest op index
    u32 mask;
    u32 nb;              // number of ASG_BLOCK-op blocks
    u32* first;          // [n] smallest op index with the same MPT key, ASG_NONE for un-keyed ops
    u32* rank;           // [n] rank of a first-occurrence op among the first occurrences
    u32* blk_cnt;        // [nb + 1] first occurrences per block -> exclusive prefix; [nb] = n_mpt
    u32* blk_next;       // [nb + 1] smallest keyed op index per block -> min over the blocks AFTER b
    // Rows evaluated where they are computed (zk_state_verify_from_rw, state_fused.hpp): `rows` is null, every op's root comes out as
    // the rank it is made of, and the rank pass enters every MPT row into the State circuit's MPT index as it writes it.
    u32* root_rank;      // out [n] or nullptr: the row's root = 3 + 5 * root_rank[i]
    u32* mpt_slots;      // [mpt_mask + 1] or nullptr: the MPT index (slot value = row | hash fingerprint << 24; n < 2^24 - 1)
    u32 mpt_mask;
    u32 pad2_;
};

// Slot s of op i when the ops are the re-keyed rows of an RW table (the mapping of rwk_key / rwk_op, one slot at a time: `s` is a
// compile-time constant at every call site, so only that slot's cells are loaded).
// (`cell(k)`: cell k of the RW row — read from memory one at a time here, from registers where a whole row was loaded in one batch,
// state_fused.hpp)
template <class CELL>
ZK_HD Fr asg_slot_of_rw_row(CELL cell, u32 s) {
    if (s == ASG_RWC) return cell(0);
    if (s == ASG_RW) return cell(1);
    if (s == ASG_VLO) return cell(8);
    if (s == ASG_VHI) return cell(9);
    if (s == ASG_LEX) return fr_from_u64(1);
    const u32 tag = rwk_tag_of_target(cell(2).v[0]);  // (rows in `order` hold one of Target's values)
    if (s == ASG_TAG) return fr_from_u64(tag);
    if (s == ASG_ID) return tag == 6u ? fr_zero() : cell(3);
    if (s == ASG_ILO) return (tag == 4u || tag == 6u) ? cell(12) : fr_zero();
    if (s == ASG_IHI) return (tag == 4u || tag == 6u) ? cell(13) : fr_zero();
    if (s == ASG_KEY && tag != 10u) {
        Fr key = cell(6);
        const Fr khi = cell(7);
        key.v[4] |= khi.v[0]; key.v[5] |= khi.v[1]; key.v[6] |= khi.v[2]; key.v[7] |= khi.v[3];
        return key;
    }
    if (s == ASG_FT && tag != 5u && tag != 10u) return cell(5);
    const Fr c4 = cell(4);
    if (s == ASG_ADDR) return tag == 5u ? fr_zero() : (tag == 10u ? rwk_shr(c4, 48) : c4);
    if (s == ASG_FT) return tag == 5u ? c4 : fr_from_u64((u64)(c4.v[1] & 0xffffu));
    return fr_from_u64((u64)c4.v[0]);  // ASG_KEY of a TxLog row
}
ZK_HD Fr asg_slot_rw(const AssignArgs& a, u32 s, u64 i) {
    if (i == 0) return s == ASG_TAG ? fr_from_u64(1) : fr_zero();  // StartOp (rwk_emit_start)
    const u64* p = a.rw + (u64)a.order[i - 1] * (RWK_RW_NCELLS * 4);
    return asg_slot_of_rw_row([p](int k) { return rwk_cell(p, k); }, s);
}
// RW: the ops are the re-keyed rows of an RW table (a compile-time switch: the device has one instantiation of every assignment
// kernel per source, so the op-list form pays nothing for the other one — a run-time branch cost assign_rows_kernel 14 VGPRs, +13 %)
template <bool RW = false>
ZK_HD Fr asg_slot(const AssignArgs& a, u32 s, u64 i) {
    if (RW) return asg_slot_rw(a, s, i);
    return fr_load(a.ops + ((u64)s * a.n + i) * 4);
}
template <bool RW = false>
ZK_HD u32 asg_flags(const AssignArgs& a, u64 i) {
    if (!RW) return a.op_flags[i];
    if (i == 0) return 0u;
    const u32 r = a.order[i - 1];
    const u32 tag = rwk_tag_of_target(rwk_cell(a.rw + (u64)r * (RWK_RW_NCELLS * 4), 2).v[0]);
    return ((a.rw_flags ? a.rw_flags[r] : 0u) & 1u) | ((tag == 4u || tag == 6u) ? 2u : 0u) | (tag == 6u ? 4u : 0u);
}
// FQ(int) of a 256-bit Python int: x < 2^256 < 6p
ZK_HD Fr asg_reduce(Fr x) {
    const Fr p = fr_modulus();
    if (x.v[7] < p.v[7]) return x;  // below 2^224 * (top limb of p): already reduced — every well-formed slot (the subtraction ladder below
                                     // is 5 x 16 limb operations for each of the six reduced slots of a row)
#pragma unroll
    for (int it = 0; it < 5; it++) {
        Fr t;
        const u32 bw = u256_sub(t, x, p);
#pragma unroll
        for (int j = 0; j < 8; j++) x.v[j] = bw ? x.v[j] : t.v[j];
    }
    return x;
}
// `op.tag != Tag.Account and op.tag != Tag.Storage` on the raw int (state_circuit.py:898)
ZK_HD bool asg_has_key(const Fr& tag) { return fr_eq_u64(tag, 4) || fr_eq_u64(tag, 6); }

struct AsgKey {
    Fr addr, ft, key;  // FQ(address), FQ(field_tag), storage_key (lo/hi halves compared together)
};
template <bool RW = false>
ZK_HD AsgKey asg_key_of(const AssignArgs& a, u64 i) {
    AsgKey k;
    k.addr = asg_reduce(asg_slot<RW>(a, ASG_ADDR, i));
    k.ft = asg_reduce(asg_slot<RW>(a, ASG_FT, i));
    k.key = asg_slot<RW>(a, ASG_KEY, i);
    return k;
}
ZK_HD bool asg_key_eq(const AsgKey& x, const AsgKey& y) { return fr_eq(x.addr, y.addr) && fr_eq(x.ft, y.ft) && fr_eq(x.key, y.key); }
ZK_HD u64 asg_key_hash(const AsgKey& k) { return zk_hash_cell(zk_hash_cell(zk_hash_cell(0x6d7074u, k.addr), k.ft), k.key); }

// Claim / join the slot of op i's key; afterwards the slot holds the smallest op index with that key.
template <bool RW = false>
ZK_HD void asg_insert(const AssignArgs& a, u32 i) {
    const AsgKey k = asg_key_of<RW>(a, i);
    u32 s = (u32)asg_key_hash(k) & a.mask;
    for (;;) {
#if defined(ZK_HOSTSIM)
        u32 cur = a.slots[s];
        if (cur == ZK_EMPTY_SLOT) { a.slots[s] = i; return; }
#else
        u32 cur = __atomic_load_n(&a.slots[s], __ATOMIC_RELAXED);
        if (cur == ZK_EMPTY_SLOT) {
            cur = atomicCAS(&a.slots[s], ZK_EMPTY_SLOT, i);
            if (cur == ZK_EMPTY_SLOT) return;
        }
#endif
        // `cur` is some op with the slot's key (the occupant only ever changes to an op with the same key)
        if (asg_key_eq(asg_key_of<RW>(a, cur), k)) {
#if defined(ZK_HOSTSIM)
            if (i < a.slots[s]) a.slots[s] = i;
#else
            atomicMin(&a.slots[s], i);
#endif
            return;
        }
        s = (s + 1) & a.mask;
    }
}
template <bool RW = false>
ZK_HD u32 asg_find_first(const AssignArgs& a, u32 i) {
    const AsgKey k = asg_key_of<RW>(a, i);
    u32 s = (u32)asg_key_hash(k) & a.mask;
    for (;;) {
        const u32 cur = a.slots[s];
        if (cur == ZK_EMPTY_SLOT) return ASG_NONE;  // unreachable after asg_insert
        if (cur == i || asg_key_eq(asg_key_of<RW>(a, cur), k)) return cur;
        s = (s + 1) & a.mask;
    }
}

ZK_HD void asg_store(u64* out, const Fr& x) {
    uint4* q = (uint4*)out;
    uint4 lo, hi;
    lo.x = x.v[0]; lo.y = x.v[1]; lo.z = x.v[2]; lo.w = x.v[3];
    hi.x = x.v[4]; hi.y = x.v[5]; hi.z = x.v[6]; hi.w = x.v[7];
    q[0] = lo;  // (plain stores: the two halves of a cell merge in L2; non-temporal stores were measured at 2.1x the kernel time, round 6)
    q[1] = hi;
}
ZK_HD void asg_store_u64(u64* out, u64 x) { asg_store(out, fr_from_u64(x)); }

// Word(x.int_value()) of a WordOrValue with cells (lo, hi): v = lo + (hi << 128); false when v >= 2^256
// (the sanity assert of Word.__init__, util/arithmetic.py:116).  out_lo / out_hi = the two 128-bit halves.
ZK_HD bool asg_word_of(const Fr& lo, const Fr& hi, Fr& out_lo, Fr& out_hi) {
    u64 c = (u64)lo.v[4] + hi.v[0];
    const u32 w4 = (u32)c; c >>= 32;
    c += (u64)lo.v[5] + hi.v[1];
    const u32 w5 = (u32)c; c >>= 32;
    c += (u64)lo.v[6] + hi.v[2];
    const u32 w6 = (u32)c; c >>= 32;
    c += (u64)lo.v[7] + hi.v[3];
    const u32 w7 = (u32)c; c >>= 32;
    out_lo = fr_zero();
    out_lo.v[0] = lo.v[0]; out_lo.v[1] = lo.v[1]; out_lo.v[2] = lo.v[2]; out_lo.v[3] = lo.v[3];
    out_hi = fr_zero();
    out_hi.v[0] = w4; out_hi.v[1] = w5; out_hi.v[2] = w6; out_hi.v[3] = w7;
    return c == 0 && (hi.v[4] | hi.v[5] | hi.v[6] | hi.v[7]) == 0;
}

// Status of a first-occurrence op inside _mock_mpt_updates (:904-934); 0 for every other op.
ZK_HD u32 asg_mock_status(u32 flags, const Fr& ft, const Fr& vlo, const Fr& vhi, const Fr& ilo, const Fr& ihi) {
    Fr t0, t1;
    if ((flags & 4u) && !(fr_fits64(ft) && fr_lo64(ft) >= 1 && fr_lo64(ft) <= 4)) return ZK_CODE(ZK_UNSUPPORTED, ASG_SITE_FIELD_TAG);
    if (!asg_word_of(vlo, vhi, t0, t1)) return ZK_CODE(ZK_ASSERT, ASG_SITE_MPT_VALUE);
    if (!asg_word_of(ilo, ihi, t0, t1)) return ZK_CODE(ZK_ASSERT, ASG_SITE_MPT_INITIAL);
    return 0;
}

// MPTTableRow of first-occurrence op i with rank r (:921-929): address, proof_type, storage_key lo/hi,
// root lo/hi, root_prev lo/hi, value lo/hi, value_prev lo/hi.
template <bool RW = false>
ZK_HD void asg_mpt_cells(const AssignArgs& a, u64 i, u32 r, Fr q[ASG_MPT_NCELLS]) {
    const u32 flags = asg_flags<RW>(a, i);
    const Fr ft = asg_slot<RW>(a, ASG_FT, i);
    const Fr key = asg_slot<RW>(a, ASG_KEY, i);
    const Fr vlo = asg_slot<RW>(a, ASG_VLO, i), vhi = asg_slot<RW>(a, ASG_VHI, i);
    const Fr ilo = asg_slot<RW>(a, ASG_ILO, i), ihi = asg_slot<RW>(a, ASG_IHI, i);
    q[0] = asg_reduce(asg_slot<RW>(a, ASG_ADDR, i));
    // isinstance(field_tag, AccountFieldTag) -> from_account_field_tag (table.py:341-350: Nonce..NonExisting -> 1..4), else StorageMod
    q[1] = fr_from_u64((flags & 4u) ? fr_lo64(ft) : 6ull);
    q[2] = u256_lo(key);
    q[3] = u256_hi(key);
    const u64 root_prev = 3ull + 5ull * r;
    q[4] = fr_from_u64(root_prev + 5);
    q[5] = fr_zero();
    q[6] = fr_from_u64(root_prev);
    q[7] = fr_zero();
    asg_word_of(vlo, vhi, q[8], q[9]);
    asg_word_of(ilo, ihi, q[10], q[11]);
}
template <bool RW = false>
ZK_HD void asg_write_mpt(const AssignArgs& a, u64 i, u32 r) {
    u64* out = a.mpt + (u64)r * (ASG_MPT_NCELLS * 4);
    Fr q[ASG_MPT_NCELLS];
    asg_mpt_cells<RW>(a, i, r, q);
#pragma unroll
    for (int c = 0; c < ASG_MPT_NCELLS; c++) asg_store(out + 4 * c, q[c]);
}

// op2row (:827-852) with the back-filled root; returns the op's status code.
template <bool RW = false>
ZK_HD u32 asg_write_row(const AssignArgs& a, u64 i, u64 root, bool is_first) {
    const u64 n = a.n;
    u64* rows = a.rows;
#define ASG_OUT(c) (rows + ((u64)(((c) >= 50 && a.compact) ? (c) - 42 : (c)) * n + i) * 4)
    const u32 flags = asg_flags<RW>(a, i);
    const Fr addr = asg_slot<RW>(a, ASG_ADDR, i);
    const Fr key = asg_slot<RW>(a, ASG_KEY, i);
    const Fr ft = asg_slot<RW>(a, ASG_FT, i);
    const Fr vlo = asg_slot<RW>(a, ASG_VLO, i), vhi = asg_slot<RW>(a, ASG_VHI, i);
    const Fr ilo = asg_slot<RW>(a, ASG_ILO, i), ihi = asg_slot<RW>(a, ASG_IHI, i);
    // RW source: every slot is read before the first store (the stores may alias the loads for the compiler: each later slot would
    // re-read the row's target cell); op-list source: the four slots are loaded where they are stored (8 fewer live registers each)
    const Fr rwc = RW ? asg_slot<RW>(a, ASG_RWC, i) : fr_zero(), rw = RW ? asg_slot<RW>(a, ASG_RW, i) : fr_zero();
    const Fr tag = RW ? asg_slot<RW>(a, ASG_TAG, i) : fr_zero(), id = RW ? asg_slot<RW>(a, ASG_ID, i) : fr_zero();
    const Fr lex = RW ? asg_slot<RW>(a, ASG_LEX, i) : fr_zero();
    asg_store(ASG_OUT(0), asg_reduce(RW ? rwc : asg_slot<RW>(a, ASG_RWC, i)));
    asg_store_u64(ASG_OUT(1), fr_is_zero(RW ? rw : asg_slot<RW>(a, ASG_RW, i)) ? 0 : 1);  // `op.rw == RW.Read` :829
    asg_store(ASG_OUT(2), asg_reduce(RW ? tag : asg_slot<RW>(a, ASG_TAG, i)));
    asg_store(ASG_OUT(3), asg_reduce(RW ? id : asg_slot<RW>(a, ASG_ID, i)));
    asg_store(ASG_OUT(4), asg_reduce(addr));
    asg_store(ASG_OUT(5), asg_reduce(ft));
    asg_store(ASG_OUT(6), u256_lo(key));
    asg_store(ASG_OUT(7), u256_hi(key));
    if (!a.compact) {
#pragma unroll
        for (int k = 0; k < 10; k++) asg_store_u64(ASG_OUT(8 + k), (addr.v[k >> 1] >> (16 * (k & 1))) & 0xffffu);
#pragma unroll
        for (int k = 0; k < 32; k++) asg_store_u64(ASG_OUT(18 + k), fr_byte(key, k));
    }
    asg_store(ASG_OUT(50), vlo);
    asg_store(ASG_OUT(51), vhi);
    asg_store(ASG_OUT(52), ilo);
    asg_store(ASG_OUT(53), ihi);
    asg_store_u64(ASG_OUT(54), root);
    asg_store_u64(ASG_OUT(55), 0);
    asg_store(ASG_OUT(56), RW ? lex : asg_slot<RW>(a, ASG_LEX, i));
#undef ASG_OUT
    a.row_flags[i] = flags & 3u;
    u32 code = is_first ? asg_mock_status(flags, ft, vlo, vhi, ilo, ihi) : 0u;
    // op.address.to_bytes(20, "little") :834 -> OverflowError
    if (!code && (addr.v[5] | addr.v[6] | addr.v[7])) code = ZK_CODE(ZK_OVERFLOW_ERROR, ASG_SITE_ADDRESS);
    return code;
}
