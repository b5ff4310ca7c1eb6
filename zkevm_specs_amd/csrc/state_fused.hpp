// State circuit over rows that are computed where they are evaluated (zk_state_verify_from_rw): verify(assign(ops)) without the
// witness in between.  assign_state_circuit's op2row (state_circuit.py:827-852, state_assign.hpp asg_write_row) gives a row's
// fifteen wide cells — the 42 limb / byte cells are decompositions of two of them and check_state_row's constraints on those
// (:504-517) hold by construction, exactly as in the ZK_OPT_STATE_COMPACT form — and check_state_row (:492-613,
// state_check_loaded) runs on them in registers.  The results are those of zk_state_assign_* followed by zk_state_verify on its
// output, bit for bit (tests/test_state_fused.py); what disappears is 1,824 B written and read back per row.
#pragma once
#include "state_assign.hpp"
#include "state_circuit.hpp"

// rows.cells is null (nothing reads it), rows.n = the number of ops, mpt = the assignment's MPT rows with the index the rank pass built
// (mpt.n = the number of OPS: a capacity — the lookup only uses it to tell an empty table and to size the hash fingerprint, and an
// index without entries already answers "not found").
template <bool RW>
struct StateFusedArgs : StateArgs {
    AssignArgs asg;
};

// Row i from op i: op2row's cells (asg_write_row), then the row's own checks and key packing as the compact loader's.
// asg_code = the op's assignment status (the mock MPT update's asserts and address.to_bytes(20): asg_write_row's return value).
template <class SLOT>
ZK_HD void state_row_from_slots(SLOT slot, u32 flags, bool is_first, u32 root_rank, StRow& R, u32& code, u32& asg_code) {
    const Fr addr = slot(ASG_ADDR);
    const Fr key = slot(ASG_KEY);
    const Fr ft = slot(ASG_FT);
    R.flags = flags & 3u;
    R.rwc = asg_reduce(slot(ASG_RWC));
    const Fr is_write = fr_from_u64(fr_is_zero(slot(ASG_RW)) ? 0 : 1);  // `op.rw == RW.Read` :829
    R.tag = asg_reduce(slot(ASG_TAG));
    R.id = asg_reduce(slot(ASG_ID));
    R.addr = asg_reduce(addr);
    R.ftag = asg_reduce(ft);
    R.key_lo = u256_lo(key);
    R.key_hi = u256_hi(key);
    R.val_lo = slot(ASG_VLO);
    R.val_hi = slot(ASG_VHI);
    R.init_lo = slot(ASG_ILO);
    R.init_hi = slot(ASG_IHI);
    R.root_lo = fr_from_u64(3ull + 5ull * (u64)root_rank);
    R.root_hi = fr_zero();
    state_finish_row_compact(R, is_write, code);
    asg_code = is_first ? asg_mock_status(flags, ft, R.val_lo, R.val_hi, R.init_lo, R.init_hi) : 0u;
    if (!asg_code && (addr.v[5] | addr.v[6] | addr.v[7])) asg_code = ZK_CODE(ZK_OVERFLOW_ERROR, ASG_SITE_ADDRESS);
}
template <bool RW>
ZK_HD void state_row_from_op(const AssignArgs& g, u64 i, StRow& R, u32& code, u32& asg_code) {
    const u32 root_rank = g.root_rank[i];
    const bool is_first = g.first[i] == (u32)i;
    if (!RW) {
        state_row_from_slots([&g, i](u32 s) { return asg_slot<false>(g, s, i); }, asg_flags<false>(g, i), is_first, root_rank, R, code, asg_code);
        return;
    }
    // the op is a re-keyed RW row reached through the sorted order: ALL fourteen cells of the row in one batch of loads behind the
    // one dependent read of the order — slot by slot (asg_slot_rw) a cell that depends on the row's tag is a further round trip after the
    // tag cell, three to four dependent latencies per row in a kernel that waits for memory 85 % of its time (round 6: 141 us for 694 k rows)
    const bool start = i == 0;  // StartOp (rwk_emit_start): no RW row — row 0 is loaded and not used
    const u32 r = start ? 0u : g.order[i - 1];
    const u64* p = g.rw + (u64)r * (RWK_RW_NCELLS * 4);
    Fr c[RWK_RW_NCELLS];
#pragma unroll
    for (int k = 0; k < RWK_RW_NCELLS; k++) c[k] = rwk_cell(p, k);
    const u32 rwf = g.rw_flags ? g.rw_flags[r] : 0u;
    const u32 tag = rwk_tag_of_target(c[2].v[0]);
    const u32 flags = start ? 0u : ((rwf & 1u) | ((tag == 4u || tag == 6u) ? 2u : 0u) | (tag == 6u ? 4u : 0u));  // asg_flags<true>
    state_row_from_slots([&c, start](u32 s) {
        if (start) return s == ASG_TAG ? fr_from_u64(1) : fr_zero();
        return asg_slot_of_rw_row([&c](int k) { return c[k]; }, s);
    }, flags, is_first, root_rank, R, code, asg_code);
}

// the two reads of state_check_loaded that are not in a StRow (state_circuit.hpp st_src_*), from the ops
template <bool RW>
ZK_HD Fr st_src_lex(const StateFusedArgs<RW>& a, u64 i) { return asg_slot<RW>(a.asg, ASG_LEX, i); }
template <bool RW>
ZK_HD u32 st_src_next_keys_diff(const StateFusedArgs<RW>& a, u64 j, const Fr* const mine[6]) {
    const AssignArgs& g = a.asg;
    const Fr key = asg_slot<RW>(g, ASG_KEY, j);
    const Fr c[6] = {asg_reduce(asg_slot<RW>(g, ASG_TAG, j)), asg_reduce(asg_slot<RW>(g, ASG_ID, j)), asg_reduce(asg_slot<RW>(g, ASG_ADDR, j)),
                     asg_reduce(asg_slot<RW>(g, ASG_FT, j)), u256_lo(key), u256_hi(key)};
    u32 d = 0;
#pragma unroll
    for (int x = 0; x < 6; x++)
#pragma unroll
        for (int k = 0; k < 8; k++) d |= c[x].v[k] ^ mine[x]->v[k];
    return d;
}
