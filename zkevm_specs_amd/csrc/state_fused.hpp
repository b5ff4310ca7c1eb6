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
template <bool RW>
ZK_HD void state_row_from_op(const AssignArgs& g, u64 i, StRow& R, u32& code, u32& asg_code) {
    const u32 flags = asg_flags<RW>(g, i);
    const Fr addr = asg_slot<RW>(g, ASG_ADDR, i);
    const Fr key = asg_slot<RW>(g, ASG_KEY, i);
    const Fr ft = asg_slot<RW>(g, ASG_FT, i);
    R.flags = flags & 3u;
    R.rwc = asg_reduce(asg_slot<RW>(g, ASG_RWC, i));
    const Fr is_write = fr_from_u64(fr_is_zero(asg_slot<RW>(g, ASG_RW, i)) ? 0 : 1);  // `op.rw == RW.Read` :829
    R.tag = asg_reduce(asg_slot<RW>(g, ASG_TAG, i));
    R.id = asg_reduce(asg_slot<RW>(g, ASG_ID, i));
    R.addr = asg_reduce(addr);
    R.ftag = asg_reduce(ft);
    R.key_lo = u256_lo(key);
    R.key_hi = u256_hi(key);
    R.val_lo = asg_slot<RW>(g, ASG_VLO, i);
    R.val_hi = asg_slot<RW>(g, ASG_VHI, i);
    R.init_lo = asg_slot<RW>(g, ASG_ILO, i);
    R.init_hi = asg_slot<RW>(g, ASG_IHI, i);
    R.root_lo = fr_from_u64(3ull + 5ull * (u64)g.root_rank[i]);
    R.root_hi = fr_zero();
    state_finish_row_compact(R, is_write, code);
    asg_code = g.first[i] == (u32)i ? asg_mock_status(flags, ft, R.val_lo, R.val_hi, R.init_lo, R.init_hi) : 0u;
    if (!asg_code && (addr.v[5] | addr.v[6] | addr.v[7])) asg_code = ZK_CODE(ZK_OVERFLOW_ERROR, ASG_SITE_ADDRESS);
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
