// EVM circuit: per-step constraint evaluation (one execution step pair (curr, next) per lane).
//
// Reference: src/zkevm_specs/evm_circuit/main.py:14-63 (`verify_steps` / `verify_step`), the
// `Instruction` toolbox in evm_circuit/instruction.py and the execution-state gadgets in
// evm_circuit/execution/*.py; lookup semantics evm_circuit/table.py:864-884.
//
// Wire layouts (all cells canonical 4xu64):
//   steps   ROW-major [n_steps][13] (lanes evaluate steps in state-sorted order, so a step's 13
//           cells are kept contiguous - 416 B, 4 cache lines - instead of 13 strided columns):
//           execution_state, rw_counter, call_id, is_root, is_create,
//           code_hash lo,hi, program_counter, stack_pointer, gas_left, memory_word_size,
//           reversible_write_counter, log_id                      (evm_circuit/step.py:16-75)
//   rw      row-major [n][14]: rw_counter, rw, key0(Target), id, address, field_tag, storage_key lo,hi,
//           value lo,hi, value_prev lo,hi, aux0 lo,hi; flags bit0 value.is_word bit1 value_prev.is_word
//                                                                  (evm_circuit/table.py:447-457)
//   bytecode row-major [n][6]: hash lo,hi, field_tag, index, is_code, value   (table.py:438-443)
//   tx      row-major [n][5]: tx_id, field_tag, index, value lo,hi; flags bit0 value.is_word (:421-426)
//   block   row-major [n][4]: field_tag, block_number, value lo,hi; flags bit0 value.is_word (:413-417)
//
// Status code of a step = (kind << 24) | seq: `seq` is the ordinal of the failing *checkpoint*
// (every primitive that can raise counts one, in the reference's evaluation order) — the same
// numbering oracle/evm_oracle.py uses, so complete codes are comparable in the parity tests.
#pragma once
#include "common.hpp"
#include "evm_tables.h"
#include "keccak.hpp"
#include "secp_constants.h"
#include "bigz.hpp"

enum { S_STATE = 0, S_RWC, S_CALL_ID, S_IS_ROOT, S_IS_CREATE, S_CH_LO, S_CH_HI, S_PC, S_SP, S_GAS, S_MWS, S_REV, S_LOG, STEP_NCELLS };
enum { R_RWC = 0, R_RW, R_TAG, R_ID, R_ADDR, R_FT, R_KEY_LO, R_KEY_HI, R_VAL_LO, R_VAL_HI, R_PREV_LO, R_PREV_HI, R_AUX_LO, R_AUX_HI, RW_NCELLS };
enum { B_HASH_LO = 0, B_HASH_HI, B_TAG, B_INDEX, B_IS_CODE, B_VALUE, BYTECODE_NCELLS };
enum { TX_NCELLS = 5, BLOCK_NCELLS = 4 };
// CopyTableRow (table.py:494-507) and ExpTableRow (:538-548) as the EVM circuit looks them up
enum { CT_IS_FIRST = 0, CT_SRC_ID_LO, CT_SRC_ID_HI, CT_SRC_TAG, CT_DST_ID_LO, CT_DST_ID_HI, CT_DST_TAG, CT_SRC_ADDR,
       CT_SRC_ADDR_END, CT_DST_ADDR, CT_LENGTH, CT_RLC_ACC, CT_RWC, CT_RWC_INC, COPY_T_NCELLS };
enum { XT_IS_STEP = 0, XT_ID, XT_IS_LAST, XT_BASE0, XT_BASE1, XT_BASE2, XT_BASE3, XT_EXP_LO, XT_EXP_HI, XT_RES_LO, XT_RES_HI,
       EXP_T_NCELLS };
// SigTableRow (table.py:552-558) and EccTableRow (:562-575): every lookup gives all the fields
enum { SIG_T_NCELLS = 9, ECC_T_NCELLS = 13 };
enum { AUX_PAIR = 3, AUX_ECRECOVER = 5, AUX_ECADD = 6, AUX_ECMUL = 7, AUX_ECPAIRING = 8 };  // flatten_step_aux kinds
enum { CDT_Bytecode = 1, CDT_Memory, CDT_TxCalldata, CDT_TxLog, CDT_RlcAcc };  // CopyDataTypeTag (table.py:336-353)

// Session-open verdicts that stay in HBM: zk_evm_open enqueues the kernels that compute them and returns without reading
// anything back (no host synchronisation at open); the evaluation kernels patch their copy of EvmArgs from this block at
// entry (evm_args_resolve).  Zero-initialised, so every field is phrased such that 0 is the "nothing built" state.
struct EvmDyn {
    u32 rw_sparse;      // set by rw_prepare_kernel when the RW rows are NOT consecutive rw_counters from rw_base
    u32 codes_n;        // directory entries (0: no directory, generic bytecode index only)
    u64 rw_base;
    u32 codes_mask;
    u32 agg_max_txs, agg_total_txs, agg_invalid_txs, agg_bad_invalid_rows, agg_total_wds;
    u32 dir_entries;    // directory build: groups counted so far (may exceed the capacity; then codes_n stays 0)
    u32 n_deferred;     // pairs the fast (hot) kernel handed to the general build this pass (reset at the start of every pass)
};

struct EvmArgs {
    const EvmDyn* dyn;  // optional (device sessions): see EvmDyn; nullptr = the fields below are final (CPU logic harness)
    const u64* steps;  // [n_steps][13][4]
    const u32* step_recs;  // device sessions: [n_steps][EVM_REC_WORDS] packed step records (see "step records"); nullptr on the CPU
    u64 n_steps;
    ZkTable rw, bytecode, tx, block;
    ZkTable copy, keccak, exp;  // optional (n == 0 when the trace has no copy / SHA3 / EXP steps)
    ZkTable sig, ecc;           // optional (only the ecRecover / ecAdd / ecMul / ecPairing precompile states read them)
    ZkTable withdrawals;        // optional WithdrawalTableRow rows (id, validator_id, address, amount), sorted by id
    // whole-table aggregates EndBlock's last step needs (end_block.py:55-91), computed once per session on the host
    u32 agg_max_txs, agg_total_txs, agg_invalid_txs, agg_bad_invalid_rows, agg_total_wds;
    const u64* aux;             // optional StepState.aux_data: [n_steps][aux_cells][4] cells ...
    const u32* aux_kind;        // ... and kinds (0 none, 1 Word, 2 int, 3 pair, 4 not representable, 5-8 precompile inputs)
    u32 aux_cells;              // cells per step in `aux` (2, or 12 when a precompile state is present)
    u32 rw_dense;     // 1: the RW rows are sorted with consecutive rw_counters (row = rw_counter - rw_base); 0: generic index
    u64 rw_base;
    const u64* rw_keys;  // optional [n_rw][4]: packed (rw, tag, field_tag, id, address) of every row, see rw_pack_row
    ZkCodeDir codes;          // n == 0 = generic index only
    unsigned long long* prof;  // optional phase timestamps (tuning aid): [block][8]
    u32* defer_list;   // device sessions: [n_pairs] pairs the EVM_FAST build could not decide (see EVM_FAST) ...
    u32* defer_count;  // ... and their count (= &dyn->n_deferred)
    const u32* perm;  // optional: lane t evaluates pair perm[t] (state-sorted order); nullptr = identity
    u32 n_pairs;      // n_steps - 1
    u32 opts;         // bit0 begin_with_first_step, bit1 end_with_last_step
    u32* defer_count_twin;  // optional (resident device sessions): the NEXT pass's deferred-pair counter — the hot launch clears it, and the twin of its
                            // tally, for the pass after it (kernels.hpp tally_clear_twin), so that no reset kernel sits in front of every pass
};

ZK_HD void evm_args_resolve(EvmArgs& a) {
#ifdef EVM_RESOLVE_OFF
    return;
#endif
    if (a.dyn) {
        const EvmDyn d = *a.dyn;  // uniform address: scalar loads
        a.rw_dense = (a.rw.n != 0u && d.rw_sparse == 0u) ? 1u : 0u;
        a.rw_base = d.rw_base;
        a.codes.n = d.codes_n;
        a.codes.mask = d.codes_mask;
        a.agg_max_txs = d.agg_max_txs; a.agg_total_txs = d.agg_total_txs; a.agg_invalid_txs = d.agg_invalid_txs;
        a.agg_bad_invalid_rows = d.agg_bad_invalid_rows; a.agg_total_wds = d.agg_total_wds;
    }
}

// LDS staging of the step pair (hot kernel), one u32 entry per lane, lane-major (conflict-free ds_read_b32 / ds_write_b32).
// Per step s (0 curr, 1 next) 12 entries at s * 12: the ten cells that are small integers in every well-formed witness
// (state, rw_counter, call_id, is_root, is_create, pc, sp, memory_word_size, reversible_write_counter, log_id) as 32 bits
// each, gas_left as 64; then the two code-hash cells as 128 bits each at 24 + s * 8.  40 entries = 160 B per lane (it was
// 240 B with one u64 per cell): 40 KB per 256-lane block, so three blocks share a CU's 160 KB instead of two.  A pair with
// a cell wider than its entry (malformed witnesses only) is not staged: that lane reads its step rows from HBM.
// Lanes per workgroup of the hot kernel.  A workgroup is dispatched when all of its wavefronts find a slot, and wavefronts
// of one group finish far apart (execution states differ 10x in length): with 4-wavefront groups the chip ran 1600-1800 of
// its 2048 slots in mid-kernel, with 2-wavefront groups it stays full (kernel 83.4 -> 80.6 us at 2^18 steps); 1-wavefront
// groups pay the LDS directory mirror per wavefront and were slower (88.5 us).
#ifndef EVM_HOT_BLOCK
#define EVM_HOT_BLOCK 128
#endif
#define EVM_STAGE_LANES EVM_HOT_BLOCK
// tuning aid: slot of this wavefront in EvmArgs::prof (the first 4096 wavefronts of the grid)
#define EV_PROF_WAVE ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)
#define EV_PROF_ON(a) ((a).prof && (threadIdx.x & 63) == 0 && EV_PROF_WAVE < 4096u)
#define EVM_STAGE_STRIDE (EVM_STAGE_LANES + 1)  // u32 words between a lane's consecutive entries: odd, so that a quad's writes of one pair's entries hit different banks
#define EVM_STAGE_ENTRIES 40
#define EVM_STAGE_GAS 10  // entry pair of gas_left within a step's 12
ZK_HD constexpr int evm_stage_entry(int s, int c) {
    return c == S_CH_LO || c == S_CH_HI ? 24 + s * 8 + (c - S_CH_LO) * 4
           : s * 12 + (c < S_CH_LO ? c : c == S_PC ? 5 : c == S_SP ? 6 : c == S_MWS ? 7 : c == S_REV ? 8 : c == S_LOG ? 9 : EVM_STAGE_GAS);
}
#define EVM_DIR_MAX_SLOTS 128   // directory mirrored in LDS when it has at most this many slots ...
#define EVM_DIR_MAX_ENTRIES 32  // ... and entries (96 B each)
#define EVM_DIR_SLOT_U64 (EVM_DIR_MAX_SLOTS / 2)
#define EVM_DIR_LDS_U64 (EVM_DIR_SLOT_U64 + EVM_DIR_MAX_ENTRIES * 12)
#if defined(ZK_HOSTSIM)
typedef const u64* EVM_LDS_PTR;
typedef const u32* EVM_LDS32_PTR;
#else
typedef const __attribute__((address_space(3))) u64* EVM_LDS_PTR;
typedef const __attribute__((address_space(3))) u32* EVM_LDS32_PTR;
#endif

struct Word {
    Fr lo, hi;
};
struct WordOrValue {
    Word w;
    bool is_word;
};

// EVM_FAST (k_evm_hot.hip): the hot kernel is compiled with the common case only — dense RW index with packed key records,
// step pairs staged from narrow cells, regular bytecode, word cells below 2^128.  Wherever the general code would take a
// fallback (generic open-addressing probe with the query cells in a stack array, cell-by-cell key compare, step rows read from
// HBM, tx / block table lookups) the fast build sets Ins::defer instead; the kernel appends the pair to the session's deferred
// list and the cold launch — the same gadget sources compiled WITHOUT this macro — evaluates it.  Verdicts are identical by
// construction (a deferred pair is evaluated by exactly the code that evaluated it before round 3); what the hot kernel sheds
// is the fallbacks' register pressure (the 14-cell query structs alone are 112 VGPRs) and their stack frames.
#ifndef EVM_FAST
#define EVM_FAST 0
#endif
struct Ins {
    const EvmArgs* a;
    u32 defer;  // EVM_FAST: this pair needs a fallback path -> the general build evaluates it (status / tally untouched here)
    u64 idx;   // pair index: curr = idx, next = idx + 1
    u32 err;   // first failure's status code (0 = none yet)
    u32 seq;   // checkpoint counter
    u64 rw_off;  // rw_counter_offset (copy gadgets add a table value to it)
    u32 pc_off;
    int sp_off;
    Fr rwc, call_id, sp, pc;  // curr step cells every lookup needs (loaded once)
    // bytecode-directory entry of curr.code_hash, probed once per step (every opcode/push-data
    // lookup of the step goes to the same code): state 0 = not probed, 1 = regular entry cached,
    // 2 = hash absent from the table, 3 = use the generic index
    u32 code_state, code_header_row, code_byte_base, code_n_bytes, code_header_ok;
    u64 code_header_value;
    // the 26 cells of (curr, next) staged in LDS by the hot kernel (evm_stage_steps): lane's entry e at stage[e * EVM_STAGE_STRIDE];
    // nullptr = read the step rows from HBM (cold kernel, hostsim, or a lane whose cells exceed the staged widths)
    EVM_LDS32_PTR stage;
    // the hot kernel's LDS mirror of the bytecode directory (slots as u32 pairs in the first EVM_DIR_SLOT_U64 words, then
    // the entries, 12 u64 each); nullptr = probe the directory in HBM
    EVM_LDS_PTR dir_lds;
    // requested ahead of the gadget (evm_prefetch): the packed bytecode record at curr.program_counter.  opcode_lookup takes
    // it instead of issuing its own dependent load; the checks are unchanged.
    u32 pre_op;
    bool pre_op_ok;
    // the opcode's packed (responsible state, validity, constant gas) word, requested when the opcode byte arrives so that the
    // shared transition tail does not start with a dependent table load; valid for opcode byte op_info_byte (0x100 = none)
    u32 op_info, op_info_byte;
};

#if defined(ZK_HOSTSIM)
#define EV_PROF(I, k) do { } while (0)
#else
#define EV_PROF(I, k) do { if (EV_PROF_ON(*(I).a)) (I).a->prof[EV_PROF_WAVE * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#endif

// ---- checkpoints --------------------------------------------------------------------------
ZK_HD void ev_fail(Ins& I, u32 kind) {
    if (I.err == 0u) I.err = ZK_CODE(kind, I.seq);
}
// branch-free (a select on two flags): hundreds of these sit in straight-line gadget code, and as `if (!cond) ev_fail()` each
// became a compare + exec-mask save + branch, whose scalar dependency stalls cost more than the select
ZK_HD void ev_require(Ins& I, bool cond, u32 kind = ZK_ASSERT) {
    I.seq++;
    const u32 take = (cond ? 0u : 1u) & (I.err == 0u ? 1u : 0u);
    I.err = take ? ZK_CODE(kind, I.seq) : I.err;
}
#define EV_TRY(stmt) do { stmt; if (I.err) return; } while (0)
#define EV_TRYV(stmt, ret) do { stmt; if (I.err) return ret; } while (0)

ZK_HD Fr ev_step_cell(const EvmArgs& a, u64 step, int c) { return fr_load(a.steps + (step * STEP_NCELLS + c) * 4); }
ZK_HD Fr ev_staged_cell(const Ins& I, int s, int c) {
    Fr r = fr_zero();
    const int e = evm_stage_entry(s, c);
    const int n = (c == S_CH_LO || c == S_CH_HI) ? 4 : c == S_GAS ? 2 : 1;
#pragma unroll
    for (int w = 0; w < n; w++) r.v[w] = I.stage[(e + w) * EVM_STAGE_STRIDE];
    return r;
}
#if EVM_FAST
ZK_HD Fr ev_curr(const Ins& I, int c) { return ev_staged_cell(I, 0, c); }  // unstaged (wide) pairs are deferred before any gadget runs
ZK_HD Fr ev_next(const Ins& I, int c) { return ev_staged_cell(I, 1, c); }
#else
ZK_HD Fr ev_curr(const Ins& I, int c) { return I.stage ? ev_staged_cell(I, 0, c) : ev_step_cell(*I.a, I.idx, c); }
ZK_HD Fr ev_next(const Ins& I, int c) { return I.stage ? ev_staged_cell(I, 1, c) : ev_step_cell(*I.a, I.idx + 1, c); }
#endif
ZK_HD Fr fr_u(u64 x) { return fr_from_u64(x); }
ZK_HD Word word_of(const Fr& lo, const Fr& hi) {
    Word w;
    w.lo = lo;
    w.hi = hi;
    return w;
}
ZK_HD Word word_zero() { return word_of(fr_zero(), fr_zero()); }
ZK_HD bool word_eq(const Word& a, const Word& b) { return fr_eq(a.lo, b.lo) && fr_eq(a.hi, b.hi); }

ZK_HD void constrain_zero(Ins& I, const Fr& v) { ev_require(I, fr_is_zero(v)); }
ZK_HD void constrain_equal(Ins& I, const Fr& a, const Fr& b) { ev_require(I, fr_eq(a, b)); }
ZK_HD void constrain_equal_word(Ins& I, const Word& a, const Word& b) { ev_require(I, word_eq(a, b)); }
// range_check (instruction.py:529-534): value fits n_bytes, else ConstraintUnsatFailure is raised
ZK_HD void range_check(Ins& I, const Fr& v, int n_bytes) { ev_require(I, fr_byte_len(v) <= n_bytes, ZK_CONSTRAINT); }
// WordOrValue.value() (util/arithmetic.py:186-189)
ZK_HD Fr value_of(Ins& I, const WordOrValue& v) {
    ev_require(I, !v.is_word);
    return v.w.lo;
}
ZK_HD bool word_cells_fit(const Word& w) { return fr_fits128(w.lo) && fr_fits128(w.hi); }
// Word.to_le_bytes / to_64s (util/arithmetic.py:155-168): int.to_bytes(16) raises OverflowError
ZK_HD U256 to_u256(Ins& I, const Word& w) {
    ev_require(I, word_cells_fit(w), ZK_OVERFLOW_ERROR);
    return u256_from_lo_hi(w.lo, w.hi);
}
// Integer value of a word for the witness computations the reference does with Python big
// ints (`Word.int_value`, util/arithmetic.py:127-129), for cells known to fit 128 bits.  Gadgets test
// `words_wide` first: cells >= 2^128 (malformed words) make the reference compute with unbounded
// integers, which is `wide_witness` (bigz.hpp) in the general build; the fast build defers the pair.
ZK_HD U256 int_value(Ins& I, const Word& w) {
#if EVM_FAST
    if (!word_cells_fit(w)) I.defer = 1u;
#else
    if (!word_cells_fit(w) && I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq);  // a call site that did not test words_wide
#endif
    return u256_from_lo_hi(w.lo, w.hi);
}
ZK_HD bool words_wide(Ins& I, const Word& a, const Word& b, const Word& c, const Word& d) {
    const bool wide = !(word_cells_fit(a) && word_cells_fit(b) && word_cells_fit(c) && word_cells_fit(d));
#if EVM_FAST
    if (wide) I.defer = 1u;
    return false;
#else
    return wide;
#endif
}
ZK_HD bool words_wide(Ins& I, const Word& a, const Word& b, const Word& c) { return words_wide(I, a, b, c, c); }
ZK_HD bool words_wide(Ins& I, const Word& a, const Word& b) { return words_wide(I, a, b, b, b); }
ZK_HD bool words_wide(Ins& I, const Word& a) { return words_wide(I, a, a, a, a); }
ZK_HD bool word_is_zero_int(const Word& w) { return fr_is_zero(w.lo) && fr_is_zero(w.hi); }  // int_value() == 0: both cells (they are >= 0)
ZK_HD Word word_from_u256(const U256& v) { return word_of(u256_lo(v), u256_hi(v)); }
// Word(int) (util/arithmetic.py:115-122): `neg` -> OverflowError, `too_big` (>= 2^256) -> AssertionError
ZK_HD Word word_from_int(Ins& I, const U256& v, bool neg = false, bool too_big = false) {
    I.seq++;
    if (too_big) ev_fail(I, ZK_ASSERT);
    else if (neg) ev_fail(I, ZK_OVERFLOW_ERROR);
    return word_from_u256(v);
}
#if !EVM_FAST
ZK_HD WideRes wide_words(u32 op, const Word& x0, const Word& x1, const Word& x2, const Word& x3) {
    return wide_witness(op, x0.lo, x0.hi, x1.lo, x1.hi, x2.lo, x2.hi, x3.lo, x3.hi);
}
// Word(int) of a wide_witness output: flag bit 0 = negative (OverflowError), bit 1 = not below 2^256 (AssertionError)
ZK_HD Word word_from_wide(Ins& I, const WideRes& W, int k) { return word_from_int(I, W.o[k], (W.fl[k] & 1u) != 0u, (W.fl[k] & 2u) != 0u); }
#endif
// Word.int_value().to_bytes(32, "little") (instruction.py:1349-1350, precompiles/ecrecover.py:49-52): cells >= 2^128 add into
// each other, and a sum that needs more than 32 bytes raises OverflowError outside every checkpoint
ZK_HD U256 int_bytes32(Ins& I, const Word& w) {
    if (word_cells_fit(w)) return u256_from_lo_hi(w.lo, w.hi);
#if EVM_FAST
    I.defer = 1u;
    return fr_zero();
#else
    const WideRes W = wide_words(WIDE_INT256, w, w, w, w);
    if ((W.fl[0] & 2u) && I.err == 0u) I.err = ZK_CODE(ZK_OVERFLOW_ERROR, I.seq);
    return W.o[0];
#endif
}
// Word((lo, hi)) with check=True (util/arithmetic.py:110-114)
ZK_HD Word word_checked(Ins& I, const Fr& lo, const Fr& hi) {
    ev_require(I, fr_fits128(lo) && fr_fits128(hi));
    return word_of(lo, hi);
}
// Instruction.compare (instruction.py:447-451): both operands must fit n_bytes
ZK_HD void ev_compare(Ins& I, const Fr& lhs, const Fr& rhs, int n_bytes, u32& lt, u32& eq) {
    ev_require(I, fr_byte_len(lhs) <= n_bytes && fr_byte_len(rhs) <= n_bytes);
    lt = fr_lt(lhs, rhs);
    eq = fr_eq(lhs, rhs);
}
// compare_word (instruction.py:453-463): hi first, then lo
ZK_HD void compare_word(Ins& I, const Word& a, const Word& b, u32& lt, u32& eq) {
    u32 hl, he, ll, le;
    ev_compare(I, a.hi, b.hi, 16, hl, he);
    ev_compare(I, a.lo, b.lo, 16, ll, le);
    lt = hl + he * ll;
    eq = he * le;
}
// Instruction.select asserts the condition is boolean (instruction.py:419-423)
ZK_HD bool ev_select(Ins& I, const Fr& cond) {
    ev_require(I, fr_le_u64(cond, 1));
    return fr_eq_u64(cond, 1);
}
ZK_HD bool ev_select_b(Ins& I, u32 cond01) {  // structurally boolean condition: checkpoint only
    I.seq++;
    return cond01 == 1u;
}
ZK_HD u32 is_zero_word(const Word& w) { return fr_is_zero(fr_add(w.lo, w.hi)); }  // instruction.py:489
ZK_HD u32 is_equal_word(const Word& a, const Word& b) {
    return fr_is_zero(fr_add(fr_sub(a.lo, b.lo), fr_sub(a.hi, b.hi)));
}
// x mod m for a small modulus (x is a canonical cell: `.n % m`)
ZK_HD u32 fr_mod_small(const Fr& x, u32 m) {
    u64 rem = 0;
    for (int k = 7; k >= 0; k--) rem = ((rem << 32) | x.v[k]) % m;
    return (u32)rem;
}
ZK_HD Word evm_empty_code_hash() {  // util/hash.py:13 (keccak256(""))
    return word_of(fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull), fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull));
}
// word_to_fq (instruction.py:480-484)
ZK_HD Fr word_to_fq(Ins& I, const Word& w, int n_bytes) {
    U256 v = to_u256(I, w);
    bool hi_zero = true;
    for (int b = n_bytes; b < 32; b++) hi_zero = hi_zero && fr_byte(v, b) == 0;
    ev_require(I, hi_zero, ZK_CONSTRAINT);
    Fr r = fr_zero();
    for (int b = 0; b < n_bytes; b++) r.v[b >> 2] |= fr_byte(v, b) << (8 * (b & 3));
    return r;
}

// ---- lookups ------------------------------------------------------------------------------
ZK_HD u64 rw_key_hash_cell(const Fr& rwc) { return zk_hash_cell(0x51ed27u, rwc); }
ZK_HD u64 rw_key_hash(const ZkTable& t, u32 r) { return rw_key_hash_cell(zk_table_cell(t, r, R_RWC)); }
ZK_HD u64 bc_key_hash_cells(const Fr& lo, const Fr& hi, const Fr& tag, const Fr& index) {
    return zk_hash_cell(zk_hash_cell(zk_hash_cell(zk_hash_cell(0xb7e15u, lo), hi), tag), index);
}
ZK_HD u64 bc_key_hash(const ZkTable& t, u32 r) {
    return bc_key_hash_cells(zk_table_cell(t, r, 0), zk_table_cell(t, r, 1), zk_table_cell(t, r, 2), zk_table_cell(t, r, 3));
}
ZK_HD u64 tx_key_hash_cells(const Fr& a, const Fr& b, const Fr& c) {
    return zk_hash_cell(zk_hash_cell(zk_hash_cell(0x7f4a7u, a), b), c);
}
ZK_HD u64 tx_key_hash(const ZkTable& t, u32 r) {
    return tx_key_hash_cells(zk_table_cell(t, r, 0), zk_table_cell(t, r, 1), zk_table_cell(t, r, 2));
}
ZK_HD u64 blk_key_hash_cells(const Fr& a, const Fr& b) { return zk_hash_cell(zk_hash_cell(0x3c6efu, a), b); }
ZK_HD u64 blk_key_hash(const ZkTable& t, u32 r) { return blk_key_hash_cells(zk_table_cell(t, r, 0), zk_table_cell(t, r, 1)); }

ZK_HD u64 copy_key_hash_cells(const Fr& rwc, const Fr& src_addr) { return zk_hash_cell(zk_hash_cell(0xc09fu, rwc), src_addr); }
ZK_HD u64 copy_key_hash(const ZkTable& t, u32 r) { return copy_key_hash_cells(zk_table_cell(t, r, CT_RWC), zk_table_cell(t, r, CT_SRC_ADDR)); }
// exp table: keyed on (identifier, is_last, exponent.lo) — every exp lookup gives the exponent, and the rows of one exponentiation share
// (identifier, is_last = 0): keyed on those two alone, a lookup walked the whole trace of its exponentiation, two dependent HBM round
// trips per row (a wavefront of EXP steps took ~400 us)
ZK_HD u64 expt_key_hash_cells(const Fr& id, const Fr& is_last, const Fr& exp_lo) { return zk_hash_cell(zk_hash_cell(zk_hash_cell(0xe4b7u, id), is_last), exp_lo); }
ZK_HD u64 sig_key_hash_cells(const Fr& msg_lo, const Fr& r_lo, const Fr& s_lo) { return zk_hash_cell(zk_hash_cell(zk_hash_cell(0x516u, msg_lo), r_lo), s_lo); }
ZK_HD u64 sig_key_hash(const ZkTable& t, u32 r) { return sig_key_hash_cells(zk_table_cell(t, r, 0), zk_table_cell(t, r, 3), zk_table_cell(t, r, 5)); }
ZK_HD u64 ecc_key_hash_cells(const Fr& op, const Fr& px_lo, const Fr& qx_lo, const Fr& input_rlc) {
    return zk_hash_cell(zk_hash_cell(zk_hash_cell(zk_hash_cell(0xecc0u, op), px_lo), qx_lo), input_rlc);
}
ZK_HD u64 ecc_key_hash(const ZkTable& t, u32 r) {
    return ecc_key_hash_cells(zk_table_cell(t, r, 0), zk_table_cell(t, r, 1), zk_table_cell(t, r, 5), zk_table_cell(t, r, 9));
}
ZK_HD u64 expt_key_hash(const ZkTable& t, u32 r) { return expt_key_hash_cells(zk_table_cell(t, r, XT_ID), zk_table_cell(t, r, XT_IS_LAST), zk_table_cell(t, r, XT_EXP_LO)); }


// Generic "exactly one distinct matching row" lookup (table.py:864-884) over the open-addressing
// index: `q` holds the query cells, bit c of `mask` says cell c is part of the query.  Out of
// line; returns row | (kind << 32) with kind 0 / ZK_LOOKUP_UNSAT / ZK_LOOKUP_AMBIGUOUS.
// (Round 3 measured a form with four slots per round trip and batched row compares (zk_row_matches): the Copy kernel got slower,
// 42 -> 47 us at 2^15 rows and 0.179 -> 0.202 ms at 2^19 — the masked cell-by-cell compare reads fewer bytes — and the warm EVM
// gadgets did not get faster; the batched compare stays where all cells are compared: keccak_contains, the State MPT lookup.)
ZK_NOINLINE u64 table_probe_generic(ZkTable t, u64 h, const Fr* q, u32 mask) {
    u32 found = ZK_EMPTY_SLOT;
    bool ambiguous = false;
    if (t.n != 0) {
        u32 slot = (u32)h & t.mask;
        for (u32 probes = 0; probes <= t.mask; probes++) {
            const u32 r = t.slots[slot];
            if (r == ZK_EMPTY_SLOT) break;
            bool m = true;
            for (u32 c = 0; c < t.ncells; c++)
                if ((mask >> c) & 1u) m = m && fr_eq(zk_table_cell(t, r, c), q[c]);
            if (m) {
                if (found == ZK_EMPTY_SLOT) found = r;
                else if (!rows_identical(t, found, r)) ambiguous = true;
            }
            slot = (slot + 1) & t.mask;
        }
    }
    if (found == ZK_EMPTY_SLOT) return (u64)ZK_LOOKUP_UNSAT << 32;
    return (u64)found | (ambiguous ? ((u64)ZK_LOOKUP_AMBIGUOUS << 32) : 0ull);
}
template <int NCELLS>
ZK_HD u32 table_lookup(Ins& I, const ZkTable& t, u64 h, const Fr (&q)[NCELLS], u32 mask) {
#if EVM_FAST
    I.defer = 2u;  // the generic probe lives in the general build
    I.seq++;
    return 0u;
#endif
    I.seq++;
    Fr tmp[NCELLS];  // private copy: only this cold path's array escapes to the out-of-line probe
    for (int c = 0; c < NCELLS; c++) tmp[c] = ((mask >> c) & 1u) ? q[c] : fr_zero();
    const u64 res = table_probe_generic(t, h, tmp, mask);
    const u32 kind = (u32)(res >> 32);
    if (kind) ev_fail(I, kind);
    return kind == ZK_LOOKUP_UNSAT ? 0u : (u32)res;
}

// The same lookup for the tables the warm gadgets query on every step (copy / keccak / exp tables): cell count and query mask
// are compile-time constants, so the query stays in registers (no private copy handed to an out-of-line probe: that copy was the
// warm kernel's scratch) and the candidate row is compared in 4-cell groups — the loads of a group in flight together,
// folded into one difference word — instead of one dependent round trip per cell.  Same verdicts as table_probe_generic.
template <int NCELLS, u32 MASK>
ZK_HD u32 table_lookup_inline(Ins& I, const ZkTable& t, u64 h, const Fr (&q)[NCELLS], Fr* out0 = nullptr, int out0_cell = 0, Fr* out1 = nullptr,
                              int out1_cell = 0) {
#if EVM_FAST
    // the pair goes to the general build: the outputs are defined (zero) all the same, never left uninitialised for a caller
    // that would one day branch or address on them
    if (out0) *out0 = fr_zero();
    if (out1) *out1 = fr_zero();
    I.defer = 2u;
    I.seq++;
    return 0u;
#endif
#ifdef ZK_WARM_GENERIC_LOOKUP  // tuning build: the out-of-line generic probe, for A / B timelines
    {
        const u32 rg = table_lookup<NCELLS>(I, t, h, q, MASK);
        if (out0) *out0 = zk_table_cell(t, rg, out0_cell);
        if (out1) *out1 = zk_table_cell(t, rg, out1_cell);
        return rg;
    }
#endif
    I.seq++;
    u32 kind;
    const u32 found = table_probe_inline<NCELLS, MASK>(t, h, q, kind, out0, out0_cell, out1, out1_cell);
    if (kind == (u32)ZK_LOOKUP_UNSAT) { ev_fail(I, ZK_LOOKUP_UNSAT); return 0u; }
    if (kind) ev_fail(I, ZK_LOOKUP_AMBIGUOUS);
    return found;
}

struct RwQ {
    Fr q[RW_NCELLS];
    u32 mask;
};
ZK_HD void rwq_init(RwQ& Q, u32 rw, u32 tag) {
    Q.mask = (1u << R_RWC) | (1u << R_RW) | (1u << R_TAG);
    Q.q[R_RW] = fr_u(rw);
    Q.q[R_TAG] = fr_u(tag);
}
ZK_HD void rwq_set(RwQ& Q, int c, const Fr& v) {
    Q.q[c] = v;
    Q.mask |= 1u << c;
}
ZK_HD void rwq_set_word(RwQ& Q, int c, const Word& w) {
    rwq_set(Q, c, w.lo);
    rwq_set(Q, c + 1, w.hi);
}
// Packed key record of an RW row: the five cells nearly every lookup compares (rw, tag, id, address,
// field_tag: 160 B on the wire) in 32 B, so a lookup reads 2 x 16 B instead of 10 x 16 B.
//   w0: bit 63 = "fits" | bits 0-7 rw | 8-15 tag | 16-23 field_tag | 24-55 address[128..160) | 56-57 the row's
//       type bits (value.is_word, value_prev.is_word)
//   w1: id | w2: address[0..64) | w3: address[64..128)
// A row whose cells exceed those widths (only malformed witnesses) has fits = 0 and is compared cell
// by cell.  Built once per session for the dense index (rw_pack_kernel), like the other indices.
struct RwKey {
    u64 w[4];
};
ZK_HD bool rw_key_fields_fit(const Fr& rw, const Fr& tag, const Fr& id, const Fr& addr, const Fr& ft) {
    return fr_le_u64(rw, 255) && fr_le_u64(tag, 255) && fr_le_u64(ft, 255) && fr_fits64(id) && (addr.v[5] | addr.v[6] | addr.v[7]) == 0u;
}
ZK_HD RwKey rw_pack_fields(const Fr& rw, const Fr& tag, const Fr& id, const Fr& addr, const Fr& ft) {
    RwKey k;
    k.w[0] = (1ull << 63) | (u64)(rw.v[0] & 0xffu) | ((u64)(tag.v[0] & 0xffu) << 8) | ((u64)(ft.v[0] & 0xffu) << 16) | ((u64)addr.v[4] << 24);
    k.w[1] = fr_lo64(id);
    k.w[2] = fr_lo64(addr);
    k.w[3] = fr_hi64of128(addr);
    return k;
}
ZK_HD RwKey rw_pack_row(const ZkTable& t, u32 r) {
    const Fr rw = zk_table_cell(t, r, R_RW), tag = zk_table_cell(t, r, R_TAG), id = zk_table_cell(t, r, R_ID);
    const Fr addr = zk_table_cell(t, r, R_ADDR), ft = zk_table_cell(t, r, R_FT);
    if (!rw_key_fields_fit(rw, tag, id, addr, ft)) {
        RwKey k;
        k.w[0] = k.w[1] = k.w[2] = k.w[3] = 0;
        return k;
    }
    RwKey k = rw_pack_fields(rw, tag, id, addr, ft);
    k.w[0] |= (u64)((t.flags ? t.flags[r] : 3u) & 3u) << 56;
    return k;
}
// row of the dense RW table a lookup at `rwc` addresses (row 0 when out of range: tables keep one zero row when empty)
ZK_HD u32 rw_dense_row(const EvmArgs& a, const Fr& rwc, bool& ok) {
    const u64 off = fr_lo64(rwc) - a.rw_base;
    ok = fr_fits64(rwc) && fr_lo64(rwc) >= a.rw_base && off < (u64)a.rw.n;
    return ok ? (u32)off : 0u;
}
// An RW row requested ahead of its lookup (rw_rows_fetch): the packed key record and the two value cells a stack lookup
// returns.  The lookups of a gadget's opening run sit at consecutive rw_counters, so their rows are known before the first
// one is checked; fetched one lookup at a time each costs a dependent HBM round trip (~2.5 us under load, and a SIMD holds
// only two wavefronts to hide it).
struct RwRow {
    uint4 k01, k23;
    Fr v_lo, v_hi;
    u32 r;  // the dense-table row these were fetched from: a lookup only uses them when it addresses that very row
};
template <int N>
struct RwRows {
    bool valid;  // dense RW table with packed key records; otherwise the lookups fetch for themselves
    RwRow row[N];
};
// Instruction.rw_lookup (instruction.py:792-824); rw_counter = curr.rw_counter + offset unless given.  `pre` = the row's
// key record, fetched ahead (it must be the row this lookup addresses: rw_rows_fetch at the same running offset).
ZK_HD u32 rw_lookup(Ins& I, RwQ& Q, const Fr* rw_counter = nullptr, const RwRow* pre = nullptr) {
    if (rw_counter) {
        Q.q[R_RWC] = *rw_counter;
    } else {
        Q.q[R_RWC] = fr_add_u64(I.rwc, I.rw_off);
        I.rw_off++;
    }
    if (I.a->rw_dense) {
        // direct index: the only row with this rw_counter is row (rw_counter - base)
        I.seq++;
        bool ok;
        const u32 r = rw_dense_row(*I.a, Q.q[R_RWC], ok);
        // branch-free compare: the row loads do not depend on earlier lookups' outcomes, so the
        // loads of consecutive lookups (MLOAD: 32 rows, PUSH32: 33 rows) overlap in flight
        bool key_cells = true;  // compare cells 1..5 one by one (no packed record, or the row does not fit it)
#if EVM_FAST
        if (!I.a->rw_keys) I.defer = 3u;
#endif
        if (I.a->rw_keys) {
            uint4 k01, k23;
            if (pre) {
                // the row requested ahead must be the row this lookup addresses: a gadget whose batch offsets drifted from its lookup
                // order gets no verdict (never a silently compared wrong row)
                if (pre->r != r && I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq);
                k01 = pre->k01;
                k23 = pre->k23;
            } else {
                const uint4* kp = reinterpret_cast<const uint4*>(I.a->rw_keys + (u64)r * 4);
                k01 = kp[0];
                k23 = kp[1];
            }
            const u64 w0 = (u64)k01.x | ((u64)k01.y << 32), w1 = (u64)k01.z | ((u64)k01.w << 32);
            const u64 w2 = (u64)k23.x | ((u64)k23.y << 32), w3 = (u64)k23.z | ((u64)k23.w << 32);
            if (w0 >> 63) {  // the row's key cells fit the packed widths: a queried cell matches iff it fits and is equal
                const u32 m = Q.mask;
                if ((m >> R_RW) & 1u) ok = ok & (fr_le_u64(Q.q[R_RW], 255) & ((u32)(w0 & 0xffu) == Q.q[R_RW].v[0]));
                if ((m >> R_TAG) & 1u) ok = ok & (fr_le_u64(Q.q[R_TAG], 255) & ((u32)((w0 >> 8) & 0xffu) == Q.q[R_TAG].v[0]));
                if ((m >> R_FT) & 1u) ok = ok & (fr_le_u64(Q.q[R_FT], 255) & ((u32)((w0 >> 16) & 0xffu) == Q.q[R_FT].v[0]));
                if ((m >> R_ID) & 1u) ok = ok & (fr_fits64(Q.q[R_ID]) & (w1 == fr_lo64(Q.q[R_ID])));
                if ((m >> R_ADDR) & 1u) {
                    const Fr& qa = Q.q[R_ADDR];
                    ok = ok & (((qa.v[5] | qa.v[6] | qa.v[7]) == 0u) & (w2 == fr_lo64(qa)) & (w3 == fr_hi64of128(qa)) &
                               ((u32)((w0 >> 24) & 0xffffffffull) == qa.v[4]));
                }
                key_cells = false;
            }
        }
#if EVM_FAST
        if (key_cells) I.defer = 4u;  // a row whose key cells exceed the packed widths: cell-by-cell compare in the general build
#else
        if (key_cells) {
#pragma unroll
            for (int c = 1; c <= R_FT; c++)
                if ((Q.mask >> c) & 1u) ok = ok & fr_eq(zk_table_cell(I.a->rw, r, c), Q.q[c]);
        }
#endif
#pragma unroll
        for (int c = R_FT + 1; c < RW_NCELLS; c++)
            if ((Q.mask >> c) & 1u) ok = ok & fr_eq(zk_table_cell(I.a->rw, r, c), Q.q[c]);
        if (!ok) ev_fail(I, ZK_LOOKUP_UNSAT);
        return r;
    }
    return table_lookup<RW_NCELLS>(I, I.a->rw, rw_key_hash_cell(Q.q[R_RWC]), Q.q, Q.mask);
}
// Request the rows of the next N lookups at the running counter (no checkpoint, no state change).
template <int N>
ZK_HD void rw_rows_fetch_at(const Ins& I, RwRows<N>& R, u64 off) {  // rows of the lookups at offsets off .. off + N - 1
    const EvmArgs& a = *I.a;
    R.valid = a.rw_dense && a.rw_keys != nullptr;
    if (!R.valid) return;
#pragma unroll
    for (int k = 0; k < N; k++) {
        bool ok;
        const u32 r = rw_dense_row(a, fr_add_u64(I.rwc, off + (u64)k), ok);
        const uint4* kp = reinterpret_cast<const uint4*>(a.rw_keys + (u64)r * 4);
        R.row[k].k01 = kp[0];
        R.row[k].k23 = kp[1];
        R.row[k].r = r;
        R.row[k].v_lo = zk_table_cell(a.rw, r, R_VAL_LO);
        R.row[k].v_hi = zk_table_cell(a.rw, r, R_VAL_LO + 1);
    }
}
template <int N>
ZK_HD void rw_rows_fetch(const Ins& I, RwRows<N>& R) { rw_rows_fetch_at(I, R, I.rw_off); }
ZK_HD Fr rw_cell(const Ins& I, u32 row, int c) { return zk_table_cell(I.a->rw, row, c); }
ZK_HD Word rw_word(const Ins& I, u32 row, int c) { return word_of(rw_cell(I, row, c), rw_cell(I, row, c + 1)); }
ZK_HD WordOrValue rw_value(const Ins& I, u32 row) {
    WordOrValue v;
    v.w = rw_word(I, row, R_VAL_LO);
    v.is_word = I.a->rw.flags ? (I.a->rw.flags[row] & 1u) : true;
    return v;
}
ZK_HD WordOrValue rw_value_prev(const Ins& I, u32 row) {
    WordOrValue v;
    v.w = rw_word(I, row, R_PREV_LO);
    v.is_word = I.a->rw.flags ? (I.a->rw.flags[row] & 2u) : true;
    return v;
}

// Directory entry of a code hash, resolved once per step: I.code_state = 1 regular code (rows
// addressable directly), 2 no row carries the hash, 3 irregular -> generic index.
ZK_HD void code_dir_resolve(Ins& I, const Word& code_hash) {
    if (I.code_state != 0) return;
    const ZkCodeDir& dir = I.a->codes;
    I.code_state = 3;
#if !defined(ZK_HOSTSIM)
    static_assert(sizeof(ZkCodeEntry) == 96, "LDS mirror layout: 12 u64 per directory entry");
    if (dir.n != 0 && I.dir_lds && fr_fits128(code_hash.lo) && fr_fits128(code_hash.hi)) {  // same probe sequence over the LDS mirror
        u32 slot = (u32)zk_code_hash_key(code_hash.lo, code_hash.hi) & dir.mask;
        I.code_state = 2;
        const u64 h0 = fr_lo64(code_hash.lo), h1 = fr_hi64of128(code_hash.lo), h4 = fr_lo64(code_hash.hi), h5 = fr_hi64of128(code_hash.hi);
        for (u32 probes = 0; probes <= dir.mask; probes++) {
            const u64 pair = I.dir_lds[slot >> 1];
            const u32 k = (slot & 1u) ? (u32)(pair >> 32) : (u32)pair;
            if (k == ZK_EMPTY_SLOT) break;
            EVM_LDS_PTR c = I.dir_lds + EVM_DIR_SLOT_U64 + k * 12;
            if (c[0] == h0 && c[1] == h1 && (c[2] | c[3]) == 0ull && c[4] == h4 && c[5] == h5 && (c[6] | c[7]) == 0ull) {
                const u64 w8 = c[8], w9 = c[9], w11 = c[11];
                I.code_state = (u32)(w9 >> 32) ? 1 : 3;  // regular
                I.code_header_row = (u32)w8;
                I.code_byte_base = (u32)(w8 >> 32);
                I.code_n_bytes = (u32)w9;
                I.code_header_value = c[10];
                I.code_header_ok = (u32)w11;
                break;
            }
            slot = (slot + 1) & dir.mask;
        }
        return;
    }
#endif
    if (dir.n != 0) {
        u32 slot = (u32)zk_code_hash_key(code_hash.lo, code_hash.hi) & dir.mask;
        I.code_state = 2;
        for (u32 probes = 0; probes <= dir.mask; probes++) {
            const u32 k = dir.slots[slot];
            if (k == ZK_EMPTY_SLOT) break;
            const ZkCodeEntry* c = dir.entries + k;
            if (fr_eq(fr_load(c->hash), code_hash.lo) && fr_eq(fr_load(c->hash + 4), code_hash.hi)) {
                I.code_state = c->regular ? 1 : 3;
                I.code_header_row = c->header_row;
                I.code_byte_base = c->byte_base;
                I.code_n_bytes = c->n_bytes;
                I.code_header_value = c->header_value;
                I.code_header_ok = c->header_ok;
                break;
            }
            slot = (slot + 1) & dir.mask;
        }
    }
}
// Tables.bytecode_lookup (table.py:718-731); is_code < 0 = not part of the query
ZK_HD u32 bytecode_lookup(Ins& I, const Word& code_hash, u32 tag, const Fr& index, int is_code, bool foreign = false) {
    Fr q[BYTECODE_NCELLS];
    q[B_HASH_LO] = code_hash.lo;
    q[B_HASH_HI] = code_hash.hi;
    q[B_TAG] = fr_u(tag);
    q[B_INDEX] = index;
    q[B_IS_CODE] = fr_u(is_code > 0 ? 1 : 0);
    q[B_VALUE] = fr_zero();
    u32 mask = 0xfu | (is_code >= 0 ? (1u << B_IS_CODE) : 0u);
    // a hash other than curr.code_hash (EXTCODESIZE, ...) goes through the generic index
    if (foreign) return table_lookup<BYTECODE_NCELLS>(I, I.a->bytecode, bc_key_hash_cells(q[0], q[1], q[2], q[3]), q, mask);
    // all other bytecode lookups of a step query curr.code_hash: probe the directory once per step
    code_dir_resolve(I, code_hash);
    if (I.code_state == 2) {  // no row carries this hash
        I.seq++;
        ev_fail(I, ZK_LOOKUP_UNSAT);
        return 0;
    }
    if (I.code_state == 1) {
        I.seq++;
        bool ok;
        u32 r = 0;
        if (tag == 1) { ok = fr_is_zero(index); r = I.code_header_row; }
        else { ok = tag == 2 && fr_fits64(index) && fr_lo64(index) < (u64)I.code_n_bytes; r = ok ? I.code_byte_base + (u32)fr_lo64(index) : 0u; }
        if (is_code >= 0) ok = ok & fr_eq(zk_table_cell(I.a->bytecode, r, B_IS_CODE), q[B_IS_CODE]);
        if (!ok) ev_fail(I, ZK_LOOKUP_UNSAT);
        return r;
    }
    return table_lookup<BYTECODE_NCELLS>(I, I.a->bytecode, bc_key_hash_cells(q[0], q[1], q[2], q[3]), q, mask);
}
ZK_HD Word curr_code_hash(const Ins& I) { return word_of(ev_curr(I, S_CH_LO), ev_curr(I, S_CH_HI)); }
// bytecode_lookup(...).value for curr.code_hash.  Regular codes (directory hit) are served from the
// packed per-row record / the directory entry: 2 bytes instead of two 32-byte cells per lookup.
ZK_HD Fr bytecode_value(Ins& I, u32 tag, const Fr& index, int is_code) {
    code_dir_resolve(I, curr_code_hash(I));
    if (I.code_state == 1) {
        const uint16_t* packed = I.a->codes.packed;
        if (tag == 1) {
            if (I.code_header_ok && is_code == 0) {
                I.seq++;
                if (!fr_is_zero(index)) ev_fail(I, ZK_LOOKUP_UNSAT);
                return fr_u(I.code_header_value);
            }
        } else if (packed && tag == 2 && fr_fits64(index) && fr_lo64(index) < (u64)I.code_n_bytes) {
            const u32 p = packed[I.code_byte_base + (u32)fr_lo64(index)];
            if (p >> 15) {
                I.seq++;
                if (is_code >= 0 && ((p >> 8) & 1u) != (u32)(is_code > 0 ? 1 : 0)) ev_fail(I, ZK_LOOKUP_UNSAT);
                return fr_u(p & 0xffu);
            }
        }
    }
    u32 r = bytecode_lookup(I, curr_code_hash(I), tag, index, is_code);
    return zk_table_cell(I.a->bytecode, r, B_VALUE);
}
ZK_HD u32 zk_opinfo(u32 opcode_byte) {  // responsible state | valid << 8 | constant gas << 16 of an opcode (evm_tables.h)
    static const uint32_t opinfo[256] = ZK_OPINFO_INIT;
    return opinfo[opcode_byte & 0xffu];
}
ZK_HD Fr opcode_lookup_at(Ins& I, const Fr& index, bool is_code) {  // instruction.py:789-790
    return bytecode_value(I, 2, index, is_code ? 1 : 0);
}
ZK_HD Fr opcode_lookup(Ins& I, bool is_code) {  // instruction.py:784-787
    if (I.pc_off == 0u && I.pre_op_ok && (I.pre_op >> 15)) {  // the record at curr.program_counter, requested ahead: bytecode_value's packed form
        I.pc_off++;
        I.seq++;
        if (((I.pre_op >> 8) & 1u) != (is_code ? 1u : 0u)) ev_fail(I, ZK_LOOKUP_UNSAT);
        I.op_info_byte = I.pre_op & 0xffu;
        I.op_info = zk_opinfo(I.op_info_byte);
        return fr_u(I.pre_op & 0xffu);
    }
    Fr index = fr_add_u64(I.pc, I.pc_off);
    I.pc_off++;
    return opcode_lookup_at(I, index, is_code);
}
ZK_HD Fr bytecode_length(Ins& I, const Word& code_hash, bool foreign = false) {  // instruction.py:771-774
    if (!foreign) return bytecode_value(I, 1, fr_zero(), 0);  // every non-foreign caller passes curr.code_hash
    u32 r = bytecode_lookup(I, code_hash, 1, fr_zero(), 0, foreign);
    return zk_table_cell(I.a->bytecode, r, B_VALUE);
}
ZK_HD WordOrValue tx_lookup(Ins& I, const Fr& tx_id, u32 field_tag, const Fr* index = nullptr) {  // table.py:697-706
    Fr q[TX_NCELLS];
    q[0] = tx_id;
    q[1] = fr_u(field_tag);
    q[2] = index ? *index : fr_zero();
    q[3] = fr_zero();
    q[4] = fr_zero();
    u32 r = table_lookup<TX_NCELLS>(I, I.a->tx, tx_key_hash_cells(q[0], q[1], q[2]), q, 0x7u);
    WordOrValue v;
    v.w = word_of(zk_table_cell(I.a->tx, r, 3), zk_table_cell(I.a->tx, r, 4));
    v.is_word = I.a->tx.flags ? (I.a->tx.flags[r] & 1u) : true;
    return v;
}
ZK_HD WordOrValue block_lookup(Ins& I, u32 field_tag, const Fr* number = nullptr) {  // table.py:690-695
    Fr q[BLOCK_NCELLS];
    q[0] = fr_u(field_tag);
    q[1] = number ? *number : fr_zero();
    q[2] = fr_zero();
    q[3] = fr_zero();
    u32 r = table_lookup<BLOCK_NCELLS>(I, I.a->block, blk_key_hash_cells(q[0], q[1]), q, 0x3u);
    WordOrValue v;
    v.w = word_of(zk_table_cell(I.a->block, r, 2), zk_table_cell(I.a->block, r, 3));
    v.is_word = I.a->block.flags ? (I.a->block.flags[r] & 1u) : true;
    return v;
}

// Tables.copy_lookup (table.py:760-787) for non-TxLog destinations: nine query cells
struct CopyRes {
    Fr rwc_inc, rlc_acc;
};
ZK_HD CopyRes copy_lookup(Ins& I, const Word& src_id, u32 src_tag, const Word& dst_id, u32 dst_tag, const Fr& src_addr,
                          const Fr& src_addr_end, const Fr& dst_addr, const Fr& length, const Fr& rw_counter) {
    Fr q[COPY_T_NCELLS];
    q[CT_IS_FIRST] = fr_zero();
    q[CT_SRC_ID_LO] = src_id.lo; q[CT_SRC_ID_HI] = src_id.hi; q[CT_SRC_TAG] = fr_u(src_tag);
    q[CT_DST_ID_LO] = dst_id.lo; q[CT_DST_ID_HI] = dst_id.hi; q[CT_DST_TAG] = fr_u(dst_tag);
    q[CT_SRC_ADDR] = src_addr; q[CT_SRC_ADDR_END] = src_addr_end; q[CT_DST_ADDR] = dst_addr;
    q[CT_LENGTH] = length; q[CT_RLC_ACC] = fr_zero(); q[CT_RWC] = rw_counter; q[CT_RWC_INC] = fr_zero();
    constexpr u32 mask = ((1u << COPY_T_NCELLS) - 1u) & ~((1u << CT_IS_FIRST) | (1u << CT_RLC_ACC) | (1u << CT_RWC_INC));
    CopyRes R;
    table_lookup_inline<COPY_T_NCELLS, mask>(I, I.a->copy, copy_key_hash_cells(rw_counter, src_addr), q, &R.rwc_inc, CT_RWC_INC, &R.rlc_acc, CT_RLC_ACC);
    return R;
}
ZK_HD Word word_value(const Fr& v) { return word_of(v, fr_zero()); }  // WordOrValue(FQ): hi cell is 0
// Tables.keccak_lookup (table.py:789-795): state_tag = 2 (Finalize), input_len, input_rlc -> output
ZK_HD Word keccak_lookup(Ins& I, const Fr& length, const Fr& value_rlc) {
    Fr q[KECCAK_NCELLS];
    q[0] = fr_u(2); q[1] = value_rlc; q[2] = length; q[3] = fr_zero(); q[4] = fr_zero();
    Word out;
    table_lookup_inline<KECCAK_NCELLS, 0x7u>(I, I.a->keccak, keccak_key_hash_cells(value_rlc, length), q, &out.lo, 3, &out.hi, 4);
    return out;
}
// Tables.exp_lookup (table.py:797-814): is_step = 1, identifier, is_last, base limbs, exponent -> exponentiation
ZK_HD Word exp_lookup(Ins& I, const Fr& identifier, const Fr& is_last, const u64 base_limbs[4], const Word& exponent) {
    Fr q[EXP_T_NCELLS];
    q[XT_IS_STEP] = fr_u(1); q[XT_ID] = identifier; q[XT_IS_LAST] = is_last;
    for (int k = 0; k < 4; k++) q[XT_BASE0 + k] = fr_u(base_limbs[k]);
    q[XT_EXP_LO] = exponent.lo; q[XT_EXP_HI] = exponent.hi; q[XT_RES_LO] = fr_zero(); q[XT_RES_HI] = fr_zero();
    Word out;
    table_lookup_inline<EXP_T_NCELLS, 0x1ffu>(I, I.a->exp, expt_key_hash_cells(identifier, is_last, exponent.lo), q, &out.lo, XT_RES_LO, &out.hi, XT_RES_HI);
    return out;
}

// Tables.sig_lookup (table.py:816-833) and ecc_lookup (:835-858): the query names every field of the row
ZK_HD void sig_lookup(Ins& I, const Word& msg_hash, const Fr& sig_v, const Word& sig_r, const Word& sig_s, const Fr& recovered_addr,
                      const Fr& is_valid) {
    Fr q[SIG_T_NCELLS];
    q[0] = msg_hash.lo; q[1] = msg_hash.hi; q[2] = sig_v; q[3] = sig_r.lo; q[4] = sig_r.hi; q[5] = sig_s.lo; q[6] = sig_s.hi;
    q[7] = recovered_addr; q[8] = is_valid;
    table_lookup<SIG_T_NCELLS>(I, I.a->sig, sig_key_hash_cells(msg_hash.lo, sig_r.lo, sig_s.lo), q, (1u << SIG_T_NCELLS) - 1u);
}
ZK_HD void ecc_lookup(Ins& I, u32 op_type, const Word& px, const Word& py, const Word& qx, const Word& qy, const Fr& input_rlc,
                      const Fr& outx, const Fr& outy, const Fr& is_valid) {
    Fr q[ECC_T_NCELLS];
    q[0] = fr_u(op_type); q[1] = px.lo; q[2] = px.hi; q[3] = py.lo; q[4] = py.hi; q[5] = qx.lo; q[6] = qx.hi; q[7] = qy.lo; q[8] = qy.hi;
    q[9] = input_rlc; q[10] = outx; q[11] = outy; q[12] = is_valid;
    table_lookup<ECC_T_NCELLS>(I, I.a->ecc, ecc_key_hash_cells(q[0], px.lo, qx.lo, input_rlc), q, (1u << ECC_T_NCELLS) - 1u);
}

// Fixed-table membership in closed form (table.py:37-103, 673-688): exact 4-tuple semantics,
// operands are field elements (anything >= 256 / >= range is simply not in the table).
ZK_HD void fixed_lookup(Ins& I, u32 tag, const Fr& v0, const Fr& v1, const Fr& v2) {
    I.seq++;
    bool ok = false;
    const bool b0 = fr_le_u64(v0, 255), b1 = fr_le_u64(v1, 255);
    const u32 x = v0.v[0], y = v1.v[0];
    switch (tag) {
    case FX_Range5: ok = fr_le_u64(v0, 4) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range16: ok = fr_le_u64(v0, 15) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range32: ok = fr_le_u64(v0, 31) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range64: ok = fr_le_u64(v0, 63) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range256: ok = b0 && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range512: ok = fr_le_u64(v0, 511) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range1024: ok = fr_le_u64(v0, 1023) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_Range24_576: ok = fr_le_u64(v0, 24575) && fr_is_zero(v1) && fr_is_zero(v2); break;
    case FX_SignByte: ok = b0 && fr_eq_u64(v1, (x >> 7) * 0xffu) && fr_is_zero(v2); break;
    case FX_BitwiseAnd: ok = b0 && b1 && fr_eq_u64(v2, x & y); break;
    case FX_BitwiseOr: ok = b0 && b1 && fr_eq_u64(v2, x | y); break;
    case FX_BitwiseXor: ok = b0 && b1 && fr_eq_u64(v2, x ^ y); break;
    case FX_ResponsibleOpcode: {
        static const uint8_t resp[256] = ZK_OPCODE_RESP_STATE_INIT;
        static const uint8_t valid[256] = ZK_OPCODE_VALID_INIT;
        ok = fr_is_zero(v2) && b1 && fr_le_u64(v0, 255) && x != 0 && resp[y] == x;
        if (fr_eq_u64(v0, ES_ErrorInvalidOpcode)) {  // execution_state.py:355-356
            ok = fr_is_zero(v2) && b1 && !valid[y];
        } else if (fr_eq_u64(v0, ES_ErrorStack) && b1 && valid[y]) {  // opcode.py:369-384: (opcode, stack_pointer)
            static const int16_t mn[256] = ZK_OPCODE_MIN_SP_INIT;
            static const int16_t mx[256] = ZK_OPCODE_MAX_SP_INIT;
            const bool small = fr_le_u64(v2, 1024);
            const int sp = (int)v2.v[0];
            ok = small && (sp < mn[y] || sp >= mx[y] + 1);
        } else if (fr_eq_u64(v0, ES_ErrorWriteProtection)) {  // opcode.py:395-407
            ok = fr_is_zero(v2) && b1 && (y == OP_SSTORE || (y >= OP_LOG0 && y <= OP_LOG4) || y == OP_CREATE || y == OP_CALL ||
                                          y == OP_CREATE2 || y == OP_SELFDESTRUCT);
        }
        break;
    }
    case FX_PrecompileInfo: {  // precompile.py:46-70: (execution state, address, base gas)
        static const uint8_t pstate[10] = {0, ES_ECRECOVER, ES_SHA256, ES_RIPEMD160, ES_DATACOPY, ES_BIGMODEXP, ES_BN254_ADD, ES_BN254_SCALAR_MUL,
                                           ES_BN254_PAIRING, ES_BLAKE2F};
        static const uint16_t pgas[10] = {0, 3000, 60, 600, 15, 0, 150, 6000, 45000, 0};
        ok = fr_le_u64(v1, 9) && y >= 1 && fr_eq_u64(v0, pstate[y]) && fr_eq_u64(v2, pgas[y]);
        break;
    }
    case FX_OpcodeConstantGas: {  // opcode.py:387-392
        static const uint8_t valid[256] = ZK_OPCODE_VALID_INIT;
        static const uint8_t dyn[256] = ZK_OPCODE_DYNAMIC_GAS_INIT;
        static const uint16_t cgas[256] = ZK_OPCODE_CONST_GAS_INIT;
        ok = b0 && valid[x] && !dyn[x] && cgas[x] > 0 && fr_eq_u64(v1, cgas[x]) && fr_is_zero(v2);
        break;
    }
    case FX_Pow2: {
        if (b0) {
            Fr p = fr_zero();
            p.v[(x & 127u) >> 5] = 1u << (x & 31u);
            ok = x < 128 ? (fr_eq(v1, p) && fr_is_zero(v2)) : (fr_is_zero(v1) && fr_eq(v2, p));
        }
        break;
    }
    default:
        if (I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq);
        return;
    }
    if (!ok) ev_fail(I, ZK_LOOKUP_UNSAT);
}

// ---- stack / memory / call context (instruction.py:866-935) --------------------------------
ZK_HD Word stack_lookup(Ins& I, u32 rw, int off) {
    RwQ Q;
    rwq_init(Q, rw, TG_Stack);
    rwq_set(Q, R_ID, I.call_id);
    const Fr& sp = I.sp;
    rwq_set(Q, R_ADDR, off >= 0 ? fr_add_u64(sp, (u64)off) : fr_sub_u64(sp, (u64)(-off)));
    u32 r = rw_lookup(I, Q);
    return rw_word(I, r, R_VAL_LO);
}
ZK_HD Word stack_pop(Ins& I) {
    int off = I.sp_off;
    I.sp_off++;
    return stack_lookup(I, 0, off);
}
ZK_HD Word stack_push(Ins& I) {
    I.sp_off--;
    return stack_lookup(I, 1, I.sp_off);
}
// The same lookups on a row requested ahead (rw_rows_fetch at the offset this lookup runs at)
ZK_HD Word stack_lookup_row(Ins& I, u32 rw, int off, const RwRow& row) {
    RwQ Q;
    rwq_init(Q, rw, TG_Stack);
    rwq_set(Q, R_ID, I.call_id);
    const Fr& sp = I.sp;
    rwq_set(Q, R_ADDR, off >= 0 ? fr_add_u64(sp, (u64)off) : fr_sub_u64(sp, (u64)(-off)));
    rw_lookup(I, Q, nullptr, &row);  // (checks that `row` is the row it addresses)
    return word_of(row.v_lo, row.v_hi);
}
template <int N>
ZK_HD Word stack_pop(Ins& I, const RwRows<N>& R, int k) {
    if (!R.valid) return stack_pop(I);
    int off = I.sp_off;
    I.sp_off++;
    return stack_lookup_row(I, 0, off, R.row[k]);
}
template <int N>
ZK_HD Word stack_push(Ins& I, const RwRows<N>& R, int k) {
    if (!R.valid) return stack_push(I);
    I.sp_off--;
    return stack_lookup_row(I, 1, I.sp_off, R.row[k]);
}
ZK_HD Fr memory_lookup(Ins& I, u32 rw, const Fr& addr, const Fr* call_id = nullptr) {
    RwQ Q;
    rwq_init(Q, rw, TG_Memory);
    rwq_set(Q, R_ID, call_id ? *call_id : I.call_id);
    rwq_set(Q, R_ADDR, addr);
    u32 r = rw_lookup(I, Q);
    return value_of(I, rw_value(I, r));
}
// The 32 memory_lookup calls of MLOAD / MSTORE (memory.py:32-40): rows rw_counter .. +31 of the dense RW
// table, keys (rw, Memory, call_id, address + k), each followed by `.value.value()` (type bit must be
// clear).  With the packed key records the 32 records of a lane are contiguous (1 KB): they are fetched
// in batches of eight and compared in registers; failures are reported at the same checkpoints as the
// one-by-one form.  Falls back to the generic lookups whenever a record or the query does not fit.
ZK_HD void memory_lookup_run32(Ins& I, u32 rw, const Fr& address) {
    const EvmArgs& a = *I.a;
    const Fr rwc0 = fr_add_u64(I.rwc, I.rw_off);
    const u64 off = fr_lo64(rwc0) - a.rw_base;
    bool fastpath = a.rw_dense && a.rw_keys && fr_fits64(rwc0) && fr_lo64(rwc0) >= a.rw_base && off + 32 <= (u64)a.rw.n && off + 32 > off &&
                    fr_fits64(I.call_id) && (address.v[5] | address.v[6] | address.v[7]) == 0u &&
                    !(address.v[4] == 0xffffffffu && address.v[3] == 0xffffffffu && address.v[2] == 0xffffffffu && address.v[1] == 0xffffffffu &&
                      address.v[0] >= 0xffffffe0u);  // address + 31 stays below 2^160
    u32 bad_lookup = 0, bad_type = 0;
    if (fastpath) {
        const uint4* kp = reinterpret_cast<const uint4*>(a.rw_keys + off * 4);
        const u64 id = fr_lo64(I.call_id);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint4 k01[8], k23[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { k01[j] = kp[2 * (8 * b + j)]; k23[j] = kp[2 * (8 * b + j) + 1]; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 8 * b + j;
                const Fr ak = fr_add_u64(address, (u64)k);
                const u64 w0 = (u64)k01[j].x | ((u64)k01[j].y << 32), w1 = (u64)k01[j].z | ((u64)k01[j].w << 32);
                const u64 w2 = (u64)k23[j].x | ((u64)k23[j].y << 32), w3 = (u64)k23[j].z | ((u64)k23[j].w << 32);
                // expected: fits | rw | Memory << 8 | address[128..160) << 24 (field_tag is not part of the query)
                const u64 want0 = (1ull << 63) | (u64)rw | ((u64)TG_Memory << 8) | ((u64)ak.v[4] << 24);
                const u64 cmp_mask = ~((0xffull << 16) | (3ull << 56));
                if (!(w0 >> 63)) fastpath = false;  // a record that does not fit: redo everything one by one
                const bool ok = ((w0 ^ want0) & cmp_mask) == 0 && w1 == id && w2 == fr_lo64(ak) && w3 == fr_hi64of128(ak);
                bad_lookup |= (ok ? 0u : 1u) << k;
                bad_type |= (u32)((w0 >> 56) & 1u) << k;
            }
        }
    }
    if (fastpath) {
        for (int k = 0; k < 32; k++) {
            I.seq++;
            if ((bad_lookup >> k) & 1u) ev_fail(I, ZK_LOOKUP_UNSAT);
            I.seq++;
            if ((bad_type >> k) & 1u) ev_fail(I, ZK_ASSERT);
        }
        I.rw_off += 32;
        return;
    }
    for (int k = 0; k < 32; k++) memory_lookup(I, rw, fr_add_u64(address, (u64)k));
}
ZK_HD WordOrValue call_context_lookup_word(Ins& I, u32 field_tag, u32 rw = 0, const Fr* call_id = nullptr) {
    RwQ Q;
    rwq_init(Q, rw, TG_CallContext);
    rwq_set(Q, R_ID, call_id ? *call_id : I.call_id);
    rwq_set(Q, R_ADDR, fr_u(field_tag));
    u32 r = rw_lookup(I, Q);
    return rw_value(I, r);
}
// the same lookup on a row requested ahead (rw_rows_fetch_at at the offset this lookup runs at): the value cells and the
// value's type bit (bit 56 of the packed key record, rw_pack_row) come with the row
ZK_HD WordOrValue call_context_lookup_word_row(Ins& I, u32 field_tag, u32 rw, const Fr* call_id, const RwRow& row) {
    RwQ Q;
    rwq_init(Q, rw, TG_CallContext);
    rwq_set(Q, R_ID, call_id ? *call_id : I.call_id);
    rwq_set(Q, R_ADDR, fr_u(field_tag));
    const u32 r = rw_lookup(I, Q, nullptr, &row);
    WordOrValue v;
    v.w = word_of(row.v_lo, row.v_hi);
    const u64 w0 = (u64)row.k01.x | ((u64)row.k01.y << 32);
    v.is_word = (w0 >> 63) ? (((w0 >> 56) & 1ull) != 0ull) : (I.a->rw.flags ? (I.a->rw.flags[r] & 1u) : true);
    return v;
}
ZK_HD Fr call_context_lookup(Ins& I, u32 field_tag, u32 rw = 0, const Fr* call_id = nullptr) {
    WordOrValue v = call_context_lookup_word(I, field_tag, rw, call_id);
    return value_of(I, v);
}
struct Reversion {
    Fr end, persistent, rwc;
};
ZK_HD Reversion reversion_info(Ins& I, const Fr* call_id = nullptr) {  // instruction.py:901-913
    Reversion rv;
    rv.end = call_context_lookup(I, CC_RwCounterEndOfReversion, 0, call_id);
    rv.persistent = call_context_lookup(I, CC_IsPersistent, 0, call_id);
    rv.rwc = call_id ? fr_zero() : ev_curr(I, S_REV);
    return rv;
}
// call-context reads of the current call on rows requested ahead (R.row[k] must be the row the lookup runs at)
template <int N>
ZK_HD WordOrValue call_context_word(Ins& I, u32 field_tag, const RwRows<N>& R, int k) {
    return R.valid ? call_context_lookup_word_row(I, field_tag, 0, nullptr, R.row[k]) : call_context_lookup_word(I, field_tag);
}
template <int N>
ZK_HD Fr call_context_value(Ins& I, u32 field_tag, const RwRows<N>& R, int k) {
    WordOrValue v = call_context_word(I, field_tag, R, k);
    return value_of(I, v);
}
template <int N>
ZK_HD Reversion reversion_info(Ins& I, const RwRows<N>& R, int k) {  // instruction.py:901-913 on rows k, k + 1
    Reversion rv;
    rv.end = call_context_value(I, CC_RwCounterEndOfReversion, R, k);
    rv.persistent = call_context_value(I, CC_IsPersistent, R, k + 1);
    rv.rwc = ev_curr(I, S_REV);
    return rv;
}
// state_write (instruction.py:826-863): the write plus, when not persistent, its reversion row
ZK_HD u32 state_write(Ins& I, RwQ& Q, Reversion& rv) {
    const u32 tag = Q.q[R_TAG].v[0];
    u32 r = rw_lookup(I, Q);
    if (I.err) return r;
    if (fr_is_zero(rv.persistent)) {
        Fr rwc = fr_sub(rv.end, rv.rwc);
        rv.rwc = fr_add_u64(rv.rwc, 1);
        RwQ R;
        rwq_init(R, 1, tag);
        rwq_set(R, R_ID, rw_cell(I, r, R_ID));
        rwq_set(R, R_ADDR, rw_cell(I, r, R_ADDR));
        rwq_set(R, R_FT, rw_cell(I, r, R_FT));
        rwq_set_word(R, R_KEY_LO, rw_word(I, r, R_KEY_LO));
        rwq_set_word(R, R_VAL_LO, rw_word(I, r, R_PREV_LO));
        rwq_set_word(R, R_PREV_LO, rw_word(I, r, R_VAL_LO));
        rwq_set_word(R, R_AUX_LO, rw_word(I, r, R_AUX_LO));
        rw_lookup(I, R, &rwc);
    }
    return r;
}

// ---- 256-bit word arithmetic ------------------------------------------------------------------
// add_words (util/arithmetic.py:236-242) for two addends
ZK_HD Word add_words2(Ins& I, const Word& x, const Word& y, Fr& carry_hi) {
    Fr slo = fr_add(x.lo, y.lo);  // FQ sum, then divmod by 2^128 on its canonical integer
    Fr sum_lo = slo, c_lo = fr_zero();
    for (int k = 4; k < 8; k++) { c_lo.v[k - 4] = slo.v[k]; sum_lo.v[k] = 0; }
    Fr shi = fr_add(fr_add(x.hi, y.hi), c_lo);
    Fr sum_hi = shi;
    carry_hi = fr_zero();
    for (int k = 4; k < 8; k++) { carry_hi.v[k - 4] = shi.v[k]; sum_hi.v[k] = 0; }
    return word_checked(I, sum_lo, sum_hi);
}

struct Limbs64 {
    u64 v[4];
};
ZK_HD Limbs64 to_64s(Ins& I, const Word& w) {
    U256 x = to_u256(I, w);
    Limbs64 l;
    for (int k = 0; k < 4; k++) l.v[k] = u256_limb64(x, k);
    return l;
}
// 64x64 -> 128 accumulate into a 256-bit integer at a 64-bit limb offset
ZK_HD void acc_mul64(U256& acc, u64 a, u64 b, int limb_off) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0, p01 = (u64)a0 * b1, p10 = (u64)a1 * b0, p11 = (u64)a1 * b1;
    u32 w[4];
    u64 c = p00;
    w[0] = (u32)c;
    c = (c >> 32) + (u32)p01 + (u32)p10;
    w[1] = (u32)c;
    c = (c >> 32) + (p01 >> 32) + (p10 >> 32) + (u32)p11;
    w[2] = (u32)c;
    c = (c >> 32) + (p11 >> 32);
    w[3] = (u32)c;
    u64 carry = 0;
    for (int k = 0; k < 8 - 2 * limb_off; k++) {
        carry += (u64)acc.v[2 * limb_off + k] + (k < 4 ? w[k] : 0u);
        acc.v[2 * limb_off + k] = (u32)carry;
        carry >>= 32;
    }
}
// t-values of mul_add_words* (instruction.py:604-611 / 643-652): plain integers < 2^196
struct MulT {
    U256 lo;    // t0 + t1 * 2^64
    U256 mid;   // t2 + t3 * 2^64
    U256 hi;    // t4 + t5 * 2^64
    U256 t6;
    U256 ovf;   // a1b3 + a2b2 + a3b1 + a2b3 + a3b2 + a3b3 (mul_add_words' overflow terms)
};
ZK_HD MulT mul_terms(const Limbs64& a, const Limbs64& b) {
    MulT t;
    t.lo = fr_zero(); t.mid = fr_zero(); t.hi = fr_zero(); t.t6 = fr_zero(); t.ovf = fr_zero();
    acc_mul64(t.lo, a.v[0], b.v[0], 0);
    acc_mul64(t.lo, a.v[0], b.v[1], 1);
    acc_mul64(t.lo, a.v[1], b.v[0], 1);
    acc_mul64(t.mid, a.v[0], b.v[2], 0);
    acc_mul64(t.mid, a.v[1], b.v[1], 0);
    acc_mul64(t.mid, a.v[2], b.v[0], 0);
    acc_mul64(t.mid, a.v[0], b.v[3], 1);
    acc_mul64(t.mid, a.v[1], b.v[2], 1);
    acc_mul64(t.mid, a.v[2], b.v[1], 1);
    acc_mul64(t.mid, a.v[3], b.v[0], 1);
    acc_mul64(t.hi, a.v[1], b.v[3], 0);
    acc_mul64(t.hi, a.v[2], b.v[2], 0);
    acc_mul64(t.hi, a.v[3], b.v[1], 0);
    acc_mul64(t.hi, a.v[2], b.v[3], 1);
    acc_mul64(t.hi, a.v[3], b.v[2], 1);
    acc_mul64(t.t6, a.v[3], b.v[3], 0);
    acc_mul64(t.ovf, a.v[1], b.v[3], 0);
    acc_mul64(t.ovf, a.v[2], b.v[2], 0);
    acc_mul64(t.ovf, a.v[3], b.v[1], 0);
    acc_mul64(t.ovf, a.v[2], b.v[3], 0);
    acc_mul64(t.ovf, a.v[3], b.v[2], 0);
    acc_mul64(t.ovf, a.v[3], b.v[3], 0);
    return t;
}
// mul_add_words (instruction.py:599-632): constrains a*b + c == d (mod 2^256), returns overflow.
// The two constrain_equal calls (:629-630) are identities of the field (carry is *defined* as
// (lhs - d)/2^128), so they only advance the checkpoint counter.
ZK_HD Fr mul_add_words(Ins& I, const Word& a, const Word& b, const Word& c, const Word& d) {
    Limbs64 a64 = to_64s(I, a);
    Limbs64 b64 = to_64s(I, b);
    MulT t = mul_terms(a64, b64);
    Fr carry_lo = div_2p128_for_range9(fr_add(t.lo, c.lo), d.lo);
    Fr carry_hi = div_2p128_for_range9(fr_add(fr_add(t.mid, c.hi), carry_lo), d.hi);
    Fr overflow = fr_add(carry_hi, t.ovf);
    range_check(I, carry_lo, 9);
    range_check(I, carry_hi, 9);
    I.seq += 2;
    return overflow;
}
// mul_add_words_512 (instruction.py:634-665): a*b + c == d*2^256 + e
ZK_HD void mul_add_words_512(Ins& I, const Word& a, const Word& b, const Word& c, const Word& d, const Word& e) {
    Limbs64 a64 = to_64s(I, a);
    Limbs64 b64 = to_64s(I, b);
    MulT t = mul_terms(a64, b64);
    Fr c0 = div_2p128_for_range9(fr_add(t.lo, c.lo), e.lo);
    Fr c1 = div_2p128_for_range9(fr_add(fr_add(t.mid, c.hi), c0), e.hi);
    Fr c2 = div_2p128_for_range9(fr_add(t.hi, c1), d.lo);
    range_check(I, c0, 9);
    range_check(I, c1, 9);
    range_check(I, c2, 9);
    I.seq += 3;  // three identities (:660-662)
    constrain_equal(I, fr_add(t.t6, c2), d.hi);
}

// ---- step-state transition (instruction.py:206-264, 365-394) ----------------------------------
struct Trans {
    u32 kind;  // 0 same, 1 delta, 2 to
    Fr value;
};
ZK_HD Trans t_same() { Trans t; t.kind = 0; t.value = fr_zero(); return t; }
ZK_HD Trans t_delta(const Fr& v) { Trans t; t.kind = 1; t.value = v; return t; }
ZK_HD Trans t_delta_i(long long v) { return t_delta(v >= 0 ? fr_u((u64)v) : fr_neg(fr_u((u64)(-v)))); }
ZK_HD Trans t_to(const Fr& v) { Trans t; t.kind = 2; t.value = v; return t; }
ZK_HD void transition(Ins& I, int cell, const Trans& t) {
    Fr c = ev_curr(I, cell), n = ev_next(I, cell);
    Fr expect = t.kind == 0 ? c : (t.kind == 1 ? fr_add(c, t.value) : t.value);
    ev_require(I, fr_eq(n, expect));
}
// The step-state transition every same-context gadget ends with (instruction.py:365-394) is
// evaluated ONCE after the gadget dispatch: gadgets only record its parameters here.
struct Tail {
    Fr opcode, pc_val, mws_val, dyn_gas;
    int rw_delta, sp_delta, rev_delta;
    u32 pc_kind, mws_kind;  // Trans kinds: 0 same, 1 delta, 2 to
    u32 rwc_mode;           // 0: rw_counter delta is rw_delta; 1 / 2: the gadget already compared next.rw_counter
                            // with a field-valued delta (copy gadgets: rw_counter_offset + rwc_inc) -> equal / different
    u32 log_mode;           // same scheme for log_id: 0 = same, 1 / 2 = precomputed delta matched / did not
    bool enabled;
    u32 err_tail;  // error states: 1 = constrain_error_state, 2 = out-of-gas compare (cost in dyn_gas) first
};
ZK_HD void set_tail(Tail& T, const Fr& opcode, int rw_delta, const Trans& pc, int sp_delta, const Trans& mws,
                    int rev_delta, const Fr& dyn_gas) {
    T.opcode = opcode;
    T.rw_delta = rw_delta;
    T.pc_kind = pc.kind;
    T.pc_val = pc.value;
    T.sp_delta = sp_delta;
    T.mws_kind = mws.kind;
    T.mws_val = mws.value;
    T.rev_delta = rev_delta;
    T.dyn_gas = dyn_gas;
    T.rwc_mode = 0;
    T.log_mode = 0;
    T.enabled = true;
}
// rw_counter = Transition.delta(field value): compare now, report at the transition's checkpoint
ZK_HD void set_tail_rwc_delta(Ins& I, Tail& T, const Fr& delta) {
    T.rwc_mode = fr_eq(ev_next(I, S_RWC), fr_add(ev_curr(I, S_RWC), delta)) ? 1u : 2u;
}
ZK_HD void set_tail3(Tail& T, const Fr& opcode, int rwc, int pc, int sp) {
    set_tail(T, opcode, rwc, t_delta_i(pc), sp, t_same(), 0, fr_zero());
}
ZK_HD Trans t_int(int d) { return d == 0 ? t_same() : t_delta_i(d); }
#if !defined(ZK_HOSTSIM)
// next == curr + d in the field for two cells known to fit 64 bits and a small signed d: the sum leaves [0, 2^64) exactly
// when the field value does (curr + d >= 2^64, or p - |curr + d|), and then it cannot equal `next`
ZK_HD bool stage_delta_ok(u64 c, u64 n, long long d) {
    if (d >= 0) {
        const u64 e = c + (u64)d;
        return e >= c && e == n;
    }
    const u64 m = (u64)(-d);
    return c >= m && c - m == n;
}
ZK_HD bool stage_trans_ok(u64 c, u64 n, u32 kind, const Fr& v) {  // v fits 64 bits when kind == 1 (caller-checked)
    if (kind == 0u) return n == c;
    if (kind == 1u) {
        const u64 e = c + fr_lo64(v);
        return e >= c && e == n;
    }
    return fr_fits64(v) && fr_lo64(v) == n;
}
// same_context on the LDS-staged pair: every cell is known to fit 64 bits (128 for the code hash), so the eleven
// transitions are 64-bit compares instead of 256-bit field additions.  Same checkpoints, same order, same verdicts.
ZK_HD void same_context_staged(Ins& I, const Tail& T, u64 dyn_gas) {
    const Fr& opcode = T.opcode;
    const bool op_byte = fr_le_u64(opcode, 255);
    const u32 info = (opcode.v[0] & 0xffu) == I.op_info_byte ? I.op_info : zk_opinfo(opcode.v[0]);
#define STG(s, c) ((u64)I.stage[evm_stage_entry(s, c) * EVM_STAGE_STRIDE])
#define STG_GAS(s) (STG(s, S_GAS) | ((u64)I.stage[(evm_stage_entry(s, S_GAS) + 1) * EVM_STAGE_STRIDE] << 32))
    const u64 st = STG(0, S_STATE);
    I.seq++;
    if (!(op_byte && st != 0u && (u64)(info & 0xffu) == st)) ev_fail(I, ZK_LOOKUP_UNSAT);
    const bool op_ok = op_byte && ((info >> 8) & 1u);
    ev_require(I, op_ok, ZK_VALUE_ERROR);
    const u64 gas_cost = (u64)(op_ok ? (info >> 16) : 0u) + dyn_gas;  // caller: no 64-bit overflow
    const u64 c_gas = STG_GAS(0);
    ev_require(I, c_gas >= gas_cost, ZK_CONSTRAINT);  // range_check(gas_left - gas_cost, 8)
    if (T.rwc_mode == 0u) ev_require(I, stage_delta_ok(STG(0, S_RWC), STG(1, S_RWC), T.rw_delta));
    else ev_require(I, T.rwc_mode == 1u);
    ev_require(I, stage_trans_ok(STG(0, S_PC), STG(1, S_PC), T.pc_kind, T.pc_val));
    ev_require(I, stage_delta_ok(STG(0, S_SP), STG(1, S_SP), T.sp_delta));
    ev_require(I, c_gas >= gas_cost && c_gas - gas_cost == STG_GAS(1));
    ev_require(I, stage_trans_ok(STG(0, S_MWS), STG(1, S_MWS), T.mws_kind, T.mws_val));
    ev_require(I, stage_delta_ok(STG(0, S_REV), STG(1, S_REV), T.rev_delta));
    if (T.log_mode == 0u) ev_require(I, STG(0, S_LOG) == STG(1, S_LOG));
    else ev_require(I, T.log_mode == 1u);
    ev_require(I, STG(0, S_CALL_ID) == STG(1, S_CALL_ID));
    ev_require(I, STG(0, S_IS_ROOT) == STG(1, S_IS_ROOT));
    ev_require(I, STG(0, S_IS_CREATE) == STG(1, S_IS_CREATE));
    bool same_hash = true;
#pragma unroll
    for (int w = 0; w < 8; w++) same_hash = same_hash & (I.stage[(24 + w) * EVM_STAGE_STRIDE] == I.stage[(32 + w) * EVM_STAGE_STRIDE]);
    ev_require(I, same_hash);
#undef STG_GAS
#undef STG
}
#endif
ZK_HD void same_context(Ins& I, const Tail& T) {
#if !defined(ZK_HOSTSIM)
    // (EVM_FAST: every pair that gets here is staged — and `I.stage` must not be tested there: the stage column of the block's
    // lane 0 starts at LDS address 0, which reads as a null pointer; in the general build that lane silently took the
    // field-arithmetic form below)
    if ((EVM_FAST || I.stage) && fr_fits64(T.dyn_gas) && fr_lo64(T.dyn_gas) < (1ull << 62) && (T.pc_kind != 1u || fr_fits64(T.pc_val)) &&
        (T.mws_kind != 1u || fr_fits64(T.mws_val))) {
        same_context_staged(I, T, fr_lo64(T.dyn_gas));
        return;
    }
#endif
#if EVM_FAST
    I.defer = 5u;  // transition operands beyond 64 bits: the field-arithmetic form below, in the general build
    return;
#endif
    const Fr& opcode = T.opcode;
    // responsible-opcode membership (success states: (state, opcode, 0) rows, table.py:71-79), opcode
    // validity and the constant gas come from ONE packed table word
    static const uint32_t opinfo[256] = ZK_OPINFO_INIT;
    const bool op_byte = fr_le_u64(opcode, 255);
    const u32 info = opinfo[opcode.v[0] & 0xff];
    const u32 st = ev_curr(I, S_STATE).v[0];
    I.seq++;
    if (!(op_byte && st != 0u && (info & 0xffu) == st && fr_fits32(ev_curr(I, S_STATE)))) ev_fail(I, ZK_LOOKUP_UNSAT);
    const bool op_ok = op_byte && ((info >> 8) & 1u);
    ev_require(I, op_ok, ZK_VALUE_ERROR);  // Opcode(opcode.n)
    Fr gas_cost = fr_add(fr_u(op_ok ? (info >> 16) : 0), T.dyn_gas);
    range_check(I, fr_sub(ev_curr(I, S_GAS), gas_cost), 8);
    Trans pc, mws;
    pc.kind = T.pc_kind; pc.value = T.pc_val;
    mws.kind = T.mws_kind; mws.value = T.mws_val;
    if (T.rwc_mode == 0u) transition(I, S_RWC, t_delta_i(T.rw_delta));  // Transition.delta(0) == same
    else ev_require(I, T.rwc_mode == 1u);
    transition(I, S_PC, pc);
    transition(I, S_SP, t_int(T.sp_delta));
    transition(I, S_GAS, t_delta(fr_neg(gas_cost)));
    transition(I, S_MWS, mws);
    transition(I, S_REV, t_int(T.rev_delta));
    if (T.log_mode == 0u) transition(I, S_LOG, t_same());
    else ev_require(I, T.log_mode == 1u);
    transition(I, S_CALL_ID, t_same());
    transition(I, S_IS_ROOT, t_same());
    transition(I, S_IS_CREATE, t_same());
    ev_require(I, fr_eq(ev_next(I, S_CH_LO), ev_curr(I, S_CH_LO)) && fr_eq(ev_next(I, S_CH_HI), ev_curr(I, S_CH_HI)));
}

// constant_divmod (instruction.py:440-445) with a small constant denominator (power of two here)
ZK_HD Fr constant_divmod_shift(Ins& I, const Fr& num, int shift, int n_bytes) {
    Fr q = fr_zero();
    for (int k = 0; k < 8; k++) {
        u32 lo = num.v[k] >> shift;
        u32 hi = (k + 1 < 8) ? (num.v[k + 1] << (32 - shift)) : 0u;
        q.v[k] = lo | hi;
    }
    range_check(I, q, n_bytes);
    return q;
}
ZK_HD Fr memory_gas_cost(Ins& I, const Fr& size) {  // instruction.py:1122-1129
    // size * size in the field; a word count below 2^32 squares inside 64 bits (no Montgomery products)
    const Fr sq = fr_fits32(size) ? fr_u((u64)size.v[0] * (u64)size.v[0]) : fr_mul(size, size);
    Fr q = constant_divmod_shift(I, sq, 9, 8);
    return fr_add(q, fr_add(fr_add(size, size), size));
}
ZK_HD void memory_expansion(Ins& I, const Fr& offset, const Fr& length, Fr& next_size, Fr& gas) {  // :1131-1148
    Fr mws = ev_curr(I, S_MWS);
    Fr mem_size = fr_zero();
    if (!fr_is_zero(length)) mem_size = constant_divmod_shift(I, fr_add_u64(fr_add(length, offset), 31), 5, 4);
    u32 lt, eq;
    ev_compare(I, mws, mem_size, 4, lt, eq);
    next_size = ev_select_b(I, lt) ? mem_size : mws;
    Fr g0 = memory_gas_cost(I, mws);
    Fr g1 = memory_gas_cost(I, next_size);
    gas = fr_sub(g1, g0);
}

// ---- gadgets (evm_circuit/execution/*.py) --------------------------------------------------------
ZK_HD void g_add_sub(Ins& I, Tail& T) {  // add_sub.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_sub = fr_eq_u64(opcode, OP_SUB);
    Word a, b, c;
    a = stack_pop(I, R, 0);
    b = stack_pop(I, R, 1);
    c = stack_push(I, R, 2);
    Word x = ev_select_b(I, is_sub) ? c : a;
    Fr carry;
    Word res = add_words2(I, x, b, carry);
    Word y = ev_select_b(I, is_sub) ? a : c;
    constrain_equal_word(I, res, y);
    set_tail3(T, opcode, 3, 1, 1);
}

// x * s in the field; `is01` = s is known to be 0 or 1 (then the product is a select, no Montgomery multiplication)
ZK_HD Fr fr_sel01(const Fr& x, const Fr& s, bool is01) {
    if (is01) return fr_is_zero(s) ? fr_zero() : x;
    return fr_mul(x, s);
}
ZK_HD void g_mul_div_mod(Ins& I, Tail& T) {  // mul_div_mod.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    // is_mul/is_div/is_mod are field expressions of the opcode (:14-16)
    // for the three opcodes the gadget is responsible for they are exactly 0 / 1, and every product with them below is a select
    const bool op_known = fr_eq_u64(opcode, OP_MUL) || fr_eq_u64(opcode, OP_DIV) || fr_eq_u64(opcode, OP_MOD);
    Fr is_mul, is_div, is_mod;
    if (op_known) {
        is_mul = fr_u(fr_eq_u64(opcode, OP_MUL) ? 1 : 0);
        is_div = fr_u(fr_eq_u64(opcode, OP_DIV) ? 1 : 0);
        is_mod = fr_u(fr_eq_u64(opcode, OP_MOD) ? 1 : 0);
    } else {
        Fr op_m2 = fr_sub_u64(opcode, 2), op_m4 = fr_sub_u64(opcode, 4);
        Fr f4_op = fr_sub(fr_u(4), opcode), f6_op = fr_sub(fr_u(6), opcode);
        is_mul = fr_mulc(fr_mul(f4_op, f6_op), frm_inv8());
        is_div = fr_mulc(fr_mul(op_m2, f6_op), frm_inv4());
        is_mod = fr_mulc(fr_mul(op_m2, op_m4), frm_inv8());
    }
    Word pop1, pop2, push;
    pop1 = stack_pop(I, R, 0);
    pop2 = stack_pop(I, R, 1);
    push = stack_push(I, R, 2);
    Word a, b, c, d;
    if (fr_eq_u64(is_mul, 1)) {
        a = pop1; b = pop2; c = word_from_int(I, fr_zero()); d = push;
    } else if (fr_eq_u64(is_div, 1)) {
        d = pop1; b = pop2; a = push;
        if (words_wide(I, d, b, a)) {
#if !EVM_FAST
            const WideRes W = wide_words(WIDE_SUB_MUL, d, b, a, a);
            EV_TRY(c = word_from_wide(I, W, 0));
#endif
        } else {
            U256 dv, bv, av;
            EV_TRY(dv = int_value(I, d)); EV_TRY(bv = int_value(I, b)); EV_TRY(av = int_value(I, a));
            U512 prod = u256_mul_full(bv, av);
            U256 plo = u512_lo(prod), rem;
            bool neg = !fr_is_zero(u512_hi(prod)) || u256_sub(rem, dv, plo);
            EV_TRY(c = word_from_int(I, rem, neg));
        }
    } else {
        d = pop1; b = pop2;
        if (word_is_zero_int(b)) {
            c = d; a = word_from_int(I, fr_zero());
        } else if (words_wide(I, d, b, push)) {
            c = push;
#if !EVM_FAST
            const WideRes W = wide_words(WIDE_MOD_QUOT, d, b, c, c);
            EV_TRY(a = word_from_wide(I, W, 0));
#endif
        } else {
            c = push;
            U256 dv, bv, cv;
            EV_TRY(dv = int_value(I, d)); EV_TRY(bv = int_value(I, b)); EV_TRY(cv = int_value(I, c));
            U256 diff, q, r;
            bool neg = u256_sub(diff, dv, cv);
            u256_divmod(diff, bv, q, r);
            EV_TRY(a = word_from_int(I, q, neg));
        }
    }
    const u32 dz = is_zero_word(b);
    Fr overflow; EV_TRY(overflow = mul_add_words(I, a, b, c, d));
    bool sel = ev_select(I, is_mul); if (I.err) return;
    constrain_equal_word(I, pop1, sel ? a : d);
    constrain_equal_word(I, pop2, b);
    Fr nz = fr_u(1 - dz);
    Fr s1 = fr_sel01(is_div, nz, op_known), s2 = fr_sel01(is_mod, nz, op_known);
    const bool sel_known = op_known;  // then s1, s2 are 0 / 1 too
    Word w1 = word_checked(I, fr_sel01(d.lo, is_mul, op_known), fr_sel01(d.hi, is_mul, op_known));
    Word w2 = word_checked(I, fr_sel01(a.lo, s1, sel_known), fr_sel01(a.hi, s1, sel_known));
    Word w12 = word_checked(I, fr_add(w1.lo, w2.lo), fr_add(w1.hi, w2.hi));
    Word w3 = word_checked(I, fr_sel01(c.lo, s2, sel_known), fr_sel01(c.hi, s2, sel_known));
    Word rhs = word_checked(I, fr_add(w12.lo, w3.lo), fr_add(w12.hi, w3.hi));
    constrain_equal_word(I, push, rhs);
    U256 cb = to_u256(I, c); if (I.err) return;
    u32 csum = 0;
    for (int k = 0; k < 32; k++) csum += fr_byte(cb, k);
    constrain_zero(I, fr_sel01(fr_u(csum), is_mul, op_known));
    u32 lt, eq; compare_word(I, c, b, lt, eq); if (I.err) return;
    Fr one_m_mul = fr_sub(fr_u(1), is_mul);
    constrain_zero(I, fr_sel01(fr_sel01(nz, one_m_mul, op_known), fr_u(1 - lt), true));
    constrain_zero(I, fr_sel01(overflow, one_m_mul, op_known));
    set_tail3(T, opcode, 3, 1, 1);
}

ZK_HD void g_cmp(Ins& I, Tail& T) {  // comparator.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_eq = fr_eq_u64(opcode, OP_EQ), is_gt = fr_eq_u64(opcode, OP_GT);
    Word a, b, c;
    a = stack_pop(I, R, 0); b = stack_pop(I, R, 1); c = stack_push(I, R, 2);
    Word aa = is_gt ? b : a, bb = is_gt ? a : b;
    u32 lt_lo, eq_lo, lt_hi, eq_hi;
    ev_compare(I, aa.lo, bb.lo, 16, lt_lo, eq_lo);
    ev_compare(I, aa.hi, bb.hi, 16, lt_hi, eq_hi);
    u32 lt = ev_select_b(I, lt_hi) ? 1u : eq_hi * lt_lo;
    u32 eq = eq_lo * eq_hi;
    u32 result = is_eq ? eq : lt;
    Word rw = word_checked(I, fr_u(result), fr_zero());
    constrain_equal_word(I, rw, c);
    set_tail3(T, opcode, 3, 1, 1);
}

// slt_sgt.py:31-36 / addmod.py:7-19: lo compare, hi compare, inner select, outer select
ZK_HD u32 lt_u256_sel(Ins& I, const Word& a, const Word& b) {
    u32 lt_lo, eq_lo, lt_hi, eq_hi;
    ev_compare(I, a.lo, b.lo, 16, lt_lo, eq_lo);
    ev_compare(I, a.hi, b.hi, 16, lt_hi, eq_hi);
    u32 inner = ev_select_b(I, eq_hi * lt_lo) ? 1u : 0u;
    return ev_select_b(I, lt_hi) ? 1u : inner;
}

ZK_HD void g_scmp(Ins& I, Tail& T) {  // slt_sgt.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_sgt = fr_eq_u64(opcode, OP_SGT);
    Word a, b, c;
    a = stack_pop(I, R, 0); b = stack_pop(I, R, 1); c = stack_push(I, R, 2);
    Word aa = is_sgt ? b : a, bb = is_sgt ? a : b;
    U256 a8, b8, c8;
    EV_TRY(a8 = to_u256(I, aa)); EV_TRY(b8 = to_u256(I, bb)); EV_TRY(c8 = to_u256(I, c));
    ev_require(I, fr_byte(c8, 31) == 0); if (I.err) return;
    Fr cc = c8;  // low 31 bytes (byte 31 is zero) as a field element
    u32 a_lt_b; EV_TRY(a_lt_b = lt_u256_sel(I, aa, bb));
    const u32 am = fr_byte(a8, 31), bm = fr_byte(b8, 31);
    if (am >= 128 && bm < 128) constrain_equal(I, cc, fr_u(1));
    else if (bm >= 128 && am < 128) constrain_equal(I, cc, fr_zero());
    else constrain_equal(I, cc, fr_u(a_lt_b));
    set_tail3(T, opcode, 3, 1, 1);
}

ZK_HD void g_iszero(Ins& I, Tail& T) {  // iszero.py
    RwRows<2> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word value; value = stack_pop(I, R, 0);
    Word z = word_checked(I, fr_u(is_zero_word(value)), fr_zero());
    Word push; push = stack_push(I, R, 1);
    constrain_equal_word(I, z, push);
    set_tail3(T, opcode, 2, 1, 0);
}

// 32 byte-wise fixed lookups (BitwiseAnd / Or / Xor, table.py:56-69) over three 256-bit words at once: `diff` has a non-zero
// byte k exactly where lookup k misses; the checkpoints are those of the 32 one-by-one lookups (first miss wins)
ZK_HD void bitwise_lookups32(Ins& I, const U256& diff) {
    int first = -1;
    for (int k = 31; k >= 0; k--)
        if (fr_byte(diff, k) != 0u) first = k;
    if (first >= 0 && I.err == 0u) I.err = ZK_CODE(ZK_LOOKUP_UNSAT, I.seq + (u32)first + 1u);
    I.seq += 32;
}
ZK_HD void g_not(Ins& I, Tail& T) {  // not_.py
    RwRows<2> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word a; a = stack_pop(I, R, 0);
    U256 a8; EV_TRY(a8 = to_u256(I, a));
    Word b; b = stack_push(I, R, 1);
    U256 b8; EV_TRY(b8 = to_u256(I, b));
    {   // (a_byte, b_byte, 255) in BitwiseXor for every byte  <=>  a ^ b == 0xff..ff
        U256 d;
        for (int k = 0; k < 8; k++) d.v[k] = ~(a8.v[k] ^ b8.v[k]);
        bitwise_lookups32(I, d);
    }
    if (I.err) return;
    set_tail3(T, opcode, 2, 1, 0);
}

ZK_HD void g_bitwise(Ins& I, Tail& T) {  // bitwise.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word a, b, c;
    a = stack_pop(I, R, 0); b = stack_pop(I, R, 1); c = stack_push(I, R, 2);
    U256 a8, b8, c8;
    EV_TRY(a8 = to_u256(I, a)); EV_TRY(b8 = to_u256(I, b)); EV_TRY(c8 = to_u256(I, c));
    // tag = BitwiseAnd + (opcode.n - AND) as a Python int, then FixedTableTag(tag)
    Fr tagf = fr_zero();
    bool tag_ok;
    {
        // opcode.n - 0x16 + 10 = opcode.n - 12 (integer arithmetic on the canonical value)
        Fr twelve = fr_u(12);
        bool neg = u256_sub(tagf, opcode, twelve);
        tag_ok = !neg && fr_lo64(tagf) >= 1 && fr_le_u64(tagf, 16);
    }
    ev_require(I, tag_ok, ZK_VALUE_ERROR); if (I.err) return;
    const u32 tag = tagf.v[0];
    if (tag == FX_BitwiseAnd || tag == FX_BitwiseOr || tag == FX_BitwiseXor) {
        U256 d;
        for (int k = 0; k < 8; k++) {
            const u32 x = a8.v[k], y = b8.v[k];
            d.v[k] = (tag == FX_BitwiseAnd ? (x & y) : (tag == FX_BitwiseOr ? (x | y) : (x ^ y))) ^ c8.v[k];
        }
        bitwise_lookups32(I, d);
    } else {
        for (int k = 0; k < 32; k++) fixed_lookup(I, tag, fr_u(fr_byte(a8, k)), fr_u(fr_byte(b8, k)), fr_u(fr_byte(c8, k)));
    }
    if (I.err) return;
    set_tail3(T, opcode, 3, 1, 1);
}

ZK_HD void g_byte(Ins& I, Tail& T) {  // byte.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word a, b, c;
    a = stack_pop(I, R, 0); b = stack_pop(I, R, 1); c = stack_push(I, R, 2);
    U256 index, value;
    EV_TRY(index = to_u256(I, a)); EV_TRY(value = to_u256(I, b));
    bool msb_zero = true;
    for (int k = 1; k < 32; k++) msb_zero = msb_zero && fr_byte(index, k) == 0;
    const u32 i0 = fr_byte(index, 0);
    u32 selected = 0;
    for (int k = 0; k < 32; k++) selected += (i0 == (u32)(31 - k) && msb_zero) ? fr_byte(value, k) : 0u;
    Word sw = word_checked(I, fr_u(selected), fr_zero());
    constrain_equal_word(I, sw, c);
    set_tail3(T, opcode, 3, 1, 1);
}

ZK_HD void g_signextend(Ins& I, Tail& T) {  // signextend.py (is_equal results are discarded there)
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word index, value, result;
    index = stack_pop(I, R, 0); value = stack_pop(I, R, 1); result = stack_push(I, R, 2);
    U256 ib, vb, rb;
    EV_TRY(ib = to_u256(I, index)); EV_TRY(vb = to_u256(I, value)); EV_TRY(rb = to_u256(I, result));
    bool msb_zero = true;
    for (int k = 1; k < 32; k++) msb_zero = msb_zero && fr_byte(ib, k) == 0;
    const u32 i0 = fr_byte(ib, 0);
    const u32 sign_byte = i0 < 31 ? (fr_byte(vb, (int)i0) >> 7) * 0xffu : 0u;
    u32 selected = 0;
    for (int k = 0; k < 31; k++) selected += (i0 == (u32)k && msb_zero) ? fr_byte(vb, k) : 0u;
    fixed_lookup(I, FX_SignByte, fr_u(selected), fr_u(sign_byte), fr_zero()); if (I.err) return;
    set_tail3(T, opcode, 3, 1, 1);
}

// constrain_zero on bytes [lo, hi) of v, one checkpoint each (first non-zero byte wins), byte positions static
ZK_HD void push_zero_run(Ins& I, const U256& v, u32 lo, u32 hi) {
    int first = -1;
#pragma unroll
    for (int k = 31; k >= 0; k--)
        if ((u32)k >= lo && (u32)k < hi && fr_byte(v, k) != 0u) first = k;
    if (first >= 0 && I.err == 0u) I.err = ZK_CODE(ZK_ASSERT, I.seq + ((u32)first - lo) + 1u);
    I.seq += hi - lo;
}
// v >> (8 * n), n <= 32
ZK_HD U256 u256_shr_bytes(const U256& v, u32 n) {
    U256 r = v;
#pragma unroll
    for (int bit = 0; bit < 5; bit++) {  // shift by 1, 2, 4, 8, 16 bytes
        const bool on = (n >> bit) & 1u;
        const int bytes = 1 << bit;
        U256 t;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int src = j + bytes / 4;
            const u32 lo_w = src < 8 ? r.v[src] : 0u, hi_w = src + 1 < 8 ? r.v[src + 1] : 0u;
            const int sh = 8 * (bytes % 4);
            t.v[j] = sh ? ((lo_w >> sh) | (hi_w << (32 - sh))) : lo_w;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) r.v[j] = on ? t.v[j] : r.v[j];
    }
    if (n >= 32u) r = fr_zero();
    return r;
}
ZK_HD u32 zk_brev32(u32 x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}
ZK_HD void g_push(Ins& I, Tail& T) {  // push.py
    // The pushed bytes of a regular code are the 2-byte packed records right after the program counter: request all 32 before
    // anything else is waited for (independent loads, in flight together with the opcode record and the stack row) instead
    // of one dependent lookup per byte.  A record beyond the code's end reads as 0 = "does not fit the packed form".
    RwRows<1> R; rw_rows_fetch(I, R);
    u32 raw[32];
    bool run_ok = false;
    {
        code_dir_resolve(I, curr_code_hash(I));
        const uint16_t* packed = I.a->codes.packed;
        if (I.code_state == 1u && packed != nullptr && fr_fits64(I.pc) && fr_lo64(I.pc) < (1ull << 62)) {
            run_ok = true;
            const u64 first = fr_lo64(I.pc) + 1, nb = (u64)I.code_n_bytes;
            const uint16_t* src = packed + (u64)I.code_byte_base + first;
#pragma unroll
            for (int i = 0; i < 32; i++) raw[i] = first + (u64)i < nb ? (u32)src[i] : 0u;
        }
    }
    Fr opcode; opcode = opcode_lookup(I, true);
    Fr num_pushed = fr_sub_u64(opcode, OP_PUSH0);
    Fr code_length; code_length = bytecode_length(I, curr_code_hash(I));
    const Fr& pc = I.pc;
    Fr left = fr_sub_u64(fr_sub(code_length, pc), 1);
    u32 oob, eq; ev_compare(I, left, num_pushed, 8, oob, eq); if (I.err) return;
    Fr num_padding = oob ? fr_sub(num_pushed, left) : fr_zero();
    Word value; value = stack_push(I, R, 0);
    U256 vb; EV_TRY(vb = to_u256(I, value));
    if (fr_fits64(num_pushed) && fr_fits64(num_padding) && fr_fits64(pc) && fr_lo64(num_pushed) <= 32 && fr_lo64(pc) < (1ull << 62)) {
        // the usual case: byte counts and the program counter are small integers.  In byte order the loop of push.py:27-36 is
        // [0, pad) constrain_zero, [pad, np) bytecode lookup + constrain_equal, [np, 32) constrain_zero: the two zero runs
        // are scanned with static byte positions, the lookups walk a shifted copy (no dynamically indexed registers)
        const u32 np = (u32)fr_lo64(num_pushed);
        const u32 pad = fr_lo64(num_padding) < (u64)np ? (u32)fr_lo64(num_padding) : np;
        const u64 base = fr_lo64(pc) + np;
        push_zero_run(I, vb, 0, pad);
        // value byte k in [pad, np) is looked up at index pc + np - k, i.e. record i = np - 1 - k of the run fetched above
        const u32 cnt = np - pad;
        const u32 need = cnt >= 32u ? 0xffffffffu : ((1u << cnt) - 1u);
        u32 fits_i = 0, code_i = 0;
        if (run_ok) {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                fits_i |= ((raw[i] >> 15) & 1u) << i;
                code_i |= ((raw[i] >> 8) & 1u) << i;
            }
        }
        if (run_ok && cnt != 0u && (fits_i & need) == need) {
            // every record fits: run the cnt (lookup, constrain_equal) checkpoint pairs on bit masks.  The lookup of byte k
            // fails when the record is an opcode byte (is_code = 0 is queried: LookupUnsatFailure), constrain_equal when
            // the values differ; the first failing checkpoint in loop order is reported, as the byte-by-byte loop does.
            U256 rev = fr_zero();  // byte 31 - i = record i's value
#pragma unroll
            for (int i = 0; i < 32; i++) rev.v[(31 - i) >> 2] |= (raw[i] & 0xffu) << (8 * ((31 - i) & 3));
            const U256 want = u256_shr_bytes(rev, 32u - np);  // byte k = record np - 1 - k
            const u32 code_k = zk_brev32(code_i) >> (32u - np);
            u32 neq_k = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) neq_k |= (fr_byte(vb, k) != fr_byte(want, k) ? 1u : 0u) << k;
            const u32 range = (np >= 32u ? 0xffffffffu : ((1u << np) - 1u)) & ~((1u << pad) - 1u);
            const u32 bad = (code_k | neq_k) & range;
            if (bad != 0u && I.err == 0u) {
                const u32 k = (u32)__builtin_ctz(bad);
                I.err = ((code_k >> k) & 1u) ? ZK_CODE(ZK_LOOKUP_UNSAT, I.seq + 2u * (k - pad) + 1u) : ZK_CODE(ZK_ASSERT, I.seq + 2u * (k - pad) + 2u);
            }
            I.seq += 2u * cnt;
        } else {
            U256 cur = u256_shr_bytes(vb, pad);
            for (u32 k = pad; k < np; k++) {
                Fr byte = opcode_lookup_at(I, fr_u(base - (u64)k), false);
                constrain_equal(I, fr_u(cur.v[0] & 0xffu), byte);
#pragma unroll
                for (int j = 0; j < 7; j++) cur.v[j] = (cur.v[j] >> 8) | (cur.v[j + 1] << 24);
                cur.v[7] >>= 8;
            }
        }
        push_zero_run(I, vb, np, 32);
    } else {
        for (int k = 0; k < 32; k++) {
            const bool pushed = fr_lt(fr_u((u64)k), num_pushed), padding = fr_lt(fr_u((u64)k), num_padding);
            if (pushed && !padding) {
                Fr index = fr_sub_u64(fr_add(pc, num_pushed), (u64)k);
                Fr byte = opcode_lookup_at(I, index, false);
                constrain_equal(I, fr_u(fr_byte(vb, k)), byte);
            } else {
                constrain_zero(I, fr_u(fr_byte(vb, k)));
            }
        }
    }
    set_tail(T, opcode, 1, t_delta(fr_add_u64(num_pushed, 1)), -1, t_same(), 0, fr_zero());
}

ZK_HD void g_pop(Ins& I, Tail& T) {  // pop.py
    RwRows<1> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    stack_pop(I, R, 0);
    set_tail3(T, opcode, 1, 1, 1);
}

ZK_HD void g_shl_shr(Ins& I, Tail& T) {  // shl_shr.py
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word pop1, pop2, push;
    pop1 = stack_pop(I, R, 0); pop2 = stack_pop(I, R, 1); push = stack_push(I, R, 2);
    // gen_witness (:103-127)
    Fr is_shl = fr_sub(fr_u(OP_SHR), opcode);
    const bool op_known = fr_eq_u64(opcode, OP_SHL) || fr_eq_u64(opcode, OP_SHR);  // is_shl, is_shr are 0 / 1: products below are selects
    Word shift = pop1;
    U256 sb; EV_TRY(sb = to_u256(I, shift));
    const u32 shf0 = fr_byte(sb, 0);
    bool rest_zero = true;
    for (int k = 1; k < 32; k++) rest_zero = rest_zero && fr_byte(sb, k) == 0;
    U256 dvs = fr_zero();
    if (rest_zero) dvs.v[shf0 >> 5] = 1u << (shf0 & 31u);
    Word divisor = word_from_int(I, dvs);
    Word dividend, quotient, remainder;
    if (fr_eq_u64(is_shl, 1)) {
        dividend = push; quotient = pop2; remainder = word_from_int(I, fr_zero());
    } else {
        dividend = pop2; quotient = push;
        if (words_wide(I, dividend, quotient)) {
#if !EVM_FAST
            const WideRes W = wide_words(WIDE_SUB_MUL, dividend, quotient, divisor, divisor);
            EV_TRY(remainder = word_from_wide(I, W, 0));
#endif
        } else {
            U256 dv, qv;
            EV_TRY(dv = int_value(I, dividend)); EV_TRY(qv = int_value(I, quotient));
            U512 prod = u256_mul_full(qv, dvs);
            U256 rem;
            bool neg = !fr_is_zero(u512_hi(prod)) || u256_sub(rem, dv, u512_lo(prod));
            EV_TRY(remainder = word_from_int(I, rem, neg));
        }
    }
    // check_witness (:35-100)
    Fr is_shr = fr_sub(fr_u(1), is_shl);
    EV_TRY(sb = to_u256(I, shift));
    const u32 shf_lt256 = rest_zero ? 1u : 0u;
    const u32 dz = is_zero_word(divisor);
    constrain_equal_word(I, pop1, shift);
    {
        Word w1 = word_checked(I, fr_sel01(quotient.lo, is_shl, op_known), fr_sel01(quotient.hi, is_shl, op_known));
        Word w2 = word_checked(I, fr_sel01(dividend.lo, is_shr, op_known), fr_sel01(dividend.hi, is_shr, op_known));
        Word s = word_checked(I, fr_add(w1.lo, w2.lo), fr_add(w1.hi, w2.hi));
        constrain_equal_word(I, pop2, s);
    }
    {
        Fr s = fr_sel01(is_shr, fr_u(1 - dz), true);
        Word w1 = word_checked(I, fr_sel01(dividend.lo, is_shl, op_known), fr_sel01(dividend.hi, is_shl, op_known));
        Word w2 = word_checked(I, fr_sel01(quotient.lo, s, op_known), fr_sel01(quotient.hi, s, op_known));
        Word sum = word_checked(I, fr_add(w1.lo, w2.lo), fr_add(w1.hi, w2.hi));
        constrain_equal_word(I, push, sum);
    }
    constrain_zero(I, fr_zero());  // shf0 - shift_le_bytes[0]: same byte by construction (:67)
    {
        Word lhs = dz ? word_checked(I, fr_zero(), fr_zero()) : word_checked(I, shift.lo, shift.hi);
        word_checked(I, fr_u(shf0), fr_zero());
        Word rhs = word_checked(I, dz ? fr_zero() : fr_u(shf0), fr_zero());
        constrain_equal_word(I, lhs, rhs);
    }
    ev_require(I, (1 - (int)dz - (int)shf_lt256) == 0);
    u32 rlt, req; compare_word(I, remainder, divisor, rlt, req);
    ev_require(I, dz == 1 || rlt == 1);
    constrain_zero(I, is_zero_word(remainder) ? fr_zero() : is_shl);
    if (I.err) return;
    Fr overflow; EV_TRY(overflow = mul_add_words(I, quotient, divisor, remainder, dividend));
    constrain_zero(I, fr_sel01(overflow, is_shr, op_known));
    if (dz == 0) fixed_lookup(I, FX_Pow2, fr_u(shf0), divisor.lo, divisor.hi);
    if (I.err) return;
    set_tail3(T, opcode, 3, 1, 1);
}

// sar.py.  Once the three words pass their to_le_bytes() checks, every constraint of
// check_witness (:53-151) except `b64s[idx] == bytes_to_fq(b_le_bytes[..])` is an identity of
// gen_witness's own outputs (:154-199): those only advance the checkpoint counter.
ZK_HD void g_sar(Ins& I, Tail& T) {
    RwRows<3> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    Word shift, a, b;
    shift = stack_pop(I, R, 0); a = stack_pop(I, R, 1); b = stack_push(I, R, 2);
    // is_neg = a.int_value() >> 255 (:158) only feeds witness values computed after a.to_64s() (:166), which raises for cells
    // >= 2^128: with such cells its value is never used
    (void)words_wide(I, a);
    const U256 av = u256_from_lo_hi(a.lo, a.hi);
    U256 sb; EV_TRY(sb = to_u256(I, shift));
    EV_TRY(to_u256(I, a));  // a.to_64s()
    const u32 is_neg = av.v[7] >> 31;
    const u32 shf0 = fr_byte(sb, 0), dv = shf0 >> 6, md = shf0 & 63u;
    bool rest_zero = true;
    for (int k = 1; k < 32; k++) rest_zero = rest_zero && fr_byte(sb, k) == 0;
    const u64 fill = is_neg ? ~0ull : 0ull;
    u64 b64[4] = {fill, fill, fill, fill};
    if (rest_zero) {
        u64 a_lo[4], a_hi[4];
        for (int k = 0; k < 4; k++) {
            const u64 x = u256_limb64(av, k);
            a_hi[k] = x >> md;
            a_lo[k] = md ? (x << (64 - md)) : 0ull;  // (x mod 2^md) * 2^(64-md)
        }
        const u64 p_top = (is_neg && md) ? ~0ull << (64 - md) : 0ull;  // is_neg * (2^64 - 2^(64-md))
        for (int k = 0; k < 4; k++) {
            const u32 src = k + dv;
            if (src == 3) b64[k] = a_hi[3] + p_top;
            else if (src < 3) b64[k] = a_hi[src] + a_lo[src + 1];
        }
    }
    U256 bb;
    I.seq++;  // a.to_le_bytes()
    EV_TRY(bb = to_u256(I, b));
    I.seq += 2;  // shift.to_le_bytes(), compare(127, a_le_bytes[31], 1)
    for (int k = 0; k < 4; k++) {
        I.seq++;
        ev_require(I, b64[k] == u256_limb64(bb, k));
        if (I.err) return;
        I.seq += 5;
    }
    I.seq += 15;  // merge constraints, shift decomposition, is_neg, sign-byte / pow2 lookups
    set_tail3(T, opcode, 3, 1, 1);
}

// two's complement negation of a 256-bit integer (get_int_neg, util/arithmetic.py:283-284)
ZK_HD U256 u256_neg(const U256& x) {
    U256 r;
    u256_sub(r, fr_zero(), x);
    return r;
}
// Instruction.abs_word (instruction.py:539-571): is_neg from the hi cell; the six constraints are
// identities of the witness x_abs = x or 2^256 - x.
ZK_HD Word abs_word(Ins& I, const Word& x, u32& is_neg) {
    u32 eq;
    is_neg = 0;
    ev_compare(I, fr_from_u128(~0ull, 0x7fffffffffffffffull), x.hi, 16, is_neg, eq);
    if (I.err) return x;
    Word x_abs = x;
    if (is_neg) {
        if (words_wide(I, x)) {  // x.hi fits (the compare above), x.lo does not: 2^256 - (lo + (hi << 128)) in full
#if !EVM_FAST
            const WideRes W = wide_words(WIDE_NEG256, x, x, x, x);
            x_abs = word_from_wide(I, W, 0);
            if (I.err) return x;
#endif
        } else {
            U256 v = int_value(I, x);
            if (I.err) return x;
            x_abs = word_from_int(I, u256_neg(v));
        }
    }
    I.seq += 6;
    return x_abs;
}
ZK_HD void g_sdiv_smod(Ins& I, Tail& T) {  // sdiv_smod.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word pop1, pop2, push;
    pop1 = stack_pop(I); pop2 = stack_pop(I); push = stack_push(I);
    // gen_witness (:79-119); is_sdiv = (SMOD - opcode) / 2 equals 1 exactly for SDIV
    const bool is_sdiv = fr_eq_u64(opcode, OP_SDIV);
    Word quotient, divisor = pop2, remainder, dividend = pop1;
    if (words_wide(I, pop1, pop2, push)) {
#if !EVM_FAST
        // unbounded-integer forms (get_int_abs of a value >= 2^256 is negative, `//` floors, a divisor of exactly 2^256 has
        // get_int_abs == 0: ZeroDivisionError, raised outside every checkpoint)
        if (is_sdiv) {
            quotient = push;
            const WideRes W = wide_words(WIDE_SDIV, pop1, pop2, push, push);
            EV_TRY(remainder = word_from_wide(I, W, 0));
        } else {
            const WideRes W = wide_words(WIDE_SMOD, pop1, pop2, pop2, pop2);
            if (W.b1) { if (I.err == 0u) I.err = ZK_CODE(ZK_ZERO_DIVISION, I.seq); return; }
            if (W.b0) quotient = word_from_int(I, fr_zero());
            else EV_TRY(quotient = word_from_wide(I, W, 0));
            remainder = W.b0 ? pop1 : push;
        }
#endif
    } else {
        U256 v1, v2, vp;
        EV_TRY(v1 = int_value(I, pop1)); EV_TRY(v2 = int_value(I, pop2)); EV_TRY(vp = int_value(I, push));
        const u32 n1 = v1.v[7] >> 31, n2 = v2.v[7] >> 31, np = vp.v[7] >> 31;
        const U256 a1 = n1 ? u256_neg(v1) : v1, a2 = n2 ? u256_neg(v2) : v2, ap = np ? u256_neg(vp) : vp;
        if (is_sdiv) {
            quotient = push;
            U512 prod = u256_mul_full(ap, a2);
            U256 rem;
            const bool neg = u256_sub(rem, a1, u512_lo(prod)) || !fr_is_zero(u512_hi(prod));
            if (n1 == 0) EV_TRY(remainder = word_from_int(I, rem, neg));
            else EV_TRY(remainder = word_from_int(I, u256_neg(rem), false, neg));  // 2^256 - rem >= 2^256 when rem < 0
        } else {
            if (fr_is_zero(v2)) {
                quotient = word_from_int(I, fr_zero());
            } else {
                U256 q, r;
                u256_divmod(a1, a2, q, r);
                quotient = word_from_int(I, n1 == n2 ? q : u256_neg(q));
            }
            remainder = fr_is_zero(v2) ? pop1 : push;
        }
    }
    // check_witness (:34-76)
    u32 q_neg, d_neg, r_neg, n_neg;
    Word q_abs, d_abs, r_abs, n_abs;
    EV_TRY(q_abs = abs_word(I, quotient, q_neg));
    EV_TRY(d_abs = abs_word(I, divisor, d_neg));
    EV_TRY(r_abs = abs_word(I, remainder, r_neg));
    EV_TRY(n_abs = abs_word(I, dividend, n_neg));
    const u32 q_nz = 1 - is_zero_word(quotient), d_nz = 1 - is_zero_word(divisor), r_nz = 1 - is_zero_word(remainder);
    Fr overflow; EV_TRY(overflow = mul_add_words(I, q_abs, d_abs, r_abs, n_abs));
    constrain_zero(I, overflow);
    u32 lt, eq; compare_word(I, r_abs, d_abs, lt, eq); if (I.err) return;
    ev_require(I, lt == 1 || d_nz == 0);
    ev_require(I, n_neg == r_neg || !(q_nz && d_nz && r_nz));
    u32 so; ev_compare(I, fr_from_u128(~0ull, 0x7fffffffffffffffull), n_abs.hi, 16, so, eq); if (I.err) return;
    ev_require(I, (q_neg ^ d_neg) == n_neg || !(q_nz && d_nz && !so));
    set_tail3(T, opcode, 3, 1, 1);
}

// 512-bit / 256-bit helpers for ADDMOD / MULMOD witness values
ZK_HD void divmod_512(const U512& num, const U256& den, U512& q, U256& r) { u512_divmod(num, den, q, r, 512); }

ZK_HD void g_addmod(Ins& I, Tail& T) {  // addmod.py
    RwRows<4> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_ADDMOD));
    Word a, b, n, pushed_r;
    a = stack_pop(I, R, 0); b = stack_pop(I, R, 1); n = stack_pop(I, R, 2); pushed_r = stack_push(I, R, 3);
    const bool wide = words_wide(I, a, b, n, pushed_r);
    bool n_zero = word_is_zero_int(n);
    U256 a_red = fr_zero(), k = fr_zero(), d = fr_zero();
    Word r, kw, arw;
#if !EVM_FAST
    WideRes W;
    if (wide) {  // a % n, a // n, (a % n + b) // n on unbounded integers; every Word(int) below can raise
        W = wide_words(WIDE_ADDMOD, a, b, n, pushed_r);
        if (n_zero) EV_TRY(r = word_from_wide(I, W, 3));
        else r = pushed_r;
        EV_TRY(kw = word_from_wide(I, W, 0));
        EV_TRY(arw = word_from_wide(I, W, 1));
    } else
#endif
    {
        U256 av, bv, nv;
        EV_TRY(av = int_value(I, a)); EV_TRY(bv = int_value(I, b)); EV_TRY(nv = int_value(I, n));
        if (n_zero) {
            a_red = av;
            U256 s; u256_add(s, a_red, bv);  // (a_red + b) % 2^256
            r = word_from_int(I, s);
        } else {
            u256_divmod(av, nv, k, a_red);
            U256 s; u32 carry = u256_add(s, a_red, bv);
            U512 num = u512_from(s, fr_u(carry)), q; U256 rem;
            divmod_512(num, nv, q, rem);
            d = u512_lo(q);
            r = pushed_r;
        }
        kw = word_from_int(I, k);
        arw = word_from_int(I, a_red);
    }
    Fr overflow; EV_TRY(overflow = mul_add_words(I, kw, n, arw, a));
    constrain_zero(I, overflow);
    Word arw2, dw;
    Fr carry_hi;
    Word a_red_plus_b;
#if !EVM_FAST
    if (wide) {
        EV_TRY(arw2 = word_from_wide(I, W, 1));
        a_red_plus_b = add_words2(I, arw2, b, carry_hi);
        EV_TRY(dw = word_from_wide(I, W, 2));
    } else
#endif
    {
        arw2 = word_from_int(I, a_red);
        a_red_plus_b = add_words2(I, arw2, b, carry_hi);
        dw = word_from_int(I, d);
    }
    Word ow = n_zero ? word_from_int(I, fr_zero()) : word_checked(I, carry_hi, fr_zero());
    EV_TRY(mul_add_words_512(I, dw, n, r, ow, a_red_plus_b));
    const u32 nz = is_zero_word(n);
    u32 r_lt_n; EV_TRY(r_lt_n = lt_u256_sel(I, r, n));
    Word arw3;
#if !EVM_FAST
    if (wide) EV_TRY(arw3 = word_from_wide(I, W, 1));
    else
#endif
        arw3 = word_from_int(I, a_red);
    u32 a_lt_n; EV_TRY(a_lt_n = lt_u256_sel(I, arw3, n));
    ev_require(I, 2 == a_lt_n + r_lt_n + 2 * nz);
    // pushed_r.int_value() == FQ(r.int_value() * (1 - n_is_zero)).n  (reduction mod p, addmod.py:61)
#if !EVM_FAST
    if (wide) ev_require(I, W.b1 != 0u);
    else
#endif
    {
        U256 pv; EV_TRY(pv = int_value(I, pushed_r));
        U256 rv; EV_TRY(rv = int_value(I, r));
        U256 rhs = fr_zero();
        if (!nz) {  // r_int mod p: r_int < 2^256 < 6p
            rhs = rv;
            Fr p = fr_modulus();
            for (int t = 0; t < 6; t++) { U256 s; if (!u256_sub(s, rhs, p)) rhs = s; }
        }
        ev_require(I, fr_eq(pv, rhs));
    }
    set_tail3(T, opcode, 4, 1, 2);
}

// kw = Word(a.int_value() // n.int_value()) (0 when n == 0), made by the caller (it has a // n from the same division as a % n)
ZK_HD void mulmod_mod(Ins& I, const Word& a, const Word& n, const Word& r, const U256& a_div_n, u32 k_flags) {  // mulmod.py:6-29
    Word a_or_zero;
    U256 k = fr_zero();
    u32 kf = 0;
    if (word_is_zero_int(n)) {
        a_or_zero = word_from_int(I, fr_zero());
    } else {
        a_or_zero = a;
        k = a_div_n;  // a // n, computed once by the caller together with a % n
        kf = k_flags;
    }
    Word kw; EV_TRY(kw = word_from_int(I, k, (kf & 1u) != 0u, (kf & 2u) != 0u));
    EV_TRY(mul_add_words(I, kw, n, r, a_or_zero));
    const u32 eq = is_equal_word(a, a_or_zero);
    u32 lt, e2; compare_word(I, r, n, lt, e2);
    const u32 nz = is_zero_word(n), aoz = is_zero_word(a_or_zero);
    ev_require(I, (1 - eq) * (1 - nz * aoz) == 0);
    ev_require(I, (1 - (int)lt - (int)nz) == 0);
}

ZK_HD void g_mulmod(Ins& I, Tail& T) {  // mulmod.py
    RwRows<4> R; rw_rows_fetch(I, R);  // the stack rows, in flight together with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_MULMOD));
    Word a, b, n, r;
    a = stack_pop(I, R, 0); b = stack_pop(I, R, 1); n = stack_pop(I, R, 2); r = stack_push(I, R, 3);
    const bool wide = words_wide(I, a, b, n, r);
    U256 a_red = fr_zero(), k = fr_zero(), q0 = fr_zero();
    u32 q0_flags = 0;
    Word e, d, arw, arw2, kw;
#if !EVM_FAST
    WideRes W;
    if (wide) {  // a % n, (a % n * b) // n, the product's halves and a // n on unbounded integers; every Word(int) can raise
        W = wide_words(WIDE_MULMOD, a, b, n, r);
        EV_TRY(e = word_from_wide(I, W, 2));
        EV_TRY(d = word_from_wide(I, W, 3));
        ev_require(I, W.b1 != 0u);
        EV_TRY(arw = word_from_wide(I, W, 0));
        const WideRes Q = wide_words(WIDE_DIV, a, n, n, n);
        q0 = Q.o[0];
        q0_flags = Q.fl[0];
    } else
#endif
    {
        U256 av, bv, nv, rv;
        EV_TRY(av = int_value(I, a)); EV_TRY(bv = int_value(I, b)); EV_TRY(nv = int_value(I, n)); EV_TRY(rv = int_value(I, r));
        U512 prod;
        bool safety;
        if (fr_is_zero(nv)) {
            prod = u256_mul_full(a_red, bv);  // 0
            safety = fr_is_zero(rv);          // 0 == 0*0 + r
        } else {
            u256_divmod(av, nv, q0, a_red);
            prod = u256_mul_full(a_red, bv);
            U512 q; U256 rem; divmod_512(prod, nv, q, rem);
            k = u512_lo(q);      // k < b < 2^256
            safety = fr_eq(rem, rv);  // prod == k*n + r  <=>  r == prod mod n
        }
        e = word_from_int(I, u512_lo(prod));
        d = word_from_int(I, u512_hi(prod));
        ev_require(I, safety);
        arw = word_from_int(I, a_red);
    }
    EV_TRY(mulmod_mod(I, a, n, arw, q0, q0_flags));
#if !EVM_FAST
    if (wide) EV_TRY(arw2 = word_from_wide(I, W, 0));
    else
#endif
        arw2 = word_from_int(I, a_red);
    Word zero = word_from_int(I, fr_zero());
    EV_TRY(mul_add_words_512(I, arw2, b, zero, d, e));
#if !EVM_FAST
    if (wide) EV_TRY(kw = word_from_wide(I, W, 1));
    else
#endif
        kw = word_from_int(I, k);
    EV_TRY(mul_add_words_512(I, kw, n, r, d, e));
    const u32 nz = is_zero_word(n);
    u32 lt, eq; compare_word(I, r, n, lt, eq);
    ev_require(I, (1 - (int)lt - (int)nz) == 0);
    set_tail3(T, opcode, 4, 1, 2);
}

ZK_HD void g_memory(Ins& I, Tail& T) {  // memory.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word aw; aw = stack_pop(I);
    Fr address; EV_TRY(address = word_to_fq(I, aw, 20));
    const bool is_mload = fr_eq_u64(opcode, OP_MLOAD), is_mstore8 = fr_eq_u64(opcode, OP_MSTORE8);
    const bool is_store = !is_mload, is_not8 = !is_mstore8;
    Word value;
    if (is_mload) { value = stack_push(I); } else { value = stack_pop(I); }
    EV_TRY(to_u256(I, value));
    Fr next_size, gas;
    EV_TRY(memory_expansion(I, ev_curr(I, S_MWS), fr_add_u64(address, 1 + (is_not8 ? 31 : 0)), next_size, gas));
    if (is_mstore8) memory_lookup(I, 1, address);
    if (is_not8) memory_lookup_run32(I, is_store ? 1 : 0, address);
    if (I.err) return;
    set_tail(T, opcode, 34 - (is_mstore8 ? 31 : 0), t_delta_i(1), is_store ? 2 : 0, t_to(next_size), 0, gas);
}

ZK_HD void ctx_push_word(Ins& I, Tail& T, const Fr& opcode, const Word& w) {
    Word push; push = stack_push(I);
    constrain_equal_word(I, w, push);
    set_tail3(T, opcode, 2, 1, -1);
}
ZK_HD void g_ctx_word(Ins& I, Tail& T, u32 expected_opcode, u32 field_tag) {  // caller.py / callvalue.py / address.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(expected_opcode));
    WordOrValue v; v = call_context_lookup_word(I, field_tag);
    ctx_push_word(I, T, opcode, v.w);
}
ZK_HD void g_ctx_value(Ins& I, Tail& T, u32 expected_opcode, u32 field_tag) {  // calldatasize.py / returndatasize.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(expected_opcode));
    Fr v; v = call_context_lookup(I, field_tag);
    Word w = word_checked(I, v, fr_zero());
    ctx_push_word(I, T, opcode, w);
}
ZK_HD void g_tx_word(Ins& I, Tail& T, u32 expected_opcode, u32 tx_field_tag) {  // origin.py / gasprice.py
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(expected_opcode));
    WordOrValue v; v = tx_lookup(I, tx_id, tx_field_tag);
    ctx_push_word(I, T, opcode, v.w);
}
ZK_HD void g_selfbalance(Ins& I, Tail& T) {  // selfbalance.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_SELFBALANCE));
    WordOrValue cw; cw = call_context_lookup_word(I, CC_CalleeAddress);
    Fr callee; EV_TRY(callee = word_to_fq(I, cw.w, 20));
    RwQ Q;
    rwq_init(Q, 0, TG_Account);
    rwq_set(Q, R_ADDR, callee);
    rwq_set(Q, R_FT, fr_u(ACC_Balance));
    u32 r; r = rw_lookup(I, Q);
    Word bal = rw_word(I, r, R_VAL_LO);
    Word push; push = stack_push(I);
    constrain_equal_word(I, push, bal);
    set_tail3(T, opcode, 3, 1, -1);
}
// Common head of balance.py / extcodesize.py / extcodehash.py: opcode check, address from the
// stack, TxId, reversion info and the access-list write (instruction.py:1044-1057).
struct AccountAccess {
    Fr opcode, address;
    bool warm, warm_is_bool;
};
ZK_HD AccountAccess account_access(Ins& I, u32 expected_opcode) {
    AccountAccess A;
    A.warm = false;
    A.warm_is_bool = true;
    A.address = fr_zero();
    A.opcode = opcode_lookup(I, true);
    constrain_equal(I, A.opcode, fr_u(expected_opcode));
    Word aw; aw = stack_pop(I);
    EV_TRYV(A.address = word_to_fq(I, aw, 20), A);
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    Reversion rv; EV_TRYV(rv = reversion_info(I), A);
    RwQ W;
    rwq_init(W, 1, TG_TxAccessListAccount);
    rwq_set(W, R_ID, tx_id);
    rwq_set(W, R_ADDR, A.address);
    rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
    u32 wr; wr = state_write(I, W, rv);
    Fr is_warm; EV_TRYV(is_warm = value_of(I, rw_value_prev(I, wr)), A);
    A.warm = fr_eq_u64(is_warm, 1);
    A.warm_is_bool = fr_le_u64(is_warm, 1);  // checked by the later select
    return A;
}
ZK_HD Word account_read_word(Ins& I, const Fr& address, u32 field_tag) {  // instruction.py:957-962
    RwQ Q;
    rwq_init(Q, 0, TG_Account);
    rwq_set(Q, R_ADDR, address);
    rwq_set(Q, R_FT, fr_u(field_tag));
    u32 r; r = rw_lookup(I, Q);
    return rw_word(I, r, R_VAL_LO);
}
// the trailing `select(is_warm, 0, EXTRA_GAS_COST_ACCOUNT_COLD_ACCESS)` (asserts a boolean)
ZK_HD Fr account_access_gas(Ins& I, const AccountAccess& A) {
    ev_require(I, A.warm_is_bool);
    return fr_u(A.warm ? 0 : 2500);
}
ZK_HD void g_balance(Ins& I, Tail& T) {  // balance.py
    AccountAccess A; EV_TRY(A = account_access(I, OP_BALANCE));
    Word ch; ch = account_read_word(I, A.address, ACC_CodeHash);
    if (I.err) return;
    const u32 exists = 1 - is_zero_word(ch);
    Word bal = word_zero();
    if (exists) bal = account_read_word(I, A.address, ACC_Balance);
    else I.seq++;  // Word(0)
    I.seq += 2;  // Word(0), select_word(exists, ..)
    Word push; push = stack_push(I);
    constrain_equal_word(I, bal, push);
    Fr dyn; EV_TRY(dyn = account_access_gas(I, A));
    set_tail(T, A.opcode, 7 + (int)exists, t_delta_i(1), 0, t_same(), 0, dyn);
}
ZK_HD void g_extcodesize(Ins& I, Tail& T) {  // extcodesize.py
    AccountAccess A; EV_TRY(A = account_access(I, OP_EXTCODESIZE));
    Word ch; ch = account_read_word(I, A.address, ACC_CodeHash);
    if (I.err) return;
    const u32 exists = 1 - is_zero_word(ch);
    Fr size = fr_zero();
    if (exists) size = bytecode_length(I, ch, true);
    I.seq++;  // select(exists, code_size, 0)
    Word w; w = word_checked(I, size, fr_zero());
    Word push; push = stack_push(I);
    constrain_equal_word(I, w, push);
    Fr dyn; EV_TRY(dyn = account_access_gas(I, A));
    set_tail(T, A.opcode, 7, t_delta_i(1), 0, t_same(), 1, dyn);
}
ZK_HD void g_extcodehash(Ins& I, Tail& T) {  // extcodehash.py
    AccountAccess A; EV_TRY(A = account_access(I, OP_EXTCODEHASH));
    Word ch; ch = account_read_word(I, A.address, ACC_CodeHash);
    Word push; push = stack_push(I);
    constrain_equal_word(I, ch, push);
    Fr dyn; EV_TRY(dyn = account_access_gas(I, A));
    set_tail(T, A.opcode, 7, t_delta_i(1), 0, t_same(), 0, dyn);
}
ZK_HD void g_blockhash(Ins& I, Tail& T) {  // blockhash.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word nw; nw = stack_pop(I);
    Fr number; EV_TRY(number = word_to_fq(I, nw, 8));
    WordOrValue cur; cur = block_lookup(I, BLK_Number);
    Fr current; EV_TRY(current = value_of(I, cur));
    Word hash; hash = stack_push(I);
    u32 block_lt, diff_lt, eq;
    EV_TRY(ev_compare(I, number, current, 8, block_lt, eq));
    EV_TRY(ev_compare(I, current, fr_add_u64(number, 256), 2, diff_lt, eq));
    Word expected = word_zero();
    if (block_lt * diff_lt == 1u) {
        WordOrValue h; h = block_lookup(I, BLK_HistoryHash, &number);
        expected = h.w;
    }
    constrain_equal_word(I, hash, expected);
    set_tail3(T, opcode, 2, 1, 0);
}
ZK_HD void g_calldataload(Ins& I, Tail& T) {  // calldataload.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_CALLDATALOAD));
    Word ow; ow = stack_pop(I);
    Fr offset; EV_TRY(offset = word_to_fq(I, ow, 8));
    const bool is_root = !fr_is_zero(ev_curr(I, S_IS_ROOT));
    Fr src_id, length, cd_offset = fr_zero();
    src_id = call_context_lookup(I, is_root ? CC_TxId : CC_CallerId);
    length = call_context_lookup(I, CC_CallDataLength);
    if (!is_root) cd_offset = call_context_lookup(I, CC_CallDataOffset);
    if (I.err) return;
    const Fr src_addr = fr_add(offset, cd_offset), src_end = fr_add(length, cd_offset);
    // BufferReaderGadget (util/__init__.py:131-166) with max_bytes = bytes_left = 32: only
    // Instruction.min's 5-byte compare can fail, the bound_dist constraints are identities
    u32 lt, eq;
    EV_TRY(ev_compare(I, src_end, src_addr, 5, lt, eq));
    I.seq += 2 + 62;
    const u64 avail = lt ? 0ull : fr_lo64(src_end) - fr_lo64(src_addr);  // both < 2^40
    U256 data = fr_zero();
    bool bytes_ok = true;
    for (int k = 0; k < 32; k++) {
        if ((u64)k < avail) {
            Fr idx = fr_add_u64(src_addr, (u64)k), b;
            if (is_root) {
                WordOrValue v; v = tx_lookup(I, src_id, TXC_CallData, &idx);
                b = value_of(I, v);
            } else {
                b = memory_lookup(I, 0, idx, &src_id);
            }
            if (I.err) return;
            I.seq += 2;  // constrain_byte
            bytes_ok = bytes_ok && fr_le_u64(b, 255);
            data.v[k >> 2] |= (b.v[0] & 0xffu) << (8 * (k & 3));
        }
    }
    ev_require(I, bytes_ok, ZK_VALUE_ERROR);  // bytes(calldata_word)
    if (I.err) return;
    Word push; push = stack_push(I);
    constrain_equal_word(I, word_from_u256(data), push);
    set_tail3(T, opcode, (int)I.rw_off, 1, 0);
}
// ---- copy-table gadgets (sha3.py, codecopy.py, calldatacopy.py, returndatacopy.py, extcodecopy.py) ----
// memory_offset_and_length (instruction.py:1122-1127)
ZK_HD void memory_offset_and_length(Ins& I, const Word& offset_word, const Word& length_word, Fr& offset, Fr& length) {
    offset = fr_zero();
    length = word_to_fq(I, length_word, 5);
    if (I.err || fr_is_zero(length)) return;
    offset = word_to_fq(I, offset_word, 5);
}
// memory_expansion_dynamic_length (instruction.py:1157-1181, rd_* = None) then memory_copier_gas_cost (:1183-1192)
ZK_HD void copy_memory_gas(Ins& I, const Fr& mem_off, const Fr& length, u32 per_word, Fr& next_size, Fr& gas) {
    Fr mws = ev_curr(I, S_MWS);
    Fr cd_size = constant_divmod_shift(I, fr_add_u64(fr_add(mem_off, length), 31), 5, 4);
    u32 lt, eq;
    ev_compare(I, mws, cd_size, 4, lt, eq);
    next_size = ev_select_b(I, lt) ? cd_size : mws;
    Fr g0 = memory_gas_cost(I, mws);
    Fr g1 = memory_gas_cost(I, next_size);
    Fr words = constant_divmod_shift(I, fr_add_u64(length, 31), 5, 4);
    gas = fr_add(fr_mul_u64(words, per_word), fr_sub(g1, g0));
    range_check(I, gas, 8);
}
ZK_HD void copy_tail(Ins& I, Tail& T, const Fr& opcode, const Fr& rwc_inc, int sp_delta, const Fr& next_size, const Fr& gas) {
    set_tail(T, opcode, 0, t_delta_i(1), sp_delta, t_to(next_size), 0, gas);
    set_tail_rwc_delta(I, T, fr_add_u64(rwc_inc, I.rw_off));
}
#if defined(ZK_WARM_STAMPS) && !defined(ZK_HOSTSIM)  // tuning build: where a SHA3 step's clocks go (tools/evm_warm_timeline.py)
#define EV_STAMP2(I, k) do { if ((I).a->prof && (threadIdx.x & 63) == 0 && EV_PROF_WAVE < 1024u) (I).a->prof[(2048u + EV_PROF_WAVE) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define EV_STAMP2(I, k) do { } while (0)
#endif
ZK_HD void g_sha3(Ins& I, Tail& T) {  // sha3.py
    EV_STAMP2(I, 0);
    Fr opcode; opcode = opcode_lookup(I, true);
    EV_STAMP2(I, 1);
    Word offset, size, sha3_value;
    offset = stack_pop(I); size = stack_pop(I); sha3_value = stack_push(I);
    EV_STAMP2(I, 2);
    Fr mem_off, length; EV_TRY(memory_offset_and_length(I, offset, size, mem_off, length));
    const Word cid = word_value(I.call_id);
    CopyRes cr; cr.rwc_inc = fr_zero(); cr.rlc_acc = fr_zero();
    EV_STAMP2(I, 3);
    if (!fr_is_zero(length))
        EV_TRY(cr = copy_lookup(I, cid, CDT_Memory, cid, CDT_RlcAcc, mem_off, fr_add(mem_off, length), fr_zero(), length,
                                fr_add_u64(I.rwc, I.rw_off)));
    EV_STAMP2(I, 4);
    Word out; EV_TRY(out = keccak_lookup(I, length, cr.rlc_acc));
    EV_STAMP2(I, 5);
    constrain_equal_word(I, out, sha3_value);
    Fr next_size, gas; EV_TRY(copy_memory_gas(I, mem_off, length, 6, next_size, gas));
    EV_STAMP2(I, 6);
    copy_tail(I, T, opcode, cr.rwc_inc, 1, next_size, gas);
    EV_STAMP2(I, 7);
}
ZK_HD void g_codecopy(Ins& I, Tail& T) {  // codecopy.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word mem_w, code_w, size_w;
    mem_w = stack_pop(I); code_w = stack_pop(I); size_w = stack_pop(I);
    Fr mem_off, size; EV_TRY(memory_offset_and_length(I, mem_w, size_w, mem_off, size));
    Fr code_off; EV_TRY(code_off = word_to_fq(I, code_w, 5));
    Fr code_size; code_size = bytecode_length(I, curr_code_hash(I));
    Fr next_size, gas; EV_TRY(copy_memory_gas(I, mem_off, size, 3, next_size, gas));
    CopyRes cr; cr.rwc_inc = fr_zero();
    if (!fr_is_zero(size))
        EV_TRY(cr = copy_lookup(I, curr_code_hash(I), CDT_Bytecode, word_value(I.call_id), CDT_Memory, code_off, code_size, mem_off,
                                size, fr_add_u64(I.rwc, I.rw_off)));
    copy_tail(I, T, opcode, cr.rwc_inc, 3, next_size, gas);
}
ZK_HD void g_calldatacopy(Ins& I, Tail& T) {  // calldatacopy.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word mem_w, data_w, len_w;
    mem_w = stack_pop(I); data_w = stack_pop(I); len_w = stack_pop(I);
    Fr mem_off, length; EV_TRY(memory_offset_and_length(I, mem_w, len_w, mem_off, length));
    Fr data_off; EV_TRY(data_off = word_to_fq(I, data_w, 5));
    const bool is_root = !fr_is_zero(ev_curr(I, S_IS_ROOT));
    Fr src_id, cd_length, cd_offset = fr_zero();
    src_id = call_context_lookup(I, is_root ? CC_TxId : CC_CallerId);
    cd_length = call_context_lookup(I, CC_CallDataLength);
    if (!is_root) cd_offset = call_context_lookup(I, CC_CallDataOffset);
    if (I.err) return;
    Fr next_size, gas; EV_TRY(copy_memory_gas(I, mem_off, length, 3, next_size, gas));
    I.seq++;  // select(FQ(is_root), TxCalldata, Memory)
    CopyRes cr; cr.rwc_inc = fr_zero();
    if (!fr_is_zero(length))
        EV_TRY(cr = copy_lookup(I, word_value(src_id), is_root ? CDT_TxCalldata : CDT_Memory, word_value(I.call_id), CDT_Memory,
                                fr_add(cd_offset, data_off), fr_add(cd_offset, cd_length), mem_off, length,
                                fr_add_u64(I.rwc, I.rw_off)));
    copy_tail(I, T, opcode, cr.rwc_inc, 3, next_size, gas);
}
ZK_HD void g_returndatacopy(Ins& I, Tail& T) {  // returndatacopy.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word mem_w, off_w, size_w;
    mem_w = stack_pop(I); off_w = stack_pop(I); size_w = stack_pop(I);
    Fr last_callee, rd_length, rd_offset;
    last_callee = call_context_lookup(I, CC_LastCalleeId);
    rd_length = call_context_lookup(I, CC_LastCalleeReturnDataLength);
    rd_offset = call_context_lookup(I, CC_LastCalleeReturnDataOffset);
    if (I.err) return;
    Fr o8, s8; EV_TRY(o8 = word_to_fq(I, off_w, 8)); EV_TRY(s8 = word_to_fq(I, size_w, 8));
    range_check(I, fr_sub(rd_length, fr_add(o8, s8)), 4); if (I.err) return;
    Fr mem_off, size; EV_TRY(memory_offset_and_length(I, mem_w, size_w, mem_off, size));
    Fr next_size, gas; EV_TRY(copy_memory_gas(I, mem_off, size, 3, next_size, gas));
    CopyRes cr;
    EV_TRY(cr = copy_lookup(I, word_value(last_callee), CDT_Memory, word_value(I.call_id), CDT_Memory, rd_offset,
                            fr_add(rd_offset, size), mem_off, size, fr_add_u64(I.rwc, I.rw_off)));
    ev_require(I, fr_eq(cr.rwc_inc, fr_add(size, size))); if (I.err) return;  // plain assert (:44)
    copy_tail(I, T, opcode, cr.rwc_inc, 3, next_size, gas);
}
ZK_HD void g_extcodecopy(Ins& I, Tail& T) {  // extcodecopy.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word aw; aw = stack_pop(I);
    Fr address; EV_TRY(address = word_to_fq(I, aw, 20));
    Word mem_w, code_w, size_w;
    mem_w = stack_pop(I); code_w = stack_pop(I); size_w = stack_pop(I);
    Fr code_off; EV_TRY(code_off = word_to_fq(I, code_w, 8));
    Fr mem_off, size; EV_TRY(memory_offset_and_length(I, mem_w, size_w, mem_off, size));
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    Reversion rv; EV_TRY(rv = reversion_info(I));
    RwQ W;
    rwq_init(W, 1, TG_TxAccessListAccount);
    rwq_set(W, R_ID, tx_id);
    rwq_set(W, R_ADDR, address);
    rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
    u32 wr; wr = state_write(I, W, rv);
    Fr is_warm; EV_TRY(is_warm = value_of(I, rw_value_prev(I, wr)));
    Word ch; ch = account_read_word(I, address, ACC_CodeHash);
    if (I.err) return;
    Fr code_size = fr_zero();
    if (!is_zero_word(ch)) code_size = bytecode_length(I, ch, true);
    Fr next_size, copier; EV_TRY(copy_memory_gas(I, mem_off, size, 3, next_size, copier));
    const bool warm = ev_select(I, is_warm); if (I.err) return;
    const Fr gas = fr_add_u64(copier, warm ? 0 : 2500);
    CopyRes cr; cr.rwc_inc = fr_zero();
    if (!fr_is_zero(size))
        EV_TRY(cr = copy_lookup(I, ch, CDT_Bytecode, word_value(I.call_id), CDT_Memory, code_off, code_size, mem_off, size,
                                fr_add_u64(I.rwc, I.rw_off)));
    copy_tail(I, T, opcode, cr.rwc_inc, 4, next_size, gas);
}
ZK_HD void g_exp(Ins& I, Tail& T) {  // exp.py
    Fr opcode; opcode = opcode_lookup(I, true);
    Word base, exponent, result;
    base = stack_pop(I); exponent = stack_pop(I); result = stack_push(I);
    const bool hi0 = fr_is_zero(exponent.hi);
    if (hi0 && fr_is_zero(exponent.lo)) {
        constrain_equal(I, result.lo, fr_u(1));
        constrain_zero(I, result.hi);
    } else if (hi0 && fr_eq_u64(exponent.lo, 1)) {
        constrain_equal(I, result.lo, base.lo);
        constrain_equal(I, result.hi, base.hi);
    } else {
        Limbs64 limbs; EV_TRY(limbs = to_64s(I, base));
        const Fr identifier = fr_add_u64(I.rwc, I.rw_off);
        const Fr single_step = fr_u(hi0 && fr_eq_u64(exponent.lo, 2) ? 1 : 0);
        Word res; EV_TRY(res = exp_lookup(I, identifier, single_step, limbs.v, exponent));
        Word two; two = word_checked(I, fr_u(2), fr_zero());
        Word int_res; EV_TRY(int_res = exp_lookup(I, identifier, fr_u(1), limbs.v, two));
        Word zero; zero = word_from_int(I, fr_zero());
        EV_TRY(mul_add_words(I, base, base, zero, int_res));
        constrain_equal_word(I, res, result);
    }
    if (I.err) return;
    U256 eb; EV_TRY(eb = to_u256(I, exponent));  // byte_size (instruction.py:492-494)
    set_tail(T, opcode, 3, t_delta_i(1), 1, t_same(), 0, fr_u(50u * (u32)fr_byte_len(eb)));
}
// tx_log_lookup_word (instruction.py:708-720): address = index + (field_tag << 32) + (log_id.n << 48)
ZK_HD Fr tx_log_address(const Fr& log_id, u32 field_tag, u32 index) {
    // log_id.n << 48 as an integer, reduced mod p: multiply by 2^48 in the field
    return fr_add(fr_mul_u64(log_id, 1ull << 48), fr_u(((u64)field_tag << 32) + index));
}
ZK_HD Word tx_log_lookup_word(Ins& I, const Fr& tx_id, const Fr& log_id, u32 field_tag, u32 index) {
    I.seq++;  // Word(0) storage key
    RwQ Q;
    rwq_init(Q, 1, TG_TxLog);
    rwq_set(Q, R_ID, tx_id);
    rwq_set(Q, R_ADDR, tx_log_address(log_id, field_tag, index));
    rwq_set(Q, R_FT, fr_zero());
    rwq_set_word(Q, R_KEY_LO, word_zero());
    u32 r; r = rw_lookup(I, Q);
    return rw_word(I, r, R_VAL_LO);
}
ZK_HD void g_log(Ins& I, Tail& T) {  // log.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const Fr topics_f = fr_sub_u64(opcode, OP_LOG0);
    fixed_lookup(I, FX_Range5, topics_f, fr_zero(), fr_zero()); if (I.err) return;
    const int topic_count = (int)topics_f.v[0];
    Word w1; w1 = stack_pop(I);
    Fr mstart; EV_TRY(mstart = word_to_fq(I, w1, 8));
    Word w2; w2 = stack_pop(I);
    Fr msize; EV_TRY(msize = word_to_fq(I, w2, 8));
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    Fr is_static; is_static = call_context_lookup(I, CC_IsStatic);
    constrain_equal(I, fr_zero(), is_static);
    WordOrValue contract; contract = call_context_lookup_word(I, CC_CalleeAddress);
    Fr is_persistent; is_persistent = call_context_lookup(I, CC_IsPersistent);
    if (I.err) return;
    const Fr log_id = fr_add_u64(ev_curr(I, S_LOG), 1);
    const bool persistent_nz = !fr_is_zero(is_persistent);
    if (persistent_nz) {
        Word a; a = tx_log_lookup_word(I, tx_id, log_id, 1, 0);
        constrain_equal_word(I, contract.w, a);
    }
    for (int k = 0; k < 4; k++) {
        if (k < topic_count) {
            Word topic; topic = stack_pop(I);
            if (persistent_nz) {
                Word t; t = tx_log_lookup_word(I, tx_id, log_id, 2, (u32)k);
                constrain_equal_word(I, topic, t);
            }
        }
    }
    if (I.err) return;
    I.seq += 7;  // constrain_bool on the constant topic selectors (:70-74)
    CopyRes cr; cr.rwc_inc = fr_zero();
    if (!fr_is_zero(msize) && fr_eq_u64(is_persistent, 1))
        EV_TRY(cr = copy_lookup(I, word_value(I.call_id), CDT_Memory, word_value(tx_id), CDT_TxLog, mstart, fr_add(mstart, msize),
                                tx_log_address(log_id, 3, 0), msize, fr_add_u64(I.rwc, I.rw_off)));
    Fr next_size = ev_curr(I, S_MWS), gas;
    {
        Fr mws = next_size;
        Fr cd_size; EV_TRY(cd_size = constant_divmod_shift(I, fr_add_u64(fr_add(mstart, msize), 31), 5, 4));
        u32 lt, eq; EV_TRY(ev_compare(I, mws, cd_size, 4, lt, eq));
        next_size = ev_select_b(I, lt) ? cd_size : mws;
        Fr g0; EV_TRY(g0 = memory_gas_cost(I, mws));
        Fr g1; EV_TRY(g1 = memory_gas_cost(I, next_size));
        gas = fr_sub(g1, g0);
    }
    const Fr dyn = fr_add(fr_add_u64(fr_mul_u64(msize, 8), 375u + 375u * (u32)topic_count), gas);
    set_tail(T, opcode, 0, t_delta_i(1), 2 + topic_count, t_to(next_size), 0, dyn);
    set_tail_rwc_delta(I, T, fr_add_u64(cr.rwc_inc, I.rw_off));
    T.log_mode = fr_eq(ev_next(I, S_LOG), fr_add(ev_curr(I, S_LOG), is_persistent)) ? 1u : 2u;
}
ZK_HD void g_blockctx(Ins& I, Tail& T) {  // block_ctx.py
    Fr opcode; opcode = opcode_lookup(I, true);
    u32 tag = 0;
    if (fr_eq_u64(opcode, OP_COINBASE)) tag = BLK_Coinbase;
    else if (fr_eq_u64(opcode, OP_TIMESTAMP)) tag = BLK_Timestamp;
    else if (fr_eq_u64(opcode, OP_NUMBER)) tag = BLK_Number;
    else if (fr_eq_u64(opcode, OP_GASLIMIT)) tag = BLK_GasLimit;
    else if (fr_eq_u64(opcode, OP_PREVRANDAO)) tag = BLK_PrevRandao;
    else if (fr_eq_u64(opcode, OP_BASEFEE)) tag = BLK_BaseFee;
    else if (fr_eq_u64(opcode, OP_CHAINID)) tag = BLK_ChainId;
    ev_require(I, tag != 0, ZK_NAME_ERROR); if (I.err) return;  // `op` unbound (:24)
    WordOrValue v; v = block_lookup(I, tag);
    Word push; push = stack_push(I);
    constrain_equal_word(I, v.w, push);
    set_tail3(T, opcode, 1, 1, -1);
}
ZK_HD void g_push_lo(Ins& I, Tail& T, const Fr& opcode, const Fr& lo) {  // Word.from_lo(x) == stack_push()
    Word w = word_checked(I, lo, fr_zero());
    Word push; push = stack_push(I);
    constrain_equal_word(I, w, push);
    set_tail3(T, opcode, 1, 1, -1);
}
ZK_HD void g_gas(Ins& I, Tail& T) {  // gas.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_GAS));
    g_push_lo(I, T, opcode, fr_sub_u64(ev_curr(I, S_GAS), 2));
}
ZK_HD void g_msize(Ins& I, Tail& T) {  // msize.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const Fr mws = ev_curr(I, S_MWS);
    g_push_lo(I, T, opcode, fr_fits32(mws) ? fr_u((u64)mws.v[0] * 32ull) : fr_mulc(mws, fr_to_mont(fr_u(32))));
}
ZK_HD void g_codesize(Ins& I, Tail& T) {  // codesize.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_CODESIZE));
    Fr size; size = bytecode_length(I, curr_code_hash(I));
    g_push_lo(I, T, opcode, size);
}
ZK_HD void g_jump(Ins& I, Tail& T) {  // jump.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_JUMP));
    Word dest; dest = stack_pop(I);
    constrain_zero(I, dest.hi);
    Fr byte; byte = opcode_lookup_at(I, dest.lo, true);
    constrain_equal(I, fr_u(OP_JUMPDEST), byte);
    set_tail(T, opcode, 1, t_to(dest.lo), 1, t_same(), 0, fr_zero());
}
ZK_HD void g_jumpi(Ins& I, Tail& T) {  // jumpi.py: `if is_zero_word(cond)` is always truthy (FQ has no __bool__)
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_JUMPI));
    Word dest; dest = stack_pop(I);
    constrain_zero(I, dest.hi);
    stack_pop(I);
    set_tail3(T, opcode, 2, 1, 2);
}

ZK_HD void g_sload(Ins& I, Tail& T) {  // storage.py:15-47
    RwRows<4> A; rw_rows_fetch(I, A);          // tx id, the two reversion fields, callee address ...
    RwRows<1> P; rw_rows_fetch_at(I, P, I.rw_off + 4);  // ... and the key's stack row, in flight with the opcode record
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_SLOAD));
    Fr tx_id; tx_id = call_context_value(I, CC_TxId, A, 0);
    Reversion rv; EV_TRY(rv = reversion_info(I, A, 1));
    WordOrValue cw; cw = call_context_word(I, CC_CalleeAddress, A, 3);
    Fr callee; EV_TRY(callee = word_to_fq(I, cw.w, 20));
    Word key; key = stack_pop(I, P, 0);
    RwQ Q;
    rwq_init(Q, 0, TG_AccountStorage);
    rwq_set(Q, R_ID, tx_id);
    rwq_set(Q, R_ADDR, callee);
    rwq_set_word(Q, R_KEY_LO, key);
    u32 r; r = rw_lookup(I, Q);
    Word val = rw_word(I, r, R_VAL_LO);
    Word push; push = stack_push(I);
    constrain_equal_word(I, val, push);
    RwQ W;
    rwq_init(W, 1, TG_TxAccessListAccountStorage);
    rwq_set(W, R_ID, tx_id);
    rwq_set(W, R_ADDR, callee);
    rwq_set_word(W, R_KEY_LO, key);
    rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
    u32 wr; wr = state_write(I, W, rv);
    Fr is_warm; EV_TRY(is_warm = value_of(I, rw_value_prev(I, wr)));
    bool warm = ev_select(I, is_warm); if (I.err) return;
    set_tail(T, opcode, 8, t_delta_i(1), 0, t_same(), 1, fr_u(warm ? 100 : 2100));
}

ZK_HD void g_sstore(Ins& I, Tail& T) {  // storage.py:50-153
    RwRows<4> A; rw_rows_fetch(I, A);                   // tx id, is_static, the two reversion fields ...
    RwRows<3> B; rw_rows_fetch_at(I, B, I.rw_off + 4);  // ... callee address, the two stack rows
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_SSTORE));
    Fr tx_id; tx_id = call_context_value(I, CC_TxId, A, 0);
    Fr is_static; is_static = call_context_value(I, CC_IsStatic, A, 1);
    constrain_equal(I, fr_zero(), is_static);
    Reversion rv; EV_TRY(rv = reversion_info(I, A, 2));
    WordOrValue cw; cw = call_context_word(I, CC_CalleeAddress, B, 0);
    Fr callee; EV_TRY(callee = word_to_fq(I, cw.w, 20));
    Word key, sval;
    key = stack_pop(I, B, 1); sval = stack_pop(I, B, 2);
    RwQ Q;
    rwq_init(Q, 1, TG_AccountStorage);
    rwq_set(Q, R_ID, tx_id);
    rwq_set(Q, R_ADDR, callee);
    rwq_set_word(Q, R_KEY_LO, key);
    u32 r; r = state_write(I, Q, rv);
    Word value = rw_word(I, r, R_VAL_LO), value_prev = rw_word(I, r, R_PREV_LO), original = rw_word(I, r, R_AUX_LO);
    constrain_equal_word(I, sval, value);
    RwQ W;
    rwq_init(W, 1, TG_TxAccessListAccountStorage);
    rwq_set(W, R_ID, tx_id);
    rwq_set(W, R_ADDR, callee);
    rwq_set_word(W, R_KEY_LO, key);
    rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
    u32 wr; wr = state_write(I, W, rv);
    Fr is_warm; EV_TRY(is_warm = value_of(I, rw_value_prev(I, wr)));
    RwQ Rf;
    rwq_init(Rf, 1, TG_TxRefund);
    rwq_set(Rf, R_ID, tx_id);
    u32 rr; rr = state_write(I, Rf, rv);
    Fr gas_refund; EV_TRY(gas_refund = value_of(I, rw_value(I, rr)));
    Fr refund_prev; EV_TRY(refund_prev = value_of(I, rw_value_prev(I, rr)));
    const u64 CLEARS = 4800, SET = 20000, RESET = 2900, SLOAD = 100;
    // eager nested selects in the reference's evaluation order (storage.py:84-123)
    const u32 vz = is_zero_word(value), pz = is_zero_word(value_prev), oz = is_zero_word(original);
    const u32 o_eq_v = is_equal_word(original, value), o_eq_p = is_equal_word(original, value_prev);
    const u32 p_eq_v = is_equal_word(value_prev, value), p_eq_o = is_equal_word(value_prev, original);
    Fr inner = ev_select_b(I, vz) ? fr_add_u64(refund_prev, CLEARS) : refund_prev;
    Fr nz_allne = ev_select_b(I, pz) ? fr_sub_u64(refund_prev, CLEARS) : inner;
    Fr nz_ne_ne = ev_select_b(I, 1 - o_eq_v) ? nz_allne : fr_sub_u64(fr_add_u64(nz_allne, RESET), SLOAD);
    Fr inner2 = ev_select_b(I, o_eq_v) ? fr_sub_u64(fr_add_u64(refund_prev, SET), SLOAD) : refund_prev;
    Fr ne_ne = ev_select_b(I, 1 - oz) ? nz_ne_ne : inner2;
    Fr inner3 = ev_select_b(I, (1 - oz) * vz) ? fr_add_u64(refund_prev, CLEARS) : refund_prev;
    Fr inner4 = ev_select_b(I, o_eq_p) ? inner3 : ne_ne;
    Fr refund_new = ev_select_b(I, p_eq_v) ? refund_prev : inner4;
    constrain_equal(I, gas_refund, refund_new);
    const u32 eq_prev = p_eq_v, prev_ne_orig = 1 - p_eq_o;
    const u64 inner5 = ev_select_b(I, oz) ? SET : RESET;
    const u64 warm_case = ev_select_b(I, eq_prev + prev_ne_orig - eq_prev * prev_ne_orig) ? SLOAD : inner5;
    bool warm = ev_select(I, is_warm); if (I.err) return;
    set_tail(T, opcode, 10, t_delta_i(1), 2, t_same(), 3, fr_u(warm ? warm_case : warm_case + 2100));
}

// step_state_transition_to_restored_context (instruction.py:292-363), caller_id=None form.
// Its eleven (twelve with the caller-id read) call-context lookups sit at consecutive rw_counters: their rows are requested
// in batches of four, two batches in flight, instead of one dependent round trip (or two) per lookup — a non-root STOP
// wavefront was 145k cycles, the longest of the hot kernel.  Same checks, same checkpoints.
ZK_HD RwRow rc_row(const RwRows<4>& A, const RwRows<4>& B, const RwRows<4>& C, int j) {
    switch (j) {
    case 0: return A.row[0]; case 1: return A.row[1]; case 2: return A.row[2]; case 3: return A.row[3];
    case 4: return B.row[0]; case 5: return B.row[1]; case 6: return B.row[2]; case 7: return B.row[3];
    case 8: return C.row[0]; case 9: return C.row[1]; case 10: return C.row[2]; default: return C.row[3];
    }
}
ZK_HD void restore_context(Ins& I, const Fr& rw_counter_delta_in, const Fr& gas_left, const Fr& rd_offset = fr_zero(),
                           const Fr& rd_length = fr_zero(), const Fr* caller_id_in = nullptr) {
    const Fr rw_counter_delta = fr_add_u64(rw_counter_delta_in, caller_id_in ? 11 : 12);
    const int o = caller_id_in ? 0 : 1;  // the lookups of this transition: [caller id], 8 saved fields, 3 last-callee writes
    const u64 base = I.rw_off;
    RwRows<4> A, B, C;
    rw_rows_fetch_at(I, A, base);
    rw_rows_fetch_at(I, B, base + 4);
    C.valid = false;
    const bool rows = A.valid;
    Fr caller_id;
    if (caller_id_in) caller_id = *caller_id_in;
    else caller_id = rows ? value_of(I, call_context_lookup_word_row(I, CC_CallerId, 0, nullptr, A.row[0])) : call_context_lookup(I, CC_CallerId);
    const u32 tags[8] = {CC_IsRoot, CC_IsCreate, CC_CodeHash, CC_ProgramCounter, CC_StackPointer, CC_GasLeft,
                         CC_MemorySize, CC_ReversibleWriteCounter};
    WordOrValue saved[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (rows && o + k == 4) rw_rows_fetch_at(I, C, base + 8);  // the third batch, once the first is consumed
        if (rows) EV_TRY(saved[k] = call_context_lookup_word_row(I, tags[k], 0, &caller_id, rc_row(A, B, C, o + k)));
        else EV_TRY(saved[k] = call_context_lookup_word(I, tags[k], 0, &caller_id));
    }
    {
        const u32 ltags[3] = {CC_LastCalleeId, CC_LastCalleeReturnDataOffset, CC_LastCalleeReturnDataLength};
        const Fr want[3] = {ev_curr(I, S_CALL_ID), rd_offset, rd_length};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Fr v;
            if (rows) v = value_of(I, call_context_lookup_word_row(I, ltags[k], 1, &caller_id, rc_row(A, B, C, o + 8 + k)));
            else v = call_context_lookup(I, ltags[k], 1, &caller_id);
            constrain_equal(I, v, want[k]);
        }
    }
    const u32 st = ev_curr(I, S_STATE).v[0];
    const bool halts_ok = st == ES_STOP || st == ES_RETURN || st == ES_SELFDESTRUCT;
    Fr rev = halts_ok ? ev_curr(I, S_REV) : fr_zero();
    Fr is_root = value_of(I, saved[0]);
    Fr is_create = value_of(I, saved[1]);
    Fr pc = value_of(I, saved[3]);
    Fr sp = value_of(I, saved[4]);
    Fr gas = value_of(I, saved[5]);
    Fr mem = value_of(I, saved[6]);
    Fr rwc = value_of(I, saved[7]);
    transition(I, S_RWC, t_delta(rw_counter_delta));
    transition(I, S_CALL_ID, t_to(caller_id));
    transition(I, S_IS_ROOT, t_to(is_root));
    transition(I, S_IS_CREATE, t_to(is_create));
    ev_require(I, fr_eq(ev_next(I, S_CH_LO), saved[2].w.lo) && fr_eq(ev_next(I, S_CH_HI), saved[2].w.hi));
    transition(I, S_PC, t_to(pc));
    transition(I, S_SP, t_to(sp));
    transition(I, S_GAS, t_to(fr_add(gas, gas_left)));
    transition(I, S_MWS, t_to(mem));
    transition(I, S_REV, t_to(fr_add(rwc, rev)));
}

ZK_HD void g_stop(Ins& I, Tail& T) {  // stop.py
    Fr code_length; code_length = bytecode_length(I, curr_code_hash(I));
    u32 lt, eq; ev_compare(I, code_length, ev_curr(I, S_PC), 8, lt, eq); if (I.err) return;
    if (lt + eq == 0) {
        Fr opcode; opcode = opcode_lookup(I, true);
        fixed_lookup(I, FX_ResponsibleOpcode, ev_curr(I, S_STATE), opcode, fr_zero()); if (I.err) return;
    }
    Fr is_success; is_success = call_context_lookup(I, CC_IsSuccess);
    constrain_equal(I, is_success, fr_u(1));
    const u32 to_end_tx = ev_next(I, S_STATE).v[0] == ES_EndTx ? 1u : 0u;
    Fr is_root = ev_curr(I, S_IS_ROOT);
    constrain_equal(I, is_root, fr_u(to_end_tx));
    if (I.err) return;
    if (!fr_is_zero(is_root)) {
        transition(I, S_RWC, t_delta_i(1));
        transition(I, S_CALL_ID, t_same());
    } else {
        restore_context(I, fr_u(1), ev_curr(I, S_GAS));
    }
}

// constrain_error_state (instruction.py:1426-1452)
ZK_HD void constrain_error_state(Ins& I) {
    const Fr delta = fr_add_u64(fr_add_u64(ev_curr(I, S_REV), I.rw_off), 1);  // rw_counter_offset + reversible_write_counter + 1
    Fr is_success; is_success = call_context_lookup(I, CC_IsSuccess);
    constrain_equal(I, is_success, fr_zero());
    const u32 to_end_tx = ev_next(I, S_STATE).v[0] == ES_EndTx ? 1u : 0u;
    Fr is_root = ev_curr(I, S_IS_ROOT);
    constrain_equal(I, is_root, fr_u(to_end_tx));
    if (I.err) return;
    if (!fr_is_zero(is_root)) {
        transition(I, S_RWC, t_delta(delta));
        transition(I, S_CALL_ID, t_same());
    } else {
        restore_context(I, delta, fr_zero());
    }
}
// the out-of-gas states end with compare(gas_left, cost, N_BYTES_GAS), constrain_equal(lt, 1) and
// constrain_error_state: evaluated once after the gadget switch (error_tail)
ZK_HD void oog_tail(Tail& T, const Fr& gas_cost) {
    T.dyn_gas = gas_cost;
    T.err_tail = 2;
}
ZK_HD void error_tail(Ins& I, const Tail& T) {
    if (T.err_tail == 2u) {
        u32 lt, eq; ev_compare(I, ev_curr(I, S_GAS), T.dyn_gas, 8, lt, eq); if (I.err) return;
        ev_require(I, lt == 1u); if (I.err) return;
    }
    constrain_error_state(I);
}
ZK_HD void g_error_invalid_opcode(Ins& I, Tail& T) {  // error_invalid_opcode.py
    Fr opcode; opcode = opcode_lookup(I, true);
    fixed_lookup(I, FX_ResponsibleOpcode, ev_curr(I, S_STATE), opcode, fr_zero()); if (I.err) return;
    T.err_tail = 1;
}
ZK_HD void g_error_stack(Ins& I, Tail& T) {  // error_stack.py
    Fr opcode; opcode = opcode_lookup(I, true);
    fixed_lookup(I, FX_ResponsibleOpcode, ev_curr(I, S_STATE), opcode, I.sp); if (I.err) return;
    T.err_tail = 1;
}
ZK_HD void g_error_oog_constant(Ins& I, Tail& T) {  // error_oog_constant.py
    Fr opcode; opcode = opcode_lookup(I, true);
    static const uint8_t valid[256] = ZK_OPCODE_VALID_INIT;
    static const uint16_t cgas[256] = ZK_OPCODE_CONST_GAS_INIT;
    const bool op_ok = fr_le_u64(opcode, 255) && valid[opcode.v[0] & 0xff];
    ev_require(I, op_ok, ZK_VALUE_ERROR);  // Opcode(opcode.n)
    if (I.err) return;
    const Fr gas = fr_u(cgas[opcode.v[0] & 0xff]);
    fixed_lookup(I, FX_OpcodeConstantGas, opcode, gas, fr_zero()); if (I.err) return;
    oog_tail(T, gas);
}
ZK_HD void g_error_invalid_jump(Ins& I, Tail& T) {  // error_invalid_jump.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_jumpi = fr_eq_u64(opcode, OP_JUMPI);
    ev_require(I, is_jumpi || fr_eq_u64(opcode, OP_JUMP)); if (I.err) return;
    Fr code_length; code_length = bytecode_length(I, curr_code_hash(I));
    Word dest; dest = stack_pop(I);
    if (is_jumpi) {
        Word cond; cond = stack_pop(I);
        ev_require(I, !fr_is_zero(cond.lo) || !fr_is_zero(cond.hi));
    }
    if (I.err) return;
    Fr dest_value; EV_TRY(dest_value = word_to_fq(I, dest, 8));
    u32 within, eq; ev_compare(I, dest_value, code_length, 8, within, eq); if (I.err) return;
    if (within == 1u) {  // out-of-range destinations get no further constraint (:25-33)
        u32 r; r = bytecode_lookup(I, curr_code_hash(I), 2, dest_value, -1); if (I.err) return;
        const bool is_code = !fr_is_zero(zk_table_cell(I.a->bytecode, r, B_IS_CODE));
        const bool is_dest = fr_eq_u64(zk_table_cell(I.a->bytecode, r, B_VALUE), OP_JUMPDEST);
        ev_require(I, !(is_code && is_dest)); if (I.err) return;
        T.err_tail = 1;
    }
}

ZK_HD Fr read_account_to_access_list(Ins& I, const Fr& tx_id, const Fr& address) {  // instruction.py:1059-1069
    RwQ Q;
    rwq_init(Q, 0, TG_TxAccessListAccount);
    rwq_set(Q, R_ID, tx_id);
    rwq_set(Q, R_ADDR, address);
    u32 r; r = rw_lookup(I, Q);
    return value_of(I, rw_value_prev(I, r));
}
// memory_expansion_dynamic_length's gas only (instruction.py:1157-1181)
ZK_HD Fr dyn_expansion_gas(Ins& I, const Fr& offset, const Fr& length) {
    Fr mws = ev_curr(I, S_MWS);
    Fr cd_size = constant_divmod_shift(I, fr_add_u64(fr_add(offset, length), 31), 5, 4);
    u32 lt, eq;
    ev_compare(I, mws, cd_size, 4, lt, eq);
    Fr next_size = ev_select_b(I, lt) ? cd_size : mws;
    Fr g0 = memory_gas_cost(I, mws);
    Fr g1 = memory_gas_cost(I, next_size);
    return fr_sub(g1, g0);
}
ZK_HD void g_error_oog_static_memory(Ins& I, Tail& T) {  // error_oog_static_memory_expansion.py
    Fr opcode; opcode = opcode_lookup(I, true);
    ev_require(I, fr_eq_u64(opcode, OP_MLOAD) || fr_eq_u64(opcode, OP_MSTORE) || fr_eq_u64(opcode, OP_MSTORE8)); if (I.err) return;
    Word ow; ow = stack_pop(I);
    Fr offset; EV_TRY(offset = word_to_fq(I, ow, 5));
    // `size = 1 if is_mstore8 else 32` (:21): an FQ is always truthy, so the size is 1 for all three opcodes
    Fr gas; EV_TRY(gas = dyn_expansion_gas(I, offset, fr_u(1)));
    oog_tail(T, fr_add_u64(gas, 3));
}
ZK_HD void g_error_oog_dynamic_memory(Ins& I, Tail& T) {  // error_oog_dynamic_memory_expansion.py
    Fr opcode; opcode = opcode_lookup(I, true);
    ev_require(I, fr_eq_u64(opcode, OP_RETURN) || fr_eq_u64(opcode, OP_REVERT)); if (I.err) return;
    Word ow, sw; ow = stack_pop(I); sw = stack_pop(I);
    Fr offset, size; EV_TRY(memory_offset_and_length(I, ow, sw, offset, size));
    Fr next_size, gas; EV_TRY(memory_expansion(I, offset, size, next_size, gas));
    oog_tail(T, gas);
}
ZK_HD void g_error_oog_memory_copy(Ins& I, Tail& T) {  // error_oog_memory_copy.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_ext = fr_eq_u64(opcode, OP_EXTCODECOPY);
    ev_require(I, is_ext || fr_eq_u64(opcode, OP_CALLDATACOPY) || fr_eq_u64(opcode, OP_CODECOPY) || fr_eq_u64(opcode, OP_RETURNDATACOPY));
    if (I.err) return;
    Word ext_addr = word_zero();
    int off = 0;
    if (is_ext) { ext_addr = stack_lookup(I, 0, 0); off = 1; }
    Word mem_w, size_w;
    mem_w = stack_lookup(I, 0, off); size_w = stack_lookup(I, 0, off + 2);
    if (I.err) return;
    u64 constant_gas = 3;
    if (is_ext) {
        Fr address; EV_TRY(address = word_to_fq(I, ext_addr, 5));  // N_BYTES_MEMORY_ADDRESS, as written (:41)
        Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
        Fr is_warm; EV_TRY(is_warm = read_account_to_access_list(I, tx_id, address));
        constant_gas = fr_eq_u64(is_warm, 1) ? 100 : 2600;
    }
    Fr mem_off, size; EV_TRY(memory_offset_and_length(I, mem_w, size_w, mem_off, size));
    Fr next_size, dyn; EV_TRY(copy_memory_gas(I, mem_off, size, 3, next_size, dyn));
    oog_tail(T, fr_add_u64(dyn, constant_gas));
}
ZK_HD void g_error_oog_account_access(Ins& I, Tail& T) {  // error_oog_account_access.py
    Fr opcode; opcode = opcode_lookup(I, true);
    ev_require(I, fr_eq_u64(opcode, OP_BALANCE) || fr_eq_u64(opcode, OP_EXTCODESIZE) || fr_eq_u64(opcode, OP_EXTCODEHASH)); if (I.err) return;
    Word aw; aw = stack_pop(I);
    Fr address; EV_TRY(address = word_to_fq(I, aw, 20));
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    Fr is_warm; EV_TRY(is_warm = read_account_to_access_list(I, tx_id, address));
    oog_tail(T, fr_u(fr_eq_u64(is_warm, 1) ? 100 : 2600));
}
ZK_HD void g_error_oog_log(Ins& I, Tail& T) {  // error_oog_log.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const Fr topics = fr_sub_u64(opcode, OP_LOG0);
    fixed_lookup(I, FX_Range5, topics, fr_zero(), fr_zero()); if (I.err) return;
    Word w1, w2; w1 = stack_pop(I);
    Fr mstart; EV_TRY(mstart = word_to_fq(I, w1, 5));
    w2 = stack_pop(I);
    Fr msize; EV_TRY(msize = word_to_fq(I, w2, 5));
    Fr gas; EV_TRY(gas = dyn_expansion_gas(I, mstart, msize));
    oog_tail(T, fr_add(fr_add_u64(fr_mul_u64(topics, 375), 375), fr_add(fr_mul_u64(msize, 8), gas)));
}
ZK_HD void g_error_oog_exp(Ins& I, Tail& T) {  // error_oog_exp.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_EXP));
    Word exponent; exponent = stack_lookup(I, 0, 1);
    U256 eb; EV_TRY(eb = to_u256(I, exponent));
    oog_tail(T, fr_u(50u * (u32)fr_byte_len(eb) + 10u));
}
ZK_HD void g_error_oog_sha3(Ins& I, Tail& T) {  // error_oog_sha3.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_SHA3));
    Word ow, sw; ow = stack_pop(I); sw = stack_pop(I);
    Fr mem_off, size; EV_TRY(memory_offset_and_length(I, ow, sw, mem_off, size));
    Fr gas; EV_TRY(gas = dyn_expansion_gas(I, mem_off, size));
    Fr words; EV_TRY(words = constant_divmod_shift(I, fr_add_u64(size, 31), 5, 4));
    oog_tail(T, fr_add_u64(fr_add(fr_mul_u64(words, 6), gas), 30));
}
ZK_HD void g_error_return_data_oob(Ins& I, Tail& T) {  // error_return_data_out_of_bound.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_RETURNDATACOPY));
    Word w1; w1 = stack_lookup(I, 0, 1);
    Fr data_offset; EV_TRY(data_offset = word_to_fq(I, w1, 31));
    Word w2; w2 = stack_lookup(I, 0, 2);
    Fr length; EV_TRY(length = word_to_fq(I, w2, 31));
    Fr rd_len; rd_len = call_context_lookup(I, CC_LastCalleeReturnDataLength);
    const Fr end = fr_add(data_offset, length);
    u32 over, eq; EV_TRY(ev_compare(I, rd_len, end, 31, over, eq));
    ev_require(I, !fr_fits64(data_offset) || !fr_fits64(end) || over != 0u); if (I.err) return;
    T.err_tail = 1;
}
ZK_HD void g_error_write_protection(Ins& I, Tail& T) {  // error_write_protection.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const u32 y = opcode.v[0];
    const bool ok = fr_le_u64(opcode, 255) && (y == OP_SSTORE || (y >= OP_LOG0 && y <= OP_LOG4) || y == OP_CREATE || y == OP_CALL ||
                                               y == OP_CREATE2 || y == OP_SELFDESTRUCT);
    ev_require(I, ok); if (I.err) return;
    Fr is_static; is_static = call_context_lookup(I, CC_IsStatic);
    constrain_equal(I, is_static, fr_u(1));
    if (y == OP_CALL) {
        Word value; value = stack_lookup(I, 0, 2);
        ev_require(I, !fr_is_zero(value.lo) || !fr_is_zero(value.hi));
    }
    if (I.err) return;
    T.err_tail = 1;
}

ZK_HD void g_return(Ins& I, Tail& T) {  // return_revert.py (`not is_return` never holds for an FQ; REVERT is not dispatched)
    Fr opcode; opcode = opcode_lookup(I, true);
    const u32 is_return = fr_eq_u64(opcode, OP_RETURN) ? 1u : 0u;
    Fr is_success; is_success = call_context_lookup(I, CC_IsSuccess);
    constrain_equal(I, is_success, fr_u(is_return));
    Word off_w, len_w; off_w = stack_pop(I); len_w = stack_pop(I);
    Fr ret_off; EV_TRY(ret_off = word_to_fq(I, off_w, 5));
    Fr ret_len; EV_TRY(ret_len = word_to_fq(I, len_w, 5));
    const Fr ret_end = fr_add(ret_off, ret_len);
    Fr rwc_delta = fr_u(3);
    Fr gas_left = ev_curr(I, S_GAS);
    const bool is_root = !fr_is_zero(ev_curr(I, S_IS_ROOT)), is_create = !fr_is_zero(ev_curr(I, S_IS_CREATE));
    if (is_create) {  // `curr.is_create and is_success`: an FQ is always truthy
        WordOrValue cw; cw = call_context_lookup_word(I, CC_CalleeAddress);
        Fr callee; EV_TRY(callee = word_to_fq(I, cw.w, 20));
        RwQ Q;
        rwq_init(Q, 1, TG_Account);
        rwq_set(Q, R_ADDR, callee);
        rwq_set(Q, R_FT, fr_u(ACC_CodeHash));
        u32 r; r = rw_lookup(I, Q); if (I.err) return;
        const Word code_hash = rw_word(I, r, R_VAL_LO), code_hash_prev = rw_word(I, r, R_PREV_LO);
        I.seq++;  // Word(EMPTY_HASH)
        constrain_equal_word(I, code_hash_prev, word_of(fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull),
                                                        fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull)));
        constrain_equal_word(I, code_hash, curr_code_hash(I));
        fixed_lookup(I, FX_Range24_576, ret_len, fr_zero(), fr_zero()); if (I.err) return;
        gas_left = fr_sub(gas_left, fr_mul_u64(ret_len, 200));
        if (!fr_is_zero(ret_len)) {
            CopyRes cr;
            EV_TRY(cr = copy_lookup(I, word_value(I.call_id), CDT_Memory, code_hash, CDT_Bytecode, ret_off, ret_end, fr_zero(), ret_len,
                                    fr_add_u64(I.rwc, I.rw_off)));
            constrain_equal(I, cr.rwc_inc, ret_len); if (I.err) return;
            I.rw_off += fr_lo64(ret_len);  // < 2^40
            rwc_delta = fr_add(rwc_delta, ret_len);
            Fr code_size; code_size = bytecode_length(I, code_hash, true);
            constrain_equal(I, code_size, ret_len); if (I.err) return;
        }
    }
    if (!is_root && !is_create) {
        Fr caller_off, caller_len;
        caller_off = call_context_lookup(I, CC_ReturnDataOffset);
        caller_len = call_context_lookup(I, CC_ReturnDataLength);
        if (I.err) return;
        u32 lt, eq; EV_TRY(ev_compare(I, ret_len, caller_len, 5, lt, eq));
        const Fr copy_len = ev_select_b(I, lt) ? ret_len : caller_len;
        CopyRes cr;
        EV_TRY(cr = copy_lookup(I, word_value(I.call_id), CDT_Memory, word_value(ev_next(I, S_CALL_ID)), CDT_Memory, ret_off, ret_end,
                                caller_off, copy_len, fr_add_u64(I.rwc, I.rw_off)));
        const Fr twice = fr_add(copy_len, copy_len);
        constrain_equal(I, cr.rwc_inc, twice); if (I.err) return;
        I.rw_off += fr_lo64(twice);
        rwc_delta = fr_add(fr_add_u64(rwc_delta, 2), twice);
    }
    const u32 to_end_tx = ev_next(I, S_STATE).v[0] == ES_EndTx ? 1u : 0u;
    constrain_equal(I, fr_u(is_root ? 1 : 0), fr_u(to_end_tx)); if (I.err) return;
    Fr exp_gas; EV_TRY(exp_gas = dyn_expansion_gas(I, ret_off, ret_len));
    if (is_root) {
        Fr is_persistent; is_persistent = call_context_lookup(I, CC_IsPersistent);
        constrain_equal(I, is_persistent, fr_u(is_return)); if (I.err) return;
        transition(I, S_RWC, t_delta(fr_add_u64(rwc_delta, 1)));
        transition(I, S_GAS, t_to(gas_left));
        transition(I, S_CALL_ID, t_same());
    } else {
        restore_context(I, rwc_delta, fr_sub(gas_left, exp_gas), ret_off, ret_len);
    }
}
ZK_HD void g_error_invalid_creation_code(Ins& I, Tail& T) {  // error_invalid_creation_code.py
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_RETURN));
    ev_require(I, !fr_is_zero(ev_curr(I, S_IS_CREATE))); if (I.err) return;
    Word ow; ow = stack_pop(I);
    Fr ret_off; EV_TRY(ret_off = word_to_fq(I, ow, 5));
    Fr first; first = memory_lookup(I, 0, ret_off);
    constrain_equal(I, first, fr_u(0xEF)); if (I.err) return;
    T.err_tail = 1;
}
ZK_HD void g_error_code_store(Ins& I, Tail& T) {  // error_code_store.py (ErrorMaxCodeSizeExceeded, ErrorOutOfGasCodeStore)
    Fr opcode; opcode = opcode_lookup(I, true);
    constrain_equal(I, opcode, fr_u(OP_RETURN));
    ev_require(I, !fr_is_zero(ev_curr(I, S_IS_CREATE))); if (I.err) return;
    Word lw; lw = stack_lookup(I, 0, 1);
    Fr ret_len; EV_TRY(ret_len = word_to_fq(I, lw, 5));
    Fr is_static; is_static = call_context_lookup(I, CC_IsStatic);
    constrain_equal(I, is_static, fr_zero()); if (I.err) return;
    u32 over, insufficient, eq;
    EV_TRY(ev_compare(I, fr_u(24576), ret_len, 2, over, eq));
    EV_TRY(ev_compare(I, ev_curr(I, S_GAS), fr_mul_u64(ret_len, 200), 8, insufficient, eq));
    ev_require(I, (insufficient | over) != 0u); if (I.err) return;
    T.err_tail = 1;
}

// ---- EndTx (end_tx.py) ---------------------------------------------------------------------------
// split a canonical field element at bit 128: value = hi * 2^128 + lo
ZK_HD void split128(const Fr& x, Fr& lo, Fr& hi) {
    lo = x; hi = fr_zero();
    for (int k = 4; k < 8; k++) { hi.v[k - 4] = x.v[k]; lo.v[k] = 0; }
}
// mul_word_by_u64 (instruction.py:587-597): the two products are taken in the field, then split
ZK_HD Word mul_word_by_u64(Ins& I, const Word& w, const Fr& m) {
    Fr p_lo, q_lo, p_hi, q_hi;
    split128(fr_mul(w.lo, m), p_lo, q_lo);
    split128(fr_add(fr_mul(w.hi, m), q_lo), p_hi, q_hi);
    constrain_zero(I, q_hi);
    return word_checked(I, p_lo, p_hi);
}
// sub_word (instruction.py:576-585)
ZK_HD Word sub_word(Ins& I, const Word& a, const Word& b) {
    const u32 borrow_lo = fr_lt(a.lo, b.lo) ? 1u : 0u;
    Fr two128 = fr_zero(); two128.v[4] = 1u;
    Fr diff_lo = fr_sub(a.lo, b.lo);
    if (borrow_lo) diff_lo = fr_add(diff_lo, two128);
    // minuend_hi.n < subtrahend_hi.n + borrow_lo on the integers (b.hi + 1 cannot wrap: b.hi < p)
    Fr bh; const u32 carry = u256_add(bh, b.hi, fr_u(borrow_lo));
    const u32 borrow_hi = (carry || fr_lt(a.hi, bh)) ? 1u : 0u;
    Fr diff_hi = fr_sub(fr_sub(a.hi, b.hi), fr_u(borrow_lo));
    if (borrow_hi) diff_hi = fr_add(diff_hi, two128);
    return word_checked(I, diff_lo, diff_hi);
}
// add_balance (instruction.py:987-999) with one addend and no reversion info
ZK_HD void add_balance(Ins& I, const Fr& address, const Word& value) {
    RwQ Q;
    rwq_init(Q, 1, TG_Account);
    rwq_set(Q, R_ADDR, address);
    rwq_set(Q, R_FT, fr_u(ACC_Balance));
    u32 r; r = rw_lookup(I, Q); if (I.err) return;
    const Word balance = rw_word(I, r, R_VAL_LO), balance_prev = rw_word(I, r, R_PREV_LO);
    Fr carry; Word sum; sum = add_words2(I, balance_prev, value, carry);
    constrain_equal_word(I, balance, sum);
    constrain_zero(I, carry);
}
ZK_HD Fr tx_receipt(Ins& I, u32 rw, const Fr& tx_id, u32 field_tag) {  // instruction.py:723-754
    I.seq++;  // Word(0) storage key
    RwQ Q;
    rwq_init(Q, rw, TG_TxReceipt);
    rwq_set(Q, R_ID, tx_id);
    rwq_set(Q, R_ADDR, fr_zero());
    rwq_set(Q, R_FT, fr_u(field_tag));
    rwq_set_word(Q, R_KEY_LO, word_zero());
    u32 r; r = rw_lookup(I, Q);
    return value_of(I, rw_value(I, r));
}
ZK_HD void g_end_tx(Ins& I, Tail& T) {
    Fr tx_id, is_persistent;
    tx_id = call_context_lookup(I, CC_TxId);
    is_persistent = call_context_lookup(I, CC_IsPersistent);
    if (I.err) return;
    WordOrValue v; v = tx_lookup(I, tx_id, TXC_TxInvalid);
    Fr is_tx_invalid; EV_TRY(is_tx_invalid = value_of(I, v));
    v = tx_lookup(I, tx_id, TXC_Gas);
    Fr tx_gas; EV_TRY(tx_gas = value_of(I, v));
    const Fr gas_left = ev_curr(I, S_GAS);
    const Fr gas_used = fr_sub(tx_gas, gas_left);
    Fr max_refund;
    {
        U256 q, r; u256_divmod(gas_used, fr_u(5), q, r);  // constant_divmod(gas_used, 5, N_BYTES_GAS)
        max_refund = q;
        range_check(I, max_refund, 8); if (I.err) return;
    }
    Fr refund;
    {
        RwQ Q;
        rwq_init(Q, 0, TG_TxRefund);
        rwq_set(Q, R_ID, tx_id);
        u32 r; r = rw_lookup(I, Q);
        EV_TRY(refund = value_of(I, rw_value(I, r)));
    }
    u32 lt, eq; EV_TRY(ev_compare(I, max_refund, refund, 8, lt, eq));
    const Fr effective_refund = ev_select_b(I, lt) ? max_refund : refund;
    const bool invalid = fr_eq_u64(is_tx_invalid, 1);
    if (invalid) constrain_zero(I, effective_refund);
    v = tx_lookup(I, tx_id, TXC_GasPrice); if (I.err) return;
    const Word gas_price = v.w;
    Word value; EV_TRY(value = mul_word_by_u64(I, gas_price, fr_add(gas_left, effective_refund)));
    v = tx_lookup(I, tx_id, TXC_CallerAddress);
    Fr caller; EV_TRY(caller = word_to_fq(I, v.w, 20));
    EV_TRY(add_balance(I, caller, value));
    WordOrValue bf; bf = block_lookup(I, BLK_BaseFee); if (I.err) return;
    Word tip; EV_TRY(tip = sub_word(I, gas_price, bf.w));
    Word reward; EV_TRY(reward = mul_word_by_u64(I, tip, gas_used));
    WordOrValue cb; cb = block_lookup(I, BLK_Coinbase);
    Fr coinbase; EV_TRY(coinbase = word_to_fq(I, cb.w, 20));
    EV_TRY(add_balance(I, coinbase, reward));
    Fr status; EV_TRY(status = tx_receipt(I, 1, tx_id, 1));  // PostStateOrStatus
    constrain_equal(I, fr_mul(fr_sub(fr_u(1), is_tx_invalid), is_persistent), status);
    Fr log_id; EV_TRY(log_id = tx_receipt(I, 1, tx_id, 3));  // LogLength
    constrain_equal(I, log_id, ev_curr(I, S_LOG));
    if (invalid) constrain_zero(I, log_id);
    if (I.err) return;
    const bool is_first_tx = fr_eq_u64(tx_id, 1);
    Fr cum = fr_zero();
    if (!is_first_tx) EV_TRY(cum = tx_receipt(I, 0, fr_sub_u64(tx_id, 1), 2));  // CumulativeGasUsed of the previous tx
    Fr new_cum; EV_TRY(new_cum = tx_receipt(I, 1, tx_id, 2));
    constrain_equal(I, fr_add(cum, gas_used), new_cum); if (I.err) return;
    const u32 next_state = ev_next(I, S_STATE).v[0];
    if (next_state == ES_BeginTx) {
        const Fr next_rwc = ev_next(I, S_RWC);
        Fr nxt_tx; nxt_tx = call_context_lookup(I, CC_TxId, 0, &next_rwc);
        constrain_equal(I, nxt_tx, fr_add_u64(tx_id, 1));
        transition(I, S_RWC, t_delta_i(10 - (is_first_tx ? 1 : 0)));
    }
    if (next_state == ES_EndBlock) {
        transition(I, S_RWC, t_delta_i(9 - (is_first_tx ? 1 : 0)));
        transition(I, S_CALL_ID, t_same());
    }
}

ZK_HD void g_end_block(Ins& I, Tail& T, bool is_last) {  // end_block.py
    const EvmArgs& a = *I.a;
    ev_require(I, a.agg_bad_invalid_rows == 0u); if (I.err) return;  // `.value.value()` on every TxInvalid row (:70-78)
    const u32 total_txs = a.agg_total_txs, total_valid_txs = a.agg_total_txs - a.agg_invalid_txs;
    const Fr rwc_m1 = fr_sub_u64(I.rwc, 1);
    const bool is_empty = fr_is_zero(rwc_m1);
    const Fr total_rws = is_empty ? fr_zero() : fr_add_u64(rwc_m1, 2);
    if (!is_last) {
        transition(I, S_RWC, t_same());
        transition(I, S_CALL_ID, t_same());
        return;
    }
    if (is_empty) {
        ev_require(I, total_valid_txs == 0u);
        ev_require(I, a.agg_total_wds == 0u);
        if (I.err) return;
    } else {
        Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
        constrain_equal(I, tx_id, fr_u(total_txs)); if (I.err) return;
        WordOrValue gl; gl = block_lookup(I, BLK_GasLimit);
        Fr gas_limit; EV_TRY(gas_limit = value_of(I, gl));
        Fr cumulative; EV_TRY(cumulative = tx_receipt(I, 0, fr_u(total_txs), 2));
        u32 exceeded, eq; EV_TRY(ev_compare(I, gas_limit, cumulative, 8, exceeded, eq));
        ev_require(I, exceeded == 0u); if (I.err) return;
        // balance updates of the validators' withdrawals, in id order (:150-156)
        for (u32 k = 0; k < a.withdrawals.n; k++) {
            const Fr amount = zk_table_cell(a.withdrawals, k, 3);
            if (fr_is_zero(amount)) continue;
            // Word(int(amount) * 10^9): a 254-bit amount times 2^30 can exceed 2^256 -> AssertionError
            const U512 prod = u256_mul_full(amount, fr_u(1000000000ull));
            I.seq++;
            if (!fr_is_zero(u512_hi(prod))) { ev_fail(I, ZK_ASSERT); return; }
            EV_TRY(add_balance(I, zk_table_cell(a.withdrawals, k, 2), word_from_u256(u512_lo(prod))));
        }
        I.seq++;  // padding count == max_withdrawals - total_withdrawals: holds by construction
    }
    if (total_txs != a.agg_max_txs) {
        WordOrValue cw; cw = tx_lookup(I, fr_u((u64)total_txs + 1), TXC_CallerAddress); if (I.err) return;
        I.seq++;  // Word(0)
        constrain_equal_word(I, cw.w, word_zero()); if (I.err) return;
    }
    {   // rw_table_start_lookup(1), rw_table_start_lookup(max_rws - total_rws - total_withdrawals) (instruction.py:897-899)
        RwQ Q;
        rwq_init(Q, 0, TG_Start);
        Fr one = fr_u(1);
        rw_lookup(I, Q, &one); if (I.err) return;
        RwQ R;
        rwq_init(R, 0, TG_Start);
        Fr c2 = fr_sub(fr_sub(fr_u(a.rw.n), total_rws), fr_u(a.agg_total_wds));
        rw_lookup(I, R, &c2);
    }
}

// ---- BeginTx (begin_tx.py) -----------------------------------------------------------------------
// add_words (util/arithmetic.py:236-242) for three addends
ZK_HD Word add_words3(Ins& I, const Word& x, const Word& y, const Word& z, Fr& carry_hi) {
    Fr slo = fr_add(fr_add(x.lo, y.lo), z.lo), sum_lo, c_lo;
    split128(slo, sum_lo, c_lo);
    Fr shi = fr_add(fr_add(fr_add(x.hi, y.hi), z.hi), c_lo), sum_hi;
    split128(shi, sum_hi, carry_hi);
    return word_checked(I, sum_lo, sum_hi);
}
// constrain_zero(add_account_to_access_list(tx_id, address)) without reversion info
ZK_HD void access_list_must_be_cold(Ins& I, const Fr& tx_id, const Fr& address) {
    RwQ W;
    rwq_init(W, 1, TG_TxAccessListAccount);
    rwq_set(W, R_ID, tx_id);
    rwq_set(W, R_ADDR, address);
    rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
    u32 r; r = rw_lookup(I, W); if (I.err) return;
    Fr prev = value_of(I, rw_value_prev(I, r)); if (I.err) return;
    constrain_zero(I, prev);
}
ZK_HD Fr tx_value_of(Ins& I, const Fr& tx_id, u32 tag) {  // tx_context_lookup (instruction.py:686-687)
    WordOrValue v; v = tx_lookup(I, tx_id, tag);
    return value_of(I, v);
}
ZK_HD void g_begin_tx(Ins& I, Tail& T, bool is_first) {
    const Fr call_id = I.rwc;
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId, 0, &call_id);
    Reversion rv; EV_TRY(rv = reversion_info(I, &call_id));
    Fr is_success; is_success = call_context_lookup(I, CC_IsSuccess, 0, &call_id);
    constrain_equal(I, is_success, rv.persistent);
    if (is_first) constrain_equal(I, tx_id, fr_u(1));
    if (I.err) return;
    WordOrValue cbw; cbw = block_lookup(I, BLK_Coinbase);
    Fr coinbase; EV_TRY(coinbase = word_to_fq(I, cbw.w, 20));
    WordOrValue caller_w; caller_w = tx_lookup(I, tx_id, TXC_CallerAddress);
    Fr caller; EV_TRY(caller = word_to_fq(I, caller_w.w, 20));
    WordOrValue callee_w; callee_w = tx_lookup(I, tx_id, TXC_CalleeAddress);
    Fr callee; EV_TRY(callee = word_to_fq(I, callee_w.w, 20));
    Fr tx_is_create; EV_TRY(tx_is_create = tx_value_of(I, tx_id, TXC_IsCreate));
    WordOrValue tx_value; tx_value = tx_lookup(I, tx_id, TXC_Value);
    Fr cd_length; EV_TRY(cd_length = tx_value_of(I, tx_id, TXC_CallDataLength));
    ev_require(I, !fr_is_zero(caller)); if (I.err) return;
    Fr is_tx_invalid; EV_TRY(is_tx_invalid = tx_value_of(I, tx_id, TXC_TxInvalid));
    Fr tx_nonce; EV_TRY(tx_nonce = tx_value_of(I, tx_id, TXC_Nonce));
    Fr nonce, nonce_prev;
    {
        RwQ Q;
        rwq_init(Q, 1, TG_Account);
        rwq_set(Q, R_ADDR, caller);
        rwq_set(Q, R_FT, fr_u(ACC_Nonce));
        u32 r; r = rw_lookup(I, Q); if (I.err) return;
        EV_TRY(nonce = value_of(I, rw_value(I, r)));
        EV_TRY(nonce_prev = value_of(I, rw_value_prev(I, r)));
    }
    const u32 is_nonce_valid = fr_eq(tx_nonce, nonce_prev) ? 1u : 0u;
    constrain_equal(I, nonce, fr_sub(fr_add_u64(nonce_prev, 1), is_tx_invalid)); if (I.err) return;
    Fr tx_gas; EV_TRY(tx_gas = tx_value_of(I, tx_id, TXC_Gas));
    WordOrValue gpw; gpw = tx_lookup(I, tx_id, TXC_GasPrice); if (I.err) return;
    Word gas_fee; EV_TRY(gas_fee = mul_word_by_u64(I, gpw.w, tx_gas));
    Fr calldata_gas; EV_TRY(calldata_gas = tx_value_of(I, tx_id, TXC_CallDataGasCost));
    const bool is_create = fr_eq_u64(tx_is_create, 1);
    Fr cost = fr_u(21000);
    if (is_create) {
        Fr words; EV_TRY(words = constant_divmod_shift(I, fr_add_u64(cd_length, 31), 5, 8));
        cost = fr_add_u64(fr_add(words, words), 53000);
    }
    Fr accesslist_gas; EV_TRY(accesslist_gas = tx_value_of(I, tx_id, TXC_AccessListGasCost));
    const Fr intrinsic = fr_add(fr_add(calldata_gas, cost), accesslist_gas);
    u32 gas_not_enough, eq; EV_TRY(ev_compare(I, tx_gas, intrinsic, 31, gas_not_enough, eq));
    const Fr gas_left = gas_not_enough ? tx_gas : fr_sub(tx_gas, intrinsic);
    // generate_contract_address (keccak of rlp([caller, nonce])): the value is only used by creations
    const Fr contract = is_create ? keccak_create_address(caller, tx_nonce) : fr_zero();
    I.seq++;  // address_to_word(contract_address)
    const Word contract_w = word_of(fr_from_u128(fr_lo64(contract), fr_hi64of128(contract)), fr_u((u64)contract.v[4]));
    const Fr callee_address = is_create ? contract : callee;
    EV_TRY(access_list_must_be_cold(I, tx_id, coinbase));
    EV_TRY(access_list_must_be_cold(I, tx_id, caller));
    EV_TRY(access_list_must_be_cold(I, tx_id, callee_address));
    const bool invalid = fr_eq_u64(is_tx_invalid, 1);
    Word value = tx_value.w, fee = gas_fee;
    if (invalid) { I.seq += 2; value = word_zero(); fee = word_zero(); }
    Word sender_prev;
    {   // transfer_with_gas_fee (instruction.py:1099-1109): sub_balance then add_balance, both reversible
        RwQ Q;
        rwq_init(Q, 1, TG_Account);
        rwq_set(Q, R_ADDR, caller);
        rwq_set(Q, R_FT, fr_u(ACC_Balance));
        u32 r; r = state_write(I, Q, rv); if (I.err) return;
        const Word balance = rw_word(I, r, R_VAL_LO);
        sender_prev = rw_word(I, r, R_PREV_LO);
        Fr carry; Word sum; sum = add_words3(I, balance, value, fee, carry);
        constrain_equal_word(I, sender_prev, sum);
        constrain_zero(I, carry);
        if (I.err) return;
        RwQ R;
        rwq_init(R, 1, TG_Account);
        rwq_set(R, R_ADDR, callee_address);
        rwq_set(R, R_FT, fr_u(ACC_Balance));
        r = state_write(I, R, rv); if (I.err) return;
        const Word rbal = rw_word(I, r, R_VAL_LO), rprev = rw_word(I, r, R_PREV_LO);
        Fr carry2; Word sum2; sum2 = add_words2(I, rprev, value, carry2);
        constrain_equal_word(I, rbal, sum2);
        constrain_zero(I, carry2);
        if (I.err) return;
    }
    Fr lhs; EV_TRY(lhs = word_to_fq(I, sender_prev, 31));
    Fr v31; EV_TRY(v31 = word_to_fq(I, tx_value.w, 31));
    Fr f31; EV_TRY(f31 = word_to_fq(I, gas_fee, 31));
    u32 balance_not_enough; EV_TRY(ev_compare(I, lhs, fr_add(v31, f31), 31, balance_not_enough, eq));
    const u32 invalid_tx = 1u - (1u - balance_not_enough) * (1u - gas_not_enough) * is_nonce_valid;
    constrain_equal(I, is_tx_invalid, fr_u(invalid_tx)); if (I.err) return;
    const Word empty_hash = word_of(fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull), fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull));
    Word code_hash = word_zero();
    bool to_end_tx;
    if (is_create) {
        to_end_tx = invalid || fr_is_zero(cd_length);
        if (!to_end_tx) {
            // the creation code is the tx calldata: its keccak is the code hash, and it is copied to the bytecode table
            CopyRes cr;
            EV_TRY(cr = copy_lookup(I, word_value(tx_id), CDT_TxCalldata, word_value(call_id), CDT_RlcAcc, fr_zero(), cd_length, fr_zero(), cd_length,
                                    fr_add_u64(I.rwc, I.rw_off)));
            ev_require(I, fr_is_zero(cr.rwc_inc)); if (I.err) return;
            EV_TRY(code_hash = keccak_lookup(I, cd_length, cr.rlc_acc));
            EV_TRY(cr = copy_lookup(I, word_value(tx_id), CDT_TxCalldata, code_hash, CDT_Bytecode, fr_zero(), cd_length, fr_zero(), cd_length,
                                    fr_add_u64(I.rwc, I.rw_off)));
            ev_require(I, fr_is_zero(cr.rwc_inc)); if (I.err) return;
        }
    } else {
        I.seq++;
        if (fr_fits64(callee) && fr_lo64(callee) >= 1 && fr_lo64(callee) <= 9) { ev_fail(I, ZK_NOT_IMPLEMENTED); return; }  // precompile callee
        code_hash = account_read_word(I, callee, ACC_CodeHash); if (I.err) return;
        I.seq++;  // Word(EMPTY_CODE_HASH)
        to_end_tx = is_equal_word(code_hash, empty_hash) || invalid;
    }
    if (to_end_tx) {
        constrain_equal(I, rv.persistent, fr_u(1));
        ev_require(I, ev_next(I, S_STATE).v[0] == ES_EndTx && fr_fits32(ev_next(I, S_STATE)));
        transition(I, S_RWC, t_delta(fr_u(I.rw_off)));
        transition(I, S_CALL_ID, t_to(call_id));
        return;
    }
    const u32 tags[13] = {CC_Depth, CC_CallerAddress, CC_CalleeAddress, CC_CallDataOffset, CC_CallDataLength, CC_Value, CC_IsStatic,
                          CC_LastCalleeId, CC_LastCalleeReturnDataOffset, CC_LastCalleeReturnDataLength, CC_IsRoot, CC_IsCreate, CC_CodeHash};
    for (int k = 0; k < 13; k++) {
        Word want = word_zero();
        switch (k) {
        case 0: case 10: want = word_value(fr_u(1)); break;
        case 1: want = caller_w.w; break;
        case 2: want = is_create ? contract_w : callee_w.w; break;
        case 4: want = word_value(cd_length); break;
        case 5: want = tx_value.w; break;
        case 11: want = word_value(fr_u(is_create ? 1 : 0)); break;
        case 12: want = code_hash; break;
        default: break;
        }
        WordOrValue got; got = call_context_lookup_word(I, tags[k], 0, &call_id);
        constrain_equal_word(I, got.w, want);
        if (I.err) return;
    }
    // step_state_transition_to_new_context (instruction.py:266-290)
    transition(I, S_RWC, t_delta(fr_u(I.rw_off)));
    transition(I, S_CALL_ID, t_to(call_id));
    transition(I, S_IS_ROOT, t_to(fr_u(1)));
    transition(I, S_IS_CREATE, t_to(fr_u(is_create ? 1 : 0)));
    ev_require(I, fr_eq(ev_next(I, S_CH_LO), code_hash.lo) && fr_eq(ev_next(I, S_CH_HI), code_hash.hi));
    transition(I, S_GAS, t_to(gas_left));
    transition(I, S_REV, t_to(fr_u(2)));
    transition(I, S_LOG, t_to(fr_zero()));
    transition(I, S_PC, t_to(fr_zero()));
    transition(I, S_SP, t_to(fr_u(1024)));
    transition(I, S_MWS, t_to(fr_zero()));
}

// ---- CALL / CALLCODE / DELEGATECALL / STATICCALL (callop.py, util/call_gadget.py, error_oog_call.py) ----
struct CallGadget {
    Word value, callee_code_hash;
    Fr gas, callee_address, cd_offset, cd_length, rd_offset, rd_length, next_memory_size, memory_expansion_gas, is_success;
    u32 is_u64_gas, has_value, is_empty_code_hash, callee_not_exists;
};
ZK_HD void call_gadget(Ins& I, CallGadget& C, bool is_success_call, u32 opcode_v) {  // call_gadget.py:39-106
    const bool is_call = opcode_v == OP_CALL, is_callcode = opcode_v == OP_CALLCODE;
    const bool is_delegatecall = opcode_v == OP_DELEGATECALL, is_staticcall = opcode_v == OP_STATICCALL;
    ev_require(I, is_call || is_callcode || is_delegatecall || is_staticcall); if (I.err) return;
    Word gas_w, callee_w;
    gas_w = stack_pop(I); callee_w = stack_pop(I);
    if (is_call || is_callcode) C.value = stack_pop(I);
    else { I.seq++; C.value = word_zero(); }  // Word(0)
    Word cd_off_w, cd_len_w, rd_off_w, rd_len_w, result;
    cd_off_w = stack_pop(I); cd_len_w = stack_pop(I); rd_off_w = stack_pop(I); rd_len_w = stack_pop(I);
    result = stack_push(I);
    if (I.err) return;
    C.is_success = result.lo;
    Word sw; sw = word_checked(I, C.is_success, fr_zero());
    constrain_equal_word(I, sw, result);
    ev_require(I, fr_le_u64(C.is_success, 1));
    if (!is_success_call) constrain_zero(I, C.is_success);
    if (I.err) return;
    C.gas = word_to_fq(I, gas_w, 8); if (I.err) return;
    {
        U256 gb = to_u256(I, gas_w); if (I.err) return;
        C.is_u64_gas = (gb.v[2] | gb.v[3] | gb.v[4] | gb.v[5] | gb.v[6] | gb.v[7]) == 0u ? 1u : 0u;
    }
    const bool no_value_op = is_delegatecall || is_staticcall;
    C.has_value = no_value_op ? 0u : 1u - is_zero_word(C.value);
    if (no_value_op) { ev_require(I, fr_is_zero(C.value.lo) && fr_is_zero(C.value.hi)); if (I.err) return; }
    C.callee_address = word_to_fq(I, callee_w, 20); if (I.err) return;
    memory_offset_and_length(I, cd_off_w, cd_len_w, C.cd_offset, C.cd_length); if (I.err) return;
    memory_offset_and_length(I, rd_off_w, rd_len_w, C.rd_offset, C.rd_length); if (I.err) return;
    {   // memory_expansion_dynamic_length with the return-data range (instruction.py:1157-1181)
        const Fr mws = ev_curr(I, S_MWS);
        Fr cd_size = constant_divmod_shift(I, fr_add_u64(fr_add(C.cd_offset, C.cd_length), 31), 5, 4); if (I.err) return;
        u32 lt, eq; ev_compare(I, mws, cd_size, 4, lt, eq); if (I.err) return;
        Fr nxt = ev_select_b(I, lt) ? cd_size : mws;
        Fr rd_size = constant_divmod_shift(I, fr_add_u64(fr_add(C.rd_offset, C.rd_length), 31), 5, 4); if (I.err) return;
        ev_compare(I, nxt, rd_size, 4, lt, eq); if (I.err) return;
        nxt = ev_select_b(I, lt) ? rd_size : nxt;
        Fr g0 = memory_gas_cost(I, mws); if (I.err) return;
        Fr g1 = memory_gas_cost(I, nxt); if (I.err) return;
        C.next_memory_size = nxt;
        C.memory_expansion_gas = fr_sub(g1, g0);
    }
    C.callee_code_hash = account_read_word(I, C.callee_address, ACC_CodeHash); if (I.err) return;
    I.seq++;  // Word(EMPTY_CODE_HASH)
    const Word empty_hash = word_of(fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull), fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull));
    C.is_empty_code_hash = is_equal_word(C.callee_code_hash, empty_hash);
    C.callee_not_exists = is_zero_word(C.callee_code_hash);
}
ZK_HD Fr call_gas_cost(Ins& I, const CallGadget& C, const Fr& is_warm, bool is_call) {  // call_gadget.py:108-124
    const bool warm = ev_select(I, is_warm);
    u64 g = warm ? 100 : 2600;
    if (C.has_value) {
        g += 9000;
        if (is_call && C.callee_not_exists && fr_eq_u64(C.is_success, 1)) g += 25000;
    }
    return fr_add_u64(C.memory_expansion_gas, g);
}
ZK_HD void g_error_oog_call(Ins& I, Tail& T) {  // error_oog_call.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const u32 ov = fr_le_u64(opcode, 255) ? opcode.v[0] : 0u;
    ev_require(I, ov == OP_CALL || ov == OP_CALLCODE || ov == OP_DELEGATECALL || ov == OP_STATICCALL); if (I.err) return;
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId); if (I.err) return;
    CallGadget C; call_gadget(I, C, false, ov); if (I.err) return;
    Fr is_warm; EV_TRY(is_warm = read_account_to_access_list(I, tx_id, C.callee_address));
    Fr gas_cost; EV_TRY(gas_cost = call_gas_cost(I, C, is_warm, true));
    oog_tail(T, gas_cost);
}
// state_write of an Account.Balance row (sub_balance / add_balance, instruction.py:987-1013)
ZK_HD void balance_move(Ins& I, const Fr& address, const Word& value, Reversion& rv, bool subtract) {
    RwQ Q;
    rwq_init(Q, 1, TG_Account);
    rwq_set(Q, R_ADDR, address);
    rwq_set(Q, R_FT, fr_u(ACC_Balance));
    u32 r; r = state_write(I, Q, rv); if (I.err) return;
    const Word bal = rw_word(I, r, R_VAL_LO), prev = rw_word(I, r, R_PREV_LO);
    Fr carry; Word sum; sum = add_words2(I, subtract ? bal : prev, value, carry);
    constrain_equal_word(I, subtract ? prev : bal, sum);
    constrain_zero(I, carry);
}
// StepState.aux_data of the current step (EvmArgs::aux): kind and the two cells
ZK_HD u32 aux_kind(const Ins& I) { return I.a->aux_kind ? I.a->aux_kind[I.idx] : 0u; }
ZK_HD Fr aux_cell(const Ins& I, u32 k) { return fr_load(I.a->aux + (I.idx * I.a->aux_cells + k) * 4); }
ZK_HD Word aux_word(const Ins& I, u32 k = 0) { return word_of(aux_cell(I, k), aux_cell(I, k + 1)); }
// aux_data of the shape a precompile gadget expects (flatten_step_aux); anything else is not evaluated
ZK_HD bool aux_expect(Ins& I, u32 kind, u32 n_cells) {
    if (aux_kind(I) == kind && I.a->aux_cells >= n_cells) return true;
    if (I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq);
    return false;
}
ZK_HD void g_callop(Ins& I, Tail& T) {  // callop.py; precompile callees (StepState.aux_data) -> ZK_UNSUPPORTED
    Fr opcode; opcode = opcode_lookup(I, true);
    const u32 ov = fr_le_u64(opcode, 255) ? opcode.v[0] : 0u;
    const bool is_call = ov == OP_CALL, is_callcode = ov == OP_CALLCODE, is_delegatecall = ov == OP_DELEGATECALL;
    fixed_lookup(I, FX_ResponsibleOpcode, ev_curr(I, S_STATE), opcode, fr_zero()); if (I.err) return;
    const Fr callee_call_id = I.rwc;
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    Reversion rv; EV_TRY(rv = reversion_info(I));
    WordOrValue ctx_caller_w; ctx_caller_w = call_context_lookup_word(I, CC_CalleeAddress);
    Fr ctx_caller; EV_TRY(ctx_caller = word_to_fq(I, ctx_caller_w.w, 20));
    Fr is_static, depth;
    is_static = call_context_lookup(I, CC_IsStatic);
    depth = call_context_lookup(I, CC_Depth);
    if (I.err) return;
    Word parent_caller_w = word_zero(), parent_value = word_zero();
    if (is_delegatecall) {
        WordOrValue a, b;
        a = call_context_lookup_word(I, CC_CallerAddress);
        b = call_context_lookup_word(I, CC_Value);
        parent_caller_w = a.w; parent_value = b.w;
    } else {
        I.seq += 2;  // Word(0), Word(0)
    }
    if (I.err) return;
    CallGadget C; call_gadget(I, C, true, ov); if (I.err) return;
    I.seq++;  // select(is_callcode + is_delegatecall, ..)
    const Fr callee_address = (is_callcode || is_delegatecall) ? ctx_caller : C.callee_address;
    I.seq++;  // address_to_word: both candidates passed word_to_fq(.., 20)
    const Word callee_address_w = word_of(fr_from_u128(fr_lo64(callee_address), fr_hi64of128(callee_address)),
                                          fr_u((u64)callee_address.v[4]));
    I.seq++;  // select_word(is_delegatecall, ..)
    const Word caller_w = is_delegatecall ? parent_caller_w : ctx_caller_w.w;
    Fr caller_address; EV_TRY(caller_address = word_to_fq(I, caller_w, 20));
    Fr is_warm;
    {
        RwQ W;
        rwq_init(W, 1, TG_TxAccessListAccount);
        rwq_set(W, R_ID, tx_id);
        rwq_set(W, R_ADDR, C.callee_address);
        rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
        u32 wr; wr = state_write(I, W, rv); if (I.err) return;
        EV_TRY(is_warm = value_of(I, rw_value_prev(I, wr)));
    }
    constrain_zero(I, C.has_value ? is_static : fr_zero()); if (I.err) return;
    Reversion crv; EV_TRY(crv = reversion_info(I, &callee_call_id));
    constrain_equal(I, crv.persistent, fr_mul(rv.persistent, C.is_success)); if (I.err) return;
    const bool success = fr_eq_u64(C.is_success, 1);
    if (success && fr_is_zero(rv.persistent)) {
        const Fr want = fr_sub(rv.end, rv.rwc);
        rv.rwc = fr_add_u64(rv.rwc, 1);
        constrain_equal(I, crv.end, want); if (I.err) return;
    }
    u32 insufficient = 0;
    if (is_call || is_callcode) {
        Word caller_balance; caller_balance = account_read_word(I, caller_address, ACC_Balance); if (I.err) return;
        u32 eqw; compare_word(I, caller_balance, C.value, insufficient, eqw); if (I.err) return;
    }
    u32 depth_ok, eq; EV_TRY(ev_compare(I, depth, fr_u(1025), 2, depth_ok, eq));
    const bool precheck_ok = depth_ok == 1u && insufficient == 0u;
    if (!precheck_ok) { constrain_zero(I, C.is_success); if (I.err) return; }
    if (is_call && precheck_ok) {  // transfer (instruction.py:1111-1120) under the callee's reversion info
        EV_TRY(balance_move(I, caller_address, C.value, crv, true));
        EV_TRY(balance_move(I, callee_address, C.value, crv, false));
    }
    if (is_callcode && success) { ev_require(I, insufficient == 0u); if (I.err) return; }
    Fr gas_cost; EV_TRY(gas_cost = call_gas_cost(I, C, is_warm, is_call));
    const Fr gas_available = fr_sub(ev_curr(I, S_GAS), gas_cost);
    Fr one_64th; EV_TRY(one_64th = constant_divmod_shift(I, gas_available, 6, 8));
    const Fr all_but = fr_sub(gas_available, one_64th);
    u32 lt; EV_TRY(ev_compare(I, all_but, C.gas, 8, lt, eq));
    const Fr capped = ev_select_b(I, lt) ? all_but : C.gas;
    Fr callee_gas_left = ev_select_b(I, C.is_u64_gas) ? capped : all_but;
    const bool is_precompile = fr_fits64(C.callee_address) && fr_lo64(C.callee_address) >= 1 && fr_lo64(C.callee_address) <= 9;
    {
        const Fr ns = ev_next(I, S_STATE);
        const u32 nsv = ns.v[0];
        const bool nxt_pre = fr_fits32(ns) && (nsv == ES_ECRECOVER || nsv == ES_SHA256 || nsv == ES_RIPEMD160 || nsv == ES_DATACOPY ||
                                                nsv == ES_BIGMODEXP || nsv == ES_BN254_ADD || nsv == ES_BN254_SCALAR_MUL ||
                                                nsv == ES_BN254_PAIRING || nsv == ES_BLAKE2F);
        ev_require(I, is_precompile == nxt_pre); if (I.err) return;
    }
    const int sp_delta = 5 + (is_call ? 1 : 0) + (is_callcode ? 1 : 0);
    const u32 no_callee_code = C.is_empty_code_hash + C.callee_not_exists;
    if (!precheck_ok || (no_callee_code == 1u && !is_precompile)) {
        const u32 tags[3] = {CC_LastCalleeId, CC_LastCalleeReturnDataOffset, CC_LastCalleeReturnDataLength};
        for (int k = 0; k < 3; k++) {
            Fr v; v = call_context_lookup(I, tags[k], 1);
            constrain_equal(I, v, fr_zero()); if (I.err) return;
        }
        transition(I, S_RWC, t_delta(fr_u(I.rw_off)));
        transition(I, S_PC, t_delta_i(1));
        transition(I, S_SP, t_delta_i(sp_delta));
        transition(I, S_GAS, t_delta(fr_sub(fr_u(C.has_value ? 2300 : 0), gas_cost)));
        transition(I, S_MWS, t_to(C.next_memory_size));
        transition(I, S_REV, t_delta_i(3));
        transition(I, S_CALL_ID, t_same());
        transition(I, S_IS_ROOT, t_same());
        transition(I, S_IS_CREATE, t_same());
        ev_require(I, fr_eq(ev_next(I, S_CH_LO), ev_curr(I, S_CH_LO)) && fr_eq(ev_next(I, S_CH_HI), ev_curr(I, S_CH_HI)));
        return;
    }
    if (is_precompile) {  // callop.py:154-276
        if (!aux_expect(I, AUX_PAIR, 2)) return;
        const Fr input_len = aux_cell(I, 0), return_len = aux_cell(I, 1);  // Python ints < p on the host
        const Fr min_rd_copy_size = fr_lt(C.rd_length, return_len) ? C.rd_length : return_len;
        ev_require(I, no_callee_code == 1u); if (I.err) return;
        constrain_equal(I, is_warm, fr_u(1)); if (I.err) return;
        {
            const u32 tags[7] = {CC_IsSuccess, CC_CalleeAddress, CC_CallerId, CC_CallDataOffset, CC_CallDataLength, CC_ReturnDataOffset,
                                 CC_ReturnDataLength};
            for (int k = 0; k < 7; k++) {
                Word want;
                switch (k) {
                case 0: want = word_value(C.is_success); break;
                case 1: want = callee_address_w; break;
                case 2: want = word_value(I.call_id); break;
                case 3: want = word_value(C.cd_offset); break;
                case 4: want = word_value(C.cd_length); break;
                case 5: want = word_value(C.rd_offset); break;
                default: want = word_value(C.rd_length); break;
                }
                WordOrValue got; got = call_context_lookup_word(I, tags[k], 1, &callee_call_id);
                constrain_equal_word(I, got.w, want); if (I.err) return;
            }
        }
        {
            const u32 tags[8] = {CC_ProgramCounter, CC_StackPointer, CC_GasLeft, CC_MemorySize, CC_ReversibleWriteCounter, CC_LastCalleeId,
                                 CC_LastCalleeReturnDataOffset, CC_LastCalleeReturnDataLength};
            for (int k = 0; k < 8; k++) {
                Fr want;
                switch (k) {
                case 0: want = fr_add_u64(I.pc, 1); break;
                case 1: want = fr_add_u64(I.sp, (u64)sp_delta); break;
                case 2: want = fr_sub(fr_sub(ev_curr(I, S_GAS), gas_cost), callee_gas_left); break;
                case 3: want = C.next_memory_size; break;
                case 4: want = fr_add_u64(ev_curr(I, S_REV), 1); break;
                case 5: want = callee_call_id; break;
                case 6: want = fr_zero(); break;
                default: want = return_len; break;
                }
                Fr v; v = call_context_lookup(I, tags[k], 1);
                constrain_equal(I, v, want); if (I.err) return;
            }
        }
        Fr rwc_inc = fr_u(I.rw_off);
        if (!fr_is_zero(input_len)) {
            CopyRes cr;
            EV_TRY(cr = copy_lookup(I, word_value(I.call_id), CDT_Memory, word_value(callee_call_id), CDT_RlcAcc, C.cd_offset,
                                    fr_add(C.cd_offset, input_len), fr_zero(), input_len, fr_add(I.rwc, rwc_inc)));
            rwc_inc = fr_add(rwc_inc, cr.rwc_inc);
        }
        if (success && !fr_is_zero(return_len)) {
            CopyRes cr;
            EV_TRY(cr = copy_lookup(I, word_value(callee_call_id), CDT_Memory, word_value(callee_call_id), CDT_RlcAcc, fr_zero(), return_len,
                                    fr_zero(), return_len, fr_add(I.rwc, rwc_inc)));
            rwc_inc = fr_add(rwc_inc, cr.rwc_inc);
            EV_TRY(cr = copy_lookup(I, word_value(callee_call_id), CDT_Memory, word_value(I.call_id), CDT_Memory, fr_zero(), min_rd_copy_size,
                                    C.rd_offset, min_rd_copy_size, fr_add(I.rwc, rwc_inc)));
            rwc_inc = fr_add(rwc_inc, cr.rwc_inc);
        }
        Fr mem_words; EV_TRY(mem_words = constant_divmod_shift(I, fr_add_u64(min_rd_copy_size, 31), 5, 4));
        if (C.has_value) callee_gas_left = fr_add_u64(callee_gas_left, 2300);
        transition(I, S_RWC, t_delta(rwc_inc));
        transition(I, S_CALL_ID, t_to(callee_call_id));
        transition(I, S_IS_ROOT, t_to(fr_zero()));
        transition(I, S_IS_CREATE, t_to(fr_zero()));
        {
            const Word empty = evm_empty_code_hash();
            ev_require(I, fr_eq(ev_next(I, S_CH_LO), empty.lo) && fr_eq(ev_next(I, S_CH_HI), empty.hi));
        }
        transition(I, S_GAS, t_to(callee_gas_left));
        transition(I, S_REV, t_to(fr_u(2)));
        transition(I, S_PC, t_delta_i(1));
        transition(I, S_SP, t_same());
        transition(I, S_MWS, t_to(mem_words));
        transition(I, S_LOG, t_same());
        if (I.err) return;
        // PrecompileGadget (util/precompile_gadget.py:9-41); the address is 1..9 here
        ev_require(I, true);
        const u64 addr = fr_lo64(C.callee_address);
        if (addr == 4) constrain_equal(I, return_len, C.cd_length);
        else if (addr == 1) ev_require(I, fr_eq_u64(return_len, 32) || fr_is_zero(return_len));
        else if (addr == 6) constrain_equal(I, C.cd_length, fr_u(128));
        else if (addr == 7) constrain_equal(I, C.cd_length, fr_u(96));
        else if (addr == 8) ev_require(I, fr_mod_small(C.cd_length, 192) == 0u);
        return;
    }
    {   // save the caller's call state
        const u32 tags[5] = {CC_ProgramCounter, CC_StackPointer, CC_GasLeft, CC_MemorySize, CC_ReversibleWriteCounter};
        for (int k = 0; k < 5; k++) {
            Fr want;
            switch (k) {
            case 0: want = fr_add_u64(I.pc, 1); break;
            case 1: want = fr_add_u64(I.sp, (u64)sp_delta); break;
            case 2: want = fr_sub(fr_sub(ev_curr(I, S_GAS), gas_cost), callee_gas_left); break;
            case 3: want = C.next_memory_size; break;
            default: want = fr_add_u64(ev_curr(I, S_REV), 1); break;
            }
            Fr v; v = call_context_lookup(I, tags[k], 1);
            constrain_equal(I, v, want); if (I.err) return;
        }
    }
    I.seq++;  // select_word(is_delegatecall, parent_call_value, call.value), evaluated while the list is built
    {
        const u32 tags[18] = {CC_CallerId, CC_TxId, CC_Depth, CC_CallerAddress, CC_CalleeAddress, CC_CallDataOffset, CC_CallDataLength,
                              CC_ReturnDataOffset, CC_ReturnDataLength, CC_Value, CC_IsSuccess, CC_IsStatic, CC_LastCalleeId,
                              CC_LastCalleeReturnDataOffset, CC_LastCalleeReturnDataLength, CC_IsRoot, CC_IsCreate, CC_CodeHash};
        for (int k = 0; k < 18; k++) {
            Word want = word_zero();
            switch (k) {
            case 0: want = word_value(I.call_id); break;
            case 1: want = word_value(tx_id); break;
            case 2: want = word_value(fr_add_u64(depth, 1)); break;
            case 3: want = caller_w; break;
            case 4: want = callee_address_w; break;
            case 5: want = word_value(C.cd_offset); break;
            case 6: want = word_value(C.cd_length); break;
            case 7: want = word_value(C.rd_offset); break;
            case 8: want = word_value(C.rd_length); break;
            case 9: want = is_delegatecall ? parent_value : C.value; break;
            case 10: want = word_value(C.is_success); break;
            case 11: want = word_value(is_static); break;
            case 17: want = C.callee_code_hash; break;
            default: break;
            }
            WordOrValue got; got = call_context_lookup_word(I, tags[k], 0, &callee_call_id);
            constrain_equal_word(I, got.w, want); if (I.err) return;
        }
    }
    if (C.has_value) callee_gas_left = fr_add_u64(callee_gas_left, 2300);
    transition(I, S_RWC, t_delta(fr_u(I.rw_off)));
    transition(I, S_CALL_ID, t_to(callee_call_id));
    transition(I, S_IS_ROOT, t_to(fr_zero()));
    transition(I, S_IS_CREATE, t_to(fr_zero()));
    ev_require(I, fr_eq(ev_next(I, S_CH_LO), C.callee_code_hash.lo) && fr_eq(ev_next(I, S_CH_HI), C.callee_code_hash.hi));
    transition(I, S_GAS, t_to(callee_gas_left));
    transition(I, S_REV, t_to(fr_u(2)));
    transition(I, S_LOG, t_same());
    transition(I, S_PC, t_to(fr_zero()));
    transition(I, S_SP, t_to(fr_u(1024)));
    transition(I, S_MWS, t_to(fr_zero()));
}

ZK_HD void g_error_oog_sload_sstore(Ins& I, Tail& T) {  // error_oog_sload_sstore.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_sstore = fr_eq_u64(opcode, OP_SSTORE), is_sload = fr_eq_u64(opcode, OP_SLOAD);
    ev_require(I, is_sstore || is_sload); if (I.err) return;
    Word key; key = stack_pop(I);
    Fr tx_id; tx_id = call_context_lookup(I, CC_TxId);
    WordOrValue cw; cw = call_context_lookup_word(I, CC_CalleeAddress);
    Fr callee; EV_TRY(callee = word_to_fq(I, cw.w, 20));
    Fr is_warm;
    {   // read_account_storage_to_access_list returns row.value (instruction.py:1088-1097)
        RwQ Q;
        rwq_init(Q, 0, TG_TxAccessListAccountStorage);
        rwq_set(Q, R_ID, tx_id);
        rwq_set(Q, R_ADDR, callee);
        rwq_set_word(Q, R_KEY_LO, key);
        u32 r; r = rw_lookup(I, Q);
        EV_TRY(is_warm = value_of(I, rw_value(I, r)));
    }
    u64 gas_cost;
    if (is_sload) {
        gas_cost = fr_eq_u64(is_warm, 1) ? 100 : 2100;
    } else {
        Word value; value = stack_pop(I);
        RwQ Q;
        rwq_init(Q, 0, TG_AccountStorage);
        rwq_set(Q, R_ID, tx_id);
        rwq_set(Q, R_ADDR, callee);
        rwq_set_word(Q, R_KEY_LO, key);
        u32 r; r = rw_lookup(I, Q); if (I.err) return;
        const Word value_prev = rw_word(I, r, R_VAL_LO);
        if (aux_kind(I) != 2u) { if (I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq); return; }  // Word(curr.aux_data)
        I.seq++;
        const Word orig = aux_word(I);
        if (word_eq(value, value_prev)) gas_cost = 100;
        else if (word_eq(value_prev, orig)) { I.seq++; gas_cost = (fr_is_zero(orig.lo) && fr_is_zero(orig.hi)) ? 20000 : 2900; }
        else gas_cost = 100;
        if (fr_is_zero(is_warm)) gas_cost += 2100;
    }
    u32 insufficient, eq; EV_TRY(ev_compare(I, ev_curr(I, S_GAS), fr_u(gas_cost), 8, insufficient, eq));
    if (is_sload) {
        ev_require(I, insufficient == 1u);
    } else {
        u32 lt; EV_TRY(ev_compare(I, ev_curr(I, S_GAS), fr_u(2300), 8, lt, eq));
        ev_require(I, (lt + eq + insufficient) != 0u);
    }
    if (I.err) return;
    T.err_tail = 1;
}

// ---- CREATE / CREATE2 (create.py) ----------------------------------------------------------------
ZK_HD void g_create(Ins& I, Tail& T) {
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_create = fr_eq_u64(opcode, OP_CREATE), is_create2 = fr_eq_u64(opcode, OP_CREATE2);
    fixed_lookup(I, FX_ResponsibleOpcode, ev_curr(I, S_STATE), opcode, fr_zero()); if (I.err) return;
    const Fr callee_call_id = I.rwc;
    Word value_w, offset_w, size_w, salt_w = word_zero(), ret_addr_w;
    value_w = stack_pop(I); offset_w = stack_pop(I); size_w = stack_pop(I);
    if (is_create2) salt_w = stack_pop(I);
    else I.seq++;  // Word(0)
    ret_addr_w = stack_push(I);
    if (I.err) return;
    Fr offset; EV_TRY(offset = word_to_fq(I, offset_w, 5));
    Fr size; EV_TRY(size = word_to_fq(I, size_w, 5));
    Fr depth, tx_id;
    depth = call_context_lookup(I, CC_Depth);
    tx_id = call_context_lookup(I, CC_TxId);
    WordOrValue caller_w; caller_w = call_context_lookup_word(I, CC_CallerAddress);
    if (I.err) return;
    Fr caller; EV_TRY(caller = word_to_fq(I, caller_w.w, 20));
    Fr nonce, nonce_prev, balance;
    {
        RwQ Q;
        rwq_init(Q, 1, TG_Account);
        rwq_set(Q, R_ADDR, caller);
        rwq_set(Q, R_FT, fr_u(ACC_Nonce));
        u32 r; r = rw_lookup(I, Q); if (I.err) return;
        EV_TRY(nonce = value_of(I, rw_value(I, r)));
        EV_TRY(nonce_prev = value_of(I, rw_value_prev(I, r)));
        RwQ B;
        rwq_init(B, 0, TG_Account);
        rwq_set(B, R_ADDR, caller);
        rwq_set(B, R_FT, fr_u(ACC_Balance));
        r = rw_lookup(I, B); if (I.err) return;
        EV_TRY(balance = value_of(I, rw_value(I, r)));
    }
    Fr is_success, is_static;
    is_success = call_context_lookup(I, CC_IsSuccess);
    is_static = call_context_lookup(I, CC_IsStatic);  // is_zero(is_static): result discarded (:48)
    (void)is_static;
    Reversion rv; EV_TRY(rv = reversion_info(I));
    const bool has_init_code = !fr_is_zero(size);
    Fr next_mem, mem_gas; EV_TRY(memory_expansion(I, offset, size, next_mem, mem_gas));
    Fr word_len; EV_TRY(word_len = constant_divmod_shift(I, fr_add_u64(size, 31), 5, 4));
    const Fr gas_left = ev_curr(I, S_GAS);
    Fr gas_cost = fr_add(fr_add_u64(mem_gas, 32000), fr_add(word_len, word_len));
    if (is_create2) gas_cost = fr_add(gas_cost, fr_mul_u64(word_len, 6));
    const Fr gas_available = fr_sub(gas_left, gas_cost);
    Fr one_64th; EV_TRY(one_64th = constant_divmod_shift(I, gas_available, 6, 8));
    const Fr all_but = fr_sub(gas_available, one_64th);
    ev_require(I, fr_fits128(gas_left), ZK_OVERFLOW_ERROR); if (I.err) return;  // WordOrValue(gas_left).to_le_bytes()
    const u32 is_u64_gas = fr_fits64(gas_left) ? 1u : 0u;
    u32 lt, eq; EV_TRY(ev_compare(I, all_but, gas_left, 8, lt, eq));
    const Fr capped = ev_select_b(I, lt) ? all_but : gas_left;
    const Fr callee_gas_left = ev_select_b(I, is_u64_gas) ? capped : all_but;
    u32 depth_ok; EV_TRY(ev_compare(I, depth, fr_u(1025), 2, depth_ok, eq));
    I.seq++;  // Word(balance.n)
    u32 insufficient, eqw;
    EV_TRY(compare_word(I, word_from_u256(balance), value_w, insufficient, eqw));
    u32 nonce_ok; EV_TRY(ev_compare(I, nonce_prev, fr_u(0xffffffffffffffffull), 8, nonce_ok, eq));
    const bool precheck_ok = depth_ok == 1u && insufficient == 0u && nonce_ok == 1u;
    const int sp_delta = 2 + (is_create2 ? 1 : 0);
    bool nac = false;
    if (precheck_ok) {
        Word code_hash;
        if (has_init_code) {
            if (aux_kind(I) != 1u) { if (I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq); return; }  // curr.aux_data: a Word
            code_hash = aux_word(I);
        } else {
            I.seq++;  // Word(EMPTY_CODE_HASH)
            code_hash = word_of(fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull), fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull));
        }
        Fr contract;
        if (is_create) {
            contract = keccak_create_address(caller, nonce);
        } else {
            U256 salt_v, hash_v;
            EV_TRY(salt_v = int_bytes32(I, salt_w));
            EV_TRY(hash_v = int_bytes32(I, code_hash));
            contract = keccak_create2_address(caller, salt_v, hash_v);
        }
        I.seq++;  // address_to_word
        const Word contract_w = word_of(fr_from_u128(fr_lo64(contract), fr_hi64of128(contract)), fr_u((u64)contract.v[4]));
        {
            RwQ W;
            rwq_init(W, 1, TG_TxAccessListAccount);
            rwq_set(W, R_ID, tx_id);
            rwq_set(W, R_ADDR, contract);
            rwq_set_word(W, R_VAL_LO, word_of(fr_u(1), fr_zero()));
            u32 r; r = rw_lookup(I, W); if (I.err) return;
            EV_TRY(value_of(I, rw_value_prev(I, r)));
        }
        Word callee_code_hash; callee_code_hash = account_read_word(I, contract, ACC_CodeHash); if (I.err) return;
        Fr callee_nonce;
        {
            RwQ Q;
            rwq_init(Q, 0, TG_Account);
            rwq_set(Q, R_ADDR, contract);
            rwq_set(Q, R_FT, fr_u(ACC_Nonce));
            u32 r; r = rw_lookup(I, Q); if (I.err) return;
            EV_TRY(callee_nonce = value_of(I, rw_value(I, r)));
        }
        I.seq += 2;  // Word(EMPTY_CODE_HASH), Word(0)
        const Word empty_hash = word_of(fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull), fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull));
        nac = fr_is_zero(callee_nonce) && (is_equal_word(callee_code_hash, empty_hash) || is_equal_word(callee_code_hash, word_zero()));
        if (nac) {
            Fr ret_addr; EV_TRY(ret_addr = word_to_fq(I, ret_addr_w, 20));
            constrain_equal(I, ret_addr, fr_mul(is_success, contract)); if (I.err) return;
            Reversion crv; EV_TRY(crv = reversion_info(I, &callee_call_id));
            constrain_equal(I, crv.persistent, fr_mul(rv.persistent, is_success)); if (I.err) return;
            EV_TRY(balance_move(I, caller, value_w, crv, true));
            EV_TRY(balance_move(I, contract, value_w, crv, false));
            {
                RwQ Q;
                rwq_init(Q, 1, TG_Account);
                rwq_set(Q, R_ADDR, contract);
                rwq_set(Q, R_FT, fr_u(ACC_Nonce));
                u32 r; r = rw_lookup(I, Q); if (I.err) return;
                Fr n2; EV_TRY(n2 = value_of(I, rw_value(I, r)));
                EV_TRY(value_of(I, rw_value_prev(I, r)));
                constrain_equal(I, n2, fr_u(1)); if (I.err) return;
            }
            if (has_init_code) {
                const Word next_hash = word_of(ev_next(I, S_CH_LO), ev_next(I, S_CH_HI));
                CopyRes cr;
                EV_TRY(cr = copy_lookup(I, word_value(I.call_id), CDT_Memory, next_hash, CDT_Bytecode, offset, fr_add(offset, size), fr_zero(),
                                        size, fr_add_u64(I.rwc, I.rw_off)));
                if (!(fr_fits64(cr.rwc_inc) && fr_lo64(cr.rwc_inc) < (1ull << 62))) { if (I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq); return; }
                I.rw_off += fr_lo64(cr.rwc_inc);
                Fr code_size; code_size = bytecode_length(I, next_hash, true);
                constrain_equal(I, code_size, size); if (I.err) return;
                {
                    const u32 tags[5] = {CC_ProgramCounter, CC_StackPointer, CC_GasLeft, CC_MemorySize, CC_ReversibleWriteCounter};
                    for (int k = 0; k < 5; k++) {
                        Fr want;
                        switch (k) {
                        case 0: want = fr_add_u64(I.pc, 1); break;
                        case 1: want = fr_add_u64(I.sp, (u64)sp_delta); break;
                        case 2: want = fr_sub(fr_sub(gas_left, gas_cost), callee_gas_left); break;
                        case 3: want = next_mem; break;
                        default: want = fr_add_u64(ev_curr(I, S_REV), 1); break;
                        }
                        Fr v; v = call_context_lookup(I, tags[k], 1);
                        constrain_equal(I, v, want); if (I.err) return;
                    }
                }
                {
                    const u32 tags[10] = {CC_CallerId, CC_TxId, CC_Depth, CC_CallerAddress, CC_CalleeAddress, CC_IsSuccess, CC_IsStatic, CC_IsRoot,
                                          CC_IsCreate, CC_CodeHash};
                    for (int k = 0; k < 10; k++) {
                        Word want = word_zero();
                        switch (k) {
                        case 0: want = word_value(I.call_id); break;
                        case 1: want = word_value(tx_id); break;
                        case 2: want = word_value(fr_add_u64(depth, 1)); break;
                        case 3: want = caller_w.w; break;
                        case 4: want = contract_w; break;
                        case 5: want = word_value(is_success); break;
                        case 8: want = word_value(fr_u(1)); break;
                        case 9: want = code_hash; break;
                        default: break;
                        }
                        WordOrValue got; got = call_context_lookup_word(I, tags[k], 0, &callee_call_id);
                        constrain_equal_word(I, got.w, want); if (I.err) return;
                    }
                }
                transition(I, S_RWC, t_delta(fr_u(I.rw_off)));
                transition(I, S_CALL_ID, t_to(callee_call_id));
                transition(I, S_IS_ROOT, t_to(fr_zero()));
                transition(I, S_IS_CREATE, t_to(fr_u(1)));
                I.seq++;  // code_hash = Transition.to_word(next.code_hash)
                transition(I, S_GAS, t_to(callee_gas_left));
                transition(I, S_REV, t_to(fr_u(3)));
                transition(I, S_LOG, t_same());
                transition(I, S_PC, t_to(fr_zero()));
                transition(I, S_SP, t_to(fr_u(1024)));
                transition(I, S_MWS, t_to(fr_zero()));
                if (I.err) return;
            }
        }
    }
    if (!precheck_ok || !nac || !has_init_code) {
        if (!precheck_ok || !nac) { constrain_equal(I, is_success, fr_zero()); if (I.err) return; }
        const u32 tags[3] = {CC_LastCalleeId, CC_LastCalleeReturnDataOffset, CC_LastCalleeReturnDataLength};
        for (int k = 0; k < 3; k++) {
            Fr v; v = call_context_lookup(I, tags[k], 1);
            constrain_equal(I, v, fr_zero()); if (I.err) return;
        }
        transition(I, S_RWC, t_delta(fr_u(I.rw_off)));
        transition(I, S_PC, t_delta_i(1));
        transition(I, S_SP, t_delta_i(sp_delta));
        transition(I, S_REV, t_int((nac && !has_init_code) ? 3 : 0));
        transition(I, S_GAS, t_delta(fr_neg(gas_cost)));
        transition(I, S_MWS, t_to(next_mem));
        transition(I, S_CALL_ID, t_same());
        transition(I, S_IS_ROOT, t_same());
        transition(I, S_IS_CREATE, t_same());
        ev_require(I, fr_eq(ev_next(I, S_CH_LO), ev_curr(I, S_CH_LO)) && fr_eq(ev_next(I, S_CH_HI), ev_curr(I, S_CH_HI)));
    }
}

// ---- DATACOPY precompile (dataCopy.py) and ErrorOutOfGasPrecompile (precompiles/error_oog_precompile.py) ----
// memory_copier_gas_cost(length, 0, per_word) (instruction.py:1183-1192)
ZK_HD Fr copier_gas_only(Ins& I, const Fr& length, u32 per_word) {
    Fr words = constant_divmod_shift(I, fr_add_u64(length, 31), 5, 4);
    Fr gas = fr_mul_u64(words, per_word);
    range_check(I, gas, 8);
    return gas;
}
ZK_HD void g_datacopy(Ins& I, Tail& T) {
    WordOrValue aw; aw = call_context_lookup_word(I, CC_CalleeAddress);
    Fr address; EV_TRY(address = word_to_fq(I, aw.w, 20));
    fixed_lookup(I, FX_PrecompileInfo, ev_curr(I, S_STATE), address, fr_u(15)); if (I.err) return;
    Fr caller_id, cd_offset, cd_length, rd_offset, rd_length;
    caller_id = call_context_lookup(I, CC_CallerId);
    cd_offset = call_context_lookup(I, CC_CallDataOffset);
    cd_length = call_context_lookup(I, CC_CallDataLength);
    rd_offset = call_context_lookup(I, CC_ReturnDataOffset);
    rd_length = call_context_lookup(I, CC_ReturnDataLength);
    if (I.err) return;
    const Fr size = cd_length;
    Fr copier; EV_TRY(copier = copier_gas_only(I, cd_length, 3));
    const Fr gas_cost = fr_add_u64(copier, 15);
    // the first copy's `length` argument really is return_data_offset + return_data_length (:41-51)
    CopyRes cr;
    EV_TRY(cr = copy_lookup(I, word_value(caller_id), CDT_Memory, word_value(caller_id), CDT_Memory, cd_offset, fr_add(cd_offset, size), rd_offset,
                            fr_add(rd_offset, rd_length), fr_add_u64(I.rwc, I.rw_off)));
    CopyRes cr2;
    EV_TRY(cr2 = copy_lookup(I, word_value(caller_id), CDT_Memory, word_value(I.call_id), CDT_Memory, cd_offset, fr_add(cd_offset, size), fr_zero(),
                             rd_length, fr_add(fr_add_u64(I.rwc, I.rw_off), cr.rwc_inc)));
    if (!(fr_fits64(size) && fr_lo64(size) < (1ull << 60))) { if (I.err == 0u) I.err = ZK_CODE(ZK_UNSUPPORTED, I.seq); return; }
    I.rw_off += 4 * fr_lo64(size);
    restore_context(I, fr_u(I.rw_off), fr_sub(ev_curr(I, S_GAS), gas_cost), fr_zero(), size, &caller_id);
}
// ---- precompile states reading the sig / ecc tables (execution/precompiles/*.py)
// RLC(bytes(reversed(data)), r, n_bytes = len(data)).expr() == Horner over `data` front to back
struct RlcAcc {
    Fr acc, rM;
};
ZK_HD RlcAcc rlc_begin(const Fr& r) {
    RlcAcc a;
    a.acc = fr_zero();
    a.rM = fr_to_mont(r);
    return a;
}
ZK_HD void rlc_le_bytes32(RlcAcc& a, const U256& v) {  // the 32 little-endian bytes of v, in order
    for (int k = 0; k < 32; k++) a.acc = fr_add(fr_mulc(a.acc, a.rM), fr_u(fr_byte(v, k)));
}
ZK_HD void precompile_prelude(Ins& I, u64 base_gas, Fr& is_success, Fr* calldata_len) {
    is_success = call_context_lookup(I, CC_IsSuccess);
    if (calldata_len) *calldata_len = call_context_lookup(I, CC_CallDataLength);
    WordOrValue aw; aw = call_context_lookup_word(I, CC_CalleeAddress);
    if (I.err) return;
    Fr address; address = word_to_fq(I, aw.w, 20); if (I.err) return;
    fixed_lookup(I, FX_PrecompileInfo, ev_curr(I, S_STATE), address, fr_u(base_gas));
}
ZK_HD void g_ecrecover(Ins& I, Tail& T) {  // precompiles/ecrecover.py:26-94
    Fr is_success; precompile_prelude(I, 3000, is_success, nullptr); if (I.err) return;
    if (!aux_expect(I, AUX_ECRECOVER, 12)) return;
    const Word msg_hash = aux_word(I, 0), sig_v = aux_word(I, 2), sig_r = aux_word(I, 4), sig_s = aux_word(I, 6);
    const Fr recovered_addr = aux_cell(I, 8), rand = aux_cell(I, 11);
    const bool is_recovered = !fr_is_zero(recovered_addr);
    RlcAcc in = rlc_begin(rand);
    {
        const Word* ws[4] = {&msg_hash, &sig_v, &sig_r, &sig_s};
        U256 vals[4];
        for (int k = 0; k < 4; k++) { vals[k] = int_bytes32(I, *ws[k]); if (I.err) return; }  // int_value().to_bytes(32, "little")
        for (int k = 0; k < 4; k++) rlc_le_bytes32(in, vals[k]);
    }
    constrain_equal(I, aux_cell(I, 9), in.acc); if (I.err) return;
    RlcAcc out = rlc_begin(rand);
    rlc_le_bytes32(out, recovered_addr);
    constrain_equal(I, aux_cell(I, 10), out.acc); if (I.err) return;
    constrain_equal(I, is_success, fr_u(1)); if (I.err) return;
    const Fr n_limbs = {SECP_N_LIMBS};
    const Word n_word = word_from_u256(n_limbs);
    u32 r_ub, s_ub, eq;
    compare_word(I, sig_r, n_word, r_ub, eq); if (I.err) return;
    compare_word(I, sig_s, n_word, s_ub, eq); if (I.err) return;
    const u32 r_nz = 1u - is_zero_word(sig_r), s_nz = 1u - is_zero_word(sig_s);
    const bool valid_r_s = r_ub + s_ub + r_nz + s_nz == 4u;
    const bool valid_v = is_equal_word(sig_v, word_value(fr_u(27))) + is_equal_word(sig_v, word_value(fr_u(28))) == 1u;
    if (valid_r_s && valid_v) {
        sig_lookup(I, msg_hash, fr_sub_u64(sig_v.lo, 27), sig_r, sig_s, recovered_addr, fr_u(is_recovered ? 1 : 0)); if (I.err) return;
    } else {
        ev_require(I, !is_recovered); if (I.err) return;
        constrain_zero(I, recovered_addr); if (I.err) return;
    }
    restore_context(I, fr_u(I.rw_off), fr_sub_u64(ev_curr(I, S_GAS), 3000), fr_zero(), fr_u(is_recovered ? 32 : 0));
}
ZK_HD void g_ecadd_ecmul(Ins& I, Tail& T, bool is_mul) {  // precompiles/ecadd.py:10-48, ecmul.py:10-55
    const u64 gas = is_mul ? 6000 : 150;
    Fr is_success; precompile_prelude(I, gas, is_success, nullptr); if (I.err) return;
    if (!aux_expect(I, is_mul ? AUX_ECMUL : AUX_ECADD, is_mul ? 8 : 10)) return;
    const Word px = aux_word(I, 0), py = aux_word(I, 2), qx = aux_word(I, 4);  // qx: the scalar for ecMul
    const Word qy = is_mul ? word_zero() : aux_word(I, 6);
    const Fr outx = aux_cell(I, is_mul ? 6 : 8), outy = aux_cell(I, is_mul ? 7 : 9);
    const bool zero_word = [](const Word& w) { return fr_is_zero(w.lo) && fr_is_zero(w.hi); }(qx);
    const bool p_inf = fr_is_zero(px.lo) && fr_is_zero(px.hi) && fr_is_zero(py.lo) && fr_is_zero(py.hi);
    if (fr_is_zero(is_success) || (is_mul && (zero_word || p_inf))) {
        constrain_zero(I, outx); if (I.err) return;
        constrain_zero(I, outy); if (I.err) return;
    }
    ecc_lookup(I, is_mul ? 2u : 1u, px, py, qx, qy, fr_zero(), outx, outy, is_success); if (I.err) return;
    const bool ok = fr_eq_u64(is_success, 1);
    restore_context(I, fr_u(I.rw_off), ok ? fr_sub_u64(ev_curr(I, S_GAS), gas) : fr_zero(), fr_zero(), fr_u(ok ? 64 : 0));
}
ZK_HD void g_ecpairing(Ins& I, Tail& T) {  // precompiles/ecpairing.py:13-78
    Fr is_success, calldata_len; precompile_prelude(I, 45000, is_success, &calldata_len); if (I.err) return;
    if (!aux_expect(I, AUX_ECPAIRING, 4)) return;
    const Fr input_rlc = aux_cell(I, 0), input_pairs = aux_cell(I, 1), is_valid_input = aux_cell(I, 2), output = aux_cell(I, 3);
    constrain_equal(I, is_success, is_valid_input); if (I.err) return;
    if (fr_mod_small(calldata_len, 192) != 0u) {
        constrain_equal(I, output, fr_zero()); if (I.err) return;
        constrain_equal(I, is_valid_input, fr_zero()); if (I.err) return;
    } else {
        constrain_equal(I, calldata_len, fr_mul_u64(input_pairs, 192)); if (I.err) return;
        if (fr_is_zero(calldata_len)) {
            constrain_zero(I, input_pairs); if (I.err) return;
            constrain_zero(I, input_rlc); if (I.err) return;
            constrain_equal(I, output, fr_u(1)); if (I.err) return;
        }
    }
    ecc_lookup(I, 3u, word_zero(), word_zero(), word_zero(), word_zero(), input_rlc, fr_zero(), output, is_valid_input); if (I.err) return;
    const Fr gas_left = fr_eq_u64(is_success, 1) ? fr_sub(fr_sub_u64(ev_curr(I, S_GAS), 45000), fr_mul_u64(input_pairs, 34000)) : fr_zero();
    restore_context(I, fr_u(I.rw_off), gas_left, fr_zero(), fr_u(fr_eq_u64(is_valid_input, 1) ? 32 : 0));
}
// [tx_calldata_lookup(tx_id, FQ(idx)) for idx in range(n)] (instruction.py:694-699): number of bytes and of
// non-zero bytes; ends at the first missing row (LookupUnsatFailure), so the walk is bounded by the tx table
ZK_HD void tx_calldata_scan(Ins& I, const Fr& tx_id, const Fr& n, u64& len, u64& nz) {
    len = 0;
    nz = 0;
    for (Fr idx = fr_zero(); fr_lt(idx, n); idx = fr_add_u64(idx, 1)) {
        WordOrValue v; v = tx_lookup(I, tx_id, TXC_CallData, &idx); if (I.err) return;
        Fr b; b = value_of(I, v); if (I.err) return;
        len++;
        if (!fr_is_zero(b)) nz++;
    }
}
ZK_HD void g_error_oog_create(Ins& I, Tail& T) {  // error_oog_create.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const bool is_create2 = fr_eq_u64(opcode, OP_CREATE2);
    ev_require(I, is_create2 || fr_eq_u64(opcode, OP_CREATE)); if (I.err) return;
    Word ow, sw; ow = stack_lookup(I, 0, 1); sw = stack_lookup(I, 0, 2); if (I.err) return;
    Fr offset, size; EV_TRY(memory_offset_and_length(I, ow, sw, offset, size));
    Fr is_root; is_root = call_context_lookup(I, CC_IsRoot); if (I.err) return;
    Fr gas_cost;
    if (fr_eq_u64(is_root, 1)) {  // creation tx: 53000 + 16 / 4 per non-zero / zero calldata byte
        Fr tx_id; tx_id = call_context_lookup(I, CC_TxId); if (I.err) return;
        u64 len, nz; EV_TRY(tx_calldata_scan(I, tx_id, size, len, nz));
        gas_cost = fr_u(53000u + nz * 16u + (len - nz) * 4u);
    } else {
        Fr next_size, gas; EV_TRY(memory_expansion(I, offset, size, next_size, gas));
        gas_cost = fr_add_u64(gas, 32000);
    }
    Fr word_size; EV_TRY(word_size = constant_divmod_shift(I, fr_add_u64(size, 31), 5, 4));
    gas_cost = fr_add(gas_cost, fr_mul_u64(word_size, is_create2 ? 8 : 2));  // EIP-3860 init-code words (+ CREATE2 hashing)
    u32 exceeds, insufficient, eq;
    EV_TRY(ev_compare(I, fr_u(49152), size, 8, exceeds, eq));  // MAX_INIT_CODE_SIZE
    EV_TRY(ev_compare(I, ev_curr(I, S_GAS), gas_cost, 8, insufficient, eq));
    ev_require(I, insufficient + exceeds != 0u); if (I.err) return;
    T.err_tail = 1;
}
// calc_mem_size64_with_uint / calc_mem_size64 / memory_size (instruction.py:1198-1327)
ZK_HD void calc_mem_size64_with_uint(Ins& I, const Word& offset_w, const Fr& length64, Fr& val, u32& over) {
    val = fr_zero();
    over = 0;
    if (fr_is_zero(length64)) return;
    Fr offset; EV_TRY(offset = word_to_fq(I, offset_w, 31));
    if (!fr_fits64(offset)) { over = 1; return; }
    Fr offset64; EV_TRY(offset64 = word_to_fq(I, offset_w, 5));
    val = fr_add(offset64, length64);
    over = fr_lt(val, offset64) ? 1u : 0u;
}
ZK_HD void calc_mem_size64(Ins& I, const Word& offset_w, const Word& length_w, Fr& val, u32& over) {
    val = fr_zero();
    over = 0;
    Fr len; EV_TRY(len = word_to_fq(I, length_w, 31));
    if (!fr_fits64(len)) { over = 1; return; }
    calc_mem_size64_with_uint(I, offset_w, len, val, over);
}
// false: the opcode is not one memory_size knows (it returns None there)
ZK_HD bool memory_size(Ins& I, u32 op, Fr& val, u32& over) {
    val = fr_zero();
    over = 0;
    Word a, b;
    switch (op) {
    case OP_SHA3: case OP_RETURN: case OP_REVERT: case OP_LOG0: case OP_LOG0 + 1: case OP_LOG0 + 2: case OP_LOG0 + 3: case OP_LOG4:
        a = stack_pop(I); b = stack_pop(I);
        calc_mem_size64(I, a, b, val, over);
        return true;
    case OP_CALLDATACOPY: case OP_RETURNDATACOPY: case OP_CODECOPY:
        stack_pop(I); a = stack_pop(I); b = stack_pop(I);
        calc_mem_size64(I, a, b, val, over);
        return true;
    case OP_EXTCODECOPY:
        stack_pop(I); stack_pop(I); a = stack_pop(I); b = stack_pop(I);
        calc_mem_size64(I, a, b, val, over);
        return true;
    case OP_MLOAD:
        a = stack_pop(I);
        calc_mem_size64_with_uint(I, a, fr_u(32), val, over);
        return true;
    case OP_MSTORE: case OP_MSTORE8:
        a = stack_pop(I); stack_pop(I);
        calc_mem_size64_with_uint(I, a, fr_u(32), val, over);
        return true;
    case OP_CREATE: case OP_CREATE2:
        stack_pop(I); a = stack_pop(I); b = stack_pop(I);
        if (op == OP_CREATE2) stack_pop(I);
        calc_mem_size64(I, a, b, val, over);
        return true;
    case OP_CALL: case OP_CALLCODE: case OP_DELEGATECALL: case OP_STATICCALL: {
        if (op == OP_CALL || op == OP_CALLCODE) stack_pop(I);
        stack_pop(I); stack_pop(I);
        Word cd_off, cd_len; cd_off = stack_pop(I); cd_len = stack_pop(I);
        a = stack_pop(I); b = stack_pop(I);
        Fr x, y;
        calc_mem_size64(I, a, b, x, over);
        if (I.err || over) { val = fr_zero(); return true; }
        calc_mem_size64(I, cd_off, cd_len, y, over);
        if (I.err || over) { val = fr_zero(); return true; }
        val = fr_lt(y, x) ? x : y;
        return true;
    }
    default: return false;
    }
}
ZK_HD void g_error_gas_uint_overflow(Ins& I, Tail& T) {  // error_gas_uint_overflow.py
    Fr opcode; opcode = opcode_lookup(I, true);
    const u32 op = fr_le_u64(opcode, 255) ? opcode.v[0] : 0x100u;
    const bool is_create = op == OP_CREATE || op == OP_CREATE2;
    Fr calldata_length, tx_id, is_root;
    calldata_length = call_context_lookup(I, CC_CallDataLength);
    tx_id = call_context_lookup(I, CC_TxId);
    is_root = call_context_lookup(I, CC_IsRoot);
    if (I.err) return;
    u32 calldata_over = 0, initcode_over = 0, eq;
    if (fr_eq_u64(is_root, 1)) {  // intrinsic gas of the tx calldata
        u64 len, nz; EV_TRY(tx_calldata_scan(I, tx_id, calldata_length, len, nz));
        if (len > 0) {
            u64 gas = is_create ? 53000u : 21000u;
            u32 nz_over, z_over = 0;
            EV_TRY(ev_compare(I, fr_u((~0ull - gas) / 16u), fr_u(nz), 8, nz_over, eq));
            gas += nz * 16u;
            if (nz_over == 0u) {
                const u64 z = len - nz;
                EV_TRY(ev_compare(I, fr_u((~0ull - gas) / 4u), fr_u(z), 8, z_over, eq));
                gas += z * 4u;
            }
            if (is_create) {
                Fr len_words; EV_TRY(len_words = constant_divmod_shift(I, fr_u(len + 31u), 5, 8));
                EV_TRY(ev_compare(I, fr_u((~0ull - gas) / 2u), len_words, 8, initcode_over, eq));
            }
            calldata_over = nz_over + z_over;
        }
    }
    // `if is_dynamic_gas:` (:149): an FQ is always truthy, so memory_size runs for every opcode and unpacking
    // its None for the opcodes it does not list raises TypeError
    Fr mem; u32 size_over;
    const bool listed = memory_size(I, op, mem, size_over); if (I.err) return;
    if (!listed) { ev_require(I, false, ZK_TYPE_ERROR); return; }
    // to_word_size + safe_mul (:1329-1336): words * 32 exceeds u64 exactly when mem_size > MAX_U64 - 31
    const u32 mul_over = (!fr_fits64(mem) || fr_lo64(mem) > ~0ull - 31u) ? 1u : 0u;
    ev_require(I, size_over + mul_over + calldata_over + initcode_over != 0u); if (I.err) return;
    T.err_tail = 1;
}
ZK_HD void g_error_oog_precompile(Ins& I, Tail& T) {
    WordOrValue aw; aw = call_context_lookup_word(I, CC_CalleeAddress);
    Fr address; EV_TRY(address = word_to_fq(I, aw.w, 20));
    Fr calldata_len; calldata_len = call_context_lookup(I, CC_CallDataLength); if (I.err) return;
    const bool is_pre = fr_fits64(address) && fr_lo64(address) >= 1 && fr_lo64(address) <= 9;
    ev_require(I, is_pre); if (I.err) return;
    static const uint16_t pgas[10] = {0, 3000, 60, 600, 15, 0, 150, 6000, 45000, 0};
    const u32 av = address.v[0];
    Fr gas_cost = fr_u(pgas[av]);
    if (av == 8u) {  // BN254PAIRING: pairs = calldata_len / 192 in the field
        gas_cost = fr_add(gas_cost, fr_mul_u64(fr_mulc(calldata_len, frm_inv192()), 34000));
    } else if (av == 4u) {  // DATACOPY
        Fr copier; EV_TRY(copier = copier_gas_only(I, calldata_len, 3));
        gas_cost = fr_add(gas_cost, copier);
    } else {
        // the base cost stays a plain int and compare() calls .expr() on it (:33): AttributeError for every other
        // precompile -- after compare's own check of the left operand (instruction.py:447-451)
        I.seq++;
        ev_fail(I, fr_fits64(ev_curr(I, S_GAS)) ? ZK_ATTRIBUTE_ERROR : ZK_ASSERT);
        return;
    }
    oog_tail(T, gas_cost);
}

// ExecutionState transition constraint (instruction.py:189-204)
ZK_HD bool state_bit(u64 lo, u64 hi, u32 state) {  // bit `state` of a 128-bit immediate
    return state < 64 ? ((lo >> state) & 1ull) : (state < 128 ? ((hi >> (state - 64)) & 1ull) : 0ull);
}
ZK_HD bool state_transition_ok(u32 curr, u32 next) {
    if (curr == ES_EndTx && !(next == ES_BeginTx || next == ES_EndBlock)) return false;
    if (curr == ES_EndBlock && next != ES_EndBlock) return false;
    if (next == ES_BeginTx) return curr == ES_EndTx;
    if (next == ES_EndTx) return (curr < ES_COUNT && state_bit(ZK_STATE_HALTS_MASK_LO, ZK_STATE_HALTS_MASK_HI, curr)) || curr == ES_BeginTx;
    if (next == ES_EndBlock) return curr == ES_EndTx || curr == ES_EndBlock;
    return true;
}

#define GROUP_OF_ES_ADD 2
#define GROUP_OF_ES_SAR 2
#define GROUP_OF_ES_SDIV_SMOD 1
#define GROUP_OF_ES_ADDMOD 1
#define GROUP_OF_ES_ADDRESS 2
#define GROUP_OF_ES_BITWISE 2
#define GROUP_OF_ES_BYTE 2
#define GROUP_OF_ES_BlockCtx 4
#define GROUP_OF_ES_CALLDATASIZE 2
#define GROUP_OF_ES_CALLER 2
#define GROUP_OF_ES_CALLVALUE 2
#define GROUP_OF_ES_CMP 2
#define GROUP_OF_ES_CODESIZE 2
#define GROUP_OF_ES_GAS 2
#define GROUP_OF_ES_GASPRICE 4
#define GROUP_OF_ES_ISZERO 2
#define GROUP_OF_ES_JUMP 2
#define GROUP_OF_ES_JUMPI 2
#define GROUP_OF_ES_MEMORY 0
#define GROUP_OF_ES_MSIZE 2
#define GROUP_OF_ES_MUL 1
#define GROUP_OF_ES_MULMOD 1
#define GROUP_OF_ES_NOT 2
#define GROUP_OF_ES_ORIGIN 4
#define GROUP_OF_ES_POP 2
#define GROUP_OF_ES_PUSH 2
#define GROUP_OF_ES_RETURNDATASIZE 2
#define GROUP_OF_ES_SCMP 2
#define GROUP_OF_ES_SELFBALANCE 2
#define GROUP_OF_ES_SHL_SHR 1
#define GROUP_OF_ES_SIGNEXTEND 2
#define GROUP_OF_ES_SLOAD 0
#define GROUP_OF_ES_SSTORE 0
#define GROUP_OF_ES_STOP 0

// Kernel specialisation groups: the step pairs are sorted by (group, state) and each group is
// evaluated by its own kernel instantiation, which contains only that group's gadget bodies
// (smaller instruction footprint, fewer live registers -> more resident wavefronts).
// numbered in launch order: the long-running gadgets get the first lanes so that their wavefronts start
// first and the short ones fill in behind them (no long tail at the end of the kernel)
// EVM_GROUP_COLD: states outside BASELINE config 3's opcode mix (error states, copy / account-access
// gadgets, EXP, SDIV/SMOD, RETURN, LOG, EndBlock).  They get their own kernel instantiation so that
// their code and register pressure stay out of the hot kernel (EVM_GROUP_ALL = the three hot groups).
// EVM_GROUP_WARM (round 3): the copy- / keccak- / exp-table gadgets (SHA3, the *COPY family, LOG, EXP).  A block has a few thousand of
// these steps (BASELINE configs[4]); inside the cold instantiation — every error state, the CALL family, CREATE, Begin / EndTx in one
// function: 512 registers + 2 KB of scratch — a wavefront of them took ~350 us.  Their own, much smaller instantiation keeps the
// block's pass bound by the State kernel.
enum { EVM_GROUP_MEM = 0, EVM_GROUP_MUL = 1, EVM_GROUP_LIGHT = 2, EVM_GROUP_WARM = 3, EVM_GROUP_COLD = 4, EVM_N_GROUPS = 5, EVM_GROUP_ALL = -1 };
#define ZK_NOT_MINE 0xffffffffu  // evm_check_step<G>: the state belongs to the other instantiation
#define ZK_DEFERRED_BASE 0xfffffff0u  // evm_check_step (EVM_FAST build) returns BASE + reason (1 wide word cell, 2 generic probe, 3 no packed
                                     // keys, 4 key cells beyond the packed widths, 5 wide transition operands; 8 = unstaged pair): the pair needs a
                                     // fallback path of the general build
ZK_HD int evm_state_group(u32 state) {
    switch (state) {
    case ES_EXP: case ES_SHA3: case ES_CODECOPY: case ES_CALLDATACOPY: case ES_RETURNDATACOPY: case ES_EXTCODECOPY: case ES_LOG: return EVM_GROUP_WARM;
    case ES_MUL: case ES_SHL_SHR: case ES_ADDMOD: case ES_MULMOD: return EVM_GROUP_MUL;
    case ES_MEMORY: case ES_SLOAD: case ES_SSTORE: case ES_STOP: return EVM_GROUP_MEM;
    case ES_SDIV_SMOD: case ES_BALANCE: case ES_EXTCODESIZE: case ES_EXTCODEHASH: case ES_BLOCKHASH:
    case ES_CALLDATALOAD: case ES_ErrorInvalidOpcode: case ES_ErrorStack: case ES_ErrorOutOfGasConstant:
    case ES_ErrorInvalidJump: case ES_ErrorOutOfGasStaticMemoryExpansion: case ES_ErrorOutOfGasDynamicMemoryExpansion:
    case ES_ErrorOutOfGasMemoryCopy: case ES_ErrorOutOfGasAccountAccess: case ES_ErrorOutOfGasLOG: case ES_ErrorOutOfGasEXP:
    case ES_ErrorOutOfGasSHA3: case ES_ErrorReturnDataOutOfBound: case ES_ErrorWriteProtection: case ES_RETURN:
    case ES_ErrorInvalidCreationCode: case ES_ErrorMaxCodeSizeExceeded: case ES_ErrorOutOfGasCodeStore: case ES_EndBlock: case ES_EndTx: case ES_BeginTx: case ES_CALL_OP:
    case ES_ErrorOutOfGasCall: case ES_ErrorOutOfGasSloadSstore: case ES_CREATE: case ES_CREATE2: case ES_DATACOPY:
    case ES_ErrorOutOfGasPrecompile: case ES_ErrorOutOfGasCREATE: case ES_ErrorGasUintOverflow: case ES_ECRECOVER: case ES_BN254_ADD:
    case ES_BN254_SCALAR_MUL: case ES_BN254_PAIRING: return EVM_GROUP_COLD;
    // readers of the tx / block tables (generic open-addressing lookups: not in the fast build)
    case ES_ORIGIN: case ES_GASPRICE: case ES_BlockCtx: return EVM_GROUP_COLD;
    default: return EVM_GROUP_LIGHT;
    }
}
// sort bin of a state.  The cold states keep their own 128 bins at the end (the cold kernel's lane range starts at
// group_start[EVM_GROUP_COLD]); the hot kernel evaluates every other state, so the first 384 bins are ONE list ordered by
// measured wavefront time, longest first (tools/evm_wave_timeline.py): a longest-processing-time-first schedule over the
// 2 x 1024 wavefront slots, with the short POP / STOP wavefronts making the kernel's tail.
ZK_HD u32 evm_state_bin(u32 state) {
    const int grp = evm_state_group(state);
    if (grp >= EVM_GROUP_WARM) return (u32)grp * 128u + (state & 127u);  // warm / cold: their own bin ranges (not padded)
    u32 key = (state & 127u) + 32u;  // everything not listed below, in state order (arbitrary cell values stay below 384)
    switch (state) {
    case ES_STOP: key = 0; break;  // non-root STOP restores the caller's context: a dozen lookups, the longest wavefronts
    case ES_ADDMOD: key = 1; break;
    case ES_MULMOD: key = 2; break;
    case ES_MEMORY: key = 3; break;
    case ES_SSTORE: key = 4; break;
    case ES_SLOAD: key = 5; break;
    case ES_MUL: key = 6; break;
    case ES_SHL_SHR: key = 7; break;
    case ES_PUSH: key = 8; break;
    case ES_BITWISE: key = 9; break;
    case ES_CMP: key = 10; break;
    case ES_SCMP: key = 11; break;
    case ES_ADD: key = 12; break;
    case ES_BYTE: key = 13; break;
    case ES_SIGNEXTEND: key = 14; break;
    case ES_NOT: key = 15; break;
    case ES_ISZERO: key = 16; break;
    case ES_POP: key = 383; break;
    default: break;
    }
    return key;
}
#define EVM_N_BINS (EVM_N_GROUPS * 128)
#define EVM_NO_PAIR 0xffffffffu                   // a pad lane of the sorted mapping (hot bins are padded to whole wavefronts)
#define EVM_PERM_PAD (64u * EVM_GROUP_WARM * 128u)  // upper bound of the padding (the hot bins)

// Request the packed bytecode record at curr.program_counter (every hot gadget's opcode_lookup) before the common checks
// and the gadget prologue run, so that its HBM round trip overlaps them.  (Requesting the leading RW rows the same way was
// tried in round 2: three rows are 72 registers held across the gadget switch, and the merged kernel has none to spare.)
ZK_HD void evm_prefetch(Ins& I) {
    const EvmArgs& a = *I.a;
    code_dir_resolve(I, curr_code_hash(I));
    const uint16_t* packed = a.codes.packed;
    if (I.code_state == 1u && packed != nullptr && fr_fits64(I.pc) && fr_lo64(I.pc) < (u64)I.code_n_bytes) {
        I.pre_op = packed[I.code_byte_base + (u32)fr_lo64(I.pc)];
        I.pre_op_ok = true;
    }
}

// verify_step (main.py:47-63) for pair `idx`; G selects which gadget bodies are compiled in
#if !defined(ZK_HOSTSIM)
// (Round 2 staged a pair from its 832 bytes of step-table rows, 52 chunks by lane quads: kernel 78.8 -> 76.9 us against a
// lane-by-lane fill; the wavefront walking its 64 x 832 B in address order was also tried — ~110 instructions of address /
// scatter arithmetic per chunk, 40-55k cycles.  Neither changes what the first round costs: 2,048 resident wavefronts asking
// for 53 KB each at t = 0 is 109 MB at HBM rate, ~37k cycles.  Round 3 stages from packed step records instead.)
// ---- step records (round 3) ----------------------------------------------------------------------------------------------
// A well-formed step's 13 cells (416 B on the wire) hold 40 + 8 + 32 = 80 bytes of information: ten 32-bit integers, the
// 64-bit gas_left and the two 128-bit code-hash cells.  zk_evm_open packs every step into a 96-byte record (one streaming
// read of the step table, riding on the open launch); the hot kernel then stages a pair (curr, next) from 192 contiguous
// bytes instead of 832: the staging phase was 14k-36k of a wavefront's 33k-130k cycles and its first round (2,048 wavefronts
// asking for 53 KB each at t = 0) ran at HBM rate.  A step with a cell wider than its record field (malformed witnesses only)
// sets the record's wide flag: a pair touching it is not staged, its lane reads the step rows from HBM as before.
// Record layout (u32 words; chosen so that the two lanes that hold the low cell halves in the builder each store whole
// 16-byte chunks):   0 state  1 call_id  2 is_create  3 sp | 4 memory_word_size  5 log_id  6 WIDE  7 - | 8-11 code_hash.hi |
//                   12 rw_counter  13 is_root  14 pc  15 reversible_write_counter | 16-17 gas_left  18-19 - | 20-23 code_hash.lo
#define EVM_REC_WORDS 24
#define EVM_REC_WIDE_WORD 6
// stage entry of word i of record chunk cj for step s (0xff = not staged): see evm_stage_entry
ZK_HD u32 evm_rec_stage_entry(u32 s, u32 cj, u32 i) {
    const u32 step12 = s * 12u, ch = 24u + s * 8u;
    switch (cj) {
    case 0: return step12 + (i == 0 ? 0u : i == 1 ? 2u : i == 2 ? 4u : 6u);              // state, call_id, is_create, sp
    case 1: return i == 0 ? step12 + 7u : i == 1 ? step12 + 9u : 0xffu;                  // memory_word_size, log_id
    case 2: return ch + 4u + i;                                                          // code_hash.hi
    case 3: return step12 + (i == 0 ? 1u : i == 1 ? 3u : i == 2 ? 5u : 8u);              // rw_counter, is_root, pc, reversible_write_counter
    case 4: return i < 2 ? step12 + (u32)EVM_STAGE_GAS + i : 0xffu;                      // gas_left
    default: return ch + i;                                                              // code_hash.lo
    }
}
// Builder: four lanes share a step row (26 chunks of 16 bytes, lane q loads chunks q, q + 4, ...): lane 0 ends up with the low
// halves of cells 0, 2, ..., 12, lane 2 with those of cells 1, 3, ..., 11, lanes 1 and 3 with the high halves (all zero in a
// well-formed step).
ZK_HD u32 evm_quad_or(u32 v) {
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
    return v;
}
ZK_HD void evm_step_record_quad(const u64* steps, u32 n_steps, u32* recs, u32 vthread) {
    const u32 st = vthread >> 2, q = vthread & 3u;
    const bool in = st < n_steps;
    const uint4 Z = make_uint4(0u, 0u, 0u, 0u);
    uint4 c[7];
    const uint4* src = reinterpret_cast<const uint4*>(steps + (u64)(in ? st : 0u) * (STEP_NCELLS * 4)) + q;
#pragma unroll
    for (int k = 0; k < 7; k++) c[k] = (k < 6 || q < 2) ? src[4 * k] : Z;  // chunks 24, 25 exist for lanes 0, 1 only
    u32 bad = 0;
    if (q & 1u) {  // high halves
#pragma unroll
        for (int k = 0; k < 7; k++) bad |= c[k].x | c[k].y | c[k].z | c[k].w;
    } else if (q == 0) {  // cells 0 state, 2 call_id, 4 is_create, 6 code_hash.hi, 8 sp, 10 memory_word_size, 12 log_id
#pragma unroll
        for (int k = 0; k < 7; k++) if (k != 3) bad |= c[k].y | c[k].z | c[k].w;
    } else {              // cells 1 rw_counter, 3 is_root, 5 code_hash.lo, 7 pc, 9 gas_left, 11 reversible_write_counter
#pragma unroll
        for (int k = 0; k < 6; k++) if (k != 2) bad |= (k == 4 ? 0u : c[k].y) | c[k].z | c[k].w;
    }
    const u32 wide = evm_quad_or(bad) != 0u ? 1u : 0u;
    if (!in) return;
    uint4* out = reinterpret_cast<uint4*>(recs + (u64)st * EVM_REC_WORDS);
    if (q == 0) {
        out[0] = make_uint4(c[0].x, c[1].x, c[2].x, c[4].x);
        out[1] = make_uint4(c[5].x, c[6].x, wide, 0u);
        out[2] = c[3];
    } else if (q == 2) {
        out[3] = make_uint4(c[0].x, c[1].x, c[3].x, c[5].x);
        out[4] = make_uint4(c[4].x, c[4].y, 0u, 0u);
        out[5] = c[2];
    }
}
// Fill the wavefront's LDS stage (layout: EVM_STAGE_ENTRIES) with its 64 lanes' step pairs from the step records, by lane quads:
// a pair is records idx and idx + 1 = 192 contiguous bytes = 12 chunks; lane r of the quad takes chunks r, r + 4, r + 8 (round
// it: the quad's pair is 16 it + q) — twelve 16-byte loads per lane for the whole wavefront instead of fifty-two — and scatters
// their words into the pair's column of the stage.  Returns whether this lane's own pair touches a wide step (malformed
// witnesses only): that lane then reads its step rows from HBM.
ZK_HD bool evm_stage_steps_quad(const EvmArgs& a, u32 idx, bool mine, __attribute__((address_space(3))) u32* wave_stage) {
    const u32 lane = threadIdx.x & 63u, q = lane >> 2, r = lane & 3u;
    u32 wide_rounds = 0;  // bit it: the pair this quad fetched in round it is wide
    uint4 v[4][3];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 p = 16u * (u32)it + q;
        const u32 idx_p = (u32)__shfl((int)idx, (int)p);
        const bool on = __shfl(mine ? 1 : 0, (int)p) != 0;  // a lane without a pair: fetch pair 0, nobody reads that column
        const uint4* src = reinterpret_cast<const uint4*>(a.step_recs + (u64)(on ? idx_p : 0u) * EVM_REC_WORDS) + r;
#pragma unroll
        for (int k = 0; k < 3; k++) v[it][k] = src[4 * k];
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 p = 16u * (u32)it + q;  // the pair (stage column) of this quad in this round
        u32 bad = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const u32 j = r + 4u * (u32)k, s = j >= 6u ? 1u : 0u, cj = j - 6u * s;  // chunk j of the pair = chunk cj of record s
            const uint4 x = v[it][k];
            const u32 w[4] = {x.x, x.y, x.z, x.w};
            if (cj == 1u) bad |= x.z;  // EVM_REC_WIDE_WORD = word 2 of chunk 1
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 e = evm_rec_stage_entry(s, cj, (u32)i);
                if (e != 0xffu) wave_stage[e * EVM_STAGE_STRIDE + p] = w[i];
            }
        }
        if (evm_quad_or(bad) != 0u) wide_rounds |= 1u << it;
    }
    const u32 theirs = (u32)__shfl((int)wide_rounds, (int)(4u * (lane & 15u)));  // the quad that fetched this lane's pair
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ((theirs >> (lane >> 4)) & 1u) != 0u;
}
// The same from the step table's own rows (a pair = 832 contiguous bytes = 52 chunks, lane r of the quad takes chunks r, r + 4,
// ... r + 48; the even lanes write the low words into the pair's column, the odd ones only have to see zeros): used by sessions
// without step records — the one-shot entry zk_evm_verify, where one streaming pass over the step table to build the records
// (+34 us in the HBM-bound open launch at 2^18 steps) costs more than the single evaluation pass gains from them (7 us).
ZK_HD bool evm_stage_steps_rows_quad(const EvmArgs& a, u32 idx, bool mine, __attribute__((address_space(3))) u32* wave_stage) {
    const u32 lane = threadIdx.x & 63u, q = lane >> 2, r = lane & 3u, r2 = r >> 1;
    const bool hi_half = (r & 1u) != 0u;
    u32 wide_rounds = 0;  // bit it: the pair this quad fetched in round it is wide
    // all 52 loads first (round it: the quad's pair is 16 it + q), then the scatter: one dependent round trip for the lot
    uint4 v[4][STEP_NCELLS];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 p = 16u * (u32)it + q;
        const u32 idx_p = (u32)__shfl((int)idx, (int)p);
        const bool on = __shfl(mine ? 1 : 0, (int)p) != 0;  // a lane without a pair: fetch pair 0, nobody reads that column
        const uint4* src = (const uint4*)(a.steps + (u64)(on ? idx_p : 0u) * (STEP_NCELLS * 4)) + r;
#pragma unroll
        for (int k = 0; k < STEP_NCELLS; k++) v[it][k] = src[4 * k];
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 p = 16u * (u32)it + q;  // the pair (stage column) of this quad in this round
        u32 bad = 0;
#pragma unroll
        for (int k = 0; k < STEP_NCELLS; k++) {
            const uint4 x = v[it][k];
            if (hi_half) {
                bad |= x.x | x.y | x.z | x.w;
            } else {
                // cell 2k + r2 of the pair (0..12 curr, 13..25 next): both candidates are compile-time, r2 selects
                const int cA = 2 * k, cB = 2 * k + 1;
                const int eA = evm_stage_entry(cA / STEP_NCELLS, cA % STEP_NCELLS), eB = evm_stage_entry(cB / STEP_NCELLS, cB % STEP_NCELLS);
                const bool chA = cA % STEP_NCELLS == S_CH_LO || cA % STEP_NCELLS == S_CH_HI, chB = cB % STEP_NCELLS == S_CH_LO || cB % STEP_NCELLS == S_CH_HI;
                const bool gasA = cA % STEP_NCELLS == S_GAS, gasB = cB % STEP_NCELLS == S_GAS;
                const u32 e = r2 ? (u32)eB : (u32)eA;
                const bool is_ch = r2 ? chB : chA, is_gas = r2 ? gasB : gasA;
                __attribute__((address_space(3))) u32* dst = wave_stage + e * EVM_STAGE_STRIDE + p;
                dst[0] = x.x;
                if (is_ch || is_gas) dst[EVM_STAGE_STRIDE] = x.y;
                if (is_ch) {
                    dst[2 * EVM_STAGE_STRIDE] = x.z;
                    dst[3 * EVM_STAGE_STRIDE] = x.w;
                }
                bad |= is_ch ? 0u : ((is_gas ? 0u : x.y) | x.z | x.w);
            }
        }
        if (evm_quad_or(bad) != 0u) wide_rounds |= 1u << it;
    }
    const u32 theirs = (u32)__shfl((int)wide_rounds, (int)(4u * (lane & 15u)));  // the quad that fetched this lane's pair
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ((theirs >> (lane >> 4)) & 1u) != 0u;
}
#endif

// Cold families (tuning builds: -DZK_COLD_MASK=<bits> keeps only some of them, to see which gadgets a build's registers / stack come
// from): 0 error states | 1 account / block / tx readers | 2 SDIV_SMOD | 3 precompiles | 4 CREATE | 5 CALL family | 6 tx / block framing
#ifndef ZK_COLD_MASK
#define ZK_COLD_MASK 0x7fu
#endif
#define EVM_COLD_FAM(f) (G == EVM_GROUP_COLD && ((ZK_COLD_MASK >> (f)) & 1u))
// (Measured with it, round 5: error states alone 592 B of stack per lane, readers 464, SDIV_SMOD 996, precompiles 624, CREATE 736,
// CALL 592, tx / block framing 816 — 2,492 together; 448 of every figure is table_lookup's private query copy.  Putting each gadget
// behind an out-of-line call made it 3,004: the callee-saved registers of a 256-VGPR gadget go to the stack too.)
template <int G>
ZK_HD u32 evm_check_step(const EvmArgs& a, u64 idx, EVM_LDS32_PTR stage = nullptr, EVM_LDS_PTR dir_lds = nullptr) {
    Ins I;
    I.a = &a;
    I.stage = stage;
    I.dir_lds = dir_lds;
    I.idx = idx;
    I.err = 0;
    I.seq = 0;
    I.rw_off = 0;
    I.pc_off = 0;
    I.sp_off = 0;
    I.code_state = 0;
    I.pre_op = 0;
    I.pre_op_ok = false;
    I.op_info = 0;
    I.op_info_byte = 0x100u;
    EV_PROF(I, 0);
    Fr statef, next_statef;
    I.defer = 0;
#if EVM_FAST
    {   // (unstaged pairs never get here: the kernel defers them)
#else
    if (I.stage) {  // one branch and one batch of LDS reads for the cells every step needs (not one of each per cell)
#endif
        I.rwc = ev_staged_cell(I, 0, S_RWC);
        I.call_id = ev_staged_cell(I, 0, S_CALL_ID);
        I.sp = ev_staged_cell(I, 0, S_SP);
        I.pc = ev_staged_cell(I, 0, S_PC);
        statef = ev_staged_cell(I, 0, S_STATE);
        next_statef = ev_staged_cell(I, 1, S_STATE);
    }
#if !EVM_FAST
    else {
        I.rwc = ev_step_cell(a, idx, S_RWC);
        I.call_id = ev_step_cell(a, idx, S_CALL_ID);
        I.sp = ev_step_cell(a, idx, S_SP);
        I.pc = ev_step_cell(a, idx, S_PC);
        statef = ev_step_cell(a, idx, S_STATE);
        next_statef = ev_step_cell(a, idx + 1, S_STATE);
    }
#endif
    const bool is_first = (a.opts & 1u) && idx == 0;
    const bool is_last = (a.opts & 2u) && idx == (u64)a.n_pairs - 1;
    const u32 state = statef.v[0];
    const u32 next_state = next_statef.v[0];  // used unless is_last
    {   // the hot instantiation (ALL) owns the three hot groups, the warm and the cold one their own
        const int grp = evm_state_group(state);
        if (G == EVM_GROUP_ALL ? grp >= EVM_GROUP_WARM : grp != G) return ZK_NOT_MINE;
    }
#if !defined(ZK_HOSTSIM)
    if (EV_PROF_ON(a)) a.prof[EV_PROF_WAVE * 8 + 4] = state;
#endif
    if (is_first) {
        ev_require(I, state == ES_BeginTx || state == ES_EndBlock);
        constrain_equal(I, ev_curr(I, S_RWC), fr_u(1));
    }
    if (is_last) ev_require(I, state == ES_EndBlock);
    else ev_require(I, state_transition_ok(state, next_state));
    if (I.err) return I.err;
    I.seq++;
    if (state >= ES_COUNT || !state_bit(ZK_STATE_REF_IMPL_MASK_LO, ZK_STATE_REF_IMPL_MASK_HI, state)) {
        ev_fail(I, ZK_NOT_IMPLEMENTED);
        return I.err;
    }
    if (G == EVM_GROUP_ALL) evm_prefetch(I);
    EV_PROF(I, 1);
    Tail T;
    T.enabled = false;
    T.err_tail = 0;
#ifdef ZK_ONLY_STATE
    if (state != ZK_ONLY_STATE) return 0;  // ISA-inspection builds: keep a single gadget
#endif
    switch (state) {
    case ES_ADD: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_ADD) { g_add_sub(I, T); } break;
    case ES_MUL: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_MUL) { g_mul_div_mod(I, T); } break;
    case ES_CMP: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_CMP) { g_cmp(I, T); } break;
    case ES_SCMP: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SCMP) { g_scmp(I, T); } break;
    case ES_ISZERO: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_ISZERO) { g_iszero(I, T); } break;
    case ES_NOT: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_NOT) { g_not(I, T); } break;
    case ES_BITWISE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_BITWISE) { g_bitwise(I, T); } break;
    case ES_BYTE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_BYTE) { g_byte(I, T); } break;
    case ES_SIGNEXTEND: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SIGNEXTEND) { g_signextend(I, T); } break;
    case ES_PUSH: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_PUSH) { g_push(I, T); } break;
    case ES_POP: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_POP) { g_pop(I, T); } break;
    case ES_SHL_SHR: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SHL_SHR) { g_shl_shr(I, T); } break;
    case ES_ADDMOD: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_ADDMOD) { g_addmod(I, T); } break;
    case ES_MULMOD: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_MULMOD) { g_mulmod(I, T); } break;
    case ES_MEMORY: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_MEMORY) { g_memory(I, T); } break;
    case ES_CALLER: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_CALLER) { g_ctx_word(I, T, OP_CALLER, CC_CallerAddress); } break;
    case ES_CALLVALUE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_CALLVALUE) { g_ctx_word(I, T, OP_CALLVALUE, CC_Value); } break;
    case ES_ADDRESS: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_ADDRESS) { g_ctx_word(I, T, OP_ADDRESS, CC_CalleeAddress); } break;
    case ES_CALLDATASIZE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_CALLDATASIZE) { g_ctx_value(I, T, OP_CALLDATASIZE, CC_CallDataLength); } break;
    case ES_RETURNDATASIZE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_RETURNDATASIZE) { g_ctx_value(I, T, OP_RETURNDATASIZE, CC_LastCalleeReturnDataLength); } break;
    case ES_ORIGIN: if (EVM_COLD_FAM(1)) { g_tx_word(I, T, OP_ORIGIN, TXC_CallerAddress); } break;
    case ES_GASPRICE: if (EVM_COLD_FAM(1)) { g_tx_word(I, T, OP_GASPRICE, TXC_GasPrice); } break;
    case ES_SELFBALANCE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SELFBALANCE) { g_selfbalance(I, T); } break;
    case ES_BlockCtx: if (EVM_COLD_FAM(1)) { g_blockctx(I, T); } break;
    case ES_GAS: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_GAS) { g_gas(I, T); } break;
    case ES_MSIZE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_MSIZE) { g_msize(I, T); } break;
    case ES_CODESIZE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_CODESIZE) { g_codesize(I, T); } break;
    case ES_STOP: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_STOP) { g_stop(I, T); } break;
    case ES_SAR: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SAR) { g_sar(I, T); } break;
    case ES_ErrorOutOfGasStaticMemoryExpansion: if (EVM_COLD_FAM(0)) { g_error_oog_static_memory(I, T); } break;
    case ES_ErrorOutOfGasDynamicMemoryExpansion: if (EVM_COLD_FAM(0)) { g_error_oog_dynamic_memory(I, T); } break;
    case ES_ErrorOutOfGasMemoryCopy: if (EVM_COLD_FAM(0)) { g_error_oog_memory_copy(I, T); } break;
    case ES_ErrorOutOfGasAccountAccess: if (EVM_COLD_FAM(0)) { g_error_oog_account_access(I, T); } break;
    case ES_ErrorOutOfGasLOG: if (EVM_COLD_FAM(0)) { g_error_oog_log(I, T); } break;
    case ES_ErrorOutOfGasEXP: if (EVM_COLD_FAM(0)) { g_error_oog_exp(I, T); } break;
    case ES_ErrorOutOfGasSHA3: if (EVM_COLD_FAM(0)) { g_error_oog_sha3(I, T); } break;
    case ES_ErrorReturnDataOutOfBound: if (EVM_COLD_FAM(0)) { g_error_return_data_oob(I, T); } break;
    case ES_ErrorWriteProtection: if (EVM_COLD_FAM(0)) { g_error_write_protection(I, T); } break;
    case ES_DATACOPY: if (EVM_COLD_FAM(3)) { g_datacopy(I, T); } break;
    case ES_ECRECOVER: if (EVM_COLD_FAM(3)) { g_ecrecover(I, T); } break;
    case ES_BN254_ADD: if (EVM_COLD_FAM(3)) { g_ecadd_ecmul(I, T, false); } break;
    case ES_BN254_SCALAR_MUL: if (EVM_COLD_FAM(3)) { g_ecadd_ecmul(I, T, true); } break;
    case ES_BN254_PAIRING: if (EVM_COLD_FAM(3)) { g_ecpairing(I, T); } break;
    case ES_ErrorOutOfGasPrecompile: if (EVM_COLD_FAM(0)) { g_error_oog_precompile(I, T); } break;
    case ES_ErrorOutOfGasCREATE: if (EVM_COLD_FAM(0)) { g_error_oog_create(I, T); } break;
    case ES_ErrorGasUintOverflow: if (EVM_COLD_FAM(0)) { g_error_gas_uint_overflow(I, T); } break;
    case ES_CREATE: case ES_CREATE2: if (EVM_COLD_FAM(4)) { g_create(I, T); } break;
    case ES_ErrorOutOfGasSloadSstore: if (EVM_COLD_FAM(0)) { g_error_oog_sload_sstore(I, T); } break;
    case ES_CALL_OP: if (EVM_COLD_FAM(5)) { g_callop(I, T); } break;
    case ES_ErrorOutOfGasCall: if (EVM_COLD_FAM(0)) { g_error_oog_call(I, T); } break;
    case ES_BeginTx: if (EVM_COLD_FAM(6)) { g_begin_tx(I, T, is_first); } break;
    case ES_EndTx: if (EVM_COLD_FAM(6)) { g_end_tx(I, T); } break;
    case ES_RETURN: if (EVM_COLD_FAM(6)) { g_return(I, T); } break;
    case ES_ErrorInvalidCreationCode: if (EVM_COLD_FAM(0)) { g_error_invalid_creation_code(I, T); } break;
    case ES_ErrorMaxCodeSizeExceeded: case ES_ErrorOutOfGasCodeStore: if (EVM_COLD_FAM(0)) { g_error_code_store(I, T); } break;
    case ES_EndBlock: if (EVM_COLD_FAM(6)) { g_end_block(I, T, is_last); } break;
    case ES_ErrorInvalidOpcode: if (EVM_COLD_FAM(0)) { g_error_invalid_opcode(I, T); } break;
    case ES_ErrorStack: if (EVM_COLD_FAM(0)) { g_error_stack(I, T); } break;
    case ES_ErrorOutOfGasConstant: if (EVM_COLD_FAM(0)) { g_error_oog_constant(I, T); } break;
    case ES_ErrorInvalidJump: if (EVM_COLD_FAM(0)) { g_error_invalid_jump(I, T); } break;
    case ES_LOG: if (G == EVM_GROUP_WARM) { g_log(I, T); } break;
    case ES_SHA3: if (G == EVM_GROUP_WARM) { g_sha3(I, T); } break;
    case ES_CODECOPY: if (G == EVM_GROUP_WARM) { g_codecopy(I, T); } break;
    case ES_CALLDATACOPY: if (G == EVM_GROUP_WARM) { g_calldatacopy(I, T); } break;
    case ES_RETURNDATACOPY: if (G == EVM_GROUP_WARM) { g_returndatacopy(I, T); } break;
    case ES_EXTCODECOPY: if (G == EVM_GROUP_WARM) { g_extcodecopy(I, T); } break;
    case ES_EXP: if (G == EVM_GROUP_WARM) { g_exp(I, T); } break;
    case ES_BALANCE: if (EVM_COLD_FAM(1)) { g_balance(I, T); } break;
    case ES_EXTCODESIZE: if (EVM_COLD_FAM(1)) { g_extcodesize(I, T); } break;
    case ES_EXTCODEHASH: if (EVM_COLD_FAM(1)) { g_extcodehash(I, T); } break;
    case ES_BLOCKHASH: if (EVM_COLD_FAM(1)) { g_blockhash(I, T); } break;
    case ES_CALLDATALOAD: if (EVM_COLD_FAM(1)) { g_calldataload(I, T); } break;
    case ES_SDIV_SMOD: if (EVM_COLD_FAM(2)) { g_sdiv_smod(I, T); } break;
    case ES_JUMP: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_JUMP) { g_jump(I, T); } break;
    case ES_JUMPI: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_JUMPI) { g_jumpi(I, T); } break;
    case ES_SLOAD: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SLOAD) { g_sload(I, T); } break;
    case ES_SSTORE: if (G == EVM_GROUP_ALL || G == GROUP_OF_ES_SSTORE) { g_sstore(I, T); } break;
    default: ev_fail(I, ZK_UNSUPPORTED); break;
    }
    EV_PROF(I, 2);
    if (I.err == 0u && T.enabled) same_context(I, T);
    if (I.err == 0u && T.err_tail) error_tail(I, T);
    EV_PROF(I, 3);
#if EVM_FAST
    if (I.defer) return ZK_DEFERRED_BASE + (I.defer & 7u);  // whatever the fast path concluded after a skipped fallback does not count
#endif
    return I.err;
}
