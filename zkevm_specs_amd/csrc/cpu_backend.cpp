// libzkevm_cpu.so — the C ABI of include/zkevm_hip.h evaluated on the host cores (SURVEY.md §2 native piece vi, §8b
// `backend` selector, §8d "C++ CPU backend on 1 and all cores").
//
// The per-row / per-step device functions of csrc/*.hpp are plain C++ when compiled without hipcc (the ZK_HOSTSIM switch of
// fr.hpp selects the host flavour of the ZK_HD / ZK_NOINLINE macros); this file puts the same session protocol around them
// that zkevm_hip.hip puts around the kernels: open = copy the caller's arrays and build the indices, launch = one pass over the
// rows with an OpenMP parallel-for, collect = the tally, read_status = the per-row codes.  What it is for:
//   * BASELINE configs[0] ("Bytecode circuit ... pure CPU path (plumbing, no GPU)") through the real boundary;
//   * the fair optimised-CPU line next to the GPU numbers (bench.py cpu_baseline legs `cpu_backend_1core` / `_allcores`).
// It is selected explicitly (ZK_BACKEND=cpu -> zkevm_specs_amd/_lib.py loads this library instead of libzkevm_hip.so); nothing
// falls back to it, and it shares no code with oracle/ (the Python restatement used as the test oracle).
// Not implemented here (they return an error that says so): the device-side witness assignments (zk_state_assign*,
// zk_bytecode_assign*, zk_copy_assign*), ZK_OPT_DEVICE_PTRS, zk_session_set_stream.
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <functional>
#include <string>
#include <vector>

#include "../../include/zkevm_hip.h"
#include "dist_tally.hpp"
#include "state_circuit.hpp"
#include "evm_circuit.hpp"
#include "host_index.hpp"
#include "row_circuits.hpp"
#include "copy_circuit.hpp"
#include "sign_circuit.hpp"
#include "keccak_table.hpp"
#include "secp256k1.hpp"
#include "pi_circuit.hpp"
#include "state_assign.hpp"
#include "bytecode_assign.hpp"
#include "state_rekey.hpp"

static thread_local std::string g_err;
#define ARG_TRY(cond, msg) do { if (!(cond)) { g_err = msg; return -1; } } while (0)
#include "copy_assign_plan.hpp"  // (uses ARG_TRY)

extern "C" const char* zk_last_error(void) { return g_err.c_str(); }
extern "C" int zk_init(int) { return 0; }
extern "C" void zk_shutdown(void) {}
extern "C" int zk_set_stream(void*) { return 0; }

// ---------------------------------------------------------------------------------------
// tables: a private copy of the caller's rows + the open-addressing index
// ---------------------------------------------------------------------------------------
struct CpuTable {
    ZkTable t;
    std::vector<u64> cells;
    std::vector<u32> flags, slots;
};
static void cpu_table(CpuTable& h, const u64* cells, const u32* flags, u64 n, u32 ncells, u64 (*hash_of)(const ZkTable&, u32)) {
    static const u64 zero_row[64] = {0};
    static const u32 zero_flag[1] = {0};
    if (n) h.cells.assign(cells, cells + n * ncells * 4);
    if (n && flags) h.flags.assign(flags, flags + n);
    h.t.cells = n ? h.cells.data() : zero_row;  // empty table: one readable zero row (as in the HIP library)
    h.t.flags = n ? (flags ? h.flags.data() : nullptr) : zero_flag;
    h.t.n = (u32)n;
    h.t.ncells = ncells;
    u32 cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    h.slots.assign(cap, ZK_EMPTY_SLOT);
    h.t.slots = h.slots.data();
    h.t.mask = cap - 1;
    if (hash_of)
        for (u32 r = 0; r < (u32)n; r++) {
            u32 s = (u32)hash_of(h.t, r) & h.t.mask;
            while (h.slots[s] != ZK_EMPTY_SLOT) s = (s + 1) & h.t.mask;
            h.slots[s] = r;
        }
}
static Fr cell_of(const u64* p) { return fr_load(p); }

// ---------------------------------------------------------------------------------------
// sessions
// ---------------------------------------------------------------------------------------
struct zk_session {
    u64 n = 0, lo = 0, hi = 0;            // rows; evaluated range
    std::function<u32(u64)> row;          // status code of row i
    std::vector<u32> status;
    u32 launches = 0;
    double ms = 0;
    u64 fail_count = 0, first_row = ~0ull;
    u32 first_code = 0;
    bool range_ok = false;                // zk_set_range applies (row circuits)
    // owned inputs (kept alive for `row`)
    std::vector<u64> a64[4];
    std::vector<u32> a32[4];
    std::vector<uint8_t> a8, a8b;
    std::vector<uint16_t> a16;
    std::vector<u64> w64[4];              // assignment sessions: work / output buffers
    std::vector<u32> out32;
    void (*pass)(zk_session*) = nullptr;  // assignment sessions: one pass computes the outputs and fills `status`
    int assign_kind = 0;                  // 1 state, 2 bytecode, 3 copy, 4 RW -> State ops
    u64 n_ops = 0;                        // RW -> State ops: 1 + kept rows
    std::vector<u32> rekey_plan;          // RW -> State ops: the RwkHostPlan's plan (as words) ...
    std::vector<u32> rekey_jobs;          // ... and its rank jobs (cls, field, base, count)
    u64 n_mpt = 0;
    CpuTable tab[12];
    std::vector<u64> keccak_rows;         // keccak sessions: the table
    // per-circuit argument blocks (one is used)
    StateArgs state;
    EvmArgs evm;
    BytecodeArgs bytecode;
    ExpArgs exp;
    CopyArgs copy;
    SignArgs sign;
    PiArgs pi;
    PiCopyArgs picopy;
    AssignArgs assign;
    BcaArgs bca;
    std::vector<BcaChunk> bca_chunks;
    std::vector<uint8_t> bca_map;
    CpaArgs cpa;
    CpaPlan cpa_plan;
    EcdsaArgs ecdsa;
    KeccakGenArgs kgen;
    ZkRwMeta rw_meta;
    HostCodeDir dir;
    std::vector<u64> aux64;
    bool status_external = false;  // the last pass wrote to the caller's buffer (zk_read_status refuses then, like the HIP library)
};

static void run_pass(zk_session* s, u32* status_out) {
    const auto t0 = std::chrono::steady_clock::now();
    u32* st = status_out ? status_out : s->status.data();
    const long long lo = (long long)s->lo, hi = (long long)s->hi;
    if (s->pass) {  // witness assignment: sequential host loops over the device functions
        s->pass(s);
        if (status_out) memcpy(status_out, s->status.data(), s->n * sizeof(u32));
    } else {
#pragma omp parallel for schedule(dynamic, 256)
        for (long long i = lo; i < hi; i++) st[i] = s->row((u64)i);
    }
    u64 fails = 0, first = ~0ull;
    for (long long i = lo; i < hi; i++)
        if (st[i]) {
            if (!fails) first = (u64)i;
            fails++;
        }
    s->fail_count = fails;
    s->first_row = first;
    s->first_code = fails ? st[first] : 0u;
    s->ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s->launches++;
}
extern "C" int zk_launch(zk_session* s, uint32_t* status_dev) {
    ARG_TRY(s, "zk_launch: null session");
    s->status_external = status_dev != nullptr;
    run_pass(s, status_dev);
    return 0;
}
extern "C" int zk_collect(zk_session* s, zk_result* r) {
    ARG_TRY(s && r, "zk_collect: bad arguments");
    r->fail_count = s->fail_count;
    r->first_fail_row = s->first_row == ~0ull ? UINT64_MAX : s->first_row;
    r->first_fail_code = s->first_code;
    r->launches = s->launches;
    r->rows_evaluated = s->hi - s->lo;
    r->kernel_ms = s->launches ? s->ms / s->launches : 0.0;
    s->launches = 0;
    s->ms = 0;
    return 0;
}
extern "C" int zk_read_status(zk_session* s, uint32_t* status_host) {
    ARG_TRY(s && status_host, "zk_read_status: bad arguments");
    // same contract as the HIP library: the last pass's codes are the caller's when it handed over its own buffer
    ARG_TRY(!s->status_external, "zk_read_status: the last pass wrote its statuses to the caller's status_dev buffer, not the session's");
    memcpy(status_host, s->status.data(), s->n * sizeof(u32));
    return 0;
}
extern "C" int zk_close(zk_session* s) {
    delete s;
    return 0;
}
extern "C" int zk_session_set_stream(zk_session*, void*) { return 0; }
// device-side event spans: nothing to measure on the host (callers read -1 = not measured)
extern "C" int zk_session_timing(zk_session*, double* open_ms, double* span_ms) {
    if (open_ms) *open_ms = -1.0;
    if (span_ms) *span_ms = -1.0;
    return 0;
}
// Multi-GPU tally.  World 1 is the identity with the row offset applied.  World > 1 goes through the same collective binding as
// the HIP library (dist_tally.hpp) with HOST buffers — which RCCL itself does not accept, so it needs ZK_RCCL_LIB to name a
// collective library that does (tests/fakerccl: an all-gather over a shared-memory file); without one, ranks of a CPU job
// exchange their tallies on the host (zkevm_specs_amd.distributed.reduce_tally over gloo).
struct zk_comm {
    int rank = 0, world = 1;
    void* nccl = nullptr;
    std::vector<uint64_t> buf;  // TALLY_WORDS of this rank | TALLY_WORDS * world gathered
};
#define RCCL_TRY(expr, what)                                                                                                                   \
    do {                                                                                                                                       \
        const int r_ = (expr);                                                                                                                 \
        if (r_ != 0) {                                                                                                                         \
            char buf_[256];                                                                                                                    \
            snprintf(buf_, sizeof buf_, "%s: RCCL error %d (%s)", what, r_, zkdist::rccl().GetErrorString ? zkdist::rccl().GetErrorString(r_) : "?"); \
            g_err = buf_;                                                                                                                      \
            return -3;                                                                                                                         \
        }                                                                                                                                      \
    } while (0)
// a collective library that takes HOST buffers: it has to say so itself (`zk_collective_buffers() == 0`, dist_tally.hpp) — ZK_RCCL_LIB
// may just as well name the integrator's own build of RCCL, whose ncclAllGather must never see host pointers
static bool host_collective() { return getenv("ZK_RCCL_LIB") && zkdist::rccl().ok && zkdist::rccl().named_by_env && zkdist::rccl().buffers == 0; }
extern "C" int zk_dist_unique_id(uint8_t* id) {
    ARG_TRY(id, "zk_dist_unique_id: id is null");
    memset(id, 0, ZK_DIST_ID_BYTES);
    if (host_collective()) {
        zkdist::RcclId u;
        RCCL_TRY(zkdist::rccl().GetUniqueId(&u), "ncclGetUniqueId");
        memcpy(id, u.internal, ZK_DIST_ID_BYTES);
    }
    return 0;
}
extern "C" int zk_dist_close(zk_comm* c) {
    if (c && c->nccl) (void)zkdist::rccl().CommDestroy(c->nccl);
    delete c;
    return 0;
}
extern "C" int zk_dist_init(const uint8_t* id, int rank, int world, zk_comm** out) {
    ARG_TRY(id && out && world >= 1 && rank >= 0 && rank < world, "zk_dist_init: bad arguments");
    ARG_TRY(world == 1 || host_collective(),
            "zk_dist_init: the CPU backend has no collective of its own (world must be 1, or ZK_RCCL_LIB must name a library that gathers host buffers); reduce on the host");
    zk_comm* c = new zk_comm();
    c->rank = rank;
    c->world = world;
    c->buf.assign((size_t)(world + 1) * zkdist::TALLY_WORDS, 0);
    if (world > 1) {
        zkdist::RcclId u;
        memcpy(u.internal, id, ZK_DIST_ID_BYTES);
        if (int r = zkdist::rccl().CommInitRank(&c->nccl, world, u, rank)) {
            char buf[256];
            snprintf(buf, sizeof buf, "ncclCommInitRank: RCCL error %d", r);
            g_err = buf;
            c->nccl = nullptr;
            zk_dist_close(c);
            return -3;
        }
    }
    *out = c;
    return 0;
}
extern "C" int zk_dist_tally(zk_comm* c, const zk_result* local, uint64_t row_offset, zk_result* global) {
    ARG_TRY(c && local && global, "zk_dist_tally: bad arguments");
    uint64_t* mine = c->buf.data();
    uint64_t* all = mine + zkdist::TALLY_WORDS;
    zkdist::tally_pack(mine, local, row_offset);
    if (c->world == 1) memcpy(all, mine, zkdist::TALLY_WORDS * sizeof(uint64_t));
    else RCCL_TRY(zkdist::rccl().AllGather(mine, all, zkdist::TALLY_WORDS, zkdist::RCCL_UINT64, c->nccl, nullptr), "ncclAllGather");
    zkdist::tally_reduce(all, c->world, local, global);
    return 0;
}
extern "C" int zk_last_host_phases(double* us4) {
    if (us4) for (int k = 0; k < 4; k++) us4[k] = -1.0;
    return 0;
}
extern "C" int zk_timing_sums(double* sums_ms, uint64_t* count, int) {  // (no device spans on this backend)
    if (sums_ms) for (int k = 0; k < 3; k++) sums_ms[k] = -1.0;
    if (count) *count = 0;
    return 0;
}
extern "C" int zk_last_timing(double* open_ms, double* pass_ms, double* span_ms) {
    if (open_ms) *open_ms = -1.0;
    if (pass_ms) *pass_ms = -1.0;
    if (span_ms) *span_ms = -1.0;
    return 0;
}
extern "C" int zk_set_range(zk_session* s, uint64_t row_lo, uint64_t row_hi) {
    ARG_TRY(s && s->range_ok, "zk_set_range: not a row-circuit session");
    ARG_TRY(row_lo < row_hi && row_hi <= s->n, "zk_set_range: bad range");
    s->lo = row_lo;
    s->hi = row_hi;
    std::fill(s->status.begin(), s->status.end(), 0u);
    return 0;
}
extern "C" int zk_state_set_range(zk_session* s, uint64_t row_lo, uint64_t row_hi) { return zk_set_range(s, row_lo, row_hi); }

static zk_session* new_session(u64 n, bool range_ok) {
    zk_session* s = new zk_session();
    s->n = n;
    s->lo = 0;
    s->hi = n;
    s->status.assign(n ? n : 1, 0u);
    s->range_ok = range_ok;
    return s;
}
static int one_shot(zk_session* s, uint32_t* status_out, zk_result* result) {
    int rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && status_out) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}
#define NO_DEVICE_PTRS(opts, name) ARG_TRY(!((opts) & ZK_OPT_DEVICE_PTRS), name ": ZK_OPT_DEVICE_PTRS has no meaning on the CPU backend")
#ifndef ZK_OPT_STATE_COMPACT
#define ZK_OPT_STATE_COMPACT 32u
#endif

// ---- Fr vector ops ------------------------------------------------------------------------------------------------------
extern "C" int zk_fr_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n, uint32_t opts) {
    NO_DEVICE_PTRS(opts, "zk_fr_op");
    ARG_TRY(a && b && out, "zk_fr_op: null pointer");
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) {
        const Fr x = fr_load(a + 4 * i), y = fr_load(b + 4 * i);
        Fr r;
        switch (op) {
        case 0: r = fr_add(x, y); break;
        case 1: r = fr_sub(x, y); break;
        case 2: r = fr_mul(x, y); break;
        case 3: r = fr_mont(x, y); break;
        case 4: r = fr_neg(x); break;
        case 5: r = fr_inv(x); break;
        case 6: r = fr_div(x, y); break;
        default: r = fr_zero();
        }
        for (int k = 0; k < 4; k++) out[4 * i + k] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
    }
    return 0;
}

// ---- State circuit ------------------------------------------------------------------------------------------------------
extern "C" int zk_state_open(const uint64_t* rows, const uint32_t* flags, uint64_t n, const uint64_t* mpt, uint64_t n_mpt,
                             uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_state_open");
    ARG_TRY(out && rows && n > 0 && n < (1ull << 32) && n_mpt < (1ull << 31), "zk_state_open: bad arguments");
    zk_session* s = new_session(n, true);
    const bool compact = opts & ZK_OPT_STATE_COMPACT;  // 15-cell rows, limb / byte decompositions derived (include/zkevm_hip.h)
    s->a64[0].assign(rows, rows + n * (compact ? 15 : ST_NCELLS) * 4);
    s->state.rows.skip = compact ? 42u : 0u;
    if (flags) s->a32[0].assign(flags, flags + n);
    else s->a32[0].assign(n, 0u);
    cpu_table(s->tab[0], mpt, nullptr, n_mpt, MPT_NCELLS, nullptr);
    {  // the MPT index carries a hash fingerprint in each slot (state_mpt_slot_value)
        CpuTable& h = s->tab[0];
        for (u32 r = 0; r < (u32)n_mpt; r++) {
            const u64 hv = state_mpt_key_hash(h.t, r);
            u32 k = (u32)hv & h.t.mask;
            while (h.slots[k] != ZK_EMPTY_SLOT) k = (k + 1) & h.t.mask;
            h.slots[k] = state_mpt_slot_value(h.t, r, hv);
        }
    }
    s->state.rows.cells = s->a64[0].data();
    s->state.rows.flags = s->a32[0].data();
    s->state.rows.n = n;
    s->state.mpt = s->tab[0].t;
    s->state.eval_lo = 0;
    s->state.eval_hi = n;
    s->row = [s](u64 i) { return state_check_row(s->state, i); };
    *out = s;
    return 0;
}
extern "C" int zk_state_verify(const uint64_t* rows, const uint32_t* flags, uint64_t n, const uint64_t* mpt, uint64_t n_mpt,
                               uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_state_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_state_open(rows, flags, n, mpt, n_mpt, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

// ---- EVM circuit --------------------------------------------------------------------------------------------------------
extern "C" int zk_evm_open(const zk_evm_tables* t, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_evm_open");
    ARG_TRY(t && out && t->steps && t->n_steps >= 2 && t->n_steps < (1ull << 32), "zk_evm_open: bad arguments");
    ARG_TRY(t->aux_cells == 0 || (t->aux_cells >= 2 && t->aux_cells <= 64), "zk_evm_open: aux_cells must be 0 (= 2) or 2..64");
    zk_session* s = new_session(t->n_steps - 1, false);
    EvmArgs& a = s->evm;
    s->a64[0].assign(t->steps, t->steps + t->n_steps * STEP_NCELLS * 4);
    a.dyn = nullptr;
    a.step_recs = nullptr;
    a.defer_list = nullptr;
    a.defer_count = nullptr;
    a.steps = s->a64[0].data();
    a.n_steps = t->n_steps;
    cpu_table(s->tab[0], t->rw, t->rw_flags, t->n_rw, RW_NCELLS, rw_key_hash);
    cpu_table(s->tab[1], t->bytecode, nullptr, t->n_bytecode, BYTECODE_NCELLS, bc_key_hash);
    cpu_table(s->tab[2], t->tx, t->tx_flags, t->n_tx, TX_NCELLS, tx_key_hash);
    cpu_table(s->tab[3], t->block, t->block_flags, t->n_block, BLOCK_NCELLS, blk_key_hash);
    cpu_table(s->tab[4], t->copy, nullptr, t->n_copy, COPY_T_NCELLS, copy_key_hash);
    cpu_table(s->tab[5], t->keccak, nullptr, t->n_keccak, KECCAK_NCELLS, keccak_key_hash);
    cpu_table(s->tab[6], t->exp, nullptr, t->n_exp, EXP_T_NCELLS, expt_key_hash);
    cpu_table(s->tab[7], t->sig, nullptr, t->n_sig, SIG_T_NCELLS, sig_key_hash);
    cpu_table(s->tab[8], t->ecc, nullptr, t->n_ecc, ECC_T_NCELLS, ecc_key_hash);
    cpu_table(s->tab[9], t->withdrawals, nullptr, t->n_withdrawals, 4, nullptr);  // walked in order: no index
    a.rw = s->tab[0].t; a.bytecode = s->tab[1].t; a.tx = s->tab[2].t; a.block = s->tab[3].t; a.copy = s->tab[4].t;
    a.keccak = s->tab[5].t; a.exp = s->tab[6].t; a.sig = s->tab[7].t; a.ecc = s->tab[8].t; a.withdrawals = s->tab[9].t;
    {
        const HostEvmAgg g = evm_aggregates_host(t->tx, t->tx_flags, t->n_tx, t->withdrawals, t->n_withdrawals);
        a.agg_max_txs = g.max_txs; a.agg_total_txs = g.total_txs; a.agg_invalid_txs = g.invalid_txs;
        a.agg_bad_invalid_rows = g.bad_invalid_rows; a.agg_total_wds = g.total_wds;
    }
    a.aux = nullptr;
    a.aux_kind = nullptr;
    a.aux_cells = t->aux_cells ? t->aux_cells : 2u;
    if (t->aux && t->aux_kind) {
        s->aux64.assign(t->aux, t->aux + t->n_steps * a.aux_cells * 4);
        s->a32[1].assign(t->aux_kind, t->aux_kind + t->n_steps);
        a.aux = s->aux64.data();
        a.aux_kind = s->a32[1].data();
    }
    a.perm = nullptr;
    a.prof = nullptr;
    a.n_pairs = (u32)(t->n_steps - 1);
    a.opts = (t->begin_with_first_step ? 1u : 0u) | (t->end_with_last_step ? 2u : 0u);
    a.rw_dense = 0;
    a.rw_base = 0;
    a.rw_keys = nullptr;
    a.codes.n = 0;
    a.codes.mask = 0;
    a.codes.packed = nullptr;
    a.codes.entries = nullptr;
    a.codes.slots = nullptr;
    if (!(opts & ZK_OPT_GENERIC_INDEX)) {
        s->rw_meta = rw_dense_meta_host(s->tab[0].t.n ? s->tab[0].cells.data() : nullptr, t->n_rw);
        a.rw_dense = s->rw_meta.dense;
        a.rw_base = s->rw_meta.base;
        if (s->rw_meta.dense) {  // packed key records, as the device open builds them
            s->a64[1].resize((size_t)t->n_rw * 4);
            for (u64 r = 0; r < t->n_rw; r++) {
                const RwKey k = rw_pack_row(a.rw, (u32)r);
                for (int j = 0; j < 4; j++) s->a64[1][4 * r + j] = k.w[j];
            }
            a.rw_keys = s->a64[1].data();
        }
        if (t->n_bytecode) {
            build_code_dir(s->tab[1].cells.data(), t->n_bytecode, s->dir);
            a.codes.entries = s->dir.entries.data();
            a.codes.slots = s->dir.slots.data();
            a.codes.mask = s->dir.mask;
            a.codes.n = (u32)s->dir.entries.size();
            a.codes.packed = s->dir.packed.data();
        }
    }
    s->row = [s](u64 i) {  // as on the device: the hot and the cold instantiation split the states
        u32 c = evm_check_step<EVM_GROUP_ALL>(s->evm, i);
        if (c == ZK_NOT_MINE) c = evm_check_step<EVM_GROUP_WARM>(s->evm, i);
        if (c == ZK_NOT_MINE) c = evm_check_step<EVM_GROUP_COLD>(s->evm, i);
        return c;
    };
    *out = s;
    return 0;
}
extern "C" int zk_evm_verify(const zk_evm_tables* t, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_evm_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_evm_open(t, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

extern "C" int zk_evm_verify_batch(const zk_evm_tables* const* t, uint64_t n, uint32_t opts, zk_result* results) {
    ARG_TRY((t && results) || n == 0, "zk_evm_verify_batch: bad arguments");
    for (uint64_t i = 0; i < n; i++) ARG_TRY(t[i], "zk_evm_verify_batch: null witness");
    for (uint64_t i = 0; i < n; i++) {  // nothing to pipeline on the host: one witness after the other
        const int rc = zk_evm_verify(t[i], opts, nullptr, &results[i]);
        if (rc) return rc;
    }
    return 0;
}

// ---- Bytecode / Exp circuits --------------------------------------------------------------------------------------------
extern "C" int zk_bytecode_open(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* randomness,
                                uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_bytecode_open");
    ARG_TRY(out && rows && randomness && n > 0 && n < (1ull << 32) && n_keccak < (1ull << 31), "zk_bytecode_open: bad arguments");
    zk_session* s = new_session(n, true);
    s->a64[0].assign(rows, rows + n * BC_NCELLS * 4);
    cpu_table(s->tab[0], keccak, nullptr, n_keccak, KECCAK_NCELLS, keccak_key_hash);
    s->bytecode.rows.cells = s->a64[0].data();
    s->bytecode.rows.flags = nullptr;
    s->bytecode.rows.n = n;
    s->bytecode.keccak = s->tab[0].t;
    s->bytecode.r = cell_of(randomness);
    s->bytecode.r_mont = nullptr;
    s->row = [s](u64 i) { return bytecode_check_row(s->bytecode, i); };
    *out = s;
    return 0;
}
extern "C" int zk_bytecode_verify(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* randomness,
                                  uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_bytecode_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_bytecode_open(rows, n, keccak, n_keccak, randomness, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}
extern "C" int zk_exp_open(const uint64_t* rows, uint64_t n, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_exp_open");
    ARG_TRY(out && rows && n > 0 && n < (1ull << 32), "zk_exp_open: bad arguments");
    zk_session* s = new_session(n, true);
    s->a64[0].assign(rows, rows + n * EX_NCELLS * 4);
    s->exp.rows.cells = s->a64[0].data();
    s->exp.rows.flags = nullptr;
    s->exp.rows.n = n;
    s->row = [s](u64 i) { return exp_check_row(s->exp, i); };
    *out = s;
    return 0;
}
extern "C" int zk_exp_verify(const uint64_t* rows, uint64_t n, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_exp_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_exp_open(rows, n, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

// ---- Copy circuit -------------------------------------------------------------------------------------------------------
extern "C" int zk_copy_open(const zk_copy_tables* t, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_copy_open");
    ARG_TRY(t && out && t->rows && t->randomness && t->n_rows > 0 && t->n_rows < (1ull << 32), "zk_copy_open: bad arguments");
    zk_session* s = new_session(t->n_rows, true);
    s->a64[0].assign(t->rows, t->rows + t->n_rows * CP_NCELLS * 4);
    if (t->row_flags) s->a32[0].assign(t->row_flags, t->row_flags + t->n_rows);
    cpu_table(s->tab[0], t->rw, t->rw_flags, t->n_rw, RW_NCELLS, rw_key_hash);
    cpu_table(s->tab[1], t->bytecode, nullptr, t->n_bytecode, BYTECODE_NCELLS, bc_key_hash);
    cpu_table(s->tab[2], t->tx, t->tx_flags, t->n_tx, TX_NCELLS, tx_key_hash);
    s->copy.rows.cells = s->a64[0].data();
    s->copy.rows.flags = t->row_flags ? s->a32[0].data() : nullptr;
    s->copy.rows.n = t->n_rows;
    s->copy.rw = s->tab[0].t;
    s->copy.bytecode = s->tab[1].t;
    s->copy.tx = s->tab[2].t;
    s->rw_meta = rw_dense_meta_host(t->n_rw ? s->tab[0].cells.data() : nullptr, t->n_rw);
    s->copy.rw_meta = (opts & ZK_OPT_GENERIC_INDEX) ? nullptr : &s->rw_meta;
    s->copy.r = cell_of(t->randomness);
    s->row = [s](u64 i) { return copy_check_row(s->copy, i); };
    *out = s;
    return 0;
}
extern "C" int zk_copy_verify(const zk_copy_tables* t, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_copy_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_copy_open(t, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

// ---- Tx / Sig circuits --------------------------------------------------------------------------------------------------
extern "C" int zk_sign_open(const zk_sign_units* t, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_sign_open");
    ARG_TRY(t && out && t->bytes && t->cells && t->meta && t->randomness && t->n_units > 0 && t->n_units < (1ull << 32), "zk_sign_open: bad arguments");
    const u64 n = t->n_units;
    zk_session* s = new_session(n, true);
    s->a8.assign(t->bytes, t->bytes + n * SG_NBYTES_ROWS * 32);
    s->a64[0].assign(t->cells, t->cells + n * SG_NCELLS * 4);
    s->a32[0].assign(t->meta, t->meta + n * 4);
    cpu_table(s->tab[0], t->keccak, nullptr, t->n_keccak, KECCAK_NCELLS, keccak_key_hash);
    cpu_table(s->tab[1], t->tx_rows, t->tx_flags, t->n_tx_rows, TX_NCELLS, nullptr);
    SignArgs& a = s->sign;
    a.bytes = s->a8.data();
    a.cells.cells = s->a64[0].data();
    a.cells.flags = nullptr;
    a.cells.n = n;
    a.meta = s->a32[0].data();
    a.keccak = s->tab[0].t;
    a.tx_rows = s->tab[1].t;
    a.tx_rows.n = (u32)t->n_tx_rows;
    a.r = cell_of(t->randomness);
    s->a64[1].resize(64 * 4);
    sign_fill_rpow(a.r, s->a64[1].data());
    a.rpow = s->a64[1].data();
    a.is_sig = t->is_sig ? 1u : 0u;
    s->row = [s](u64 i) { return sign_check_unit(s->sign, i); };
    *out = s;
    return 0;
}
extern "C" int zk_sign_verify(const zk_sign_units* t, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_sign_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_sign_open(t, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

// ---- PI circuit ---------------------------------------------------------------------------------------------------------
extern "C" int zk_pi_open(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* gas, uint64_t n_gas,
                          uint64_t circuit_len, const uint64_t* keccak_rand, const uint64_t* byte_pow_base, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_pi_open");
    ARG_TRY(out && rows && keccak_rand && byte_pow_base && n > 0 && n < (1ull << 32), "zk_pi_open: bad arguments");
    zk_session* s = new_session(n, true);
    s->a64[0].assign(rows, rows + n * 24 * 4);
    cpu_table(s->tab[0], keccak, nullptr, n_keccak, KECCAK_NCELLS, keccak_key_hash);
    cpu_table(s->tab[1], gas, nullptr, n_gas, PI_GAS_NCELLS, pi_gas_key_hash);
    PiArgs& a = s->pi;
    a.rows.cells = s->a64[0].data();
    a.rows.flags = nullptr;
    a.rows.n = n;
    a.keccak = s->tab[0].t;
    a.gas = s->tab[1].t;
    a.keccak_rand_m = fr_to_mont(cell_of(keccak_rand));
    a.byte_pow_base_m = fr_to_mont(cell_of(byte_pow_base));
    a.circuit_len = fr_from_u64(circuit_len);
    s->row = [s](u64 i) { return pi_check_row(s->pi, i); };
    *out = s;
    return 0;
}
extern "C" int zk_pi_verify(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* gas, uint64_t n_gas,
                            uint64_t circuit_len, const uint64_t* keccak_rand, const uint64_t* byte_pow_base, uint32_t opts,
                            uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_pi_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_pi_open(rows, n, keccak, n_keccak, gas, n_gas, circuit_len, keccak_rand, byte_pow_base, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

extern "C" int zk_pi_copy_open(const uint64_t* cells, const uint8_t* bytes, const uint32_t* lens, uint64_t n, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_pi_copy_open");
    ARG_TRY(out && cells && bytes && lens && n > 0 && n < (1ull << 32), "zk_pi_copy_open: bad arguments");
    zk_session* s = new_session(n, false);
    s->a64[0].assign(cells, cells + n * 4);
    s->a8.assign(bytes, bytes + n * 32);
    s->a32[0].assign(lens, lens + n);
    s->picopy.cells = s->a64[0].data();
    s->picopy.bytes = s->a8.data();
    s->picopy.lens = s->a32[0].data();
    s->picopy.n = n;
    s->row = [s](u64 i) { return pi_copy_check(s->picopy, i); };
    *out = s;
    return 0;
}
extern "C" int zk_pi_copy_verify(const uint64_t* cells, const uint8_t* bytes, const uint32_t* lens, uint64_t n, uint32_t opts, uint32_t* status_out,
                                 zk_result* result) {
    ARG_TRY(result, "zk_pi_copy_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_pi_copy_open(cells, bytes, lens, n, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

// ---- Keccak table generation --------------------------------------------------------------------------------------------
extern "C" int zk_keccak_open(const uint8_t* data, uint64_t n_bytes, const uint64_t* offsets, uint64_t n_msgs, const uint64_t* randomness,
                              uint32_t mode, uint64_t* rows_dev, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_keccak_open");
    ARG_TRY(out && offsets && randomness && n_msgs > 0 && n_msgs < (1ull << 32) && (data || n_bytes == 0) && mode <= 1u && !rows_dev,
            "zk_keccak_open: bad arguments");
    for (u64 k = 0; k < n_msgs; k++) ARG_TRY(offsets[k] <= offsets[k + 1], "zk_keccak_open: offsets must be non-decreasing");
    ARG_TRY(offsets[n_msgs] <= n_bytes, "zk_keccak_open: offsets exceed the data buffer");
    zk_session* s = new_session(n_msgs, false);
    s->a8.assign(data, data + n_bytes);
    s->a64[0].assign(offsets, offsets + n_msgs + 1);
    s->a64[1].resize(KT_RPOW_ROWS * 4);
    kt_fill_rpow(cell_of(randomness), s->a64[1].data());
    s->keccak_rows.assign(n_msgs * KT_NCELLS * 4, 0);
    KeccakGenArgs& g = s->kgen;
    g.data = s->a8.data();
    g.offsets = s->a64[0].data();
    g.n = n_msgs;
    g.rpow = s->a64[1].data();
    g.rows = s->keccak_rows.data();
    g.mode = mode;
    g.long_list = nullptr;
    g.long_count = nullptr;
    s->row = [s](u64 i) { return keccak_table_row(s->kgen, i); };
    *out = s;
    return 0;
}
extern "C" int zk_keccak_read_rows(zk_session* s, uint64_t* rows_host) {
    ARG_TRY(s && rows_host && !s->keccak_rows.empty(), "zk_keccak_read_rows: bad arguments");
    memcpy(rows_host, s->keccak_rows.data(), s->keccak_rows.size() * 8);
    return 0;
}
extern "C" int zk_keccak_table(const uint8_t* data, uint64_t n_bytes, const uint64_t* offsets, uint64_t n_msgs, const uint64_t* randomness,
                               uint32_t mode, uint64_t* rows_out, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result && rows_out, "zk_keccak_table: null output");
    zk_session* s = nullptr;
    int rc = zk_keccak_open(data, n_bytes, offsets, n_msgs, randomness, mode, nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc) rc = zk_keccak_read_rows(s, rows_out);
    if (!rc && status_out) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// ---- secp256k1 ECDSA verification ---------------------------------------------------------------------------------------
extern "C" int zk_ecdsa_open_batches(const zk_ecdsa_batch* bt, uint32_t n_batches, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_ecdsa_open_batches");
    ARG_TRY(out && bt && (n_batches == 1 || n_batches == 2), "zk_ecdsa_open_batches: one or two batches");
    u64 n = 0;
    for (u32 k = 0; k < n_batches; k++) {
        ARG_TRY(bt[k].bytes && bt[k].n > 0 && bt[k].n < (1ull << 31) && bt[k].layout <= 2u && (!bt[k].v || bt[k].v_stride >= 1) && !bt[k].out_dev,
                "zk_ecdsa_open: bad arguments");
        n += bt[k].n;
    }
    static const u32 OFF[2][5] = {{0, 32, 64, 96, 128}, {64, 96, 160, 224, 256}};
    zk_session* s = new_session(n, false);
    EcdsaArgs& a = s->ecdsa;
    a.n = n;
    ecdsa_single_batch(a);
    a.n0 = bt[0].n;
    for (u32 k = 0; k < n_batches; k++) {
        const u64 stride = bt[k].layout ? 288 : 160;
        std::vector<uint8_t>& hb = k == 0 ? s->a8 : s->a8b;
        std::vector<u32>& hv = s->a32[k];
        hb.assign(bt[k].bytes, bt[k].bytes + bt[k].n * stride);
        if (bt[k].v) hv.assign(bt[k].v, bt[k].v + bt[k].n * bt[k].v_stride);
        if (k == 0) {
            a.stride = stride; a.bytes = hb.data(); a.v = bt[k].v ? hv.data() : nullptr; a.v_stride = bt[k].v_stride; a.msg_be = bt[k].layout != 1u;
            for (int c = 0; c < 5; c++) a.off[c] = OFF[bt[k].layout ? 1 : 0][c];
        } else {
            a.stride1 = stride; a.bytes1 = hb.data(); a.v1 = bt[k].v ? hv.data() : nullptr; a.v_stride1 = bt[k].v_stride; a.msg_be1 = bt[k].layout != 1u;
            for (int c = 0; c < 5; c++) a.off1[c] = OFF[bt[k].layout ? 1 : 0][c];
        }
    }
    a.first = 0;
    a.out = nullptr;
    a.out_stride = 0;
    a.qtab = nullptr;
    a.qtab_lanes = 1;
    a.lanes_per_sig = 1;
    s->row = [s](u64 i) {
        u32 tab[15 * 24];  // the key's multiples: per call on the CPU
        return ecdsa_verify_one(s->ecdsa, i, tab, 1);
    };
    *out = s;
    return 0;
}
extern "C" int zk_ecdsa_open(const uint8_t* bytes, uint32_t layout, const uint32_t* v, uint32_t v_stride, uint64_t n, uint32_t* out_dev,
                             uint32_t out_stride, uint32_t opts, zk_session** out) {
    zk_ecdsa_batch b;
    b.bytes = bytes; b.layout = layout; b.v = v; b.v_stride = v_stride; b.n = n; b.out_dev = out_dev; b.out_stride = out_stride;
    return zk_ecdsa_open_batches(&b, 1, opts, out);
}
extern "C" int zk_ecdsa_verify(const uint8_t* bytes, uint32_t layout, const uint32_t* v, uint32_t v_stride, uint64_t n, uint32_t opts,
                               uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_ecdsa_verify: result is null");
    zk_session* s = nullptr;
    const int rc = zk_ecdsa_open(bytes, layout, v, v_stride, n, nullptr, 0, opts, &s);
    return rc ? rc : one_shot(s, status_out, result);
}

// ---- witness assignment (state_circuit.py:827-934, bytecode_circuit.py:104-167, evm_circuit/typing.py CopyCircuit.copy): the device
// functions of csrc/state_assign.hpp / bytecode_assign.hpp / copy_assign.hpp in plain host loops.  A pass (zk_launch) computes the
// outputs into session-owned buffers; zk_*_assign_read copies them out.  Output pointers at open need ZK_OPT_DEVICE_PTRS, which
// has no meaning here.
static void state_assign_pass(zk_session* s) {
    AssignArgs& a = s->assign;
    const u64 n = a.n;
    std::fill(s->a32[1].begin(), s->a32[1].end(), ZK_EMPTY_SLOT);  // slots
    std::fill(s->a32[2].begin(), s->a32[2].end(), ASG_NONE);       // first
    // ops are inserted in REVERSE order so that the "smallest index wins" rule is what makes the result right
    for (u64 i = n; i-- > 0;)
        if (asg_has_key(asg_slot(a, ASG_TAG, i))) asg_insert(a, (u32)i);
    u32 r = 0;
    u32* first = s->a32[2].data();
    u32* rank = s->a32[3].data();
    for (u64 i = 0; i < n; i++) {
        if (!asg_has_key(asg_slot(a, ASG_TAG, i))) continue;
        first[i] = asg_find_first(a, (u32)i);
        if (first[i] == (u32)i) { rank[i] = r; asg_write_mpt(a, i, r); r++; }
    }
    s->n_mpt = r;
    u32 nxt = ASG_NONE;
    for (u64 i = n; i-- > 0;) {
        const u64 root = 3ull + 5ull * (nxt == ASG_NONE ? r : rank[first[nxt]]);
        s->status[i] = asg_write_row(a, i, root, first[i] == (u32)i);
        if (first[i] != ASG_NONE) nxt = (u32)i;
    }
}
extern "C" int zk_state_assign_open(const uint64_t* ops, const uint32_t* op_flags, uint64_t n, uint64_t* rows_dev, uint32_t* row_flags_dev,
                                    uint64_t* mpt_dev, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_state_assign_open");
    ARG_TRY(out && ops && op_flags && n > 0 && n < (1ull << 31) && !rows_dev && !row_flags_dev && !mpt_dev, "zk_state_assign_open: bad arguments");
    zk_session* s = new_session(n, false);
    s->a64[0].assign(ops, ops + n * ASG_NSLOTS * 4);
    s->a32[0].assign(op_flags, op_flags + n);
    const bool compact = opts & ZK_OPT_STATE_COMPACT;
    s->a64[1].assign(n * (compact ? 15 : ASG_ROW_NCELLS) * 4, 0);  // rows
    s->a64[2].assign(n * ASG_MPT_NCELLS * 4, 0);  // mpt
    s->out32.assign(n, 0);                        // row flags
    u32 cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    s->a32[1].assign(cap, ZK_EMPTY_SLOT);
    s->a32[2].assign(n, ASG_NONE);
    s->a32[3].assign(n, 0);
    AssignArgs& a = s->assign;
    a.ops = s->a64[0].data(); a.op_flags = s->a32[0].data(); a.n = n;
    a.rows = s->a64[1].data(); a.row_flags = s->out32.data(); a.mpt = s->a64[2].data();
    a.slots = s->a32[1].data(); a.mask = cap - 1; a.first = s->a32[2].data(); a.rank = s->a32[3].data();
    a.nb = 0; a.blk_cnt = nullptr; a.blk_next = nullptr;
    a.compact = compact ? 1u : 0u;
    s->pass = state_assign_pass;
    s->assign_kind = 1;
    *out = s;
    return 0;
}
extern "C" int zk_state_assign_read(zk_session* s, uint64_t* rows_host, uint32_t* row_flags_host, uint64_t* mpt_host, uint64_t mpt_capacity_rows,
                                    uint64_t* n_mpt_out) {
    ARG_TRY(s && s->assign_kind == 1, "zk_state_assign_read: bad arguments");
    if (n_mpt_out) *n_mpt_out = s->n_mpt;
    if (rows_host) memcpy(rows_host, s->a64[1].data(), s->a64[1].size() * 8);
    if (row_flags_host) memcpy(row_flags_host, s->out32.data(), s->out32.size() * 4);
    if (mpt_host) {
        ARG_TRY(mpt_capacity_rows >= s->n_mpt, "zk_state_assign_read: mpt buffer too small");
        memcpy(mpt_host, s->a64[2].data(), (size_t)s->n_mpt * ASG_MPT_NCELLS * 32);
    }
    return 0;
}
extern "C" int zk_state_assign(const uint64_t* ops, const uint32_t* op_flags, uint64_t n, uint64_t* rows_out, uint32_t* row_flags_out,
                               uint64_t* mpt_out, uint64_t* n_mpt_out, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result && n_mpt_out, "zk_state_assign: null output");
    zk_session* s = nullptr;
    int rc = zk_state_assign_open(ops, op_flags, n, nullptr, nullptr, nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc) rc = zk_state_assign_read(s, rows_out, row_flags_out, mpt_out, n, n_mpt_out);
    if (!rc && status_out) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// ---- RW table -> State-circuit operations (state_rekey.hpp): the device's per-row functions and its compact-key plan, the sort
// itself a std::stable_sort over the compact keys (the device sorts the same keys with LSD radix passes).
#include "state_rekey_plan.hpp"
#include <algorithm>
static void rekey_scan_host(const u64* rw, u64 n, std::vector<u32>& masks) {
    masks.assign(2 * RWK_MASK_WORDS_H + RWK_NCLASSES, 0u);
    for (u64 i = 0; i < n; i++) {
        const RwkKey k = rwk_key(rw + i * (RWK_RW_NCELLS * 4));
        masks[2 * RWK_MASK_WORDS_H + k.cls]++;
        if (k.cls == RWK_CLASS_DROPPED) continue;
        for (int f = 0; f < RWK_NFIELDS; f++)
            for (int w = 0; w < 8; w++) {
                const u32 slot = (k.cls * RWK_NFIELDS + f) * 8 + w;
                masks[slot] |= k.f[f].v[w];
                masks[RWK_MASK_WORDS_H + slot] |= ~k.f[f].v[w];
            }
    }
}
static void state_rekey_pass(zk_session* s) {
    const u64 n = s->n;
    const u64* rw = s->a64[0].data();
    const u32* fl = s->a32[0].data();
    RwkPlan plan;
    memcpy(&plan, s->rekey_plan.data(), sizeof(plan));
    // ranks of the plan's wide fields: rank = number of class members with a smaller value
    std::vector<u32> ranks[RWK_NFIELDS];
    for (size_t j = 0; j + 3 < s->rekey_jobs.size(); j += 4) {
        const u32 cls = s->rekey_jobs[j], field = s->rekey_jobs[j + 1];
        if (ranks[field].empty()) ranks[field].assign(n, 0u);
        std::vector<std::pair<Fr, u32>> mem;
        for (u64 i = 0; i < n; i++) {
            const RwkKey k = rwk_key(rw + i * (RWK_RW_NCELLS * 4));
            if (k.cls == cls) mem.push_back({k.f[field], (u32)i});
        }
        std::sort(mem.begin(), mem.end(), [](const std::pair<Fr, u32>& x, const std::pair<Fr, u32>& y) { return fr_lt(x.first, y.first); });
        u32 r = 0;
        for (size_t q = 0; q < mem.size(); q++) {
            if (q && fr_lt(mem[q - 1].first, mem[q].first)) r = (u32)q;
            ranks[field][mem[q].second] = r;
        }
    }
    const u32 kw = plan.key_words;
    std::vector<u32> keys((size_t)kw * n);
    for (u64 i = 0; i < n; i++) {
        const RwkKey k = rwk_key(rw + i * (RWK_RW_NCELLS * 4));
        u32 rk[RWK_NFIELDS];
        for (int f = 0; f < RWK_NFIELDS; f++) rk[f] = ranks[f].empty() ? 0u : ranks[f][i];
        rwk_pack(plan, plan.cls[k.cls], k, rk, keys.data() + i, n);
        s->status[i] = k.status;
    }
    std::vector<u32> order(n);
    for (u64 i = 0; i < n; i++) order[i] = (u32)i;
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) {
        for (u32 w = 0; w < kw; w++) {
            const u32 a = keys[(size_t)w * n + x], b = keys[(size_t)w * n + y];
            if (a != b) return a < b;
        }
        return false;
    });
    u64* ops = s->a64[1].data();
    u32* of = s->out32.data();
    rwk_emit_start(ops, of, s->n_ops);
    for (u64 j = 1; j < s->n_ops; j++) {
        const u64* p = rw + (u64)order[j - 1] * (RWK_RW_NCELLS * 4);
        const RwkKey k = rwk_key(p);
        rwk_emit(ops, of, s->n_ops, j, k, rwk_op(p, fl[order[j - 1]], k));
    }
}
extern "C" int zk_state_ops_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint64_t* ops_dev, uint32_t* op_flags_dev,
                                         uint32_t opts, uint64_t* n_ops_out, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_state_ops_from_rw_open");
    ARG_TRY(out && rw && n > 0 && n < (1ull << 31) && !ops_dev && !op_flags_dev, "zk_state_ops_from_rw_open: bad arguments");
    zk_session* s = new_session(n, false);
    s->a64[0].assign(rw, rw + n * RWK_RW_NCELLS * 4);
    if (rw_flags) s->a32[0].assign(rw_flags, rw_flags + n);
    else s->a32[0].assign(n, 0u);
    std::vector<u32> masks;
    rekey_scan_host(s->a64[0].data(), n, masks);
    RwkHostPlan hp;
    const char* e = getenv("ZK_REKEY_NO_RANKS");
    rwk_build_plan(masks.data(), !(e && e[0] == '1'), hp);
    s->rekey_plan.assign((sizeof(RwkPlan) + 3) / 4, 0u);
    memcpy(s->rekey_plan.data(), &hp.plan, sizeof(RwkPlan));
    for (const RwkRankJob& j : hp.jobs) { s->rekey_jobs.push_back(j.cls); s->rekey_jobs.push_back(j.field); s->rekey_jobs.push_back(j.base); s->rekey_jobs.push_back(j.count); }
    s->n_ops = 1 + hp.n_kept;
    s->a64[1].assign(s->n_ops * RWK_NSLOTS * 4, 0);
    s->out32.assign(s->n_ops, 0);
    s->pass = state_rekey_pass;
    s->assign_kind = 4;
    if (n_ops_out) *n_ops_out = s->n_ops;
    *out = s;
    return 0;
}
extern "C" int zk_state_ops_from_rw_read(zk_session* s, uint64_t* ops_host, uint32_t* op_flags_host, uint64_t* n_ops_out) {
    ARG_TRY(s && s->assign_kind == 4, "zk_state_ops_from_rw_read: bad arguments");
    if (n_ops_out) *n_ops_out = s->n_ops;
    if (ops_host) memcpy(ops_host, s->a64[1].data(), s->a64[1].size() * 8);
    if (op_flags_host) memcpy(op_flags_host, s->out32.data(), s->out32.size() * 4);
    return 0;
}
extern "C" int zk_state_ops_from_rw(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint64_t* ops_out, uint32_t* op_flags_out,
                                    uint64_t* n_ops_out, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result && n_ops_out, "zk_state_ops_from_rw: null output");
    zk_session* s = nullptr;
    int rc = zk_state_ops_from_rw_open(rw, rw_flags, n, nullptr, nullptr, opts, n_ops_out, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc) rc = zk_state_ops_from_rw_read(s, ops_out, op_flags_out, n_ops_out);
    if (!rc && status_out) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// RW table -> State rows in one session: here simply the two host passes back to back (the device reads its ops straight from
// the RW rows; the results are the same by construction of asg_slot_rw, which tests/test_state_rekey.py checks on the device).
extern "C" int zk_state_assign_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n_rw, uint64_t* rows_dev,
                                            uint32_t* row_flags_dev, uint64_t* mpt_dev, uint32_t opts, uint64_t* n_ops_out, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_state_assign_from_rw_open");
    ARG_TRY(out && rw && n_rw > 0 && !rows_dev && !row_flags_dev && !mpt_dev, "zk_state_assign_from_rw_open: bad arguments");
    zk_session* r = nullptr;
    uint64_t n_ops = 0;
    int rc = zk_state_ops_from_rw_open(rw, rw_flags, n_rw, nullptr, nullptr, opts, &n_ops, &r);
    if (rc) return rc;
    run_pass(r, nullptr);
    if (r->fail_count) {
        g_err = "zk_state_assign_from_rw_open: the RW table has rows the re-keying rejects (zk_state_ops_from_rw reports them)";
        zk_close(r);
        return -1;
    }
    rc = zk_state_assign_open(r->a64[1].data(), r->out32.data(), n_ops, nullptr, nullptr, nullptr, opts, out);
    zk_close(r);
    if (!rc && n_ops_out) *n_ops_out = n_ops;
    return rc;
}

// RW table -> State verdict: the device evaluates the rows where it computes them (state_fused.hpp); here the three host passes back
// to back — re-keying, assignment (15-cell rows), State circuit — with the errors of the first two reported by the open.
extern "C" int zk_state_verify_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n_rw, uint32_t opts, uint64_t* n_ops_out,
                                            zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_state_verify_from_rw_open");
    ARG_TRY(out && rw && n_rw > 0, "zk_state_verify_from_rw_open: bad arguments");
    zk_session* a = nullptr;
    uint64_t n_ops = 0;
    int rc = zk_state_assign_from_rw_open(rw, rw_flags, n_rw, nullptr, nullptr, nullptr, opts | ZK_OPT_STATE_COMPACT, &n_ops, &a);
    if (rc) return rc;
    run_pass(a, nullptr);
    if (a->fail_count) {
        char msg[200];
        snprintf(msg, sizeof msg, "zk_state_verify_from_rw: the State witness assignment failed for %llu ops (first: %llu, code 0x%08x; zk_state_assign_from_rw reports each)",
                 (unsigned long long)a->fail_count, (unsigned long long)a->first_row, (unsigned)a->first_code);
        g_err = msg;
        zk_close(a);
        return -1;
    }
    rc = zk_state_open(a->a64[1].data(), a->out32.data(), n_ops, a->a64[2].data(), a->n_mpt, opts | ZK_OPT_STATE_COMPACT, out);
    zk_close(a);
    if (!rc && n_ops_out) *n_ops_out = n_ops;
    return rc;
}
extern "C" int zk_state_verify_from_rw(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint32_t opts, uint32_t* status_out,
                                       uint64_t* n_ops_out, zk_result* result) {
    ARG_TRY(result, "zk_state_verify_from_rw: result is null");
    zk_session* s = nullptr;
    int rc = zk_state_verify_from_rw_open(rw, rw_flags, n, opts, n_ops_out, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && status_out) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// the block one-shot is four chains on four device streams: there is nothing to overlap on the host — callers of the CPU backend use the
// per-circuit entries (zkevm_specs_amd/super_circuit.py does, on host arrays)
extern "C" int zk_block_verify(const zk_block*, uint32_t, zk_result*, double*) {
    g_err = "zk_block_verify: not available on the CPU backend (it needs ZK_OPT_DEVICE_PTRS); use the per-circuit entries";
    return -1;
}

static void bytecode_assign_pass(zk_session* s) {
    BcaArgs& a = s->bca;
    for (u64 c = 0; c < a.n_chunks; c++) bca_chunk(a, c);
    for (u64 j = 0; j < a.n_codes; j++) bca_prefix_code(a, j);
    for (u64 c = 0; c < a.n_chunks; c++) bca_rlc_chunk(a, c);
    for (u64 i = 0; i < a.n_out; i++) bca_write_row(a, i);
}
extern "C" int zk_bytecode_assign_open(const uint64_t* in_rows, uint64_t n_rows, const uint64_t* offsets, const uint64_t* lengths, uint64_t n_codes,
                                       uint32_t k, const uint64_t* randomness, uint64_t* rows_dev, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_bytecode_assign_open");
    ARG_TRY(out && offsets && lengths && randomness && k >= 1 && k <= 30 && n_rows < (1ull << 31) && n_codes < (1ull << 31) && (in_rows || n_rows == 0) &&
            !rows_dev, "zk_bytecode_assign_open: bad arguments");
    ARG_TRY(offsets[0] == 0 && offsets[n_codes] == n_rows, "zk_bytecode_assign_open: offsets must span the rows");
    for (u64 j = 0; j < n_codes; j++) ARG_TRY(offsets[j] <= offsets[j + 1], "zk_bytecode_assign_open: offsets must be non-decreasing");
    zk_session* s = new_session(1ull << k, false);
    s->a64[0].assign(in_rows, in_rows + n_rows * 6 * 4);
    s->a64[1].assign(offsets, offsets + n_codes + 1);
    s->a64[2].assign(lengths, lengths + n_codes);
    std::vector<BcaChunk>& chunks = s->bca_chunks;
    s->a32[0].assign(n_codes + 1, 0);  // code_chunk0
    for (u64 j = 0; j < n_codes; j++) {
        s->a32[0][j] = (u32)chunks.size();
        for (u64 g = offsets[j]; g < offsets[j + 1]; g += BCA_CHUNK) {
            BcaChunk c;
            c.code = (u32)j; c.start = (u32)g;
            c.count = (u32)((offsets[j + 1] - g < BCA_CHUNK) ? offsets[j + 1] - g : BCA_CHUNK);
            c.first = g == offsets[j] ? 1u : 0u;
            chunks.push_back(c);
        }
    }
    s->a32[0][n_codes] = (u32)chunks.size();
    s->a64[3].assign(BCA_RPOW_ROWS * 4, 0);
    bca_fill_rpow(cell_of(randomness), s->a64[3].data());
    s->w64[0].assign(chunks.size() * 4 + 4, 0);  // chunk_acc
    s->w64[1].assign(chunks.size() * 4 + 4, 0);  // chunk_in
    s->w64[2].assign(n_rows * 4 + 4, 0);         // rlc
    s->w64[3].assign((size_t)(1ull << k) * 12 * 4, 0);  // output rows
    s->a32[1].assign(chunks.size() + 1, 0);      // chunk_m
    s->a32[2].assign(n_rows + 1, 0);             // row_code
    s->a32[3].assign(chunks.size() + 1, 0);      // chunk_state
    s->bca_map.assign((chunks.size() + 1) * BCA_MAP_STRIDE, 0);
    s->a8.assign(2 * n_rows + 2, 0);             // track
    BcaArgs& a = s->bca;
    a.in_rows = s->a64[0].data(); a.offsets = s->a64[1].data(); a.lengths = s->a64[2].data(); a.n_in = n_rows; a.n_codes = n_codes; a.n_out = 1ull << k;
    a.rpow = s->a64[3].data(); a.chunks = chunks.data(); a.code_chunk0 = s->a32[0].data(); a.n_chunks = chunks.size();
    a.track = s->a8.data(); a.chunk_acc = s->w64[0].data(); a.chunk_m = s->a32[1].data(); a.chunk_in = s->w64[1].data(); a.rlc = s->w64[2].data();
    a.row_code = s->a32[2].data(); a.rows = s->w64[3].data();
    a.chunk_map = s->bca_map.data(); a.chunk_state = s->a32[3].data();
    s->pass = bytecode_assign_pass;
    s->assign_kind = 2;
    *out = s;
    return 0;
}
extern "C" int zk_bytecode_assign_read(zk_session* s, uint64_t* rows_host) {
    ARG_TRY(s && rows_host && s->assign_kind == 2, "zk_bytecode_assign_read: bad arguments");
    memcpy(rows_host, s->w64[3].data(), s->w64[3].size() * 8);
    return 0;
}
extern "C" int zk_bytecode_assign(const uint64_t* in_rows, uint64_t n_rows, const uint64_t* offsets, const uint64_t* lengths, uint64_t n_codes, uint32_t k,
                                  const uint64_t* randomness, uint64_t* rows_out, uint32_t opts, zk_result* result) {
    ARG_TRY(result && rows_out, "zk_bytecode_assign: null output");
    zk_session* s = nullptr;
    int rc = zk_bytecode_assign_open(in_rows, n_rows, offsets, lengths, n_codes, k, randomness, nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc) rc = zk_bytecode_assign_read(s, rows_out);
    zk_close(s);
    return rc;
}

static void copy_assign_pass(zk_session* s) {
    CpaArgs& a = s->cpa;
    for (u64 c = 0; c < a.n_chunks; c++) cpa_chunk(a, c);
    for (u64 e = 0; e < a.n_events; e++) cpa_prefix_event(a, e);
    for (u64 j = 0; j < a.n_rows; j++) cpa_write_row(a, j);
}
extern "C" int zk_copy_assign_sizes(const zk_copy_events* t, uint32_t opts, uint64_t* n_rows, uint64_t* n_table, uint64_t* n_rw) {
    NO_DEVICE_PTRS(opts, "zk_copy_assign_sizes");
    ARG_TRY(t && t->n_events > 0 && t->events && t->data_offsets, "zk_copy_assign_sizes: bad arguments");
    CpaPlan pl;
    const int rc = cpa_plan(t->events, t->flags, t->data_offsets, t->n_events, pl);
    if (rc) return rc;
    if (n_rows) *n_rows = pl.n_rows;
    if (n_table) *n_table = pl.n_table;
    if (n_rw) *n_rw = pl.n_rw;
    return 0;
}
extern "C" int zk_copy_assign_open(const zk_copy_events* t, uint64_t* rows_dev, uint32_t* row_flags_dev, uint64_t* table_dev, uint64_t* rw_dev,
                                   uint32_t* rw_flags_dev, uint32_t opts, zk_session** out) {
    NO_DEVICE_PTRS(opts, "zk_copy_assign_open");
    ARG_TRY(t && out && t->n_events > 0 && t->n_events < (1ull << 31) && t->events && t->data_offsets && t->randomness, "zk_copy_assign_open: bad arguments");
    ARG_TRY(!rows_dev && !row_flags_dev && !table_dev && !rw_dev && !rw_flags_dev, "zk_copy_assign_open: output buffers need ZK_OPT_DEVICE_PTRS");
    zk_session* s = new_session(1, false);
    CpaPlan& pl = s->cpa_plan;
    const int rc = cpa_plan(t->events, t->flags, t->data_offsets, t->n_events, pl);
    if (rc) { delete s; return rc; }
    s->n = s->hi = pl.n_rows;
    s->status.assign(pl.n_rows ? pl.n_rows : 1, 0u);
    s->a64[0].assign(t->events, t->events + t->n_events * CPA_EV_NCELLS * 4);
    s->a16.assign(t->data, t->data + pl.n_data);
    s->a64[1].assign(CPA_RPOW_ROWS * 4, 0);
    cpa_fill_rpow(cell_of(t->randomness), s->a64[1].data());
    s->w64[0].assign(pl.chunks.size() * 4 + 4, 0);  // chunk_acc
    s->w64[1].assign(pl.chunks.size() * 4 + 4, 0);  // chunk_in
    s->w64[2].assign(t->n_events * 4 + 4, 0);       // ev_rlc
    s->w64[3].assign(pl.n_rlc * 4 + 4, 0);          // rlc
    s->a64[2].assign(pl.n_rows * CPA_ROW_NCELLS * 4 + 4, 0);   // rows
    s->a64[3].assign(pl.n_table * CPA_TABLE_NCELLS * 4 + 4, 0);  // table
    s->keccak_rows.assign(pl.n_rw * CPA_RW_NCELLS * 4 + 4, 0);  // rw rows
    s->out32.assign(pl.n_rows + 1, 0);              // row flags
    s->a32[0].assign(pl.n_rw + 1, 0);               // rw flags
    CpaArgs& a = s->cpa;
    a.events = s->a64[0].data(); a.ev = pl.ev.data(); a.row0 = pl.row0.data(); a.n_events = t->n_events; a.n_rows = pl.n_rows; a.data = s->a16.data();
    a.rpow = s->a64[1].data(); a.chunks = pl.chunks.data(); a.n_chunks = pl.chunks.size(); a.chunk_acc = s->w64[0].data(); a.chunk_in = s->w64[1].data();
    a.ev_rlc = s->w64[2].data(); a.rlc = s->w64[3].data(); a.rows = s->a64[2].data(); a.row_flags = s->out32.data(); a.table = s->a64[3].data();
    a.rw = s->keccak_rows.data(); a.rw_flags = s->a32[0].data();
    s->pass = copy_assign_pass;
    s->assign_kind = 3;
    *out = s;
    return 0;
}
extern "C" int zk_copy_assign_read(zk_session* s, uint64_t* rows_host, uint32_t* row_flags_host, uint64_t* table_host, uint64_t* rw_host,
                                   uint32_t* rw_flags_host) {
    ARG_TRY(s && s->assign_kind == 3, "zk_copy_assign_read: bad arguments");
    const CpaPlan& pl = s->cpa_plan;
    if (rows_host) memcpy(rows_host, s->a64[2].data(), (size_t)pl.n_rows * CPA_ROW_NCELLS * 32);
    if (row_flags_host) memcpy(row_flags_host, s->out32.data(), (size_t)pl.n_rows * 4);
    if (table_host && pl.n_table) memcpy(table_host, s->a64[3].data(), (size_t)pl.n_table * CPA_TABLE_NCELLS * 32);
    if (rw_host && pl.n_rw) memcpy(rw_host, s->keccak_rows.data(), (size_t)pl.n_rw * CPA_RW_NCELLS * 32);
    if (rw_flags_host && pl.n_rw) memcpy(rw_flags_host, s->a32[0].data(), (size_t)pl.n_rw * 4);
    return 0;
}
extern "C" int zk_copy_assign(const zk_copy_events* t, uint64_t* rows_out, uint32_t* row_flags_out, uint64_t* table_out, uint64_t* rw_out,
                              uint32_t* rw_flags_out, uint32_t opts, zk_result* result) {
    ARG_TRY(result && rows_out && row_flags_out, "zk_copy_assign: null output");
    zk_session* s = nullptr;
    int rc = zk_copy_assign_open(t, nullptr, nullptr, nullptr, nullptr, nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc) rc = zk_copy_assign_read(s, rows_out, row_flags_out, table_out, rw_out, rw_flags_out);
    zk_close(s);
    return rc;
}
