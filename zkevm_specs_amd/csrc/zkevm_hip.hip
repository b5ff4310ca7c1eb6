// libzkevm_hip.so — HIP kernels (gfx950) + C ABI (include/zkevm_hip.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <chrono>
#include <string>
#include <vector>

#include <atomic>
#include <mutex>
#include <dlfcn.h>

#include "../../include/zkevm_hip.h"
#include "kernels.hpp"
#include "dist_tally.hpp"
#include "host_index.hpp"
#include "code_dir_build.hpp"

// ---------------------------------------------------------------------------------------
// engine state
// ---------------------------------------------------------------------------------------
// Process-wide state is limited to one lazily created stream per device (immutable once created).  Everything else is
// per thread (the "current" device / stream that zk_*_open captures, the error text) or per session (device, stream,
// buffers, events): sessions are independent contexts, calls on different sessions may come from different threads.
#define ZK_MAX_DEVICES 64
#define ZK_ECDSA_CHUNK_LANES (1ull << 17)  // lanes per ECDSA launch (x 1,440 B of key tables = 189 MB); a multiple of 64
static std::mutex g_dev_mutex;
static hipStream_t g_own_stream[ZK_MAX_DEVICES] = {nullptr};
static hipStream_t g_batch_stream[ZK_MAX_DEVICES][2] = {{nullptr}};
// EVM passes: the warm and cold builds walk their own lane ranges of the sorted mapping and are independent of the hot build's;
// they run on this stream, forked from / joined to the session's stream by events, so a pass lasts max(hot, warm + cold) instead
// of their sum.  Opt-in per session (ZK_OPT_SIDE_STREAM): the fork / join barriers cost more than the two launches when the
// session has the device to itself.
static hipStream_t g_side_stream[ZK_MAX_DEVICES] = {nullptr};

static u32* g_secp_comb[ZK_MAX_DEVICES] = {nullptr};  // per device: the ECDSA kernel's 8-bit fixed-base table of G (secp256k1.hpp), built at the first ECDSA open  // zk_evm_verify_batch: the two pipeline slots of a device
static void* g_zero_row[ZK_MAX_DEVICES] = {nullptr};  // 512 zero bytes per device: "row 0" of every empty table
static thread_local int t_device = -1;              // device selected by this thread's last zk_init
static thread_local hipStream_t t_stream = nullptr;  // stream new sessions of this thread are bound to
static thread_local std::string g_err;
static thread_local double t_host_phase[4] = {0, 0, 0, 0};  // zk_last_host_phases: host microseconds inside open / launch / collect / close of the last one-shot
static thread_local double t_timing_sum[3] = {0, 0, 0};  // zk_timing_sums
static thread_local uint64_t t_timing_count = 0;
static thread_local double t_timing[3] = {0, 0, 0};  // zk_last_timing: open span, pass span, first open dispatch -> last pass dispatch (ms)

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char buf_[512];                                                                   \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                     \
            g_err = buf_;                                                                     \
            return -(int)e_ - 1000;                                                           \
        }                                                                                     \
    } while (0)
#define ARG_TRY(cond, msg)  \
    do {                    \
        if (!(cond)) {      \
            g_err = msg;    \
            return -1;      \
        }                   \
    } while (0)

extern "C" const char* zk_last_error(void) { return g_err.c_str(); }

// The circuits of a block run as concurrent sessions, one HIP stream each (SuperCircuit: six).  The HIP runtime multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run back to back: with 4 queues the
// EVM and State sessions of the 2^20-row block landed on one queue and the pass took their SUM (0.49 ms vs 0.37 ms with 8).  The
// runtime reads the variable at its first API call, so a default set when this library is loaded is early enough for any host that
// has not used HIP yet; a host that has, or that sets the variable itself, keeps its own choice.  16 since round 4: a process
// that also holds the batch entry's two pipeline streams, the side stream and a Tx / Sig pass's streams mapped two of the block's
// sessions onto one of 8 queues again (0.44 ms per block pass inside the default bench line against 0.35 alone; 0.35 with 16).
__attribute__((constructor)) static void zk_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
static void arena_release_all();

// The four streams of zk_block_verify's chains.  Created when the device is first selected, not when the first block arrives: the
// runtime multiplexes streams onto a fixed number of hardware queues in creation order, and a process that has made many streams by
// then (PyTorch creates its pools of 32 per priority at once) leaves late-comers on shared queues — two chains of a block behind
// each other: 1.08-1.11 ms per block instead of 0.94 (profiles/r06_block_oneshot_v3.txt).
static hipStream_t g_block_stream[ZK_MAX_DEVICES][4] = {{nullptr}};
static int block_streams_ensure(int device) {  // g_dev_mutex held
    for (int c = 0; c < 4; c++)
        if (!g_block_stream[device][c]) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            // the State chain on the low priority, the three chains of short kernels on the high one: 0.96 ms per block against 1.02 the other
            // way round (ZK_BLOCK_STATE_PRIO=hi) and 1.00 without priorities (ZK_BLOCK_PRIO=none).  (A CU-mask split between the State chain
            // and the others was measured: no gain at 4 / 8 / 16 of every 32 CUs — the chains slow each other through the memory system,
            // not through shared CUs; profiles/r06_experiments.txt)
            static const bool state_low = [] { const char* e = getenv("ZK_BLOCK_STATE_PRIO"); return !(e && e[0] == 'h'); }();
            static const bool flat = [] { const char* e = getenv("ZK_BLOCK_PRIO"); return e && e[0] == 'n'; }();
            if (flat) HIP_TRY(hipStreamCreateWithFlags(&g_block_stream[device][c], hipStreamNonBlocking));
            else HIP_TRY(hipStreamCreateWithPriority(&g_block_stream[device][c], hipStreamNonBlocking, (c == 0) == state_low ? lo : hi));
        }
    return 0;
}
extern "C" int zk_init(int device) {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    ARG_TRY(device >= 0 && device < count && device < ZK_MAX_DEVICES, "zk_init: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    hipStream_t own;
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        if (!g_own_stream[device]) {
            HIP_TRY(hipStreamCreateWithFlags(&g_own_stream[device], hipStreamNonBlocking));
            static const bool lazy = [] { const char* e = getenv("ZK_BLOCK_STREAMS_LAZY"); return e && e[0] == '1'; }();  // A/B aid
            if (!lazy) { int brc = block_streams_ensure(device); if (brc) return brc; }
        }
        own = g_own_stream[device];
        if (!g_zero_row[device]) {
            HIP_TRY(hipMalloc(&g_zero_row[device], 512));
            HIP_TRY(hipMemset(g_zero_row[device], 0, 512));
        }
    }
    // a stream set by zk_set_stream belongs to the device it was set on: re-selecting the same device keeps it, selecting
    // another device falls back to that device's own stream (never a handle of the previous device)
    if (t_device != device || !t_stream) t_stream = own;
    t_device = device;
    return 0;
}
extern "C" void zk_shutdown(void) {
    // sessions own their buffers (zk_close); the per-device streams and the buffer arena are released here.  Callers close sessions first.
    arena_release_all();
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    for (int d = 0; d < ZK_MAX_DEVICES; d++)
        if (g_own_stream[d]) {
            if (hipSetDevice(d) == hipSuccess) {
                (void)hipStreamDestroy(g_own_stream[d]);
                if (g_zero_row[d]) (void)hipFree(g_zero_row[d]);
                if (g_secp_comb[d]) (void)hipFree(g_secp_comb[d]);
                g_secp_comb[d] = nullptr;
            }
            for (int k = 0; k < 2; k++) {
                if (g_batch_stream[d][k]) (void)hipStreamDestroy(g_batch_stream[d][k]);
                g_batch_stream[d][k] = nullptr;
            }
            if (g_side_stream[d]) (void)hipStreamDestroy(g_side_stream[d]);
            g_side_stream[d] = nullptr;
            for (int k = 0; k < 4; k++) {
                if (g_block_stream[d][k]) (void)hipStreamDestroy(g_block_stream[d][k]);
                g_block_stream[d][k] = nullptr;
            }
            g_own_stream[d] = nullptr;
            g_zero_row[d] = nullptr;
        }
    t_stream = nullptr;
    t_device = -1;
}
extern "C" int zk_set_stream(void* st) {
    ARG_TRY(t_device >= 0, "zk_set_stream: call zk_init first");
    t_stream = st ? (hipStream_t)st : g_own_stream[t_device];
    return 0;
}
struct zk_session;
static int session_rebind_stream(zk_session* s, hipStream_t st);
extern "C" int zk_session_set_stream(zk_session* s, void* st) {
    ARG_TRY(s, "zk_session_set_stream: null session");
    return session_rebind_stream(s, (hipStream_t)st);
}

// ---------------------------------------------------------------------------------------
// tally + index kernels
// ---------------------------------------------------------------------------------------
__global__ void tally_reset_kernel(ZkTally* t, u32* counter = nullptr) {
    t[threadIdx.x].fail_count = 0ull;
    t[threadIdx.x].first_fail = ~0ull;
    if (counter && threadIdx.x == 0) *counter = 0u;  // EVM sessions in trace order: the deferred-pair count of the pass
}


__global__ void slots_fill_kernel(u32* slots, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = ZK_EMPTY_SLOT;
}

// unless_dense: the verdict of rw_dense_check_kernel (earlier on the stream) — a dense RW table is looked up by position and its
// open-addressing index is never read (copy_rw_lookup): neither filled nor built then
template <u64 (*HASH)(const ZkTable&, u32)>
__global__ void index_build_kernel(ZkTable t, u32* slots, const ZkRwMeta* unless_dense = nullptr) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.n || (unless_dense && unless_dense->dense)) return;
    u32 s = (u32)HASH(t, r) & t.mask;
    while (atomicCAS(&slots[s], ZK_EMPTY_SLOT, r) != ZK_EMPTY_SLOT) s = (s + 1) & t.mask;
}
__global__ void slots_fill_unless_dense_kernel(u32* slots, u32 n, const ZkRwMeta* unless_dense) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !unless_dense->dense) slots[i] = ZK_EMPTY_SLOT;
}
// MPT index: slot value = row | hash fingerprint (state_mpt_slot_value)
__global__ void mpt_index_build_kernel(ZkTable t, u32* slots) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.n) return;
    const u64 h = state_mpt_key_hash(t, r);
    u32 s = (u32)h & t.mask;
    const u32 v = state_mpt_slot_value(t, r, h);
    while (atomicCAS(&slots[s], ZK_EMPTY_SLOT, v) != ZK_EMPTY_SLOT) s = (s + 1) & t.mask;
}
// RW-table density check (see ZkRwMeta): meta->dense must be pre-set to 1.
__global__ void rw_dense_check_kernel(ZkTable t, ZkRwMeta* meta) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.n) return;
    const u64* p = t.cells + (u64)r * RW_NCELLS * 4;
    const u64* p0 = t.cells;
    const u64 base = p0[0];
    const bool ok = (p0[1] | p0[2] | p0[3]) == 0 && (p[1] | p[2] | p[3]) == 0 && p[0] == base + r && base + r >= base;
    if (r == 0) meta->base = base;
    if (!ok) atomicAnd(&meta->dense, 0u);
}

// Counting sort of the step pairs by (group, state): histogram, scan, scatter.
struct EvmSortArgs {  // one pass of the counting sort (evm_build_perm); also rides on the session-open launches for the first pass
    const u64* steps;
    u32 n_pairs;
    u32* hist;       // this pass's histogram (zero on entry)
    u32* hist_next;  // the other buffer: cleared for the pass after this one
    u32* taken;      // per-bin scatter cursors
    uint16_t* bin16;
    u32* group_start;
    u32* perm;
    ZkTally* tally;
    u32* defer_count;  // reset with the tally
    u32* early_host;   // the session's page-locked block (device alias), words EVM_EARLY_WORD..: warm lanes, cold lanes, sequence number
    u32 early_seq;     // (nullptr / 0: nobody is waiting for the lane ranges)
};
#define EVM_EARLY_WORD 36u  // behind the result block (32 words) and its flag word
__device__ __forceinline__ void evm_state_hist_body(u32 vblock, const u64* steps, u32 n_pairs, u32* hist, u32* taken, uint16_t* bin16, ZkTally* tally,
                                                    u32* defer_count) {
    __shared__ u32 local[EVM_N_BINS];
    if (vblock == 0) {
        if (threadIdx.x == 0 && defer_count) *defer_count = 0u;
        if (threadIdx.x == 0) {  // fused tally reset (saves a launch per pass)
            tally->fail_count = 0ull;
            tally->first_fail = ~0ull;
        }
        for (u32 k = threadIdx.x; k < EVM_N_BINS; k += blockDim.x) taken[k] = 0;  // the scatter's per-bin cursors
    }
    for (u32 k = threadIdx.x; k < EVM_N_BINS; k += blockDim.x) local[k] = 0;
    __syncthreads();
    u32 i = vblock * blockDim.x + threadIdx.x;
    if (i < n_pairs) {
        const u32 bin = evm_state_bin((u32)steps[((u64)i * STEP_NCELLS + S_STATE) * 4]);
        bin16[i] = (uint16_t)bin;  // the scatter reads this compact copy instead of the 416-byte-strided state cells
        atomicAdd(&local[bin], 1u);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < EVM_N_BINS; k += blockDim.x)
        if (local[k]) atomicAdd(&hist[k], local[k]);
}
__global__ void evm_state_hist_kernel(const u64* steps, u32 n_pairs, u32* hist, u32* taken, uint16_t* bin16, ZkTally* tally, u32* defer_count) {
    evm_state_hist_body(blockIdx.x, steps, n_pairs, hist, taken, bin16, tally, defer_count);
}
// Scatter with block-level aggregation.  Every block scans the (complete) histogram itself — 512 bins, Hillis-Steele in
// LDS — instead of waiting for a separate one-block scan launch; block 0 publishes the group boundaries and clears the
// OTHER histogram buffer for the next pass (the two alternate).  Ranks inside a block come from LDS atomics, one global
// atomic per (block, bin present).  Order inside a bin is irrelevant for correctness.
// Every hot bin's lane range is padded to a multiple of 64 (pad lanes = EVM_NO_PAIR): a wavefront never holds two execution
// states.  A mixed wavefront runs both gadget bodies one after the other, and with the longest states sorted first those
// boundary wavefronts (STOP + ADDMOD, MEMORY + SSTORE: 190k cycles against a 118k median) were the last to leave the kernel.
// (block size: at least EVM_N_BINS threads)
__device__ __forceinline__ void evm_state_scatter_body(u32 vblock, const uint16_t* bin16, u32 n_pairs, const u32* hist, u32* hist_next,
                                                       u32* taken, u32* group_start, u32* perm, u32* early_host = nullptr, u32 early_seq = 0u) {
    __shared__ u32 sa[EVM_N_BINS], sb[EVM_N_BINS];
    __shared__ u32 local[EVM_N_BINS];
    __shared__ u32 base[EVM_N_BINS];
    __shared__ u32 gsh[EVM_N_GROUPS + 1];
    const u32 k = threadIdx.x;
    u32 c = 0, c_real = 0;
    if (k < EVM_N_BINS) {
        c_real = hist[k];
        c = k < (u32)EVM_GROUP_WARM * 128u ? ((c_real + 63u) & ~63u) : c_real;  // hot bins: whole wavefronts
        sa[k] = c;
        local[k] = 0;
        if (vblock == 0) hist_next[k] = 0;
    }
    __syncthreads();
    u32* src = sa;
    u32* dst = sb;
    for (u32 off = 1; off < EVM_N_BINS; off <<= 1) {
        if (k < EVM_N_BINS) dst[k] = src[k] + (k >= off ? src[k - off] : 0u);
        __syncthreads();
        u32* t = src; src = dst; dst = t;
    }
    u32 excl = 0;
    if (k < EVM_N_BINS) {
        excl = src[k] - c;
        if (vblock == 0) {
            if ((k & 127u) == 0) group_start[k >> 7] = gsh[k >> 7] = excl;
            if (k == EVM_N_BINS - 1) group_start[EVM_N_GROUPS] = gsh[EVM_N_GROUPS] = src[k];
            for (u32 j = c_real; j < c; j++) perm[excl + j] = EVM_NO_PAIR;
        }
    }
    const u32 i = vblock * blockDim.x + threadIdx.x;
    u32 bin = 0, rank = 0;
    if (i < n_pairs) {
        bin = bin16[i];
        rank = atomicAdd(&local[bin], 1u);
    }
    __syncthreads();
    if (vblock == 0 && k == 0 && early_host) {
        // The host learns how many warm / cold lanes this witness has while the evaluation kernels are still queued behind this launch
        // (zk_launch's lazy tail: an empty range is then never launched).  Two words + a sequence number into the session's page-locked
        // block, from the first block of a short launch: nothing waits for it but the end of this kernel.
        __hip_atomic_store(&early_host[EVM_EARLY_WORD], gsh[EVM_GROUP_WARM + 1] - gsh[EVM_GROUP_WARM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&early_host[EVM_EARLY_WORD + 1], gsh[EVM_GROUP_COLD + 1] - gsh[EVM_GROUP_COLD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&early_host[EVM_EARLY_WORD + 2], early_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (k < EVM_N_BINS && local[k]) base[k] = excl + atomicAdd(&taken[k], local[k]);
    __syncthreads();
    if (i < n_pairs) perm[base[bin] + rank] = i;
}
__global__ __launch_bounds__(1024) void evm_state_scatter_kernel(const uint16_t* bin16, u32 n_pairs, const u32* hist, u32* hist_next,
                                                                 u32* taken, u32* group_start, u32* perm) {
    evm_state_scatter_body(blockIdx.x, bin16, n_pairs, hist, hist_next, taken, group_start, perm);
}
// ---------------------------------------------------------------------------------------
// EVM session open: three launches, whatever the tables (the host enqueues them and returns; nothing is read back).
//   evm_open_fill_kernel    every slot table / "min" array of the session to 0xFF.., every counter / status array to 0
//   evm_open_phase1_kernel  block ranges: small-table index inserts + EndBlock aggregates + tally reset | RW table:
//                           density verdict + packed key records (one read of the key cells) | bytecode directory events
//   evm_open_phase2_kernel  block ranges: directory entries | generic RW index (does nothing when the rows are dense)
// A launch costs the host ~10 us on this stack and the device ~3 us of gap: round 2's open was 20 launches, 15 hipMallocs and
// two host synchronisations (0.49 ms for 2^18 steps); the independent builds now share launches and overlap on the device
// (the RW pass streams HBM, the others are latency-bound).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void evm_open_fill_kernel(uint4* ff, u64 n_ff, uint4* zero, u64 n_zero) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const uint4 F = make_uint4(~0u, ~0u, ~0u, ~0u), Z = make_uint4(0u, 0u, 0u, 0u);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_ff; i += stride) ff[i] = F;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += stride) zero[i] = Z;
}
// RW table: ONE read of the key cells of every row gives (a) the density verdict of the dense index (ZkRwMeta / EvmDyn:
// rw_counter of row r == rw_counter of row 0 + r) and (b) the packed key record (RwKey).  The verdict stays on the device
// (EvmDyn::rw_sparse); the packed records are only consulted when it is "dense".
// Four lanes share a row: the 192 bytes of its six key cells are 12 chunks of 16 bytes, lane q of the quad loads chunks q,
// q + 4, q + 8 — every load instruction of the wavefront is 16 rows x 64 contiguous bytes instead of 64 rows x 16 bytes
// 448 bytes apart — and the quad ORs its partial key words together (rw_pack_row's record, bit for bit).
//   lane 0: rw_counter[0..128)   tag[0..128)   address[0..128)        lane 2: rw[0..128)   id[0..128)   field_tag[0..128)
//   lane 1: rw_counter[128..256) tag[128..256) address[128..256)      lane 3: rw[128..)    id[128..)    field_tag[128..)
__device__ __forceinline__ void rw_prepare_quad(const ZkTable& t, u64* keys, EvmDyn* dyn, u32 vthread) {
    const u32 r = vthread >> 2, q = vthread & 3u;
    const bool in = r < t.n;
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0;
    u64 base = 0;
    u32 flags = 3u;
    if (in) {
        const uint4* p = reinterpret_cast<const uint4*>(t.cells + (u64)r * RW_NCELLS * 4) + q;
        c0 = p[0];
        c1 = p[4];
        c2 = p[8];
        if (q == 0) {
            base = t.cells[0];
            if (t.flags) flags = t.flags[r];
        }
    }
    // per-lane verdicts and partial key words
    bool fit, dense_ok;
    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    const bool z0hi = (c0.y | c0.z | c0.w) == 0u, z1hi = (c1.y | c1.z | c1.w) == 0u, z2hi = (c2.y | c2.z | c2.w) == 0u;
    const bool z0 = z0hi && c0.x == 0u, z1 = z1hi && c1.x == 0u, z2 = z2hi && c2.x == 0u;
    if (q == 0) {        // rw_counter low half, tag low half, address low half
        const u64 rwc = (u64)c0.x | ((u64)c0.y << 32);
        dense_ok = (c0.z | c0.w) == 0u && rwc == base + r && base + r >= base;
        fit = z1hi && c1.x <= 255u;
        w0 = (u64)(c1.x & 0xffu) << 8 | ((u64)(flags & 3u) << 56) | (1ull << 63);
        w2 = (u64)c2.x | ((u64)c2.y << 32);
        w3 = (u64)c2.z | ((u64)c2.w << 32);
    } else if (q == 1) { // the high halves: rw_counter and tag must be zero there, the address may reach bit 160
        dense_ok = z0;
        fit = z1 && z2hi;
        w0 = (u64)c2.x << 24;
    } else if (q == 2) { // rw, id, field_tag low halves
        dense_ok = true;
        fit = z0hi && c0.x <= 255u && (c1.z | c1.w) == 0u && z2hi && c2.x <= 255u;
        w0 = (u64)(c0.x & 0xffu) | ((u64)(c2.x & 0xffu) << 16);
        w1 = (u64)c1.x | ((u64)c1.y << 32);
    } else {             // rw, id, field_tag high halves: all zero in a record that fits
        dense_ok = true;
        fit = z0 && z1 && z2;
    }
    // quad reductions: the two verdicts by ballot, the key words by two xor-shuffle steps
    const u32 qshift = threadIdx.x & 60u;
    const bool fits = ((__ballot(fit) >> qshift) & 0xfull) == 0xfull;
    const unsigned long long nd = __ballot(in && !dense_ok);
    if (nd != 0ull && (threadIdx.x & 63u) == 0) atomicOr(&dyn->rw_sparse, 1u);
    if (r == 0 && q == 0 && in) dyn->rw_base = base;
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        w0 |= __shfl_xor(w0, m);
        w1 |= __shfl_xor(w1, m);
        w2 |= __shfl_xor(w2, m);
        w3 |= __shfl_xor(w3, m);
    }
    if (!fits) w0 = w1 = w2 = w3 = 0;
    if (in && q < 2) {  // lanes 0 and 1 store the record's two halves: 32 contiguous bytes per quad
        const u64 a = q == 0 ? w0 : w2, b2 = q == 0 ? w1 : w3;
        reinterpret_cast<uint4*>(keys + (u64)r * 4)[q] = make_uint4((u32)a, (u32)(a >> 32), (u32)b2, (u32)(b2 >> 32));
    }
}
#define EVM_OPEN_TABLES 8
struct EvmOpenTables {
    ZkTable t[EVM_OPEN_TABLES];      // tx, block, bytecode, copy, keccak, exp, sig, ecc (slots writable: they are this session's)
    u32 block_start[EVM_OPEN_TABLES + 2];  // first block of each table's range; [8] = withdrawals, [9] = end of the small tables
    const u64* wds;
    u32 n_wds;
    EvmDyn* dyn;
    ZkTally* tally;
    ZkTable rw;          // phase 1: rw_prepare over [rw_block0, dir_block0)
    u64* rw_keys;        // nullptr: generic indices only
    u32 dir_block0, rw_block0;  // phase-1 block ranges: [0, dir_block0) small tables, [dir_block0, hist_block0) directory rows, [hist_block0, rw_block0) histogram, then RW rows
    DirBuild dir;        // dir.n == 0: no directory
    const u64* steps;    // phase 1: step records (evm_step_record_quad) over [rec_block0, end)
    u32 n_steps;
    u32* step_recs;
    u32 rec_block0;
    EvmSortArgs sort;    // sort.perm != nullptr: the first pass's counting sort rides on the two launches (histogram in phase 1, scatter in phase 2)
    u32 hist_block0;
    u32 lat_period;      // phase 1: every lat_period-th block of the launch's first part is one of the latency-bound ranges' (0 / 1: they all come first)
#ifdef ZK_DIAG_P1
    u32 diag_skip;       // tuning builds only (tools/p1_ranges.py): ranges of phase 1 that return at once — results are INVALID
#endif
};
// small-table index inserts + EndBlock's whole-table aggregates over the tx and withdrawal rows (end_block.py:55-91;
// host_index.hpp's evm_aggregates_host is the CPU statement of the same counts)
__device__ __forceinline__ void evm_open_small_tables(const EvmOpenTables& o, u32 block) {
    int k = 0;
#pragma unroll
    for (int j = 1; j <= EVM_OPEN_TABLES; j++) k += block >= o.block_start[j] ? 1 : 0;
    const u32 r = (block - o.block_start[k]) * blockDim.x + threadIdx.x;
    if (k == EVM_OPEN_TABLES) {  // withdrawals: rows with a non-zero amount
        const bool nz = r < o.n_wds && !fr_is_zero(fr_load(o.wds + ((u64)r * 4 + 3) * 4));
        const unsigned long long b = __ballot(nz);
        if (b && (threadIdx.x & 63u) == 0) atomicAdd(&o.dyn->agg_total_wds, (u32)__popcll(b));
        return;
    }
    const ZkTable& t = o.t[k];
    const bool in = r < t.n;
    if (k == 0) {  // tx rows: CallerAddress rows count MAX_TXS / the txs present, TxInvalid rows the invalid ones
        bool caller = false, present = false, invalid = false, bad = false;
        if (in) {
            const Fr tag = zk_table_cell(t, r, 1), lo = zk_table_cell(t, r, 3), hi = zk_table_cell(t, r, 4);
            caller = fr_eq_u64(tag, 4);
            present = caller && !(fr_is_zero(lo) && fr_is_zero(hi));
            const bool inv_row = fr_eq_u64(tag, 10);
            bad = inv_row && t.flags && (t.flags[r] & 1u);
            invalid = inv_row && fr_eq_u64(lo, 1);
        }
        const unsigned long long bc = __ballot(caller), bp = __ballot(present), bi = __ballot(invalid), bb = __ballot(bad);
        if ((threadIdx.x & 63u) == 0) {
            if (bc) atomicAdd(&o.dyn->agg_max_txs, (u32)__popcll(bc));
            if (bp) atomicAdd(&o.dyn->agg_total_txs, (u32)__popcll(bp));
            if (bi) atomicAdd(&o.dyn->agg_invalid_txs, (u32)__popcll(bi));
            if (bb) atomicOr(&o.dyn->agg_bad_invalid_rows, 1u);
        }
    }
    if (!in) return;
    u64 h;
    switch (k) {
    case 0: h = tx_key_hash(t, r); break;
    case 1: h = blk_key_hash(t, r); break;
    case 2: h = bc_key_hash(t, r); break;
    case 3: h = copy_key_hash(t, r); break;
    case 4: h = keccak_key_hash(t, r); break;
    case 5: h = expt_key_hash(t, r); break;
    case 6: h = sig_key_hash(t, r); break;
    default: h = ecc_key_hash(t, r); break;
    }
    u32* slots = const_cast<u32*>(t.slots);
    u32 s = (u32)h & t.mask;
    while (atomicCAS(&slots[s], ZK_EMPTY_SLOT, r) != ZK_EMPTY_SLOT) s = (s + 1) & t.mask;
}
__global__ __launch_bounds__(256) void evm_open_phase1_kernel(EvmOpenTables o) {
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        o.tally[threadIdx.x].fail_count = 0ull;
        o.tally[threadIdx.x].first_fail = ~0ull;
    }
    // Virtual block order [0, rw_block0): the latency-bound ranges (dependent loads, atomics) | [rw_block0, rec_block0): the RW
    // rows, streaming | step records, streaming.  The ~1,400 latency-bound blocks used to be dispatched first: for the first
    // ~10 us they held two thirds of the chip's block slots and the stream ran at half rate (tools/p1_ranges.py: 14 us of the
    // launch).  They are now dealt into the RW range, one every lat_period blocks, so the stream has most of the slots from
    // the first cycle and the chains finish under it.
    u32 b = blockIdx.x;
    if (o.lat_period > 1u && b < o.rec_block0) {
        const u32 q = b / o.lat_period;
        b = (b == q * o.lat_period && q < o.rw_block0) ? q : o.rw_block0 + b - (q + 1u < o.rw_block0 ? q + 1u : o.rw_block0);
    }
#ifdef ZK_DIAG_P1
    {
        const u32 range = b < o.dir_block0 ? 1u : b < o.hist_block0 ? 2u : b < o.rw_block0 ? 4u : b < o.rec_block0 ? 8u : 16u;
        if (o.diag_skip & range) return;
    }
#endif
    if (b < o.dir_block0) evm_open_small_tables(o, b);
    else if (b < o.hist_block0) dirb_events_row(o.dir, (b - o.dir_block0) * blockDim.x + threadIdx.x);
    else if (b < o.rw_block0) evm_state_hist_body(b - o.hist_block0, o.sort.steps, o.sort.n_pairs, o.sort.hist, o.sort.taken, o.sort.bin16, o.sort.tally, o.sort.defer_count);
    else if (b < o.rec_block0) rw_prepare_quad(o.rw, o.rw_keys, o.dyn, (b - o.rw_block0) * blockDim.x + threadIdx.x);
    else evm_step_record_quad(o.steps, o.n_steps, o.step_recs, (b - o.rec_block0) * blockDim.x + threadIdx.x);
}
// Phase 2.  The generic RW index is only needed when the rows are not dense: the blocks are always launched (the host does
// not know the verdict), the work is conditional; grid-stride so that the idle case is 256 blocks that exit at once.
#define EVM_OPEN_RW_GENERIC_BLOCKS 64u
#define EVM_OPEN_P2_BLOCK 1024u  // the scatter scans EVM_N_BINS bins with one thread each
__global__ __launch_bounds__(1024) void evm_open_phase2_kernel(EvmOpenTables o, u32 scatter_blocks, u32 dir_blocks, u32 force_generic) {
    if (blockIdx.x < scatter_blocks) {  // the first pass's state-sorted permutation
        evm_state_scatter_body(blockIdx.x, o.sort.bin16, o.sort.n_pairs, o.sort.hist, o.sort.hist_next, o.sort.taken, o.sort.group_start, o.sort.perm,
                               o.sort.early_host, o.sort.early_seq);
        return;
    }
    const u32 b = blockIdx.x - scatter_blocks;
    if (b < dir_blocks) {
        dirb_finalize_entry(o.dir, b * blockDim.x + threadIdx.x);
        return;
    }
    if (!force_generic && o.dyn->rw_sparse == 0u) return;
    if (force_generic && b == dir_blocks && threadIdx.x == 0) o.dyn->rw_sparse = 1u;  // generic-index sessions never use the dense path
    u32* slots = const_cast<u32*>(o.rw.slots);
    const u32 stride = (gridDim.x - scatter_blocks - dir_blocks) * blockDim.x;
    for (u32 r = (b - dir_blocks) * blockDim.x + threadIdx.x; r < o.rw.n; r += stride) {
        u32 s = (u32)rw_key_hash(o.rw, r) & o.rw.mask;
        while (atomicCAS(&slots[s], ZK_EMPTY_SLOT, r) != ZK_EMPTY_SLOT) s = (s + 1) & o.rw.mask;
    }
}

__global__ void fr_op_kernel(int op, const u64* a, const u64* b, u64* out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = fr_load(a + 4 * i), y = fr_load(b + 4 * i), r;
    switch (op) {
    case 0: r = fr_add(x, y); break;
    case 1: r = fr_sub(x, y); break;
    case 2: r = fr_mul(x, y); break;
    case 3: r = fr_mont(x, y); break;
    case 4: r = fr_neg(x); break;
    case 5: r = fr_inv(x); break;
    case 6: r = fr_div(x, y); break;
    case 16: r = sp_mul_p(x, y); break;   // secp256k1 base field (unit-test hooks of csrc/secp256k1.hpp): a * b mod P, residues in and out
    case 17: r = sp_sqr_p(x); break;
    default: r = fr_zero();
    }
    for (int k = 0; k < 4; k++) out[4 * i + k] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
}

// The 128-byte result block travels to the host by a one-wavefront kernel at the end of the pass (ZK_POLL_RESULT=0: by a copy
// dispatch + hipStreamSynchronize, as until round 4) — 32 lanes store its 32 words into the session's page-locked (device-mapped, coherent) block, a system-scope fence, then the
// sequence number into the word behind it — and zk_collect polls that word instead of waiting for the runtime's completion signal
// of a copy dispatch.  (Not the in-kernel publication round 4 measured: nothing of the evaluation launches changes.)
__global__ __launch_bounds__(64) void evm_publish_kernel(const u32* d, u32* h, u32 seq) {
    const u32 i = threadIdx.x;
    if (i < 32u) __hip_atomic_store(&h[i], d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();  // one wavefront: every lane's store has left before the flag does
    if (i == 0u) __hip_atomic_store(&h[32], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ---------------------------------------------------------------------------------------
// sessions
// ---------------------------------------------------------------------------------------
// What zk_collect reads back from an EVM session, contiguous at the start of the session's zero-filled region.
struct EvmResultBlock {
    EvmDyn dyn;                      // n_deferred
    u32 n_deferred_twin;             // the deferred-pair counter of the odd passes of a resident session (EvmArgs::defer_count_twin)
    u32 pad0[(64 - sizeof(EvmDyn)) / 4 - 1];
    ZkTally tally[2];                // offset 64
    u32 group_start[EVM_N_GROUPS + 1];  // offset 96
    u32 pad1[8 - (EVM_N_GROUPS + 1)];
};
static_assert(sizeof(EvmDyn) <= 64 && sizeof(EvmResultBlock) == 128, "EvmResultBlock layout");
enum SessionKind { SESSION_REKEY = 14, SESSION_PICOPY = 13, SESSION_PI = 12, SESSION_CPA = 11, SESSION_STATE = 1, SESSION_EVM = 2, SESSION_BYTECODE = 3, SESSION_EXP = 4, SESSION_COPY = 5, SESSION_SIGN = 6, SESSION_KECCAK = 7, SESSION_ASSIGN = 8, SESSION_ECDSA = 9, SESSION_BCA = 10 };

struct zk_session {
    SessionKind kind;
    int device = t_device;            // captured at open: the session is its own context from here on
    hipStream_t stream = t_stream;
    u64 n = 0;                      // rows per pass
    u64 eval_lo = 0, eval_hi = 0;   // row sessions: rows [eval_lo, eval_hi) are evaluated (zk_set_range); 0, 0 = all n
    std::vector<void*> owned;       // device buffers (arena) returned at close
    std::vector<int> owned_class;   // their arena size classes
    ZkTally* d_tally = nullptr;
    u32* d_status = nullptr;        // internal per-row status (zeroed at open; what zk_read_status copies)
    bool status_external = false;   // the last pass wrote to the caller's status_dev instead
    std::vector<hipEvent_t> ev;     // start/stop pairs
    u32 launches = 0;               // since last collect
    u32 tally_pass = 0;             // passes of a twin-tally session since open
    ZkTally* tally_last = nullptr;  // the tally the latest pass accumulated into
    StateArgs state;
    EvmArgs evm;
    BytecodeArgs bytecode;
    ExpArgs exp;
    CopyArgs copy;
    SignArgs sign;
    KeccakGenArgs keccak_gen;
    AssignArgs assign;
    EcdsaArgs ecdsa;
    BcaArgs bca;
    CpaArgs cpa;
    PiArgs pi;
    PiCopyArgs picopy;
    RekeyArgs rekey;
    RwkPlan rekey_plan_host;      // the compact-key plan as uploaded (host copy owned by the session: the upload needs no synchronisation of its own)
    bool assign_from_rw = false;  // SESSION_ASSIGN over an RW table: every pass starts with the re-keying and the sort (rekey)
    bool fused_order_ready = false;  // fused_verify: a collect has seen a pass without rejected RW rows / failed ops — the sorted order, the
                                     // first-access links and the MPT root ranks stay with the session (they are to this form what the
                                     // assigned rows are to the 57-cell form) and later passes launch the evaluation kernel alone
    bool fused_verify = false;    // SESSION_ASSIGN whose rows are evaluated where they are computed (zk_state_verify_from_rw_open):
                                  // d_tally[0] = the State circuit's tally, d_tally[1] = rejected RW rows + failed assignments
    u32* rekey_status = nullptr;  // ... whose per-RW-row codes go here (the session's statuses are per op)
    u64 cpa_n_table = 0, cpa_n_rw = 0;
    u32* d_hist = nullptr;   // EVM: (group, state) bins (histogram -> cursors)
    u32* d_cursor = nullptr; // EVM: per-bin scatter cursors (cleared by every histogram pass)
    u32* d_hist2 = nullptr;  // EVM: the other histogram buffer (the two alternate between passes)
    uint16_t* d_bin16 = nullptr;  // EVM: sort bin of every pair, written by the histogram pass
    u32 evm_pass = 0;
    bool perm_ready = false; // EVM: zk_evm_open already enqueued the counting sort of the first pass
    u32 evm_tally_idx = 0;   // EVM: which of the result block's two tallies / deferred counters the current pass uses (resident sessions alternate)
    bool perm_valid = false; // EVM: the state-sorted mapping of this session's (fixed) step table exists: later passes evaluate through it
    int evm_ranges_known = 0;       // EVM: 1 once a collect has read the warm / cold lane ranges of this session's (fixed) step table ...
    bool evm_warm_empty = false, evm_cold_empty = false;  // ... empty ranges are not launched again
    u32 evm_warm_lanes = 0;         // ... and the warm launch is sized to its range (0 = not known yet: sized for every pair)
    bool deferred_pending = false;  // EVM: the last pass may have left deferred pairs (evm_finish_deferred has not looked yet)
    u32* last_status = nullptr;     // EVM: where the last pass wrote its statuses
    u32* d_group_start = nullptr;  // EVM: lane range of each kernel group inside d_perm
    u32* d_perm = nullptr;   // EVM: state-sorted lane -> pair permutation
    hipEvent_t ev_open0 = nullptr, ev_open1 = nullptr;  // EVM: ride on the first / last dispatch of zk_evm_open (zk_session_timing)
    bool side_stream = false;                           // EVM: ZK_OPT_SIDE_STREAM
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;    // EVM: fork to / join from the device's side stream (warm + cold builds)
    // EVM: EvmDyn, the two tallies and the lane ranges sit in ONE 128-byte device block (EvmResultBlock) that zk_collect reads
    // back with ONE copy into page-locked host memory (three copies into pageable memory cost three blocking round trips)
    void* d_result = nullptr;
    void* h_result = nullptr;
    u32 publish_seq = 0;        // ZK_POLL_RESULT: sequence number of the last evm_publish_kernel (the flag word behind the block)
    bool publish_enqueued = false;  // ... which is already in the stream behind the pass (evm_publish_enqueue: the batch entry enqueues it a witness ahead of the collect)
    u32* h_result_dev = nullptr;  // the device alias of h_result (queried once, at open)
    // Lazy tail (one-shot sessions): the open's scatter writes the warm / cold lane counts into the page-locked block (sequence number
    // early_seq); zk_launch enqueues the hot build only, and whoever consumes the pass first (evm_enqueue_tail) enqueues the builds whose
    // range is not empty — the host has the counts long before the hot kernel ends, so nothing waits and an empty build is never launched.
    u32 early_seq = 0;          // 0 = the open did not arm it
    bool tail_pending = false;
    u32 tail_cold_grid = 0;
    u32* tail_status = nullptr;
    hipEvent_t tail_e1 = nullptr;  // the pass's stop event (rides on the hot dispatch; moved to the last tail dispatch if there is one)
    bool stream_drained = false;  // the host has seen the stream's last kernel finish without a hipStreamSynchronize (zk_close may skip its own)
};

static const int MAX_EVENT_PAIRS = 256;

// Move a session to another stream of its device (NULL = the device's own stream).  Work already enqueued on the old
// stream is waited for first, so the passes of one session stay ordered.
static int session_rebind_stream(zk_session* s, hipStream_t st) {
    HIP_TRY(hipSetDevice(s->device));
    if (!st) st = g_own_stream[s->device];
    if (st == s->stream) return 0;
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->stream = st;
    return 0;
}

// ---------------------------------------------------------------------------------------
// per-device buffer arena
// ---------------------------------------------------------------------------------------
// hipMalloc / hipFree cost tens of microseconds each and hipFree synchronises the device; a verifier opens a session per
// witness.  Session buffers therefore come from per-device free lists of power-of-two size classes: a closed session
// returns its buffers (after its stream has drained), the next open re-uses them.  The cache is bounded
// (ZK_ARENA_MAX_BYTES, default 16 GiB per device: beyond that buffers go back to the runtime); zk_shutdown releases it.
// ZK_NO_ARENA=1 turns it off (every buffer a hipMalloc / hipFree pair, as before round 3).
#define ZK_ARENA_CLASSES 48
struct DevArena {
    std::mutex m;
    std::vector<void*> free_[ZK_ARENA_CLASSES];
    std::vector<hipEvent_t> events;
    std::vector<void*> pinned;  // ZK_PINNED_BYTES-byte blocks of page-locked host memory (result read-backs)
    size_t cached_bytes = 0;
};
static DevArena g_arena[ZK_MAX_DEVICES];
static size_t arena_limit() {
    static const size_t lim = [] { const char* e = getenv("ZK_ARENA_MAX_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)16 << 30); }();
    return lim;
}
static bool arena_off() {
    static const bool off = [] { const char* e = getenv("ZK_NO_ARENA"); return e && e[0] == '1'; }();
    return off;
}
static int arena_class(size_t bytes) {
    int c = 8;  // 256 B: hipMalloc's own granularity
    while (((size_t)1 << c) < bytes) c++;
    return c;
}
static int arena_take(int device, size_t bytes, void** p, int* cls) {
    const int c = arena_class(bytes ? bytes : 16);
    *cls = c;
    if (!arena_off() && c < ZK_ARENA_CLASSES) {
        DevArena& A = g_arena[device];
        std::lock_guard<std::mutex> lock(A.m);
        if (!A.free_[c].empty()) {
            *p = A.free_[c].back();
            A.free_[c].pop_back();
            A.cached_bytes -= (size_t)1 << c;
            return 0;
        }
    }
    HIP_TRY(hipMalloc(p, arena_off() ? (bytes ? bytes : 16) : ((size_t)1 << c)));
    return 0;
}
static void arena_give(int device, void* p, int cls) {
    if (!arena_off() && cls < ZK_ARENA_CLASSES) {
        DevArena& A = g_arena[device];
        std::lock_guard<std::mutex> lock(A.m);
        if (A.cached_bytes + ((size_t)1 << cls) <= arena_limit()) {
            A.free_[cls].push_back(p);
            A.cached_bytes += (size_t)1 << cls;
            return;
        }
    }
    (void)hipFree(p);
}
#define ZK_PINNED_BYTES 256
static bool evm_poll_enabled() {
    static const bool poll = [] { const char* e = getenv("ZK_POLL_RESULT"); return !(e && e[0] == '0'); }();  // default on (round 5: step 0.179 -> 0.174 ms)
    return poll;
}
static bool evm_lazy_tail_enabled() {  // ZK_LAZY_TAIL=0: every one-shot pass launches the warm and cold builds, empty or not (as until round 6)
    static const bool lazy = [] { const char* e = getenv("ZK_LAZY_TAIL"); return !(e && e[0] == '0'); }();
    return lazy && evm_poll_enabled();
}
static int arena_pinned(int device, void** p) {
    {
        DevArena& A = g_arena[device];
        std::lock_guard<std::mutex> lock(A.m);
        if (!A.pinned.empty()) {
            *p = A.pinned.back();
            A.pinned.pop_back();
            return 0;
        }
    }
    HIP_TRY(hipHostMalloc(p, ZK_PINNED_BYTES, hipHostMallocDefault));
    return 0;
}
static int arena_event(int device, hipEvent_t* e) {
    {
        DevArena& A = g_arena[device];
        std::lock_guard<std::mutex> lock(A.m);
        if (!A.events.empty()) {
            *e = A.events.back();
            A.events.pop_back();
            return 0;
        }
    }
    HIP_TRY(hipEventCreate(e));
    return 0;
}
static void arena_release_all() {
    for (int d = 0; d < ZK_MAX_DEVICES; d++) {
        DevArena& A = g_arena[d];
        std::lock_guard<std::mutex> lock(A.m);
        bool any = !A.events.empty() || !A.pinned.empty();
        for (int c = 0; c < ZK_ARENA_CLASSES; c++) any = any || !A.free_[c].empty();
        if (!any || hipSetDevice(d) != hipSuccess) continue;
        for (int c = 0; c < ZK_ARENA_CLASSES; c++) {
            for (void* p : A.free_[c]) (void)hipFree(p);
            A.free_[c].clear();
        }
        for (hipEvent_t e : A.events) (void)hipEventDestroy(e);
        A.events.clear();
        for (void* h : A.pinned) (void)hipHostFree(h);
        A.pinned.clear();
        A.cached_bytes = 0;
    }
}

static int dev_alloc(zk_session* s, void** p, size_t bytes) {
    int cls = 0;
    int rc = arena_take(s->device, bytes, p, &cls);
    if (rc) return rc;
    s->owned.push_back(*p);
    s->owned_class.push_back(cls);
    return 0;
}
// Bring a buffer to the device unless the caller already handed a device pointer.
static int stage(zk_session* s, const void* src, size_t bytes, bool device_ptrs, const void** out) {
    if (bytes == 0 || !src) {
        // empty table: "row 0" is always readable on the device — the device's shared block of zeros (read-only)
        *out = g_zero_row[s->device];
        return 0;
    }
    if (device_ptrs) {
        *out = src;
        return 0;
    }
    void* d = nullptr;
    int rc = dev_alloc(s, &d, bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s->stream));
    *out = d;
    return 0;
}

// Small device -> host read on the SESSION's stream.  (A plain hipMemcpy runs on the legacy default stream and waits for every
// other stream's work first: with several sessions being opened concurrently — block.py's chains — each open then queued behind
// the longest chain's kernels, 0.6 - 0.9 ms per open.)  Device-pointer inputs must be complete, or ordered before this stream.
static int d2h_now(hipStream_t st, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}
// Host copies of small device inputs that a caller already holds (zk_block_verify reads the block's randomness cells and offsets once
// and opens a dozen sessions over them): an open that needs such a value on the host takes it from here instead of waiting for its
// stream — which, once a chain has work queued, is a wait for that work.
struct HostView { const void* dev; const void* host; size_t bytes; };
static thread_local HostView t_host_view[4];
static thread_local int t_n_host_view = 0;
static int fetch_small(hipStream_t st, void* dst, const void* src, size_t bytes) {
    for (int k = 0; k < t_n_host_view; k++)
        if (t_host_view[k].dev == src && t_host_view[k].bytes >= bytes) { memcpy(dst, t_host_view[k].host, bytes); return 0; }
    return d2h_now(st, dst, src, bytes);
}

template <u64 (*HASH)(const ZkTable&, u32)>
static int build_index(zk_session* s, ZkTable& t, bool mpt_fingerprints = false, const ZkRwMeta* unless_dense = nullptr) {
    u32 cap = 16;
    while (cap < 2 * t.n + 2) cap <<= 1;
    u32* slots = nullptr;
    int rc = dev_alloc(s, (void**)&slots, (size_t)cap * sizeof(u32));
    if (rc) return rc;
    t.mask = cap - 1;
    t.slots = slots;
    if (unless_dense) hipLaunchKernelGGL(slots_fill_unless_dense_kernel, dim3((cap + 255) / 256), dim3(256), 0, s->stream, slots, cap, unless_dense);
    else hipLaunchKernelGGL(slots_fill_kernel, dim3((cap + 255) / 256), dim3(256), 0, s->stream, slots, cap);
    if (t.n && mpt_fingerprints) hipLaunchKernelGGL(mpt_index_build_kernel, dim3((t.n + 255) / 256), dim3(256), 0, s->stream, t, slots);
    else if (t.n) hipLaunchKernelGGL(HIP_KERNEL_NAME(index_build_kernel<HASH>), dim3((t.n + 255) / 256), dim3(256), 0, s->stream, t, slots, unless_dense);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int session_common_init(zk_session* s) {
    int rc = dev_alloc(s, (void**)&s->d_tally, 2 * sizeof(ZkTally));
    if (rc) return rc;
    hipLaunchKernelGGL(tally_reset_kernel, dim3(1), dim3(2), 0, s->stream, s->d_tally, (u32*)nullptr);
    s->tally_last = s->d_tally;
    rc = dev_alloc(s, (void**)&s->d_status, (size_t)s->n * sizeof(u32));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(s->d_status, 0, (size_t)s->n * sizeof(u32), s->stream));
    return 0;
}

extern "C" int zk_close(zk_session* s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    // nothing enqueued may still read the buffers that go back to the arena (skipped when zk_collect has already seen the stream's
    // last kernel finish through the polled result block and nothing was enqueued since)
    if (!s->stream_drained) (void)hipStreamSynchronize(s->stream);
    for (size_t k = 0; k < s->owned.size(); k++) arena_give(s->device, s->owned[k], s->owned_class[k]);
    {
        DevArena& A = g_arena[s->device];
        std::lock_guard<std::mutex> lock(A.m);
        for (hipEvent_t e : s->ev) A.events.push_back(e);
        if (s->ev_open0) A.events.push_back(s->ev_open0);
        if (s->ev_open1) A.events.push_back(s->ev_open1);
        if (s->ev_fork) A.events.push_back(s->ev_fork);
        if (s->ev_join) A.events.push_back(s->ev_join);
        if (s->h_result) A.pinned.push_back(s->h_result);
    }
    delete s;
    return 0;
}

extern "C" int zk_state_open(const uint64_t* rows, const uint32_t* flags, uint64_t n, const uint64_t* mpt,
                             uint64_t n_mpt, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_state_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rows && n > 0 && n < (1ull << 32), "zk_state_open: bad arguments");
    ARG_TRY(n_mpt < (1ull << 31), "zk_state_open: MPT table too large");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    const bool compact = opts & ZK_OPT_STATE_COMPACT;
    zk_session* s = new zk_session();
    s->kind = SESSION_STATE;
    s->n = n;
    int rc = 0;
    const void* p = nullptr;
    if ((rc = stage(s, rows, (size_t)n * (compact ? 15 : ST_NCELLS) * 32, dev, &p))) goto fail;
    s->state.rows.cells = (const u64*)p;
    s->state.rows.skip = compact ? 42u : 0u;
    if ((rc = stage(s, flags, (size_t)n * 4, dev, &p))) goto fail;
    s->state.rows.flags = (const u32*)p;
    s->state.rows.n = n;
    s->state.eval_lo = 0;
    s->state.eval_hi = n;
    if ((rc = stage(s, mpt, (size_t)n_mpt * MPT_NCELLS * 32, dev, &p))) goto fail;
    s->state.mpt.cells = (const u64*)p;
    s->state.mpt.flags = nullptr;
    s->state.mpt.n = (u32)n_mpt;
    s->state.mpt.ncells = MPT_NCELLS;
    if ((rc = build_index<state_mpt_key_hash>(s, s->state.mpt, /*mpt_fingerprints=*/true))) goto fail;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}

static int table_stage(zk_session* s, ZkTable& t, const uint64_t* cells, const uint32_t* flags, uint64_t n,
                       u32 ncells, bool dev) {
    const void* p = nullptr;
    int rc;
    if ((rc = stage(s, cells, (size_t)n * ncells * 32, dev, &p))) return rc;
    t.cells = (const u64*)p;
    if ((rc = stage(s, flags, (size_t)n * 4, dev, &p))) return rc;
    t.flags = (n && flags) ? (const u32*)p : nullptr;
    t.n = (u32)n;
    t.ncells = ncells;
    return 0;
}

// (Re)build the state-sorted permutation of the step pairs.
#ifndef ZK_HIST_BLOCK
#define ZK_HIST_BLOCK 1024  // threads per block of the histogram pass (256 was measured: 4x the global atomics, whole pass 0.124 -> 0.137 ms)
#endif
static int evm_build_perm(zk_session* s) {
    const u32 n = s->evm.n_pairs;
    // the histogram buffer of this pass is zero on entry: cleared at open, then by the previous pass's scatter
    u32* h_cur = (s->evm_pass & 1u) ? s->d_hist2 : s->d_hist;
    u32* h_next = (s->evm_pass & 1u) ? s->d_hist : s->d_hist2;
    s->evm_pass++;
    hipLaunchKernelGGL(evm_state_hist_kernel, dim3((n + ZK_HIST_BLOCK - 1) / ZK_HIST_BLOCK), dim3(ZK_HIST_BLOCK), 0, s->stream, s->evm.steps, n, h_cur, s->d_cursor, s->d_bin16, s->d_tally, s->evm.defer_count);
    hipLaunchKernelGGL(evm_state_scatter_kernel, dim3((n + 1023) / 1024), dim3(1024), 0, s->stream, s->d_bin16, n, h_cur, h_next, s->d_cursor,
                       s->d_group_start, s->d_perm);
    HIP_TRY(hipGetLastError());
    return 0;
}

// slots of an open-addressing index over n rows
static inline u32 index_cap(u64 n) {
    u32 cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    return cap;
}

// Open an EVM session.  With device-resident inputs (ZK_OPT_DEVICE_PTRS) nothing is read back and the host never waits:
// buffers come from the arena, every index / verdict is built by kernels enqueued on the session's stream —
//   memset 0xFF (all slot tables, one region)  ·  memset 0 (EvmDyn, histograms, per-pair status: one region)
//   evm_open_tables_kernel   small-table indices + EndBlock aggregates + tally reset
//   rw_prepare_kernel        density verdict + packed key records, one read of the RW key cells
//   rw_generic_index_kernel  (does nothing when the rows are dense)
//   dirb_* x 5               bytecode directory
// and the evaluation kernels pick the verdicts up from EvmDyn (evm_args_resolve).
extern "C" int zk_evm_open(const zk_evm_tables* t, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_evm_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(t && out && t->steps && t->n_steps >= 2 && t->n_steps < (1ull << 32), "zk_evm_open: bad arguments");
    ARG_TRY(t->n_rw < (1ull << 31) && t->n_bytecode < (1ull << 31) && t->n_tx < (1ull << 31) && t->n_block < (1ull << 31) &&
            t->n_copy < (1ull << 31) && t->n_keccak < (1ull << 31) && t->n_exp < (1ull << 31) && t->n_withdrawals < (1ull << 31) &&
            t->n_sig < (1ull << 31) && t->n_ecc < (1ull << 31),
            "zk_evm_open: table too large");
    ARG_TRY(t->aux_cells == 0 || (t->aux_cells >= 2 && t->aux_cells <= 64), "zk_evm_open: aux_cells must be 0 (= 2) or 2..64");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    const bool generic = opts & ZK_OPT_GENERIC_INDEX;
    zk_session* s = new zk_session();
    s->kind = SESSION_EVM;
    s->n = t->n_steps - 1;
    s->side_stream = (opts & ZK_OPT_SIDE_STREAM) != 0;
    int rc = 0;
    const void* p = nullptr;
    EvmArgs& E = s->evm;
    if ((rc = stage(s, t->steps, (size_t)t->n_steps * STEP_NCELLS * 32, dev, &p))) goto fail;
    E.steps = (const u64*)p;
    E.n_steps = t->n_steps;
    E.n_pairs = (u32)(t->n_steps - 1);
    if ((rc = table_stage(s, E.rw, t->rw, t->rw_flags, t->n_rw, RW_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.bytecode, t->bytecode, nullptr, t->n_bytecode, BYTECODE_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.tx, t->tx, t->tx_flags, t->n_tx, TX_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.block, t->block, t->block_flags, t->n_block, BLOCK_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.copy, t->copy, nullptr, t->n_copy, COPY_T_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.keccak, t->keccak, nullptr, t->n_keccak, KECCAK_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.exp, t->exp, nullptr, t->n_exp, EXP_T_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.sig, t->sig, nullptr, t->n_sig, SIG_T_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.ecc, t->ecc, nullptr, t->n_ecc, ECC_T_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, E.withdrawals, t->withdrawals, nullptr, t->n_withdrawals, 4, dev))) goto fail;
    E.withdrawals.slots = nullptr;
    E.withdrawals.mask = 0;
    E.aux = nullptr;
    E.aux_kind = nullptr;
    E.aux_cells = t->aux_cells ? t->aux_cells : 2u;
    if (t->aux && t->aux_kind) {
        if ((rc = stage(s, t->aux, (size_t)t->n_steps * E.aux_cells * 32, dev, &p))) goto fail;
        E.aux = (const u64*)p;
        if ((rc = stage(s, t->aux_kind, (size_t)t->n_steps * 4, dev, &p))) goto fail;
        E.aux_kind = (const u32*)p;
    }
    {
        ZkTable* const small[EVM_OPEN_TABLES] = {&E.tx, &E.block, &E.bytecode, &E.copy, &E.keccak, &E.exp, &E.sig, &E.ecc};
        const bool want_dir = !generic && t->n_bytecode != 0;
        // ---- one 0xFF region (u32 units, every piece a multiple of 4 words): all slot tables + the directory's `first` -------
        size_t n_ff = 0;
        u32 caps[EVM_OPEN_TABLES];
        for (int k = 0; k < EVM_OPEN_TABLES; k++) { caps[k] = index_cap(small[k]->n); n_ff += caps[k]; }
        const u32 cap_rw = index_cap(E.rw.n), cap_big = want_dir ? index_cap(t->n_bytecode) : 0u;
        n_ff += (size_t)cap_rw + 2 * (size_t)cap_big + (want_dir ? DIRB_SMALL_SLOTS : 0u);
        u32* ff = nullptr;
        if ((rc = dev_alloc(s, (void**)&ff, n_ff * sizeof(u32)))) goto fail;
        u32* cur = ff;
        for (int k = 0; k < EVM_OPEN_TABLES; k++) { small[k]->slots = cur; small[k]->mask = caps[k] - 1; cur += caps[k]; }
        E.rw.slots = cur; E.rw.mask = cap_rw - 1; cur += cap_rw;
        u32* const dir_rep = cur; cur += cap_big;
        u32* const dir_first = cur; cur += cap_big;
        u32* const small_slots = cur;
        // ---- one zero region: EvmDyn, the two histograms, the per-pair status, the directory's last / runs / bad -------------
        const size_t dyn_bytes = 256, hist_bytes = EVM_N_BINS * sizeof(u32), status_bytes = (((size_t)s->n * sizeof(u32)) + 15) & ~(size_t)15;
        const size_t zero_bytes = dyn_bytes + 2 * hist_bytes + status_bytes + 3 * (size_t)cap_big * 4;
        char* zero = nullptr;
        if ((rc = dev_alloc(s, (void**)&zero, zero_bytes))) goto fail;
        EvmResultBlock* const rb = (EvmResultBlock*)zero;  // the first 128 of the region's 256 leading bytes
        EvmDyn* dyn = &rb->dyn;
        s->d_result = rb;
        if ((rc = arena_pinned(s->device, &s->h_result))) goto fail;
        {
            void* hd = nullptr;
            if (hipHostGetDevicePointer(&hd, s->h_result, 0) == hipSuccess) s->h_result_dev = (u32*)hd;
            (void)hipGetLastError();
        }
        static_assert(sizeof(EvmResultBlock) <= 256, "the result block outgrew its slot");
        static_assert((EVM_N_BINS * sizeof(u32)) % 16 == 0, "zero region pieces are 16-byte multiples");
        s->d_hist = (u32*)(zero + dyn_bytes);
        s->d_hist2 = (u32*)(zero + dyn_bytes + hist_bytes);
        s->d_status = (u32*)(zero + dyn_bytes + 2 * hist_bytes);
        u32* const dir_last = (u32*)(zero + dyn_bytes + 2 * hist_bytes + status_bytes);
        E.dyn = dyn;
        E.defer_count = &dyn->n_deferred;
        if ((rc = dev_alloc(s, (void**)&E.defer_list, (size_t)E.n_pairs * sizeof(u32)))) goto fail;
        s->d_tally = rb->tally;
        s->tally_last = s->d_tally;
        {
            const u64 n16 = n_ff / 4 + zero_bytes / 16;
            const u32 fill_grid = (u32)(n16 / 256 < 2048 ? (n16 + 255) / 256 : 2048);
            // the open's device span is measured by two events that ride on its first and last dispatch (no event packets of their
            // own between the kernels): zk_session_timing / zk_last_timing
            if (arena_event(s->device, &s->ev_open0) || arena_event(s->device, &s->ev_open1)) s->ev_open0 = s->ev_open1 = nullptr;
            hipExtLaunchKernelGGL(evm_open_fill_kernel, dim3(fill_grid ? fill_grid : 1u), dim3(256), 0, s->stream, s->ev_open0, nullptr, 0, (uint4*)ff,
                                  (u64)(n_ff / 4), (uint4*)zero, (u64)(zero_bytes / 16));
        }
        E.rw_dense = 0;
        E.rw_base = 0;
        E.rw_keys = nullptr;
        E.codes.n = 0;
        E.codes.mask = 0;
        E.codes.packed = nullptr;
        E.codes.entries = nullptr;
        E.codes.slots = nullptr;
        E.agg_max_txs = E.agg_total_txs = E.agg_invalid_txs = E.agg_bad_invalid_rows = E.agg_total_wds = 0;
        EvmOpenTables o;
        memset(&o, 0, sizeof o);
        u32 blk = 0;
        for (int k = 0; k < EVM_OPEN_TABLES; k++) { o.t[k] = *small[k]; o.block_start[k] = blk; blk += (small[k]->n + 255u) / 256u; }
        o.block_start[EVM_OPEN_TABLES] = blk;
        blk += (u32)((t->n_withdrawals + 255) / 256);
        o.block_start[EVM_OPEN_TABLES + 1] = blk;
        o.wds = E.withdrawals.cells;
        o.n_wds = (u32)t->n_withdrawals;
        o.dyn = dyn;
        o.tally = s->d_tally;
        o.rw = E.rw;
        if (t->n_rw && !generic) {
            u64* d_keys = nullptr;
            if ((rc = dev_alloc(s, (void**)&d_keys, (size_t)t->n_rw * 32))) goto fail;
            E.rw_keys = d_keys;
            o.rw_keys = d_keys;
        }
        const u32 rw_row_blocks = o.rw_keys ? (u32)((t->n_rw + 63) / 64) : 0u;  // four lanes per row
        o.dir_block0 = blk;
        u32 dir_row_blocks = 0;
        if (want_dir) {  // bytecode directory (code_dir_build.hpp)
            DirBuild& d = o.dir;
            d.rows = E.bytecode.cells;
            d.n = (u32)t->n_bytecode;
            d.rep = dir_rep;
            d.first = dir_first;
            d.last = dir_last;
            d.runs = dir_last + cap_big;
            d.bad = dir_last + 2 * (size_t)cap_big;
            d.big_mask = cap_big - 1;
            d.small_slots = small_slots;
            d.dyn = dyn;
            if ((rc = dev_alloc(s, (void**)&d.packed, (size_t)d.n * sizeof(uint16_t)))) goto fail;
            if ((rc = dev_alloc(s, (void**)&d.entries, (size_t)DIRB_MAX_ENTRIES * sizeof(ZkCodeEntry)))) goto fail;
            if ((rc = dev_alloc(s, (void**)&d.list, (size_t)DIRB_MAX_ENTRIES * 4))) goto fail;
            dir_row_blocks = (d.n + 255u) / 256u;
            E.codes.entries = d.entries;
            E.codes.slots = d.small_slots;
            E.codes.packed = d.packed;
        }
        // the first pass's counting sort rides on the same two launches (it only needs the step rows): zk_launch then goes
        // straight to the evaluation kernels
        if ((rc = dev_alloc(s, (void**)&s->d_cursor, EVM_N_BINS * sizeof(u32)))) goto fail;
        if ((rc = dev_alloc(s, (void**)&s->d_bin16, (size_t)E.n_pairs * sizeof(uint16_t)))) goto fail;
        s->d_group_start = rb->group_start;
        if ((rc = dev_alloc(s, (void**)&s->d_perm, ((size_t)E.n_pairs + EVM_PERM_PAD + 2 * EVM_HOT_BLOCK) * sizeof(u32)))) goto fail;  // + slack: the hot kernel reads perm[t] for every lane of its grid
        const bool sorted = !(opts & ZK_OPT_NO_STATE_SORT);
        o.hist_block0 = o.dir_block0 + dir_row_blocks;
        u32 hist_blocks = 0, scatter_blocks = 0;
        if (sorted) {
            o.sort.steps = E.steps; o.sort.n_pairs = E.n_pairs; o.sort.hist = s->d_hist; o.sort.hist_next = s->d_hist2;
            o.sort.taken = s->d_cursor; o.sort.bin16 = s->d_bin16; o.sort.group_start = s->d_group_start; o.sort.perm = s->d_perm;
            o.sort.tally = s->d_tally;
            o.sort.defer_count = E.defer_count;
            if ((opts & ZK_OPT_SINGLE_PASS) && s->h_result_dev && evm_lazy_tail_enabled() && !(opts & ZK_OPT_SIDE_STREAM)) {
                static std::atomic<u32> g_early_seq{0};
                u32 q = ++g_early_seq;
                if (!q) q = ++g_early_seq;  // never 0
                __atomic_store_n(&((u32*)s->h_result)[EVM_EARLY_WORD + 2], 0u, __ATOMIC_RELEASE);
                s->early_seq = q;
                o.sort.early_host = s->h_result_dev;
                o.sort.early_seq = q;
            }
            hist_blocks = (E.n_pairs + 255u) / 256u;
            scatter_blocks = (E.n_pairs + EVM_OPEN_P2_BLOCK - 1u) / EVM_OPEN_P2_BLOCK;
            s->evm_pass = 1;        // the next pass's histogram is d_hist2 (cleared by this scatter)
            s->perm_ready = true;
        }
        o.rw_block0 = o.hist_block0 + hist_blocks;
        // packed step records (evm_circuit.hpp "step records"): the other streaming range of the launch
        o.rec_block0 = o.rw_block0 + rw_row_blocks;
        u32 rec_blocks = 0;
        E.step_recs = nullptr;
        if (!(opts & ZK_OPT_SINGLE_PASS)) {
            u32* recs = nullptr;
            if ((rc = dev_alloc(s, (void**)&recs, (size_t)t->n_steps * EVM_REC_WORDS * sizeof(u32)))) goto fail;
            o.steps = E.steps; o.n_steps = (u32)t->n_steps; o.step_recs = recs;
            E.step_recs = recs;
            rec_blocks = (u32)((t->n_steps + 63) / 64);  // four lanes per step
        }
        const u32 grid1 = o.rec_block0 + rec_blocks;
        {   // latency-bound blocks dealt into the first 1 / ZK_P1_LAT_SPREAD of the RW range (evm_open_phase1_kernel)
            static const u32 spread = [] { const char* e = getenv("ZK_P1_LAT_SPREAD"); const int v = e ? atoi(e) : 2; return (u32)(v < 0 ? 0 : v); }();
            o.lat_period = (o.rw_block0 && spread) ? o.rec_block0 / (spread * o.rw_block0) : 0u;
        }
#ifdef ZK_DIAG_P1
        if (const char* e = getenv("ZK_DIAG_P1_SKIP")) o.diag_skip = (u32)atoi(e);
#endif
        {
            const u32 dir_blocks = want_dir ? DIRB_MAX_ENTRIES / EVM_OPEN_P2_BLOCK : 0u;
            const u32 rw_blocks = t->n_rw ? EVM_OPEN_RW_GENERIC_BLOCKS : 0u;
            bool phase2 = scatter_blocks + dir_blocks + rw_blocks != 0;
#ifdef ZK_DIAG_P1
            if (o.diag_skip) phase2 = false;  // its inputs are missing
#endif
            hipExtLaunchKernelGGL(evm_open_phase1_kernel, dim3(grid1 ? grid1 : 1u), dim3(256), 0, s->stream, nullptr, phase2 ? nullptr : s->ev_open1, 0, o);
            if (phase2)
                hipExtLaunchKernelGGL(evm_open_phase2_kernel, dim3(scatter_blocks + dir_blocks + rw_blocks), dim3(EVM_OPEN_P2_BLOCK), 0, s->stream, nullptr,
                                      s->ev_open1, 0, o, scatter_blocks, dir_blocks, generic ? 1u : 0u);
        }
        if (hipGetLastError() != hipSuccess) { rc = -2; g_err = "zk_evm_open: a build kernel failed to launch"; goto fail; }
    }
    E.opts = (t->begin_with_first_step ? 1u : 0u) | (t->end_with_last_step ? 2u : 0u);
    E.prof = nullptr;
    if (getenv("ZK_EVM_PROF")) {
        if ((rc = dev_alloc(s, (void**)&E.prof, 1024 * 4 * 8 * sizeof(unsigned long long)))) goto fail;
        if (hipMemsetAsync(E.prof, 0, 1024 * 4 * 8 * sizeof(unsigned long long), s->stream) != hipSuccess) { rc = -2; goto fail; }
    }
    E.perm = (opts & ZK_OPT_NO_STATE_SORT) ? nullptr : s->d_perm;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}

static void session_timing(zk_session* s, double pass_ms, double* open_ms, double* span_ms);
static int evm_enqueue_tail(zk_session* s);
static bool evm_publish_enqueue(zk_session* s);
extern "C" int zk_evm_verify(const zk_evm_tables* t, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_evm_verify: result is null");
    zk_session* s = nullptr;
    const auto h0 = std::chrono::steady_clock::now();
    int rc = zk_evm_open(t, opts | ZK_OPT_SINGLE_PASS, &s);  // one pass: the step records would not pay for themselves
    if (rc) return rc;
    const auto h1 = std::chrono::steady_clock::now();
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    const auto h2 = std::chrono::steady_clock::now();
    if (!rc) rc = zk_collect(s, result);
    const auto h3 = std::chrono::steady_clock::now();
    if (!rc) {
        t_timing[1] = result->kernel_ms;
        session_timing(s, result->kernel_ms, &t_timing[0], &t_timing[2]);
        for (int k = 0; k < 3; k++) t_timing_sum[k] += t_timing[k];
        t_timing_count++;
    }
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    const auto h4 = std::chrono::steady_clock::now();
    t_host_phase[0] = std::chrono::duration<double, std::micro>(h1 - h0).count();
    t_host_phase[1] = std::chrono::duration<double, std::micro>(h2 - h1).count();
    t_host_phase[2] = std::chrono::duration<double, std::micro>(h3 - h2).count();
    t_host_phase[3] = std::chrono::duration<double, std::micro>(h4 - h3).count();
    return rc;
}

// Batch verification: n independent witnesses, each opened, evaluated and collected exactly as zk_evm_verify does, two in flight —
// witness i + 1 is opened and launched on the device's other pipeline stream before the host waits for witness i's tally, so the
// HBM-bound open of one (key-record / index builds streaming the RW table) runs under the latency-bound evaluation kernel of the
// other, and the host's launch and synchronisation latencies are hidden.  Results are written in input order; per-pair statuses
// are not returned (zk_evm_verify on the failing witness gives them).
extern "C" int zk_evm_verify_batch(const zk_evm_tables* const* t, uint64_t n, uint32_t opts, zk_result* results) {
    ARG_TRY(t_device >= 0, "zk_evm_verify_batch: call zk_init first");
    ARG_TRY((t && results) || n == 0, "zk_evm_verify_batch: bad arguments");
    // every witness pointer is checked before the first session opens: an early return from inside the pipeline would leave the
    // other slot's session open with kernels in flight (arena buffers, events and the pinned block leaked)
    for (uint64_t i = 0; i < n; i++) ARG_TRY(t[i], "zk_evm_verify_batch: null witness");
    HIP_TRY(hipSetDevice(t_device));
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        for (int k = 0; k < 2; k++)
            if (!g_batch_stream[t_device][k]) HIP_TRY(hipStreamCreateWithFlags(&g_batch_stream[t_device][k], hipStreamNonBlocking));
    }
    // the caller's stream orders its own uploads before this call: the pipeline streams start behind it
    const hipStream_t caller = t_stream;
    HIP_TRY(hipStreamSynchronize(caller));
    zk_session* pend[2] = {nullptr, nullptr};
    int rc = 0;
    for (uint64_t i = 0; i < n + 2 && !rc; i++) {
        const int slot = (int)(i & 1u);
        if (pend[slot]) {  // witness i - 2
            rc = zk_collect(pend[slot], &results[i - 2]);
            zk_close(pend[slot]);
            pend[slot] = nullptr;
        }
        if (!rc && i < n) {
            t_stream = g_batch_stream[t_device][slot];
            rc = zk_evm_open(t[i], opts | ZK_OPT_SINGLE_PASS, &pend[slot]);
            t_stream = caller;
            if (!rc) rc = zk_launch(pend[slot], nullptr);
        }
        // the witness launched one iteration ago (the other slot): the open's sort has long told the host its lane ranges — enqueue what is
        // left of its pass and its publish kernel now, behind its hot kernel, instead of when the next iteration comes to collect it
        zk_session* const o = pend[slot ^ 1];
        if (!rc && o && !o->publish_enqueued) {
            rc = evm_enqueue_tail(o);
            if (!rc) o->publish_enqueued = evm_publish_enqueue(o);
        }
    }
    for (int k = 0; k < 2; k++)
        if (pend[k]) zk_close(pend[k]);
    return rc;
}

extern "C" int zk_bytecode_open(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak,
                                const uint64_t* randomness, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_bytecode_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rows && randomness && n > 0 && n < (1ull << 32) && n_keccak < (1ull << 31), "zk_bytecode_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_BYTECODE;
    s->n = n;
    int rc = 0;
    const void* p = nullptr;
    u64 rh[4];
    if ((rc = stage(s, rows, (size_t)n * BC_NCELLS * 32, dev, &p))) goto fail;
    s->bytecode.rows.cells = (const u64*)p;
    s->bytecode.rows.flags = nullptr;
    s->bytecode.rows.n = n;
    if ((rc = table_stage(s, s->bytecode.keccak, keccak, nullptr, n_keccak, KECCAK_NCELLS, dev))) goto fail;
    if ((rc = build_index<keccak_key_hash>(s, s->bytecode.keccak))) goto fail;
    if (dev) {
        if (fetch_small(s->stream, rh, randomness, 32)) { rc = -2; g_err = "randomness download failed"; goto fail; }
    } else {
        memcpy(rh, randomness, 32);
    }
    for (int k = 0; k < 4; k++) { s->bytecode.r.v[2 * k] = (u32)rh[k]; s->bytecode.r.v[2 * k + 1] = (u32)(rh[k] >> 32); }
    {
        u64* d_rm = nullptr;
        if ((rc = dev_alloc(s, (void**)&d_rm, 32))) goto fail;
        zk_launch_fr_to_mont(s->stream, s->bytecode.r, d_rm);
        s->bytecode.r_mont = d_rm;
    }
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}

extern "C" int zk_exp_open(const uint64_t* rows, uint64_t n, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_exp_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rows && n > 0 && n < (1ull << 32), "zk_exp_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_EXP;
    s->n = n;
    int rc = 0;
    const void* p = nullptr;
    if ((rc = stage(s, rows, (size_t)n * EX_NCELLS * 32, dev, &p))) goto fail;
    s->exp.rows.cells = (const u64*)p;
    s->exp.rows.flags = nullptr;
    s->exp.rows.n = n;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}

// dense RW-index metadata (see ZkRwMeta), verified on the device
static int build_rw_meta(zk_session* s, const ZkTable& rw, const ZkRwMeta** out) {
    ZkRwMeta* d_meta = nullptr;
    int rc = dev_alloc(s, (void**)&d_meta, sizeof(ZkRwMeta));
    if (rc) return rc;
    static const ZkRwMeta init_dense = {1u, 0u, 0ull}, init_empty = {0u, 0u, 0ull};  // static storage: no wait for the copy
    HIP_TRY(hipMemcpyAsync(d_meta, rw.n ? &init_dense : &init_empty, sizeof(ZkRwMeta), hipMemcpyHostToDevice, s->stream));
    if (rw.n) hipLaunchKernelGGL(rw_dense_check_kernel, dim3((rw.n + 255) / 256), dim3(256), 0, s->stream, rw, d_meta);
    *out = d_meta;
    return 0;
}

extern "C" int zk_copy_open(const zk_copy_tables* t, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_copy_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(t && out && t->rows && t->randomness && t->n_rows > 0 && t->n_rows < (1ull << 32), "zk_copy_open: bad arguments");
    ARG_TRY(t->n_rw < (1ull << 31) && t->n_bytecode < (1ull << 31) && t->n_tx < (1ull << 31), "zk_copy_open: table too large");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_COPY;
    s->n = t->n_rows;
    int rc = 0;
    const void* p = nullptr;
    u64 rh[4];
    if ((rc = stage(s, t->rows, (size_t)t->n_rows * CP_NCELLS * 32, dev, &p))) goto fail;
    s->copy.rows.cells = (const u64*)p;
    if ((rc = stage(s, t->row_flags, (size_t)t->n_rows * 4, dev, &p))) goto fail;
    s->copy.rows.flags = t->row_flags ? (const u32*)p : nullptr;
    s->copy.rows.n = t->n_rows;
    if ((rc = table_stage(s, s->copy.rw, t->rw, t->rw_flags, t->n_rw, RW_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, s->copy.bytecode, t->bytecode, nullptr, t->n_bytecode, BYTECODE_NCELLS, dev))) goto fail;
    if ((rc = table_stage(s, s->copy.tx, t->tx, t->tx_flags, t->n_tx, TX_NCELLS, dev))) goto fail;
    s->copy.rw_meta = nullptr;
    if (!(opts & ZK_OPT_GENERIC_INDEX) && (rc = build_rw_meta(s, s->copy.rw, &s->copy.rw_meta))) goto fail;
    if ((rc = build_index<rw_key_hash>(s, s->copy.rw, false, s->copy.rw_meta))) goto fail;  // (neither filled nor built when the rows are dense)
    if ((rc = build_index<bc_key_hash>(s, s->copy.bytecode))) goto fail;
    if ((rc = build_index<tx_key_hash>(s, s->copy.tx))) goto fail;
    if (dev) {
        if (fetch_small(s->stream, rh, t->randomness, 32)) { rc = -2; g_err = "randomness download failed"; goto fail; }
    } else {
        memcpy(rh, t->randomness, 32);
    }
    for (int k = 0; k < 4; k++) { s->copy.r.v[2 * k] = (u32)rh[k]; s->copy.r.v[2 * k + 1] = (u32)(rh[k] >> 32); }
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}

static int one_shot(zk_session* s, bool dev, uint32_t* status_out, zk_result* result);
extern "C" int zk_sign_open(const zk_sign_units* t, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_sign_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(t && out && t->bytes && t->cells && t->meta && t->randomness && t->n_units > 0 && t->n_units < (1ull << 32),
            "zk_sign_open: bad arguments");
    ARG_TRY(t->n_keccak < (1ull << 31) && t->n_tx_rows < (1ull << 32), "zk_sign_open: table too large");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_SIGN;
    s->n = t->n_units;
    int rc = 0;
    const void* p = nullptr;
    u64 rh[4];
    if ((rc = stage(s, t->bytes, (size_t)t->n_units * SG_NBYTES_ROWS * 32, dev, &p))) goto fail;
    s->sign.bytes = (const uint8_t*)p;
    if ((rc = stage(s, t->cells, (size_t)t->n_units * SG_NCELLS * 32, dev, &p))) goto fail;
    s->sign.cells.cells = (const u64*)p;
    s->sign.cells.flags = nullptr;
    s->sign.cells.n = t->n_units;
    if ((rc = stage(s, t->meta, (size_t)t->n_units * 16, dev, &p))) goto fail;
    s->sign.meta = (const u32*)p;
    if ((rc = table_stage(s, s->sign.keccak, t->keccak, nullptr, t->n_keccak, KECCAK_NCELLS, dev))) goto fail;
    if ((rc = build_index<keccak_key_hash>(s, s->sign.keccak))) goto fail;
    if ((rc = table_stage(s, s->sign.tx_rows, t->tx_rows, t->tx_flags, t->n_tx_rows, TX_NCELLS, dev))) goto fail;
    s->sign.tx_rows.slots = nullptr;
    s->sign.tx_rows.mask = 0;
    s->sign.is_sig = t->is_sig ? 1u : 0u;
    if (dev) {
        if (fetch_small(s->stream, rh, t->randomness, 32)) { rc = -2; g_err = "randomness download failed"; goto fail; }
    } else {
        memcpy(rh, t->randomness, 32);
    }
    for (int k = 0; k < 4; k++) { s->sign.r.v[2 * k] = (u32)rh[k]; s->sign.r.v[2 * k + 1] = (u32)(rh[k] >> 32); }
    {
        u64* d_rpow = nullptr;
        if ((rc = dev_alloc(s, (void**)&d_rpow, 64 * 4 * sizeof(u64)))) goto fail;
        zk_launch_sign_rpow(s->stream, s->sign.r, d_rpow);
        s->sign.rpow = d_rpow;
    }
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_sign_verify(const zk_sign_units* t, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_sign_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_sign_open(t, opts, &s);
    if (rc) return rc;
    return one_shot(s, opts & ZK_OPT_DEVICE_PTRS, status_out, result);
}

// ---- Keccak table generation
extern "C" int zk_keccak_open(const uint8_t* data, uint64_t n_bytes, const uint64_t* offsets, uint64_t n_msgs,
                              const uint64_t* randomness, uint32_t mode, uint64_t* rows_dev, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_keccak_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && offsets && randomness && n_msgs > 0 && n_msgs < (1ull << 32) && (data || n_bytes == 0) && mode <= 1u,
            "zk_keccak_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    ARG_TRY(dev || !rows_dev, "zk_keccak_open: rows_dev needs ZK_OPT_DEVICE_PTRS");
    if (!dev) {
        for (u64 k = 0; k < n_msgs; k++) ARG_TRY(offsets[k] <= offsets[k + 1], "zk_keccak_open: offsets must be non-decreasing");
        ARG_TRY(offsets[n_msgs] <= n_bytes, "zk_keccak_open: offsets exceed the data buffer");
    }
    zk_session* s = new zk_session();
    s->kind = SESSION_KECCAK;
    s->n = n_msgs;
    int rc = 0;
    const void* p = nullptr;
    u64 rh[4];
    Fr r;
    if ((rc = stage(s, data, (size_t)n_bytes, dev, &p))) goto fail;
    s->keccak_gen.data = (const uint8_t*)p;
    if ((rc = stage(s, offsets, (size_t)(n_msgs + 1) * 8, dev, &p))) goto fail;
    s->keccak_gen.offsets = (const u64*)p;
    s->keccak_gen.n = n_msgs;
    s->keccak_gen.mode = mode;
    s->keccak_gen.long_list = nullptr;
    s->keccak_gen.long_count = nullptr;
    if (mode == KT_MODE_CIRCUIT && !getenv("ZK_KECCAK_NO_GROUPS")) {  // the lane-group kernel's work list (keccak_table.hpp)
        u32* d_list = nullptr;
        if ((rc = dev_alloc(s, (void**)&d_list, ((size_t)n_msgs + 1) * sizeof(u32)))) goto fail;
        s->keccak_gen.long_list = d_list + 1;
        s->keccak_gen.long_count = d_list;
    }
    if (dev) {
        if (fetch_small(s->stream, rh, randomness, 32)) { rc = -2; g_err = "randomness download failed"; goto fail; }
    } else {
        memcpy(rh, randomness, 32);
    }
    for (int k = 0; k < 4; k++) { r.v[2 * k] = (u32)rh[k]; r.v[2 * k + 1] = (u32)(rh[k] >> 32); }
    {
        u64* d_rpow = nullptr;
        if ((rc = dev_alloc(s, (void**)&d_rpow, KT_RPOW_ROWS * 4 * sizeof(u64)))) goto fail;
        zk_launch_keccak_rpow(s->stream, r, d_rpow);
        s->keccak_gen.rpow = d_rpow;
    }
    if (rows_dev) {
        s->keccak_gen.rows = rows_dev;
    } else {
        u64* d_rows = nullptr;
        if ((rc = dev_alloc(s, (void**)&d_rows, (size_t)n_msgs * KT_NCELLS * 32))) goto fail;
        s->keccak_gen.rows = d_rows;
    }
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_keccak_read_rows(zk_session* s, uint64_t* rows_host) {
    ARG_TRY(s && rows_host && s->kind == SESSION_KECCAK, "zk_keccak_read_rows: bad arguments");
    HIP_TRY(hipMemcpyAsync(rows_host, s->keccak_gen.rows, (size_t)s->n * KT_NCELLS * 32, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}
extern "C" int zk_keccak_table(const uint8_t* data, uint64_t n_bytes, const uint64_t* offsets, uint64_t n_msgs,
                               const uint64_t* randomness, uint32_t mode, uint64_t* rows_out, uint32_t opts,
                               uint32_t* status_out, zk_result* result) {
    ARG_TRY(result && rows_out, "zk_keccak_table: null output");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = nullptr;
    int rc = zk_keccak_open(data, n_bytes, offsets, n_msgs, randomness, mode, dev ? rows_out : nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && !dev) rc = zk_keccak_read_rows(s, rows_out);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// ---- State-circuit witness assignment
extern "C" int zk_state_assign_open(const uint64_t* ops, const uint32_t* op_flags, uint64_t n, uint64_t* rows_dev,
                                    uint32_t* row_flags_dev, uint64_t* mpt_dev, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_state_assign_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && ops && op_flags && n > 0 && n < (1ull << 31), "zk_state_assign_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    ARG_TRY(dev || (!rows_dev && !row_flags_dev && !mpt_dev), "zk_state_assign_open: output buffers need ZK_OPT_DEVICE_PTRS");
    zk_session* s = new zk_session();
    s->kind = SESSION_ASSIGN;
    s->n = n;
    AssignArgs& a = s->assign;
    int rc = 0;
    const void* p = nullptr;
    u32 cap = 16;
    if ((rc = stage(s, ops, (size_t)n * ASG_NSLOTS * 32, dev, &p))) goto fail;
    a.ops = (const u64*)p;
    if ((rc = stage(s, op_flags, (size_t)n * 4, dev, &p))) goto fail;
    a.op_flags = (const u32*)p;
    a.n = n;
    a.nb = (u32)((n + ASG_BLOCK - 1) / ASG_BLOCK);
    a.rows = rows_dev;
    a.row_flags = row_flags_dev;
    a.mpt = mpt_dev;
    a.compact = (opts & ZK_OPT_STATE_COMPACT) ? 1u : 0u;
    if (!a.rows && (rc = dev_alloc(s, (void**)&a.rows, (size_t)n * (a.compact ? 15 : ASG_ROW_NCELLS) * 32))) goto fail;
    if (!a.row_flags && (rc = dev_alloc(s, (void**)&a.row_flags, (size_t)n * 4))) goto fail;
    if (!a.mpt && (rc = dev_alloc(s, (void**)&a.mpt, (size_t)n * ASG_MPT_NCELLS * 32))) goto fail;
    while (cap < 2 * n + 2) cap <<= 1;
    a.mask = cap - 1;
    if ((rc = dev_alloc(s, (void**)&a.slots, (size_t)cap * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.first, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.rank, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.blk_cnt, (size_t)(a.nb + 1) * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.blk_next, (size_t)(a.nb + 1) * 4))) goto fail;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_state_assign_read(zk_session* s, uint64_t* rows_host, uint32_t* row_flags_host, uint64_t* mpt_host,
                                    uint64_t mpt_capacity_rows, uint64_t* n_mpt_out) {
    ARG_TRY(s && s->kind == SESSION_ASSIGN, "zk_state_assign_read: bad arguments");
    const AssignArgs& a = s->assign;
    u32 n_mpt = 0;
    HIP_TRY(hipMemcpyAsync(&n_mpt, a.blk_cnt + a.nb, 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (n_mpt_out) *n_mpt_out = n_mpt;
    if (rows_host) HIP_TRY(hipMemcpyAsync(rows_host, a.rows, (size_t)a.n * (a.compact ? 15 : ASG_ROW_NCELLS) * 32, hipMemcpyDeviceToHost, s->stream));
    if (row_flags_host) HIP_TRY(hipMemcpyAsync(row_flags_host, a.row_flags, (size_t)a.n * 4, hipMemcpyDeviceToHost, s->stream));
    if (mpt_host) {
        ARG_TRY(mpt_capacity_rows >= n_mpt, "zk_state_assign_read: mpt buffer too small");
        if (n_mpt) HIP_TRY(hipMemcpyAsync(mpt_host, a.mpt, (size_t)n_mpt * ASG_MPT_NCELLS * 32, hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}
extern "C" int zk_state_assign(const uint64_t* ops, const uint32_t* op_flags, uint64_t n, uint64_t* rows_out,
                               uint32_t* row_flags_out, uint64_t* mpt_out, uint64_t* n_mpt_out, uint32_t opts,
                               uint32_t* status_out, zk_result* result) {
    ARG_TRY(result && n_mpt_out, "zk_state_assign: null output");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = nullptr;
    int rc = zk_state_assign_open(ops, op_flags, n, dev ? rows_out : nullptr, dev ? row_flags_out : nullptr,
                                  dev ? mpt_out : nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc) rc = dev ? zk_state_assign_read(s, nullptr, nullptr, nullptr, 0, n_mpt_out)
                      : zk_state_assign_read(s, rows_out, row_flags_out, mpt_out, n, n_mpt_out);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// ---- RW table -> State-circuit operations (state_rekey.hpp)
#include "state_rekey_plan.hpp"
// Stage the RW table, run the class scan, build the plan and allocate the sort's buffers (s->rekey); with want_ops the op list's
// buffers too (ops_dev / op_flags_dev: the caller's, or session-owned when null).  s->n is left to the caller.
static int rekey_setup(zk_session* s, const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, bool dev, bool want_ops,
                       uint64_t* ops_dev, uint32_t* op_flags_dev) {
    RekeyArgs& a = s->rekey;
    memset(&a, 0, sizeof(a));
    int rc = 0;
    const void* p = nullptr;
    RwkHostPlan hp;
    std::vector<u32> h_masks(2 * RWK_MASK_WORDS_H + RWK_NCLASSES);
    const size_t mask_bytes = h_masks.size() * 4;
    RwkPlan* d_plan = nullptr;
    if ((rc = stage(s, rw, (size_t)n * RWK_RW_NCELLS * 32, dev, &p))) return rc;
    a.rw = (const u64*)p;
    if (rw_flags) {
        if ((rc = stage(s, rw_flags, (size_t)n * 4, dev, &p))) return rc;
        a.rw_flags = (const u32*)p;
    }
    a.n = n;
    if ((rc = dev_alloc(s, (void**)&a.masks, mask_bytes))) return rc;
    HIP_TRY(hipMemsetAsync(a.masks, 0, mask_bytes, s->stream));
    zk_launch_rekey_scan(s->stream, a);
    HIP_TRY(hipMemcpyAsync(h_masks.data(), a.masks, mask_bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    {
        const char* e = getenv("ZK_REKEY_NO_RANKS");  // (read per open: the tests switch it inside one process)
        rwk_build_plan(h_masks.data(), !(e && e[0] == '1'), hp);
    }
    a.key_words = hp.plan.key_words;
    a.n_passes = hp.plan.n_passes;
    a.n_ops = 1 + hp.n_kept;
    a.ntiles = (u32)((n + 4095) / 4096);
    if ((rc = dev_alloc(s, (void**)&d_plan, sizeof(RwkPlan)))) return rc;
    s->rekey_plan_host = hp.plan;  // (lives as long as the session: no wait for the copy here — one host round trip less per open)
    HIP_TRY(hipMemcpyAsync(d_plan, &s->rekey_plan_host, sizeof(RwkPlan), hipMemcpyHostToDevice, s->stream));
    a.plan = d_plan;
    {
        const char* e = getenv("ZK_REKEY_NO_FAST");
        const bool no_fast = e && e[0] == '1';
        a.fast = (!no_fast && a.key_words <= 2 && a.n_passes <= 8 && n < (1ull << 30)) ? 1u : 0u;
    }
    a.ntiles_fast = (u32)((n + RWK_SW_TILE - 1) / RWK_SW_TILE);
    if ((rc = dev_alloc(s, (void**)&a.idx_a, (size_t)n * 4))) return rc;
    if ((rc = dev_alloc(s, (void**)&a.idx_b, (size_t)n * 4))) return rc;
    if (a.fast) {
        if ((rc = dev_alloc(s, (void**)&a.key64_a, (size_t)n * 8))) return rc;
        if ((rc = dev_alloc(s, (void**)&a.key64_b, (size_t)n * 8))) return rc;
        if ((rc = dev_alloc(s, (void**)&a.sweep, ((size_t)RWK_SWEEP_HEAD + (size_t)a.n_passes * a.ntiles_fast * 256) * 4))) return rc;
    } else {
        if ((rc = dev_alloc(s, (void**)&a.keys, (size_t)a.key_words * n * 4))) return rc;
        if ((rc = dev_alloc(s, (void**)&a.hist, (size_t)a.ntiles * 256 * 4))) return rc;
    }
    a.n_jobs = (u32)hp.jobs.size();
    if (a.n_jobs) {
        u32 members = 0;
        for (u32 j = 0; j < a.n_jobs; j++) { a.jobs[j] = hp.jobs[j]; members += hp.jobs[j].count; }
        for (int f = 0; f < RWK_NFIELDS; f++)
            if (hp.rank_field[f] && (rc = dev_alloc(s, (void**)&a.ranks[f], (size_t)n * 4))) return rc;
        if ((rc = dev_alloc(s, (void**)&a.job_cursor, (size_t)RWK_MAX_JOBS * 4))) return rc;
        if ((rc = dev_alloc(s, (void**)&a.job_rows, (size_t)members * 4))) return rc;
        if ((rc = dev_alloc(s, (void**)&a.job_vals, (size_t)members * 32))) return rc;
    }
    if (want_ops) {
        a.ops = ops_dev;
        a.op_flags = op_flags_dev;
        if (!a.ops && (rc = dev_alloc(s, (void**)&a.ops, (size_t)a.n_ops * RWK_NSLOTS * 32))) return rc;
        if (!a.op_flags && (rc = dev_alloc(s, (void**)&a.op_flags, (size_t)a.n_ops * 4))) return rc;
    }
    return 0;
}
// the index buffer the last radix pass of a rekey session writes: the sorted order of the kept rows (then the left-out ones)
static const u32* rekey_order(const RekeyArgs& a) { return ((a.n_passes - 1u) & 1u) ? a.idx_b : a.idx_a; }

extern "C" int zk_state_ops_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint64_t* ops_dev, uint32_t* op_flags_dev,
                                         uint32_t opts, uint64_t* n_ops_out, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_state_ops_from_rw_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rw && n > 0 && n < (1ull << 31), "zk_state_ops_from_rw_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    ARG_TRY(dev || (!ops_dev && !op_flags_dev), "zk_state_ops_from_rw_open: output buffers need ZK_OPT_DEVICE_PTRS");
    zk_session* s = new zk_session();
    s->kind = SESSION_REKEY;
    s->n = n;
    int rc = rekey_setup(s, rw, rw_flags, n, dev, true, ops_dev, op_flags_dev);
    if (!rc) rc = session_common_init(s);
    if (rc) { zk_close(s); return rc; }
    if (n_ops_out) *n_ops_out = s->rekey.n_ops;
    *out = s;
    return 0;
}
// RW table -> State-circuit rows in one session: the re-keying and the sort as above, then the witness assignment reads its ops
// straight from the RW rows through the sorted order (state_assign.hpp asg_slot_rw): the op list is never materialised.
extern "C" int zk_state_assign_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n_rw, uint64_t* rows_dev,
                                            uint32_t* row_flags_dev, uint64_t* mpt_dev, uint32_t opts, uint64_t* n_ops_out, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_state_assign_from_rw_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rw && n_rw > 0 && n_rw < (1ull << 31) - 1, "zk_state_assign_from_rw_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    ARG_TRY(dev || (!rows_dev && !row_flags_dev && !mpt_dev), "zk_state_assign_from_rw_open: output buffers need ZK_OPT_DEVICE_PTRS");
    zk_session* s = new zk_session();
    s->kind = SESSION_ASSIGN;
    int rc = rekey_setup(s, rw, rw_flags, n_rw, dev, false, nullptr, nullptr);
    if (rc) { zk_close(s); return rc; }
    const u64 n = s->rekey.n_ops;
    s->n = n;
    s->assign_from_rw = true;
    AssignArgs& a = s->assign;
    u32 cap = 16;
    a.n = n;
    a.rw = s->rekey.rw;
    a.rw_flags = s->rekey.rw_flags;
    a.order = rekey_order(s->rekey);
    a.nb = (u32)((n + ASG_BLOCK - 1) / ASG_BLOCK);
    a.rows = rows_dev;
    a.row_flags = row_flags_dev;
    a.mpt = mpt_dev;
    a.compact = (opts & ZK_OPT_STATE_COMPACT) ? 1u : 0u;
    if (!a.rows && (rc = dev_alloc(s, (void**)&a.rows, (size_t)n * (a.compact ? 15 : ASG_ROW_NCELLS) * 32))) goto fail;
    if (!a.row_flags && (rc = dev_alloc(s, (void**)&a.row_flags, (size_t)n * 4))) goto fail;
    if (!a.mpt && (rc = dev_alloc(s, (void**)&a.mpt, (size_t)n * ASG_MPT_NCELLS * 32))) goto fail;
    while (cap < 2 * n + 2) cap <<= 1;
    a.mask = cap - 1;
    if ((rc = dev_alloc(s, (void**)&a.slots, (size_t)cap * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.first, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.rank, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.blk_cnt, (size_t)(a.nb + 1) * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.blk_next, (size_t)(a.nb + 1) * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&s->rekey_status, (size_t)n_rw * 4))) goto fail;
    if ((rc = session_common_init(s))) goto fail;
    if (n_ops_out) *n_ops_out = n;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
// RW table -> State circuit verdict in one session: the re-keying, the sort and the mock MPT as above; op2row's rows are evaluated in
// the registers they are computed in (state_fused.hpp) and never stored.
extern "C" int zk_state_verify_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n_rw, uint32_t opts, uint64_t* n_ops_out,
                                            zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_state_verify_from_rw_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rw && n_rw > 0 && n_rw < (1ull << 31) - 1, "zk_state_verify_from_rw_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_ASSIGN;
    int rc = rekey_setup(s, rw, rw_flags, n_rw, dev, false, nullptr, nullptr);
    if (rc) { zk_close(s); return rc; }
    const u64 n = s->rekey.n_ops;
    s->n = n;
    s->assign_from_rw = true;
    s->fused_verify = true;
    AssignArgs& a = s->assign;
    StateArgs& v = s->state;
    u32 cap = 16, mcap = 16;
    a.n = n;
    a.rw = s->rekey.rw;
    a.rw_flags = s->rekey.rw_flags;
    a.order = rekey_order(s->rekey);
    a.nb = (u32)((n + ASG_BLOCK - 1) / ASG_BLOCK);
    if ((rc = dev_alloc(s, (void**)&a.mpt, (size_t)n * ASG_MPT_NCELLS * 32))) goto fail;
    while (cap < 2 * n + 2) cap <<= 1;
    mcap = cap;  // (as many MPT rows as ops at most)
    a.mask = cap - 1;
    a.mpt_mask = mcap - 1;
    if ((rc = dev_alloc(s, (void**)&a.slots, ((size_t)cap + mcap) * 4))) goto fail;  // key slots, then the MPT index: one fill
    a.mpt_slots = a.slots + cap;
    if ((rc = dev_alloc(s, (void**)&a.first, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.rank, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.root_rank, (size_t)n * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.blk_cnt, (size_t)(a.nb + 1) * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&a.blk_next, (size_t)(a.nb + 1) * 4))) goto fail;
    if ((rc = dev_alloc(s, (void**)&s->rekey_status, (size_t)n_rw * 4))) goto fail;
    v.rows.cells = nullptr;
    v.rows.flags = nullptr;
    v.rows.n = n;
    v.eval_lo = 0;
    v.eval_hi = n;
    v.mpt.cells = a.mpt;
    v.mpt.flags = nullptr;
    v.mpt.n = (u32)n;  // a capacity: see StateFusedArgs
    v.mpt.ncells = MPT_NCELLS;
    v.mpt.slots = a.mpt_slots;
    v.mpt.mask = a.mpt_mask;
    if ((rc = session_common_init(s))) goto fail;
    if (n_ops_out) *n_ops_out = n;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_state_verify_from_rw(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint32_t opts, uint32_t* status_out,
                                       uint64_t* n_ops_out, zk_result* result) {
    ARG_TRY(result, "zk_state_verify_from_rw: result is null");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = nullptr;
    int rc = zk_state_verify_from_rw_open(rw, rw_flags, n, opts, n_ops_out, &s);
    if (rc) return rc;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}
extern "C" int zk_state_ops_from_rw_read(zk_session* s, uint64_t* ops_host, uint32_t* op_flags_host, uint64_t* n_ops_out) {
    ARG_TRY(s && s->kind == SESSION_REKEY, "zk_state_ops_from_rw_read: bad arguments");
    const RekeyArgs& a = s->rekey;
    if (n_ops_out) *n_ops_out = a.n_ops;
    if (ops_host) HIP_TRY(hipMemcpyAsync(ops_host, a.ops, (size_t)a.n_ops * RWK_NSLOTS * 32, hipMemcpyDeviceToHost, s->stream));
    if (op_flags_host) HIP_TRY(hipMemcpyAsync(op_flags_host, a.op_flags, (size_t)a.n_ops * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}
extern "C" int zk_state_ops_from_rw(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint64_t* ops_out, uint32_t* op_flags_out,
                                    uint64_t* n_ops_out, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result && n_ops_out, "zk_state_ops_from_rw: null output");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = nullptr;
    int rc = zk_state_ops_from_rw_open(rw, rw_flags, n, dev ? ops_out : nullptr, dev ? op_flags_out : nullptr, opts, n_ops_out, &s);
    if (rc) return rc;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && !dev) rc = zk_state_ops_from_rw_read(s, ops_out, op_flags_out, n_ops_out);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

// ---- secp256k1 ECDSA verification
extern "C" int zk_ecdsa_open_batches(const zk_ecdsa_batch* bt, uint32_t n_batches, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_ecdsa_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && bt && (n_batches == 1 || n_batches == 2), "zk_ecdsa_open_batches: one or two batches");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    u64 n = 0;
    for (u32 k = 0; k < n_batches; k++) {
        ARG_TRY(bt[k].bytes && bt[k].n > 0 && bt[k].n < (1ull << 31) && bt[k].layout <= 2u && (!bt[k].v || bt[k].v_stride >= 1), "zk_ecdsa_open: bad arguments");
        ARG_TRY(dev || !bt[k].out_dev, "zk_ecdsa_open: out_dev needs ZK_OPT_DEVICE_PTRS");
        ARG_TRY(!bt[k].out_dev || bt[k].out_stride >= 1, "zk_ecdsa_open: out_stride must be >= 1");
        n += bt[k].n;
    }
    zk_session* s = new zk_session();
    s->kind = SESSION_ECDSA;
    s->n = n;
    EcdsaArgs& a = s->ecdsa;
    int rc = 0;
    const void* p = nullptr;
    // layout 0: packed uint8[n][5][32] (msg_hash big-endian); 1 / 2: the Tx / Sig units' byte rows uint8[n][9][32]
    // (rows 2, 3, 5, 7, 8; the Tx chip keeps msg_hash little-endian, the Sig chip big-endian)
    static const u32 OFF[2][5] = {{0, 32, 64, 96, 128}, {64, 96, 160, 224, 256}};
    a.n = n;
    ecdsa_single_batch(a);
    a.n0 = bt[0].n;
    {   // the fixed-base table of G: built once per device (ordered before this session's first pass: same stream + a synchronisation)
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        if (!g_secp_comb[s->device]) {
            u32* tab = nullptr;
            if (hipMalloc(&tab, (size_t)ECDSA_COMB_ENTRIES * 16 * sizeof(u32)) != hipSuccess) { rc = -2; g_err = "zk_ecdsa_open: table allocation failed"; goto fail; }
            zk_launch_ecdsa_comb_build(s->stream, tab);
            if (hipStreamSynchronize(s->stream) != hipSuccess) { (void)hipFree(tab); rc = -2; g_err = "zk_ecdsa_open: table build failed"; goto fail; }
            g_secp_comb[s->device] = tab;
        }
        a.gcomb = g_secp_comb[s->device];
    }
    for (u32 k = 0; k < n_batches; k++) {
        const u64 stride = bt[k].layout ? 288 : 160;
        const uint8_t* d_bytes;
        const u32* d_v = nullptr;
        if ((rc = stage(s, bt[k].bytes, (size_t)bt[k].n * stride, dev, &p))) goto fail;
        d_bytes = (const uint8_t*)p;
        if (bt[k].v) {
            if ((rc = stage(s, bt[k].v, (size_t)bt[k].n * 4 * bt[k].v_stride, dev, &p))) goto fail;
            d_v = (const u32*)p;
        }
        if (k == 0) {
            a.stride = stride; a.msg_be = bt[k].layout != 1u; a.bytes = d_bytes; a.v = d_v; a.v_stride = bt[k].v_stride;
            a.out = bt[k].out_dev; a.out_stride = bt[k].out_stride;
            for (int c = 0; c < 5; c++) a.off[c] = OFF[bt[k].layout ? 1 : 0][c];
        } else {
            a.stride1 = stride; a.msg_be1 = bt[k].layout != 1u; a.bytes1 = d_bytes; a.v1 = d_v; a.v_stride1 = bt[k].v_stride;
            a.out1 = bt[k].out_dev; a.out_stride1 = bt[k].out_stride;
            for (int c = 0; c < 5; c++) a.off1[c] = OFF[bt[k].layout ? 1 : 0][c];
        }
    }
    // lane pairs while the batch cannot fill the chip anyway (the pass is then bound by one lane's dependent chain: halve it);
    // one lane per signature beyond that (less total work); lane quads up to 2^14 signatures — one wavefront per SIMD at most —, where the
    // two extra lanes take u1 G off the pair's chain (secp256k1.hpp ecdsa_partial4: 1.09-1.11 -> 0.99-1.02 ms; 2^15 signatures as quads
    // are two wavefronts per SIMD: 1.67 ms against 1.13).  ZK_ECDSA_LANES=1|2|4 overrides (tuning / tests).
    a.lanes_per_sig = n <= (1ull << 14) ? 4u : n <= (1ull << 16) ? 2u : 1u;
    if (const char* e = getenv("ZK_ECDSA_LANES")) { const int v = atoi(e); a.lanes_per_sig = v == 4 ? 4u : v == 2 ? 2u : 1u; }
    if (a.lanes_per_sig == 4u && !a.gcomb) a.lanes_per_sig = 2u;  // (the four-lane form takes u1 G from the comb table)
    a.first = 0;
    a.qtab_lanes = ((n * a.lanes_per_sig + 63) / 64) * 64;
    if (a.qtab_lanes > ZK_ECDSA_CHUNK_LANES) a.qtab_lanes = ZK_ECDSA_CHUNK_LANES;  // larger batches: chunked launches over one set of tables
    if ((rc = dev_alloc(s, (void**)&a.qtab, (size_t)a.qtab_lanes * 15 * 24 * sizeof(u32)))) goto fail;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_ecdsa_open(const uint8_t* bytes, uint32_t layout, const uint32_t* v, uint32_t v_stride, uint64_t n,
                             uint32_t* out_dev, uint32_t out_stride, uint32_t opts, zk_session** out) {
    zk_ecdsa_batch b;
    b.bytes = bytes; b.layout = layout; b.v = v; b.v_stride = v_stride; b.n = n; b.out_dev = out_dev; b.out_stride = out_stride;
    return zk_ecdsa_open_batches(&b, 1, opts, out);
}
extern "C" int zk_ecdsa_verify(const uint8_t* bytes, uint32_t layout, const uint32_t* v, uint32_t v_stride, uint64_t n,
                               uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_ecdsa_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_ecdsa_open(bytes, layout, v, v_stride, n, nullptr, 0, opts, &s);
    if (rc) return rc;
    return one_shot(s, opts & ZK_OPT_DEVICE_PTRS, status_out, result);
}

// ---- Bytecode-circuit witness assignment
extern "C" int zk_bytecode_assign_open(const uint64_t* in_rows, uint64_t n_rows, const uint64_t* offsets, const uint64_t* lengths,
                                       uint64_t n_codes, uint32_t k, const uint64_t* randomness, uint64_t* rows_dev, uint32_t opts,
                                       zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_bytecode_assign_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && randomness && k >= 1 && k <= 28 && n_rows < (1ull << 31) && n_codes < (1ull << 31) && (n_codes == 0 || (offsets && lengths)) &&
            (n_rows == 0 || in_rows), "zk_bytecode_assign_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    ARG_TRY(dev || !rows_dev, "zk_bytecode_assign_open: rows_dev needs ZK_OPT_DEVICE_PTRS");
    zk_session* s = new zk_session();
    s->kind = SESSION_BCA;
    s->n = 1ull << k;
    BcaArgs& a = s->bca;
    int rc = 0;
    const void* p = nullptr;
    std::vector<u64> h_off((size_t)n_codes + 1, 0);
    std::vector<BcaChunk> chunks;
    std::vector<u32> code_chunk0((size_t)n_codes + 1, 0);
    u64 rh[4];
    Fr r;
    // the chunk table is index plumbing over the row offsets (not witness data): built on the host
    if (n_codes) {
        if (dev) {
            if (fetch_small(s->stream, h_off.data(), offsets, (n_codes + 1) * 8)) { rc = -2; g_err = "offsets download failed"; goto fail; }
        } else {
            memcpy(h_off.data(), offsets, (n_codes + 1) * 8);
        }
    }
    for (u64 j = 0; j < n_codes; j++) {
        if (h_off[j] > h_off[j + 1] || h_off[j + 1] > n_rows) { rc = -1; g_err = "zk_bytecode_assign_open: offsets must be non-decreasing and within the rows"; goto fail; }
        code_chunk0[j] = (u32)chunks.size();
        for (u64 g = h_off[j]; g < h_off[j + 1]; g += BCA_CHUNK) {
            BcaChunk c;
            c.code = (u32)j; c.start = (u32)g;
            c.count = (u32)((h_off[j + 1] - g < BCA_CHUNK) ? h_off[j + 1] - g : BCA_CHUNK);
            c.first = g == h_off[j] ? 1u : 0u;
            chunks.push_back(c);
        }
    }
    code_chunk0[n_codes] = (u32)chunks.size();
    if (n_codes && (h_off[0] != 0 || h_off[n_codes] != n_rows)) { rc = -1; g_err = "zk_bytecode_assign_open: offsets must cover every row"; goto fail; }
    if ((rc = stage(s, in_rows, (size_t)n_rows * BCA_IN_NCELLS * 32, dev, &p))) goto fail;
    a.in_rows = (const u64*)p;
    if ((rc = stage(s, n_codes ? offsets : nullptr, (size_t)(n_codes + 1) * 8, dev, &p))) goto fail;
    a.offsets = (const u64*)p;
    if ((rc = stage(s, lengths, (size_t)n_codes * 8, dev, &p))) goto fail;
    a.lengths = (const u64*)p;
    a.n_in = n_rows; a.n_codes = n_codes; a.n_out = 1ull << k; a.n_chunks = chunks.size();
    if (dev) {
        if (fetch_small(s->stream, rh, randomness, 32)) { rc = -2; g_err = "randomness download failed"; goto fail; }
    } else {
        memcpy(rh, randomness, 32);
    }
    for (int q = 0; q < 4; q++) { r.v[2 * q] = (u32)rh[q]; r.v[2 * q + 1] = (u32)(rh[q] >> 32); }
    {
        u64* d_rpow = nullptr;
        void* d = nullptr;
        if ((rc = dev_alloc(s, (void**)&d_rpow, BCA_RPOW_ROWS * 32))) goto fail;
        zk_launch_bca_rpow(s->stream, r, d_rpow);
        a.rpow = d_rpow;
        if ((rc = stage(s, chunks.empty() ? nullptr : chunks.data(), chunks.size() * sizeof(BcaChunk), false, &p))) goto fail;
        a.chunks = (const BcaChunk*)p;
        if ((rc = stage(s, code_chunk0.data(), code_chunk0.size() * 4, false, &p))) goto fail;
        a.code_chunk0 = (const u32*)p;
        // the host vectors above go out of scope with this call
        if (hipStreamSynchronize(s->stream) != hipSuccess) { rc = -2; g_err = "zk_bytecode_assign_open: upload failed"; goto fail; }
        if ((rc = dev_alloc(s, &d, (size_t)n_rows * 2))) goto fail;
        a.track = (uint8_t*)d;
        if ((rc = dev_alloc(s, &d, chunks.size() * 32))) goto fail;
        a.chunk_acc = (u64*)d;
        if ((rc = dev_alloc(s, &d, chunks.size() * 4))) goto fail;
        a.chunk_m = (u32*)d;
        if ((rc = dev_alloc(s, &d, chunks.size() * (size_t)BCA_MAP_STRIDE))) goto fail;
        a.chunk_map = (uint8_t*)d;
        if ((rc = dev_alloc(s, &d, chunks.size() * 4))) goto fail;
        a.chunk_state = (u32*)d;
        if ((rc = dev_alloc(s, &d, chunks.size() * 32))) goto fail;
        a.chunk_in = (u64*)d;
        if ((rc = dev_alloc(s, &d, (size_t)n_rows * 32))) goto fail;
        a.rlc = (u64*)d;
        if ((rc = dev_alloc(s, &d, (size_t)n_rows * 4))) goto fail;
        a.row_code = (u32*)d;
        if ((rc = dev_alloc(s, &d, (size_t)n_rows * 4))) goto fail;
        a.row_chunk = (u32*)d;
    }
    a.rows = rows_dev;
    if (!a.rows && (rc = dev_alloc(s, (void**)&a.rows, (size_t)a.n_out * BCA_OUT_NCELLS * 32))) goto fail;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_bytecode_assign_read(zk_session* s, uint64_t* rows_host) {
    ARG_TRY(s && rows_host && s->kind == SESSION_BCA, "zk_bytecode_assign_read: bad arguments");
    HIP_TRY(hipMemcpyAsync(rows_host, s->bca.rows, (size_t)s->bca.n_out * BCA_OUT_NCELLS * 32, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}
extern "C" int zk_bytecode_assign(const uint64_t* in_rows, uint64_t n_rows, const uint64_t* offsets, const uint64_t* lengths,
                                  uint64_t n_codes, uint32_t k, const uint64_t* randomness, uint64_t* rows_out, uint32_t opts,
                                  zk_result* result) {
    ARG_TRY(result && rows_out, "zk_bytecode_assign: null output");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = nullptr;
    int rc = zk_bytecode_assign_open(in_rows, n_rows, offsets, lengths, n_codes, k, randomness, dev ? rows_out : nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && !dev) rc = zk_bytecode_assign_read(s, rows_out);
    zk_close(s);
    return rc;
}


// ---- Copy-circuit witness assignment
#include "copy_assign_plan.hpp"
// host copies of the event arrays when the caller's are device pointers
static int cpa_fetch(const zk_copy_events* t, bool dev, std::vector<u64>& cells, std::vector<u32>& flags, std::vector<u64>& offs,
                     const u64** pc, const u32** pf, const u64** po) {
    *pc = t->events; *pf = t->flags; *po = t->data_offsets;
    if (!dev) return 0;
    cells.resize((size_t)t->n_events * CPA_EV_NCELLS * 4);
    flags.resize((size_t)t->n_events);
    offs.resize((size_t)t->n_events + 1);
    const hipStream_t st = t_stream ? t_stream : g_own_stream[t_device];  // (the calling thread's stream: see d2h_now)
    HIP_TRY(hipMemcpyAsync(cells.data(), t->events, cells.size() * 8, hipMemcpyDeviceToHost, st));
    if (t->flags) HIP_TRY(hipMemcpyAsync(flags.data(), t->flags, flags.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(offs.data(), t->data_offsets, offs.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *pc = cells.data(); *pf = t->flags ? flags.data() : nullptr; *po = offs.data();
    return 0;
}
extern "C" int zk_copy_assign_sizes(const zk_copy_events* t, uint32_t opts, uint64_t* n_rows, uint64_t* n_table, uint64_t* n_rw) {
    ARG_TRY(t && t->n_events > 0 && t->events && t->data_offsets, "zk_copy_assign_sizes: bad arguments");
    std::vector<u64> cells, offs;
    std::vector<u32> flags;
    const u64 *pc, *po;
    const u32* pf;
    int rc = cpa_fetch(t, opts & ZK_OPT_DEVICE_PTRS, cells, flags, offs, &pc, &pf, &po);
    if (rc) return rc;
    CpaPlan pl;
    if ((rc = cpa_plan(pc, pf, po, t->n_events, pl))) return rc;
    if (n_rows) *n_rows = pl.n_rows;
    if (n_table) *n_table = pl.n_table;
    if (n_rw) *n_rw = pl.n_rw;
    return 0;
}
extern "C" int zk_copy_assign_open(const zk_copy_events* t, uint64_t* rows_dev, uint32_t* row_flags_dev, uint64_t* table_dev,
                                   uint64_t* rw_dev, uint32_t* rw_flags_dev, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_copy_assign_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(t && out && t->n_events > 0 && t->n_events < (1ull << 31) && t->events && t->data_offsets && t->randomness, "zk_copy_assign_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    ARG_TRY(dev || (!rows_dev && !row_flags_dev && !table_dev && !rw_dev && !rw_flags_dev), "zk_copy_assign_open: output buffers need ZK_OPT_DEVICE_PTRS");
    std::vector<u64> cells, offs;
    std::vector<u32> flags;
    const u64 *pc, *po;
    const u32* pf;
    int rc = cpa_fetch(t, dev, cells, flags, offs, &pc, &pf, &po);
    if (rc) return rc;
    CpaPlan pl;
    if ((rc = cpa_plan(pc, pf, po, t->n_events, pl))) return rc;
    zk_session* s = new zk_session();
    s->kind = SESSION_CPA;
    s->n = pl.n_rows;
    s->cpa_n_table = pl.n_table;
    s->cpa_n_rw = pl.n_rw;
    CpaArgs& a = s->cpa;
    const void* p = nullptr;
    void* d = nullptr;
    u64 rh[4];
    Fr r;
    if ((rc = stage(s, t->events, (size_t)t->n_events * CPA_EV_NCELLS * 32, dev, &p))) goto fail;
    a.events = (const u64*)p;
    if ((rc = stage(s, t->data, (size_t)pl.n_data * 2, dev, &p))) goto fail;
    a.data = (const uint16_t*)p;
    if ((rc = stage(s, pl.ev.data(), pl.ev.size() * sizeof(CpaEvent), false, &p))) goto fail;
    a.ev = (const CpaEvent*)p;
    if ((rc = stage(s, pl.row0.data(), pl.row0.size() * 8, false, &p))) goto fail;
    a.row0 = (const u64*)p;
    if ((rc = stage(s, pl.chunks.empty() ? nullptr : pl.chunks.data(), pl.chunks.size() * sizeof(CpaChunk), false, &p))) goto fail;
    a.chunks = (const CpaChunk*)p;
    a.n_events = t->n_events; a.n_rows = pl.n_rows; a.n_chunks = pl.chunks.size();
    if (dev) {
        if (fetch_small(s->stream, rh, t->randomness, 32)) { rc = -2; g_err = "randomness download failed"; goto fail; }
    } else {
        memcpy(rh, t->randomness, 32);
    }
    for (int q = 0; q < 4; q++) { r.v[2 * q] = (u32)rh[q]; r.v[2 * q + 1] = (u32)(rh[q] >> 32); }
    if ((rc = dev_alloc(s, &d, CPA_RPOW_ROWS * 32))) goto fail;
    a.rpow = (const u64*)d;
    zk_launch_cpa_rpow(s->stream, r, (u64*)d);
    // the plan vectors go out of scope with this call
    if (hipStreamSynchronize(s->stream) != hipSuccess) { rc = -2; g_err = "zk_copy_assign_open: upload failed"; goto fail; }
    if ((rc = dev_alloc(s, &d, pl.chunks.size() * 32))) goto fail;
    a.chunk_acc = (u64*)d;
    if ((rc = dev_alloc(s, &d, pl.chunks.size() * 32))) goto fail;
    a.chunk_in = (u64*)d;
    if ((rc = dev_alloc(s, &d, (size_t)t->n_events * 32))) goto fail;
    a.ev_rlc = (u64*)d;
    a.rlc = nullptr;  // the row lanes compute the running values themselves (copy_assign.hpp cpa_rlc_value)
    a.rows = rows_dev; a.row_flags = row_flags_dev; a.table = table_dev; a.rw = rw_dev; a.rw_flags = rw_flags_dev;
    if (!a.rows && (rc = dev_alloc(s, (void**)&a.rows, (size_t)pl.n_rows * CPA_ROW_NCELLS * 32))) goto fail;
    if (!a.row_flags && (rc = dev_alloc(s, (void**)&a.row_flags, (size_t)pl.n_rows * 4))) goto fail;
    if (!a.table && (rc = dev_alloc(s, (void**)&a.table, (size_t)pl.n_table * CPA_TABLE_NCELLS * 32))) goto fail;
    if (!a.rw && (rc = dev_alloc(s, (void**)&a.rw, (size_t)pl.n_rw * CPA_RW_NCELLS * 32))) goto fail;
    if (!a.rw_flags && (rc = dev_alloc(s, (void**)&a.rw_flags, (size_t)pl.n_rw * 4))) goto fail;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_copy_assign_read(zk_session* s, uint64_t* rows_host, uint32_t* row_flags_host, uint64_t* table_host,
                                   uint64_t* rw_host, uint32_t* rw_flags_host) {
    ARG_TRY(s && s->kind == SESSION_CPA, "zk_copy_assign_read: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    const CpaArgs& a = s->cpa;
    if (rows_host) HIP_TRY(hipMemcpyAsync(rows_host, a.rows, (size_t)a.n_rows * CPA_ROW_NCELLS * 32, hipMemcpyDeviceToHost, s->stream));
    if (row_flags_host) HIP_TRY(hipMemcpyAsync(row_flags_host, a.row_flags, (size_t)a.n_rows * 4, hipMemcpyDeviceToHost, s->stream));
    if (table_host && s->cpa_n_table) HIP_TRY(hipMemcpyAsync(table_host, a.table, (size_t)s->cpa_n_table * CPA_TABLE_NCELLS * 32, hipMemcpyDeviceToHost, s->stream));
    if (rw_host && s->cpa_n_rw) HIP_TRY(hipMemcpyAsync(rw_host, a.rw, (size_t)s->cpa_n_rw * CPA_RW_NCELLS * 32, hipMemcpyDeviceToHost, s->stream));
    if (rw_flags_host && s->cpa_n_rw) HIP_TRY(hipMemcpyAsync(rw_flags_host, a.rw_flags, (size_t)s->cpa_n_rw * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}
extern "C" int zk_copy_assign(const zk_copy_events* t, uint64_t* rows_out, uint32_t* row_flags_out, uint64_t* table_out,
                              uint64_t* rw_out, uint32_t* rw_flags_out, uint32_t opts, zk_result* result) {
    ARG_TRY(result && rows_out && row_flags_out, "zk_copy_assign: null output");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = nullptr;
    int rc = zk_copy_assign_open(t, dev ? rows_out : nullptr, dev ? row_flags_out : nullptr, dev ? table_out : nullptr,
                                 dev ? rw_out : nullptr, dev ? rw_flags_out : nullptr, opts, &s);
    if (rc) return rc;
    rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && !dev) rc = zk_copy_assign_read(s, rows_out, row_flags_out, table_out, rw_out, rw_flags_out);
    zk_close(s);
    return rc;
}

// ---- Public-inputs circuit
extern "C" int zk_pi_open(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* gas, uint64_t n_gas,
                          uint64_t circuit_len, const uint64_t* keccak_rand, const uint64_t* byte_pow_base, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_pi_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && rows && keccak_rand && byte_pow_base && n > 0 && n < (1ull << 32) && n_keccak < (1ull << 31) && n_gas < (1ull << 31), "zk_pi_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_PI;
    s->n = n;
    int rc = 0;
    const void* p = nullptr;
    u64 rh[8];
    if ((rc = stage(s, rows, (size_t)n * PI_NCELLS * 32, dev, &p))) goto fail;
    s->pi.rows.cells = (const u64*)p;
    s->pi.rows.flags = nullptr;
    s->pi.rows.n = n;
    if ((rc = table_stage(s, s->pi.keccak, keccak, nullptr, n_keccak, KECCAK_NCELLS, dev))) goto fail;
    if ((rc = build_index<keccak_key_hash>(s, s->pi.keccak))) goto fail;
    if ((rc = table_stage(s, s->pi.gas, gas, nullptr, n_gas, PI_GAS_NCELLS, dev))) goto fail;
    if ((rc = build_index<pi_gas_key_hash>(s, s->pi.gas))) goto fail;
    if (dev) {
        if (d2h_now(s->stream, rh, keccak_rand, 32) || d2h_now(s->stream, rh + 4, byte_pow_base, 32)) {
            rc = -2; g_err = "randomness download failed"; goto fail;
        }
    } else {
        memcpy(rh, keccak_rand, 32);
        memcpy(rh + 4, byte_pow_base, 32);
    }
    {
        Fr kr, bp;
        for (int k = 0; k < 4; k++) {
            kr.v[2 * k] = (u32)rh[k]; kr.v[2 * k + 1] = (u32)(rh[k] >> 32);
            bp.v[2 * k] = (u32)rh[4 + k]; bp.v[2 * k + 1] = (u32)(rh[4 + k] >> 32);
        }
        u64* d_m = nullptr;
        u64 hm[8];
        if ((rc = dev_alloc(s, (void**)&d_m, 64))) goto fail;
        zk_launch_fr_to_mont(s->stream, kr, d_m);
        zk_launch_fr_to_mont(s->stream, bp, d_m + 4);
        if (hipMemcpyAsync(hm, d_m, 64, hipMemcpyDeviceToHost, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
            rc = -2; g_err = "constant download failed"; goto fail;
        }
        for (int k = 0; k < 4; k++) {
            s->pi.keccak_rand_m.v[2 * k] = (u32)hm[k]; s->pi.keccak_rand_m.v[2 * k + 1] = (u32)(hm[k] >> 32);
            s->pi.byte_pow_base_m.v[2 * k] = (u32)hm[4 + k]; s->pi.byte_pow_base_m.v[2 * k + 1] = (u32)(hm[4 + k] >> 32);
        }
    }
    s->pi.circuit_len = Fr{{(u32)circuit_len, (u32)(circuit_len >> 32), 0, 0, 0, 0, 0, 0}};
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
// PI circuit copy constraints (pi_circuit.py:355-445; csrc/pi_circuit.hpp pi_copy_check)
extern "C" int zk_pi_copy_open(const uint64_t* cells, const uint8_t* bytes, const uint32_t* lens, uint64_t n, uint32_t opts, zk_session** out) {
    ARG_TRY(t_device >= 0, "zk_pi_copy_open: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(out && cells && bytes && lens && n > 0 && n < (1ull << 32), "zk_pi_copy_open: bad arguments");
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    zk_session* s = new zk_session();
    s->kind = SESSION_PICOPY;
    s->n = n;
    int rc = 0;
    const void* p = nullptr;
    if ((rc = stage(s, cells, (size_t)n * 32, dev, &p))) goto fail;
    s->picopy.cells = (const u64*)p;
    if ((rc = stage(s, bytes, (size_t)n * 32, dev, &p))) goto fail;
    s->picopy.bytes = (const uint8_t*)p;
    if ((rc = stage(s, lens, (size_t)n * 4, dev, &p))) goto fail;
    s->picopy.lens = (const u32*)p;
    s->picopy.n = n;
    if ((rc = session_common_init(s))) goto fail;
    *out = s;
    return 0;
fail:
    zk_close(s);
    return rc;
}
extern "C" int zk_pi_copy_verify(const uint64_t* cells, const uint8_t* bytes, const uint32_t* lens, uint64_t n, uint32_t opts, uint32_t* status_out,
                                 zk_result* result) {
    ARG_TRY(result, "zk_pi_copy_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_pi_copy_open(cells, bytes, lens, n, opts, &s);
    if (rc) return rc;
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}
extern "C" int zk_pi_verify(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* gas, uint64_t n_gas,
                            uint64_t circuit_len, const uint64_t* keccak_rand, const uint64_t* byte_pow_base, uint32_t opts,
                            uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_pi_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_pi_open(rows, n, keccak, n_keccak, gas, n_gas, circuit_len, keccak_rand, byte_pow_base, opts, &s);
    if (rc) return rc;
    return one_shot(s, opts & ZK_OPT_DEVICE_PTRS, status_out, result);
}

extern "C" int zk_copy_verify(const zk_copy_tables* t, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_copy_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_copy_open(t, opts, &s);
    if (rc) return rc;
    return one_shot(s, opts & ZK_OPT_DEVICE_PTRS, status_out, result);
}

static int one_shot(zk_session* s, bool dev, uint32_t* status_out, zk_result* result) {
    int rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}
extern "C" int zk_bytecode_verify(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak,
                                  const uint64_t* randomness, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_bytecode_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_bytecode_open(rows, n, keccak, n_keccak, randomness, opts, &s);
    if (rc) return rc;
    return one_shot(s, opts & ZK_OPT_DEVICE_PTRS, status_out, result);
}
extern "C" int zk_exp_verify(const uint64_t* rows, uint64_t n, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_exp_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_exp_open(rows, n, opts, &s);
    if (rc) return rc;
    return one_shot(s, opts & ZK_OPT_DEVICE_PTRS, status_out, result);
}

static inline u64 range_lo(const zk_session* s) { return s->eval_hi ? s->eval_lo : 0; }
static inline u64 range_hi(const zk_session* s) { return s->eval_hi ? s->eval_hi : s->n; }
extern "C" int zk_set_range(zk_session* s, uint64_t row_lo, uint64_t row_hi) {
    ARG_TRY(s && (s->kind == SESSION_STATE || s->kind == SESSION_BYTECODE || s->kind == SESSION_COPY || s->kind == SESSION_EXP ||
                  s->kind == SESSION_SIGN || s->kind == SESSION_PI),
            "zk_set_range: not a row-circuit session (EVM sessions shard by the steps they are opened over)");
    ARG_TRY(row_lo < row_hi && row_hi <= s->n, "zk_set_range: bad range");
    HIP_TRY(hipSetDevice(s->device));
    s->eval_lo = row_lo;
    s->eval_hi = row_hi;
    if (s->kind == SESSION_STATE) {
        s->state.eval_lo = row_lo;
        s->state.eval_hi = row_hi;
    }
    HIP_TRY(hipMemsetAsync(s->d_status, 0, (size_t)s->n * sizeof(u32), s->stream));
    return 0;
}
extern "C" int zk_state_set_range(zk_session* s, uint64_t row_lo, uint64_t row_hi) {
    ARG_TRY(s && s->kind == SESSION_STATE, "zk_state_set_range: not a State session");
    return zk_set_range(s, row_lo, row_hi);
}

// tuning aid (not part of the public ABI): the pairs the fast EVM kernel deferred to the general build in the last pass
extern "C" int zk_debug_read_deferred(zk_session* s, uint32_t* count, uint32_t* list, uint32_t max) {
    if (!s || s->kind != SESSION_EVM || !s->evm.defer_count) return -1;
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(count, s->evm.defer_count, 4, hipMemcpyDeviceToHost));
    const uint32_t n = *count < max ? *count : max;
    if (n && list) HIP_TRY(hipMemcpy(list, s->evm.defer_list, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}
// tuning aid (not part of the public ABI): copy the phase timestamps of the last pass
extern "C" int zk_debug_read_prof(zk_session* s, unsigned long long* out) {
    if (!s || !s->evm.prof) return -1;
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out, s->evm.prof, 1024 * 4 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// profiling aid (not part of the public ABI): read `nbytes` exactly once in the EVM kernel's
// access pattern -- every lane walks its own `lane_bytes`-byte record with 16-byte loads -- so the
// FETCH_SIZE counter can be calibrated against a known byte count (MI355X guide, HBM section).
__global__ void __launch_bounds__(256) calib_gather_kernel(const uint4* __restrict__ buf, u64 n_rec, u32 vec_per_rec,
                                                           u32* __restrict__ sink) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const uint4* p = buf + i * vec_per_rec;
    u32 acc = 0;
    for (u32 k = 0; k < vec_per_rec; ++k) {
        const uint4 v = p[k];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;  // keeps the loads alive
}
extern "C" int zk_debug_calib_gather(uint64_t nbytes, uint32_t lane_bytes, float* ms_out) {
    ARG_TRY(lane_bytes && lane_bytes % 16 == 0 && nbytes >= lane_bytes, "zk_debug_calib_gather: bad sizes");
    const u64 n_rec = nbytes / lane_bytes;
    uint4* buf = nullptr;
    u32* sink = nullptr;
    HIP_TRY(hipMalloc(&buf, n_rec * lane_bytes));
    HIP_TRY(hipMalloc(&sink, 64));
    HIP_TRY(hipMemsetAsync(buf, 0x5a, n_rec * lane_bytes, t_stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, t_stream));
    calib_gather_kernel<<<dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, t_stream>>>(buf, n_rec, lane_bytes / 16, sink);
    HIP_TRY(hipEventRecord(e1, t_stream));
    HIP_TRY(hipStreamSynchronize(t_stream));
    if (ms_out) HIP_TRY(hipEventElapsedTime(ms_out, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(buf);
    hipFree(sink);
    return 0;
}

// The lazy tail of a one-shot EVM pass (zk_session::early_seq): wait for the lane counts the open's scatter published — they arrive
// while the hot kernel is still queued or running —, then enqueue the warm / cold builds whose range is not empty, the warm one sized
// to its range.  Called by everything that consumes or follows a pass (zk_collect, zk_read_status, the next zk_launch).
static int evm_enqueue_tail(zk_session* s) {
    if (s->kind != SESSION_EVM || !s->tail_pending) return 0;
    s->tail_pending = false;
    s->stream_drained = false;
    const u32* hw = (const u32*)s->h_result;
    const auto t0 = std::chrono::steady_clock::now();
    u64 spins = 0;
    bool seen = true;
    while (__atomic_load_n(&hw[EVM_EARLY_WORD + 2], __ATOMIC_ACQUIRE) != s->early_seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { seen = false; break; }  // never expected
    }
    u32 warm_lanes = 0;
    bool run_warm = true, run_cold = true;
    if (seen) {
        warm_lanes = hw[EVM_EARLY_WORD];
        const u32 cold_lanes = hw[EVM_EARLY_WORD + 1];
        run_warm = warm_lanes != 0;
        run_cold = cold_lanes != 0;
        s->evm_ranges_known = 1;  // what a collect's read of group_start would have said
        s->evm_warm_empty = !run_warm;
        s->evm_warm_lanes = warm_lanes;
        s->evm_cold_empty = !run_cold;
    }
    s->early_seq = 0;
    if (run_warm) zk_launch_evm_warm(s->stream, s->tail_cold_grid, warm_lanes, s->evm, s->d_group_start, s->tail_status, s->tally_last, run_cold ? nullptr : s->tail_e1);
    if (run_cold) zk_launch_evm_cold(s->stream, s->tail_cold_grid, s->evm, s->d_group_start, s->tail_status, s->tally_last, s->tail_e1);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int zk_launch(zk_session* s, uint32_t* status_dev) {
    ARG_TRY(s, "zk_launch: null session");
    s->stream_drained = false;
    HIP_TRY(hipSetDevice(s->device));
    { int trc = evm_enqueue_tail(s); if (trc) return trc; }  // (a pass launched twice without a collect in between)
    s->publish_enqueued = false;  // (a publish kernel already behind the previous pass says nothing about this one)
    s->status_external = status_dev != nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = s->launches < (u32)MAX_EVENT_PAIRS;
    if (timed) {
        if (s->ev.size() < 2 * (size_t)(s->launches + 1)) {
            int erc = arena_event(s->device, &e0);
            if (erc) return erc;
            s->ev.push_back(e0);
            if ((erc = arena_event(s->device, &e1))) return erc;
            s->ev.push_back(e1);
        }
        e0 = s->ev[2 * s->launches];
        e1 = s->ev[2 * s->launches + 1];
    }
    const bool twin_tally = s->kind == SESSION_STATE || s->kind == SESSION_BYTECODE || s->kind == SESSION_COPY ||
                            s->kind == SESSION_SIGN || s->kind == SESSION_EXP || s->kind == SESSION_PI || s->kind == SESSION_PICOPY;
    ZkTally* const tally = twin_tally ? s->d_tally + (s->tally_pass++ & 1u) : s->d_tally;
    s->tally_last = tally;
    if (!twin_tally && !(s->kind == SESSION_EVM && s->evm.perm))
        hipLaunchKernelGGL(tally_reset_kernel, dim3(1), dim3((s->fused_verify && !s->fused_order_ready) ? 2 : 1), 0, s->stream, s->d_tally,
                           s->kind == SESSION_EVM ? s->evm.defer_count : (u32*)nullptr);
    u32* status = status_dev ? status_dev : s->d_status;
    // the state-sorted EVM pass attaches its two timing events to the kernel dispatches themselves (hipExtLaunchKernelGGL):
    // no separate event packets between the sort passes and the evaluation kernels
    bool evm_ext_events = timed && s->kind == SESSION_EVM && s->evm.perm;
    hipStream_t side = nullptr;
    if (s->kind == SESSION_EVM && s->evm.perm && s->side_stream &&
        !(s->evm_ranges_known && s->evm_warm_empty && s->evm_cold_empty)) {
        {   // lazily created under the device mutex (two threads launching side-stream sessions on one device)
            std::lock_guard<std::mutex> lock(g_dev_mutex);
            if (!g_side_stream[s->device]) HIP_TRY(hipStreamCreateWithFlags(&g_side_stream[s->device], hipStreamNonBlocking));
        }
        if (!s->ev_fork) {
            int erc = arena_event(s->device, &s->ev_fork);
            if (!erc) erc = arena_event(s->device, &s->ev_join);
            if (erc) return erc;
        }
        side = g_side_stream[s->device];
        if (side == s->stream) side = nullptr;
    }
    if (side) evm_ext_events = false;  // the pass's events are the fork / join records themselves
    // State sessions: one kernel per pass — its two events ride on the dispatch too (back-to-back passes of 2^16 rows were 28.6 us of
    // kernel in 35.5 us per pass: two event packets between consecutive launches)
    const bool state_ext_events = timed && s->kind == SESSION_STATE && zk_state_rows_events_ride(s->state);
    if (state_ext_events) evm_ext_events = true;  // (same handling below: nothing is recorded around the launch)
    if (timed && !evm_ext_events && !side) HIP_TRY(hipEventRecord(e0, s->stream));
    switch (s->kind) {
    case SESSION_STATE: zk_launch_state_rows(s->stream, s->state, status, tally, state_ext_events ? e0 : nullptr, state_ext_events ? e1 : nullptr); break;
    case SESSION_BYTECODE: zk_launch_bytecode_rows(s->stream, s->bytecode, range_lo(s), range_hi(s), status, tally); break;
    case SESSION_COPY: zk_launch_copy_rows(s->stream, s->copy, range_lo(s), range_hi(s), status, tally); break;
    case SESSION_SIGN: zk_launch_sign_units(s->stream, s->sign, range_lo(s), range_hi(s), status, tally); break;
    case SESSION_EXP: zk_launch_exp_rows(s->stream, s->exp, range_lo(s), range_hi(s), status, tally); break;
    case SESSION_KECCAK: zk_launch_keccak_table(s->stream, s->keccak_gen, status, s->d_tally); break;
    case SESSION_ASSIGN:
        if (s->fused_verify) {
            if (!s->fused_order_ready) {  // (a session's RW table does not change between passes)
                zk_launch_state_rekey(s->stream, s->rekey, s->rekey_status, s->d_tally + 1);
                zk_launch_state_assign(s->stream, s->assign, nullptr, nullptr);  // (the roots alone: root_rank is set)
            }
            zk_launch_state_rows_fused(s->stream, s->state, s->assign, status, s->d_tally, s->d_tally + 1);
            break;
        }
        if (s->assign_from_rw) zk_launch_state_rekey(s->stream, s->rekey, s->rekey_status, s->d_tally);  // rejected RW rows count in the same tally
        zk_launch_state_assign(s->stream, s->assign, status, s->d_tally);
        break;
    case SESSION_ECDSA: zk_launch_ecdsa(s->stream, s->ecdsa, status, s->d_tally); break;
    case SESSION_BCA: zk_launch_bytecode_assign(s->stream, s->bca, status, s->d_tally); break;
    case SESSION_PI: zk_launch_pi_rows(s->stream, s->pi, range_lo(s), range_hi(s), status, tally); break;
    case SESSION_PICOPY: zk_launch_pi_copy(s->stream, s->picopy, status, tally); break;
    case SESSION_CPA: zk_launch_copy_assign(s->stream, s->cpa, status, s->d_tally); break;
    case SESSION_REKEY: zk_launch_state_rekey(s->stream, s->rekey, status, s->d_tally); break;
    case SESSION_EVM: {
        // the state-sorted lane mapping is derived from the step column on every pass
        if (s->evm.perm) {
            // A session's step table does not change between passes (its packed step records and indices, built at open, already assume
            // that): the state-sorted mapping is derived once — by the open's launches, or by the first pass of a session opened without
            // them — and every later pass only resets its tally.  (Until round 6 every pass re-derived it: histogram 10.5 + scatter 7.5 us
            // of a 74-us resident pass, 26 + 19 us beside the other circuits of a block pass.  ZK_EVM_RESORT=1 restores that.)
            static const bool resort = [] { const char* e = getenv("ZK_EVM_RESORT"); return e && e[0] == '1'; }();
            // ... and needs no reset kernel either: the passes alternate between the result block's two tallies / deferred-pair counters, and
            // every hot launch clears the pair the NEXT pass will use (EvmArgs::defer_count_twin), as the single-kernel row sessions do
            if (s->perm_ready) { s->perm_ready = false; s->perm_valid = true; s->evm_tally_idx = 0; }  // the open's launches carried this pass's sort (and reset tally 0)
            else if (s->perm_valid && !resort) s->evm_tally_idx ^= 1u;
            else { int prc = evm_build_perm(s); if (prc) return prc; s->perm_valid = true; s->evm_tally_idx = 0; }
            {
                EvmResultBlock* const rb = (EvmResultBlock*)s->d_result;
                u32* const dc[2] = {&rb->dyn.n_deferred, &rb->n_deferred_twin};
                s->evm.defer_count = dc[s->evm_tally_idx];
                s->evm.defer_count_twin = resort ? nullptr : dc[s->evm_tally_idx ^ 1u];
                s->tally_last = s->d_tally + s->evm_tally_idx;
            }
        }
        // with the sorted mapping the hot lane range is padded per state (EVM_PERM_PAD bounds the padding); blocks past its end exit
        const u32 grid = (u32)((s->n + (s->evm.perm ? EVM_PERM_PAD : 0) + EVM_HOT_BLOCK - 1) / EVM_HOT_BLOCK);
        const u32 all_grid = (u32)((s->n + 255) / 256);
        const u32 cold_grid = s->evm.perm ? (all_grid < 256u ? all_grid : 256u) : all_grid;
        // one kernel with every gadget; with `perm` the lanes are state-sorted (heavy gadget families first); then the
        // rarely-taken states (evm_state_group == COLD): a small grid-stride launch, empty for most traces.  With the sorted
        // mapping the two timing events ride on the dispatches themselves (no event packets between the kernels).
        s->deferred_pending = true;
        s->last_status = status;
        // The warm (copy- / keccak- / exp-table gadgets) and cold (everything rare) instantiations walk their own lane ranges of
        // the sorted mapping.  Which states a session's step table contains does not change between passes: once a collect has
        // seen a range empty (the usual case for the cold one, and for both on BASELINE config 3's mix) it is not launched again.
        const bool sorted = s->evm.perm != nullptr;
        const bool run_warm = !(sorted && s->evm_ranges_known && s->evm_warm_empty);
        const bool run_cold = !(sorted && s->evm_ranges_known && s->evm_cold_empty);
        if (side) {
            // fork after the sort: the small warm / cold launches go first and finish under the hot one
            hipEvent_t fork = timed ? e0 : s->ev_fork;
            HIP_TRY(hipEventRecord(fork, s->stream));
            HIP_TRY(hipStreamWaitEvent(side, fork, 0));
            if (run_warm)
                zk_launch_evm_warm(side, cold_grid, s->evm_ranges_known ? s->evm_warm_lanes : 0u, s->evm, s->d_group_start, status, s->tally_last, nullptr);
            if (run_cold) zk_launch_evm_cold(side, cold_grid, s->evm, s->d_group_start, status, s->tally_last, nullptr);
            HIP_TRY(hipEventRecord(s->ev_join, side));
            zk_launch_evm_hot(s->stream, grid, s->evm, s->d_group_start, status, s->tally_last, nullptr, nullptr);
            HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_join, 0));
            if (status_dev && s->evm.defer_count) {
                zk_launch_evm_deferred(s->stream, s->evm, status, s->tally_last);
                s->deferred_pending = false;
            }
            break;
        }
        if (sorted && !s->evm_ranges_known && s->early_seq && !status_dev && evm_ext_events) {
            // lazy tail: the hot build alone (it carries both events); the warm / cold builds follow from evm_enqueue_tail once the
            // open's scatter has told the host which of their ranges are not empty
            zk_launch_evm_hot(s->stream, grid, s->evm, s->d_group_start, status, s->tally_last, e0, e1);
            s->tail_pending = true;
            s->tail_cold_grid = cold_grid;
            s->tail_status = status;
            s->tail_e1 = e1;
            break;
        }
        hipEvent_t e_hot1 = (evm_ext_events && !run_warm && !run_cold) ? e1 : nullptr;  // the hot dispatch carries both events then
        zk_launch_evm_hot(s->stream, grid, s->evm, s->d_group_start, status, s->tally_last, evm_ext_events ? e0 : nullptr, e_hot1);
        if (run_warm)
            zk_launch_evm_warm(s->stream, cold_grid, (sorted && s->evm_ranges_known) ? s->evm_warm_lanes : 0u, s->evm, s->d_group_start, status, s->tally_last,
                               (evm_ext_events && !run_cold) ? e1 : nullptr);
        if (run_cold) zk_launch_evm_cold(s->stream, cold_grid, s->evm, s->d_group_start, status, s->tally_last, evm_ext_events ? e1 : nullptr);
        // A caller who hands over its own status buffer may read it after its own stream synchronisation, without zk_collect:
        // the general build is enqueued right behind the pass (it reads the deferred count on the device and returns at once
        // when it is zero), so that buffer — and the tally — are final in stream order.  Passes into the session's own buffer
        // keep the lazy form (zk_collect / zk_read_status look at the count in the synchronisation they need anyway).
        if (status_dev && s->evm.defer_count) {
            zk_launch_evm_deferred(s->stream, s->evm, status, s->tally_last);
            s->deferred_pending = false;
        }
        break;
    }
    }
    if (timed && !evm_ext_events) HIP_TRY(hipEventRecord(e1, s->stream));
    HIP_TRY(hipGetLastError());
    s->launches++;
    return 0;
}

// EVM sessions: the hot kernel is the fast build (EVM_FAST); a pair that needs one of the general build's fallback paths — only
// malformed or oddly shaped witnesses produce any — is put on the session's deferred list.  Whoever asks for the pass's results
// first (zk_collect, zk_read_status) looks at the count and, if there are such pairs, runs the general build over them before
// answering.  A well-formed witness costs one extra 4-byte read, in the same synchronisation as the tally.
static int evm_finish_deferred(zk_session* s) {
    { int trc = evm_enqueue_tail(s); if (trc) return trc; }
    if (s->kind != SESSION_EVM || !s->deferred_pending || !s->evm.defer_count) return 0;
    u32 n_def = 0;
    HIP_TRY(hipMemcpyAsync(&n_def, s->evm.defer_count, 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->deferred_pending = false;
    if (n_def) {
        s->stream_drained = false;  // (something is enqueued behind the polled result: zk_close must wait again)
        zk_launch_evm_deferred(s->stream, s->evm, s->last_status, s->tally_last);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// The publish kernel of a pass (zk_collect's polled result block), enqueued behind whatever the pass has in the stream.  zk_collect calls it;
// zk_evm_verify_batch calls it a witness ahead, so that the block is on its way to the host when the collect comes to poll for it.
static bool evm_publish_enqueue(zk_session* s) {
    if (s->kind != SESSION_EVM || !s->d_result || !s->h_result || !s->h_result_dev || !evm_poll_enabled()) return false;
    u32* hw = (u32*)s->h_result;
    const u32 seq = ++s->publish_seq ? s->publish_seq : ++s->publish_seq;  // never 0
    __atomic_store_n(&hw[32], 0u, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(evm_publish_kernel, dim3(1), dim3(64), 0, s->stream, (const u32*)s->d_result, s->h_result_dev, seq);
    return hipGetLastError() == hipSuccess;
}

extern "C" int zk_collect(zk_session* s, zk_result* r) {
    ARG_TRY(s && r, "zk_collect: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    { int trc = evm_enqueue_tail(s); if (trc) return trc; }
    ZkTally t;
    u32 n_def = 0;
    const bool check_deferred = s->kind == SESSION_EVM && s->deferred_pending && s->evm.defer_count;
    u32 gs[EVM_N_GROUPS + 1] = {0};
    const bool read_ranges = s->kind == SESSION_EVM && s->evm.perm && !s->evm_ranges_known && s->launches > 0;
    if (s->kind == SESSION_EVM && s->d_result && s->h_result) {
        // the 128-byte result block (deferred count, both tallies, lane ranges) in page-locked memory: published by evm_publish_kernel and
        // polled (default), or one copy dispatch + one wait.  (Having the pass's last block write the block to the host itself — system-
        // scope stores + fence from inside the cold launch — was measured in round 4: pass 76 -> 90 us.  The separate one-wavefront
        // kernel costs the evaluation launches nothing and saves the runtime's completion-signal path: step 0.179 -> 0.174 ms.)
        const bool poll = evm_poll_enabled();
        bool polled = false;
        if (poll) {
            static_assert(sizeof(EvmResultBlock) == 128 && ZK_PINNED_BYTES >= 4 * (EVM_EARLY_WORD + 3), "flag word behind the result block, early words behind that");
            u32* hw = (u32*)s->h_result;
            void* const hd = s->h_result_dev;
            if (hd) {
                const bool launched = s->publish_enqueued || evm_publish_enqueue(s);
                s->publish_enqueued = false;
                const u32 seq = s->publish_seq;
                if (launched) {
                    const auto t0 = std::chrono::steady_clock::now();
                    u64 spins = 0;
                    while (__atomic_load_n(&hw[32], __ATOMIC_ACQUIRE) != seq) {
                        __builtin_ia32_pause();  // (the host side of this file is x86-64: a polite spin)
                        if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;  // never expected: fall back to the runtime's wait
                    }
                    polled = __atomic_load_n(&hw[32], __ATOMIC_ACQUIRE) == seq;
                }
            }
            (void)hipGetLastError();
        }
        if (polled) {
            s->stream_drained = true;  // the publish kernel is the stream's last command and its stores are done
        } else {
            HIP_TRY(hipMemcpyAsync(s->h_result, s->d_result, sizeof(EvmResultBlock), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
        }
        const EvmResultBlock* h = (const EvmResultBlock*)s->h_result;
        n_def = s->evm_tally_idx ? h->n_deferred_twin : h->dyn.n_deferred;
        for (int k = 0; k <= EVM_N_GROUPS; k++) gs[k] = h->group_start[k];
        t = h->tally[s->tally_last - s->d_tally];
    } else {
        if (check_deferred) HIP_TRY(hipMemcpyAsync(&n_def, s->evm.defer_count, 4, hipMemcpyDeviceToHost, s->stream));  // rides on the tally's synchronisation
        if (read_ranges) HIP_TRY(hipMemcpyAsync(gs, s->d_group_start, sizeof gs, hipMemcpyDeviceToHost, s->stream));
        ZkTally t_asg = {0ull, ~0ull};
        if (s->fused_verify) HIP_TRY(hipMemcpyAsync(&t_asg, s->d_tally + 1, sizeof t_asg, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(&t, s->tally_last, sizeof t, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (t_asg.fail_count) {  // no witness: what assign_state_circuit raises in the reference is an error here, not a verdict
            char msg[240];
            snprintf(msg, sizeof msg, "zk_state_verify_from_rw: the State witness assignment failed for %llu RW rows / ops (first: %llu, code 0x%08x; "
                     "zk_state_assign_from_rw reports each)", (unsigned long long)t_asg.fail_count, (unsigned long long)(t_asg.first_fail >> 32),
                     (unsigned)(t_asg.first_fail & 0xffffffffull));
            g_err = msg;
            s->launches = 0;
            return -1;
        }
        if (s->fused_verify && s->launches > 0) s->fused_order_ready = true;
    }
    if (read_ranges) {
        s->evm_ranges_known = 1;
        s->evm_warm_empty = gs[EVM_GROUP_WARM] == gs[EVM_GROUP_WARM + 1];
        s->evm_warm_lanes = gs[EVM_GROUP_WARM + 1] - gs[EVM_GROUP_WARM];
        s->evm_cold_empty = gs[EVM_GROUP_COLD] == gs[EVM_GROUP_COLD + 1];
    }
    if (check_deferred) {
        s->deferred_pending = false;
        if (n_def) {  // the general build decides the pairs the fast kernel left (see evm_finish_deferred), then the tally is final
            s->stream_drained = false;  // (something is enqueued behind the polled result: zk_close must wait again)
        zk_launch_evm_deferred(s->stream, s->evm, s->last_status, s->tally_last);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&t, s->tally_last, sizeof t, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
        }
    }
    double ms = 0;
    u32 timed = s->launches < (u32)MAX_EVENT_PAIRS ? s->launches : (u32)MAX_EVENT_PAIRS;
    for (u32 k = 0; k < timed; k++) {
        float f = 0;
        hipError_t ee = hipEventElapsedTime(&f, s->ev[2 * k], s->ev[2 * k + 1]);
        if (ee == hipErrorNotReady) {  // (polled result: the runtime may not have looked at the dispatch's signal yet)
            (void)hipGetLastError();
            (void)hipEventSynchronize(s->ev[2 * k + 1]);
            ee = hipEventElapsedTime(&f, s->ev[2 * k], s->ev[2 * k + 1]);
        }
        HIP_TRY(ee);
        ms += f;
    }
    r->fail_count = t.fail_count;
    r->first_fail_row = t.first_fail == ~0ull ? UINT64_MAX : (t.first_fail >> 32);
    r->first_fail_code = t.first_fail == ~0ull ? 0u : (u32)(t.first_fail & 0xffffffffull);
    r->launches = s->launches;
    r->rows_evaluated = range_hi(s) - range_lo(s);
    r->kernel_ms = timed ? ms / timed : 0.0;
    s->launches = 0;
    return 0;
}

extern "C" int zk_read_status(zk_session* s, uint32_t* status_host) {
    ARG_TRY(s && status_host, "zk_read_status: bad arguments");
    ARG_TRY(!s->status_external, "zk_read_status: the last pass wrote its statuses to the caller's status_dev buffer, not the session's");
    HIP_TRY(hipSetDevice(s->device));
    { int drc = evm_finish_deferred(s); if (drc) return drc; }
    HIP_TRY(hipMemcpyAsync(status_host, s->d_status, (size_t)s->n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

// Device-side spans of an EVM session, by HIP events riding on the dispatches: `open_ms` = first to last kernel of zk_evm_open,
// `pass_ms` = the mean evaluation-kernel span of the passes of the last zk_collect, `span_ms` = first open kernel to the end of the
// FIRST pass's last evaluation kernel (what a witness seen once costs on the device).  Valid after a zk_collect; -1 = not measured.
static void session_timing(zk_session* s, double pass_ms, double* open_ms, double* span_ms) {
    *open_ms = *span_ms = -1.0;
    if (s->kind != SESSION_EVM || !s->ev_open0 || !s->ev_open1) return;
    float f = 0;
    if (hipEventElapsedTime(&f, s->ev_open0, s->ev_open1) == hipSuccess) *open_ms = f;
    if (pass_ms > 0 && s->ev.size() >= 2 && hipEventElapsedTime(&f, s->ev_open0, s->ev[1]) == hipSuccess) *span_ms = f;
    (void)hipGetLastError();
}
extern "C" int zk_session_timing(zk_session* s, double* open_ms, double* span_ms) {
    ARG_TRY(s && open_ms && span_ms, "zk_session_timing: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    session_timing(s, 1.0, open_ms, span_ms);
    return 0;
}
extern "C" int zk_last_host_phases(double* us4) {
    if (us4) for (int k = 0; k < 4; k++) us4[k] = t_host_phase[k];
    return 0;
}
extern "C" int zk_timing_sums(double* sums_ms, uint64_t* count, int reset) {
    if (sums_ms) for (int k = 0; k < 3; k++) sums_ms[k] = t_timing_sum[k];
    if (count) *count = t_timing_count;
    if (reset) { t_timing_sum[0] = t_timing_sum[1] = t_timing_sum[2] = 0; t_timing_count = 0; }
    return 0;
}
extern "C" int zk_last_timing(double* open_ms, double* pass_ms, double* span_ms) {
    if (open_ms) *open_ms = t_timing[0];
    if (pass_ms) *pass_ms = t_timing[1];
    if (span_ms) *span_ms = t_timing[2];
    return 0;
}

extern "C" int zk_state_verify(const uint64_t* rows, const uint32_t* flags, uint64_t n, const uint64_t* mpt,
                               uint64_t n_mpt, uint32_t opts, uint32_t* status_out, zk_result* result) {
    ARG_TRY(result, "zk_state_verify: result is null");
    zk_session* s = nullptr;
    int rc = zk_state_open(rows, flags, n, mpt, n_mpt, opts, &s);
    if (rc) return rc;
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    rc = zk_launch(s, (dev && status_out) ? status_out : nullptr);
    if (!rc) rc = zk_collect(s, result);
    if (!rc && status_out && !dev) rc = zk_read_status(s, status_out);
    zk_close(s);
    return rc;
}

extern "C" int zk_fr_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n, uint32_t opts) {
    ARG_TRY(t_device >= 0, "zk_fr_op: call zk_init first");
    HIP_TRY(hipSetDevice(t_device));
    ARG_TRY(a && b && out, "zk_fr_op: null pointer");
    if (n == 0) return 0;
    const bool dev = opts & ZK_OPT_DEVICE_PTRS;
    const u64 *da = a, *db = b;
    u64* dout = out;
    void *ta = nullptr, *tb = nullptr, *to = nullptr;
    if (!dev) {
        HIP_TRY(hipMalloc(&ta, n * 32));
        HIP_TRY(hipMalloc(&tb, n * 32));
        HIP_TRY(hipMalloc(&to, n * 32));
        HIP_TRY(hipMemcpyAsync(ta, a, n * 32, hipMemcpyHostToDevice, t_stream));
        HIP_TRY(hipMemcpyAsync(tb, b, n * 32, hipMemcpyHostToDevice, t_stream));
        da = (const u64*)ta;
        db = (const u64*)tb;
        dout = (u64*)to;
    }
    hipLaunchKernelGGL(fr_op_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, t_stream, op, da, db, dout, n);
    HIP_TRY(hipGetLastError());
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(out, to, n * 32, hipMemcpyDeviceToHost, t_stream));
        HIP_TRY(hipStreamSynchronize(t_stream));
        (void)hipFree(ta);
        (void)hipFree(tb);
        (void)hipFree(to);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// Multi-GPU tally through RCCL (include/zkevm_hip.h "Multi-GPU tally").  librccl is bound at first use: nothing else in the
// library needs it, and torch — when it is in the process — has usually loaded the same SONAME already.
// ---------------------------------------------------------------------------------------
using zkdist::rccl;
using zkdist::RcclId;
static const int RCCL_UINT64 = zkdist::RCCL_UINT64;
#define RCCL_TRY(expr, what)                                                                                        \
    do {                                                                                                            \
        const int r_ = (expr);                                                                                      \
        if (r_ != 0) {                                                                                              \
            char buf_[256];                                                                                         \
            snprintf(buf_, sizeof buf_, "%s: RCCL error %d (%s)", what, r_, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
            g_err = buf_;                                                                                           \
            return -3;                                                                                              \
        }                                                                                                           \
    } while (0)

struct zk_comm {
    void* nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    u64* d_buf = nullptr;  // ZK_TALLY_WORDS of this rank | ZK_TALLY_WORDS * world gathered
    u64* h_buf = nullptr;  // page-locked, same layout, gathered part first
};
#define ZK_TALLY_WORDS zkdist::TALLY_WORDS

extern "C" int zk_dist_unique_id(uint8_t* id) {
    ARG_TRY(id, "zk_dist_unique_id: id is null");
    ARG_TRY(rccl().ok, "zk_dist_unique_id: librccl.so.1 could not be loaded");
    RcclId u;
    RCCL_TRY(rccl().GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, ZK_DIST_ID_BYTES);
    return 0;
}
extern "C" int zk_dist_close(zk_comm* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}
extern "C" int zk_dist_init(const uint8_t* id, int rank, int world, zk_comm** out) {
    ARG_TRY(t_device >= 0, "zk_dist_init: call zk_init first");
    ARG_TRY(id && out && world >= 1 && rank >= 0 && rank < world, "zk_dist_init: bad arguments");
    ARG_TRY(rccl().ok, "zk_dist_init: librccl.so.1 could not be loaded");
    ARG_TRY(rccl().buffers == 1, "zk_dist_init: ZK_RCCL_LIB names a collective library that takes HOST buffers (zk_collective_buffers() == 0): the "
                                 "HIP library hands it device memory");
    HIP_TRY(hipSetDevice(t_device));
    zk_comm* c = new zk_comm();
    c->rank = rank; c->world = world; c->device = t_device;
    RcclId u;
    memcpy(u.internal, id, ZK_DIST_ID_BYTES);
    int rc = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->d_buf, (size_t)(world + 1) * ZK_TALLY_WORDS * sizeof(u64)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_buf, (size_t)(world + 1) * ZK_TALLY_WORDS * sizeof(u64), hipHostMallocDefault) != hipSuccess) {
        g_err = "zk_dist_init: buffer allocation failed";
        rc = -2;
    } else if (int r = rccl().CommInitRank(&c->nccl, world, u, rank)) {
        char buf[256];
        snprintf(buf, sizeof buf, "ncclCommInitRank: RCCL error %d (%s)", r, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
        g_err = buf;
        c->nccl = nullptr;
        rc = -3;
    }
    if (rc) { zk_dist_close(c); return rc; }
    *out = c;
    return 0;
}
extern "C" int zk_dist_tally(zk_comm* c, const zk_result* local, uint64_t row_offset, zk_result* global) {
    ARG_TRY(c && local && global, "zk_dist_tally: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    u64* mine = c->h_buf + (size_t)c->world * ZK_TALLY_WORDS;
    zkdist::tally_pack(mine, local, row_offset);
    HIP_TRY(hipMemcpyAsync(c->d_buf, mine, ZK_TALLY_WORDS * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(rccl().AllGather(c->d_buf, c->d_buf + ZK_TALLY_WORDS, ZK_TALLY_WORDS, RCCL_UINT64, c->nccl, c->stream), "ncclAllGather");
    HIP_TRY(hipMemcpyAsync(c->h_buf, c->d_buf + ZK_TALLY_WORDS, (size_t)c->world * ZK_TALLY_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    zkdist::tally_reduce(c->h_buf, c->world, local, global);
    return 0;
}

// ---------------------------------------------------------------------------------------
// zk_block_verify: a block as a one-shot (include/zkevm_hip.h).  The chains are the C entries above, driven by four host threads:
// every open has its own small host work (class scan + plan, chunk tables, event plan), so the overlap has to come from threads.
// The threads are persistent (four per device, parked on a condition variable); within a chain every session is opened and launched
// before the first one is collected, so the host's opens overlap the device's kernels and a chain waits once, at its end.
// ---------------------------------------------------------------------------------------
#include <condition_variable>
#include <functional>
#include <thread>
struct BlockWorkers {  // (never destroyed: the threads are parked, not joined, when the process ends)
    std::mutex call;   // one block at a time per device
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::function<void()> job[4];
    bool has_job[4] = {false, false, false, false};
    int pending = 0;
    bool started = false;
    void run(int c) {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lock(m);
                cv_job.wait(lock, [&] { return has_job[c]; });
                f = std::move(job[c]);
                has_job[c] = false;
            }
            f();
            {
                std::lock_guard<std::mutex> lock(m);
                pending--;
            }
            cv_done.notify_all();
        }
    }
};
static BlockWorkers* g_block_workers[ZK_MAX_DEVICES] = {nullptr};
struct BlockShared {
    std::mutex m;
    std::condition_variable cv;
    bool keccak_enqueued = false, keccak_failed = false;
    zk_session* keep = nullptr;  // the Bytecode assignment session (its rows are another chain's input)
    zk_session* keep_cpa = nullptr;  // the copy assignment session (the Copy circuit on chain 3 reads its rows)
    bool cpa_enqueued = false, cpa_failed = false;
    hipEvent_t ev_cpa = nullptr;
    hipEvent_t ev_keccak = nullptr;  // here: "the Bytecode assignment's rows are written"
    int rc[4] = {0, 0, 0, 0};
    std::string err[4];
    double end_ms[4] = {0, 0, 0, 0}, start_ms[4] = {0, 0, 0, 0};
    std::chrono::steady_clock::time_point t0;
};
#define BLK_TRY(expr) do { if ((rc = (expr))) goto done; } while (0)
static int block_run_pass(zk_session* s, zk_result* r) {
    int rc = zk_launch(s, nullptr);
    if (!rc) rc = zk_collect(s, r);
    return rc;
}
extern "C" int zk_block_verify(const zk_block* b, uint32_t opts, zk_result* results, double* chain_ms) {
    ARG_TRY(t_device >= 0, "zk_block_verify: call zk_init first");
    ARG_TRY(b && results && (opts & ZK_OPT_DEVICE_PTRS), "zk_block_verify: needs a block, a result array and ZK_OPT_DEVICE_PTRS");
    ARG_TRY(b->evm.steps && b->evm.n_steps >= 2 && b->evm.rw && b->evm.n_rw && b->randomness && b->hashed_offsets && b->n_hashed >= b->n_codes &&
            b->code_offsets && b->code_lengths && b->n_bytecodes, "zk_block_verify: bad arguments");
    HIP_TRY(hipSetDevice(t_device));
    const int device = t_device;
    const uint32_t dev_opts = ZK_OPT_DEVICE_PTRS, st_opts = ZK_OPT_DEVICE_PTRS | (opts & ZK_OPT_STATE_COMPACT);
    BlockShared sh;
    sh.t0 = std::chrono::steady_clock::now();
    BlockWorkers* W = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        { int brc = block_streams_ensure(device); if (brc) return brc; }
        if (!g_block_workers[device]) g_block_workers[device] = new BlockWorkers();
        W = g_block_workers[device];
    }
    std::lock_guard<std::mutex> one_block(W->call);
    if (!W->started) {
        for (int c = 0; c < 4; c++) std::thread([W, c] { W->run(c); }).detach();
        W->started = true;
    }
    for (int c = 0; c < ZK_BLOCK_NCIRCUITS; c++) { memset(&results[c], 0, sizeof(zk_result)); results[c].first_fail_row = UINT64_MAX; }
    // the small inputs every open wants on the host (HostView): the randomness cells and the bytecode offsets, fetched once by this
    // thread while the chains start — the State chain needs none of them and does not wait
    u64 h_cells[3][4];
    std::vector<u64> h_code_off((size_t)b->n_bytecodes + 1);
    HostView views[4];
    int n_views = 0;
    bool views_ready = false, views_failed = false;
    { int erc = arena_event(device, &sh.ev_keccak); if (erc) return erc; }
    { int erc = arena_event(device, &sh.ev_cpa); if (erc) return erc; }
    // Chains (each a host thread on its own stream):
    //   0  State:     RW table -> State verdict (rows evaluated where they are computed; or -> State witness -> State circuit)
    //   1  Bytecode:  keccak of the contracts (long messages: the slow pass) -> [Bytecode rows ready: event from chain 3] -> Bytecode circuit
    //   2  EVM:       copy assignment (+ event) -> keccak of the SHA3 inputs (its own short pass: the EVM circuit's keccak table does not
    //                 hold the contracts' hashes, so this chain never waits for chain 1) -> EVM open + pass
    //   3  rest:      Bytecode assignment (+ event) -> Exp -> Tx -> [copy rows ready: event from chain 2] -> Copy circuit
    auto enter = [&](int chain) {
        (void)zk_init(device);
        (void)zk_set_stream(g_block_stream[device][chain]);
        sh.start_ms[chain] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t0).count();
        if (chain == 0) return true;
        std::unique_lock<std::mutex> lock(sh.m);
        sh.cv.wait(lock, [&] { return views_ready; });
        for (int k = 0; k < n_views; k++) t_host_view[k] = views[k];
        t_n_host_view = n_views;
        return !views_failed;
    };
    auto leave = [&](int chain, int rc) {
        t_n_host_view = 0;
        sh.rc[chain] = rc;
        if (rc) sh.err[chain] = g_err;
        sh.end_ms[chain] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t0).count();
    };
    // ZK_BLOCK_TRACE=1: host-side marks of every chain to stderr (where a chain's host thread is when)
    static const bool trace = [] { const char* e = getenv("ZK_BLOCK_TRACE"); return e && e[0] == '1'; }();
    struct Mark { int chain; const char* what; double ms; };
    std::vector<Mark> marks[4];
    auto mark = [&](int chain, const char* what) {
        if (trace) marks[chain].push_back(Mark{chain, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t0).count()});
    };
    const u64* bca_rows = nullptr;  // (written by chain 3 before it sets keccak_enqueued — here: "bytecode rows enqueued")
    auto chain_state = [&] {
        enter(0);
        int rc = 0;
        zk_session *a = nullptr, *s = nullptr;
        uint64_t n_ops = 0, n_mpt = 0;
        zk_result ra;
        u32 n_mpt32 = 0;
        if (!(opts & (ZK_OPT_BLOCK_STATE_ROWS | ZK_OPT_STATE_COMPACT))) {  // rows evaluated where they are computed: one session, no witness
            BLK_TRY(zk_state_verify_from_rw_open(b->evm.rw, b->evm.rw_flags, b->evm.n_rw, dev_opts, &n_ops, &a));
            mark(0, "state opened (class scan + plan)");
            BLK_TRY(zk_launch(a, nullptr));
            mark(0, "state launched");
            BLK_TRY(zk_collect(a, &results[ZK_BLOCK_STATE]));
            mark(0, "state collected");
            goto done;
        }
        BLK_TRY(zk_state_assign_from_rw_open(b->evm.rw, b->evm.rw_flags, b->evm.n_rw, nullptr, nullptr, nullptr, st_opts, &n_ops, &a));
        BLK_TRY(zk_launch(a, nullptr));
        // the MPT row count rides on the collect's synchronisation (zk_state_assign_read would be two more round trips)
        if (hipMemcpyAsync(&n_mpt32, a->assign.blk_cnt + a->assign.nb, 4, hipMemcpyDeviceToHost, a->stream) != hipSuccess) { rc = -2; g_err = "zk_block_verify: copy failed"; goto done; }
        BLK_TRY(zk_collect(a, &ra));
        if (ra.fail_count) { rc = -1; g_err = "zk_block_verify: the State witness assignment failed (zk_state_assign_from_rw reports the op)"; goto done; }
        n_mpt = n_mpt32;
        BLK_TRY(zk_state_open(a->assign.rows, a->assign.row_flags, n_ops, a->assign.mpt, n_mpt, st_opts, &s));
        BLK_TRY(block_run_pass(s, &results[ZK_BLOCK_STATE]));
    done:
        if (s) zk_close(s);
        if (a) zk_close(a);
        leave(0, rc);
    };
    auto chain_bytecode = [&] {
        const bool entered = enter(1);
        int rc = 0;
        zk_session *ks = nullptr, *bs = nullptr;
        zk_result rk;
        if (!entered) { rc = -2; g_err = "zk_block_verify: reading the block's randomness / offsets failed"; goto done; }
        if (b->n_codes) {
            BLK_TRY(zk_keccak_open(b->hashed_data, b->hashed_bytes, b->hashed_offsets, b->n_codes, b->randomness, 0, nullptr, dev_opts, &ks));
            BLK_TRY(zk_launch(ks, nullptr));
            mark(1, "contracts' keccak launched");
        }
        {
            std::unique_lock<std::mutex> lock(sh.m);
            sh.cv.wait(lock, [&] { return sh.keccak_enqueued; });  // chain 3 has enqueued the Bytecode assignment (or failed)
            if (sh.keccak_failed) { rc = -1; g_err = "zk_block_verify: the Bytecode assignment failed before it was enqueued"; goto done; }
        }
        // the Bytecode circuit's session is opened behind the keccak pass on this stream (its index over the keccak rows is a kernel in
        // stream order); its evaluation pass also waits for the Bytecode rows of chain 3
        BLK_TRY(zk_bytecode_open(bca_rows, (uint64_t)1 << b->k, ks ? ks->keccak_gen.rows : nullptr, b->n_codes, b->randomness, dev_opts, &bs));
        if (hipStreamWaitEvent(g_block_stream[device][1], sh.ev_keccak, 0) != hipSuccess) { rc = -2; g_err = "zk_block_verify: stream wait failed"; goto done; }
        BLK_TRY(zk_launch(bs, nullptr));
        mark(1, "bytecode circuit launched");
        if (ks) {
            BLK_TRY(zk_collect(ks, &rk));
            if (rk.fail_count) { rc = -1; g_err = "zk_block_verify: a contract was rejected by the keccak table generation"; goto done; }
        }
        BLK_TRY(zk_collect(bs, &results[ZK_BLOCK_BYTECODE]));
    done:
        if (bs) zk_close(bs);
        if (ks) zk_close(ks);
        leave(1, rc);
    };
    auto chain_evm = [&] {
        const bool entered = enter(2);
        int rc = 0;
        zk_session *ca = nullptr, *es = nullptr, *ks = nullptr;
        zk_result rc_assign, rk;
        zk_evm_tables t = b->evm;
        t.copy = nullptr; t.n_copy = 0; t.keccak = nullptr; t.n_keccak = 0;
        bool cpa_signalled = false;
        if (!entered) { rc = -2; g_err = "zk_block_verify: reading the block's randomness / offsets failed"; goto done; }
        if (b->copy_events.n_events) {
            BLK_TRY(zk_copy_assign_open(&b->copy_events, nullptr, nullptr, nullptr, nullptr, nullptr, dev_opts, &ca));
            BLK_TRY(zk_launch(ca, nullptr));
            if (hipEventRecord(sh.ev_cpa, ca->stream) != hipSuccess) { rc = -2; g_err = "zk_block_verify: event record failed"; goto done; }
            sh.keep_cpa = ca;
            mark(2, "copy assignment launched");
            t.copy = ca->cpa.table;
            t.n_copy = ca->cpa_n_table;
        }
        { std::lock_guard<std::mutex> lock(sh.m); sh.cpa_enqueued = true; }
        sh.cv.notify_all();
        cpa_signalled = true;
        if (b->n_hashed > b->n_codes) {  // the SHA3 steps' inputs: the EVM circuit's keccak table (execution/sha3.py:31), on this chain's own stream
            BLK_TRY(zk_keccak_open(b->hashed_data, b->hashed_bytes, b->hashed_offsets + b->n_codes, b->n_hashed - b->n_codes, b->randomness, 0, nullptr, dev_opts, &ks));
            BLK_TRY(zk_launch(ks, nullptr));
            t.keccak = ks->keccak_gen.rows;
            t.n_keccak = b->n_hashed - b->n_codes;
            mark(2, "SHA3 keccak launched");
        }
        // the EVM session's build kernels and its pass queue up behind the two table generations: nothing is collected before everything
        // of this chain is enqueued
        BLK_TRY(zk_evm_open(&t, dev_opts | ZK_OPT_SINGLE_PASS | ZK_OPT_SIDE_STREAM, &es));
        mark(2, "evm opened");
        BLK_TRY(zk_launch(es, nullptr));
        mark(2, "evm launched");
        BLK_TRY(zk_collect(es, &results[ZK_BLOCK_EVM]));
        mark(2, "evm collected");
        if (ca) {
            BLK_TRY(zk_collect(ca, &rc_assign));
            if (rc_assign.fail_count) { rc = -1; g_err = "zk_block_verify: a copy event was rejected by the copy assignment"; goto done; }
        }
        if (ks) {
            BLK_TRY(zk_collect(ks, &rk));
            if (rk.fail_count) { rc = -1; g_err = "zk_block_verify: a SHA3 input was rejected by the keccak table generation"; goto done; }
        }
    done:
        if (!cpa_signalled) {
            { std::lock_guard<std::mutex> lock(sh.m); sh.cpa_failed = true; sh.cpa_enqueued = true; }
            sh.cv.notify_all();
        }
        if (es) zk_close(es);
        if (ks) zk_close(ks);
        // (the copy assignment's rows are read by chain 3's Copy circuit: closed after the chains have ended)
        leave(2, rc);
    };
    auto chain_rest = [&] {
        const bool entered = enter(3);
        int rc = 0;
        zk_session *ex = nullptr, *tx = nullptr, *ba = nullptr, *cs = nullptr;
        zk_result rb;
        bool signalled = false;
        if (!entered) { rc = -2; g_err = "zk_block_verify: reading the block's randomness / offsets failed"; goto done; }
        BLK_TRY(zk_bytecode_assign_open(b->evm.bytecode, b->evm.n_bytecode, b->code_offsets, b->code_lengths, b->n_bytecodes, b->k, b->randomness, nullptr, dev_opts, &ba));
        BLK_TRY(zk_launch(ba, nullptr));
        if (hipEventRecord(sh.ev_keccak, ba->stream) != hipSuccess) { rc = -2; g_err = "zk_block_verify: event record failed"; goto done; }
        bca_rows = ba->bca.rows;
        { std::lock_guard<std::mutex> lock(sh.m); sh.keccak_enqueued = true; }
        sh.cv.notify_all();
        signalled = true;
        mark(3, "bytecode assignment launched");
        if (b->n_exp_rows) {
            BLK_TRY(zk_exp_open(b->exp_rows, b->n_exp_rows, dev_opts, &ex));
            BLK_TRY(zk_launch(ex, nullptr));
        }
        if (b->tx.n_units) {
            BLK_TRY(zk_sign_open(&b->tx, dev_opts, &tx));
            BLK_TRY(zk_launch(tx, nullptr));
        }
        mark(3, "exp + tx launched");
        {   // the Copy circuit over the rows chain 2's copy assignment writes: its lookup indices are built right away, its pass is ordered
            // behind the assignment on the device
            std::unique_lock<std::mutex> lock(sh.m);
            sh.cv.wait(lock, [&] { return sh.cpa_enqueued; });
            if (sh.cpa_failed) { rc = -1; g_err = "zk_block_verify: the copy assignment failed before it was enqueued"; goto done; }
        }
        if (sh.keep_cpa && sh.keep_cpa->cpa.n_rows) {
            zk_session* ca = sh.keep_cpa;
            zk_copy_tables ct;
            memset(&ct, 0, sizeof ct);
            ct.rows = ca->cpa.rows; ct.row_flags = ca->cpa.row_flags; ct.n_rows = ca->cpa.n_rows; ct.randomness = b->copy_events.randomness;
            ct.rw = b->evm.rw; ct.rw_flags = b->evm.rw_flags; ct.n_rw = b->evm.n_rw;
            ct.bytecode = b->evm.bytecode; ct.n_bytecode = b->evm.n_bytecode;
            ct.tx = b->evm.tx; ct.tx_flags = b->evm.tx_flags; ct.n_tx = b->evm.n_tx;
            BLK_TRY(zk_copy_open(&ct, dev_opts, &cs));
            if (hipStreamWaitEvent(g_block_stream[device][3], sh.ev_cpa, 0) != hipSuccess) { rc = -2; g_err = "zk_block_verify: stream wait failed"; goto done; }
            BLK_TRY(zk_launch(cs, nullptr));
            mark(3, "copy circuit launched");
        }
        if (tx) BLK_TRY(zk_collect(tx, &results[ZK_BLOCK_TX]));
        if (ex) BLK_TRY(zk_collect(ex, &results[ZK_BLOCK_EXP]));
        BLK_TRY(zk_collect(ba, &rb));
        if (rb.fail_count) { rc = -1; g_err = "zk_block_verify: the Bytecode assignment rejected its input"; goto done; }
        if (cs) BLK_TRY(zk_collect(cs, &results[ZK_BLOCK_COPY]));
    done:
        if (cs) zk_close(cs);
        if (!signalled) {
            { std::lock_guard<std::mutex> lock(sh.m); sh.keccak_failed = true; sh.keccak_enqueued = true; }
            sh.cv.notify_all();
        }
        if (tx) zk_close(tx);
        if (ex) zk_close(ex);
        // (the Bytecode assignment's rows are read by chain 1's Bytecode circuit: closed after the chains have ended)
        sh.keep = ba;
        leave(3, rc);
    };
    {
        std::unique_lock<std::mutex> lock(W->m);
        W->job[0] = chain_state; W->job[1] = chain_bytecode; W->job[2] = chain_evm; W->job[3] = chain_rest;
        for (int c = 0; c < 4; c++) W->has_job[c] = true;
        W->pending = 4;
        W->cv_job.notify_all();
    }
    {
        const hipStream_t st = t_stream ? t_stream : g_own_stream[device];
        const uint64_t* cells[3] = {b->randomness, b->copy_events.n_events ? b->copy_events.randomness : nullptr, b->tx.n_units ? b->tx.randomness : nullptr};
        bool ok = true;
        for (int k = 0; k < 3 && ok; k++) {
            if (!cells[k]) continue;
            bool seen = false;
            for (int j = 0; j < n_views; j++) seen = seen || views[j].dev == cells[k];
            if (seen) continue;
            ok = hipMemcpyAsync(h_cells[k], cells[k], 32, hipMemcpyDeviceToHost, st) == hipSuccess;
            views[n_views++] = HostView{cells[k], h_cells[k], 32};
        }
        ok = ok && hipMemcpyAsync(h_code_off.data(), b->code_offsets, h_code_off.size() * 8, hipMemcpyDeviceToHost, st) == hipSuccess;
        views[n_views++] = HostView{b->code_offsets, h_code_off.data(), h_code_off.size() * 8};
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        { std::lock_guard<std::mutex> lock(sh.m); views_failed = !ok; views_ready = true; }
        sh.cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lock(W->m);
        W->cv_done.wait(lock, [&] { return W->pending == 0; });
    }
    const double joined_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t0).count();
    if (sh.keep) zk_close(sh.keep);
    if (sh.keep_cpa) zk_close(sh.keep_cpa);
    {
        DevArena& A = g_arena[device];
        std::lock_guard<std::mutex> lock(A.m);
        A.events.push_back(sh.ev_keccak);
        A.events.push_back(sh.ev_cpa);
    }
    if (trace)
        for (int c = 0; c < 4; c++) {
            fprintf(stderr, "[zk_block_verify] chain %d: start %.3f", c, sh.start_ms[c]);
            for (const Mark& k : marks[c]) fprintf(stderr, " | %s %.3f", k.what, k.ms);
            fprintf(stderr, " | end %.3f\n", sh.end_ms[c]);
        }
    if (chain_ms) {
        for (int c = 0; c < 4; c++) { chain_ms[c] = sh.end_ms[c]; chain_ms[4 + c] = sh.start_ms[c]; }
        chain_ms[8] = joined_ms;
        chain_ms[9] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t0).count();
    }
    for (int c = 0; c < 4; c++)
        if (sh.rc[c]) { g_err = sh.err[c]; return sh.rc[c]; }
    return 0;
}
