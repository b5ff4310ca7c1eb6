// Multi-GPU tally (include/zkevm_hip.h "Multi-GPU tally"): what both backends share — the binding of the
// collective library and the packing / reduction of the per-rank tally words.  The HIP library stages the words through device
// memory and calls RCCL on its own stream; the CPU backend hands the collective library host buffers (only a library named
// by ZK_RCCL_LIB can take those: RCCL itself wants device memory).
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/zkevm_hip.h"

namespace zkdist {
struct RcclId { char internal[ZK_DIST_ID_BYTES]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, void* /* hipStream_t */) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    bool named_by_env = false;  // ZK_RCCL_LIB chose the library (the integrator's own build of RCCL, or a host-buffer stand-in)
    int buffers = 1;            // what its ncclAllGather takes: 1 device memory (RCCL; any library without the marker), 0 host memory
                                // (a stand-in that exports `int zk_collective_buffers(void)` returning 0: tests/fakerccl)
};
static const int RCCL_UINT64 = 5;  // ncclUint64 (rccl.h ncclDataType_t)
// librccl is bound at first use: nothing else in the library needs it, and torch — when it is in the process — has usually
// loaded the same SONAME already.  ZK_RCCL_LIB (a path) is tried first.
inline RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        if (const char* e = getenv("ZK_RCCL_LIB")) {
            if (*e && (h = dlopen(e, RTLD_NOW | RTLD_GLOBAL))) a.named_by_env = true;
        }
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return a;
        a.GetUniqueId = (int (*)(RcclId*))dlsym(h, "ncclGetUniqueId");
        a.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(h, "ncclCommInitRank");
        a.AllGather = (int (*)(const void*, void*, size_t, int, void*, void*))dlsym(h, "ncclAllGather");
        a.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy;
        if (auto marker = (int (*)(void))dlsym(h, "zk_collective_buffers")) a.buffers = marker() ? 1 : 0;
        return a;
    }();
    return api;
}

// per rank: fail count | first failing GLOBAL row (UINT64_MAX: none) | its code | rows evaluated | kernel_ms (the double's bits)
static const int TALLY_WORDS = 5;
inline void tally_pack(uint64_t* mine, const zk_result* local, uint64_t row_offset) {
    mine[0] = local->fail_count;
    mine[1] = local->first_fail_row == UINT64_MAX ? UINT64_MAX : local->first_fail_row + row_offset;
    mine[2] = local->first_fail_row == UINT64_MAX ? 0u : (uint64_t)local->first_fail_code;
    mine[3] = local->rows_evaluated;
    memcpy(&mine[4], &local->kernel_ms, 8);
}
// SUM of the counts, the smallest failing global row with its code (global rows of different ranks are distinct: no tie to
// break), MAX of the kernel times — identically on every rank
inline void tally_reduce(const uint64_t* gathered, int world, const zk_result* local, zk_result* global) {
    *global = *local;
    uint64_t total = 0, rows = 0, row = UINT64_MAX, code = 0;
    double kmax = 0.0;
    for (int r = 0; r < world; r++) {
        const uint64_t* w = gathered + (size_t)r * TALLY_WORDS;
        total += w[0];
        rows += w[3];
        if (w[1] < row) { row = w[1]; code = w[2]; }
        double k;
        memcpy(&k, &w[4], 8);
        if (k > kmax) kmax = k;
    }
    global->rows_evaluated = rows;
    global->fail_count = total;
    global->first_fail_row = row;
    global->first_fail_code = row == UINT64_MAX ? 0u : (uint32_t)code;
    global->kernel_ms = kmax;
}
}  // namespace zkdist
