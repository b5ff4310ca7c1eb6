// Micro-benchmark (tuning aid, not product code): how fast can the RW table's key cells be streamed for the packed-key build
// of zk_evm_open?  Rows are 448 B (14 cells); the six key cells are the first 192 B.  Variants differ in how the loads are
// spread over lanes / how many are in flight per lane.  hipcc --offload-arch=gfx950 -O3 -o rw_pack rw_pack.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
#define ROW_U4 28  // uint4 per row

__device__ __forceinline__ uint4 orr(uint4 a, uint4 b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }

// v0: one quarter-row per thread (the product's form): lane q of a quad loads chunks q, q+4, q+8
__global__ __launch_bounds__(1024) void v0(const uint4* rows, u32 n, uint4* keys) {
    const u32 vt = blockIdx.x * blockDim.x + threadIdx.x, r = vt >> 2, q = vt & 3u;
    if (r >= n) return;
    const uint4* p = rows + (u64)r * ROW_U4 + q;
    uint4 c = orr(orr(p[0], p[4]), p[8]);
    c.x |= __shfl_xor(c.x, 1); c.y |= __shfl_xor(c.y, 2);
    if (q < 2) keys[(u64)r * 2 + q] = c;
}
// v1: grid-stride, R rows per quad per iteration (3R loads in flight per lane)
template <int R>
__global__ __launch_bounds__(256) void v1(const uint4* rows, u32 n, uint4* keys) {
    const u32 quads = gridDim.x * blockDim.x / 4;
    const u32 q = threadIdx.x & 3u;
    for (u32 r0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; r0 < n; r0 += quads * R) {
        uint4 c[R][3];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const u32 r = r0 + k * quads;
            const uint4* p = rows + (u64)(r < n ? r : 0) * ROW_U4 + q;
            c[k][0] = p[0]; c[k][1] = p[4]; c[k][2] = p[8];
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const u32 r = r0 + k * quads;
            uint4 x = orr(orr(c[k][0], c[k][1]), c[k][2]);
            x.x |= __shfl_xor(x.x, 1); x.y |= __shfl_xor(x.y, 2);
            if (r < n && q < 2) keys[(u64)r * 2 + q] = x;
        }
    }
}
// v2: the wavefront walks the key bytes of 16 consecutive rows as 16 x 12 chunks: lane l takes chunk (l % 12) of row (l / 12)...
// simpler: 12 lanes per row (5 rows per wavefront, 4 lanes idle), one 16-byte load per lane: every load instruction covers
// 5 x 192 contiguous-per-row bytes
__global__ __launch_bounds__(256) void v2(const uint4* rows, u32 n, uint4* keys) {
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const u32 sub = lane / 12u, ch = lane % 12u;
    const u32 r = wave * 5u + sub;
    uint4 c = make_uint4(0, 0, 0, 0);
    if (sub < 5u && r < n) c = rows[(u64)r * ROW_U4 + ch];
    c.x |= __shfl_xor(c.x, 1); c.y |= __shfl_xor(c.y, 2); c.z |= __shfl_xor(c.z, 4);
    if (sub < 5u && r < n && ch < 2) keys[(u64)r * 2 + ch] = c;
}
// v3: whole rows (all 28 chunks): 4 lanes per row, 7 chunks each — what a record that also carries the value cells would read
__global__ __launch_bounds__(256) void v3(const uint4* rows, u32 n, uint4* keys) {
    const u32 vt = blockIdx.x * blockDim.x + threadIdx.x, r = vt >> 2, q = vt & 3u;
    if (r >= n) return;
    const uint4* p = rows + (u64)r * ROW_U4 + q;
    uint4 c = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 7; k++) c = orr(c, p[4 * k]);
    c.x |= __shfl_xor(c.x, 1); c.y |= __shfl_xor(c.y, 2);
    keys[(u64)r * 4 + q] = c;  // 64-byte record
}
// v4: whole rows, fully linear: thread t loads uint4 t, t + T, ... (pure streaming reference; no per-row result)
__global__ __launch_bounds__(256) void v4(const uint4* rows, u64 n_u4, uint4* keys) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    uint4 c = make_uint4(0, 0, 0, 0);
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n_u4; i += 4 * stride) {
        uint4 a = rows[i], b = rows[i + stride], d = rows[i + 2 * stride], e = rows[i + 3 * stride];
        c = orr(c, orr(orr(a, b), orr(d, e)));
    }
    for (; i < n_u4; i += stride) c = orr(c, rows[i]);
    if ((c.x | c.y | c.z | c.w) == 0x12345u) keys[0] = c;
}
// v5: key cells only, linear over (row, chunk<12): thread walks chunk indices 0..12n, address = row*28 + chunk: coalesced within a row's 192 B
__global__ __launch_bounds__(256) void v5(const uint4* rows, u32 n, uint4* keys) {
    const u64 total = (u64)n * 12, stride = (u64)gridDim.x * blockDim.x;
    uint4 c = make_uint4(0, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const u64 r = i / 12, ch = i % 12;
        c = orr(c, rows[r * ROW_U4 + ch]);
    }
    if ((c.x | c.y | c.z | c.w) == 0x12345u) keys[0] = c;
}
// read-only sweep over 2 GiB of unrelated data (a write sweep leaves dirty lines whose write-back then competes with the measured kernel)
__global__ void flush_k(const uint4* p, u64 n, uint4* sink) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    uint4 c = make_uint4(0, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c = orr(c, p[i]);
    if ((c.x | c.y | c.z | c.w) == 0x12345u) sink[0] = c;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const u32 n = 816377;
    const u64 bytes = (u64)n * 448;
    uint4 *rows, *keys, *fl;
    CK(hipMalloc(&rows, bytes)); CK(hipMalloc(&keys, (u64)n * 64)); CK(hipMalloc(&fl, 2ull << 30)); CK(hipMemset(fl, 3, 2ull << 30));
    CK(hipMemset(rows, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double mb) {
        float best = 1e9, sum = 0;
        for (int it = 0; it < 6; it++) {
            hipLaunchKernelGGL(flush_k, dim3(4096), dim3(256), 0, 0, fl, (u64)(2ull << 30) / 16, keys);
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-28s best %7.1f us  mean %7.1f us   %6.2f TB/s of %.0f MB touched (best)\n", name, best * 1e3, sum / 5 * 1e3, mb / 1e6 / best * 1e3 / 1e3, mb);
    };
    const double key_lines = (double)n * 256, all = (double)bytes;
    run("v0 quarter-row/thread", [&] { hipLaunchKernelGGL(v0, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, rows, n, keys); }, key_lines);
    run("v0 (blocks of 1024)", [&] { hipLaunchKernelGGL(v0, dim3((n * 4 + 1023) / 1024), dim3(1024), 0, 0, rows, n, keys); }, key_lines);
    run("v0 (blocks of 64)", [&] { hipLaunchKernelGGL(v0, dim3((n * 4 + 63) / 64), dim3(64), 0, 0, rows, n, keys); }, key_lines);
    for (int g : {1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "v1<2> grid %d", g); run(nm, [&] { hipLaunchKernelGGL(v1<2>, dim3(g), dim3(256), 0, 0, rows, n, keys); }, key_lines);
        snprintf(nm, 64, "v1<4> grid %d", g); run(nm, [&] { hipLaunchKernelGGL(v1<4>, dim3(g), dim3(256), 0, 0, rows, n, keys); }, key_lines);
    }
    run("v1<8> grid 2048", [&] { hipLaunchKernelGGL(v1<8>, dim3(2048), dim3(256), 0, 0, rows, n, keys); }, key_lines);
    run("v2 12 lanes/row", [&] { hipLaunchKernelGGL(v2, dim3((((n + 4) / 5) * 64 + 255) / 256), dim3(256), 0, 0, rows, n, keys); }, key_lines);
    run("v3 whole rows 4 lanes/row", [&] { hipLaunchKernelGGL(v3, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, rows, n, keys); }, all);
    run("v4 linear whole table", [&] { hipLaunchKernelGGL(v4, dim3(4096), dim3(256), 0, 0, rows, bytes / 16, keys); }, all);
    run("v5 linear key chunks", [&] { hipLaunchKernelGGL(v5, dim3(4096), dim3(256), 0, 0, rows, n, keys); }, key_lines);
    return 0;
}
