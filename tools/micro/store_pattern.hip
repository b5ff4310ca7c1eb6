// Micro-benchmark: how should a lane-per-row kernel write column-major 32-byte cells?  (assign_rows_kernel: 57 cells x n rows)
//   A  every lane stores its own 32 bytes as two 16-byte stores (what asg_store does: each store instruction covers 2 KB half-filled)
//   B  the wavefront's 64 x 32 bytes of a cell go out as two fully contiguous 1 KB store instructions (lane l: bytes [16 l, 16 l + 16)
//      of each KB) — the data a lane stores is synthetic here (the LDS transpose a real kernel would need is not part of the timing)
//   C  like B but with the transpose through LDS included
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run: ./store_pattern [log2 rows = 20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define NC 57
__global__ __launch_bounds__(256) void kA(uint4* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 lo = {(unsigned)i, 1, 2, 3}, hi = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NC; c++) {
        uint4* p = out + ((size_t)c * n + i) * 2;
        uint4 v = lo; v.y = c;
        p[0] = v;
        p[1] = hi;
    }
}
__global__ __launch_bounds__(256) void kB(uint4* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave0 = i - lane;
    const uint4 lo = {(unsigned)i, 1, 2, 3};
#pragma unroll
    for (int c = 0; c < NC; c++) {
        uint4* p = out + ((size_t)c * n + wave0) * 2;  // the wavefront's 2 KB of this cell
        uint4 v = lo; v.y = c;
        p[lane] = v;
        p[64 + lane] = v;
    }
}
__global__ __launch_bounds__(256) void kC(uint4* out, size_t n) {
    __shared__ uint4 s[4][128];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const size_t wave0 = i - lane;
    const uint4 lo = {(unsigned)i, 1, 2, 3}, hi = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NC; c++) {
        uint4 v = lo; v.y = c;
        s[w][2 * lane] = v;
        s[w][2 * lane + 1] = hi;
        __builtin_amdgcn_wave_barrier();
        const uint4 a = s[w][lane], b = s[w][64 + lane];
        __builtin_amdgcn_wave_barrier();
        uint4* p = out + ((size_t)c * n + wave0) * 2;
        p[lane] = a;
        p[64 + lane] = b;
    }
}
int main(int argc, char** argv) {
    const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 20);
    uint4* out;
    hipMalloc(&out, n * NC * 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 3; k++) {
        const char* name = k == 0 ? "A two 16-B stores per lane, 32-B stride" : k == 1 ? "B contiguous 1 KB per store instruction" : "C = B + LDS transpose";
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (k == 0) kA<<<n / 256, 256>>>(out, n);
            else if (k == 1) kB<<<n / 256, 256>>>(out, n);
            else kC<<<n / 256, 256>>>(out, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("%-45s %.3f ms  %.2f TB/s\n", name, best, n * NC * 32.0 / best / 1e9);
    }
    return 0;
}
