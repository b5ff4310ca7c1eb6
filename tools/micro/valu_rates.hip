// Issue-rate micro-benchmark for the integer VALU instructions the modular arithmetic is made of (gfx950).
// One block of 64 / 128 / 256 threads per CU-sized grid; every lane runs N independent or dependent ops between two
// s_memtime reads; prints shader cycles per instruction per wavefront.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned int u32;
#define N_ITER 512

template <int KIND, int CHAINS>
__global__ void k(u64* out, u32 seed) {
    u32 a[CHAINS], b = seed | 1u;
    u64 c[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; c[i] = a[i]; }
    const u64 t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (KIND == 0) c[i] = (u64)a[i] * b + c[i];                          // v_mad_u64_u32
            if (KIND == 1) a[i] = a[i] * b + (u32)c[i];                          // v_mul_lo_u32 + add (or v_mad_u32_u24?)
            if (KIND == 2) a[i] = __umulhi(a[i], b) + 1u;                        // v_mul_hi_u32
            if (KIND == 3) { c[i] += a[i]; }                                     // 64-bit add: v_add_co + v_addc
            if (KIND == 4) a[i] = (a[i] ^ b) + (a[i] >> 3);                      // plain 32-bit VALU
            if (KIND == 5) { c[i] = (u64)a[i] * b + (c[i] >> 32) + (u32)c[i]; }  // mad + carry-style adds (CIOS inner step)
        }
    }
    const u64 t1 = clock64();
    u64 acc = 0;
    for (int i = 0; i < CHAINS; i++) acc += c[i] + a[i];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (acc == 0x1234567) out[blockIdx.x * 2 + 1] = acc;
}

template <int KIND, int CHAINS>
void run(const char* name, int threads) {
    u64* d;
    hipMalloc(&d, 1024 * 16);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<KIND, CHAINS>), dim3(256), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    u64 h[512];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; i++) s += (double)h[2 * i];
    s /= 256;
    printf("%-34s chains=%d threads/block=%3d (waves/SIMD=%.2f): %.2f cycles per instruction-group per wave\n", name, CHAINS, threads, threads / 256.0,
           s / ((double)N_ITER * CHAINS));
    hipFree(d);
}
int main() {
    for (int threads : {64, 256, 512}) {
        run<0, 1>("v_mad_u64_u32 dependent", threads);
        run<0, 8>("v_mad_u64_u32 8 independent", threads);
        run<1, 8>("mul_lo+add 8 independent", threads);
        run<2, 8>("mul_hi+add 8 independent", threads);
        run<3, 8>("add64 8 independent", threads);
        run<4, 8>("xor/shift/add 8 independent", threads);
        run<5, 8>("mad64 + split-carry adds 8 indep", threads);
    }
    return 0;
}
