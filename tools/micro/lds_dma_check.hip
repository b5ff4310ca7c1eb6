// Does global_load_lds_dwordx4 land lane t's 16 bytes at base + 16 t, and do counted vmcnt waits order LDS-DMA pieces?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
template <int WAIT>
__global__ __launch_bounds__(64) void probe(const uint4* src, u32* out) {
    extern __shared__ uint4 lds[];
    const u32 lane = threadIdx.x;
    // 8 pieces of 1 KiB; piece p reads src[p * 64 + lane]
#pragma unroll
    for (int p = 0; p < 8; p++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)blockIdx.x * 512 + p * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(lds + p * 64), 16, 0, 0);
    // wait for piece 0..3 only (4 pieces issued after them) or for all
    if (WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    const u32 addr = (u32)(size_t)(__attribute__((address_space(3))) void*)lds + lane * 16;
    u32 bad = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        u32x4 v;
        if (p == 0) asm volatile("ds_read_b128 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        if (p == 1) asm volatile("ds_read_b128 %0, %1 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        if (p == 2) asm volatile("ds_read_b128 %0, %1 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        if (p == 3) asm volatile("ds_read_b128 %0, %1 offset:3072\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        const u32 e = (blockIdx.x * 512 + p * 64 + lane) * 4;
        if (v.x != e || v.y != e + 1 || v.z != e + 2 || v.w != e + 3) bad |= 1u << p;
    }
    if (bad) atomicAdd(out + (WAIT ? 1 : 0), 1);
    if (bad && blockIdx.x == 0 && lane < 4) printf("WAIT=%d lane %u bad mask %x\n", WAIT, lane, bad);
}
int main() {
    const size_t n = 4096ull * 512;  // uint4s
    uint4* src; u32* out;
    CHECK(hipMalloc(&src, n * 16)); CHECK(hipMalloc(&out, 8)); CHECK(hipMemset(out, 0, 8));
    u32* h = (u32*)malloc(n * 16);
    for (size_t i = 0; i < n * 4; i++) h[i] = (u32)i;
    CHECK(hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(probe<0>), dim3(4096), dim3(64), 8192, 0, src, out);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(probe<1>), dim3(4096), dim3(64), 8192, 0, src, out);
    CHECK(hipDeviceSynchronize());
    u32 r[2]; CHECK(hipMemcpy(r, out, 8, hipMemcpyDeviceToHost));
    printf("lanes with wrong data: vmcnt(0) %u, vmcnt(4) %u (of %u)\n", r[0], r[1], 4096 * 64);
    return 0;
}
