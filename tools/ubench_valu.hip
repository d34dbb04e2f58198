// ubench_valu.hip -- empirical integer-VALU ceilings on gfx950 for the Hamming
// kernel's instruction mix: how many lane-operations per second the chip
// sustains for (a) v_xor_b32 + accumulating v_bcnt_u32_b32 pairs and (b) the
// full 19-op comparison body (8 xor, 8 bcnt, key pack, min, med3).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_xor_bcnt(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t q[8], acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) q[i] = seed * (threadIdx.x + 1) + i * 0x9e3779b9u;
    uint32_t t = seed ^ blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t x = q[i] ^ (t + r);
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[r]) : "v"(x));
            }
        }
        t = t * 1664525u + 1013904223u; // scalar
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void k_cmp_body(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t q[8], b1 = ~0u, b2 = ~0u;
    for (int i = 0; i < 8; ++i) q[i] = seed * (threadIdx.x + 1) + i * 0x9e3779b9u;
    uint32_t t = seed ^ blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t d = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t x = q[i] ^ (t + r * 77 + i);
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
            }
            uint32_t key = (d << 23) | (uint32_t)(it * 4 + r);
            uint32_t nb2;
            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(nb2) : "v"(key), "v"(b1), "v"(b2));
            b1 = min(key, b1);
            b2 = nb2;
        }
        t = t * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = b1 ^ b2;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int iters = 4096;
    uint32_t* out;
    CHK(hipMalloc(&out, (size_t)p.multiProcessorCount * 16 * 256 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int bpc = 1; bpc <= 16; bpc *= 2)
    for (int which = 0; which < 2; ++which) {
        const int blocks = p.multiProcessorCount * bpc;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_xor_bcnt, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u + rep);
            else hipLaunchKernelGGL(k_cmp_body, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u + rep);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double lanes = (double)blocks * 256;
        if (which == 0) {
            double ops = lanes * iters * 4 * 16; // 8 xor + 8 bcnt per row
            printf("[%2d waves/SIMD] xor+bcnt: %.3f ms, %.2f T lane-ops/s, = %.1f G cmp/s at 16 ops/cmp\n", bpc, best,
                   ops / best / 1e9, ops / 16 / best / 1e6);
        } else {
            double cmps = lanes * iters * 4;
            printf("[%2d waves/SIMD] cmp body (19 VALU ops): %.3f ms, %.1f G cmp/s, %.2f T lane-ops/s\n", bpc, best,
                   cmps / best / 1e6, cmps * 19 / best / 1e9);
        }
    }
    return 0;
}
