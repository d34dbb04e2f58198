// ubench_mfma.hip -- cycles per v_mfma_i32_32x32x32_i8 on gfx950: four independent
// accumulator chains per wave, one / two / three waves per SIMD, every CU busy.
// Prints the time per MFMA per SIMD in ns and in cycles of the wall clock
// (s_memrealtime runs at 100 MHz; the shader clock is read from the device
// properties), and the chip's int8 rate that follows.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o ubench_mfma ubench_mfma.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// mode 0: constant small operands; mode 1: operands as the match kernel has them,
// random train bits as {0, 64} bytes and random query bits as {+64, -64} bytes,
// eight different register sets used in turn; mode 2: the same with {0, 1} and {+1, -1}
__global__ __launch_bounds__(256) void k_mfma(int* out, int iters, int seed, int mode)
{
    v4i a[8], b[8];
    uint32_t x = (uint32_t)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u + 1u;
    for (int k = 0; k < 8; ++k)
        for (int n = 0; n < 4; ++n) {
            x = x * 1664525u + 1013904223u;
            uint32_t bits = (x >> 8) & 0x01010101u; // one random bit per byte
            uint32_t y = x * 22695477u + 1u;
            uint32_t qb = (y >> 9) & 0x01010101u;
            if (mode == 0) { a[k][n] = seed + n; b[k][n] = (int)threadIdx.x + n; }
            else if (mode == 1) { a[k][n] = (int)(bits << 6); b[k][n] = (int)((((qb << 8) - qb) & 0x80808080u) | 0x40404040u); }
            else { a[k][n] = (int)bits; b[k][n] = (int)(((qb << 8) - qb) | 0x01010101u); }
        }
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[(k + 3) & 7], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k + 1], b[k + 1], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k + 1], b[(k + 4) & 7], c3, 0, 0, 0);
        }
    }
    int s = 0;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Two accumulator chains (as the match kernel has) with NV independent vector
// operations (min / med3 on registers the MFMAs do not touch) after every MFMA:
// how much vector work the issue port takes before the matrix pipe starves.
template <int NV>
__global__ __launch_bounds__(256) void k_mfma_valu(int* out, int iters, int seed)
{
    v4i a[8], b[8];
    uint32_t x = (uint32_t)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u + 1u;
    for (int k = 0; k < 8; ++k)
        for (int n = 0; n < 4; ++n) {
            x = x * 1664525u + 1013904223u;
            a[k][n] = (int)(((x >> 8) & 0x01010101u) << 6);
            uint32_t qb = ((x * 22695477u + 1u) >> 9) & 0x01010101u;
            b[k][n] = (int)((((qb << 8) - qb) & 0x80808080u) | 0x40404040u);
        }
    v16i c0 = {0}, c1 = {0};
    uint32_t m1 = x, m2 = x * 3u, w = x * 7u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[k], c0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                w = w * 1u + (uint32_t)v; // folded to an add
                uint32_t lo = m1 < w ? m1 : w, hi = m1 < w ? w : m1;
                m2 = hi < m2 ? hi : m2;
                m1 = lo;
            }
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[(k + 3) & 7], c1, 0, 0, 0);
        }
    }
    int s = (int)(m1 ^ m2);
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
static int run_valu(int* out, int cus, int iters)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 3; ++wps) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_mfma_valu<NV>, dim3(cus * wps), dim3(256), 0, 0, out, iters, rep);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double mfma_per_simd = (double)iters * 16 * wps;
        printf("2 chains, %d x (add, cmp-based min/max: about 4 VALU) per pair of MFMAs, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD = %.1f cycles at 2400 MHz\n",
               NV, wps, best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.4);
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int iters = 40000;
    int* out;
    CHK(hipMalloc(&out, (size_t)p.multiProcessorCount * 4 * 256 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode)
    for (int wps = 1; wps <= 3; ++wps) { // workgroups of 4 waves per CU = waves per SIMD
        const int blocks = p.multiProcessorCount * wps;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, rep, mode);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double mfma_per_simd = (double)iters * 8 * wps; // 16 MFMAs per two iterations
        const double ns = best * 1e6 / mfma_per_simd;
        const double tops = 2.0 * 32 * 32 * 32 * mfma_per_simd * p.multiProcessorCount * 4 / (best * 1e-3) / 1e12;
        printf("mode %d, %d wave(s) per SIMD: %.3f ms, %.2f ns per MFMA per SIMD = %.1f cycles at %d MHz, %.0f TOP/s\n", mode, wps,
               best, ns, ns * p.clockRate / 1e6, p.clockRate / 1000, tops);
    }
    if (run_valu<0>(out, p.multiProcessorCount, 5000)) return 1;
    if (run_valu<2>(out, p.multiProcessorCount, 5000)) return 1;
    if (run_valu<4>(out, p.multiProcessorCount, 5000)) return 1;
    if (run_valu<6>(out, p.multiProcessorCount, 5000)) return 1;
    return 0;
}
