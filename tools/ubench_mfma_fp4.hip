// ubench_mfma_fp4.hip -- Hamming distances from v_mfma_scale_f32_32x32x64_f8f6f4 with
// FP4 (E2M1) operands on gfx950: (1) exactness and result layout against popcount
// on the host, with block scales that multiply every product by 8192; (2) cycles
// per MFMA per SIMD for chains of this instruction on random bits, alone and with
// float min / med3 fillers as a top-2 update has them.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o ubench_mfma_fp4 ubench_mfma_fp4.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 8 bits -> 8 FP4 nibbles, bit i in nibble i: set -> hi, clear -> lo
__host__ __device__ static inline uint32_t expand8(uint32_t byte, uint32_t lo, uint32_t hi)
{
    uint32_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (((byte >> i) & 1u) ? hi : lo) << (4 * i);
    return v;
}

#define SCALE_A 131 /* 2^4 */
#define SCALE_B 132 /* 2^5 */

// one wave: 32 train rows x 32 query rows of 64 bits each
__global__ void k_check(const uint32_t* train, const uint32_t* query, float* out)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    // lane half h takes dword h of the 64-bit rows: 32 bits -> 32 nibbles
    const uint32_t tw = train[r * 2 + h], qw = query[r * 2 + h];
    v8i a = {0}, b = {0};
    for (int n = 0; n < 4; ++n) {
        a[n] = (int)expand8((tw >> (8 * n)) & 0xffu, 0x0u, 0x6u); // 0 / +4
        b[n] = (int)expand8((qw >> (8 * n)) & 0xffu, 0x6u, 0xEu); // +4 / -4  (b' = 1 - 2b)
    }
    v16f c;
    for (int j = 0; j < 16; ++j) c[j] = (float)(1000 + (j & 3) + 8 * (j >> 2) + 4 * h);
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, SCALE_A, 0, SCALE_B);
    for (int j = 0; j < 16; ++j) out[lane * 16 + j] = c[j];
}

typedef int v4i __attribute__((ext_vector_type(4)));
// FP4 operands are 4 registers; the builtin takes 8, of which the upper 4 are not read
#define WIDE(x) __builtin_shufflevector(x, x, 0, 1, 2, 3, -1, -1, -1, -1)

// the unscaled form (both scales the constant 0 select v_mfma_f32_32x32x64_f8f6f4)
template <int NV>
__global__ __launch_bounds__(256) void k_rate_unscaled(float* out, int iters, int seed)
{
    v4i a[4], b[4];
    uint32_t x = (uint32_t)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u + 1u;
    for (int k = 0; k < 4; ++k)
        for (int n = 0; n < 4; ++n) {
            x = x * 1664525u + 1013904223u;
            a[k][n] = (int)expand8((x >> 11) & 0xffu, 0x0u, 0x6u);
            b[k][n] = (int)expand8((x >> 3) & 0xffu, 0x6u, 0xEu);
        }
    v16f c0 = {0}, c1 = {0};
    float m1 = (float)(x & 0xffff), m2 = m1 * 3.f, w = m1 * 7.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(WIDE(a[k]), WIDE(b[k]), c0, 4, 4, 0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                w += 1.0f;
                float n2 = __builtin_amdgcn_fmed3f(w, m1, m2);
                m1 = fminf(w, m1);
                m2 = n2;
            }
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(WIDE(a[k]), WIDE(b[(k + 1) & 3]), c1, 4, 4, 0, 0, 0, 0);
        }
    }
    float s = m1 + m2;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, int seed)
{
    v8i a[4], b[4];
    uint32_t x = (uint32_t)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u + 1u;
    for (int k = 0; k < 4; ++k) {
        a[k] = (v8i){0};
        b[k] = (v8i){0};
        for (int n = 0; n < 4; ++n) {
            x = x * 1664525u + 1013904223u;
            a[k][n] = (int)expand8((x >> 11) & 0xffu, 0x0u, 0x6u);
            b[k][n] = (int)expand8((x >> 3) & 0xffu, 0x6u, 0xEu);
        }
    }
    v16f c0 = {0}, c1 = {0};
    float m1 = (float)(x & 0xffff), m2 = m1 * 3.f, w = m1 * 7.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k], b[k], c0, 4, 4, 0, SCALE_A, 0, SCALE_B);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                w += 1.0f;
                float n2 = __builtin_amdgcn_fmed3f(w, m1, m2);
                m1 = fminf(w, m1);
                m2 = n2;
            }
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k], b[(k + 1) & 3], c1, 4, 4, 0, SCALE_A, 0, SCALE_B);
        }
    }
    float s = m1 + m2;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool SCALED>
static int run_rate(float* out, int cus, int iters)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 2; ++wps) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (SCALED) hipLaunchKernelGGL(k_rate<NV>, dim3(cus * wps), dim3(256), 0, 0, out, iters, rep);
            else hipLaunchKernelGGL(k_rate_unscaled<NV>, dim3(cus * wps), dim3(256), 0, 0, out, iters, rep);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double mfma_per_simd = (double)iters * 8 * wps;
        const double ns = best * 1e6 / mfma_per_simd;
        printf("fp4 32x32x64 %s, 2 chains, %d x 3 float VALU (add, med3, min) per pair of MFMAs, %d wave(s)/SIMD: %.2f ns per MFMA per SIMD = %.1f cycles at 2400 MHz, %.0f T bit-ops/s\n",
               SCALED ? "with block scales" : "unscaled", NV, wps, ns, ns * 2.4, 2.0 * 32 * 32 * 64 * mfma_per_simd * cus * 4 / (best * 1e-3) / 1e12);
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    uint32_t ht[64], hq[64];
    srand(7);
    for (int i = 0; i < 64; ++i) { ht[i] = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand(); hq[i] = (uint32_t)rand() * 40503u ^ ((uint32_t)rand() << 3); }
    uint32_t *dt, *dq;
    float* dout;
    CHK(hipMalloc(&dt, sizeof ht));
    CHK(hipMalloc(&dq, sizeof hq));
    CHK(hipMalloc(&dout, (size_t)p.multiProcessorCount * 2 * 256 * 4 + 64 * 16 * 4));
    CHK(hipMemcpy(dt, ht, sizeof ht, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dt, dq, dout);
    float ho[64 * 16];
    CHK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 16; ++j) {
            const int q = lane & 31, row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
            int ham = 0, pq = 0;
            for (int w = 0; w < 2; ++w) { ham += __builtin_popcount(ht[row * 2 + w] ^ hq[q * 2 + w]); pq += __builtin_popcount(hq[q * 2 + w]); }
            const float want = (float)(1000 + row) + 8192.0f * (float)(ham - pq);
            if (ho[lane * 16 + j] != want) { if (bad < 5) printf("lane %d reg %d: got %.1f want %.1f\n", lane, j, ho[lane * 16 + j], want); ++bad; }
        }
    printf("exactness and layout (train row in the register, query in the lane, x8192 by block scales): %s (%d of 1024 differ)\n", bad ? "FAIL" : "ok", bad);
    if (run_rate<0, true>(dout, p.multiProcessorCount, 20000)) return 1;
    if (run_rate<0, false>(dout, p.multiProcessorCount, 20000)) return 1;
    if (run_rate<4, false>(dout, p.multiProcessorCount, 5000)) return 1;
    if (run_rate<6, false>(dout, p.multiProcessorCount, 5000)) return 1;
    if (run_rate<8, false>(dout, p.multiProcessorCount, 5000)) return 1;
    if (run_rate<8, true>(dout, p.multiProcessorCount, 5000)) return 1;
    return 0;
}
