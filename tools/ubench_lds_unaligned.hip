// ubench_lds_unaligned.hip -- do 16-byte LDS reads at 2-byte alignment return the
// right bytes on gfx950, and what do they cost next to aligned ones?
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_lds_unaligned ubench_lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_check(uint4* out, const uint16_t* in, int shift)
{
    __shared__ __attribute__((aligned(16))) uint16_t w[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) w[i] = in[i];
    __syncthreads();
    uint4 a;
    __builtin_memcpy(&a, &w[threadIdx.x * 8 + shift], 16); // (2 * shift) bytes off a 16-byte boundary
    out[threadIdx.x] = a;
}

template <int SHIFT>
__global__ __launch_bounds__(256) void k_time(uint32_t* out, const uint16_t* in, int iters)
{
    __shared__ __attribute__((aligned(16))) uint16_t w[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) w[i] = in[i & 4095];
    __syncthreads();
    uint32_t acc = 0;
    int pos = threadIdx.x * 8 + SHIFT;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint4 a;
            __builtin_memcpy(&a, &w[(pos + u * 264) & 4095], 16);
            acc += a.x ^ a.y ^ a.z ^ a.w;
        }
        pos += 88;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    std::vector<uint16_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (uint16_t)(i * 40503u + 17u);
    uint16_t* d_in;
    uint4* d_out;
    uint32_t* d_acc;
    CHK(hipMalloc(&d_in, 8192));
    CHK(hipMalloc(&d_out, 256 * 16));
    CHK(hipMalloc(&d_acc, 1024 * 256 * 4));
    CHK(hipMemcpy(d_in, h.data(), 8192, hipMemcpyHostToDevice));
    int bad = 0;
    for (int shift = 0; shift < 8; ++shift) {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, d_out, d_in, shift);
        std::vector<uint16_t> got(256 * 8);
        CHK(hipMemcpy(got.data(), d_out, 256 * 16, hipMemcpyDeviceToHost));
        for (int t = 0; t < 256; ++t)
            for (int e = 0; e < 8; ++e)
                if (got[t * 8 + e] != h[t * 8 + shift + e]) ++bad;
    }
    printf("16-byte LDS reads at 2-byte alignment: %d wrong values\n", bad);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int v = 0; v < 3; ++v) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (v == 0) hipLaunchKernelGGL(k_time<0>, dim3(1024), dim3(256), 0, 0, d_acc, d_in, 2000);
            if (v == 1) hipLaunchKernelGGL(k_time<3>, dim3(1024), dim3(256), 0, 0, d_acc, d_in, 2000);
            if (v == 2) hipLaunchKernelGGL(k_time<4>, dim3(1024), dim3(256), 0, 0, d_acc, d_in, 2000);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("shift %d elements: %.3f ms\n", v == 0 ? 0 : (v == 1 ? 3 : 4), best);
    }
    return bad != 0;
}
