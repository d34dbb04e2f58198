// ubench_patch_loads.hip -- what does the alignment of a keypoint's patch rows cost?
// A wavefront fetches a 31-row x 32-byte patch at a random position of a row-major
// 640 x 480 image, one 16-byte load per lane (62 lanes), as the orientation kernel
// does, and reduces it with a few operations; 2000 patches per frame, 256 frames.
//   variant 0: patches at arbitrary x (loads at any byte alignment)
//   variant 1: the same patches with x rounded down to a multiple of 4
//   variant 2: x rounded down to a multiple of 16 (aligned loads)
//   variant 3: arbitrary x, fetched as two aligned 16-byte loads per lane (the
//              aligned pieces that cover the lane's 16 bytes)
//   variant 4: the image in tiles of 16 x 8 pixels (one 128-byte line each): the
//              3 x 5 tiles that hold the patch, two 16-byte pieces per lane
//   variant 5: variant 2 (aligned) with non-temporal loads
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_patch_loads ubench_patch_loads.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define W 640
#define H 480
#define NKP 2000

__device__ __forceinline__ uint32_t lcg(uint32_t x) { return x * 1664525u + 1013904223u; }

template <int VARIANT>
__global__ __launch_bounds__(256) void k_fetch(const uint8_t* __restrict__ img, uint32_t* __restrict__ out)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;
    const uint8_t* frame = img + (size_t)f * W * H;
    const int row = lane >> 1, half = lane & 1;
    uint32_t acc = 0;
    for (int idx = wave; idx < NKP; idx += nwaves) {
        uint32_t r = lcg((uint32_t)(f * NKP + idx) * 2654435761u + 12345u);
        int cx = 16 + (int)((r >> 8) % (W - 64));
        r = lcg(r);
        int cy = 16 + (int)((r >> 8) % (H - 48));
        int x0 = cx - 15;
        if (VARIANT == 1) x0 &= ~3;
        if (VARIANT == 2) x0 &= ~15;
        const uint8_t* p = frame + (size_t)(cy - 15 + (row < 31 ? row : 0)) * W + x0 + 16 * half;
        uint4 a = make_uint4(0, 0, 0, 0);
        if (VARIANT == 4) {
            const int tx0 = x0 >> 4, ty0 = (cy - 15) >> 3, tpr = W / 16;
            uint4 b = make_uint4(0, 0, 0, 0);
            const int t0 = lane >> 3, t1 = (lane + 64) >> 3, rw = lane & 7;
            a = *reinterpret_cast<const uint4*>(frame + ((size_t)(ty0 + t0 / 3) * tpr + tx0 + t0 % 3) * 128 + rw * 16);
            if (lane < 56)
                b = *reinterpret_cast<const uint4*>(frame + ((size_t)(ty0 + t1 / 3) * tpr + tx0 + t1 % 3) * 128 + rw * 16);
            a.x ^= b.y; a.y ^= b.z; a.z ^= b.w; a.w ^= b.x;
        } else if (VARIANT == 5) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            if (lane < 62) {
                const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>((uintptr_t)p & ~(uintptr_t)15));
                a = make_uint4(t.x, t.y, t.z, t.w);
            }
        } else if (VARIANT < 3) {
            if (lane < 62) __builtin_memcpy(&a, p, 16);
        } else {
            const uint8_t* q = (const uint8_t*)((uintptr_t)p & ~(uintptr_t)15);
            uint4 b = make_uint4(0, 0, 0, 0);
            if (lane < 62) {
                a = *reinterpret_cast<const uint4*>(q);
                b = *reinterpret_cast<const uint4*>(q + 16);
            }
            a.x ^= b.y; a.y ^= b.z; a.z ^= b.w; a.w ^= b.x; // stands in for the realignment
        }
        uint32_t s = __builtin_amdgcn_udot4(a.x, 0x01010101u, 0, false);
        s = __builtin_amdgcn_udot4(a.y, 0x01020304u, s, false);
        s = __builtin_amdgcn_udot4(a.z, 0x04030201u, s, false);
        s = __builtin_amdgcn_udot4(a.w, 0x01010101u, s, false);
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        acc += s;
    }
    if (lane == 0) out[(size_t)f * nwaves + wave] = acc;
}

int main()
{
    const int frames = 256;
    uint8_t* img;
    uint32_t* out;
    CHK(hipMalloc(&img, (size_t)frames * W * H + 64));
    CHK(hipMemset(img, 7, (size_t)frames * W * H + 64));
    const int wgs = (NKP + 15) / 16; // 16 keypoints per workgroup, as the orientation kernel
    CHK(hipMalloc(&out, (size_t)frames * wgs * 4 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int v = 0; v < 6; ++v) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CHK(hipEventRecord(e0, 0));
            dim3 grid(wgs, frames);
            if (v == 0) hipLaunchKernelGGL(k_fetch<0>, grid, dim3(256), 0, 0, img, out);
            if (v == 1) hipLaunchKernelGGL(k_fetch<1>, grid, dim3(256), 0, 0, img, out);
            if (v == 2) hipLaunchKernelGGL(k_fetch<2>, grid, dim3(256), 0, 0, img, out);
            if (v == 3) hipLaunchKernelGGL(k_fetch<3>, grid, dim3(256), 0, 0, img, out);
            if (v == 4) hipLaunchKernelGGL(k_fetch<4>, grid, dim3(256), 0, 0, img, out);
            if (v == 5) hipLaunchKernelGGL(k_fetch<5>, grid, dim3(256), 0, 0, img, out);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("variant %d: %.4f ms per %d frames x %d patches\n", v, best, frames, NKP);
    }
    return 0;
}
