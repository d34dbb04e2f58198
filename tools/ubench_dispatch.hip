// Workgroup dispatch rate on MI355X: time launches of kernels that do nothing,
// for several workgroup sizes, with and without an LDS allocation, and with a
// 3-D grid like the per-frame kernels use.  Prints workgroups per microsecond.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dispatch tools/ubench_dispatch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int LDS_WORDS>
__global__ void k_empty(int* out, int never)
{
    __shared__ int lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
    if (LDS_WORDS > 0) lds[threadIdx.x % LDS_WORDS] = threadIdx.x;
    if (never) out[blockIdx.x] = LDS_WORDS > 0 ? lds[0] : 1;
}

struct big_arg { int v[120]; };

// a 480-byte by-value argument indexed with a value that depends on the block
__global__ void k_bigarg(big_arg a, int* out, int never)
{
    int l = a.v[(blockIdx.y * 7 + blockIdx.z) % 120];
    if (never + l == 12345) out[blockIdx.x] = l;
}

// one dependent global load and a store per thread: the shortest useful kernel
__global__ void k_touch(const int* in, int* out)
{
    int i = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x;
    out[i & 0x3ffff] = in[i & 0x3ffff] + 1;
}

template <class F>
static void run_any(const char* name, dim3 grid, int block, F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / reps;
    double wgs = (double)grid.x * grid.y * grid.z;
    printf("%-28s grid %6u x %4u x %2u block %4d : %8.2f us/launch  %7.1f WG/us  %7.1f waves/us\n", name,
           grid.x, grid.y, grid.z, block, us, wgs / us, wgs * (block / 64) / us);
}

template <int LDS_WORDS>
static void run(const char* name, dim3 grid, int block, int* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty<LDS_WORDS>, grid, dim3(block), 0, 0, d, 0);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_empty<LDS_WORDS>, grid, dim3(block), 0, 0, d, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / reps;
    double wgs = (double)grid.x * grid.y * grid.z;
    printf("%-28s grid %6u x %4u x %2u block %4d : %8.2f us/launch  %7.1f WG/us  %7.1f waves/us\n", name,
           grid.x, grid.y, grid.z, block, us, wgs / us, wgs * (block / 64) / us);
}

int main()
{
    int* d;
    hipMalloc(&d, 1 << 20);
    for (int block : {64, 256, 1024}) {
        run<0>("no LDS", dim3(65536), block, d);
        run<2048>("8 KB LDS", dim3(65536), block, d);
    }
    run<0>("no LDS, tiny grid", dim3(1), 256, d);
    run<0>("no LDS, 512 WGs", dim3(512), 256, d);
    run<0>("no LDS, 8192 WGs", dim3(8192), 256, d);
    run<0>("no LDS, 3-D grid", dim3(8, 1024, 8), 256, d);
    run<2048>("8 KB LDS, 3-D grid", dim3(8, 1024, 8), 256, d);
    big_arg a;
    for (int i = 0; i < 120; ++i) a.v[i] = i;
    int* d2;
    (void)hipMalloc(&d2, 1 << 20);
    for (dim3 g : {dim3(9088), dim3(8, 142, 8), dim3(64, 142, 1), dim3(142, 64, 1)}) {
        run_any("480-byte argument", g, 256, [&] { hipLaunchKernelGGL(k_bigarg, g, dim3(256), 0, 0, a, d, 0); });
        run_any("load + store per thread", g, 256, [&] { hipLaunchKernelGGL(k_touch, g, dim3(256), 0, 0, d2, d); });
    }
    hipFree(d2);
    hipFree(d);
    return 0;
}
