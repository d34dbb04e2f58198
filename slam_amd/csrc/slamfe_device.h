// slamfe_device.h -- device-side helpers shared by the kernel files of the
// feature front-end (gfx950, wave64).  Compile with -ffp-contract=off: the few
// float expressions (rotation of the rBRIEF pattern, fastAtan2, reprojection,
// triangulation) are a numerical contract shared with the host side
// (slamfe_math.h) and must not be fused.
#ifndef SLAMFE_DEVICE_H
#define SLAMFE_DEVICE_H

#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "slamfe.h"
#include "slamfe_dev.h"
#include "slamfe_kernels.h"
#include "slamfe_math.h"
#include "slamfe_tables.h"

#define TW SLAMFE_TILE_W
#define TH SLAMFE_TILE_H

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int find_level_by_tile(const slamfe_dev_geom& g, int t)
{
    int l = 0;
#pragma unroll
    for (int i = 1; i < SLAMFE_MAX_LEVELS; ++i)
        if (i < g.nlevels && t >= g.lv[i].tile_base) l = i;
    return l;
}

__device__ __forceinline__ int find_level_by_group(const slamfe_dev_geom& g, int t)
{
    int l = 0;
#pragma unroll
    for (int i = 1; i < SLAMFE_MAX_LEVELS; ++i)
        if (i < g.nlevels && t >= g.lv[i].group_base) l = i;
    return l;
}

__device__ __forceinline__ int find_level_by_band(const slamfe_dev_geom& g, int b)
{
    int l = 0;
#pragma unroll
    for (int i = 1; i < SLAMFE_MAX_LEVELS; ++i)
        if (i < g.nlevels && b >= g.lv[i].band_base) l = i;
    return l;
}

// Frame of a workgroup.  Per-frame kernels are launched on a grid
// (FX, items, FZ): workgroups are dealt round-robin to the 8 XCDs in linear
// order (x fastest), so with FX = 8 all workgroups of frame f run on one XCD
// (f % 8) and share that XCD's L2 -- tile halos, source rows and keypoint
// neighbourhoods are then fetched from HBM once, not once per XCD.  Batches of
// fewer than 8 frames use FX = 1 (work items interleaved over the XCDs).
__device__ __forceinline__ int wg_frame() { return (int)(blockIdx.z * gridDim.x + blockIdx.x); }
__device__ __forceinline__ int wg_item() { return (int)blockIdx.y; }

__device__ __forceinline__ int reflect101(int i, int n)
{
    // valid for -n < i < 2n-1, which covers the 3-px reach of the blur
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : i;
}

// Sum over the 64 lanes of a wavefront, result in every lane: butterflies inside
// a row of 16 lanes through DPP (no LDS traffic), then the four row sums through
// v_readlane.
__device__ __forceinline__ int wave_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false); // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false); // row_mirror
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) +
           __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// LDS accesses of one wavefront execute in program order; this keeps the
// compiler from reordering them where a wave reads what its other lanes wrote.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ int hamming256_dev(const uint32_t* __restrict__ a,
                                              const uint32_t* __restrict__ b)
{
    const uint4* pa = reinterpret_cast<const uint4*>(a);
    const uint4* pb = reinterpret_cast<const uint4*>(b);
    uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// ---------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------
// grid (FX, items, FZ) of the per-frame kernels; see wg_frame()
static inline dim3 frame_grid(int items, int nframes)
{
    int fx = nframes >= 8 ? 8 : 1;
    return dim3((unsigned)fx, (unsigned)items, (unsigned)((nframes + fx - 1) / fx));
}

#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

#endif /* SLAMFE_DEVICE_H */
