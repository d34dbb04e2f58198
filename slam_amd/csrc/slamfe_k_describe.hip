// slamfe_k_describe.hip -- orientation, keypoint records, rBRIEF-256
#include "slamfe_device.h"

__device__ __constant__ int8_t c_pattern[1024] = SLAMFE_PATTERN_INIT;

// ---------------------------------------------------------------------------
// K5  orientation by intensity centroid (ICAngles inside cv::ORB).
//
// k_orient: a wavefront sums the moments of a keypoint's 749-pixel disc.
// k_finish: one lane per keypoint turns the moments into the angle
// (fastAtan2), its cos/sin for the descriptor, and the cv::KeyPoint record.
// ---------------------------------------------------------------------------

// A wavefront walks keypoints idx = wave, wave + nwaves, ...  The disc's 31 rows
// of 31 bytes (starting at cx-15) are cut into 8 dwords each, 248 (row, dword)
// tasks = 4 per lane.  Which bytes of a task lie inside the disc, their
// x-offsets u and the row offset v depend on the lane only, so they are turned
// once per wave into byte weights for v_dot4_u32_u8:
//   T += dot4(x, inside ? u + 32 : 0)   S += dot4(x, inside ? 1 : 0)
//   V += dot4(x, inside ? v + 16 : 0)   m10 = T - 32 S,  m01 = V - 16 S
// leaving three dot products per loaded dword for every keypoint.
// Per-lane constants of the moment sums: a lane owns one 16-byte half of one of
// the 31 patch rows (62 lanes): one 16-byte load per keypoint.  (With four dword
// loads per lane the address unit was the busiest part of the CU.)
struct orient_lane {
    uint32_t wT[4], wV[4], wS[4];
    int trow, tcol;
};

__device__ __forceinline__ orient_lane orient_setup(int lane)
{
    // half-widths of the disc rows |v| = 0..15, one nibble each
    const unsigned long long umax = 0x3689abcddeeeffffULL;
    orient_lane W;
    const int row = lane >> 1, half = lane & 1;
    const int v = row - 15, av = v < 0 ? -v : v;
    const int d = row < 31 ? (int)((umax >> (4 * av)) & 15ULL) : -1;
    // The lane's 16 bytes are u = 16 half - 15 + k, k = 0..15; inside the disc: |u| <= d,
    // a run of k.  (Made from the run's ends with a few shifts: a wave handles about
    // four keypoints, and deciding the 16 bytes one by one cost about 200 vector
    // instructions per wave, half of what the kernel executed.)
    const int klo = max(15 - 16 * half - d, 0), khi = min(15 - 16 * half + d, 15);
    const uint32_t run = khi >= klo ? (2u << (khi & 15)) - (1u << (klo & 15)) : 0u; // bit k: byte k is inside
    const uint32_t vv = (uint32_t)(v + 16) * 0x01010101u;
    const uint32_t u0 = (uint32_t)(16 * half + 17) * 0x01010101u + 0x03020100u; // u + 32 of bytes 0..3
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t n = (run >> (4 * it)) & 15u;
        const uint32_t ones = (n | (n << 7) | (n << 14) | (n << 21)) & 0x01010101u; // bit j -> byte j
        const uint32_t full = (ones << 8) - ones;                                     // 0xff per inside byte
        W.wS[it] = ones;
        W.wT[it] = full & (u0 + 0x04040404u * (uint32_t)it);
        W.wV[it] = full & vv;
    }
    W.trow = row < 31 ? row : 0;
    W.tcol = 16 * half;
    return W;
}

// The lane's 16 bytes of the patch whose top-left corner is `base` (= centre - 15, - 15).
__device__ __forceinline__ uint4 orient_load(const orient_lane& W, const uint8_t* __restrict__ base, int pitch,
                                             int lane)
{
    uint4 x4 = make_uint4(0, 0, 0, 0);
    if (lane < 62) __builtin_memcpy(&x4, base + W.trow * pitch + W.tcol, 16);
    return x4;
}

// m10, m01 of the disc from the lanes' 16-byte pieces (in every lane).
__device__ __forceinline__ void orient_moments(const orient_lane& W, const uint4& x4, int& m10, int& m01)
{
    const uint32_t xs[4] = {x4.x, x4.y, x4.z, x4.w};
    uint32_t T = 0, V = 0, S = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        T = __builtin_amdgcn_udot4(xs[it], W.wT[it], T, false);
        V = __builtin_amdgcn_udot4(xs[it], W.wV[it], V, false);
        S = __builtin_amdgcn_udot4(xs[it], W.wS[it], S, false);
    }
    m10 = wave_sum((int)T - 32 * (int)S);
    m01 = wave_sum((int)V - 16 * (int)S);
}

__global__ __launch_bounds__(256) void k_orient(slamfe_dev_geom g, const uint8_t* __restrict__ pyr,
                                                int slot_base, int nslots,
                                                const uint32_t* __restrict__ kp_key,
                                                const uint32_t* __restrict__ kp_level,
                                                const int32_t* __restrict__ kp_count,
                                                int2* __restrict__ kp_mom, int max_kp, int nframes,
                                                slamfe_level0 l0)
{
    const int f = wg_frame();
    if (f >= nframes) return;
    const int slot = (slot_base + f) % nslots;
    const int lane = threadIdx.x & 63;
    // wave-uniform values are made scalar explicitly so that the per-keypoint reads
    // below (key, level, level geometry) become scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(wg_item() * 4 + (int)(threadIdx.x >> 6));
    const int nwaves = (int)gridDim.y * 4;
    int n = kp_count[slot];
    if (n > max_kp) n = max_kp;
    if (wave >= n) return;
    const orient_lane W = orient_setup(lane);
    const uint8_t* frame = pyr + (size_t)f * g.frame_bytes;
    const uint32_t* keys = kp_key + (size_t)slot * max_kp;
    const uint32_t* levels = kp_level + (size_t)slot * max_kp;
    // (Neither the arithmetic nor the wait for a patch is what a keypoint costs here: a
    // third of the vector instructions less, or two keypoints' patches requested before
    // either is summed, change the kernel's time by 1 %.  It is the ~39 lines of 128
    // bytes that a 31-row patch of a row-major level touches.)
    uint32_t next_key = keys[wave], next_level = levels[wave];
    for (int idx = wave; idx < n; idx += nwaves) { // uniform per wave
        const uint32_t key = next_key;
        const int l = (int)next_level;
        if (idx + nwaves < n) { // fetched while this keypoint is processed
            next_key = keys[idx + nwaves];
            next_level = levels[idx + nwaves];
        }
        if (l < 0 || l >= g.nlevels) continue; // never true for entries written by k_select_write
        const int w = g.lv[l].w, h = g.lv[l].h, pitch = g.lv[l].pitch;
        const int cx = (int)SLAMFE_KEY_X(key), cy = (int)SLAMFE_KEY_Y(key);
        if (cx < 16 || cy < 16 || cx >= w - 17 || cy >= h - 16) continue; // reads must stay inside
        const uint8_t* level = l == 0 ? l0.base + (size_t)f * l0.frame_stride : frame + g.lv[l].off;
        const uint8_t* base = level + (size_t)(cy - 15) * pitch + (cx - 15);
        const uint4 x4 = orient_load(W, base, pitch, lane);
        int m10, m01;
        orient_moments(W, x4, m10, m01);
        if (lane == 0) kp_mom[(size_t)slot * max_kp + idx] = make_int2(m10, m01);
    }
}

// The cv::KeyPoint record of a selected keypoint and cos / sin of its angle.
__device__ __forceinline__ slamfe_keypoint finish_record(uint32_t key, int l, float scale, int2 m, int mode,
                                                         int patch_size, float2* rot)
{
    const int cx = (int)SLAMFE_KEY_X(key), cy = (int)SLAMFE_KEY_Y(key);
    slamfe_keypoint k;
    if (mode == SLAMFE_MODE_ORB) {
        k.angle = slamfe_fast_atan2((float)m.y, (float)m.x);
        k.x = (float)cx * scale;
        k.y = (float)cy * scale;
        k.size = (float)patch_size * scale;
        k.octave = l;
    } else {
        k.angle = -1.f;
        k.x = (float)cx;
        k.y = (float)cy;
        k.size = 7.f;
        k.octave = 0;
    }
    k.response = (float)SLAMFE_KEY_S(key);
    k.class_id = -1;
    float sn, cs;
    slamfe_sincos_deg(k.angle, &sn, &cs);
    *rot = make_float2(cs, sn);
    return k;
}

__global__ __launch_bounds__(256) void k_finish(slamfe_dev_geom g, int slot_base, int nslots,
                                                const uint32_t* __restrict__ kp_key,
                                                const uint32_t* __restrict__ kp_level,
                                                const int32_t* __restrict__ kp_count,
                                                const int2* __restrict__ kp_mom,
                                                slamfe_keypoint* __restrict__ kps,
                                                float2* __restrict__ kp_rot, int max_kp, int mode,
                                                int patch_size, int nframes)
{
    const int f = wg_frame();
    if (f >= nframes) return;
    const int slot = (slot_base + f) % nslots;
    const int idx = wg_item() * 256 + threadIdx.x;
    int n = kp_count[slot];
    if (n > max_kp) n = max_kp;
    if (idx >= n) return;
    const size_t e = (size_t)slot * max_kp + idx;
    const uint32_t key = kp_key[e];
    int l = (int)kp_level[e];
    if (l < 0 || l >= g.nlevels) l = 0;
    float2 rot;
    const int2 m = mode == SLAMFE_MODE_ORB ? kp_mom[e] : make_int2(0, 0);
    kps[e] = finish_record(key, l, g.lv[l].scale, m, mode, patch_size, &rot);
    kp_rot[e] = rot;
}

// ---------------------------------------------------------------------------
// K7  rBRIEF-256 (computeOrbDescriptors inside cv::ORB, reached from
//     desc->compute, Init.cc:78).  Lane i of a wavefront evaluates tests i,
//     64+i, 128+i, 192+i, so each wave ballot is 8 descriptor bytes.
// ---------------------------------------------------------------------------
// The blurred level is stored in tiles of 16 x 8 pixels, one 128-byte line each.
// A keypoint's samples reach +-18 px (|pattern| <= 13 * sqrt 2 = 18.4), a square
// of 37 x 37 pixels that lies in at most 4 x 6 tiles: the wave fetches those
// tiles whole (8 lanes x 16 bytes = one line) into an LDS image of 48 rows x 64
// bytes, in which the keypoint sits at (cy - 8 * first tile row, cx - 16 * first
// tile column).  (A row-major level costs about 51 lines per keypoint -- 37 rows
// of 48 bytes at an arbitrary offset -- against 18 here.)
#define DP_TCOLS 4
#define DP_TROWS 6
/* Bytes of a patch row in LDS: one 16-byte piece more than the four tile columns.  With
 * 64 bytes a row is 16 of the 64 banks, rows y and y + 4 share their banks, the eight
 * rows of a tile are staged two to a bank and the samples of a keypoint fall on 40 of
 * the 64 banks; with 80 the rows start at banks 0, 20, 40, 60, 16, ...: every bank in
 * use, and the eight 16-byte pieces of a tile on eight different bank quadruples. */
#define DP_PITCH 80
#define DP_PITCH_U4 (DP_PITCH / 16)
#define DP_LDS_ROWS (8 * DP_TROWS)
#define DP_PIECES (DP_TCOLS * DP_TROWS * 8) /* 16-byte pieces of the largest tile region */

typedef float desc_v2f __attribute__((ext_vector_type(2)));
#define DP_NIT (DP_PIECES / 64)

struct describe_lane {
    desc_v2f P[4][2], Q[4][2];
    int tcol[DP_NIT], trow[DP_NIT], trw[DP_NIT]; // piece `it` of the lane: tile column, tile row, row inside the tile
};

__device__ __forceinline__ describe_lane describe_setup(int lane)
{
    describe_lane D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int8_t* t = c_pattern + (j * 64 + lane) * 4;
        D.P[j][0] = (desc_v2f){(float)t[0], (float)t[0]};
        D.Q[j][0] = (desc_v2f){(float)t[1], (float)t[1]};
        D.P[j][1] = (desc_v2f){(float)t[2], (float)t[2]};
        D.Q[j][1] = (desc_v2f){(float)t[3], (float)t[3]};
    }
#pragma unroll
    for (int it = 0; it < DP_NIT; ++it) {
        const int p = it * 64 + lane, t = p >> 3; // eight consecutive lanes fetch one tile = one line
        D.tcol[it] = t % DP_TCOLS;
        D.trow[it] = t / DP_TCOLS;
        D.trw[it] = p & 7;
    }
    return D;
}

// The tiles that hold columns cx-18 .. cx+18 and rows cy-18 .. cy+18 of a tiled level.
struct describe_region {
    int tx0, ty0, ncols, nrows;
};

__device__ __forceinline__ describe_region describe_region_of(int cx, int cy)
{
    describe_region R;
    R.tx0 = (cx - 18) >> 4;
    R.ty0 = (cy - 18) >> 3;
    R.ncols = ((cx + 18) >> 4) - R.tx0 + 1; // 3 or 4
    R.nrows = ((cy + 18) >> 3) - R.ty0 + 1; // 5 or 6
    return R;
}

// The lane's pieces of the tile region; level = first byte of the tiled level,
// tpr = tiles per tile row (= level pitch / 16).
__device__ __forceinline__ void describe_load(const describe_lane& D, const uint8_t* __restrict__ level, int tpr,
                                              const describe_region& R, uint4 (&r)[DP_NIT])
{
#pragma unroll
    for (int it = 0; it < DP_NIT; ++it) {
        r[it] = make_uint4(0, 0, 0, 0);
        if (D.tcol[it] < R.ncols && D.trow[it] < R.nrows)
            // (a level is far below 4 GiB: 32-bit offsets from the level's uniform base)
            r[it] = *reinterpret_cast<const uint4*>(
                level + (uint32_t)(((R.ty0 + D.trow[it]) * tpr + R.tx0 + D.tcol[it]) * 128 + D.trw[it] * 16));
    }
}

// Stage the pieces in the wave's LDS image, rotate the pattern by (cos, sin) = rot,
// compare the 256 sample pairs; lanes 0..3 store the 32 descriptor bytes.
__device__ __forceinline__ void describe_finish(const describe_lane& D, const uint4 (&r)[DP_NIT], uint4* my,
                                                const describe_region& R, int cx, int cy, float2 rot, int lane,
                                                uint8_t* __restrict__ out32, uint8_t* __restrict__ out32_host = nullptr)
{
#pragma unroll
    for (int it = 0; it < DP_NIT; ++it)
        if (D.tcol[it] < R.ncols && D.trow[it] < R.nrows) my[(D.trow[it] * 8 + D.trw[it]) * DP_PITCH_U4 + D.tcol[it]] = r[it];
    // only this wavefront reads what it wrote: LDS operations of one wave are
    // ordered, the fences keep the compiler from moving accesses across
    wave_lds_fence();
    typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;
    const uint8_t* c = reinterpret_cast<const uint8_t*>(my) + (cy - 8 * R.ty0) * DP_PITCH + (cx - 16 * R.tx0);
    const desc_v2f AB = {rot.x, rot.y}, NBA = {-rot.y, rot.x};
    // Adding 1.5 * 2^23 rounds to the nearest integer r (ties to even, as cvRound)
    // and leaves the float's bit pattern at 0x4B400000 + r.  The sample's LDS address
    // c + 80 * ry + rx is therefore 80 * (low 24 bits of bits(y)) + bits(x) plus a
    // constant, in 32-bit modular arithmetic: one v_mad_u32_u24 and one v_add per sample
    // (the low 24 bits are 0x400000 + ry, positive).
    const desc_v2f magic = {12582912.f, 12582912.f};
    const uint32_t c0 = (uint32_t)(uintptr_t)(lds_bytes)c - (0x400000u * (uint32_t)DP_PITCH + 0x4B400000u);
    unsigned long long word[4];
    // all eight sample reads are issued before the first comparison: one LDS latency
    // per keypoint, not four
    int v[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            desc_v2f pt = D.P[j][e] * AB + D.Q[j][e] * NBA; // (px*a - py*b, px*b + py*a)
            pt = pt + magic;
            const uint32_t at = __umul24(__float_as_uint(pt.y), (uint32_t)DP_PITCH) + __float_as_uint(pt.x) + c0;
            v[j][e] = *(lds_bytes)(uintptr_t)at;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) word[j] = __ballot(v[j][0] < v[j][1]);
    if (lane < 4) {
        unsigned long long o = lane == 0 ? word[0] : lane == 1 ? word[1] : lane == 2 ? word[2] : word[3];
        reinterpret_cast<unsigned long long*>(out32)[lane] = o;
        if (out32_host) reinterpret_cast<unsigned long long*>(out32_host)[lane] = o;
    }
    // the next keypoint's staging must not overtake these LDS reads
    wave_lds_fence();
}

__global__ __launch_bounds__(256) void k_describe(slamfe_dev_geom g, const uint8_t* __restrict__ blur,
                                                  int slot_base, int nslots,
                                                  const uint32_t* __restrict__ kp_key,
                                                  const uint32_t* __restrict__ kp_level,
                                                  const int32_t* __restrict__ kp_count,
                                                  const float2* __restrict__ kp_rot,
                                                  uint8_t* __restrict__ desc, int max_kp, int nframes)
{
    __shared__ uint4 patch128[4][DP_LDS_ROWS * DP_PITCH_U4];
    const int f = wg_frame();
    if (f >= nframes) return;
    const int slot = (slot_base + f) % nslots;
    const int wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(wg_item() * 4 + wv);
    const int nwaves = (int)gridDim.y * 4;
    int n = kp_count[slot];
    if (n > max_kp) n = max_kp;
    if (wave >= n) return;
    const describe_lane D = describe_setup(lane);
    uint4* my = patch128[wv];
    const uint8_t* frame = blur + (size_t)f * g.frame_bytes;
    const uint32_t* keys = kp_key + (size_t)slot * max_kp;
    const uint32_t* levels = kp_level + (size_t)slot * max_kp;
    const float2* rots = kp_rot + (size_t)slot * max_kp;
    uint32_t next_key = keys[wave], next_level = levels[wave];
    float2 next_rot = rots[wave];
    for (int idx = wave; idx < n; idx += nwaves) { // uniform per wave
        const uint32_t key = next_key;
        const int l = (int)next_level;
        const float2 rot = next_rot; // cos, sin of the keypoint angle
        if (idx + nwaves < n) {      // fetched while this keypoint is processed
            next_key = keys[idx + nwaves];
            next_level = levels[idx + nwaves];
            next_rot = rots[idx + nwaves];
        }
        if (l < 0 || l >= g.nlevels) continue; // never true for entries written by k_select_write
        const int w = g.lv[l].w, h = g.lv[l].h, pitch = g.lv[l].pitch;
        const int cx = (int)SLAMFE_KEY_X(key), cy = (int)SLAMFE_KEY_Y(key);
        if (cx < 18 || cy < 18 || cx >= w - 22 || cy >= h - 18) continue; // reads must stay inside
        const describe_region R = describe_region_of(cx, cy);
        uint4 r[DP_NIT];
        describe_load(D, frame + g.lv[l].off, pitch >> 4, R, r);
        describe_finish(D, r, my, R, cx, cy, rot, lane, desc + ((size_t)slot * max_kp + idx) * 32);
    }
}

// Orientation, keypoint record and descriptor of a keypoint in one pass, for
// calls of a few frames (one launch instead of three; every lane of the wave
// evaluates the keypoint's angle and its cos / sin, which a batch leaves to one
// lane per keypoint in k_finish).  The patch loads of both images are issued
// together.  Same device functions, same results as k_orient + k_finish +
// k_describe.
__global__ __launch_bounds__(256) void k_orient_describe(slamfe_dev_geom g, const uint8_t* __restrict__ pyr,
                                                         const uint8_t* __restrict__ blur, int slot_base,
                                                         int nslots, const uint32_t* __restrict__ kp_key,
                                                         const uint32_t* __restrict__ kp_level,
                                                         const int32_t* __restrict__ kp_count,
                                                         int2* __restrict__ kp_mom,
                                                         slamfe_keypoint* __restrict__ kps,
                                                         float2* __restrict__ kp_rot, uint8_t* __restrict__ desc,
                                                         int max_kp, int mode, int patch_size, int nframes,
                                                         slamfe_host_rows host, slamfe_level0 l0)
{
    __shared__ uint4 patch128[4][DP_LDS_ROWS * DP_PITCH_U4];
    const int f = wg_frame();
    if (f >= nframes) return;
    const int slot = (slot_base + f) % nslots;
    const int wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(wg_item() * 4 + wv);
    const int nwaves = (int)gridDim.y * 4;
    int n = kp_count[slot];
    // results also go straight to the caller's side (pinned host memory) if asked for:
    // the frame's count here, records and descriptor rows where they are made
    if (host.cnt && wave == 0 && lane == 0) host.cnt[f] = n;
    if (n > max_kp) n = max_kp;
    if (wave >= n) return;
    const orient_lane W = orient_setup(lane);
    const describe_lane D = describe_setup(lane);
    uint4* my = patch128[wv];
    const uint32_t* keys = kp_key + (size_t)slot * max_kp;
    const uint32_t* levels = kp_level + (size_t)slot * max_kp;
    uint32_t next_key = keys[wave], next_level = levels[wave];
    for (int idx = wave; idx < n; idx += nwaves) { // uniform per wave
        const uint32_t key = next_key;
        int l = (int)next_level;
        if (idx + nwaves < n) { // fetched while this keypoint is processed
            next_key = keys[idx + nwaves];
            next_level = levels[idx + nwaves];
        }
        if (l < 0 || l >= g.nlevels) l = 0; // never true for entries written by k_select_write
        const int w = g.lv[l].w, h = g.lv[l].h, pitch = g.lv[l].pitch;
        const int cx = (int)SLAMFE_KEY_X(key), cy = (int)SLAMFE_KEY_Y(key);
        const size_t e = (size_t)slot * max_kp + idx;
        const size_t lvl = (size_t)f * g.frame_bytes + g.lv[l].off;
        // (the guards are those of k_orient and k_describe: reads must stay inside)
        const bool can_orient = mode == SLAMFE_MODE_ORB && !(cx < 16 || cy < 16 || cx >= w - 17 || cy >= h - 16);
        const bool can_describe = !(cx < 18 || cy < 18 || cx >= w - 22 || cy >= h - 18);
        uint4 x4 = make_uint4(0, 0, 0, 0);
        uint4 r[DP_NIT];
        // (level 0 may be the caller's frame itself, used in place)
        const uint8_t* unblurred = l == 0 ? l0.base + (size_t)f * l0.frame_stride : pyr + lvl;
        if (can_orient) x4 = orient_load(W, unblurred + (size_t)(cy - 15) * pitch + (cx - 15), pitch, lane);
        const describe_region R = describe_region_of(cx, cy);
        if (can_describe) describe_load(D, blur + lvl, pitch >> 4, R, r);
        int2 m = make_int2(0, 0);
        if (can_orient) {
            orient_moments(W, x4, m.x, m.y);
            if (lane == 0) kp_mom[e] = m;
        }
        float2 rot;
        const slamfe_keypoint k = finish_record(key, l, g.lv[l].scale, m, mode, patch_size, &rot);
        if (lane == 0) {
            kps[e] = k;
            kp_rot[e] = rot;
        }
        const size_t he = (size_t)f * max_kp + idx;
        if (host.kps && lane < 7) { // every lane holds the record: lane i sends dword i of its 28 bytes
            const uint32_t word = lane == 0 ? __float_as_uint(k.x) : lane == 1 ? __float_as_uint(k.y)
                                : lane == 2 ? __float_as_uint(k.size) : lane == 3 ? __float_as_uint(k.angle)
                                : lane == 4 ? __float_as_uint(k.response) : lane == 5 ? (uint32_t)k.octave
                                                                                       : (uint32_t)k.class_id;
            reinterpret_cast<uint32_t*>(host.kps + he * sizeof(slamfe_keypoint))[lane] = word;
        }
        if (can_describe) describe_finish(D, r, my, R, cx, cy, rot, lane, desc + e * 32, host.desc ? host.desc + he * 32 : nullptr);
        else if (host.desc && lane < 4) // the row keeps what the slot holds
            reinterpret_cast<unsigned long long*>(host.desc + he * 32)[lane] =
                reinterpret_cast<const unsigned long long*>(desc + e * 32)[lane];
    }
}

// Results of a few frames straight into pinned host memory: the keypoint count
// of every frame, its cv::KeyPoint records and descriptor rows (16 bytes per
// lane).  One launch and one stream synchronisation replace the read-back of
// the counts, a second round trip for the rows and three transfer launches.
// Workgroup (chunk, frame) moves rows [256 * chunk, 256 * chunk + 256) of the frame.
__global__ __launch_bounds__(256) void k_export(const int32_t* __restrict__ kp_count,
                                                const slamfe_keypoint* __restrict__ kps,
                                                const uint8_t* __restrict__ desc, int slot_base, int nslots,
                                                int max_kp, int with_rows, int32_t* __restrict__ h_cnt,
                                                uint8_t* __restrict__ h_kps, uint8_t* __restrict__ h_desc)
{
    const int f = blockIdx.y, slot = (slot_base + f) % nslots;
    const int count = kp_count[slot];
    if (blockIdx.x == 0 && threadIdx.x == 0) h_cnt[f] = count;
    if (!with_rows) return;
    const int n = min(count, max_kp), r0 = (int)blockIdx.x * 256;
    if (r0 >= n) return;
    const int rows = min(256, n - r0);
    const size_t K = (size_t)max_kp;
    // (slots and the host arrays hold max_kp rows per frame, so the last 16-byte
    // piece may cover bytes past the frame's rows but never past its array)
    const uint4* sk = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(kps) +
                                                     ((size_t)slot * K + r0) * sizeof(slamfe_keypoint));
    uint4* dk = reinterpret_cast<uint4*>(h_kps + ((size_t)f * K + r0) * sizeof(slamfe_keypoint));
    const int nk = (rows * (int)sizeof(slamfe_keypoint) + 15) / 16;
    for (int i = threadIdx.x; i < nk; i += 256) dk[i] = sk[i];
    const uint4* sd = reinterpret_cast<const uint4*>(desc + ((size_t)slot * K + r0) * 32);
    uint4* dd = reinterpret_cast<uint4*>(h_desc + ((size_t)f * K + r0) * 32);
    for (int i = threadIdx.x; i < rows * 2; i += 256) dd[i] = sd[i];
}

int slamfe_launch_export(hipStream_t s, const int32_t* kp_count, const slamfe_keypoint* kps, const uint8_t* desc,
                         int slot_base, int nslots, int max_kp, int nframes, int with_rows, int32_t* h_cnt,
                         uint8_t* h_kps, uint8_t* h_desc)
{
    hipLaunchKernelGGL(k_export, dim3(with_rows ? (max_kp + 255) / 256 : 1, nframes), dim3(256), 0, s, kp_count,
                       kps, desc, slot_base, nslots, max_kp, with_rows, h_cnt, h_kps, h_desc);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_orient(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                         int slot_base, int nslots, const uint32_t* kp_key, const uint32_t* kp_level,
                         const int32_t* kp_count, int32_t* kp_mom, slamfe_keypoint* kps, float* kp_rot,
                         int max_kp, int mode, int patch_size, slamfe_level0 l0)
{
    if (mode == SLAMFE_MODE_ORB) {
        // about four keypoints per wavefront when the level quotas are met exactly;
        // the kernel loops, so any count up to max_kp is covered
        long long est = 0;
        for (int l = 0; l < g.nlevels; ++l) est += g.lv[l].quota;
        if (est < 64 || est > max_kp) est = max_kp;
        hipLaunchKernelGGL(k_orient, frame_grid((int)((est + 15) / 16), nframes), dim3(256), 0, s, g, pyr,
                           slot_base, nslots, kp_key, kp_level, kp_count,
                           reinterpret_cast<int2*>(kp_mom), max_kp, nframes, l0);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_finish, frame_grid((max_kp + 255) / 256, nframes), dim3(256), 0, s, g,
                       slot_base, nslots, kp_key, kp_level, kp_count,
                       reinterpret_cast<const int2*>(kp_mom), kps, reinterpret_cast<float2*>(kp_rot),
                       max_kp, mode, patch_size, nframes);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_orient_describe(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                                  const uint8_t* blur, int slot_base, int nslots, const uint32_t* kp_key,
                                  const uint32_t* kp_level, const int32_t* kp_count, int32_t* kp_mom,
                                  slamfe_keypoint* kps, float* kp_rot, uint8_t* desc, int max_kp, int mode,
                                  int patch_size, slamfe_host_rows host, slamfe_level0 l0)
{
    long long est = 0;
    for (int l = 0; l < g.nlevels; ++l) est += g.lv[l].quota;
    if (est < 64 || est > max_kp) est = max_kp;
    // two keypoints per wavefront: a call of a few frames is short of waves, not of work
    hipLaunchKernelGGL(k_orient_describe, frame_grid((int)((est + 7) / 8), nframes), dim3(256), 0, s, g, pyr,
                       blur, slot_base, nslots, kp_key, kp_level, kp_count, reinterpret_cast<int2*>(kp_mom), kps,
                       reinterpret_cast<float2*>(kp_rot), desc, max_kp, mode, patch_size, nframes, host, l0);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_describe(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* blur,
                           int slot_base, int nslots, const uint32_t* kp_key,
                           const uint32_t* kp_level, const int32_t* kp_count,
                           const float* kp_rot, uint8_t* desc, int max_kp)
{
    // about four keypoints per wavefront when the level quotas are met exactly
    // (cv::ORB), a wavefront per four slots of capacity otherwise; the kernel loops
    long long est = 0;
    for (int l = 0; l < g.nlevels; ++l) est += g.lv[l].quota;
    if (est < 64 || est > max_kp) est = max_kp;
    hipLaunchKernelGGL(k_describe, frame_grid((int)((est + 15) / 16), nframes), dim3(256), 0, s, g, blur,
                       slot_base, nslots, kp_key, kp_level, kp_count,
                       reinterpret_cast<const float2*>(kp_rot), desc, max_kp, nframes);
    LAUNCH_CHECK();
    return 0;
}
