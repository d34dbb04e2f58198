// slamfe_kernels.hip -- gfx950 kernels of the feature front-end.
//
// Everything here is integer / byte work bound by HBM or by the integer VALU;
// nothing is a dense contraction, so there is no MFMA.  Wavefronts are 64 wide.
// Compile with -ffp-contract=off: the few float expressions (rotation of the
// rBRIEF pattern, fastAtan2) are part of the numerical contract in
// slamfe_math.h and must not be fused.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <stdint.h>

#include "slamfe.h"
#include "slamfe_dev.h"
#include "slamfe_kernels.h"
#include "slamfe_math.h"
#include "slamfe_tables.h"

#define TW SLAMFE_TILE_W
#define TH SLAMFE_TILE_H

__device__ __constant__ int8_t c_pattern[1024] = SLAMFE_PATTERN_INIT;

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

// ---------------------------------------------------------------------------
// K1  frame ingest: gray copy or BGR -> gray into pyramid level 0
//     (cv::cvtColor(frame, frame, COLOR_BGR2GRAY), Init.cc:75)
// ---------------------------------------------------------------------------
// A lane produces 16 pixels (one 16-byte store; rows are padded to 64 bytes, so
// the store may run past w inside the row).  Gray rows that are 16-byte aligned
// are copied with one 16-byte load; everything else goes pixel by pixel.
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ in, int w, int h,
                                                int stride, size_t frame_stride, int channels,
                                                uint8_t* __restrict__ pyr, uint32_t frame_bytes,
                                                int pitch, int aligned16)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 16;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.z;
    if (x >= w || y >= h) return;
    const uint8_t* src = in + (size_t)f * frame_stride + (size_t)y * stride;
    uint4 o;
    if (channels == 1 && aligned16 && x + 16 <= w) {
        o = *reinterpret_cast<const uint4*>(src + x);
    } else {
        uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int xx = x + k;
            uint32_t v = 0;
            if (xx < w) {
                if (channels == 3) {
                    int b = src[3 * xx], gg = src[3 * xx + 1], r = src[3 * xx + 2];
                    v = (uint32_t)((b * SLAMFE_B2Y + gg * SLAMFE_G2Y + r * SLAMFE_R2Y +
                                    (1 << (SLAMFE_GRAY_SHIFT - 1))) >> SLAMFE_GRAY_SHIFT);
                } else {
                    v = src[xx];
                }
            }
            wd[k >> 2] |= v << (8 * (k & 3));
        }
        o = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
    *reinterpret_cast<uint4*>(pyr + (size_t)f * frame_bytes + (size_t)y * pitch + x) = o;
}

// ---------------------------------------------------------------------------
// K0  scaled ingest: cv::resize(frame, frame, Size(), fx, fy) on the BGR frame
//     (Init.cc:103, Init.cc:126; INTER_LINEAR, 11-bit fixed point) fused with
//     the BGR -> gray conversion of Init.cc:75.  A lane produces 4 pixels: the
//     resized BGR values (optionally stored, the host still needs that frame
//     for Init.cc:294) and their gray values as one dword of pyramid level 0.
//     xt[dx] = {sx | single << 31, a0 | a1 << 16}, yt[dy] = {y0 | y1 << 16, b0 | b1 << 16}.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ingest_scaled(const uint8_t* __restrict__ in, int stride,
                                                       const uint2* __restrict__ xt,
                                                       const uint2* __restrict__ yt, int dw, int dh,
                                                       uint8_t* __restrict__ gray, int gpitch,
                                                       uint8_t* __restrict__ bgr_out, int bpitch)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const uint2 ty = yt[y];
    const uint8_t* S0 = in + (size_t)(ty.x & 0xffffu) * stride;
    const uint8_t* S1 = in + (size_t)(ty.x >> 16) * stride;
    const int b0 = (int)(ty.y & 0xffffu), b1 = (int)(ty.y >> 16);
    uint32_t g4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (x + k >= dw) break;
        const uint2 tx = xt[x + k];
        const int sx = (int)(tx.x & 0x7fffffffu) * 3;
        const bool single = tx.x >> 31;
        const int a0 = (int)(tx.y & 0xffffu), a1 = (int)(tx.y >> 16);
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int r0, r1;
            if (single) {
                r0 = S0[sx + c] * 2048;
                r1 = S1[sx + c] * 2048;
            } else {
                r0 = S0[sx + c] * a0 + S0[sx + 3 + c] * a1;
                r1 = S1[sx + c] * a0 + S1[sx + 3 + c] * a1;
            }
            int t = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v[c] = min(max(t, 0), 255);
        }
        if (bgr_out) {
            uint8_t* o = bgr_out + (size_t)y * bpitch + (size_t)(x + k) * 3;
            o[0] = (uint8_t)v[0];
            o[1] = (uint8_t)v[1];
            o[2] = (uint8_t)v[2];
        }
        uint32_t gv = (uint32_t)((v[0] * SLAMFE_B2Y + v[1] * SLAMFE_G2Y + v[2] * SLAMFE_R2Y +
                                  (1 << (SLAMFE_GRAY_SHIFT - 1))) >> SLAMFE_GRAY_SHIFT);
        g4 |= gv << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(gray + (size_t)y * gpitch + x) = g4;
}

// ---------------------------------------------------------------------------
// K2  pyramid: level l from level l-1, INTER_LINEAR_EXACT fixed point
//     (the cumulative resize chain inside cv::ORB).
//
// taps: ofs | c1 << 16 per destination column / row (host-computed in double).
// A lane produces 4 adjacent pixels and stores them as one dword.  The source
// bytes of a row come from three aligned dwords starting at (ofs[0] & ~3); the
// two taps of each pixel are cut out with v_alignbyte, widened to a 16-bit pair
// and combined with their 8.8 weights by v_dot2_u32_u16; the vertical blend of
// the two 8.8 row results is one more v_dot2 with the rounding constant as the
// addend:  out = (h0*w0 + h1*w1 + 2^15) >> 16.
// ---------------------------------------------------------------------------
typedef unsigned short slamfe_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ slamfe_us2 as_us2(uint32_t x)
{
    union { uint32_t u; slamfe_us2 v; } c;
    c.u = x;
    return c.v;
}

// A wavefront produces RS_ROWS consecutive destination rows.  Everything that
// depends on the column only (source offset, byte selection, tap weights of the
// lane's 4 pixels) is decoded once per column block; the horizontal pass of a
// source row (4 values of 8.8 fixed point, two dwords) is computed once and used
// by both destination rows that blend it (consecutive destination rows share a
// source row: 1.2 source rows per destination row instead of 2).
#define RS_ROWS 4

__global__ __launch_bounds__(256) void k_resize(slamfe_dev_geom g, int l, int nframes,
                                                uint8_t* __restrict__ pyr,
                                                const uint32_t* __restrict__ taps)
{
    const slamfe_dev_level& S = g.lv[l - 1];
    const slamfe_dev_level& D = g.lv[l];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int y_first = (wg_item() * 4 + wv) * RS_ROWS;
    const int f = wg_frame();
    if (f >= nframes || y_first >= D.h) return;
    const int y_end = min(y_first + RS_ROWS, D.h);
    const uint8_t* src = pyr + (size_t)f * g.frame_bytes + S.off;
    uint8_t* dst = pyr + (size_t)f * g.frame_bytes + D.off;
    const int spitch = S.pitch, dpitch = D.pitch;
    for (int x = (threadIdx.x & 63) * 4; x < D.w; x += 256) {
        // the tap table is padded with zero entries to a multiple of 4 per level
        const uint4 tx = *reinterpret_cast<const uint4*>(taps + D.tap_x + x);
        const uint32_t txs[4] = {tx.x, tx.y, tx.z, tx.w};
        const uint32_t a = (txs[0] & 0xffffu) & ~3u; // aligned start of the 12 source bytes
        uint32_t sh[4], sel[4];
        slamfe_us2 cv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t o = (txs[k] & 0xffffu) - a; // 0 .. 10
            const uint32_t c1 = txs[k] >> 16;
            cv[k] = as_us2((256u - c1) | (c1 << 16));
            sh[k] = o;
            sel[k] = o >> 2;
        }
        // horizontal pass of one source row: the 4 pixels' 8.8 values, two per dword.
        // 12 source bytes; bytes past the row end are read but carry weight 0
        auto hpass = [&](uint32_t p0, uint32_t p1, uint32_t p2) -> uint2 {
            uint32_t h[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t lo = sel[k] == 0 ? p0 : (sel[k] == 1 ? p1 : p2);
                uint32_t hi = sel[k] == 0 ? p1 : (sel[k] == 1 ? p2 : 0u);
                uint32_t q = __builtin_amdgcn_alignbyte(hi, lo, sh[k]); // bytes o, o+1 in bits 0..15
                q = __builtin_amdgcn_perm(0u, q, 0x0c010c00u);          // widen: 0x00 b1 0x00 b0
                h[k] = __builtin_amdgcn_udot2(as_us2(q), cv[k], 0u, false); // 8.8, <= 65280
            }
            return make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        };
        // walk the source rows the strip needs in order; the next row's 12 bytes are
        // requested before the current row is processed
        int y = y_first;
        uint32_t ty = taps[D.tap_y + y];
        int yo = (int)(ty & 0xffffu);
        uint32_t w1 = ty >> 16;
        int yo1 = yo + (w1 ? 1 : 0);
        int sy = yo;
        const uint32_t* r = reinterpret_cast<const uint32_t*>(src + (size_t)sy * spitch + a);
        uint32_t p0 = r[0], p1 = r[1], p2 = r[2];
        uint2 Hprev = make_uint2(0, 0), Hcur = make_uint2(0, 0);
        while (y < y_end) { // uniform
            const int nsy = min(sy + 1, S.h - 1);
            const uint32_t* rn = reinterpret_cast<const uint32_t*>(src + (size_t)nsy * spitch + a);
            const uint32_t n0 = rn[0], n1 = rn[1], n2 = rn[2];
            Hprev = Hcur;
            Hcur = hpass(p0, p1, p2);
            while (y < y_end && yo1 == sy) { // destination rows whose second source row is sy
                const uint2 H0 = yo == sy ? Hcur : Hprev;
                const slamfe_us2 wv2 = as_us2((256u - w1) | (w1 << 16));
                // pair the two rows' values of each pixel: (h0, h1) -> one v_dot2 with the
                // rounding constant as the addend: out = (h0*w0 + h1*w1 + 2^15) >> 16
                const uint32_t e0 = __builtin_amdgcn_perm(Hcur.x, H0.x, 0x05040100u); // px0: low halves
                const uint32_t e1 = __builtin_amdgcn_perm(Hcur.x, H0.x, 0x07060302u); // px1: high halves
                const uint32_t e2 = __builtin_amdgcn_perm(Hcur.y, H0.y, 0x05040100u);
                const uint32_t e3 = __builtin_amdgcn_perm(Hcur.y, H0.y, 0x07060302u);
                uint32_t v0 = __builtin_amdgcn_udot2(as_us2(e0), wv2, 1u << 15, false) >> 16;
                uint32_t v1 = __builtin_amdgcn_udot2(as_us2(e1), wv2, 1u << 15, false) >> 16;
                uint32_t v2 = __builtin_amdgcn_udot2(as_us2(e2), wv2, 1u << 15, false) >> 16;
                uint32_t v3 = __builtin_amdgcn_udot2(as_us2(e3), wv2, 1u << 15, false) >> 16;
                const uint32_t out = min(v0, 255u) | (min(v1, 255u) << 8) | (min(v2, 255u) << 16) |
                                     (min(v3, 255u) << 24);
                // rows are padded to a multiple of 64 bytes: the dword store stays in the row
                *reinterpret_cast<uint32_t*>(dst + (size_t)y * dpitch + x) = out;
                if (++y < y_end) {
                    ty = taps[D.tap_y + y];
                    yo = (int)(ty & 0xffffu);
                    w1 = ty >> 16;
                    yo1 = yo + (w1 ? 1 : 0);
                }
            }
            ++sy;
            p0 = n0;
            p1 = n1;
            p2 = n2;
        }
    }
}

// ---------------------------------------------------------------------------
// K3+K6  FAST-9/16 (segment test, score, strict 3x3 NMS, border filter) fused
//        with the 7x7 sigma=2 blur of the same level.
//        (detector->detect, Init.cc:76, per level inside cv::ORB; and the
//        GaussianBlur(7x7, 2) cv::ORB applies before describing, Init.cc:78.)
//
// One workgroup = one 64x16 tile of one level of one frame.  The tile plus a
// 4-px halo is read from HBM once (aligned dwords, REFLECT_101 at the image
// border) into LDS and feeds both computations:
//
//  FAST   pass B: every pixel of the tile + 1-px ring gets the 4-point
//                 pretest (any arc of 9 holds one pixel of each opposite pair,
//                 so both max(p0,p8) and max(p4,p12) must exceed v+t, or both
//                 mins lie below v-t); survivors are compacted into an LDS list
//         pass C: full score for list entries only (dense lanes): max over the
//                 16 arcs of min(d) and of min(-d), minus 1
//         pass D: NMS of list entries against the LDS score tile; keypoints
//                 inside the ORB border go to the tile's candidate list and to
//                 the level's score histogram
//  blur   rows:   v_dot4_u32_u8 of byte windows cut out with v_alignbyte
//                 (3.5 VALU ops per output), 16-bit sums to LDS
//         cols:   v_dot2_u32_u16 per tap, round, saturate, 4 pixels per store
// ---------------------------------------------------------------------------
#define FL_W (TW + 32)         /* LDS tile row: image columns x0-16 .. x0+79, six 16-byte pieces */
#define FL_WD (FL_W / 4)
#define FL_X0 12               /* byte column of image column x0-4 (the first one that is used) */
#define FL_H (TH + 8)          /* 24 rows: y0-4 .. y0+19 */
#define FB_COLS (TW + 8)       /* used columns: x0-4 .. x0+67 = 18 dwords */
#define FB_COLD (FB_COLS / 4)
#define FS_W (TW + 2)          /* score tile: 66 x 18 */
#define FS_H (TH + 2)
#define FS_N (FS_W * FS_H)     /* 1188 */
#define BR_H (TH + 6)          /* rows of blur row sums: 22 */

__device__ __forceinline__ int fast_score_lds(const uint8_t* c, int thr)
{
    // ring pixel offsets in the LDS tile (pitch FL_W)
    constexpr int R[16][2] = SLAMFE_FAST_RING_INIT;
    int v = c[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = v - (int)c[R[k][0] + R[k][1] * FL_W];
    // min / max over every arc of 9 contiguous ring pixels:
    // arc(s) = d[s..s+8] = three runs of three
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int a = d[k], b = d[(k + 1) & 15], e = d[(k + 2) & 15];
        mn3[k] = min(a, min(b, e));
        mx3[k] = max(a, max(b, e));
    }
    int A = INT_MIN, Bn = INT_MAX;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int mn9 = min(mn3[k], min(mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
        int mx9 = max(mx3[k], max(mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
        A = max(A, mn9);   // ring darker than centre by at least A on some arc
        Bn = min(Bn, mx9); // ring brighter than centre by at least -Bn on some arc
    }
    int m = max(A, -Bn);
    // corner <=> some arc has all |d| > thr on one side; score = that margin - 1
    return m > thr ? m - 1 : 0;
}

__global__ __launch_bounds__(256) void k_fast_blur(slamfe_dev_geom g, const uint8_t* __restrict__ pyr,
                                                   uint8_t* __restrict__ blur, int nframes, int thr,
                                                   int edge, uint32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ cand_cnt)
{
    __shared__ uint32_t tile32[FL_H * FL_WD];
    __shared__ uint32_t sc32[(FS_N + 3) / 4];
    __shared__ uint32_t rows32[BR_H * TW / 2];
    __shared__ uint16_t pos[FS_N];
    __shared__ uint32_t list[SLAMFE_TILE_CAND];
    __shared__ uint32_t nlist, npos;
    const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);
    uint8_t* sc = reinterpret_cast<uint8_t*>(sc32);

    const int t = wg_item(), f = wg_frame(), tid = threadIdx.x;
    if (f >= nframes) return;
    const int l = find_level_by_tile(g, t);
    const slamfe_dev_level L = g.lv[l];
    const int tl = t - L.tile_base;
    const int x0 = (tl % L.tiles_x) * TW, y0 = (tl / L.tiles_x) * TH;
    const uint8_t* img = pyr + (size_t)f * g.frame_bytes + L.off;

    if (tid == 0) {
        nlist = 0;
        npos = 0;
    }
    // ---- tile load: 24 rows x six aligned 16-byte pieces (columns x0-16 .. x0+79;
    // columns x0-4 .. x0+67 are used).  Rows reflect at the top / bottom border by
    // index; pieces outside the row are zero, bytes past w inside the row's
    // padding are don't-care until the fix-up below.
    if (tid < FL_H * 6) {
        int ly = tid / 6, pc = tid - ly * 6;
        int gy = reflect101(y0 - 4 + ly, L.h);
        int gx = x0 - 16 + 16 * pc;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gx >= 0 && gx < L.pitch)
            v = *reinterpret_cast<const uint4*>(img + (size_t)gy * L.pitch + gx);
        *reinterpret_cast<uint4*>(&tile32[ly * FL_WD + 4 * pc]) = v;
    }
    for (int i = tid; i < (FS_N + 3) / 4; i += 256) sc32[i] = 0;
    __syncthreads();
    // REFLECT_101 at the left / right image border (the blur reads 3 px across
    // it): copy the mirrored bytes inside LDS.  Only border tiles take this.
    {
        const bool left = x0 == 0, right = x0 + TW + 3 >= L.w;
        if (left || right) { // uniform per workgroup
            uint8_t* tb = reinterpret_cast<uint8_t*>(tile32);
            if (tid < FL_H * 8) {
                int ly = tid >> 3, k = tid & 7; // k 0..2: left side, 4..6: right side
                if (left && k < 3) {
                    // column -(k+1) <- column k+1
                    tb[ly * FL_W + FL_X0 + 4 - (k + 1)] = tb[ly * FL_W + FL_X0 + 4 + (k + 1)];
                } else if (right && k >= 4 && k < 7) {
                    int x = L.w + (k - 4); // <- column 2(w-1) - x
                    int lx = x - (x0 - 16), sx = 2 * (L.w - 1) - x - (x0 - 16);
                    if (lx < FL_W && sx >= 0) tb[ly * FL_W + lx] = tb[ly * FL_W + sx];
                }
            }
            __syncthreads();
        }
    }

    // ---- FAST pass B: pretest on the tile + 1-px ring, 8 pixels per task.
    // Task = (score row sy of 18, dword pair m of 9); its pixels are LDS bytes
    // lx = 8m .. 8m+7 of row ly = sy + 3.  Bytes are widened to 16-bit pairs
    // (pixels 0,2 and 1,3 of a dword) so that packed 16-bit min / max / sub
    // handle two pixels per operation; a negative difference flags a survivor.
    // (162 tasks: three wave passes; with 4 pixels per task it took six.)
    {
        typedef short v2s __attribute__((ext_vector_type(2)));
        union pk { uint32_t u; v2s v; };
        const int blo = max(3, edge - 1);
        // tiles that lie entirely in the discarded border band skip FAST
        const bool any_fast = x0 + TW + 1 > blo && x0 - 1 < L.w - blo && y0 + TH + 1 > blo &&
                              y0 - 1 < L.h - blo;
        const bool interior = x0 - 1 >= blo && x0 + TW <= L.w - blo - 1 && y0 - 1 >= blo &&
                              y0 + TH <= L.h - blo - 1;
        const v2s t2 = {(short)thr, (short)thr};
        // survivors among the 4 pixels of dword C (ring pixels: U above, D below,
        // L3 / R3 three to the left / right), one flag bit per pixel
        auto pretest4 = [&](uint32_t C, uint32_t U, uint32_t D, uint32_t L3, uint32_t R3) -> uint32_t {
            uint32_t flags = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                pk v, p0, p8, p4, p12;
                if (half == 0) {
                    v.u = C & 0x00ff00ffu; p0.u = D & 0x00ff00ffu; p8.u = U & 0x00ff00ffu;
                    p4.u = R3 & 0x00ff00ffu; p12.u = L3 & 0x00ff00ffu;
                } else { // bytes 1 and 3, widened by one v_perm each
                    v.u = __builtin_amdgcn_perm(0u, C, 0x0c030c01u);
                    p0.u = __builtin_amdgcn_perm(0u, D, 0x0c030c01u);
                    p8.u = __builtin_amdgcn_perm(0u, U, 0x0c030c01u);
                    p4.u = __builtin_amdgcn_perm(0u, R3, 0x0c030c01u);
                    p12.u = __builtin_amdgcn_perm(0u, L3, 0x0c030c01u);
                }
                v2s hi = __builtin_elementwise_min(__builtin_elementwise_max(p0.v, p8.v),
                                                   __builtin_elementwise_max(p4.v, p12.v));
                v2s lo = __builtin_elementwise_max(__builtin_elementwise_min(p0.v, p8.v),
                                                   __builtin_elementwise_min(p4.v, p12.v));
                pk r1, r2;
                r1.v = (v.v + t2) - hi; // < 0  <=>  hi > v + t
                r2.v = lo - (v.v - t2); // < 0  <=>  lo < v - t
                uint32_t neg = (r1.u | r2.u) & 0x80008000u;
                // pair element 0 -> pixel `half`, element 1 -> pixel 2 + half
                flags |= ((neg >> 15) & 1u) << half;
                flags |= (neg >> 31) << (2 + half);
            }
            return flags;
        };
        for (int i = tid; any_fast && i < FS_H * (FB_COLD / 2); i += 256) {
            int sy = i / (FB_COLD / 2), m = i - sy * (FB_COLD / 2);
            int ly = sy + 3;
            const uint32_t* row = &tile32[ly * FL_WD + FL_X0 / 4 + 2 * m];
            uint32_t C0 = row[0], C1 = row[1], Cp = row[-1], Cn = row[2];
            uint32_t U0 = row[-3 * FL_WD], U1 = row[-3 * FL_WD + 1];
            uint32_t D0 = row[3 * FL_WD], D1 = row[3 * FL_WD + 1];
            uint32_t flags = pretest4(C0, U0, D0, __builtin_amdgcn_alignbyte(C0, Cp, 1),
                                      __builtin_amdgcn_alignbyte(C1, C0, 3)) |
                             pretest4(C1, U1, D1, __builtin_amdgcn_alignbyte(C1, C0, 1),
                                      __builtin_amdgcn_alignbyte(Cn, C1, 3)) << 4;
            // pixel k is a score position iff lx = 8m+k in [3, 68] and the image
            // position is inside the band that can still yield a keypoint:
            // keypoints outside [edge, w-edge) x [edge, h-edge) are discarded by
            // the ORB border filter, and NMS looks one pixel further, so scores
            // are needed on [lo, w-lo) x [lo, h-lo) with lo = max(3, edge-1)
            // (tiles whose score area lies inside that band need the column rule only)
            uint32_t valid;
            if (interior) { // uniform
                valid = m == 0 ? 0xf8u : (m == FB_COLD / 2 - 1 ? 0x1fu : 0xffu);
            } else {
                int gy = y0 - 1 + sy, gxb = x0 - 4 + 8 * m;
                int kmin = max(3 - 8 * m, blo - gxb), kmax = min(FS_W + 2 - 8 * m, L.w - blo - 1 - gxb);
                kmin = max(kmin, 0);
                kmax = min(kmax, 7);
                valid = (gy >= blo && gy < L.h - blo && kmax >= kmin)
                            ? ((2u << kmax) - 1u) & ~((1u << kmin) - 1u) : 0u;
            }
            flags &= valid;
            // wave-level compaction: exclusive prefix of the per-lane counts
            // (0..8) from four ballots, one LDS atomic per wave.  (A per-lane
            // atomicAdd of the count is turned by the compiler into a serial
            // loop over the lanes and made the kernel 12 % slower.)
            const uint32_t c = (uint32_t)__popc(flags);
            const unsigned long long m0 = __ballot(c & 1u), m1 = __ballot(c & 2u), m2 = __ballot(c & 4u),
                                     m3 = __ballot(c & 8u);
            const unsigned long long act = __ballot(1);
            uint32_t pre = lane_rank(m0) + 2u * lane_rank(m1) + 4u * lane_rank(m2) + 8u * lane_rank(m3);
            const uint32_t tot = (uint32_t)__popcll(m0) + 2u * (uint32_t)__popcll(m1) +
                                 4u * (uint32_t)__popcll(m2) + 8u * (uint32_t)__popcll(m3);
            const uint32_t rank = lane_rank(act);
            uint32_t base = 0;
            if (tot) { // wave-uniform
                if (rank == 0) base = atomicAdd(&npos, tot);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + pre;
                int sbase = sy * FS_W + 8 * m - 3;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (flags & (1u << k)) pos[base++] = (uint16_t)(sbase + k);
            }
        }
    }

    // ---- blur row pass: task = (row r of 22, group k of 8) -> 8 sums
    // (176 tasks: three wave passes)
    {
        const uint32_t KA = 18u | (34u << 8) | (49u << 16) | (55u << 24); // taps 0..3
        const uint32_t KB = 49u | (34u << 8) | (18u << 16);               // taps 4..6, 0
        // the 4 sums of the dword whose window is (d0, d1, d2): output c = 4k+j reads
        // bytes (j+1 .. j+7) of the 12-byte window
        auto rows4 = [&](uint32_t d0, uint32_t d1, uint32_t d2) -> uint2 {
            uint32_t s0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), KA, 0, false);
            s0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), KB, s0, false);
            uint32_t s1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), KA, 0, false);
            s1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), KB, s1, false);
            uint32_t s2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), KA, 0, false);
            s2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), KB, s2, false);
            uint32_t s3 = __builtin_amdgcn_udot4(d1, KA, 0, false);
            s3 = __builtin_amdgcn_udot4(d2, KB, s3, false);
            // sums are <= 257 * 255 = 65535: two per dword
            return make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
        };
        for (int i = tid; i < BR_H * 8; i += 256) {
            int r = i >> 3, k = i & 7;
            const uint32_t* p = &tile32[(r + 1) * FL_WD + FL_X0 / 4 + 2 * k]; // columns x0-4+8k .. +15
            uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
            uint2 a = rows4(d0, d1, d2), c = rows4(d1, d2, d3);
            *reinterpret_cast<uint4*>(&rows32[r * (TW / 2) + 4 * k]) = make_uint4(a.x, a.y, c.x, c.y);
        }
    }
    __syncthreads();

    // ---- FAST pass C: full score of the pretest survivors
    const uint32_t np = npos;
    for (uint32_t k = tid; k < np; k += 256) {
        int i = pos[k];
        int sy = i / FS_W, sx = i - sy * FS_W;
        sc[i] = (uint8_t)fast_score_lds(&tile[(sy + 3) * FL_W + FL_X0 + sx + 3], thr);
    }

    // ---- blur column pass: task = (row r of 16, group k of 16) -> 4 pixels
    {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        constexpr int K[7] = SLAMFE_GAUSS7_INIT;
        int r = tid >> 4, k = tid & 15;
        int gy = y0 + r, gx = x0 + 4 * k;
        if (gy < L.h && gx < L.w) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                uint2 v = *reinterpret_cast<const uint2*>(&rows32[(r + i) * (TW / 2) + 2 * k]);
                us2 lo, hi, klo, khi;
                lo.x = (unsigned short)v.x; lo.y = (unsigned short)(v.x >> 16);
                hi.x = (unsigned short)v.y; hi.y = (unsigned short)(v.y >> 16);
                klo.x = (unsigned short)K[i]; klo.y = 0;
                khi.x = 0; khi.y = (unsigned short)K[i];
                a0 = __builtin_amdgcn_udot2(lo, klo, a0, false);
                a1 = __builtin_amdgcn_udot2(lo, khi, a1, false);
                a2 = __builtin_amdgcn_udot2(hi, klo, a2, false);
                a3 = __builtin_amdgcn_udot2(hi, khi, a3, false);
            }
            a0 = min((a0 + 32768u) >> 16, 255u);
            a1 = min((a1 + 32768u) >> 16, 255u);
            a2 = min((a2 + 32768u) >> 16, 255u);
            a3 = min((a3 + 32768u) >> 16, 255u);
            uint8_t* out = blur + (size_t)f * g.frame_bytes + L.off + (size_t)gy * L.pitch + gx;
            // rows are padded to a multiple of 64 bytes, so the dword store stays
            // inside the row even when it covers pixels beyond w
            *reinterpret_cast<uint32_t*>(out) = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
        }
    }
    __syncthreads();

    // ---- FAST pass D: NMS of the survivors, border filter, emission
    for (uint32_t k = tid; k < np; k += 256) {
        int i = pos[k];
        int s = sc[i];
        int sy = i / FS_W, sx = i - sy * FS_W;
        if (s > 0 && sy >= 1 && sy <= TH && sx >= 1 && sx <= TW) {
            const uint8_t* p = &sc[i];
            int m = max(max(max(p[-1], p[1]), max(p[-FS_W - 1], p[-FS_W])),
                        max(max(p[-FS_W + 1], p[FS_W - 1]), max(p[FS_W], p[FS_W + 1])));
            int gx = x0 - 1 + sx, gy = y0 - 1 + sy;
            if (s > m && gx >= edge && gx < L.w - edge && gy >= edge && gy < L.h - edge) {
                uint32_t q = atomicAdd(&nlist, 1u);
                list[q] = SLAMFE_KEY(gx, gy, s);
            }
        }
    }
    __syncthreads();
    const uint32_t n = nlist;
    size_t tg = (size_t)f * g.total_tiles + t;
    if (tid < n) cand[tg * SLAMFE_TILE_CAND + tid] = list[tid];
    if (tid == 0) cand_cnt[tg] = n;
}

// ---------------------------------------------------------------------------
// K4  keypoint selection (KeyPointsFilter::retainBest inside cv::ORB) and
//     ordered compaction.
//
// Scores are integers below 256, so the retain-best cut-off is read from a
// histogram: cut = the quota-th best score of the level; every candidate with
// score >= cut is kept (ties included).
//   k_band_hist   one workgroup per tile row ("band"): LDS histogram of the
//                 band's candidate scores, stored whole (no global atomics: a
//                 level-wide histogram updated per keypoint serialises on a
//                 few hot bins)
//   k_level_cut   one workgroup per level: sum the band histograms, find the
//                 cut, the kept total and every band's output offset (a kernel
//                 of its own: the kernel boundary is the cheapest way to make
//                 the band histograms visible; an agent-scope fence per band
//                 writes the whole L2 back and cost 0.6 ms per batch)
//   k_select_write  per band: gather the kept keys, bitonic sort by (y, x),
//                 write at the band's offset -> (level, y, x) order
// Bands are small (a 640-px level-0 band holds ~300 candidates and keeps ~15),
// so a band is one wavefront's job and a workgroup takes four bands.
// ---------------------------------------------------------------------------
#define SLAMFE_MAX_TILES_X 64 /* 4096 / 64 */

// LDS accesses of one wavefront execute in program order; this keeps the
// compiler from reordering them where a wave reads what its other lanes wrote.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// The candidate lists of one band (a row of tiles), walked by one wavefront: a
// lane owns slots 4*lane .. 4*lane+3 of every tile's list (one 16-byte load),
// four tiles are in flight at a time.  cnt = the lane's copy of cand_cnt for
// tile `lane` of the band.  fn(key, valid) is called by all 64 lanes together.
template <class F>
__device__ __forceinline__ void wave_band_candidates(const uint32_t* __restrict__ cand, size_t t0,
                                                     int tiles_x, uint32_t cnt, F fn)
{
    const int lane = threadIdx.x & 63;
    for (int tx0 = 0; tx0 < tiles_x; tx0 += 4) { // uniform
        uint4 k[4];
        uint32_t n[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tx = tx0 + u;
            n[u] = tx < tiles_x ? (uint32_t)__builtin_amdgcn_readlane((int)cnt, tx < tiles_x ? tx : 0) : 0u;
            k[u] = make_uint4(0, 0, 0, 0);
            if ((uint32_t)lane * 4u < n[u])
                k[u] = *reinterpret_cast<const uint4*>(cand + (t0 + tx) * SLAMFE_TILE_CAND + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (n[u] == 0) continue; // uniform
            fn(k[u].x, (uint32_t)lane * 4u + 0u < n[u]);
            fn(k[u].y, (uint32_t)lane * 4u + 1u < n[u]);
            fn(k[u].z, (uint32_t)lane * 4u + 2u < n[u]);
            fn(k[u].w, (uint32_t)lane * 4u + 3u < n[u]);
        }
    }
}

// One wavefront per band, four bands per workgroup, no workgroup barriers.
__global__ __launch_bounds__(256) void k_band_hist(slamfe_dev_geom g,
                                                   const uint32_t* __restrict__ cand,
                                                   const uint32_t* __restrict__ cand_cnt,
                                                   uint32_t* __restrict__ band_hist, int nframes)
{
    __shared__ uint32_t hs[4][256];
    const int f = wg_frame(), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = __builtin_amdgcn_readfirstlane(wg_item() * 4 + wv);
    if (f >= nframes || b >= g.total_bands) return;
    const int l = find_level_by_band(g, b);
    const int tiles_x = g.lv[l].tiles_x;
    uint32_t* h = hs[wv];
    *reinterpret_cast<uint4*>(&h[lane * 4]) = make_uint4(0, 0, 0, 0);
    const size_t t0 = (size_t)f * g.total_tiles + g.lv[l].tile_base + (size_t)(b - g.lv[l].band_base) * tiles_x;
    const uint32_t cnt = lane < tiles_x ? cand_cnt[t0 + lane] : 0u;
    wave_lds_fence();
    wave_band_candidates(cand, t0, tiles_x, cnt, [&](uint32_t k, bool valid) {
        if (valid) atomicAdd(&h[SLAMFE_KEY_S(k)], 1u);
    });
    wave_lds_fence();
    *reinterpret_cast<uint4*>(&band_hist[((size_t)f * g.total_bands + b) * 256 + lane * 4]) =
        *reinterpret_cast<const uint4*>(&h[lane * 4]);
}

// Inclusive prefix sum over the 256 threads of a workgroup: shuffles inside the
// wavefronts, one barrier for the four wave totals.
__device__ __forceinline__ uint32_t block_incl_scan256(uint32_t v, uint32_t* wtot /* [4] in LDS */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += t;
    }
    if (lane == 63) wtot[wv] = v;
    __syncthreads();
    for (int k = 0; k < wv; ++k) v += wtot[k];
    return v;
}

__global__ __launch_bounds__(256) void k_level_cut(slamfe_dev_geom g,
                                                   const uint32_t* __restrict__ band_hist,
                                                   uint32_t* __restrict__ band_off,
                                                   uint32_t* __restrict__ level_kept,
                                                   int32_t* __restrict__ level_cutoff, int nframes)
{
    __shared__ uint32_t suffix[256];
    __shared__ uint32_t bkept[256];
    __shared__ uint32_t wtot[2][4];
    const int l = wg_item(), f = wg_frame(), tid = threadIdx.x;
    if (f >= nframes) return;
    const int nb = g.lv[l].tiles_y, quota = g.lv[l].quota;
    const uint32_t* bh = band_hist + ((size_t)f * g.total_bands + g.lv[l].band_base) * 256;
    // thread tid sums score s = 255 - tid over the bands, so that an inclusive
    // prefix over tid is the suffix sum over scores.  Loads are issued eight
    // bands at a time: one load per loop iteration would expose the full memory
    // latency nb times.
    const int s = 255 - tid;
    uint32_t mine = 0;
    for (int b0 = 0; b0 < nb; b0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b0 + u < nb ? bh[(size_t)(b0 + u) * 256 + s] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) mine += v[u];
    }
    bkept[tid] = 0;
    // suffix[s] = candidates of the level with score >= s
    const uint32_t suf = block_incl_scan256(mine, wtot[0]);
    suffix[s] = suf;
    // largest s with suffix[s] >= quota = (number of s with suffix[s] >= quota) - 1
    const int c = __syncthreads_count(suf >= (uint32_t)quota); // also publishes suffix[]
    const uint32_t total = suffix[0];
    int cut;
    uint32_t kept;
    if (total <= (uint32_t)quota) {
        cut = 0;
        kept = total;
    } else if (quota <= 0) {
        cut = 256;
        kept = 0;
    } else {
        cut = c - 1;
        kept = suffix[cut];
    }
    if (tid == 0) {
        level_kept[f * SLAMFE_MAX_LEVELS + l] = kept;
        level_cutoff[f * SLAMFE_MAX_LEVELS + l] = cut;
    }
    // kept count of every band (a level has at most 4096 / 16 = 256 bands): all
    // threads read a band's row together, thread = score
    for (int b0 = 0; b0 < nb; b0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = (b0 + u < nb && tid >= cut) ? bh[(size_t)(b0 + u) * 256 + tid] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int sum = wave_sum((int)v[u]);
            if ((tid & 63) == 0 && sum) atomicAdd(&bkept[b0 + u], (uint32_t)sum);
        }
    }
    __syncthreads();
    // exclusive prefix = the band's offset inside the level
    const uint32_t cb = tid < nb ? bkept[tid] : 0u;
    const uint32_t incl = block_incl_scan256(cb, wtot[1]);
    if (tid < nb) band_off[(size_t)f * g.total_bands + g.lv[l].band_base + tid] = incl - cb;
}

// Bitonic sort of 64 keys held one per lane, ascending by lane.
__device__ __forceinline__ uint32_t wave_sort64(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v, j);
            const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
            v = keep_min ? min(v, o) : max(v, o);
        }
    return v;
}

// k_select_write: four bands per workgroup.  A band that keeps at most 64
// keypoints (the usual case under cv::ORB's quotas) is handled by one wavefront
// on its own: ballot compaction, a register bitonic sort, no barriers.  Bands
// that keep more (no quota, or very wide frames) are then sorted one after the
// other by the whole workgroup in the dynamic LDS buffer.
__global__ __launch_bounds__(256) void k_select_write(
    slamfe_dev_geom g, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ cand_cnt,
    const uint32_t* __restrict__ band_off, const uint32_t* __restrict__ level_kept,
    const int32_t* __restrict__ level_cutoff, int slot_base, int nslots,
    uint32_t* __restrict__ kp_key, uint32_t* __restrict__ kp_level, int32_t* __restrict__ kp_count,
    int max_kp, int nframes)
{
    extern __shared__ uint32_t lds[]; // sort buffer for large bands: tiles_x_max * 256 entries, power of two
    __shared__ uint32_t small[4][64];
    __shared__ uint32_t big_n[4], big_off[4];
    __shared__ uint32_t nkeep;
    const int f = wg_frame(), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (f >= nframes) return;
    const int slot = (slot_base + f) % nslots;
    const int b = __builtin_amdgcn_readfirstlane(wg_item() * 4 + wv);
    if (lane == 0) big_n[wv] = 0;
    if (b < g.total_bands) { // uniform per wave
        const int l = find_level_by_band(g, b);
        const int tiles_x = g.lv[l].tiles_x, bl = b - g.lv[l].band_base;
        const int cut = level_cutoff[f * SLAMFE_MAX_LEVELS + l];
        // offset = kept keypoints of lower levels + the band's offset inside the level
        const uint32_t boff = band_off[(size_t)f * g.total_bands + b];
        const uint32_t bnext = bl + 1 < g.lv[l].tiles_y ? band_off[(size_t)f * g.total_bands + b + 1] : 0u;
        uint32_t lk[SLAMFE_MAX_LEVELS];
#pragma unroll
        for (int i = 0; i < SLAMFE_MAX_LEVELS; ++i)
            lk[i] = i < g.nlevels ? level_kept[f * SLAMFE_MAX_LEVELS + i] : 0u;
        uint32_t offset = boff, tot = 0, mykept = 0;
#pragma unroll
        for (int i = 0; i < SLAMFE_MAX_LEVELS; ++i) {
            if (i < l) offset += lk[i];
            if (i == l) mykept = lk[i];
            tot += lk[i];
        }
        if (b == 0 && lane == 0) kp_count[slot] = (int32_t)tot;
        const uint32_t n = (bl + 1 < g.lv[l].tiles_y ? bnext : mykept) - boff; // kept in this band
        if (n > 64) {
            if (lane == 0) {
                big_n[wv] = n;
                big_off[wv] = offset;
            }
        } else if (n > 0) {
            const size_t t0 = (size_t)f * g.total_tiles + g.lv[l].tile_base + (size_t)bl * tiles_x;
            const uint32_t cnt = lane < tiles_x ? cand_cnt[t0 + lane] : 0u;
            uint32_t* my = small[wv];
            uint32_t base = 0;
            wave_band_candidates(cand, t0, tiles_x, cnt, [&](uint32_t k, bool valid) {
                const bool keep = valid && (int)SLAMFE_KEY_S(k) >= cut;
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const uint32_t r = base + lane_rank(m);
                    if (r < 64) my[r] = k;
                }
                base += (uint32_t)__popcll(m);
            });
            wave_lds_fence();
            uint32_t v = (uint32_t)lane < n ? my[lane] : 0xffffffffu;
            v = wave_sort64(v);
            const uint32_t o = offset + lane;
            if ((uint32_t)lane < n && o < (uint32_t)max_kp) {
                kp_key[(size_t)slot * max_kp + o] = v;
                kp_level[(size_t)slot * max_kp + o] = (uint32_t)l;
            }
        }
    }
    __syncthreads();
    // large bands, one after the other, by the whole workgroup
    for (int w = 0; w < 4; ++w) {
        const uint32_t n = big_n[w];
        if (n == 0) continue; // uniform per workgroup
        const int bb = wg_item() * 4 + w;
        const int l = find_level_by_band(g, bb);
        const int tiles_x = g.lv[l].tiles_x, bl = bb - g.lv[l].band_base;
        const int cut = level_cutoff[f * SLAMFE_MAX_LEVELS + l];
        const uint32_t offset = big_off[w];
        const size_t t0 = (size_t)f * g.total_tiles + g.lv[l].tile_base + (size_t)bl * tiles_x;
        if (tid == 0) nkeep = 0;
        __syncthreads();
        for (int tx0 = 0; tx0 < tiles_x; tx0 += 4) {
            uint32_t k[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tx = tx0 + u;
                ok[u] = tx < tiles_x && (uint32_t)tid < cand_cnt[t0 + (tx < tiles_x ? tx : 0)];
                k[u] = ok[u] ? cand[(t0 + tx) * SLAMFE_TILE_CAND + tid] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u] && (int)SLAMFE_KEY_S(k[u]) >= cut) lds[atomicAdd(&nkeep, 1u)] = k[u];
        }
        __syncthreads();
        const uint32_t nk = nkeep; // == n
        uint32_t np2 = 1;
        while (np2 < nk) np2 <<= 1;
        for (uint32_t i = nk + tid; i < np2; i += 256) lds[i] = 0xffffffffu;
        __syncthreads();
        for (uint32_t k = 2; k <= np2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < np2; i += 256) {
                    uint32_t ixj = i ^ j;
                    if (ixj > i) {
                        uint32_t a = lds[i], c = lds[ixj];
                        bool asc = (i & k) == 0;
                        if ((a > c) == asc) {
                            lds[i] = c;
                            lds[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
        for (uint32_t i = tid; i < nk; i += 256) {
            uint32_t o = offset + i;
            if (o < (uint32_t)max_kp) {
                kp_key[(size_t)slot * max_kp + o] = lds[i];
                kp_level[(size_t)slot * max_kp + o] = (uint32_t)l;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K5  orientation by intensity centroid (ICAngles inside cv::ORB).
//
// k_orient: a wavefront sums the moments of a keypoint's 749-pixel disc.
// k_finish: one lane per keypoint turns the moments into the angle
// (fastAtan2), its cos/sin for the descriptor, and the cv::KeyPoint record.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// A wavefront walks keypoints idx = wave, wave + nwaves, ...  The disc's 31 rows
// of 31 bytes (starting at cx-15) are cut into 8 dwords each, 248 (row, dword)
// tasks = 4 per lane.  Which bytes of a task lie inside the disc, their
// x-offsets u and the row offset v depend on the lane only, so they are turned
// once per wave into byte weights for v_dot4_u32_u8:
//   T += dot4(x, inside ? u + 32 : 0)   S += dot4(x, inside ? 1 : 0)
//   V += dot4(x, inside ? v + 16 : 0)   m10 = T - 32 S,  m01 = V - 16 S
// leaving three dot products per loaded dword for every keypoint.
__global__ __launch_bounds__(256) void k_orient(slamfe_dev_geom g, const uint8_t* __restrict__ pyr,
                                                int slot_base, int nslots,
                                                const uint32_t* __restrict__ kp_key,
                                                const uint32_t* __restrict__ kp_level,
                                                const int32_t* __restrict__ kp_count,
                                                int2* __restrict__ kp_mom, int max_kp, int nframes)
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
    // half-widths of the disc rows |v| = 0..15, one nibble each
    const unsigned long long umax = 0x3689abcddeeeffffULL;
    uint32_t wT[4], wV[4], wS[4];
    int trow[4], tcol[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int t = it * 64 + lane, row = t >> 3, dw = t & 7;
        const int v = row - 15, av = v < 0 ? -v : v;
        const int d = row < 31 ? (int)((umax >> (4 * av)) & 15ULL) : -1;
        uint32_t a = 0, b = 0, c = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = 4 * dw + k - 15;
            if (u >= -d && u <= d) {
                a |= (uint32_t)(u + 32) << (8 * k);
                b |= (uint32_t)(v + 16) << (8 * k);
                c |= 1u << (8 * k);
            }
        }
        wT[it] = a;
        wV[it] = b;
        wS[it] = c;
        trow[it] = row < 31 ? row : 0;
        tcol[it] = 4 * dw;
    }
    const uint8_t* frame = pyr + (size_t)f * g.frame_bytes;
    const uint32_t* keys = kp_key + (size_t)slot * max_kp;
    const uint32_t* levels = kp_level + (size_t)slot * max_kp;
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
        const uint8_t* base = frame + g.lv[l].off + (size_t)(cy - 15) * pitch + (cx - 15);
        uint32_t T = 0, V = 0, S = 0;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            uint32_t x = 0;
            if (it < 3 || lane < 56) x = load_u32_unaligned(base + trow[it] * pitch + tcol[it]);
            T = __builtin_amdgcn_udot4(x, wT[it], T, false);
            V = __builtin_amdgcn_udot4(x, wV[it], V, false);
            S = __builtin_amdgcn_udot4(x, wS[it], S, false);
        }
        const int m10 = wave_sum((int)T - 32 * (int)S);
        const int m01 = wave_sum((int)V - 16 * (int)S);
        if (lane == 0) kp_mom[(size_t)slot * max_kp + idx] = make_int2(m10, m01);
    }
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
    const float scale = g.lv[l].scale;
    const int cx = (int)SLAMFE_KEY_X(key), cy = (int)SLAMFE_KEY_Y(key);
    slamfe_keypoint k;
    if (mode == SLAMFE_MODE_ORB) {
        int2 m = kp_mom[e];
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
    kps[e] = k;
    float sn, cs;
    slamfe_sincos_deg(k.angle, &sn, &cs);
    kp_rot[e] = make_float2(cs, sn);
}

// ---------------------------------------------------------------------------
// K7  rBRIEF-256 (computeOrbDescriptors inside cv::ORB, reached from
//     desc->compute, Init.cc:78).  Lane i of a wavefront evaluates tests i,
//     64+i, 128+i, 192+i, so each wave ballot is 8 descriptor bytes.
// ---------------------------------------------------------------------------
#define DP_ROWS 37 /* samples reach +-18 px (|pattern| <= 13 * sqrt 2 = 18.4) */
#define DP_DW 10   /* 37 bytes from column cx-18, read as 10 dwords */

// A wavefront walks keypoints idx = wave, wave + nwaves, ...  For each one it
// stages the 37-row neighbourhood (rows of 40 bytes starting at cx-18,
// unaligned dword loads) in its own LDS slice and gathers the 512 samples from
// there.  Per-lane constants (the lane's four tests as floats, its staging
// tasks) are set up once per wave.  The rotation is evaluated on float pairs
// (x, y) = P * (a, b) + Q * (-b, a): the same products and one sum per
// coordinate as the scalar expressions x = px*a - py*b, y = px*b + py*a, so the
// bits are identical.  Rounding to the nearest integer (ties to even, as
// cvRound) is the addition of 1.5 * 2^23, which leaves the integer in the low
// mantissa bits.
__global__ __launch_bounds__(256) void k_describe(slamfe_dev_geom g, const uint8_t* __restrict__ blur,
                                                  int slot_base, int nslots,
                                                  const uint32_t* __restrict__ kp_key,
                                                  const uint32_t* __restrict__ kp_level,
                                                  const int32_t* __restrict__ kp_count,
                                                  const float2* __restrict__ kp_rot,
                                                  uint8_t* __restrict__ desc, int max_kp, int nframes)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    __shared__ uint32_t patch32[4][DP_ROWS * DP_DW];
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
    v2f P[4][2], Q[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int8_t* t = c_pattern + (j * 64 + lane) * 4;
        P[j][0] = (v2f){(float)t[0], (float)t[0]};
        Q[j][0] = (v2f){(float)t[1], (float)t[1]};
        P[j][1] = (v2f){(float)t[2], (float)t[2]};
        Q[j][1] = (v2f){(float)t[3], (float)t[3]};
    }
    constexpr int NIT = (DP_ROWS * DP_DW + 63) / 64;
    int trow[NIT], tcol[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = it * 64 + lane, row = t / DP_DW;
        trow[it] = row < DP_ROWS ? row : 0;
        tcol[it] = 4 * (t - row * DP_DW);
    }
    uint32_t* my = patch32[wv];
    const uint8_t* c = reinterpret_cast<const uint8_t*>(my) + 18 * (DP_DW * 4) + 18;
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
        const uint8_t* base = frame + g.lv[l].off + (size_t)(cy - 18) * pitch + (cx - 18);
        uint32_t r[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            r[it] = 0;
            if (it < NIT - 1 || it * 64 + lane < DP_ROWS * DP_DW)
                r[it] = load_u32_unaligned(base + trow[it] * pitch + tcol[it]);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (it < NIT - 1 || it * 64 + lane < DP_ROWS * DP_DW) my[it * 64 + lane] = r[it];
        // only this wavefront reads what it wrote: LDS operations of one wave are
        // ordered, the fences keep the compiler from moving accesses across
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const v2f AB = {rot.x, rot.y}, NBA = {-rot.y, rot.x};
        const v2f magic = {12582912.f, 12582912.f};
        unsigned long long word[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v2f pt = P[j][e] * AB + Q[j][e] * NBA; // (px*a - py*b, px*b + py*a)
                pt = pt + magic;
                const int ix = (short)__float_as_int(pt.x), iy = (short)__float_as_int(pt.y);
                v[e] = c[iy * (DP_DW * 4) + ix];
            }
            word[j] = __ballot(v[0] < v[1]);
        }
        if (lane < 4) {
            unsigned long long o = lane == 0 ? word[0] : lane == 1 ? word[1] : lane == 2 ? word[2] : word[3];
            reinterpret_cast<unsigned long long*>(desc + ((size_t)slot * max_kp + idx) * 32)[lane] = o;
        }
        // the next keypoint's staging must not overtake these LDS reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------
// K8  brute-force Hamming k=2 (BFMatcher::knnMatch, Init.cc:131) + ratio test
//     (Init.cc:146-147).
//
// A lane owns QPT query rows in registers (8 dwords each).  Train rows are
// read through wave-uniform addresses (scalar loads), so one comparison is
// 8 v_xor + 8 accumulating v_bcnt; the running top-2 is kept as packed keys
// (distance << 23 | train row), for which integer min *is* the (distance,
// index) order of knnMatch.  The train set is cut into chunks over gridDim.y;
// k_match_merge folds the per-chunk top-2.
// ---------------------------------------------------------------------------
// Frame-slot addressing for batched frame-vs-frame matching: pair z (blockIdx.z)
// matches slot (q_base + z) % nslots against slot (t_base + z) % nslots; rows
// and counts of slot s are at desc + s * cap * 32 and count + s.  nslots = 0
// means "plain pointers, one pair".
struct slamfe_pair_slots {
    int q_base, t_base, nslots;
};

// One 256-bit comparison and top-2 update: 8 v_xor + 8 chained accumulating
// v_bcnt + key pack + v_min + v_med3 = 19 VALU operations.
__device__ __forceinline__ void match_row(const uint32_t (&q)[8], const uint32_t (&t)[8], uint32_t j,
                                          uint32_t& b1, uint32_t& b2)
{
    uint32_t d, x0 = q[0] ^ t[0];
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(x0));
#pragma unroll
    for (int e = 1; e < 8; ++e) {
        uint32_t x = q[e] ^ t[e];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
    }
    uint32_t key = (d << SLAMFE_MKEY_SHIFT) | j;
    // b1 <= b2 always, so the new second best is the median of (key, b1, b2)
    uint32_t nb2;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(nb2) : "v"(key), "v"(b1), "v"(b2));
    b1 = min(key, b1);
    b2 = nb2;
}

// Train rows are staged through LDS in blocks of 128 rows (4 KiB, one 16-byte
// piece per thread, coalesced); every lane then reads the same row address
// (an LDS broadcast: no bank conflicts), two ds_read_b128 per row, so the row
// sits in 8 VGPRs while QPT comparisons use it.  The next block's global load
// is issued before the current block is consumed and written to the other LDS
// buffer afterwards.
#define MATCH_STAGE_ROWS 128

// Work decomposition: one workgroup = (pair, block of 256*QPT queries, chunk of
// chunk_rows train rows).  The grid is one-dimensional and sized for the
// capacity; a workgroup derives its work item from the *actual* row counts so
// that the busy workgroups are the first ones of the grid: workgroups are dealt
// round-robin to the 8 XCDs, and a grid whose idle workgroups recur with a
// period of 8 would leave whole XCDs empty.
struct slamfe_match_geo {
    int cap;        // row capacity of a frame slot / of the query array
    int chunk_rows; // train rows per workgroup
    int nchunks;    // chunks the partial buffer is laid out for
    int npairs;
    int part_sq, part_sc; // partial index = chunk * part_sc + query * part_sq (uint2 units)
};

template <int QPT>
__global__ __launch_bounds__(256) void k_match(const uint32_t* __restrict__ query, int nq_imm,
                                               const int32_t* __restrict__ nq_dev,
                                               const uint32_t* __restrict__ train, int nt_imm,
                                               const int32_t* __restrict__ nt_dev,
                                               uint32_t* __restrict__ part, slamfe_pair_slots ps,
                                               slamfe_match_geo mg)
{
    __shared__ uint4 stage[2][MATCH_STAGE_ROWS * 2];
    const int pair = (int)(blockIdx.x % (unsigned)mg.npairs);
    const int local = (int)(blockIdx.x / (unsigned)mg.npairs);
    if (ps.nslots) {
        int sq = (ps.q_base + pair) % ps.nslots, st = (ps.t_base + pair) % ps.nslots;
        query += (size_t)sq * mg.cap * 8;
        train += (size_t)st * mg.cap * 8;
        nq_dev += sq;
        nt_dev += st;
        part += (size_t)pair * mg.nchunks * mg.cap * 2;
    }
    const int nq = nq_dev ? min(*nq_dev, mg.cap) : nq_imm;
    const int nt = nt_dev ? min(*nt_dev, mg.cap) : nt_imm;
    const int real_q = (nq + 256 * QPT - 1) / (256 * QPT);
    const int real_c = max(1, (nt + mg.chunk_rows - 1) / mg.chunk_rows);
    if (local >= real_q * real_c) return;
    const int qb = local / real_c, chunk = local - qb * real_c;
    const int tid = threadIdx.x;
    const int qbase = qb * 256 * QPT + tid; // lane owns queries qbase, qbase + 256, ...
    const int j0 = chunk * mg.chunk_rows;
    const int j1 = min(nt, j0 + mg.chunk_rows);
    uint32_t q[QPT][8];
    uint32_t b1[QPT], b2[QPT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        int qi = min(qbase + i * 256, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 8);
        uint4 lo = p[0], hi = p[1];
        q[i][0] = lo.x; q[i][1] = lo.y; q[i][2] = lo.z; q[i][3] = lo.w;
        q[i][4] = hi.x; q[i][5] = hi.y; q[i][6] = hi.z; q[i][7] = hi.w;
        b1[i] = b2[i] = SLAMFE_MKEY_NONE;
    }
    if (j0 < j1) { // uniform per workgroup
        const uint4* t4 = reinterpret_cast<const uint4*>(train);
        const int nstages = (j1 - j0 + MATCH_STAGE_ROWS - 1) / MATCH_STAGE_ROWS;
        // piece `tid` of a stage = 16 bytes at row (tid >> 1), half (tid & 1)
        auto load_piece = [&](int st) -> uint4 {
            int row = j0 + st * MATCH_STAGE_ROWS + (tid >> 1);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < j1) v = t4[(size_t)row * 2 + (tid & 1)];
            return v;
        };
        stage[0][tid] = load_piece(0);
        __syncthreads();
        for (int st = 0; st < nstages; ++st) {
            const int cur = st & 1;
            uint4 nxt = make_uint4(0, 0, 0, 0);
            if (st + 1 < nstages) nxt = load_piece(st + 1);
            const int jb = j0 + st * MATCH_STAGE_ROWS;
            const int rows = min(MATCH_STAGE_ROWS, j1 - jb);
            const uint4* sp = stage[cur];
            int r = 0;
            for (; r + 4 <= rows; r += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint4 lo = sp[(r + u) * 2], hi = sp[(r + u) * 2 + 1];
                    const uint32_t tr[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int i = 0; i < QPT; ++i)
                        match_row(q[i], tr, (uint32_t)(jb + r + u), b1[i], b2[i]);
                }
            }
            for (; r < rows; ++r) {
                uint4 lo = sp[r * 2], hi = sp[r * 2 + 1];
                const uint32_t tr[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int i = 0; i < QPT; ++i) match_row(q[i], tr, (uint32_t)(jb + r), b1[i], b2[i]);
            }
            if (st + 1 < nstages) stage[cur ^ 1][tid] = nxt;
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        int qi = qbase + i * 256;
        if (qi < nq) {
            uint2 o = make_uint2(b1[i], b2[i]);
            reinterpret_cast<uint2*>(part)[(size_t)chunk * mg.part_sc + (size_t)qi * mg.part_sq] = o;
        }
    }
}

// ---------------------------------------------------------------------------
// K8m  the same brute-force Hamming k=2 on the matrix cores.
//
// With train bits a in {0, 1} and query bits mapped to b' = 1 - 2b in {+1, -1},
//   sum_k a_k * b'_k = popcount(a) - 2 a.b = hamming(a, b) - popcount(b),
// so one v_mfma_i32_32x32x32_i8 chain over the 256 bit positions (8 MFMAs) gives
// the exact distances of 32 train rows x 32 queries, minus a per-query constant
// that does not change the order.  In the result tile the query is the lane
// (column) and the 16 registers are train rows, so the running top-2 of a query
// stays in its lane: key = (dot + 256) << 22 | row-in-tile, one v_lshl_add per
// distance, then v_min / v_med3 as in the VALU kernel.  Per comparison that is 3
// VALU operations instead of 19; the 8 MFMAs of a tile take 256 cycles per SIMD.
//
// A workgroup = 256 queries (4 waves x 2 blocks of 32, expanded to +-1 bytes in
// 64 VGPRs per lane) x one chunk of train rows.  Train rows are read packed
// (32 bytes), expanded to 0/1 bytes into LDS (one dword -> 32 bytes per thread
// and tile, rows padded to 272 bytes so that the 16-byte fragment reads are
// conflict-free), double-buffered.  Row indices inside the keys are relative to
// the current tile (older entries are moved down by 32 per tile: 2 operations
// per tile instead of re-basing 16 constants) and made absolute at the end.
// ---------------------------------------------------------------------------
#define MM_TILE 32
#define MM_SUB 2 /* tiles per LDS stage (one workgroup barrier per stage) */
#define MM_ROW 272
#define MM_BIAS 8192u
#define MM_SHIFT 22
typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef int mm_v16i __attribute__((ext_vector_type(16)));

// bits pos .. pos+3 of x -> four bytes of 0 / 1 (bit i in byte i)
__device__ __forceinline__ uint32_t bits4_to_bytes(uint32_t x, int pos)
{
    return (((x >> pos) & 0xFu) * 0x00204081u) & 0x01010101u;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_match_mfma(const uint32_t* __restrict__ query, int nq_imm,
                                                    const int32_t* __restrict__ nq_dev,
                                                    const uint32_t* __restrict__ train, int nt_imm,
                                                    const int32_t* __restrict__ nt_dev,
                                                    uint32_t* __restrict__ part, slamfe_pair_slots ps,
                                                    slamfe_match_geo mg)
{
    __shared__ __attribute__((aligned(16))) uint8_t sA[2][MM_SUB][MM_TILE * MM_ROW];
    const int pair = (int)(blockIdx.x % (unsigned)mg.npairs);
    const int local = (int)(blockIdx.x / (unsigned)mg.npairs);
    if (ps.nslots) {
        int sq = (ps.q_base + pair) % ps.nslots, st = (ps.t_base + pair) % ps.nslots;
        query += (size_t)sq * mg.cap * 8;
        train += (size_t)st * mg.cap * 8;
        nq_dev += sq;
        nt_dev += st;
        part += (size_t)pair * mg.nchunks * mg.cap * 2;
    }
    const int nq = nq_dev ? min(*nq_dev, mg.cap) : nq_imm;
    const int nt = nt_dev ? min(*nt_dev, mg.cap) : nt_imm;
    const int real_q = (nq + 255) / 256;
    const int real_c = max(1, (nt + mg.chunk_rows - 1) / mg.chunk_rows);
    if (local >= real_q * real_c) return;
    const int qb256 = local / real_c, chunk = local - qb256 * real_c;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int j0 = chunk * mg.chunk_rows;
    const int j1 = min(nt, j0 + mg.chunk_rows);
    const int qfirst = qb256 * 256 + wv * 64 + r; // + 32 for the second block

    if (j0 >= j1) { // an empty train set: every query of the workgroup gets "none"
        for (int blk = 0; blk < 2; ++blk) {
            int qi = qfirst + 32 * blk;
            if (h == 0 && qi < nq)
                reinterpret_cast<uint2*>(part)[(size_t)chunk * mg.part_sc + (size_t)qi * mg.part_sq] =
                    make_uint2(SLAMFE_MKEY_NONE, SLAMFE_MKEY_NONE);
        }
        return;
    }

    // query fragments: lane (r, h) holds bits 16h .. 16h+15 of every dword of its
    // query row as +-1 bytes
    mm_v4i B[2][8];
    int pq[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int qi = min(qfirst + 32 * blk, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t qw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int pc = 0;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            pc += __popc(qw[kb]);
            const uint32_t half = (qw[kb] >> (16 * h)) & 0xffffu;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const uint32_t t = bits4_to_bytes(half, 4 * n);
                B[blk][kb][n] = (int)(((t << 8) - t) | 0x01010101u); // bit 1 -> 0xff (-1), bit 0 -> 0x01
            }
        }
        pq[blk] = pc;
    }

    // a stage = MM_SUB tiles; the thread's piece of tile u of a stage: dword
    // (tid & 7) of row (tid >> 3)
    auto load_bits = [&](int st, int u) -> uint32_t {
        const int row = j0 + (st * MM_SUB + u) * MM_TILE + (tid >> 3);
        return row < j1 ? train[(size_t)row * 8 + (tid & 7)] : 0u;
    };
    auto store_tile = [&](int buf, int u, uint32_t bits) {
        uint4 a, b;
        a.x = bits4_to_bytes(bits, 0);  a.y = bits4_to_bytes(bits, 4);
        a.z = bits4_to_bytes(bits, 8);  a.w = bits4_to_bytes(bits, 12);
        b.x = bits4_to_bytes(bits, 16); b.y = bits4_to_bytes(bits, 20);
        b.z = bits4_to_bytes(bits, 24); b.w = bits4_to_bytes(bits, 28);
        uint4* d = reinterpret_cast<uint4*>(&sA[buf][u][(tid >> 3) * MM_ROW + (tid & 7) * 32]);
        d[0] = a;
        d[1] = b;
    };
    const int ntiles = (j1 - j0 + MM_TILE - 1) / MM_TILE;
    const int nstages = (ntiles + MM_SUB - 1) / MM_SUB;
#pragma unroll
    for (int u = 0; u < MM_SUB; ++u) store_tile(0, u, load_bits(0, u));
    __syncthreads();

    // result register j of lane half h is train row (j & 3) + 8 (j >> 2) + 4 h of the tile
    uint32_t cj[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) cj[j] = (256u << MM_SHIFT) + MM_BIAS + (uint32_t)((j & 3) + 8 * (j >> 2) + 4 * h);
    uint32_t b1[2] = {0xffffffffu, 0xffffffffu}, b2[2] = {0xffffffffu, 0xffffffffu};

    for (int st = 0; st < nstages; ++st) {
        uint32_t nbits[MM_SUB];
#pragma unroll
        for (int u = 0; u < MM_SUB; ++u) nbits[u] = st + 1 < nstages ? load_bits(st + 1, u) : 0u;
#pragma unroll
        for (int u = 0; u < MM_SUB; ++u) {
            const int t = st * MM_SUB + u;
            if (t >= ntiles) break; // uniform
            const uint8_t* base = &sA[st & 1][u][r * MM_ROW + h * 16];
            mm_v16i acc0 = {0}, acc1 = {0};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const mm_v4i a = *reinterpret_cast<const mm_v4i*>(base + kb * 32);
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B[0][kb], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B[1][kb], acc1, 0, 0, 0);
            }
            // entries of earlier tiles move down by one tile
            b1[0] -= MM_TILE; b2[0] -= MM_TILE; b1[1] -= MM_TILE; b2[1] -= MM_TILE;
            const int rows = j1 - (j0 + t * MM_TILE); // valid rows of this tile (>= 32: all)
            if (rows >= MM_TILE) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    uint32_t k0 = ((uint32_t)acc0[j] << MM_SHIFT) + cj[j];
                    uint32_t k1 = ((uint32_t)acc1[j] << MM_SHIFT) + cj[j];
                    uint32_t n0, n1;
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n0) : "v"(k0), "v"(b1[0]), "v"(b2[0]));
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n1) : "v"(k1), "v"(b1[1]), "v"(b2[1]));
                    b1[0] = min(k0, b1[0]); b2[0] = n0;
                    b1[1] = min(k1, b1[1]); b2[1] = n1;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool ok = (j & 3) + 8 * (j >> 2) + 4 * h < rows;
                    uint32_t k0 = ok ? ((uint32_t)acc0[j] << MM_SHIFT) + cj[j] : 0xffffffffu;
                    uint32_t k1 = ok ? ((uint32_t)acc1[j] << MM_SHIFT) + cj[j] : 0xffffffffu;
                    uint32_t n0, n1;
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n0) : "v"(k0), "v"(b1[0]), "v"(b2[0]));
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n1) : "v"(k1), "v"(b1[1]), "v"(b2[1]));
                    b1[0] = min(k0, b1[0]); b2[0] = n0;
                    b1[1] = min(k1, b1[1]); b2[1] = n1;
                }
            }
        }
        if (st + 1 < nstages) {
#pragma unroll
            for (int u = 0; u < MM_SUB; ++u) store_tile((st + 1) & 1, u, nbits[u]);
        }
        __syncthreads();
    }

    // the two lane halves hold disjoint train rows of the same query: merge, then
    // make distance and row absolute
    const uint32_t last = (uint32_t)(ntiles - 1) * MM_TILE;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)b1[blk], 32), o2 = (uint32_t)__shfl_xor((int)b2[blk], 32);
        const uint32_t m1 = min(b1[blk], o1);
        const uint32_t m2 = min(max(b1[blk], o1), min(b2[blk], o2));
        const int qi = qfirst + 32 * blk;
        if (h == 0 && qi < nq) {
            auto conv = [&](uint32_t k) -> uint32_t {
                if (k >= 0xC0000000u) return SLAMFE_MKEY_NONE;
                const uint32_t d = (k >> MM_SHIFT) - 256u + (uint32_t)pq[blk];
                const uint32_t row = (k & ((1u << MM_SHIFT) - 1u)) - MM_BIAS + last + (uint32_t)j0;
                return (d << SLAMFE_MKEY_SHIFT) | row;
            };
            reinterpret_cast<uint2*>(part)[(size_t)chunk * mg.part_sc + (size_t)qi * mg.part_sq] =
                make_uint2(conv(m1), conv(m2));
        }
    }
}

__device__ __forceinline__ void top2_push(uint32_t k, uint32_t& b1, uint32_t& b2)
{
    uint32_t mx = max(k, b1);
    b1 = min(k, b1);
    b2 = min(mx, b2);
}

__device__ __forceinline__ void match_emit(uint32_t b1, uint32_t b2, int qi, float ratio,
                                           slamfe_match2* __restrict__ out,
                                           uint64_t* __restrict__ out_keys, int64_t train_base)
{
    const uint32_t mask = (1u << SLAMFE_MKEY_SHIFT) - 1u;
    if (out_keys) {
        uint64_t k0 = ~0ULL, k1 = ~0ULL;
        if (b1 != SLAMFE_MKEY_NONE)
            k0 = ((uint64_t)(b1 >> SLAMFE_MKEY_SHIFT) << 32) | (uint64_t)(train_base + (b1 & mask));
        if (b2 != SLAMFE_MKEY_NONE)
            k1 = ((uint64_t)(b2 >> SLAMFE_MKEY_SHIFT) << 32) | (uint64_t)(train_base + (b2 & mask));
        out_keys[(size_t)qi * 2] = k0;
        out_keys[(size_t)qi * 2 + 1] = k1;
    }
    if (out) {
        slamfe_match2 m;
        m.idx0 = b1 != SLAMFE_MKEY_NONE ? (int32_t)(b1 & mask) : -1;
        m.idx1 = b2 != SLAMFE_MKEY_NONE ? (int32_t)(b2 & mask) : -1;
        m.d0 = (uint16_t)(b1 != SLAMFE_MKEY_NONE ? b1 >> SLAMFE_MKEY_SHIFT : 0);
        m.d1 = (uint16_t)(b2 != SLAMFE_MKEY_NONE ? b2 >> SLAMFE_MKEY_SHIFT : 0);
        m.nfound = (uint8_t)((m.idx0 >= 0) + (m.idx1 >= 0));
        m.ratio_pass = (uint8_t)(m.nfound == 2 && (float)m.d0 < ratio * (float)m.d1);
        m.reserved[0] = m.reserved[1] = 0;
        out[qi] = m;
    }
}

// Fold per-chunk top-2 keys.  Output either match rows (ratio test applied) or
// 64-bit shard keys (distance << 32 | global train index).
// LPQ = lanes per query: 1 (a thread walks the few chunks of a frame pair) or
// 64 (a wavefront strides over the hundreds of chunks of a large train set and
// folds with a shuffle butterfly; partials are then laid out query-major so the
// wave reads contiguous keys).
template <int LPQ>
__global__ __launch_bounds__(256) void k_match_merge(const uint32_t* __restrict__ part, int nq_imm,
                                                     const int32_t* __restrict__ nq_dev,
                                                     const int32_t* __restrict__ nt_dev, int nt_imm,
                                                     float ratio, slamfe_match2* __restrict__ out,
                                                     uint64_t* __restrict__ out_keys,
                                                     int64_t train_base, slamfe_pair_slots ps,
                                                     slamfe_match_geo mg)
{
    const int pair = blockIdx.y;
    if (ps.nslots) {
        int sq = (ps.q_base + pair) % ps.nslots, st = (ps.t_base + pair) % ps.nslots;
        nq_dev += sq;
        nt_dev += st;
        part += (size_t)pair * mg.nchunks * mg.cap * 2;
        out += (size_t)pair * mg.cap;
    }
    const int nq = nq_dev ? min(*nq_dev, mg.cap) : nq_imm;
    const int nt = nt_dev ? min(*nt_dev, mg.cap) : nt_imm;
    const int used = max(1, (nt + mg.chunk_rows - 1) / mg.chunk_rows);
    const uint2* p2 = reinterpret_cast<const uint2*>(part);
    uint32_t b1 = SLAMFE_MKEY_NONE, b2 = SLAMFE_MKEY_NONE;
    if (LPQ == 1) {
        int qi = blockIdx.x * 256 + threadIdx.x;
        if (qi >= nq) return;
        for (int c = 0; c < used; ++c) {
            uint2 p = p2[(size_t)c * mg.part_sc + (size_t)qi * mg.part_sq];
            top2_push(p.x, b1, b2);
            top2_push(p.y, b1, b2);
        }
        match_emit(b1, b2, qi, ratio, out, out_keys, train_base);
    } else {
        int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (qi >= nq) return;
        for (int c = lane; c < used; c += 64) {
            uint2 p = p2[(size_t)c * mg.part_sc + (size_t)qi * mg.part_sq];
            top2_push(p.x, b1, b2);
            top2_push(p.y, b1, b2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            uint32_t o1 = (uint32_t)__shfl_xor((int)b1, o), o2 = (uint32_t)__shfl_xor((int)b2, o);
            uint32_t mx = max(b1, o1);
            b1 = min(b1, o1);
            b2 = min(mx, min(b2, o2));
        }
        if (lane == 0) match_emit(b1, b2, qi, ratio, out, out_keys, train_base);
    }
}

// Merge key sets of several shards ([shard][nq][2], each ascending).
__global__ __launch_bounds__(256) void k_merge_shard_keys(const uint64_t* __restrict__ keys,
                                                          int nshards, int nq, float ratio,
                                                          slamfe_match2* __restrict__ out)
{
    int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint64_t b1 = ~0ULL, b2 = ~0ULL;
    for (int s = 0; s < nshards; ++s)
        for (int e = 0; e < 2; ++e) {
            uint64_t k = keys[((size_t)s * nq + qi) * 2 + e];
            uint64_t mx = k > b1 ? k : b1;
            b1 = k < b1 ? k : b1;
            b2 = mx < b2 ? mx : b2;
        }
    slamfe_match2 m;
    m.idx0 = b1 != ~0ULL ? (int32_t)(b1 & 0xffffffffULL) : -1;
    m.idx1 = b2 != ~0ULL ? (int32_t)(b2 & 0xffffffffULL) : -1;
    m.d0 = (uint16_t)(b1 != ~0ULL ? b1 >> 32 : 0);
    m.d1 = (uint16_t)(b2 != ~0ULL ? b2 >> 32 : 0);
    m.nfound = (uint8_t)((b1 != ~0ULL) + (b2 != ~0ULL));
    m.ratio_pass = (uint8_t)(m.nfound == 2 && (float)m.d0 < ratio * (float)m.d1);
    m.reserved[0] = m.reserved[1] = 0;
    out[qi] = m;
}

// ---------------------------------------------------------------------------
// K9  search by projection (Init.cc:229-259).
//
// Pass 1 (parallel, one workgroup per inlier): test every visible map point
// against the inlier's keypoint -- pixel distance gate first (Init.cc:389-391
// arithmetic: float differences, double squares and sqrt, narrowed to float),
// then Hamming distance against the device-resident map descriptor row -- and
// keep the PROJ_K best candidates as keys (hamming << 20 | position), sorted.
// Pass 2 (one wavefront, in inlier order): take the first candidate not yet
// consumed; that is exactly the reference's "minimum over the remaining
// points, then erase".  An inlier whose list overflowed and ran dry is
// re-scanned against all points by the whole wave.
// ---------------------------------------------------------------------------
#define PROJ_K 32
#define PROJ_LDS 2048

// The pixel-distance gate of Init.cc:389-391 and Init.cc:242: float differences,
// squares and sum in double, square root narrowed to float, compared with the
// radius.  Root and narrowing are monotone, so the test is dd <= gate_dd with
// gate_dd = the largest double whose narrowed root does not exceed the radius
// (found on the host by bisection over the doubles, slamfe_proj_gate_threshold):
// same decisions, no square root per pair.
__device__ __forceinline__ bool proj_gate(float ux, float uy, float kx, float ky, double gate_dd)
{
    float dx = ux - kx, dy = uy - ky;
    double dd = (double)dx * (double)dx + (double)dy * (double)dy;
    return dd <= gate_dd;
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

__global__ __launch_bounds__(256) void k_proj_candidates(
    const slamfe_keypoint* __restrict__ kps, const uint32_t* __restrict__ fdesc, int nkp,
    const int32_t* __restrict__ kp_idx, int n_inl, const float2* __restrict__ uv,
    const int32_t* __restrict__ point_ids, int R, const uint32_t* __restrict__ map_desc,
    double gate_dd, int max_hamming, uint32_t* __restrict__ cand_out, int32_t* __restrict__ cand_n)
{
    __shared__ uint32_t lds[PROJ_LDS];
    __shared__ uint32_t cnt;
    const int i = blockIdx.x, tid = threadIdx.x;
    const int idx = kp_idx[i];
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (idx >= 0 && idx < nkp) {
        const float kx = kps[idx].x, ky = kps[idx].y;
        const uint32_t* qd = fdesc + (size_t)idx * 8;
        // four positions per step so that the loads are in flight together
        for (int j0 = tid; j0 < R; j0 += 4 * 256) {
            float2 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 256;
                p[u] = j < R ? uv[j] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 256;
                if (j < R && proj_gate(p[u].x, p[u].y, kx, ky, gate_dd)) {
                    int h = hamming256_dev(map_desc + (size_t)point_ids[j] * 8, qd);
                    if (h <= max_hamming) {
                        uint32_t pos = atomicAdd(&cnt, 1u);
                        if (pos < PROJ_LDS) lds[pos] = ((uint32_t)h << 20) | (uint32_t)j;
                    }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t n = cnt;
    if (n > PROJ_LDS) { // too dense to rank here: leave it to the serial pass
        if (tid == 0) cand_n[i] = -1;
        return;
    }
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (uint32_t k = n + tid; k < np2; k += 256) lds[k] = 0xffffffffu;
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t e = tid; e < np2; e += 256) {
                uint32_t exj = e ^ j;
                if (exj > e) {
                    uint32_t a = lds[e], c = lds[exj];
                    bool asc = (e & k) == 0;
                    if ((a > c) == asc) {
                        lds[e] = c;
                        lds[exj] = a;
                    }
                }
            }
            __syncthreads();
        }
    if (tid < PROJ_K && tid < n) cand_out[(size_t)i * PROJ_K + tid] = lds[tid];
    // cand_n: number of stored candidates; bit 30 set if more passed the gates
    if (tid == 0) cand_n[i] = (int32_t)(n < PROJ_K ? n : (PROJ_K | (n > PROJ_K ? (1 << 30) : 0)));
}

#define PROJ_CHUNK 64 /* inliers per staged chunk */
#define PROJ_STRIDE (PROJ_K + 1) /* LDS words per staged list: odd, so lanes hit different banks */
#define PROJ_CLAIM 4096          /* entries of a proposal table */

__global__ __launch_bounds__(256) void k_proj_resolve(
    const slamfe_keypoint* __restrict__ kps, const uint32_t* __restrict__ fdesc, int nkp,
    const int32_t* __restrict__ kp_idx, int n_inl, const float2* __restrict__ uv,
    const int32_t* __restrict__ point_ids, int R, const uint32_t* __restrict__ map_desc,
    double gate_dd, int max_hamming, const uint32_t* __restrict__ cand,
    const int32_t* __restrict__ cand_n, int32_t* __restrict__ matched)
{
    // One workgroup of four waves.  Wave 0 resolves the inliers in order, 64 at
    // a time; waves 1-3 meanwhile copy the next chunk of 64 candidate lists from
    // global memory into the other LDS buffer.  LDS: two buffers of 64 lists +
    // 64 list headers, 64 answers, then one bit per visible point.
    //
    // The reference's loop is serial: inlier i takes its best point among those
    // no earlier inlier took.  Lane l of wave 0 holds inlier l of the chunk and
    // proposes the first candidate of its (sorted) list that is not taken yet.
    // If the proposals of lanes 0..k-1 are pairwise different they are exactly
    // what the serial loop would choose (each is the best of a superset of what
    // the serial loop would see, and none of them is removed by an earlier
    // lane), so that whole prefix is final at once; lane k, the first one whose
    // proposal an earlier lane also made, proposes again in the next round.
    // Contested points are rare, so a chunk usually takes one or two rounds
    // instead of 64 dependent steps.
    extern __shared__ uint32_t plds[];
    const int BUF = PROJ_CHUNK * PROJ_STRIDE + PROJ_CHUNK;
    int32_t* res = reinterpret_cast<int32_t*>(plds + 2 * BUF);
    uint32_t* claims = plds + 2 * BUF + PROJ_CHUNK; // two proposal tables, one per chunk parity
    uint32_t* taken = claims + 2 * PROJ_CLAIM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = tid; w < (R + 31) / 32; w += 256) taken[w] = 0;
    for (int w = tid; w < 2 * PROJ_CLAIM; w += 256) claims[w] = 0;

    auto load_chunk = [&](int c0, int buf, int t0, int nt) {
        // threads t0 .. t0+nt-1 of the workgroup copy chunk c0 into buffer buf
        const int nc = min(PROJ_CHUNK, n_inl - c0);
        if (nc <= 0) return;
        uint32_t* st = plds + buf * BUF;
        int32_t* sn = reinterpret_cast<int32_t*>(st + PROJ_CHUNK * PROJ_STRIDE);
        const int t = tid - t0;
        if (t < PROJ_CHUNK) sn[t] = t < nc ? cand_n[c0 + t] : 0;
        const uint32_t* src = cand + (size_t)c0 * PROJ_K;
#pragma unroll 4
        for (int e = t; e < nc * PROJ_K; e += nt) st[(e / PROJ_K) * PROJ_STRIDE + (e % PROJ_K)] = src[e];
    };

    load_chunk(0, 0, 0, 256);
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < n_inl; c0 += PROJ_CHUNK, buf ^= 1) {
        const int nc = min(PROJ_CHUNK, n_inl - c0);
        if (wave != 0) {
            load_chunk(c0 + PROJ_CHUNK, buf ^ 1, 64, 192);
            // and clear the proposal table the next chunk will use
            for (int w = tid - 64; w < PROJ_CLAIM; w += 192) claims[(buf ^ 1) * PROJ_CLAIM + w] = 0;
        } else {
            const uint32_t* stage = plds + buf * BUF;
            uint32_t* claim = claims + buf * PROJ_CLAIM;
            uint32_t round = 0;
            const int32_t* stage_n = reinterpret_cast<const int32_t*>(stage + PROJ_CHUNK * PROJ_STRIDE);
            volatile uint32_t* vt = taken;
            // list header: number of stored candidates; bit 30: more passed the gates
            // than were stored; negative: never ranked (too many points for the sort)
            const int cn = lane < nc ? stage_n[lane] : 0;
            const int nl = cn < 0 ? 0 : (cn & 0xffff);
            const bool overflow = cn < 0 || (cn & (1 << 30));
            const uint32_t* mylist = stage + lane * PROJ_STRIDE;
            int pos = 0, result = -1;
            bool done = lane >= nc;
            while (__ballot(!done)) { // every round finishes at least the first open lane
                int prop = -1; // position j of the proposed point
                bool dry = false;
                if (!done) {
                    while (pos < nl) {
                        const uint32_t j = mylist[pos] & 0xfffffu;
                        if (!((vt[j >> 5] >> (j & 31)) & 1u)) {
                            prop = (int)j;
                            break;
                        }
                        ++pos;
                    }
                    dry = prop < 0 && overflow; // the stored list ran out but more candidates exist
                }
                // Did an earlier open lane propose the same point?  Every proposal is
                // entered in a hash table with atomicMax on (round, 63 - lane, point):
                // the current round beats stale entries, the lowest lane wins its
                // slot, and a lane that does not find its own entry stops.  Two
                // different points in one slot stop the later lane needlessly, which
                // only shortens this round's prefix.
                ++round; // at most 64 rounds per chunk: each one finishes a lane
                bool stop = dry;
                if (prop >= 0) {
                    const uint32_t mine = (round << 26) | ((uint32_t)(63 - lane) << 20) | (uint32_t)prop;
                    const uint32_t slot = ((uint32_t)prop * 2654435761u) >> 20; // 12 bits
                    atomicMax(&claim[slot], mine);
                    wave_lds_fence();
                    stop = ((volatile uint32_t*)claim)[slot] != mine;
                } else {
                    wave_lds_fence();
                }
                const unsigned long long mstop = __ballot(stop && !done);
                const int k = mstop ? __ffsll((long long)mstop) - 1 : 64;
                if (!done && lane < k) {
                    result = prop;
                    if (prop >= 0) atomicOr(&taken[prop >> 5], 1u << (prop & 31));
                    done = true;
                }
                if (k < 64 && __builtin_amdgcn_readlane((int)dry, k)) { // uniform
                    // inlier k has to look at every point again: the whole wave does it
                    const int i = c0 + k;
                    const int idx = kp_idx[i];
                    uint32_t best = 0xffffffffu;
                    if (idx >= 0 && idx < nkp) {
                        const float kx = kps[idx].x, ky = kps[idx].y;
                        const uint32_t* qd = fdesc + (size_t)idx * 8;
                        for (int j = lane; j < R; j += 64) {
                            if ((vt[j >> 5] >> (j & 31)) & 1u) continue;
                            float2 p = uv[j];
                            if (!proj_gate(p.x, p.y, kx, ky, gate_dd)) continue;
                            int h = hamming256_dev(map_desc + (size_t)point_ids[j] * 8, qd);
                            if (h <= max_hamming) best = min(best, ((uint32_t)h << 20) | (uint32_t)j);
                        }
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1)
                        best = min(best, (uint32_t)__shfl_xor((int)best, o));
                    const int r = best != 0xffffffffu ? (int)(best & 0xfffffu) : -1;
                    if (lane == k) {
                        result = r;
                        if (r >= 0) atomicOr(&taken[r >> 5], 1u << (r & 31));
                        done = true;
                    }
                }
                // the next round's reads of `taken` come after these updates: LDS
                // operations of one wave execute in order
                wave_lds_fence();
            }
            if (lane < nc) res[lane] = result;
        }
        __syncthreads();
        if (tid < nc) {
            int r = res[tid];
            matched[c0 + tid] = r >= 0 ? point_ids[r] : -1;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// grid (FX, items, FZ) of the per-frame kernels; see wg_frame()
static dim3 frame_grid(int items, int nframes)
{
    int fx = nframes >= 8 ? 8 : 1;
    return dim3((unsigned)fx, (unsigned)items, (unsigned)((nframes + fx - 1) / fx));
}

#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

int slamfe_launch_ingest(hipStream_t s, const uint8_t* d_in, int w, int h, int stride,
                         size_t frame_stride, int channels, int nframes, uint8_t* pyr,
                         uint32_t frame_bytes, int pitch)
{
    dim3 grid((w + 1023) / 1024, (h + 3) / 4, nframes);
    int aligned16 = ((uintptr_t)d_in % 16 == 0) && stride % 16 == 0 && frame_stride % 16 == 0;
    hipLaunchKernelGGL(k_ingest, grid, dim3(256), 0, s, d_in, w, h, stride, frame_stride, channels,
                       pyr, frame_bytes, pitch, aligned16);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_ingest_scaled(hipStream_t s, const uint8_t* d_in, int stride, const uint32_t* xt,
                                const uint32_t* yt, int dw, int dh, uint8_t* gray, int gpitch,
                                uint8_t* bgr_out, int bpitch)
{
    dim3 grid((dw + 255) / 256, (dh + 3) / 4);
    hipLaunchKernelGGL(k_ingest_scaled, grid, dim3(256), 0, s, d_in, stride,
                       reinterpret_cast<const uint2*>(xt), reinterpret_cast<const uint2*>(yt), dw, dh,
                       gray, gpitch, bgr_out, bpitch);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_pyramid(hipStream_t s, const slamfe_dev_geom& g, int nframes, uint8_t* pyr,
                          const uint32_t* taps)
{
    for (int l = 1; l < g.nlevels; ++l) {
        hipLaunchKernelGGL(k_resize, frame_grid((g.lv[l].h + 4 * RS_ROWS - 1) / (4 * RS_ROWS), nframes), dim3(256), 0, s, g, l,
                           nframes, pyr, taps);
        LAUNCH_CHECK();
    }
    return 0;
}

int slamfe_launch_fast_blur(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                            uint8_t* blur, int thr, int edge, uint32_t* cand, uint32_t* cand_cnt)
{
    hipLaunchKernelGGL(k_fast_blur, frame_grid(g.total_tiles, nframes), dim3(256), 0, s, g, pyr, blur,
                       nframes, thr, edge, cand, cand_cnt);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_select(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint32_t* cand,
                         const uint32_t* cand_cnt, uint32_t* band_hist, uint32_t* band_off,
                         uint32_t* level_kept, int32_t* level_cutoff, int slot_base, int nslots,
                         uint32_t* kp_key, uint32_t* kp_level, int32_t* kp_count, int max_kp,
                         size_t sort_lds_bytes)
{
    hipLaunchKernelGGL(k_band_hist, frame_grid((g.total_bands + 3) / 4, nframes), dim3(256), 0, s, g,
                       cand, cand_cnt, band_hist, nframes);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_level_cut, frame_grid(g.nlevels, nframes), dim3(256), 0, s, g, band_hist,
                       band_off, level_kept, level_cutoff, nframes);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_select_write, frame_grid((g.total_bands + 3) / 4, nframes), dim3(256), sort_lds_bytes,
                       s, g, cand, cand_cnt, band_off, level_kept, level_cutoff, slot_base, nslots,
                       kp_key, kp_level, kp_count, max_kp, nframes);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_orient(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                         int slot_base, int nslots, const uint32_t* kp_key, const uint32_t* kp_level,
                         const int32_t* kp_count, int32_t* kp_mom, slamfe_keypoint* kps, float* kp_rot,
                         int max_kp, int mode, int patch_size)
{
    if (mode == SLAMFE_MODE_ORB) {
        // about four keypoints per wavefront when the level quotas are met exactly;
        // the kernel loops, so any count up to max_kp is covered
        long long est = 0;
        for (int l = 0; l < g.nlevels; ++l) est += g.lv[l].quota;
        if (est < 64 || est > max_kp) est = max_kp;
        hipLaunchKernelGGL(k_orient, frame_grid((int)((est + 15) / 16), nframes), dim3(256), 0, s, g, pyr,
                           slot_base, nslots, kp_key, kp_level, kp_count,
                           reinterpret_cast<int2*>(kp_mom), max_kp, nframes);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_finish, frame_grid((max_kp + 255) / 256, nframes), dim3(256), 0, s, g,
                       slot_base, nslots, kp_key, kp_level, kp_count,
                       reinterpret_cast<const int2*>(kp_mom), kps, reinterpret_cast<float2*>(kp_rot),
                       max_kp, mode, patch_size, nframes);
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

int slamfe_match_plan(int nq_max, int nt_max, int npairs, int kernel, int* qpt, int* chunk_rows,
                      int* nchunks)
{
    if (npairs < 1) npairs = 1;
    long long qblocks = (nq_max + 255) / 256;
    if (qblocks < 1) qblocks = 1;
    const long long have = qblocks * npairs;
    int rows, q;
    if (kernel == SLAMFE_MATCH_MATRIX || (kernel == SLAMFE_MATCH_AUTO && nq_max >= 128)) {
        // matrix-core kernel: a workgroup expands its 256 queries once (about the
        // cost of three tiles), so chunks are long: 2048 rows for large train sets,
        // 1024 for frame pairs (measured best on MI355X), less only when that
        // would leave fewer than ~512 workgroups (a single frame pair)
        q = 64;
        rows = nt_max > 8192 ? 2048 : 1024;
        while (rows > 256 && have * ((nt_max + rows - 1) / rows) < 512) rows >>= 1;
    } else {
        // few queries: the VALU kernel, one query row per lane.  Measured on MI355X:
        // 256- or 512-row chunks beat 2 or 4 rows per lane by 7-13 %, and chunk
        // sizes that are not a power of two ran up to 20 % slower.
        q = 1;
        rows = 256;
        if (have * ((nt_max + 255) / 256) < 1024) rows = 128;
        else if (have * ((nt_max + 511) / 512) >= 8192) rows = 512;
    }
    int nc = (nt_max + rows - 1) / rows;
    if (nc < 1) nc = 1;
    *qpt = q;
    *chunk_rows = rows;
    *nchunks = nc;
    return 0;
}

int slamfe_launch_match(hipStream_t s, const uint8_t* d_query, int nq, const int32_t* d_nq,
                        const uint8_t* d_train, int nt, const int32_t* d_nt, int cap, int qpt,
                        int chunk_rows, int nchunks, uint32_t* part, float ratio,
                        slamfe_match2* d_out, uint64_t* d_keys, int64_t train_base, int npairs,
                        int q_slot_base, int t_slot_base, int nslots)
{
    int nq_grid = d_nq ? cap : nq;
    if (nq_grid <= 0 || npairs <= 0) return 0;
    slamfe_pair_slots ps = {q_slot_base, t_slot_base, nslots};
    const bool wide = nchunks > 32; // many chunks: wave-per-query merge, query-major partials
    slamfe_match_geo mg;
    mg.cap = cap;
    mg.chunk_rows = chunk_rows;
    mg.nchunks = nchunks;
    mg.npairs = npairs;
    mg.part_sq = wide ? nchunks : 1;
    mg.part_sc = wide ? 1 : cap;
    int qblocks = (nq_grid + 255) / 256; // both kernels: 256 queries per workgroup
    dim3 grid((unsigned)qblocks * (unsigned)nchunks * (unsigned)npairs);
    const uint32_t* q32 = reinterpret_cast<const uint32_t*>(d_query);
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(d_train);
    if (qpt == 64)
        hipLaunchKernelGGL(k_match_mfma, grid, dim3(256), 0, s, q32, nq, d_nq, t32, nt, d_nt, part, ps, mg);
    else if (qpt == 1)
        hipLaunchKernelGGL(k_match<1>, grid, dim3(256), 0, s, q32, nq, d_nq, t32, nt, d_nt, part, ps, mg);
    else
        return (int)hipErrorInvalidValue;
    LAUNCH_CHECK();
    if (wide)
        hipLaunchKernelGGL(k_match_merge<64>, dim3((nq_grid + 3) / 4, npairs), dim3(256), 0, s, part, nq,
                           d_nq, d_nt, nt, ratio, d_out, d_keys, train_base, ps, mg);
    else
        hipLaunchKernelGGL(k_match_merge<1>, dim3((nq_grid + 255) / 256, npairs), dim3(256), 0, s, part,
                           nq, d_nq, d_nt, nt, ratio, d_out, d_keys, train_base, ps, mg);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_launch_merge_shard_keys(hipStream_t s, const uint64_t* d_keys, int nshards, int nq,
                                   float ratio, slamfe_match2* d_out)
{
    if (nq <= 0) return 0;
    hipLaunchKernelGGL(k_merge_shard_keys, dim3((nq + 255) / 256), dim3(256), 0, s, d_keys, nshards,
                       nq, ratio, d_out);
    LAUNCH_CHECK();
    return 0;
}

// Largest double dd with (float)sqrt(dd) <= radius, or -1 if there is none: the
// predicate is monotone in dd, so bisect over the bit patterns of the
// non-negative doubles (their integer order is their numeric order).
double slamfe_proj_gate_threshold(float radius)
{
    auto ok = [&](uint64_t bits) {
        double dd;
        __builtin_memcpy(&dd, &bits, sizeof dd);
        return (float)sqrt(dd) <= radius;
    };
    if (!(radius >= 0.f) || !ok(0)) return -1.0; // negative or NaN radius: nothing passes
    uint64_t lo = 0, hi = 0x7ff0000000000000ULL; // +infinity
    if (ok(hi)) lo = hi;
    while (hi - lo > 1) { // invariant: ok(lo), !ok(hi)
        uint64_t mid = lo + (hi - lo) / 2;
        if (ok(mid)) lo = mid;
        else hi = mid;
    }
    double dd;
    __builtin_memcpy(&dd, &lo, sizeof dd);
    return dd;
}

int slamfe_launch_projection(hipStream_t s, const slamfe_keypoint* kps, const uint8_t* fdesc, int nkp,
                             const int32_t* kp_idx, int n_inl, const float* uv,
                             const int32_t* point_ids, int R, const uint8_t* map_desc, float radius,
                             int max_hamming, uint32_t* cand, int32_t* cand_n, int32_t* matched)
{
    if (n_inl <= 0) return 0;
    const double gate_dd = slamfe_proj_gate_threshold(radius);
    hipLaunchKernelGGL(k_proj_candidates, dim3(n_inl), dim3(256), 0, s, kps,
                       reinterpret_cast<const uint32_t*>(fdesc), nkp, kp_idx, n_inl,
                       reinterpret_cast<const float2*>(uv), point_ids, R,
                       reinterpret_cast<const uint32_t*>(map_desc), gate_dd, max_hamming, cand, cand_n);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(256),
                       (2 * (PROJ_CHUNK * PROJ_STRIDE + PROJ_CHUNK) + PROJ_CHUNK + 2 * PROJ_CLAIM) * 4 +
                           ((size_t)R + 31) / 32 * 4 + 4,
                       s, kps,
                       reinterpret_cast<const uint32_t*>(fdesc), nkp, kp_idx, n_inl,
                       reinterpret_cast<const float2*>(uv), point_ids, R,
                       reinterpret_cast<const uint32_t*>(map_desc), gate_dd, max_hamming, cand, cand_n,
                       matched);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Map-point reprojection, Init.cc:203-218.  M = K4x4 * inverse(pose) is formed
// once on the host (the reference recomputes the same inverse for every point);
// the three kernels apply the depth gate, project, apply the image gate and
// compact the survivors in ascending map order.
//   k_reproject_test    one lane per point: gates + projection, ballot per wave
//   k_reproject_scan    one workgroup: exclusive scan of the per-wave counts
//   k_reproject_scatter ordered scatter of (u, v) and the point index
// Float arithmetic: products and sums in float, left to right, no fused
// multiply-add; pt /= s multiplies by the float value of 1.0/s taken in double.
// ---------------------------------------------------------------------------
struct slamfe_proj_mat { float m[16]; };

__global__ __launch_bounds__(256) void k_reproject_test(const float* __restrict__ xyz, int npoints,
                                                        slamfe_proj_mat M, float th, float W, float H,
                                                        float2* __restrict__ uv_all,
                                                        uint64_t* __restrict__ wave_mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool vis = false;
    float2 out = make_float2(0.f, 0.f);
    if (i < npoints) {
        const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
        if (!(z < th || fabs((double)(z - th)) > 20.0)) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[r] = M.m[r * 4] * x + M.m[r * 4 + 1] * y + M.m[r * 4 + 2] * z + M.m[r * 4 + 3] * 1.f;
            float sc = (float)(1.0 / (double)p[3]);
            p[0] = p[0] * sc;
            p[1] = p[1] * sc;
            p[2] = p[2] * sc;
            sc = (float)(1.0 / (double)p[2]);
            p[0] = p[0] * sc;
            p[1] = p[1] * sc;
            vis = !(p[0] <= 0.f || p[0] >= W || p[1] <= 0.f || p[1] >= H);
            out = make_float2(p[0], p[1]);
        }
        uv_all[i] = out;
    }
    const uint64_t b = __ballot(vis);
    if ((threadIdx.x & 63) == 0) wave_mask[(blockIdx.x * 256 + threadIdx.x) >> 6] = b;
}

__global__ __launch_bounds__(1024) void k_reproject_scan(const uint64_t* __restrict__ wave_mask, int nwaves,
                                                         int32_t* __restrict__ wave_off,
                                                         int32_t* __restrict__ total)
{
    __shared__ int part[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < nwaves; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nwaves ? __popcll(wave_mask[i]) : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) part[wv] = incl;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < wv; ++k) before += part[k];
        if (i < nwaves) wave_off[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_reproject_scatter(const float2* __restrict__ uv_all, int npoints,
                                                           const uint64_t* __restrict__ wave_mask,
                                                           const int32_t* __restrict__ wave_off,
                                                           float2* __restrict__ vis_uv,
                                                           int32_t* __restrict__ vis_idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npoints) return;
    const uint64_t m = wave_mask[i >> 6];
    const int lane = i & 63;
    if (!((m >> lane) & 1)) return;
    const int pos = wave_off[i >> 6] + __popcll(m & ((1ull << lane) - 1));
    vis_uv[pos] = uv_all[i];
    vis_idx[pos] = i;
}

int slamfe_launch_reproject(hipStream_t s, const float* xyz, int npoints, const float* M16, float th,
                            float W, float H, float* uv_all, uint64_t* wave_mask, int32_t* wave_off,
                            int32_t* total, float* vis_uv, int32_t* vis_idx)
{
    if (npoints <= 0) return 0;
    slamfe_proj_mat M;
    for (int i = 0; i < 16; ++i) M.m[i] = M16[i];
    const int nblk = (npoints + 255) / 256, nwaves = nblk * 4;
    hipLaunchKernelGGL(k_reproject_test, dim3(nblk), dim3(256), 0, s, xyz, npoints, M, th, W, H,
                       reinterpret_cast<float2*>(uv_all), wave_mask);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reproject_scan, dim3(1), dim3(1024), 0, s, wave_mask, nwaves, wave_off, total);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reproject_scatter, dim3(nblk), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(uv_all), npoints, wave_mask, wave_off,
                       reinterpret_cast<float2*>(vis_uv), vis_idx);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Triangulation of point pairs, Init::triangulate (Init.cc:5-31): one lane per
// pair builds the 4x4 system and runs the one-sided Jacobi SVD of
// slamfe_math.h (the same source the host side compiles).
// ---------------------------------------------------------------------------
struct slamfe_proj_pair { float p1[12], p2[12]; };

__global__ __launch_bounds__(64) void k_triangulate(slamfe_proj_pair P, const float2* __restrict__ pts1,
                                                    const float2* __restrict__ pts2, int n,
                                                    float* __restrict__ xyz)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float2 a = pts1[i], b = pts2[i];
    float out[3];
    slamfe_triangulate_point(P.p1, P.p2, a.x, a.y, b.x, b.y, out);
    xyz[3 * (size_t)i] = out[0];
    xyz[3 * (size_t)i + 1] = out[1];
    xyz[3 * (size_t)i + 2] = out[2];
}

int slamfe_launch_triangulate(hipStream_t s, const float* P1, const float* P2, const float* d_pts1,
                              const float* d_pts2, int n, float* d_xyz)
{
    if (n <= 0) return 0;
    slamfe_proj_pair P;
    for (int i = 0; i < 12; ++i) {
        P.p1[i] = P1[i];
        P.p2[i] = P2[i];
    }
    hipLaunchKernelGGL(k_triangulate, dim3((n + 63) / 64), dim3(64), 0, s, P,
                       reinterpret_cast<const float2*>(d_pts1), reinterpret_cast<const float2*>(d_pts2), n,
                       d_xyz);
    LAUNCH_CHECK();
    return 0;
}

// Raise the dynamic-LDS limits of the two kernels that may need more than the
// default 64 KiB (4K-wide frames; very large local maps).
int slamfe_kernels_init(size_t sort_lds_bytes)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_select_write),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sort_lds_bytes > 65536 ? sort_lds_bytes : 65536));
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_resolve),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            SLAMFE_PROJ_MAX_POINTS / 8 +
                                (2 * (PROJ_CHUNK * PROJ_STRIDE + PROJ_CHUNK) + PROJ_CHUNK + 2 * PROJ_CLAIM) * 4 + 64);
    if (e != hipSuccess) return (int)e;
    return 0;
}
