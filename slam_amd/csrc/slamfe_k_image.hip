// slamfe_k_image.hip -- frame ingest, pyramid, fused FAST + blur
#include "slamfe_device.h"

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
// two taps of each pixel are picked and widened to a 16-bit pair by one v_perm
// (selector decoded once per column block) and combined with their 8.8 weights
// by v_dot2_u32_u16; the vertical blend of
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

// A wavefront produces RS_ROWS consecutive destination rows of a block of 64
// columns (16 lanes x 4 pixels) in four frames at once (lane / 16 = frame within
// the group): level widths are nowhere near multiples of 256 columns, and with a
// whole wave on one row 30 % of the lanes idled; the row walk stays wave-uniform
// because the four frames share the geometry.  The frames of a group are 8 apart
// so that they belong to the same XCD as in the other per-frame kernels.
// Everything that
// depends on the column only (source offset, byte selection, tap weights of the
// lane's 4 pixels) is decoded once per column block; the horizontal pass of a
// source row (4 values of 8.8 fixed point, two dwords) is computed once and used
// by both destination rows that blend it (consecutive destination rows share a
// source row: 1.2 source rows per destination row instead of 2).
// (8 rows per wave: with 4 the pyramid of 1024 VGA frames took 9 % longer -- a wave's
// life is one round of loads, and per-wave set-up (taps, selectors) is amortised over
// twice the rows; with 12 or 16 the source rows held in registers cost occupancy.)
#define RS_ROWS 8
#define RS_PRE 11 /* source rows requested at once */

__global__ __launch_bounds__(256) void k_resize(slamfe_dev_geom g, int l, int nframes,
                                                uint8_t* __restrict__ pyr,
                                                const uint32_t* __restrict__ taps, slamfe_level0 l0)
{
    const slamfe_dev_level& S = g.lv[l - 1];
    const slamfe_dev_level& D = g.lv[l];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nxb = (D.w + 63) >> 6; // column blocks of the level
    const int item = wg_item();
    const int y_first = ((item / nxb) * 4 + wv) * RS_ROWS;
    if (y_first >= D.h) return;
    const int y_end = min(y_first + RS_ROWS, D.h);
    const int lane = threadIdx.x & 63;
    // frame of this lane: group member j = lane / 16 of the group blockIdx.z
    const int f = (int)((blockIdx.z * 4 + (unsigned)(lane >> 4)) * gridDim.x + blockIdx.x);
    const int x = (item % nxb) * 64 + (lane & 15) * 4;
    const uint8_t* src = l == 1 ? l0.base + (size_t)f * l0.frame_stride : pyr + (size_t)f * g.frame_bytes + S.off;
    uint8_t* dst = pyr + (size_t)f * g.frame_bytes + D.off;
    const int spitch = S.pitch, dpitch = D.pitch;
    if (f < nframes && x < D.w) {
        // the tap table is padded with zero entries to a multiple of 4 per level
        const uint4 tx = *reinterpret_cast<const uint4*>(taps + D.tap_x + x);
        const uint32_t txs[4] = {tx.x, tx.y, tx.z, tx.w};
        // aligned start of the 12 source bytes, kept inside the row: a window that would
        // run past the row's last byte starts earlier instead (the taps that matter
        // end at column w - 1, so they stay inside the shifted window; a clamped tap's
        // second byte may fall outside it and carries weight 0).  Level 1 may read
        // the caller's frames in place, where nothing follows the last row.
        const uint32_t a = min((txs[0] & 0xffffu) & ~3u, (uint32_t)S.pitch - 12u);
        // per pixel: which dword pair of the 12-byte window holds its two taps
        // (bytes o, o+1 with o = 0 .. 10: the pair (p0,p1) up to o = 6, else (p1,p2))
        // and the v_perm selector that picks them widened to 16 bits each
        bool upper[4];
        uint32_t pick[4];
        slamfe_us2 cv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t o = (txs[k] & 0xffffu) - a; // 0 .. 10
            const uint32_t c1 = txs[k] >> 16;
            cv[k] = as_us2((256u - c1) | (c1 << 16));
            upper[k] = o >= 7u;
            const uint32_t off = upper[k] ? o - 4u : o;
            pick[k] = 0x0c000c00u | ((off + 1u) << 16) | off; // 0x00 b(off+1) 0x00 b(off)
        }
        // horizontal pass of one source row: the 4 pixels' 8.8 values, two per dword.
        // 12 source bytes; bytes past the row end are read but carry weight 0
        auto hpass = [&](uint32_t p0, uint32_t p1, uint32_t p2) -> uint2 {
            uint32_t h[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t lo = upper[k] ? p1 : p0, hi = upper[k] ? p2 : p1;
                const uint32_t q = __builtin_amdgcn_perm(hi, lo, pick[k]);
                h[k] = __builtin_amdgcn_udot2(as_us2(q), cv[k], 0u, false); // 8.8, <= 65280
            }
            return make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        };
        // one destination row from the horizontal passes of its two source rows
        auto emit = [&](const uint2& H0, const uint2& H1, uint32_t w1, int y) {
            const slamfe_us2 wv2 = as_us2((256u - w1) | (w1 << 16));
            // pair the two rows' values of each pixel: (h0, h1) -> one v_dot2 with the
            // rounding constant as the addend: out = (h0*w0 + h1*w1 + 2^15) >> 16
            const uint32_t e0 = __builtin_amdgcn_perm(H1.x, H0.x, 0x05040100u); // px0: low halves
            const uint32_t e1 = __builtin_amdgcn_perm(H1.x, H0.x, 0x07060302u); // px1: high halves
            const uint32_t e2 = __builtin_amdgcn_perm(H1.y, H0.y, 0x05040100u);
            const uint32_t e3 = __builtin_amdgcn_perm(H1.y, H0.y, 0x07060302u);
            // each sum is at most 65280 * 256 + 2^15 < 2^24: its byte 2 is the pixel
            const uint32_t v0 = __builtin_amdgcn_udot2(as_us2(e0), wv2, 1u << 15, false);
            const uint32_t v1 = __builtin_amdgcn_udot2(as_us2(e1), wv2, 1u << 15, false);
            const uint32_t v2 = __builtin_amdgcn_udot2(as_us2(e2), wv2, 1u << 15, false);
            const uint32_t v3 = __builtin_amdgcn_udot2(as_us2(e3), wv2, 1u << 15, false);
            const uint32_t out = __builtin_amdgcn_perm(v1, v0, 0x0c0c0602u) |
                                 __builtin_amdgcn_perm(v3, v2, 0x06020c0cu);
            // rows are padded to a multiple of 64 bytes: the dword store stays in the row
            *reinterpret_cast<uint32_t*>(dst + (size_t)y * dpitch + x) = out;
        };
        // walk the source rows the strip needs in order
        int y = y_first;
        uint32_t ty = taps[D.tap_y + y];
        int yo = (int)(ty & 0xffffu);
        uint32_t w1 = ty >> 16;
        int yo1 = yo + (w1 ? 1 : 0);
        int sy = yo;
        const uint32_t ty_last = taps[D.tap_y + y_end - 1];
        const int sy_last = (int)(ty_last & 0xffffu) + ((ty_last >> 16) ? 1 : 0); // uniform
        uint2 Hprev = make_uint2(0, 0), Hcur = make_uint2(0, 0);
        if (sy_last - sy < RS_PRE) {
            // (the usual case: RS_ROWS rows at scale 1.2 blend at most eleven source rows)
            // all source rows are requested at once, so that the wave waits for memory
            // once and not once per row
            uint3 R[RS_PRE];
#pragma unroll
            for (int k = 0; k < RS_PRE; ++k)
                R[k] = *reinterpret_cast<const uint3*>(src + (size_t)min(min(sy + k, sy_last), S.h - 1) * spitch + a);
#pragma unroll
            for (int k = 0; k < RS_PRE; ++k) {
                if (y < y_end) { // uniform
                    Hprev = Hcur;
                    Hcur = hpass(R[k].x, R[k].y, R[k].z);
                    while (y < y_end && yo1 == sy) { // destination rows whose second source row is sy
                        emit(yo == sy ? Hcur : Hprev, Hcur, w1, y);
                        if (++y < y_end) {
                            ty = taps[D.tap_y + y];
                            yo = (int)(ty & 0xffffu);
                            w1 = ty >> 16;
                            yo1 = yo + (w1 ? 1 : 0);
                        }
                    }
                    ++sy;
                }
            }
        } else {
            // the next row's 12 bytes are requested before the current row is processed
            uint3 pw = *reinterpret_cast<const uint3*>(src + (size_t)sy * spitch + a);
            uint32_t p0 = pw.x, p1 = pw.y, p2 = pw.z;
            while (y < y_end) { // uniform
                const int nsy = min(sy + 1, S.h - 1);
                const uint3 nw = *reinterpret_cast<const uint3*>(src + (size_t)nsy * spitch + a);
                const uint32_t n0 = nw.x, n1 = nw.y, n2 = nw.z;
                Hprev = Hcur;
                Hcur = hpass(p0, p1, p2);
                while (y < y_end && yo1 == sy) { // destination rows whose second source row is sy
                    emit(yo == sy ? Hcur : Hprev, Hcur, w1, y);
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
}

// ---------------------------------------------------------------------------
// K1+K2 in one launch, for calls of a few frames where the chain of dependent
// launches (ingest + one resize per level), not the arithmetic, sets the time.
//
// A workgroup owns one SLAMFE_PT_W x SLAMFE_PT_H tile of level 0 and the
// corresponding piece of every level (host-made range tables, see
// build_fused_pyramid_tables).  It reads its level-0 region from the input frame
// (gray copy or BGR -> gray, as k_ingest), and then computes level after level
// in LDS, each from the one before -- the same taps and the same fixed-point
// arithmetic as k_resize: h = s[o] * (256 - c) + s[o + 1] * c per source row,
// out = (h0 * (256 - w) + h1 * w + 2^15) >> 16 -- storing the columns and rows it
// owns.  The region it has to compute is a little larger than what it owns (the
// taps of the next level's region reach into the neighbours' pixels); that halo
// is recomputed by both neighbours instead of being exchanged.
// ---------------------------------------------------------------------------
#ifdef SLAMFE_STAMPS
// Diagnostic build only: shader-clock and wall-clock stamps of workgroup (0,0,0)
// at the phases of k_pyramid_fused, read back with slamfe_debug_stamps().
__device__ unsigned long long g_pf_stamps[64];
#define PF_STAMP(i)                                                                       \
    do {                                                                                  \
        if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && threadIdx.x == 0) {  \
            g_pf_stamps[2 * (i)] = __builtin_amdgcn_s_memtime();                          \
            g_pf_stamps[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();                  \
        }                                                                                 \
    } while (0)
extern "C" int slamfe_debug_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pf_stamps), (size_t)n * 8);
}
#else
#define PF_STAMP(i)
#endif

#define PF_THREADS 512
#define PF_WAVES (PF_THREADS / 64)
#define PF_ROWS 8 /* rows a thread fetches per batch of independent loads */

// Four horizontally adjacent gray pixels of the input frame as one dword; `valid`
// (1..4) of them exist.  Full groups are fetched with unaligned dword loads.
__device__ __forceinline__ uint32_t pf_load4(const uint8_t* __restrict__ p, int channels, int valid)
{
    if (channels == 1) {
        if (valid == 4) return load_u32_unaligned(p);
        uint32_t v = 0;
        for (int k = 0; k < valid; ++k) v |= (uint32_t)p[k] << (8 * k);
        return v;
    }
    uint8_t q[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (valid == 4) {
        const uint32_t a = load_u32_unaligned(p), b = load_u32_unaligned(p + 4), c = load_u32_unaligned(p + 8);
        __builtin_memcpy(q, &a, 4);
        __builtin_memcpy(q + 4, &b, 4);
        __builtin_memcpy(q + 8, &c, 4);
    } else {
        for (int k = 0; k < 3 * valid; ++k) q[k] = p[k];
    }
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        v |= (uint32_t)((q[3 * k] * SLAMFE_B2Y + q[3 * k + 1] * SLAMFE_G2Y + q[3 * k + 2] * SLAMFE_R2Y +
                         (1 << (SLAMFE_GRAY_SHIFT - 1))) >> SLAMFE_GRAY_SHIFT) << (8 * k);
    return v;
}

__global__ __launch_bounds__(PF_THREADS) void k_pyramid_fused(slamfe_dev_geom g, const uint8_t* __restrict__ in,
                                                              int stride, size_t frame_stride, int channels,
                                                              uint8_t* __restrict__ pyr,
                                                              const uint32_t* __restrict__ taps)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pl[];
    uint32_t* rng = reinterpret_cast<uint32_t*>(pl);                 // [axis][level][need, own]
    uint32_t* tapc = reinterpret_cast<uint32_t*>(pl + g.pt_off_taps); // per level: x taps, then y taps
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = blockIdx.x, ty = blockIdx.y, f = blockIdx.z;
    const int L = g.nlevels;
    PF_STAMP(0);
    if (tid < 4 * L) {
        const int axis = tid / (2 * L), rem = tid - axis * 2 * L;
        const int l = rem >> 1, wsel = rem & 1;
        rng[tid] = taps[(axis ? g.ptab_y : g.ptab_x) + (size_t)(l * (axis ? g.pt_y : g.pt_x) + (axis ? ty : tx)) * 2 + wsel];
    }
    __syncthreads();
    PF_STAMP(1);
    auto need = [&](int axis, int l, int& a, int& b) {
        const uint32_t v = rng[(axis * L + l) * 2];
        a = (int)(v & 0xffffu);
        b = (int)(v >> 16);
    };
    auto own = [&](int axis, int l, int& a, int& b) {
        const uint32_t v = rng[(axis * L + l) * 2 + 1];
        a = (int)(v & 0xffffu);
        b = (int)(v >> 16);
    };
    // tap cache: the taps of every level's needed columns and rows, as one index
    // space so that a thread's loads are independent of each other (and of the
    // level-0 region loads below)
    {
        int tb[SLAMFE_MAX_LEVELS + 1]; // first cache entry of each level
        tb[0] = tb[1] = 0;
        for (int l = 1; l < L; ++l) {
            int x0, x1, y0, y1;
            need(0, l, x0, x1);
            need(1, l, y0, y1);
            tb[l + 1] = tb[l] + max(x1 - x0, 0) + max(y1 - y0, 0);
        }
        for (int e = tid; e < tb[L]; e += PF_THREADS) {
            int l = 1;
            while (l + 1 < L && e >= tb[l + 1]) ++l;
            int x0, x1, y0, y1;
            need(0, l, x0, x1);
            need(1, l, y0, y1);
            const int i = e - tb[l], nx = max(x1 - x0, 0);
            tapc[e] = i < nx ? taps[g.lv[l].tap_x + x0 + i] : taps[g.lv[l].tap_y + y0 + (i - nx)];
        }
    }
    uint8_t* bufA = pl + g.pt_off_a;
    uint8_t* bufB = pl + g.pt_off_b;
    // the owned part of a level's LDS region goes to the pyramid as dwords (owned
    // columns start at a multiple of 4; rows are padded to 64 bytes)
    // (items = groups of four columns, dealt to the threads as one index space:
    // the pieces are far narrower than 64 lanes x 4 columns)
    auto store_owned = [&](int l, const uint8_t* buf, int nx0, int ny0, int pitch) {
        int ox0, ox1, oy0, oy1;
        own(0, l, ox0, ox1);
        own(1, l, oy0, oy1);
        uint8_t* dst = pyr + (size_t)f * g.frame_bytes + g.lv[l].off;
        const int dp = g.lv[l].pitch;
        const int ng = (max(ox1 - ox0, 0) + 3) >> 2, items = ng * max(oy1 - oy0, 0);
        const float inv = 1.0f / (float)max(ng, 1);
        for (int i = tid; i < items; i += PF_THREADS) {
            const int r = (int)(((float)i + 0.5f) * inv), y = oy0 + r, x = ox0 + 4 * (i - r * ng);
            const uint8_t* sp = buf + (y - ny0) * pitch + (x - nx0);
            const uint32_t v = (uint32_t)sp[0] | ((uint32_t)sp[1] << 8) | ((uint32_t)sp[2] << 16) | ((uint32_t)sp[3] << 24);
            *reinterpret_cast<uint32_t*>(dst + (size_t)y * dp + x) = v;
        }
    };
    // ---- level 0: the needed region, four pixels per load, PF_ROWS rows per thread
    // in flight together
    int sx0, sx1, sy0, sy1;
    need(0, 0, sx0, sx1);
    need(1, 0, sy0, sy1);
    int spitch = ((sx1 - sx0 + 3) & ~3) + 4;
    {
        const uint8_t* src = in ? in + (size_t)f * frame_stride : pyr + (size_t)f * g.frame_bytes + g.lv[0].off;
        const int sstride = in ? stride : g.lv[0].pitch, ch = in ? channels : 1;
        const int nxw = sx1 - sx0, nyh = sy1 - sy0, nd = (nxw + 3) >> 2;
        for (int yb = 0; yb < nyh; yb += PF_WAVES * PF_ROWS)
            for (int xb = 0; xb < nd; xb += 64) {
                const int xd = xb + lane;
                uint32_t v[PF_ROWS];
#pragma unroll
                for (int u = 0; u < PF_ROWS; ++u) {
                    const int y = yb + wv + PF_WAVES * u;
                    v[u] = 0;
                    if (y < nyh && xd < nd)
                        v[u] = pf_load4(src + (size_t)(sy0 + y) * sstride + (size_t)(sx0 + 4 * xd) * ch, ch,
                                        min(4, nxw - 4 * xd));
                }
#pragma unroll
                for (int u = 0; u < PF_ROWS; ++u) {
                    const int y = yb + wv + PF_WAVES * u;
                    if (y < nyh && xd < nd) *reinterpret_cast<uint32_t*>(bufA + y * spitch + 4 * xd) = v[u];
                }
            }
    }
    __syncthreads();
    PF_STAMP(2);
    if (in) store_owned(0, bufA, sx0, sy0, spitch);
    PF_STAMP(3);
    // ---- levels 1 .. L-1, ping-pong between the two buffers
    const uint8_t* S = bufA;
    uint8_t* D = bufB;
    int tbase = 0;
    for (int l = 1; l < L; ++l) {
        int dx0, dx1, dy0, dy1;
        need(0, l, dx0, dx1);
        need(1, l, dy0, dy1);
        const int nx = max(dx1 - dx0, 0), ny = max(dy1 - dy0, 0);
        const int dpitch = ((nx + 3) & ~3) + 4;
        const uint32_t* txs = tapc + tbase;
        const uint32_t* tys = txs + nx;
        const int ng = (nx + 3) >> 2, items = ng * ny;
        const float inv = 1.0f / (float)max(ng, 1);
        for (int i = tid; i < items; i += PF_THREADS) {
            const int y = (int)(((float)i + 0.5f) * inv), x = 4 * (i - y * ng);
            const uint32_t tyv = tys[y];
            const int yo = (int)(tyv & 0xffffu) - sy0;
            const uint32_t w1 = tyv >> 16, w0 = 256u - w1;
            const uint8_t* r0 = S + yo * spitch;
            const uint8_t* r1 = r0 + (w1 ? spitch : 0);
            uint32_t outv = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (x + k < nx) {
                    const uint32_t txv = txs[x + k];
                    const int o = (int)(txv & 0xffffu) - sx0;
                    const uint32_t c1 = txv >> 16, c0 = 256u - c1;
                    const int o1 = o + (c1 ? 1 : 0);
                    const uint32_t h0 = r0[o] * c0 + r0[o1] * c1;
                    const uint32_t h1 = r1[o] * c0 + r1[o1] * c1;
                    outv |= ((h0 * w0 + h1 * w1 + (1u << 15)) >> 16) << (8 * k);
                }
            }
            *reinterpret_cast<uint32_t*>(D + y * dpitch + x) = outv;
        }
        __syncthreads();
        PF_STAMP(2 + 2 * l);
        store_owned(l, D, dx0, dy0, dpitch);
        PF_STAMP(3 + 2 * l);
        // the next level reads D and overwrites S, which nobody reads any more
        const uint8_t* t = S;
        S = D;
        D = const_cast<uint8_t*>(t);
        sx0 = dx0;
        sy0 = dy0;
        spitch = dpitch;
        tbase += nx + ny;
    }
}

// ---------------------------------------------------------------------------
// K3+K6  FAST-9/16 (segment test, score, strict 3x3 NMS, border filter) fused
//        with the 7x7 sigma=2 blur of the same level.
//        (detector->detect, Init.cc:76, per level inside cv::ORB; and the
//        GaussianBlur(7x7, 2) cv::ORB applies before describing, Init.cc:78.)
//
// One workgroup = one group of SLAMFE_FB_TILES vertically adjacent 64x16 tiles
// of one level of one frame (the last group of a level may hold fewer rows; all
// passes are trimmed to the rows that exist).  The group plus a 4-px halo is
// read from HBM once (aligned dwords, REFLECT_101 at the image border) into LDS
// and feeds both computations:
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
//  blur   on the matrix cores (blur_mfma below): the separable 7-tap filter is
//         a product with a banded matrix of taps on either side, exact in i32
// ---------------------------------------------------------------------------
#define FL_W (TW + 32)         /* LDS tile row: image columns x0-16 .. x0+79, six 16-byte pieces */
#define FL_WD (FL_W / 4)
#define FL_X0 12               /* byte column of image column x0-4 (the first one that is used) */
#define GH (TH * SLAMFE_FB_TILES) /* rows of a tile group */
#define FL_H (GH + 8)          /* rows y0-4 .. y0+GH+3 */
#define FB_COLS (TW + 8)       /* used columns: x0-4 .. x0+67 = 18 dwords */
#define FB_COLD (FB_COLS / 4)
#define FS_W (TW + 2)          /* score tile: 66 x (GH + 2) */
#define FS_H (GH + 2)
#define FS_N (FS_W * FS_H)

// ---------------------------------------------------------------------------
// The 7x7 blur as two chained matrix products (v_mfma_i32_32x32x32_i8).
//
// With P the LDS tile, T the band matrix of the taps k = [18,34,49,55,49,34,18]:
//   row pass     S[r][x] = sum_c P[r][c] * k[c - x + 3]           (S = P * T)
//   column pass  F[x][y] = sum_r S[r][x] * k[r - y + 3]           (F = S^T * T)
//   out[y][x]    = min((F + 2^15) >> 16, 255)
// All sums are exact integers, so any exact decomposition gives the filter's
// bytes.  One wave produces 32 columns x 32 rows of output:
//
//  * row pass: A = pixels, lane = tile row r (two blocks of 32 rows: the 38 rows
//    an output block needs), 16 consecutive bytes of the row per lane and k-step
//    (one ds_read_b128; two k-steps cover the 38 columns); B = taps, constant per
//    lane.  The instruction multiplies signed bytes: pixels are re-centred to
//    p - 128 (p ^ 0x80), which shifts every row sum by 128 * 257.
//  * the result tile has its column x on the lane and its rows in the 16
//    registers, which is the operand layout of a product that sums over rows:
//    the column pass takes it as its A operand with no lane movement.  A row sum
//    needs 16 bits, an operand byte holds 8: D + 128 (range -32768 .. 32767) is
//    split into its signed high byte h and its low byte, taken as l - 128, so
//    D = 256 h + (l - 128) and F = 256 * sum k h + sum k (l - 128) + constants:
//    two accumulator chains, joined by one shift-and-add per output.
//  * constants ride in operand bytes the filter does not use: tile columns 0 and
//    95 hold 0x82 (2 after re-centring) against a tap-matrix entry of 64 -- the
//    "+ 128" of the split; in the second k-step of the column pass (tile rows
//    33 .. 64, of which only 33 .. 38 exist) four operand bytes are constants
//    against entries of 64, which adds 128 * 257 * 257 + 2^15 (the re-centring
//    undone, the split's offsets, the rounding constant) to F.
//  * the column pass leaves the output row y on the lane and four consecutive
//    columns in registers 4q .. 4q+3: one dword store per q into the tiled
//    blurred level.
// ---------------------------------------------------------------------------
typedef int blur_v4i __attribute__((ext_vector_type(4)));
typedef int blur_v16i __attribute__((ext_vector_type(16)));

// tile row of element j of lane half h in a result tile used as an operand
constexpr int blur_rho(int h, int j) { return (j & 3) + 8 * (j >> 2) + 4 * h; }
constexpr int blur_tap(int t)
{
    constexpr int K[7] = SLAMFE_GAUSS7_INIT;
    return t >= 0 && t <= 6 ? K[t] : 0;
}
// Tap operands, per lane (n = lane & 31, h = lane >> 5), 16 bytes each:
//  b1[xb][s]: row pass, k-step s; byte j belongs to the tile column 16 + 32 xb
//             - 16 + 32 s + 16 h + j and weighs into output column 16 + 32 xb + n
//  b2[mb]:    column pass, k-step mb; byte j belongs to tile row 1 + 32 mb +
//             rho(h, j) and weighs into output row 4 + n
struct slamfe_blur_tab {
    uint32_t b1[2][2][64][4];
    uint32_t b2[2][64][4];
};
constexpr slamfe_blur_tab make_blur_tab()
{
    slamfe_blur_tab T{};
    for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 31, h = lane >> 5;
        for (int j = 0; j < 16; ++j) {
            for (int xb = 0; xb < 2; ++xb)
                for (int s = 0; s < 2; ++s) {
                    int v = blur_tap(-16 + 32 * s + 16 * h + j - n + 3);
                    // the constant column (tile column 0 for xb = 0, 95 for xb = 1)
                    if ((xb == 0 && s == 0 && h == 0 && j == 0) || (xb == 1 && s == 1 && h == 1 && j == 15)) v += 64;
                    T.b1[xb][s][lane][j >> 2] |= (uint32_t)v << (8 * (j & 3));
                }
            for (int mb = 0; mb < 2; ++mb) {
                int v = blur_tap(32 * mb + blur_rho(h, j) - n);
                if (mb == 1 && j >= 4 && j < 8) v += 64; // against the constant operand bytes
                T.b2[mb][lane][j >> 2] |= (uint32_t)v << (8 * (j & 3));
            }
        }
    }
    return T;
}
__device__ const slamfe_blur_tab g_blur_tab = make_blur_tab();
// constant operand bytes 4 .. 7 of the column pass's second k-step, both lane
// halves: 2 * 64 * (256 * (127 + 127 + 5) + 1) = 128 * 257 * 257 + 2^15
#define BLUR_CONST_H 0x00057f7fu
#define BLUR_CONST_L 0x00000001u
static_assert(2 * 64 * (256 * (127 + 127 + 5) + 1) == 128 * 257 * 257 + (1 << 15), "blur constants");

__device__ __forceinline__ void blur_mfma(const uint32_t* tile32, int xb, int lane, int nrows, int x0,
                                          int y0, int w, int pitch, uint8_t* __restrict__ out_level)
{
    const int n = lane & 31, h = lane >> 5;
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile32);
    auto tab = [&](const uint32_t* p) { return *reinterpret_cast<const blur_v4i*>(p); };
    const blur_v4i b1s0 = tab(g_blur_tab.b1[xb][0][lane]), b1s1 = tab(g_blur_tab.b1[xb][1][lane]);
    const blur_v4i b2k0 = tab(g_blur_tab.b2[0][lane]), b2k1 = tab(g_blur_tab.b2[1][lane]);
    // ---- row pass: tile rows 1 .. 32 and 33 .. 38 (lanes past row 38 read the last row)
    const int colb = 32 * xb + 16 * h;
    const uint8_t* r0 = tb + (1 + n) * FL_W + colb;
    const uint8_t* r1 = tb + min(33 + n, FL_H - 1) * FL_W + colb;
    auto pix = [&](const uint8_t* p) {
        blur_v4i a = *reinterpret_cast<const blur_v4i*>(p);
        a ^= (int)0x80808080u;
        return a;
    };
    blur_v16i d0 = {0}, d1 = {0};
    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(pix(r0), b1s0, d0, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(pix(r0 + 32), b1s1, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(pix(r1), b1s0, d1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(pix(r1 + 32), b1s1, d1, 0, 0, 0);
    // ---- split: registers 4q .. 4q+3 -> one dword of high bytes, one of (low byte - 128)
    auto split = [&](const blur_v16i& d, int q, int& hi, int& lo) {
        const uint32_t t0 = __builtin_amdgcn_perm((uint32_t)d[4 * q + 1], (uint32_t)d[4 * q], 0x05010400u);
        const uint32_t t1 = __builtin_amdgcn_perm((uint32_t)d[4 * q + 3], (uint32_t)d[4 * q + 2], 0x05010400u);
        lo = (int)(__builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u);
        hi = (int)__builtin_amdgcn_perm(t1, t0, 0x07060302u);
    };
    blur_v4i aH0, aL0, aH1, aL1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int hi, lo;
        split(d0, q, hi, lo);
        aH0[q] = hi;
        aL0[q] = lo;
    }
    {
        int hi, lo;
        split(d1, 0, hi, lo); // tile rows 33 .. 40: all that exist of the second block
        aH1 = blur_v4i{hi, (int)BLUR_CONST_H, 0, 0};
        aL1 = blur_v4i{lo, (int)BLUR_CONST_L, 0, 0};
    }
    // ---- column pass
    blur_v16i fh = {0}, fl = {0};
    fh = __builtin_amdgcn_mfma_i32_32x32x32_i8(aH0, b2k0, fh, 0, 0, 0);
    fh = __builtin_amdgcn_mfma_i32_32x32x32_i8(aH1, b2k1, fh, 0, 0, 0);
    fl = __builtin_amdgcn_mfma_i32_32x32x32_i8(aL0, b2k0, fl, 0, 0, 0);
    fl = __builtin_amdgcn_mfma_i32_32x32x32_i8(aL1, b2k1, fl, 0, 0, 0);
    // ---- join, round (the constant is in), saturate, store: lane = output row
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union pk { uint32_t u; us2 v; };
    const int gy = y0 + n;
    // The blurred level is stored in tiles of 16 x 8 pixels (one 128-byte line
    // each, tiles row-major): the descriptor kernel fetches a keypoint's
    // neighbourhood as whole lines.  Rows are padded to a multiple of 64 bytes,
    // so a store that covers pixels beyond w stays inside the level.
    // (a level is far below 4 GiB: 32-bit offsets from the level's uniform base)
    const uint32_t orow = (uint32_t)(gy >> 3) * (uint32_t)(pitch >> 4) * 128u + (uint32_t)(gy & 7) * 16u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = ((uint32_t)fh[4 * q + i] << 8) + (uint32_t)fl[4 * q + i];
        // t >> 16 is at most 257: two of them per dword as 16-bit values, clamped together
        pk p01, p23, lim;
        lim.u = 0x00ff00ffu;
        p01.u = __builtin_amdgcn_perm(t[1], t[0], 0x07060302u);
        p23.u = __builtin_amdgcn_perm(t[3], t[2], 0x07060302u);
        p01.v = __builtin_elementwise_min(p01.v, lim.v);
        p23.v = __builtin_elementwise_min(p23.v, lim.v);
        const uint32_t px = __builtin_amdgcn_perm(p23.u, p01.u, 0x06040200u);
        const int gx = x0 + 32 * xb + 8 * q + 4 * h;
        if (n < nrows && gx < w)
            *reinterpret_cast<uint32_t*>(out_level + (orow + (uint32_t)(gx >> 4) * 128u + (uint32_t)(gx & 15))) = px;
    }
}

// atomicAdd on an LDS word, written as the instruction itself: the compiler
// rewrites atomicAdd into a scan over the lanes (per-lane operands) or a
// lane-count multiplication with an election of a leader lane (uniform operand),
// both costly in a loop that is bound by instruction issue.  The hardware
// serialises the lanes that hit one address inside the LDS unit instead.
__device__ __forceinline__ uint32_t lds_fetch_add(uint32_t* p, uint32_t v)
{
    uint32_t old;
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
    return old;
}

// Appends the survivors of one pretest task to a list of 16-bit entries in LDS.
// f0 / f1: bit 7 of byte j set iff pixel j / pixel 4 + j of the task survives (the
// other bits clear); slot: where the lane's first entry goes; e2: the entry of
// pixel 0 in both halves.  Per pixel: one byte-select compare into a scalar register
// pair, one 16-bit store of either register half, one add in place -- the compiler's
// form of the same loop (test, compare, add into a copy, move back, add for the
// entry) costs twice the vector instructions, in the pass that has the most.
// The lane mask is restored before the block ends, and the stores have landed.
__device__ __forceinline__ void lds_append_survivors(uint32_t f0, uint32_t f1, uint16_t* slot, uint32_t e2)
{
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)slot;
    // (halves added separately: the first task of a row starts three entries before the row)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union pk { uint32_t u; us2 v; };
    auto plus = [&](unsigned short k) { pk a, r; a.u = e2; r.v = a.v + us2{k, (unsigned short)(k + 1)}; return r.u; };
    const uint32_t e01 = plus(0), e23 = plus(2), e45 = plus(4), e67 = plus(6);
    const uint32_t zero = 0;
    uint64_t saved, k0, k1, k2, k3, k4, k5, k6, k7;
    // (all eight compares run under the full lane mask, before any store narrows it)
#define CMP(k, f, byte) "v_cmp_ne_u32_sdwa %[" k "], %[" f "], %[z] src0_sel:BYTE_" byte " src1_sel:DWORD\n\t"
#define APPEND(k, st, e) "s_mov_b64 exec, %[" k "]\n\t" st " %[a], %[" e "]\n\tv_add_u32_e32 %[a], 2, %[a]\n\t"
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 CMP("k0", "f0", "0") CMP("k1", "f0", "1") CMP("k2", "f0", "2") CMP("k3", "f0", "3")
                 CMP("k4", "f1", "0") CMP("k5", "f1", "1") CMP("k6", "f1", "2") CMP("k7", "f1", "3")
                 APPEND("k0", "ds_write_b16", "e01") APPEND("k1", "ds_write_b16_d16_hi", "e01")
                 APPEND("k2", "ds_write_b16", "e23") APPEND("k3", "ds_write_b16_d16_hi", "e23")
                 APPEND("k4", "ds_write_b16", "e45") APPEND("k5", "ds_write_b16_d16_hi", "e45")
                 APPEND("k6", "ds_write_b16", "e67") APPEND("k7", "ds_write_b16_d16_hi", "e67")
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [sv] "=&s"(saved), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3), [k4] "=&s"(k4),
                   [k5] "=&s"(k5), [k6] "=&s"(k6), [k7] "=&s"(k7), [a] "+v"(addr)
                 : [f0] "v"(f0), [f1] "v"(f1), [z] "v"(zero), [e01] "v"(e01), [e23] "v"(e23), [e45] "v"(e45),
                   [e67] "v"(e67)
                 : "memory");
#undef APPEND
#undef CMP
}

// FAST-9 score of two pixels of the LDS tile at once, one in each 16-bit half of
// a register.  With r[0..15] the ring and d = v - r, the score is the largest
// margin m - 1 such that some arc of 9 contiguous ring pixels has all d > m - 1
// or all d < -(m - 1): m = max(v - min_s max(arc_s), max_s min(arc_s) - v) over
// the sixteen arcs, each arc three runs of three ring pixels.
// gfx950 has packed three-input minimum / maximum for 16-bit floats
// (v_pk_minimum3_f16, v_pk_maximum3_f16), and a pixel value 0 .. 255 taken as the bit
// pattern of a 16-bit float is a subnormal number (value * 2^-24; the kernel's float
// mode keeps 16-bit subnormals): minimum and maximum of such numbers are minimum and
// maximum of the integers, exactly -- no arithmetic is done on them.  The network
// (three-input minima and maxima of runs of three, then of three runs) therefore
// costs the same number of instructions for two pixels as it would for one.
typedef _Float16 fs_h2 __attribute__((ext_vector_type(2)));
typedef short fs_s2 __attribute__((ext_vector_type(2)));
union fs_pk {
    uint32_t u;
    fs_h2 h;
    fs_s2 s;
};

__device__ __forceinline__ void fast_score2_lds(const uint8_t* ca, const uint8_t* cb, int thr, int& sa, int& sb)
{
    constexpr int R[16][2] = SLAMFE_FAST_RING_INIT;
    fs_pk r[16], v;
    v.u = (uint32_t)ca[0] | ((uint32_t)cb[0] << 16);
#pragma unroll
    for (int k = 0; k < 16; ++k)
        r[k].u = (uint32_t)ca[R[k][0] + R[k][1] * FL_W] | ((uint32_t)cb[R[k][0] + R[k][1] * FL_W] << 16);
    fs_pk mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn3[k].h = __builtin_elementwise_minimum(__builtin_elementwise_minimum(r[k].h, r[(k + 1) & 15].h), r[(k + 2) & 15].h);
        mx3[k].h = __builtin_elementwise_maximum(__builtin_elementwise_maximum(r[k].h, r[(k + 1) & 15].h), r[(k + 2) & 15].h);
    }
    fs_pk mn9[16], mx9[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn9[k].h = __builtin_elementwise_minimum(__builtin_elementwise_minimum(mn3[k].h, mn3[(k + 3) & 15].h), mn3[(k + 6) & 15].h);
        mx9[k].h = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx3[k].h, mx3[(k + 3) & 15].h), mx3[(k + 6) & 15].h);
    }
    // lo: some arc lies entirely at or below it; hi: some arc lies entirely at or above it
    fs_pk lo = mx9[0], hi = mn9[0];
#pragma unroll
    for (int k = 1; k < 15; k += 2) {
        lo.h = __builtin_elementwise_minimum(__builtin_elementwise_minimum(lo.h, mx9[k].h), mx9[k + 1].h);
        hi.h = __builtin_elementwise_maximum(__builtin_elementwise_maximum(hi.h, mn9[k].h), mn9[k + 1].h);
    }
    lo.h = __builtin_elementwise_minimum(lo.h, mx9[15].h);
    hi.h = __builtin_elementwise_maximum(hi.h, mn9[15].h);
    // back to integers: margin m = max(v - lo, hi - v) per half
    fs_pk d1, d2, m;
    d1.s = v.s - lo.s;
    d2.s = hi.s - v.s;
    m.s = __builtin_elementwise_max(d1.s, d2.s);
    const int ma = (int)m.s[0], mb = (int)m.s[1]; // signed: -255 .. 255
    // corner <=> some arc has all |d| > thr on one side; score = that margin - 1
    sa = ma > thr ? ma - 1 : 0;
    sb = mb > thr ? mb - 1 : 0;
}

__global__ __launch_bounds__(256) void k_fast_blur(slamfe_dev_geom g, const uint8_t* __restrict__ pyr,
                                                   uint8_t* __restrict__ blur, int nframes, int thr,
                                                   int edge, uint32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ cand_cnt, slamfe_level0 l0)
{
    __shared__ uint32_t tile32[FL_H * FL_WD];
    __shared__ __attribute__((aligned(16))) uint32_t sc32[(FS_N + 15) / 16 * 4]; // cleared 16 bytes at a time
    __shared__ uint16_t pos[FS_N];
    __shared__ uint32_t list[SLAMFE_FB_TILES * SLAMFE_TILE_CAND];
    __shared__ uint32_t nlist[SLAMFE_FB_TILES], npos;
    const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);
    uint8_t* sc = reinterpret_cast<uint8_t*>(sc32);

    const int t = wg_item(), f = wg_frame(), tid = threadIdx.x;
    if (f >= nframes) return;
    const int l = find_level_by_group(g, t);
    const slamfe_dev_level L = g.lv[l];
    const int tl = t - L.group_base;
    const int tx = tl % L.tiles_x, ty0 = (tl / L.tiles_x) * SLAMFE_FB_TILES;
    const int x0 = tx * TW, y0 = ty0 * TH;
    const int nrows = min(GH, L.h - y0); // image rows of this group
    const uint8_t* img = l == 0 ? l0.base + (size_t)f * l0.frame_stride : pyr + (size_t)f * g.frame_bytes + L.off;

    if (tid < SLAMFE_FB_TILES) nlist[tid] = 0;
    if (tid == 0) npos = 0;
    // ---- tile load: nrows + 8 rows x six aligned 16-byte pieces (columns x0-16 .. x0+79;
    // columns x0-4 .. x0+67 are used).  Rows reflect at the top / bottom border by
    // index; pieces outside the row are zero, bytes past w inside the row's
    // padding are don't-care until the fix-up below.
    for (int i = tid; i < (nrows + 8) * 6; i += 256) {
        int ly = i / 6, pc = i - ly * 6;
        int gy = reflect101(y0 - 4 + ly, L.h);
        int gx = x0 - 16 + 16 * pc;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gx >= 0 && gx < L.pitch)
            v = *reinterpret_cast<const uint4*>(img + (size_t)gy * L.pitch + gx);
        // tile columns 0 and 95 (image columns x0-16 and x0+79, which nothing reads
        // as pixels) carry the constant 0x82 of the matrix-core blur
        if (pc == 0) v.x = (v.x & 0xffffff00u) | 0x82u;
        if (pc == 5) v.w = (v.w & 0x00ffffffu) | 0x82000000u;
        *reinterpret_cast<uint4*>(&tile32[ly * FL_WD + 4 * pc]) = v;
    }
    for (int i = tid; i < ((nrows + 2) * FS_W + 15) / 16; i += 256)
        reinterpret_cast<uint4*>(sc32)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // REFLECT_101 at the left / right image border (the blur reads 3 px across
    // it): copy the mirrored bytes inside LDS.  Only border tiles take this.
    {
        const bool left = x0 == 0, right = x0 + TW + 3 >= L.w;
        if (left || right) { // uniform per workgroup
            uint8_t* tb = reinterpret_cast<uint8_t*>(tile32);
            for (int i = tid; i < (nrows + 8) * 8; i += 256) {
                int ly = i >> 3, k = i & 7; // k 0..2: left side, 4..6: right side
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
    // (With 4 pixels per task the pass took twice as many wave passes.  A fixed
    // lane -> (row, task) mapping whose addresses advance by constants executed
    // 0.8 % more instructions, not fewer: one lane of 64 idles and the division it
    // saves is a handful of operations.)
    {
        typedef short v2s __attribute__((ext_vector_type(2)));
        union pk { uint32_t u; v2s v; };
        const int blo = max(3, edge - 1);
        // tiles that lie entirely in the discarded border band skip FAST
        const bool any_fast = x0 + TW + 1 > blo && x0 - 1 < L.w - blo && y0 + nrows + 1 > blo &&
                              y0 - 1 < L.h - blo;
        const bool interior = x0 - 1 >= blo && x0 + TW <= L.w - blo - 1 && y0 - 1 >= blo &&
                              y0 + nrows <= L.h - blo - 1;
        const v2s t2 = {(short)thr, (short)thr};
        // A dword's four pixels widened to 16-bit pairs: e = pixels 0 and 2, o = pixels 1 and 3
        struct wide { pk e, o; };
        auto widen = [&](uint32_t x) -> wide {
            wide r;
            r.e.u = x & 0x00ff00ffu;
            r.o.u = __builtin_amdgcn_perm(0u, x, 0x0c030c01u);
            return r;
        };
        // sign bits of the result: set iff the pixel survives (v the pixels, p0 / p8
        // below / above, p4 / p12 three to the right / left)
        auto pretest2 = [&](pk v, pk p0, pk p8, pk p4, pk p12) -> uint32_t {
            v2s hi = __builtin_elementwise_min(__builtin_elementwise_max(p0.v, p8.v),
                                               __builtin_elementwise_max(p4.v, p12.v));
            v2s lo = __builtin_elementwise_max(__builtin_elementwise_min(p0.v, p8.v),
                                               __builtin_elementwise_min(p4.v, p12.v));
            pk r;
            r.v = t2 - __builtin_elementwise_max(hi - v.v, v.v - lo); // < 0 <=> hi > v + t or lo < v - t
            return r.u;
        };
        // survivors among the 4 pixels of a dword (C widened, U above, D below, P / N
        // the dwords before / after C in the row, widened): bit 7 of byte j of the
        // result is set iff pixel j survives (the other bits are undefined).  The
        // pixels three to the left are P's pixels 1, 3 for C's pixels 0, 2 and
        // (P's pixel 2, C's pixel 0) for pixels 1, 3; three to the right likewise.
        auto pretest4 = [&](const wide& C, uint32_t U, uint32_t D, const wide& P, const wide& N) -> uint32_t {
            const wide u = widen(U), d = widen(D);
            pk l3o, r3e;
            l3o.u = __builtin_amdgcn_alignbyte(C.e.u, P.e.u, 2);
            r3e.u = __builtin_amdgcn_alignbyte(N.o.u, C.o.u, 2);
            const uint32_t n0 = pretest2(C.e, d.e, u.e, r3e, P.o);
            const uint32_t n1 = pretest2(C.o, d.o, u.o, N.e, l3o);
            // the high bytes of the four 16-bit results, in pixel order
            return __builtin_amdgcn_perm(n1, n0, 0x07030501u);
        };
        // (the tasks past the first 256 go to the last wave, which takes no part
        // in the blur below)
        for (int i = tid, it = 0; any_fast && it < 2; ++it, i = 256 + ((tid + 64) & 255)) {
            if (i >= (nrows + 2) * (FB_COLD / 2)) continue;
            int sy = i / (FB_COLD / 2), m = i - sy * (FB_COLD / 2);
            int ly = sy + 3;
            const uint32_t* row = &tile32[ly * FL_WD + FL_X0 / 4 + 2 * m];
            uint32_t C0 = row[0], C1 = row[1], Cp = row[-1], Cn = row[2];
            uint32_t U0 = row[-3 * FL_WD], U1 = row[-3 * FL_WD + 1];
            uint32_t D0 = row[3 * FL_WD], D1 = row[3 * FL_WD + 1];
            // flags of pixels 0 .. 3 (f0) and 4 .. 7 (f1) of the task: bit 7 of byte j
            const wide wp = widen(Cp), w0 = widen(C0), w1 = widen(C1), wn = widen(Cn);
            uint32_t f0 = pretest4(w0, U0, D0, wp, w1);
            uint32_t f1 = pretest4(w1, U1, D1, w0, wn);
            // pixel k is a score position iff lx = 8m+k in [3, 68] and the image
            // position is inside the band that can still yield a keypoint:
            // keypoints outside [edge, w-edge) x [edge, h-edge) are discarded by
            // the ORB border filter, and NMS looks one pixel further, so scores
            // are needed on [lo, w-lo) x [lo, h-lo) with lo = max(3, edge-1)
            // (tiles whose score area lies inside that band need the column rule only)
            uint32_t v0, v1;
            if (interior) { // uniform: pixels 3..7 of the first task, 0..4 of the last
                v0 = m == 0 ? 0x80000000u : 0x80808080u;
                v1 = m == FB_COLD / 2 - 1 ? 0x00000080u : 0x80808080u;
            } else {
                int gy = y0 - 1 + sy, gxb = x0 - 4 + 8 * m;
                int kmin = max(3 - 8 * m, blo - gxb), kmax = min(FS_W + 2 - 8 * m, L.w - blo - 1 - gxb);
                kmin = max(kmin, 0);
                kmax = min(kmax, 7);
                const uint32_t valid = (gy >= blo && gy < L.h - blo && kmax >= kmin)
                                           ? ((2u << kmax) - 1u) & ~((1u << kmin) - 1u) : 0u;
                // bit j of a nibble -> bit 7 + 8j (the four shifted copies of the
                // nibble inside the product do not overlap)
                v0 = ((valid & 15u) * 0x10204080u) & 0x80808080u;
                v1 = ((valid >> 4) * 0x10204080u) & 0x80808080u;
            }
            f0 &= v0;
            f1 &= v1;
            // compaction: every lane with survivors reserves its entries with one
            // LDS atomic, issued as the instruction itself (see lds_fetch_add; the
            // compiler's rewrite of atomicAdd with per-lane operands is a serial
            // loop over the lanes that made the kernel 12 % slower).  Two
            // wave-level schemes measured no better: an exclusive prefix of the
            // lane counts from four count-bit ballots with one atomic per wave
            // (same time, twice the code), and one ballot per pixel with
            // rank-addressed stores (fewer vector instructions, eight exec-mask
            // sections, 6 % slower).
            const uint32_t c = (uint32_t)(__popc(f0) + __popc(f1));
            if (c) {
                const uint32_t base = lds_fetch_add(&npos, c);
                lds_append_survivors(f0, f1, pos + base, ((uint32_t)(sy * FS_W + 8 * m - 3) & 0xffffu) * 0x00010001u);
            }
        }
    }

    // ---- blur: waves 0 and 1 each produce 32 columns x nrows rows on the matrix
    // cores (the pretest's remainder above went to wave 3); the blur reads only the
    // tile, so nothing waits for it
    if (tid < 128) // uniform per wave
        blur_mfma(tile32, tid >> 6, tid & 63, nrows, x0, y0, L.w, L.pitch,
                  blur + (size_t)f * g.frame_bytes + L.off);
    __syncthreads();

    // ---- FAST pass C: full score of the pretest survivors, two per lane (survivor k
    // with survivor k + half; with an odd count the last lane scores one of them twice)
    const uint32_t np = npos;
    const uint32_t half = (np + 1) >> 1;
    for (uint32_t k = tid; k < half; k += 256) {
        const int ia = pos[k], ib = pos[min(k + half, np - 1)];
        const int ya = ia / FS_W, xa = ia - ya * FS_W, yb = ib / FS_W, xb = ib - yb * FS_W;
        int sa, sb;
        fast_score2_lds(&tile[(ya + 3) * FL_W + FL_X0 + xa + 3], &tile[(yb + 3) * FL_W + FL_X0 + xb + 3], thr, sa, sb);
        sc[ia] = (uint8_t)sa;
        sc[ib] = (uint8_t)sb;
    }

    __syncthreads();

    // ---- FAST pass D: NMS of the survivors, border filter, emission
    // (A list of the survivors that have a score, appended to by the pass above so
    // that this pass runs on full waves, cost as much as it saved.)
    for (uint32_t k = tid; k < np; k += 256) {
        int i = pos[k];
        int s = sc[i];
        int sy = i / FS_W, sx = i - sy * FS_W;
        if (s > 0 && sy >= 1 && sy <= nrows && sx >= 1 && sx <= TW) {
            const uint8_t* p = &sc[i];
            int m = max(max(max(p[-1], p[1]), max(p[-FS_W - 1], p[-FS_W])),
                        max(max(p[-FS_W + 1], p[FS_W - 1]), max(p[FS_W], p[FS_W + 1])));
            int gx = x0 - 1 + sx, gy = y0 - 1 + sy;
            if (s > m && gx >= edge && gx < L.w - edge && gy >= edge && gy < L.h - edge) {
                const int part = (sy - 1) / TH; // the tile of the group
                uint32_t q = atomicAdd(&nlist[part], 1u);
                list[part * SLAMFE_TILE_CAND + q] = SLAMFE_KEY(gx, gy, s);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int part = 0; part < SLAMFE_FB_TILES; ++part) {
        if (ty0 + part < L.tiles_y) { // uniform
            const uint32_t n = nlist[part];
            size_t tg = (size_t)f * g.total_tiles + L.tile_base + (size_t)(ty0 + part) * L.tiles_x + tx;
            if (tid < n) cand[tg * SLAMFE_TILE_CAND + tid] = list[part * SLAMFE_TILE_CAND + tid];
            if (tid == 0) cand_cnt[tg] = n;
        }
    }
}

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
                          const uint32_t* taps, slamfe_level0 l0)
{
    for (int l = 1; l < g.nlevels; ++l) {
        // items = strips of 4 * RS_ROWS rows x blocks of 64 columns; a workgroup
        // covers four frames of the grid's frame axis (see k_resize)
        dim3 grid = frame_grid((g.lv[l].h + 4 * RS_ROWS - 1) / (4 * RS_ROWS) * ((g.lv[l].w + 63) / 64), nframes);
        grid.z = (grid.z + 3) / 4;
        hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, g, l, nframes, pyr, taps, l0);
        LAUNCH_CHECK();
    }
    return 0;
}

int slamfe_launch_pyramid_fused(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* d_in,
                                int stride, size_t frame_stride, int channels, uint8_t* pyr,
                                const uint32_t* taps)
{
    if (g.pt_x <= 0 || nframes <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pyramid_fused, dim3((unsigned)g.pt_x, (unsigned)g.pt_y, (unsigned)nframes), dim3(PF_THREADS),
                       g.pt_lds, s, g, d_in, stride, frame_stride, channels, pyr, taps);
    LAUNCH_CHECK();
    return 0;
}

int slamfe_image_init(void)
{
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyramid_fused),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SLAMFE_PT_LDS_MAX);
}

int slamfe_launch_fast_blur(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                            uint8_t* blur, int thr, int edge, uint32_t* cand, uint32_t* cand_cnt,
                            slamfe_level0 l0)
{
    hipLaunchKernelGGL(k_fast_blur, frame_grid(g.total_groups, nframes), dim3(256), 0, s, g, pyr, blur,
                       nframes, thr, edge, cand, cand_cnt, l0);
    LAUNCH_CHECK();
    return 0;
}
