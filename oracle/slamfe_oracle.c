/* slamfe_oracle.c -- CPU restatement of the reference's feature front-end.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see slamfe_oracle.h).
 *
 * Scalar C, single thread, integer arithmetic wherever the reference path is
 * integer.  Build with -ffp-contract=off: the few float expressions must not
 * be fused into multiply-adds.
 */
#include "slamfe_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slamfe_math.h"
#include "slamfe_tables.h"

static const int8_t k_pattern[1024] = SLAMFE_PATTERN_INIT;
static const int k_ring[16][2] = SLAMFE_FAST_RING_INIT;
static const int k_umax[16] = SLAMFE_UMAX_INIT;
static const int k_gauss[7] = SLAMFE_GAUSS7_INIT;

/* ------------------------------------------------------------------ gray */

/* Init.cc:75, cv::COLOR_BGR2GRAY for 8-bit: fixed-point weights, shift 14. */
void orc_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray, int gstride)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = bgr + (size_t)y * stride;
        uint8_t* d = gray + (size_t)y * gstride;
        for (int x = 0; x < w; ++x) {
            int b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
            d[x] = (uint8_t)((b * SLAMFE_B2Y + g * SLAMFE_G2Y + r * SLAMFE_R2Y +
                              (1 << (SLAMFE_GRAY_SHIFT - 1))) >> SLAMFE_GRAY_SHIFT);
        }
    }
}

/* ------------------------------------------------------- frame resize */

void orc_scaled_size(int sw, int sh, double fx, double fy, int* dw, int* dh)
{
    *dw = slamfe_round_d((double)sw * fx);
    *dh = slamfe_round_d((double)sh * fy);
}

static short sat_short(float v)
{
    int r = slamfe_round_f(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

/* Init.cc:103,126.  Per axis: position f = (float)((d + 0.5) * (1/fx) - 0.5),
 * s = floor(f), weights saturate_cast<short>((1-f')*2048), (f'*2048) in float;
 * horizontal pass in int, vertical pass
 *   ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2. */
void orc_resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, int cn, double fx,
                          double fy, uint8_t* dst, int dstride)
{
    int dw, dh;
    orc_scaled_size(sw, sh, fx, fy, &dw, &dh);
    double scale_x = 1. / fx, scale_y = 1. / fy;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    short* ia = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float f = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(f);
        f -= (float)sx;
        if (sx < 0) { f = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { f = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ia[2 * dx] = sat_short((1.f - f) * 2048.f);
        ia[2 * dx + 1] = sat_short(f * 2048.f);
    }
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
    rows[1] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; ++dy) {
        float f = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(f);
        f -= (float)sy;
        short b0 = sat_short((1.f - f) * 2048.f), b1 = sat_short(f * 2048.f);
        for (int k = 0; k < 2; ++k) {
            int y = sy + k;
            y = y >= 0 ? (y < sh ? y : sh - 1) : 0;
            const uint8_t* S = src + (size_t)y * sstride;
            int* D = rows[k];
            for (int dx = 0; dx < dw; ++dx)
                for (int c = 0; c < cn; ++c) {
                    int sx = xofs[dx] * cn + c;
                    D[dx * cn + c] = dx < xmax ? S[sx] * ia[2 * dx] + S[sx + cn] * ia[2 * dx + 1]
                                               : S[sx] * 2048;
                }
        }
        uint8_t* d = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw * cn; ++x) {
            int v = (((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2;
            d[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs);
    free(ia);
    free(rows[0]);
    free(rows[1]);
}

/* ------------------------------------------------------------------ FAST */

/* Score of a pixel that passed the segment test: with d[k] = v - ring[k],
 * max over the 16 arcs of 9 of min(d) and of min(-d), floored at the
 * threshold, minus one (cornerScore<16> behind Init.cc:76). */
static int fast_corner_score(const uint8_t* p, const int* off, int thr)
{
    int d[25];
    int v = p[0];
    for (int k = 0; k < 16; ++k) d[k] = v - p[off[k]];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    int a0 = thr;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; ++j)
            if (d[k + j] < a) a = d[k + j];
        int m = a < d[k] ? a : d[k];
        if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9];
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] > b) b = d[k + 3];
        for (int j = 4; j <= 5; ++j)
            if (d[k + j] > b) b = d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; ++j)
            if (d[k + j] > b) b = d[k + j];
        int m = b > d[k] ? b : d[k];
        if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9];
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

/* Segment test: 9 contiguous ring pixels all > v+t or all < v-t. */
static int fast_is_corner(const uint8_t* p, const int* off, int thr)
{
    int v = p[0];
    int hi = v + thr, lo = v - thr;
    unsigned bright = 0, dark = 0;
    /* any arc of 9 contains one pixel of each opposite pair: cheap reject */
    int p0 = p[off[0]], p8 = p[off[8]];
    if (!(p0 > hi || p8 > hi || p0 < lo || p8 < lo)) return 0;
    int p4 = p[off[4]], p12 = p[off[12]];
    if (!(p4 > hi || p12 > hi || p4 < lo || p12 < lo)) return 0;
    for (int k = 0; k < 16; ++k) {
        int q = p[off[k]];
        bright |= (unsigned)(q > hi) << k;
        dark |= (unsigned)(q < lo) << k;
    }
    for (int pass = 0; pass < 2; ++pass) {
        unsigned m = pass ? dark : bright;
        m |= m << 16;
        unsigned r = m & (m >> 1);
        r &= r >> 2;
        r &= r >> 4;
        r &= m >> 8;
        if (r & 0xffffu) return 1;
    }
    return 0;
}

void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int thr, uint8_t* score,
                        int sstride)
{
    int off[16];
    if (thr < 0) thr = 0;
    if (thr > 255) thr = 255;
    for (int k = 0; k < 16; ++k) off[k] = k_ring[k][0] + k_ring[k][1] * stride;
    for (int y = 0; y < h; ++y) memset(score + (size_t)y * sstride, 0, (size_t)w);
    for (int y = 3; y < h - 3; ++y) {
        const uint8_t* row = img + (size_t)y * stride;
        uint8_t* srow = score + (size_t)y * sstride;
        for (int x = 3; x < w - 3; ++x)
            if (fast_is_corner(row + x, off, thr))
                srow[x] = (uint8_t)fast_corner_score(row + x, off, thr);
    }
}

int orc_fast9(const uint8_t* img, int w, int h, int stride, int thr, int nonmax, orc_corner* out,
              int cap)
{
    if (w < 7 || h < 7) return 0;
    if (thr < 0) thr = 0;
    if (thr > 255) thr = 255;
    uint8_t* sc = (uint8_t*)malloc((size_t)w * h);
    /* a separate flag map: with thr = 0 a corner may score 0 */
    uint8_t* is = (uint8_t*)calloc((size_t)w * h, 1);
    int off[16];
    for (int k = 0; k < 16; ++k) off[k] = k_ring[k][0] + k_ring[k][1] * stride;
    memset(sc, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; ++y) {
        const uint8_t* row = img + (size_t)y * stride;
        for (int x = 3; x < w - 3; ++x)
            if (fast_is_corner(row + x, off, thr)) {
                is[(size_t)y * w + x] = 1;
                sc[(size_t)y * w + x] = (uint8_t)fast_corner_score(row + x, off, thr);
            }
    }
    int n = 0;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            size_t i = (size_t)y * w + x;
            if (!is[i]) continue;
            int s = sc[i];
            if (nonmax) {
                /* strictly greater than all 8 neighbours (non-corners score 0) */
                if (!(s > sc[i - 1] && s > sc[i + 1] && s > sc[i - w - 1] && s > sc[i - w] &&
                      s > sc[i - w + 1] && s > sc[i + w - 1] && s > sc[i + w] && s > sc[i + w + 1]))
                    continue;
            }
            if (n < cap) {
                out[n].x = x;
                out[n].y = y;
                out[n].score = s;
            }
            ++n;
        }
    free(sc);
    free(is);
    return n;
}

/* --------------------------------------------------------- ORB geometry */

/* Level scale = (float)pow(scaleFactor, level) with scaleFactor held as a
 * double converted from the float 1.2f of Init.cc:65; level size =
 * cvRound(dim * (1.0f / scale)). */
void orc_level_geometry(int w, int h, int nlevels, float scale_factor, int* lw, int* lh,
                        float* lscale)
{
    for (int l = 0; l < nlevels; ++l) {
        float scale = (float)pow((double)scale_factor, (double)l);
        float inv = 1.0f / scale;
        lscale[l] = scale;
        lw[l] = slamfe_round_f((float)w * inv);
        lh[l] = slamfe_round_f((float)h * inv);
    }
}

void orc_level_quotas(int nfeatures, int nlevels, float scale_factor, int* quota)
{
    float factor = (float)(1.0 / (double)scale_factor);
    float nd = (float)nfeatures * (1 - factor) /
               (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        quota[l] = slamfe_round_f(nd);
        sum += quota[l];
        nd *= factor;
    }
    quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* ---------------------------------------------------------------- resize */

/* Per-axis taps of INTER_LINEAR_EXACT: source offset and the 8.8 fixed-point
 * weight of the right/lower tap.  Destinations whose sample position falls
 * before the first or at/after the last source pixel copy the edge pixel. */
static void resize_taps(int ssize, int dsize, int* ofs, int* c1)
{
    double inv_scale = (double)dsize / (double)ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i0 = (int)floor(f);
        if (i0 >= 0 && ssize > 1) {
            if (i0 < ssize - 1) {
                ofs[d] = i0;
                c1[d] = slamfe_round_d((f - (double)i0) * 256.0);
            } else {
                ofs[d] = ssize - 1;
                c1[d] = 0;
            }
        } else {
            ofs[d] = 0;
            c1[d] = 0;
        }
    }
}

void orc_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw,
                             int dh, int dstride)
{
    int* xo = (int*)malloc(sizeof(int) * (size_t)dw * 2);
    int* xc = xo + dw;
    int* yo = (int*)malloc(sizeof(int) * (size_t)dh * 2);
    int* yc = yo + dh;
    resize_taps(sw, dw, xo, xc);
    resize_taps(sh, dh, yo, yc);
    for (int y = 0; y < dh; ++y) {
        const uint8_t* r0 = src + (size_t)yo[y] * sstride;
        const uint8_t* r1 = src + (size_t)(yo[y] + (yc[y] ? 1 : 0)) * sstride;
        unsigned w1 = (unsigned)yc[y], w0 = 256u - w1;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; ++x) {
            int o = xo[x];
            int o1 = o + (xc[x] ? 1 : 0);
            unsigned c1 = (unsigned)xc[x], c0 = 256u - c1;
            unsigned h0 = r0[o] * c0 + r0[o1] * c1; /* 8.8 */
            unsigned h1 = r1[o] * c0 + r1[o1] * c1;
            unsigned v = (h0 * w0 + h1 * w1 + (1u << 15)) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(xo);
    free(yo);
}

void orc_build_pyramid(const uint8_t* gray, int w, int h, int stride, int nlevels,
                       float scale_factor, uint8_t** levels)
{
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    float ls[ORC_MAX_LEVELS];
    orc_level_geometry(w, h, nlevels, scale_factor, lw, lh, ls);
    for (int y = 0; y < h; ++y) memcpy(levels[0] + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
    /* each level is resized from the previous level, not from level 0 */
    for (int l = 1; l < nlevels; ++l)
        orc_resize_linear_exact(levels[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], levels[l], lw[l],
                                lh[l], lw[l]);
}

/* ------------------------------------------------------------------ blur */

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

void orc_gaussian_blur7(const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride)
{
    /* row pass into 32-bit sums (<= 257 * 255), then column pass; borders by
     * REFLECT_101 index, the interior without index arithmetic */
    int* rows = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * stride;
        int* r = rows + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            if (x >= 3 && x < w - 3) {
                r[x] = k_gauss[0] * (s[x - 3] + s[x + 3]) + k_gauss[1] * (s[x - 2] + s[x + 2]) +
                       k_gauss[2] * (s[x - 1] + s[x + 1]) + k_gauss[3] * s[x];
            } else {
                int acc = 0;
                for (int k = -3; k <= 3; ++k) acc += k_gauss[k + 3] * s[reflect101(x + k, w)];
                r[x] = acc;
            }
        }
    }
    for (int y = 0; y < h; ++y) {
        const int* r[7];
        for (int k = -3; k <= 3; ++k) r[k + 3] = rows + (size_t)reflect101(y + k, h) * w;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int acc = k_gauss[0] * (r[0][x] + r[6][x]) + k_gauss[1] * (r[1][x] + r[5][x]) +
                      k_gauss[2] * (r[2][x] + r[4][x]) + k_gauss[3] * r[3][x];
            int v = (acc + (1 << 15)) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(rows);
}

/* ----------------------------------------------------------- orientation */

void orc_ic_moments(const uint8_t* img, int stride, int cx, int cy, int* m10_out, int* m01_out)
{
    const uint8_t* c = img + (size_t)cy * stride + cx;
    int m01 = 0, m10 = 0;
    for (int u = -SLAMFE_HALF_PATCH; u <= SLAMFE_HALF_PATCH; ++u) m10 += u * c[u];
    for (int v = 1; v <= SLAMFE_HALF_PATCH; ++v) {
        int vsum = 0, d = k_umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * stride], vm = c[u - v * stride];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    *m10_out = m10;
    *m01_out = m01;
}

float orc_fast_atan2(float y, float x) { return slamfe_fast_atan2(y, x); }

float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy)
{
    int m10, m01;
    orc_ic_moments(img, stride, cx, cy, &m10, &m01);
    return slamfe_fast_atan2((float)m01, (float)m10);
}

void orc_sincos_deg(float angle_deg, int trig_mode, float* s, float* c)
{
    if (trig_mode == ORC_TRIG_LIBM) {
        float a = angle_deg * (float)(3.14159265358979323846 / 180.0);
        *c = cosf(a);
        *s = sinf(a);
    } else {
        slamfe_sincos_deg(angle_deg, s, c);
    }
}

/* ------------------------------------------------------------ descriptor */

void orc_brief256(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg,
                  int trig_mode, uint8_t desc[32])
{
    float a, b;
    orc_sincos_deg(angle_deg, trig_mode, &b, &a); /* a = cos, b = sin */
    const uint8_t* center = blurred + (size_t)cy * stride + cx;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int j = 0; j < 8; ++j) {
            const int8_t* t = k_pattern + (i * 8 + j) * 4;
            float px0 = (float)t[0], py0 = (float)t[1], px1 = (float)t[2], py1 = (float)t[3];
            float x0 = px0 * a - py0 * b, y0 = px0 * b + py0 * a;
            float x1 = px1 * a - py1 * b, y1 = px1 * b + py1 * a;
            int t0 = center[slamfe_round_f(y0) * stride + slamfe_round_f(x0)];
            int t1 = center[slamfe_round_f(y1) * stride + slamfe_round_f(x1)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* -------------------------------------------------------------- extract */

static int cmp_int_desc(const void* a, const void* b)
{
    int x = *(const int*)a, y = *(const int*)b;
    return (x < y) - (x > y);
}

int orc_extract(const slamfe_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                int trig_mode, slamfe_keypoint* kps, uint8_t* desc, int cap)
{
    int nlevels = p->mode == SLAMFE_MODE_PROVIDED ? 1 : p->nlevels;
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS], quota[ORC_MAX_LEVELS];
    float ls[ORC_MAX_LEVELS];
    uint8_t* levels[ORC_MAX_LEVELS];
    uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
    if (channels == 3) orc_bgr2gray(img, w, h, stride, gray, w);
    else
        for (int y = 0; y < h; ++y) memcpy(gray + (size_t)y * w, img + (size_t)y * stride, (size_t)w);

    orc_level_geometry(w, h, nlevels, p->scale_factor, lw, lh, ls);
    if (p->mode == SLAMFE_MODE_ORB) orc_level_quotas(p->nfeatures, nlevels, p->scale_factor, quota);
    for (int l = 0; l < nlevels; ++l) levels[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l] + 1);
    orc_build_pyramid(gray, w, h, w, nlevels, p->scale_factor, levels);

    int edge = p->edge_threshold;
    int total = 0;
    size_t corner_cap = (size_t)w * h / 4 + 16;
    orc_corner* cand = (orc_corner*)malloc(sizeof(orc_corner) * corner_cap);
    int* resp = (int*)malloc(sizeof(int) * corner_cap);
    uint8_t* blurred = (uint8_t*)malloc((size_t)w * h);

    for (int l = 0; l < nlevels; ++l) {
        int W = lw[l], H = lh[l];
        int n = orc_fast9(levels[l], W, H, W, p->fast_threshold, 1, cand, (int)corner_cap);
        /* KeyPointsFilter::runByImageBorder: keep edge <= x < W-edge, same for y */
        int m = 0;
        if (edge > 0 && (H <= edge * 2 || W <= edge * 2)) n = 0;
        for (int i = 0; i < n; ++i)
            if (cand[i].x >= edge && cand[i].x < W - edge && cand[i].y >= edge &&
                cand[i].y < H - edge)
                cand[m++] = cand[i];
        n = m;
        if (p->mode == SLAMFE_MODE_ORB) {
            /* KeyPointsFilter::retainBest: keep everything whose response is >=
             * the quota-th best response (ties at the cut are all kept) */
            int q = quota[l];
            if (n > q) {
                if (q == 0) n = 0;
                else {
                    for (int i = 0; i < n; ++i) resp[i] = cand[i].score;
                    qsort(resp, (size_t)n, sizeof(int), cmp_int_desc);
                    int cut = resp[q - 1];
                    m = 0;
                    for (int i = 0; i < n; ++i)
                        if (cand[i].score >= cut) cand[m++] = cand[i];
                    n = m;
                }
            }
        }
        if (n == 0) continue;
        orc_gaussian_blur7(levels[l], W, H, W, blurred, W);
        float scale = ls[l];
        for (int i = 0; i < n; ++i, ++total) {
            if (total >= cap) continue;
            slamfe_keypoint* k = kps + total;
            float angle;
            if (p->mode == SLAMFE_MODE_ORB) {
                angle = orc_ic_angle(levels[l], W, cand[i].x, cand[i].y);
                k->x = (float)cand[i].x * scale;
                k->y = (float)cand[i].y * scale;
                k->size = (float)p->patch_size * scale;
                k->octave = l;
            } else {
                angle = -1.f;
                k->x = (float)cand[i].x;
                k->y = (float)cand[i].y;
                k->size = 7.f;
                k->octave = 0;
            }
            k->angle = angle;
            k->response = (float)cand[i].score;
            k->class_id = -1;
            /* descriptor centre: cvRound(pt * (1/scale)) in the level image */
            float inv = 1.f / scale;
            int cx = slamfe_round_f(k->x * inv), cy = slamfe_round_f(k->y * inv);
            orc_brief256(blurred, W, cx, cy, angle, trig_mode, desc + (size_t)total * 32);
        }
    }
    for (int l = 0; l < nlevels; ++l) free(levels[l]);
    free(gray);
    free(cand);
    free(resp);
    free(blurred);
    return total;
}

/* --------------------------------------------------------------- matching */

static inline int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int orc_hamming256(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }

/* BFMatcher(NORM_HAMMING, crossCheck=false)::knnMatch, k = 2 (Init.cc:66,
 * Init.cc:131): scan train rows in ascending order; a candidate enters the
 * sorted 2-slot list only past slots with a strictly larger distance, so equal
 * distances keep the lower train index first.  Then the ratio test of
 * Init.cc:146-147 in float. */
void orc_match_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt, float ratio,
                    slamfe_match2* out)
{
    for (int i = 0; i < nq; ++i) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        const uint8_t* q = query + (size_t)i * 32;
        for (int j = 0; j < nt; ++j) {
            int d = hamming256(q, train + (size_t)j * 32);
            if (d < d1) {
                if (d < d0) {
                    d1 = d0; i1 = i0;
                    d0 = d; i0 = j;
                } else {
                    d1 = d; i1 = j;
                }
            }
        }
        slamfe_match2* m = out + i;
        memset(m, 0, sizeof(*m));
        m->idx0 = i0;
        m->idx1 = i1;
        m->d0 = (uint16_t)(i0 >= 0 ? d0 : 0);
        m->d1 = (uint16_t)(i1 >= 0 ? d1 : 0);
        m->nfound = (uint8_t)((i0 >= 0) + (i1 >= 0));
        m->ratio_pass = (uint8_t)(m->nfound == 2 && (float)d0 < ratio * (float)d1);
    }
}

/* ------------------------------------------------- search by projection */

/* Init::euclideanDistance, Init.cc:389-391: float differences, squares and
 * square root in double, result narrowed to float. */
static float euclid_ref(float ax, float ay, float bx, float by)
{
    float dx = ax - bx, dy = ay - by;
    return (float)sqrt(pow((double)dx, 2) + pow((double)dy, 2));
}

void orc_search_by_projection(const float* kp_xy, const uint8_t* frame_desc, const int32_t* kp_idx,
                              int n_inl, const float* uv, const uint8_t* point_desc,
                              const int32_t* point_ids, int R, float radius, int max_hamming,
                              int32_t* matched_point_id)
{
    /* rPoints / rPointsIdx of Init.cc:199-200 as index lists we can erase from */
    int* live = (int*)malloc(sizeof(int) * (size_t)(R > 0 ? R : 1));
    int nlive = R;
    for (int j = 0; j < R; ++j) live[j] = j;
    for (int i = 0; i < n_inl; ++i) {
        int idx = kp_idx[i];
        matched_point_id[i] = -1;
        if (nlive == 0) continue; /* Init.cc:233 */
        int flag = 0, mIdx = 0, mPoint = 0;
        double minDistance = DBL_MAX;
        for (int j = 0; j < nlive; ++j) {
            int k = live[j];
            double distance = (double)hamming256(point_desc + (size_t)k * 32,
                                                     frame_desc + (size_t)idx * 32);
            float e = euclid_ref(uv[2 * k], uv[2 * k + 1], kp_xy[2 * idx], kp_xy[2 * idx + 1]);
            if (distance <= max_hamming && e <= radius) {
                if (distance < minDistance) {
                    minDistance = distance;
                    mPoint = k;
                    mIdx = j;
                }
                flag = 1;
            }
        }
        if (flag) {
            matched_point_id[i] = point_ids[mPoint];
            memmove(live + mIdx, live + mIdx + 1, sizeof(int) * (size_t)(nlive - mIdx - 1));
            --nlive;
        }
    }
    free(live);
}


/* ---------------------------------------------------------------------------
 * Map-point reprojection, Init.cc:203-218.
 *
 * keyframe->getPose().inv() (Init.cc:206): cv::invert with DECOMP_LU on a 4x4
 * CV_32F matrix: Gaussian elimination with partial pivoting in float on a copy
 * of the matrix and an identity right-hand side, the pivot's reciprocal kept in
 * place of the pivot, back-substitution multiplying by it.  A pivot below
 * 10*FLT_EPSILON makes the matrix singular: OpenCV then returns all zeros.
 * K4x4 * kf * pt (Init.cc:208) is evaluated left to right; OpenCV's small-matrix
 * product sums the four float products of a row left to right in float.
 * pt /= s (Init.cc:211, 213) scales by the float value of 1.0/s computed in
 * double.  No fused multiply-add anywhere (-ffp-contract=off).
 * ------------------------------------------------------------------------- */
int orc_invert4x4(const float* src, float* dst)
{
    float A[16], B[16];
    memcpy(A, src, sizeof A);
    for (int i = 0; i < 16; ++i) B[i] = (i % 5 == 0) ? 1.f : 0.f;
    const float eps = FLT_EPSILON * 10;
    for (int i = 0; i < 4; ++i) {
        int k = i;
        for (int j = i + 1; j < 4; ++j)
            if (fabsf(A[j * 4 + i]) > fabsf(A[k * 4 + i])) k = j;
        if (fabsf(A[k * 4 + i]) < eps) {
            memset(dst, 0, 16 * sizeof(float));
            return 0;
        }
        if (k != i) {
            for (int j = i; j < 4; ++j) { float t = A[i * 4 + j]; A[i * 4 + j] = A[k * 4 + j]; A[k * 4 + j] = t; }
            for (int j = 0; j < 4; ++j) { float t = B[i * 4 + j]; B[i * 4 + j] = B[k * 4 + j]; B[k * 4 + j] = t; }
        }
        float d = -1 / A[i * 4 + i];
        for (int j = i + 1; j < 4; ++j) {
            float alpha = A[j * 4 + i] * d;
            for (int c = i + 1; c < 4; ++c) A[j * 4 + c] += alpha * A[i * 4 + c];
            for (int c = 0; c < 4; ++c) B[j * 4 + c] += alpha * B[i * 4 + c];
        }
        A[i * 4 + i] = -d;
    }
    for (int i = 3; i >= 0; --i)
        for (int j = 0; j < 4; ++j) {
            float sum = B[i * 4 + j];
            for (int c = i + 1; c < 4; ++c) sum -= A[i * 4 + c] * B[c * 4 + j];
            B[i * 4 + j] = sum * A[i * 4 + i];
        }
    memcpy(dst, B, sizeof B);
    return 1;
}

void orc_matmul4x4(const float* a, const float* b, float* d)
{
    float out[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            out[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j] +
                             a[i * 4 + 3] * b[12 + j];
    memcpy(d, out, sizeof out);
}

int orc_reproject(const float* pose, const float* K, const float* xyz, int npoints, float th, float W,
                  float H, float* uv, int32_t* idx)
{
    float kf[16], K4[16], M[16];
    orc_invert4x4(pose, kf);
    for (int i = 0; i < 16; ++i) K4[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K4[r * 4 + c] = K[r * 3 + c];
    orc_matmul4x4(K4, kf, M);
    int n = 0;
    for (int i = 0; i < npoints; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (z < th || fabs((double)(z - th)) > 20) continue; /* Init.cc:205 */
        float p[4];
        for (int r = 0; r < 4; ++r)
            p[r] = M[r * 4] * x + M[r * 4 + 1] * y + M[r * 4 + 2] * z + M[r * 4 + 3] * 1.f;
        float s = (float)(1. / (double)p[3]); /* Init.cc:211 */
        for (int r = 0; r < 4; ++r) p[r] = p[r] * s;
        s = (float)(1. / (double)p[2]); /* Init.cc:213 */
        for (int r = 0; r < 4; ++r) p[r] = p[r] * s;
        if (p[0] <= 0 || p[0] >= W || p[1] <= 0 || p[1] >= H) continue; /* Init.cc:214-215 */
        idx[n] = i;
        uv[2 * n] = p[0];
        uv[2 * n + 1] = p[1];
        ++n;
    }
    return n;
}


/* Init::triangulate (Init.cc:5-31) for n point pairs.  The arithmetic is the
 * shared float contract of slamfe_math.h (as for fastAtan2 and cos/sin): the
 * one-sided Jacobi SVD OpenCV uses for small float matrices. */
void orc_triangulate(const float* P1, const float* P2, const float* pts1, const float* pts2, int n,
                     float* xyz)
{
    for (int i = 0; i < n; ++i)
        slamfe_triangulate_point(P1, P2, pts1[2 * i], pts1[2 * i + 1], pts2[2 * i], pts2[2 * i + 1],
                                 xyz + 3 * (size_t)i);
}

void orc_svd4_last_row(const float* A, float* vt3) { slamfe_svd4_last_row(A, vt3); }
