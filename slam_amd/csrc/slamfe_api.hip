// slamfe_api.hip -- host side of libslamfe.so: the C ABI of include/slamfe.h.
//
// A context owns one GPU, one stream and every device buffer the kernels use.
// Device memory is laid out for a batch of `max_batch` frames:
//
//   pyr / blur      [max_batch][frame_bytes]    unblurred / blurred pyramids,
//                                               level l at lv[l].off, rows of
//                                               lv[l].pitch bytes (64-B aligned)
//   cand            [max_batch][tiles][256] u32 FAST survivors per 64x16 tile
//   band_hist       [max_batch][bands][256]     score histogram of every tile row
//   frame slots     [frame_slots][max_kp]       keys, levels, cv::KeyPoint
//                                               records, 32-byte descriptor rows
//   map_desc        [map_capacity][32]          mirror of Point::desc
#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "slamfe.h"
#include "slamfe_dev.h"
#include "slamfe_kernels.h"
#include "slamfe_math.h"

// Calls with fewer frames than this take the fused, launch-poor kernels.
#define SLAMFE_FEW_FRAMES 8
#define SLAMFE_FEW_PIXELS 2200000 /* ... and at most this many pixels in the call */

namespace {

struct Scratch {
    void* p = nullptr;
    size_t bytes = 0;
};

} // namespace

// Fixed device memory for slamfe_match_shard_keys_static: the per-chunk top-2
// partials of one matching job.  Every GPU the library runs on holds its own
// copy of this array at its own address until the process ends; the name below
// is only the host-side handle of the symbol, never a device address: host code
// uses the per-device pointer that static_scratch_for() looks up.
#define SLAMFE_STATIC_SCRATCH_BYTES ((size_t)4 << 20)
static __device__ __align__(256) unsigned char scratch[SLAMFE_STATIC_SCRATCH_BYTES];

struct slamfe_ctx {
    slamfe_params p;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // geometry of the frame size in use
    int gw = 0, gh = 0;
    slamfe_dev_geom geom;
    std::vector<uint32_t> taps_host;
    size_t sort_lds = 0;
    // upper bounds from (max_width, max_height)
    size_t cap_frame_bytes = 0;
    int cap_tiles = 0, cap_bands = 0;
    size_t cap_taps = 0;

    uint8_t *pyr = nullptr, *blur = nullptr, *in_stage = nullptr;
    // pinned host staging for the host-pointer calls (allocated on first use):
    // pageable hipMemcpy costs milliseconds per call, pinned copies microseconds
    uint8_t *h_in = nullptr, *h_out = nullptr;
    // set around run_extract by a host-pointer call of a few frames: the last kernel
    // then writes counts and rows into these pinned arrays as well
    slamfe_host_rows host_rows = {nullptr, nullptr, nullptr};
    bool host_rows_done = false; // ... and says so here
    int32_t* h_cnt = nullptr; // allocated at creation
    uint8_t* h_pin = nullptr;  // general pinned bounce buffer (grows)
    size_t h_pin_bytes = 0;
    uint32_t* taps = nullptr;
    uint32_t *cand = nullptr, *cand_cnt = nullptr, *band_hist = nullptr, *band_off = nullptr,
             *level_kept = nullptr;
    int32_t* level_cutoff = nullptr;

    int max_kp = 0, nslots = 0;
    uint32_t *kp_key = nullptr, *kp_level = nullptr;
    int32_t* kp_mom = nullptr; // [slots][max_kp][2] intensity-centroid moments
    float* kp_rot = nullptr;   // [slots][max_kp][2] cos, sin of the keypoint angle
    int32_t* kp_count = nullptr;
    slamfe_keypoint* kps = nullptr;
    uint8_t* desc = nullptr;
    std::vector<uint64_t> slot_handle;
    std::vector<int32_t> slot_count_host; // keypoint count of a slot once the host has read it, else -1
    uint64_t next_seq = 1; // handle = sequence number; slot = seq % nslots

    slamfe_match2* match_out = nullptr; // [max_kp]
    int fm_qpt = 1, fm_chunk = 64, fm_nchunks = 1; // plan for frame-vs-frame matching
    Scratch match_part, match_q, match_t, match_o;
    uint8_t* map_desc = nullptr;
    Scratch pj_in, pj_cand, pj_candn, pj_cells, pj_sorted, pj_cnt, pj_rank;
    // The grid over the device-resident visible set (pj_cells, pj_sorted) is built by
    // slamfe_reproject for the radius of the most recent search (the reference's
    // constant before any search, Init.cc:242) and reused by the search if the
    // radius is the same.
    float pj_grid_radius = 32.0f;
    bool pj_grid_ready = false;
    hipEvent_t ev_count = nullptr; // marks the arrival of a count on the host
    Scratch ing_taps, ing_bgr;
    // reprojection: coordinates table (row = Point::id), scratch, and the visible
    // set the last slamfe_reproject call left on the device
    float* map_xyz = nullptr;
    Scratch rp_uv_all, rp_mask, rp_off, rp_total, vis_uv, vis_idx;
    Scratch tri_in, tri_out;
    int vis_count = -1;
    // where level 0 of the most recent extraction lives (the pyramid buffer, or the
    // caller's frames used in place)
    slamfe_level0 cur_l0 = {nullptr, 0};
    long long few_pixels = SLAMFE_FEW_PIXELS;
    void* static_scratch = nullptr; // this GPU's copy of `scratch` (device address)
    // in-pipeline stage probe (slamfe_debug_probe): events around one stage of
    // every extraction / batched match that is queued while the probe is set
    int probe_stage = 0;
    std::vector<hipEvent_t> probe_ev; // pairs: begin, end
    size_t probe_used = 0;            // events recorded so far
};

namespace {

int fail(slamfe_ctx* c, int code, const char* fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), SLAMFE_E_DEVICE, "%s failed: %s (%s:%d)", #call,               \
                        hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

#define KCHK(c, call)                                                                       \
    do {                                                                                    \
        int e_ = (call);                                                                    \
        if (e_ != 0)                                                                        \
            return fail((c), SLAMFE_E_DEVICE, "%s failed: %s", #call,                       \
                        hipGetErrorString((hipError_t)e_));                                 \
    } while (0)

// Host-side phase times of a call, printed to stderr when the environment
// variable SLAMFE_TRACE is set (a tuning aid; no effect otherwise).
struct PhaseTrace {
    bool on;
    const char* what;
    std::chrono::steady_clock::time_point t0, last;
    std::string line;
    explicit PhaseTrace(const char* name) : on(getenv("SLAMFE_TRACE") != nullptr), what(name)
    {
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char* phase)
    {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, " %s %.1f", phase, std::chrono::duration<double, std::micro>(now - last).count());
        line += buf;
        last = now;
    }
    ~PhaseTrace()
    {
        if (on)
            fprintf(stderr, "[slamfe %s]%s | total %.1f us\n", what, line.c_str(),
                    std::chrono::duration<double, std::micro>(last - t0).count());
    }
};

// Events around `stage` while the probe is set on it (at most 64 intervals are kept).
struct ProbeScope {
    slamfe_ctx* c;
    bool on;
    ProbeScope(slamfe_ctx* ctx, int stage) : c(ctx), on(false)
    {
        if (c->probe_stage != stage || c->probe_used + 2 > c->probe_ev.size()) return;
        on = hipEventRecord(c->probe_ev[c->probe_used], c->stream) == hipSuccess;
    }
    ~ProbeScope()
    {
        if (!on) return;
        if (hipEventRecord(c->probe_ev[c->probe_used + 1], c->stream) == hipSuccess) c->probe_used += 2;
    }
};

int grow(slamfe_ctx* c, Scratch& s, size_t bytes)
{
    if (bytes <= s.bytes) return SLAMFE_OK;
    if (s.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(s.p));
        s.p = nullptr;
        s.bytes = 0;
    }
    size_t want = bytes + bytes / 2 + 256;
    HIPCHK(c, hipMalloc(&s.p, want));
    s.bytes = want;
    return SLAMFE_OK;
}

// Device address of `scratch` on `device` (which must be the current device):
// looked up once per device, kept for the life of the process.
#define SLAMFE_MAX_DEVICES 64
std::mutex g_scratch_mutex;
void* g_scratch_ptr[SLAMFE_MAX_DEVICES];

int static_scratch_for(slamfe_ctx* c, int device, void** out)
{
    if (device < 0 || device >= SLAMFE_MAX_DEVICES)
        return fail(c, SLAMFE_E_INVALID, "device ordinal %d; at most %d devices", device, SLAMFE_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIPCHK(c, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(c, SLAMFE_E_DEVICE, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return SLAMFE_OK;
}

// Pinned bounce buffer of at least `bytes` for the host-pointer calls.
int pin_reserve(slamfe_ctx* c, size_t bytes)
{
    if (bytes <= c->h_pin_bytes) return SLAMFE_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_pin) HIPCHK(c, hipHostFree(c->h_pin));
    c->h_pin = nullptr;
    c->h_pin_bytes = 0;
    size_t want = bytes + bytes / 2 + 4096;
    HIPCHK(c, hipHostMalloc((void**)&c->h_pin, want, 0));
    c->h_pin_bytes = want;
    return SLAMFE_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Per-axis taps of the fixed-point bilinear resize: source offset and the 8.8
// weight of the right/lower tap, packed as ofs | c1 << 16.  The sample
// position is scale * (d + 0.5) - 0.5 in double with scale = 1 / (dst / src).
void resize_taps(int ssize, int dsize, uint32_t* out)
{
    double inv_scale = (double)dsize / (double)ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i0 = (int)floor(f);
        uint32_t ofs = 0, c1 = 0;
        if (i0 >= 0 && ssize > 1) {
            if (i0 < ssize - 1) {
                ofs = (uint32_t)i0;
                c1 = (uint32_t)slamfe_round_d((f - (double)i0) * 256.0);
            } else {
                ofs = (uint32_t)(ssize - 1);
            }
        }
        out[d] = ofs | (c1 << 16);
    }
}

// Range tables of the one-launch pyramid kernel.  Level 0 is cut into tiles of
// SLAMFE_PT_W x SLAMFE_PT_H pixels; tile t owns, on every level l, the columns
// [own0, own1) with own0 = floor(t * PT_W * w_l / w_0) rounded down to a multiple
// of 4 (a partition of the level; dword stores of neighbouring tiles never
// overlap) and has to compute the columns [need0, need1): its own ones plus
// whatever the taps of its needed columns on level l + 1 read.  Rows likewise
// (no rounding).  Entry (l, t) of a table is two words: need0 | need1 << 16,
// own0 | own1 << 16.  Appended to the tap buffer; sets g->pt_* (pt_x = 0 if
// the kernel's LDS budget does not hold the largest tile).
void build_fused_pyramid_tables(slamfe_dev_geom* g, std::vector<uint32_t>* taps)
{
    g->pt_x = g->pt_y = 0;
    g->pt_lds = 0;
    const int L = g->nlevels;
    if (L < 2) return;
    for (int l = 0; l < L; ++l)
        if (g->lv[l].w < 1 || g->lv[l].h < 1) return;
    const int ntx = (g->lv[0].w + SLAMFE_PT_W - 1) / SLAMFE_PT_W, nty = (g->lv[0].h + SLAMFE_PT_H - 1) / SLAMFE_PT_H;
    size_t max_need[2][SLAMFE_MAX_LEVELS] = {};
    std::vector<uint32_t> tab[2];
    for (int axis = 0; axis < 2; ++axis) {
        const int nt = axis ? nty : ntx, tile = axis ? SLAMFE_PT_H : SLAMFE_PT_W;
        tab[axis].assign((size_t)L * nt * 2, 0u);
        for (int t = 0; t < nt; ++t) {
            int need0[SLAMFE_MAX_LEVELS], need1[SLAMFE_MAX_LEVELS];
            for (int l = L - 1; l >= 0; --l) {
                const int dim = axis ? g->lv[l].h : g->lv[l].w, dim0 = axis ? g->lv[0].h : g->lv[0].w;
                auto cut = [&](int k) {
                    if (k >= nt) return dim;
                    long long v = (long long)k * tile * dim / dim0;
                    if (!axis) v &= ~3LL;
                    return (int)(v > dim ? dim : v);
                };
                const int own0 = cut(t), own1 = cut(t + 1);
                int n0 = own0, n1 = own1;
                if (l + 1 < L && need1[l + 1] > need0[l + 1]) {
                    const uint32_t* tp = taps->data() + (axis ? g->lv[l + 1].tap_y : g->lv[l + 1].tap_x);
                    const uint32_t a = tp[need0[l + 1]], b = tp[need1[l + 1] - 1];
                    int s0 = (int)(a & 0xffffu), s1 = (int)(b & 0xffffu) + ((b >> 16) ? 1 : 0) + 1;
                    if (s1 > dim) s1 = dim;
                    if (n1 <= n0) { n0 = s0; n1 = s1; }
                    else { n0 = std::min(n0, s0); n1 = std::max(n1, s1); }
                }
                need0[l] = n0;
                need1[l] = n1;
                tab[axis][((size_t)l * nt + t) * 2] = (uint32_t)n0 | ((uint32_t)n1 << 16);
                tab[axis][((size_t)l * nt + t) * 2 + 1] = (uint32_t)own0 | ((uint32_t)own1 << 16);
                max_need[axis][l] = std::max(max_need[axis][l], (size_t)(n1 > n0 ? n1 - n0 : 0));
            }
        }
    }
    // LDS: 64 words of ranges, the tap cache, two level buffers (even / odd levels)
    size_t tapc = 0, buf[2] = {0, 0};
    for (int l = 0; l < L; ++l) {
        if (l > 0) tapc += max_need[0][l] + max_need[1][l];
        const size_t pitch = (max_need[0][l] + 3) / 4 * 4 + 4;
        buf[l & 1] = std::max(buf[l & 1], pitch * max_need[1][l]);
    }
    const size_t off_taps = 64 * 4, off_a = off_taps + tapc * 4, off_b = off_a + (buf[0] + 15) / 16 * 16,
                 total = off_b + (buf[1] + 15) / 16 * 16;
    if (total > SLAMFE_PT_LDS_MAX) return;
    g->ptab_x = (uint32_t)taps->size();
    taps->insert(taps->end(), tab[0].begin(), tab[0].end());
    g->ptab_y = (uint32_t)taps->size();
    taps->insert(taps->end(), tab[1].begin(), tab[1].end());
    g->pt_x = ntx;
    g->pt_y = nty;
    g->pt_lds = (uint32_t)total;
    g->pt_off_taps = (uint32_t)off_taps;
    g->pt_off_a = (uint32_t)off_a;
    g->pt_off_b = (uint32_t)off_b;
}

// Level sizes, scales and retain-best quotas as cv::ORB computes them
// (Init.cc:65: nfeatures, 1.2f, 8 levels), plus the device layout.
void build_geometry(const slamfe_params& p, int w, int h, slamfe_dev_geom* g,
                    std::vector<uint32_t>* taps)
{
    memset(g, 0, sizeof *g);
    int nlevels = p.mode == SLAMFE_MODE_PROVIDED ? 1 : p.nlevels;
    g->nlevels = nlevels;
    // quotas
    int quota[SLAMFE_MAX_LEVELS];
    if (p.mode == SLAMFE_MODE_ORB) {
        float factor = (float)(1.0 / (double)p.scale_factor);
        float nd = (float)p.nfeatures * (1 - factor) /
                   (1 - (float)pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; ++l) {
            quota[l] = slamfe_round_f(nd);
            sum += quota[l];
            nd *= factor;
        }
        quota[nlevels - 1] = p.nfeatures - sum > 0 ? p.nfeatures - sum : 0;
    } else {
        quota[0] = INT32_MAX;
    }
    uint32_t off = 0;
    int tiles = 0, bands = 0, groups = 0;
    size_t ntaps = 0;
    for (int l = 0; l < nlevels; ++l) {
        slamfe_dev_level& L = g->lv[l];
        float scale = (float)pow((double)p.scale_factor, (double)l);
        float inv = 1.0f / scale;
        L.scale = scale;
        L.w = slamfe_round_f((float)w * inv);
        L.h = slamfe_round_f((float)h * inv);
        L.pitch = (L.w + 63) / 64 * 64;
        L.quota = quota[l];
        L.off = off;
        // rows rounded up to 8: the blurred level is stored in tiles of 16 x 8 pixels
        off += (uint32_t)L.pitch * (uint32_t)((L.h + 7) & ~7);
        off = (off + 255u) & ~255u;
        L.tiles_x = (L.w + SLAMFE_TILE_W - 1) / SLAMFE_TILE_W;
        L.tiles_y = (L.h + SLAMFE_TILE_H - 1) / SLAMFE_TILE_H;
        L.tile_base = tiles;
        L.band_base = bands;
        L.group_base = groups;
        groups += L.tiles_x * ((L.tiles_y + SLAMFE_FB_TILES - 1) / SLAMFE_FB_TILES);
        tiles += L.tiles_x * L.tiles_y;
        bands += L.tiles_y;
        // column taps are read four at a time: 4-entry alignment, zero padding
        L.tap_x = (uint32_t)ntaps;
        ntaps += ((size_t)L.w + 3) / 4 * 4;
        L.tap_y = (uint32_t)ntaps;
        ntaps += ((size_t)L.h + 3) / 4 * 4;
    }
    g->frame_bytes = off;
    g->total_tiles = tiles;
    g->total_bands = bands;
    g->total_groups = groups;
    if (taps) {
        taps->assign(ntaps, 0u);
        for (int l = 1; l < nlevels; ++l) {
            resize_taps(g->lv[l - 1].w, g->lv[l].w, taps->data() + g->lv[l].tap_x);
            resize_taps(g->lv[l - 1].h, g->lv[l].h, taps->data() + g->lv[l].tap_y);
        }
        build_fused_pyramid_tables(g, taps);
    }
}

size_t sort_lds_bytes_for(const slamfe_dev_geom& g)
{
    size_t n = 1;
    for (int l = 0; l < g.nlevels; ++l) {
        size_t need = (size_t)g.lv[l].tiles_x * SLAMFE_TILE_CAND;
        while (n < need) n <<= 1;
    }
    return n * sizeof(uint32_t);
}

int ensure_geometry(slamfe_ctx* c, int w, int h)
{
    if (w == c->gw && h == c->gh) return SLAMFE_OK;
    if (w < 1 || h < 1 || w > c->p.max_width || h > c->p.max_height)
        return fail(c, SLAMFE_E_INVALID, "frame %dx%d outside the context's maximum %dx%d", w, h,
                    c->p.max_width, c->p.max_height);
    slamfe_dev_geom g;
    std::vector<uint32_t> taps;
    build_geometry(c->p, w, h, &g, &taps);
    if (g.frame_bytes > c->cap_frame_bytes || g.total_tiles > c->cap_tiles ||
        g.total_bands > c->cap_bands || taps.size() > c->cap_taps)
        return fail(c, SLAMFE_E_CAPACITY, "internal: geometry of %dx%d exceeds buffers", w, h);
    // the previous taps may still be in use by queued kernels
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->taps_host.swap(taps);
    if (!c->taps_host.empty())
        HIPCHK(c, hipMemcpy(c->taps, c->taps_host.data(), c->taps_host.size() * sizeof(uint32_t),
                            hipMemcpyHostToDevice));
    c->geom = g;
    c->sort_lds = sort_lds_bytes_for(g);
    c->gw = w;
    c->gh = h;
    return SLAMFE_OK;
}

int slot_of(const slamfe_ctx* c, uint64_t handle)
{
    if (handle == 0) return -1;
    int s = (int)(handle % (uint64_t)c->nslots);
    return c->slot_handle[s] == handle ? s : -1;
}

// Does a call of n frames of w x h take the launch-poor kernels?  Only where the
// whole call is small enough to be bound by launch latency: large frames fill
// the chip on their own and the throughput kernels are faster there.
bool is_few(const slamfe_ctx* c, int n, int w, int h)
{
    return n < SLAMFE_FEW_FRAMES && (long long)n * w * h <= c->few_pixels;
}

// Queue the whole extraction of n device-resident frames.
// d_imgs == nullptr: level 0 of the pyramid has already been written (scaled ingest).
int run_extract(slamfe_ctx* c, const uint8_t* d_imgs, int n, int w, int h, int stride,
                size_t frame_stride, int channels, uint64_t* handles)
{
    if (n < 1 || n > c->p.max_batch)
        return fail(c, SLAMFE_E_INVALID, "batch of %d frames; context allows 1..%d", n,
                    c->p.max_batch);
    if (channels != 1 && channels != 3)
        return fail(c, SLAMFE_E_INVALID, "channels must be 1 (gray) or 3 (BGR), got %d", channels);
    if (stride < w * channels) return fail(c, SLAMFE_E_INVALID, "stride %d < row bytes", stride);
    int rc = ensure_geometry(c, w, h);
    if (rc) return rc;
    const slamfe_dev_geom& g = c->geom;
    hipStream_t s = c->stream;
    uint64_t seq0 = c->next_seq;
    int slot_base = (int)(seq0 % (uint64_t)c->nslots);
    for (int i = 0; i < n; ++i) {
        uint64_t hdl = seq0 + (uint64_t)i;
        c->slot_handle[(size_t)(hdl % (uint64_t)c->nslots)] = hdl;
        c->slot_count_host[(size_t)(hdl % (uint64_t)c->nslots)] = -1;
        if (handles) handles[i] = hdl;
    }
    c->next_seq += (uint64_t)n;

    // Calls of a few frames are bound by the chain of dependent launches, not by
    // the arithmetic: they take the fused kernels (fewer launches, some redundant
    // work); batches take the throughput kernels.  Results are identical.
    const bool few = is_few(c, n, w, h);
    c->cur_l0.base = c->pyr + g.lv[0].off;
    c->cur_l0.frame_stride = g.frame_bytes;
    if (few && g.pt_x > 0) {
        ProbeScope ps(c, SLAMFE_STAGE_PYRAMID);
        KCHK(c, slamfe_launch_pyramid_fused(s, g, n, d_imgs, stride, frame_stride, channels, c->pyr, c->taps));
    } else {
        // Gray frames whose rows already have level 0's pitch are used in place:
        // the kernels that read level 0 take its address from cur_l0, and the copy
        // into the pyramid buffer (78 MB each way per 256 VGA frames) is skipped.
        const bool in_place = d_imgs && channels == 1 && stride == g.lv[0].pitch &&
                              (uintptr_t)d_imgs % 16 == 0 && frame_stride % 16 == 0;
        if (in_place) {
            c->cur_l0.base = d_imgs;
            c->cur_l0.frame_stride = frame_stride;
        } else if (d_imgs) {
            KCHK(c, slamfe_launch_ingest(s, d_imgs, w, h, stride, frame_stride, channels, n, c->pyr,
                                         g.frame_bytes, g.lv[0].pitch));
        }
        ProbeScope ps(c, SLAMFE_STAGE_PYRAMID);
        KCHK(c, slamfe_launch_pyramid(s, g, n, c->pyr, c->taps, c->cur_l0));
    }
    {
        ProbeScope ps(c, SLAMFE_STAGE_FAST);
        KCHK(c, slamfe_launch_fast_blur(s, g, n, c->pyr, c->blur, c->p.fast_threshold,
                                        c->p.edge_threshold, c->cand, c->cand_cnt, c->cur_l0));
    }
    {
        ProbeScope ps(c, SLAMFE_STAGE_SELECT);
        KCHK(c, slamfe_launch_select(s, g, n, c->cand, c->cand_cnt, c->band_hist, c->band_off,
                                     c->level_kept, c->level_cutoff, slot_base, c->nslots, c->kp_key,
                                     c->kp_level, c->kp_count, c->max_kp, c->sort_lds, few));
    }
    if (few) {
        ProbeScope ps(c, SLAMFE_STAGE_DESCRIBE);
        KCHK(c, slamfe_launch_orient_describe(s, g, n, c->pyr, c->blur, slot_base, c->nslots, c->kp_key,
                                              c->kp_level, c->kp_count, c->kp_mom, c->kps, c->kp_rot, c->desc,
                                              c->max_kp, c->p.mode, c->p.patch_size, c->host_rows, c->cur_l0));
        c->host_rows_done = c->host_rows.cnt != nullptr;
        return SLAMFE_OK;
    }
    {
        ProbeScope ps(c, SLAMFE_STAGE_ORIENT);
        KCHK(c, slamfe_launch_orient(s, g, n, c->pyr, slot_base, c->nslots, c->kp_key, c->kp_level,
                                     c->kp_count, c->kp_mom, c->kps, c->kp_rot, c->max_kp, c->p.mode,
                                     c->p.patch_size, c->cur_l0));
    }
    {
        ProbeScope ps(c, SLAMFE_STAGE_DESCRIBE);
        KCHK(c, slamfe_launch_describe(s, g, n, c->blur, slot_base, c->nslots, c->kp_key, c->kp_level,
                                       c->kp_count, c->kp_rot, c->desc, c->max_kp));
    }
    return SLAMFE_OK;
}

int ensure_host_staging(slamfe_ctx* c)
{
    if (c->h_in) return SLAMFE_OK;
    const size_t B = (size_t)c->p.max_batch;
    HIPCHK(c, hipHostMalloc((void**)&c->h_in, B * (size_t)c->p.max_width * c->p.max_height * 3, 0));
    HIPCHK(c, hipHostMalloc((void**)&c->h_out, B * (size_t)c->max_kp * (sizeof(slamfe_keypoint) + 32), 0));
    return SLAMFE_OK;
}

// Copy the results of `nf` resident frames (slots[i]) to the caller: one
// transfer + wait for the counts, one for the rows, through pinned memory.
int download_frames(slamfe_ctx* c, const int* slots, int nf, slamfe_keypoint* kps, uint8_t* desc,
                    int cap, int* n_out, bool view = false)
{
    int rc = ensure_host_staging(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    for (int i = 0; i < nf; ++i)
        HIPCHK(c, hipMemcpyAsync(c->h_cnt + i, c->kp_count + slots[i], sizeof(int32_t),
                                 hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int worst = SLAMFE_OK;
    const size_t K = (size_t)c->max_kp;
    slamfe_keypoint* hk = reinterpret_cast<slamfe_keypoint*>(c->h_out);
    uint8_t* hd = c->h_out + (size_t)c->p.max_batch * K * sizeof(slamfe_keypoint);
    std::vector<int> ncopy((size_t)nf, 0);
    for (int i = 0; i < nf; ++i) {
        int n = c->h_cnt[i];
        c->slot_count_host[(size_t)slots[i]] = n;
        if (n_out) n_out[i] = n;
        if (n > c->max_kp) {
            worst = fail(c, SLAMFE_E_CAPACITY, "%d keypoints exceed the context's max_keypoints %d", n,
                         c->max_kp);
            continue;
        }
        if (!view && (kps || desc) && n > cap) {
            worst = fail(c, SLAMFE_E_CAPACITY, "%d keypoints exceed the caller's capacity %d", n, cap);
            continue;
        }
        ncopy[(size_t)i] = n;
    }
    // One strided transfer per run of consecutive slots (a batch occupies
    // consecutive slots of the ring, so at most two runs) instead of two small
    // transfers per frame: every frame of the run sends its first `most` rows.
    for (int i0 = 0; i0 < nf;) {
        int i1 = i0 + 1;
        while (i1 < nf && slots[i1] == slots[i1 - 1] + 1) ++i1;
        int most = 0;
        for (int i = i0; i < i1; ++i) most = std::max(most, ncopy[(size_t)i]);
        if (most > 0 && (kps || view))
            HIPCHK(c, hipMemcpy2DAsync(hk + (size_t)i0 * K, K * sizeof(slamfe_keypoint),
                                       c->kps + (size_t)slots[i0] * K, K * sizeof(slamfe_keypoint),
                                       (size_t)most * sizeof(slamfe_keypoint), (size_t)(i1 - i0),
                                       hipMemcpyDeviceToHost, s));
        if (most > 0 && (desc || view))
            HIPCHK(c, hipMemcpy2DAsync(hd + (size_t)i0 * K * 32, K * 32, c->desc + (size_t)slots[i0] * K * 32,
                                       K * 32, (size_t)most * 32, (size_t)(i1 - i0), hipMemcpyDeviceToHost, s));
        i0 = i1;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    for (int i = 0; i < nf && !view; ++i) {
        size_t n = (size_t)ncopy[(size_t)i];
        if (n && kps) memcpy(kps + (size_t)i * cap, hk + (size_t)i * K, n * sizeof(slamfe_keypoint));
        if (n && desc) memcpy(desc + (size_t)i * cap * 32, hd + (size_t)i * K * 32, n * 32);
    }
    return worst;
}

// The results of the nf frames just extracted (consecutive slots of the ring) for
// a call of a few frames: one kernel writes counts and rows into the pinned
// staging arrays, one wait, then the copies into the caller's arrays.
int export_frames(slamfe_ctx* c, const int* slots, int nf, slamfe_keypoint* kps, uint8_t* desc, int cap,
                  int* n_out, bool view = false)
{
    int rc = ensure_host_staging(c);
    if (rc) return rc;
    const size_t K = (size_t)c->max_kp;
    uint8_t* hk = c->h_out;
    uint8_t* hd = c->h_out + (size_t)c->p.max_batch * K * sizeof(slamfe_keypoint);
    // (the extraction's last kernel has written them already if it was told to)
    if (!c->host_rows_done)
        KCHK(c, slamfe_launch_export(c->stream, c->kp_count, c->kps, c->desc, slots[0], c->nslots, c->max_kp, nf,
                                     (kps || desc || view) ? 1 : 0, c->h_cnt, hk, hd));
    c->host_rows_done = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int worst = SLAMFE_OK;
    for (int i = 0; i < nf; ++i) {
        const int n = c->h_cnt[i];
        c->slot_count_host[(size_t)slots[i]] = n;
        if (n_out) n_out[i] = n;
        if (n > c->max_kp) {
            worst = fail(c, SLAMFE_E_CAPACITY, "%d keypoints exceed the context's max_keypoints %d", n,
                         c->max_kp);
            continue;
        }
        if (view) continue; // the rows stay where they are: the caller reads them in place
        if ((kps || desc) && n > cap) {
            worst = fail(c, SLAMFE_E_CAPACITY, "%d keypoints exceed the caller's capacity %d", n, cap);
            continue;
        }
        if (n && kps)
            memcpy(kps + (size_t)i * cap, hk + (size_t)i * K * sizeof(slamfe_keypoint), (size_t)n * sizeof(slamfe_keypoint));
        if (n && desc) memcpy(desc + (size_t)i * cap * 32, hd + (size_t)i * K * 32, (size_t)n * 32);
    }
    return worst;
}

int download_frame(slamfe_ctx* c, int slot, slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    return download_frames(c, &slot, 1, kps, desc, cap, n_out);
}

} // namespace

extern "C" {

int slamfe_version(void) { return 100; }

int slamfe_default_params(slamfe_params* p)
{
    if (!p) return SLAMFE_E_INVALID;
    memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->mode = SLAMFE_MODE_ORB;
    p->fast_threshold = 30; // main.cc:11
    p->nfeatures = 10000;   // main.cc:11
    p->scale_factor = 1.2f; // Init.cc:65
    p->nlevels = 8;
    p->edge_threshold = 31;
    p->patch_size = 31;
    p->ratio = 0.75f; // Init.cc:146
    p->device_id = 0;
    p->max_width = 1920;
    p->max_height = 1080;
    p->max_batch = 1;
    p->max_keypoints = 0; // chosen from mode / nfeatures
    p->frame_slots = 0;   // chosen from max_batch
    p->map_capacity = 1 << 16;
    p->stream = nullptr;
    p->match_kernel = SLAMFE_MATCH_AUTO;
    p->reserved0 = 0;
    return SLAMFE_OK;
}

void slamfe_destroy(slamfe_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->p.device_id);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* bufs[] = {c->pyr, c->blur, c->in_stage, c->taps, c->cand, c->cand_cnt, c->band_hist,
                    c->band_off, c->level_kept, c->level_cutoff, c->kp_key, c->kp_level,
                    c->kp_count, c->kps, c->desc, c->kp_mom, c->kp_rot, c->match_out, c->map_desc, c->match_part.p,
                    c->match_q.p, c->match_t.p, c->match_o.p, c->pj_in.p, c->pj_cand.p, c->pj_candn.p, c->pj_cells.p,
                    c->pj_sorted.p, c->pj_cnt.p, c->pj_rank.p, c->ing_taps.p,
                    c->ing_bgr.p, c->map_xyz, c->rp_uv_all.p, c->rp_mask.p, c->rp_off.p, c->rp_total.p,
                    c->vis_uv.p, c->vis_idx.p, c->tri_in.p, c->tri_out.p};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (hipEvent_t e : c->probe_ev) (void)hipEventDestroy(e);
    if (c->ev_count) (void)hipEventDestroy(c->ev_count);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_cnt) (void)hipHostFree(c->h_cnt);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* slamfe_last_error(const slamfe_ctx* c) { return c ? c->err.c_str() : "null context"; }

int slamfe_create(const slamfe_params* pin, slamfe_ctx** out)
{
    if (!pin || !out) return SLAMFE_E_INVALID;
    *out = nullptr;
    if (pin->struct_size != (int32_t)sizeof(slamfe_params)) return SLAMFE_E_INVALID;
    slamfe_params p = *pin;
    if (p.mode != SLAMFE_MODE_ORB && p.mode != SLAMFE_MODE_PROVIDED) return SLAMFE_E_INVALID;
    if (p.nlevels < 1 || p.nlevels > SLAMFE_MAX_LEVELS) return SLAMFE_E_INVALID;
    if (p.patch_size != 31 || p.edge_threshold < 22) return SLAMFE_E_INVALID;
    if (!(p.scale_factor > 1.0f) || p.scale_factor > 2.0f) return SLAMFE_E_INVALID;
    if (p.max_width < 8 || p.max_height < 8 || p.max_width > SLAMFE_MAX_DIM ||
        p.max_height > SLAMFE_MAX_DIM)
        return SLAMFE_E_INVALID;
    if (p.max_batch < 1 || p.nfeatures < 0 || p.map_capacity < 0) return SLAMFE_E_INVALID;
    if (p.fast_threshold < 0) p.fast_threshold = 0;
    if (p.fast_threshold > 255) p.fast_threshold = 255;
    if (p.max_keypoints <= 0)
        p.max_keypoints = p.mode == SLAMFE_MODE_ORB ? (2 * p.nfeatures > 4096 ? 2 * p.nfeatures : 4096)
                                                    : 32768;
    if (p.max_keypoints >= SLAMFE_MAX_TRAIN) return SLAMFE_E_INVALID;
    if (p.frame_slots <= 0) p.frame_slots = 2 * p.max_batch > 4 ? 2 * p.max_batch : 4;
    if (p.frame_slots < p.max_batch) return SLAMFE_E_INVALID;
    if (p.match_kernel < SLAMFE_MATCH_AUTO || p.match_kernel > SLAMFE_MATCH_MATRIX) return SLAMFE_E_INVALID;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p.device_id < 0 ||
        p.device_id >= ndev)
        return SLAMFE_E_DEVICE;
    if (hipSetDevice(p.device_id) != hipSuccess) return SLAMFE_E_DEVICE;

    slamfe_ctx* c = new slamfe_ctx();
    c->p = p;
    c->max_kp = p.max_keypoints;
    c->nslots = p.frame_slots;
    c->slot_handle.assign((size_t)c->nslots, 0);
    c->slot_count_host.assign((size_t)c->nslots, -1);
    if (const char* e = getenv("SLAMFE_FEW_PIXELS")) c->few_pixels = atoll(e); // tuning aid
    *out = c; // returned even on failure so that slamfe_last_error can be read
    if (p.stream) {
        c->stream = (hipStream_t)p.stream;
    } else {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    {
        int rc = static_scratch_for(c, p.device_id, &c->static_scratch);
        if (rc) return rc;
    }
    slamfe_dev_geom gmax;
    std::vector<uint32_t> tmax;
    build_geometry(p, p.max_width, p.max_height, &gmax, &tmax);
    c->cap_frame_bytes = (size_t)gmax.frame_bytes + 4096;
    c->cap_tiles = gmax.total_tiles + 2 * SLAMFE_MAX_LEVELS + 64;
    c->cap_bands = gmax.total_bands + SLAMFE_MAX_LEVELS;
    // (the range tables of the one-launch pyramid are part of this buffer: room
    // for the largest number of tiles whether or not the maximum size uses them)
    c->cap_taps = tmax.size() + 64 +
                  (size_t)SLAMFE_MAX_LEVELS * 2 *
                      ((p.max_width + SLAMFE_PT_W - 1) / SLAMFE_PT_W + (p.max_height + SLAMFE_PT_H - 1) / SLAMFE_PT_H + 2);
    KCHK(c, slamfe_kernels_init(sort_lds_bytes_for(gmax)));

    const size_t B = (size_t)p.max_batch;
    HIPCHK(c, hipMalloc(&c->pyr, B * c->cap_frame_bytes));
    HIPCHK(c, hipMalloc(&c->blur, B * c->cap_frame_bytes));
    HIPCHK(c, hipMalloc(&c->in_stage, B * (size_t)p.max_width * p.max_height * 3));
    HIPCHK(c, hipMalloc(&c->taps, c->cap_taps * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->cand, B * c->cap_tiles * SLAMFE_TILE_CAND * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->cand_cnt, B * c->cap_tiles * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->band_hist, B * c->cap_bands * 256 * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->band_off, B * c->cap_bands * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->level_kept, B * SLAMFE_MAX_LEVELS * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->level_cutoff, B * SLAMFE_MAX_LEVELS * sizeof(int32_t)));
    const size_t S = (size_t)c->nslots, K = (size_t)c->max_kp;
    HIPCHK(c, hipMalloc(&c->kp_key, S * K * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->kp_level, S * K * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->kp_count, S * sizeof(int32_t)));
    HIPCHK(c, hipMalloc(&c->kp_mom, S * K * 2 * sizeof(int32_t)));
    HIPCHK(c, hipMalloc(&c->kp_rot, S * K * 2 * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->kps, S * K * sizeof(slamfe_keypoint)));
    HIPCHK(c, hipMalloc(&c->desc, S * K * 32));
    HIPCHK(c, hipMemset(c->kp_count, 0, S * sizeof(int32_t)));
    HIPCHK(c, hipHostMalloc((void**)&c->h_cnt, (B > 16 ? B : 16) * sizeof(int32_t), 0));
    HIPCHK(c, hipMalloc(&c->match_out, K * sizeof(slamfe_match2)));
    slamfe_match_plan(c->max_kp, c->max_kp, 1, c->p.match_kernel, &c->fm_qpt, &c->fm_chunk, &c->fm_nchunks);
    int rc = grow(c, c->match_part, (size_t)c->fm_nchunks * K * 8);
    if (rc) return rc;
    if (p.map_capacity > 0) {
        HIPCHK(c, hipMalloc(&c->map_desc, (size_t)p.map_capacity * 32));
        HIPCHK(c, hipMemset(c->map_desc, 0, (size_t)p.map_capacity * 32));
        HIPCHK(c, hipMalloc(&c->map_xyz, (size_t)p.map_capacity * 12));
        HIPCHK(c, hipMemset(c->map_xyz, 0, (size_t)p.map_capacity * 12));
    }
    HIPCHK(c, hipDeviceSynchronize());
    return SLAMFE_OK;
}

int slamfe_sync(slamfe_ctx* c)
{
    if (!c) return SLAMFE_E_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SLAMFE_OK;
}

int slamfe_extract_resident(slamfe_ctx* c, const void* d_imgs, int n, int w, int h, int stride,
                            size_t frame_stride_bytes, int channels, uint64_t* frame_handles)
{
    if (!c || !d_imgs) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    return run_extract(c, (const uint8_t*)d_imgs, n, w, h, stride, frame_stride_bytes, channels,
                       frame_handles);
}

namespace {
int extract_host_frames(slamfe_ctx* c, const uint8_t* const* imgs, int n, int w, int h, int stride,
                        int channels, slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                        uint64_t* frame_handles, bool view)
{
    if (!c || !imgs) return SLAMFE_E_INVALID;
    PhaseTrace tr("extract_batch");
    HIPCHK(c, hipSetDevice(c->p.device_id));
    if (n < 1 || n > c->p.max_batch)
        return fail(c, SLAMFE_E_INVALID, "batch of %d frames; context allows 1..%d", n,
                    c->p.max_batch);
    if (w < 1 || h < 1 || w > c->p.max_width || h > c->p.max_height || (channels != 1 && channels != 3) ||
        stride < w * channels)
        return fail(c, SLAMFE_E_INVALID, "bad frame description %dx%d stride %d channels %d", w, h,
                    stride, channels);
    const size_t row = (size_t)w * channels, fbytes = row * h;
    int rc = ensure_host_staging(c);
    if (rc) return rc;
    // the staging buffers may still feed the previous call's queued copies
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tr.mark("setup");
    for (int i = 0; i < n; ++i)
        if (!imgs[i]) return fail(c, SLAMFE_E_INVALID, "imgs[%d] is NULL", i);
    // frames go to the pinned buffer in groups of about 2 MiB; each group's
    // transfer to the device is queued as soon as it is staged, so the DMA of one
    // group runs while the host copies the next
    const int group = (int)std::max<size_t>(1, ((size_t)2 << 20) / fbytes);
    for (int i0 = 0; i0 < n; i0 += group) {
        const int i1 = std::min(n, i0 + group);
        for (int i = i0; i < i1; ++i) {
            uint8_t* dst = c->h_in + (size_t)i * fbytes;
            if ((size_t)stride == row) memcpy(dst, imgs[i], fbytes);
            else
                for (int y = 0; y < h; ++y) memcpy(dst + (size_t)y * row, imgs[i] + (size_t)y * stride, row);
        }
        HIPCHK(c, hipMemcpyAsync(c->in_stage + (size_t)i0 * fbytes, c->h_in + (size_t)i0 * fbytes,
                                 (size_t)(i1 - i0) * fbytes, hipMemcpyHostToDevice, c->stream));
    }
    tr.mark("stage+h2d");
    std::vector<uint64_t> hd((size_t)n);
    // results: counts and rows are written into pinned memory by a kernel (one wait, no
    // round trip for the counts first), by the extraction's last kernel itself in calls
    // of a few frames
    const bool by_export = c->max_kp % 4 == 0;
    c->host_rows_done = false;
    if (by_export) {
        const size_t K = (size_t)c->max_kp;
        c->host_rows.cnt = c->h_cnt;
        const bool rows = kps || desc || view;
        c->host_rows.kps = rows ? c->h_out : nullptr;
        c->host_rows.desc = rows ? c->h_out + (size_t)c->p.max_batch * K * sizeof(slamfe_keypoint) : nullptr;
    }
    rc = run_extract(c, c->in_stage, n, w, h, (int)row, fbytes, channels, hd.data());
    c->host_rows = {nullptr, nullptr, nullptr};
    if (rc) return rc;
    tr.mark("launches");
    std::vector<int> slots((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (frame_handles) frame_handles[i] = hd[(size_t)i];
        slots[(size_t)i] = slot_of(c, hd[(size_t)i]);
    }
    rc = by_export ? export_frames(c, slots.data(), n, kps, desc, cap, n_out, view)
                   : download_frames(c, slots.data(), n, kps, desc, cap, n_out, view);
    tr.mark("results");
    return rc;
}
} // namespace

int slamfe_extract_batch(slamfe_ctx* c, const uint8_t* const* imgs, int n, int w, int h, int stride,
                         int channels, slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                         uint64_t* frame_handles)
{
    return extract_host_frames(c, imgs, n, w, h, stride, channels, kps, desc, cap, n_out, frame_handles, false);
}

int slamfe_extract_view(slamfe_ctx* c, const uint8_t* img, int w, int h, int stride, int channels,
                        const slamfe_keypoint** kps, const uint8_t** desc, int* n_out, uint64_t* frame_handle)
{
    if (!c || !img || !kps || !desc || !n_out) return SLAMFE_E_INVALID;
    *kps = nullptr;
    *desc = nullptr;
    *n_out = 0;
    const uint8_t* one[1] = {img};
    int n = 0;
    int rc = extract_host_frames(c, one, 1, w, h, stride, channels, nullptr, nullptr, 0, &n, frame_handle, true);
    *n_out = n;
    if (rc) return rc;
    // frame 0 of the pinned result arrays: records first, descriptor rows after
    // the records of max_batch frames
    *kps = reinterpret_cast<const slamfe_keypoint*>(c->h_out);
    *desc = c->h_out + (size_t)c->p.max_batch * (size_t)c->max_kp * sizeof(slamfe_keypoint);
    return SLAMFE_OK;
}

int slamfe_extract(slamfe_ctx* c, const uint8_t* img, int w, int h, int stride, int channels,
                   slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out, uint64_t* frame_handle)
{
    const uint8_t* one[1] = {img};
    return slamfe_extract_batch(c, one, 1, w, h, stride, channels, kps, desc, cap, n_out,
                                frame_handle);
}

namespace {
short sat_short(float v)
{
    int r = slamfe_round_f(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}
} // namespace

int slamfe_extract_scaled_bgr(slamfe_ctx* c, const uint8_t* bgr, int w, int h, int stride, double fx,
                              double fy, uint8_t* resized_bgr, int* out_w, int* out_h,
                              slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                              uint64_t* frame_handle)
{
    if (!c || !bgr) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    if (w < 2 || h < 2 || stride < w * 3 || !(fx > 0.5 && fx <= 1.0) || !(fy > 0.5 && fy <= 1.0))
        return fail(c, SLAMFE_E_INVALID, "scaled ingest takes a BGR frame and factors in (0.5, 1]");
    if (w > c->p.max_width || h > c->p.max_height)
        return fail(c, SLAMFE_E_INVALID, "frame %dx%d outside the context's maximum %dx%d", w, h,
                    c->p.max_width, c->p.max_height);
    const int dw = slamfe_round_d((double)w * fx), dh = slamfe_round_d((double)h * fy);
    if (out_w) *out_w = dw;
    if (out_h) *out_h = dh;
    int rc = ensure_geometry(c, dw, dh);
    if (rc) return rc;
    if ((rc = ensure_host_staging(c))) return rc;
    // taps exactly as cv::resize computes them (float positions, 11-bit weights)
    std::vector<uint32_t> taps(2 * (size_t)(dw + dh));
    const double scale_x = 1. / fx, scale_y = 1. / fy;
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float f = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(f);
        f -= (float)sx;
        if (sx < 0) { f = 0; sx = 0; }
        if (sx + 1 >= w) {
            if (dx < xmax) xmax = dx;
            if (sx >= w - 1) { f = 0; sx = w - 1; }
        }
        uint32_t a0 = (uint16_t)sat_short((1.f - f) * 2048.f), a1 = (uint16_t)sat_short(f * 2048.f);
        taps[2 * (size_t)dx] = (uint32_t)sx | (dx >= xmax ? 0x80000000u : 0u);
        taps[2 * (size_t)dx + 1] = a0 | (a1 << 16);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float f = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(f);
        f -= (float)sy;
        int y0 = sy < 0 ? 0 : (sy < h ? sy : h - 1), y1 = sy + 1 < 0 ? 0 : (sy + 1 < h ? sy + 1 : h - 1);
        uint32_t b0 = (uint16_t)sat_short((1.f - f) * 2048.f), b1 = (uint16_t)sat_short(f * 2048.f);
        taps[2 * (size_t)(dw + dy)] = (uint32_t)y0 | ((uint32_t)y1 << 16);
        taps[2 * (size_t)(dw + dy) + 1] = b0 | (b1 << 16);
    }
    const size_t tap_bytes = taps.size() * sizeof(uint32_t);
    const int bpitch = (dw * 3 + 15) / 16 * 16;
    if ((rc = grow(c, c->ing_taps, tap_bytes))) return rc;
    if (resized_bgr && (rc = grow(c, c->ing_bgr, (size_t)bpitch * dh))) return rc;
    const size_t row = (size_t)w * 3, fbytes = row * h;
    if ((rc = pin_reserve(c, tap_bytes + (resized_bgr ? (size_t)bpitch * dh : 0)))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int y = 0; y < h; ++y) memcpy(c->h_in + (size_t)y * row, bgr + (size_t)y * stride, row);
    memcpy(c->h_pin, taps.data(), tap_bytes);
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->in_stage, c->h_in, fbytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->ing_taps.p, c->h_pin, tap_bytes, hipMemcpyHostToDevice, s));
    const uint32_t* dt = (const uint32_t*)c->ing_taps.p;
    KCHK(c, slamfe_launch_ingest_scaled(s, c->in_stage, (int)row, dt, dt + 2 * (size_t)dw, dw, dh, c->pyr,
                                        c->geom.lv[0].pitch, resized_bgr ? (uint8_t*)c->ing_bgr.p : nullptr,
                                        bpitch));
    uint64_t hd = 0;
    rc = run_extract(c, nullptr, 1, dw, dh, dw, 0, 1, &hd);
    if (rc) return rc;
    if (frame_handle) *frame_handle = hd;
    if (resized_bgr)
        HIPCHK(c, hipMemcpyAsync(c->h_pin + tap_bytes, c->ing_bgr.p, (size_t)bpitch * dh,
                                 hipMemcpyDeviceToHost, s));
    int slot = slot_of(c, hd);
    rc = c->max_kp % 4 == 0 ? export_frames(c, &slot, 1, kps, desc, cap, n_out)
                            : download_frames(c, &slot, 1, kps, desc, cap, n_out);
    if (resized_bgr)
        for (int y = 0; y < dh; ++y)
            memcpy(resized_bgr + (size_t)y * dw * 3, c->h_pin + tap_bytes + (size_t)y * bpitch, (size_t)dw * 3);
    return rc;
}

int slamfe_frame_download(slamfe_ctx* c, uint64_t frame_handle, slamfe_keypoint* kps, uint8_t* desc,
                          int cap, int* n_out)
{
    if (!c) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int slot = slot_of(c, frame_handle);
    if (slot < 0) return fail(c, SLAMFE_E_HANDLE, "frame handle %llu expired or unknown",
                              (unsigned long long)frame_handle);
    return download_frame(c, slot, kps, desc, cap, n_out);
}

int slamfe_frame_device_ptrs(slamfe_ctx* c, uint64_t frame_handle, const slamfe_keypoint** d_kps,
                             const uint8_t** d_desc, const int32_t** d_count)
{
    if (!c) return SLAMFE_E_INVALID;
    int slot = slot_of(c, frame_handle);
    if (slot < 0) return fail(c, SLAMFE_E_HANDLE, "frame handle %llu expired or unknown",
                              (unsigned long long)frame_handle);
    if (d_kps) *d_kps = c->kps + (size_t)slot * c->max_kp;
    if (d_desc) *d_desc = c->desc + (size_t)slot * c->max_kp * 32;
    if (d_count) *d_count = c->kp_count + slot;
    return SLAMFE_OK;
}

int slamfe_match_knn2(slamfe_ctx* c, const uint8_t* query, int nq, const uint8_t* train, int nt,
                      slamfe_match2* out)
{
    if (!c || nq < 0 || nt < 0 || (nq > 0 && (!query || !out)) || (nt > 0 && !train))
        return SLAMFE_E_INVALID;
    if (nt >= SLAMFE_MAX_TRAIN)
        return fail(c, SLAMFE_E_INVALID, "train set of %d rows; at most %d per call", nt,
                    SLAMFE_MAX_TRAIN - 1);
    if (nq == 0) return SLAMFE_OK;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int rc;
    if ((rc = grow(c, c->match_q, (size_t)nq * 32))) return rc;
    if ((rc = grow(c, c->match_t, (size_t)(nt > 0 ? nt : 1) * 32))) return rc;
    if ((rc = grow(c, c->match_o, (size_t)nq * sizeof(slamfe_match2)))) return rc;
    int qpt, chunk, nchunks;
    slamfe_match_plan(nq, nt, 1, c->p.match_kernel, &qpt, &chunk, &nchunks);
    if ((rc = grow(c, c->match_part, (size_t)nchunks * nq * 8))) return rc;
    const size_t oq = 0, ot = align256((size_t)nq * 32), oo = ot + align256((size_t)nt * 32);
    if ((rc = pin_reserve(c, oo + (size_t)nq * sizeof(slamfe_match2)))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the bounce buffer may still be in use
    memcpy(c->h_pin + oq, query, (size_t)nq * 32);
    HIPCHK(c, hipMemcpyAsync(c->match_q.p, c->h_pin + oq, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (nt > 0) {
        memcpy(c->h_pin + ot, train, (size_t)nt * 32);
        HIPCHK(c, hipMemcpyAsync(c->match_t.p, c->h_pin + ot, (size_t)nt * 32, hipMemcpyHostToDevice,
                                 c->stream));
    }
    KCHK(c, slamfe_launch_match(c->stream, (const uint8_t*)c->match_q.p, nq, nullptr,
                                (const uint8_t*)c->match_t.p, nt, nullptr, nq, qpt, chunk, nchunks,
                                (uint32_t*)c->match_part.p, c->p.ratio, (slamfe_match2*)c->match_o.p,
                                nullptr, 0, 1, 0, 0, 0));
    HIPCHK(c, hipMemcpyAsync(c->h_pin + oo, c->match_o.p, (size_t)nq * sizeof(slamfe_match2),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->h_pin + oo, (size_t)nq * sizeof(slamfe_match2));
    return SLAMFE_OK;
}

int slamfe_match_resident(slamfe_ctx* c, uint64_t query_frame, uint64_t train_frame,
                          slamfe_match2* d_out)
{
    if (!c || !d_out) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int sq = slot_of(c, query_frame), st = slot_of(c, train_frame);
    if (sq < 0 || st < 0) return fail(c, SLAMFE_E_HANDLE, "frame handle expired or unknown");
    KCHK(c, slamfe_launch_match(c->stream, c->desc + (size_t)sq * c->max_kp * 32, 0, c->kp_count + sq,
                                c->desc + (size_t)st * c->max_kp * 32, 0, c->kp_count + st, c->max_kp,
                                c->fm_qpt, c->fm_chunk, c->fm_nchunks, (uint32_t*)c->match_part.p,
                                c->p.ratio, d_out, nullptr, 0, 1, 0, 0, 0));
    return SLAMFE_OK;
}

int slamfe_match_resident_batch(slamfe_ctx* c, uint64_t first_query, uint64_t first_train, int npairs,
                                slamfe_match2* d_out)
{
    if (!c || !d_out || npairs < 1 || npairs > c->nslots) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    for (int i = 0; i < npairs; ++i)
        if (slot_of(c, first_query + (uint64_t)i) < 0 || slot_of(c, first_train + (uint64_t)i) < 0)
            return fail(c, SLAMFE_E_HANDLE, "frame handle expired or unknown (pair %d)", i);
    int qpt, chunk, nchunks;
    slamfe_match_plan(c->max_kp, c->max_kp, npairs, c->p.match_kernel, &qpt, &chunk, &nchunks);
    int rc = grow(c, c->match_part, (size_t)npairs * nchunks * c->max_kp * 8);
    if (rc) return rc;
    ProbeScope ps(c, SLAMFE_STAGE_MATCH);
    KCHK(c, slamfe_launch_match(c->stream, c->desc, 0, c->kp_count, c->desc, 0, c->kp_count, c->max_kp,
                                qpt, chunk, nchunks, (uint32_t*)c->match_part.p,
                                c->p.ratio, d_out, nullptr, 0, npairs,
                                (int)(first_query % (uint64_t)c->nslots),
                                (int)(first_train % (uint64_t)c->nslots), c->nslots));
    return SLAMFE_OK;
}

namespace {
// Init.cc:131-153 on two resident frames with the rows brought to the host: copied
// into out[0..cap), or (view != nullptr) left in the pinned bounce buffer for the
// caller to read in place.
int match_frames_to_host(slamfe_ctx* c, uint64_t query_frame, uint64_t train_frame, slamfe_match2* out,
                         int cap, int* nq_out, const slamfe_match2** view)
{
    if (!c) return SLAMFE_E_INVALID;
    if (view) *view = nullptr;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    const int sq = slot_of(c, query_frame);
    if (sq < 0 || slot_of(c, train_frame) < 0) return fail(c, SLAMFE_E_HANDLE, "frame handle expired or unknown");
    const bool rows = out || view;
    // The query frame's keypoint count is known on the host if the frame came
    // through a host-pointer call.  Then the rows are written by the merge kernel
    // straight into pinned host memory (which the device can address): one launch
    // sequence, one wait.
    int32_t nq = c->slot_count_host[(size_t)sq];
    if (nq >= 0) {
        if (nq > c->max_kp) nq = c->max_kp;
        if (nq_out) *nq_out = nq;
        if (!rows) return SLAMFE_OK; // only the count was asked for: nothing to launch
        if (!view && nq > cap)
            return fail(c, SLAMFE_E_CAPACITY, "%d match rows exceed the caller's capacity %d", nq, cap);
        int rc = pin_reserve(c, (size_t)c->max_kp * sizeof *out);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream)); // the bounce buffer may still be in use
        if ((rc = slamfe_match_resident(c, query_frame, train_frame, reinterpret_cast<slamfe_match2*>(c->h_pin))))
            return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (view) *view = reinterpret_cast<const slamfe_match2*>(c->h_pin);
        else if (nq > 0) memcpy(out, c->h_pin, (size_t)nq * sizeof *out);
        return SLAMFE_OK;
    }
    // count not known yet: read it back first (the rows are matched only if wanted)
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->kp_count + sq, sizeof(int32_t), hipMemcpyDeviceToHost,
                             c->stream));
    if (rows) {
        int rc = slamfe_match_resident(c, query_frame, train_frame, c->match_out);
        if (rc) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    nq = c->h_cnt[0];
    c->slot_count_host[(size_t)sq] = nq;
    if (nq > c->max_kp) nq = c->max_kp;
    if (nq_out) *nq_out = nq;
    if (rows) {
        if (!view && nq > cap)
            return fail(c, SLAMFE_E_CAPACITY, "%d match rows exceed the caller's capacity %d", nq, cap);
        int rc = pin_reserve(c, (size_t)(nq > 0 ? nq : 1) * sizeof *out);
        if (rc) return rc;
        if (nq > 0) {
            HIPCHK(c, hipMemcpyAsync(c->h_pin, c->match_out, (size_t)nq * sizeof *out,
                                     hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (view) *view = reinterpret_cast<const slamfe_match2*>(c->h_pin);
        else if (nq > 0) memcpy(out, c->h_pin, (size_t)nq * sizeof *out);
    }
    return SLAMFE_OK;
}
} // namespace

int slamfe_match_frames(slamfe_ctx* c, uint64_t query_frame, uint64_t train_frame, slamfe_match2* out,
                        int cap, int* nq_out)
{
    return match_frames_to_host(c, query_frame, train_frame, out, cap, nq_out, nullptr);
}

int slamfe_match_frames_view(slamfe_ctx* c, uint64_t query_frame, uint64_t train_frame,
                             const slamfe_match2** rows, int* nq_out)
{
    if (!rows || !nq_out) return SLAMFE_E_INVALID;
    return match_frames_to_host(c, query_frame, train_frame, nullptr, 0, nq_out, rows);
}

int slamfe_match_shard_keys(slamfe_ctx* c, const void* d_query, int nq, const void* d_train, int nt,
                            int64_t train_base, uint64_t* d_keys)
{
    if (!c || nq < 0 || nt < 0 || !d_keys || (nq > 0 && !d_query) || (nt > 0 && !d_train))
        return SLAMFE_E_INVALID;
    if (nt >= SLAMFE_MAX_TRAIN)
        return fail(c, SLAMFE_E_INVALID, "shard of %d rows; at most %d per call", nt,
                    SLAMFE_MAX_TRAIN - 1);
    if (nq == 0) return SLAMFE_OK;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int qpt, chunk, nchunks, rc;
    slamfe_match_plan(nq, nt, 1, c->p.match_kernel, &qpt, &chunk, &nchunks);
    if ((rc = grow(c, c->match_part, (size_t)nchunks * nq * 8))) return rc;
    KCHK(c, slamfe_launch_match(c->stream, (const uint8_t*)d_query, nq, nullptr,
                                (const uint8_t*)d_train, nt, nullptr, nq, qpt, chunk, nchunks,
                                (uint32_t*)c->match_part.p, c->p.ratio, nullptr, d_keys, train_base, 1,
                                0, 0, 0));
    return SLAMFE_OK;
}

int slamfe_match_shard_keys_static(slamfe_ctx* c, const void* d_query, int nq, const void* d_train, int nt,
                                   int64_t train_base, uint64_t* d_keys)
{
    if (!c || nq < 0 || nt < 0 || !d_keys || (nq > 0 && !d_query) || (nt > 0 && !d_train))
        return SLAMFE_E_INVALID;
    if (nt >= SLAMFE_MAX_TRAIN)
        return fail(c, SLAMFE_E_INVALID, "shard of %d rows; at most %d per call", nt,
                    SLAMFE_MAX_TRAIN - 1);
    if (nq == 0) return SLAMFE_OK;
    if (!c->static_scratch) return fail(c, SLAMFE_E_DEVICE, "the context has no static scratch address");
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int qpt, chunk, nchunks;
    slamfe_match_plan(nq, nt, 1, c->p.match_kernel, &qpt, &chunk, &nchunks);
    const size_t need = (size_t)nchunks * (size_t)nq * 8;
    if (need > SLAMFE_STATIC_SCRATCH_BYTES)
        return fail(c, SLAMFE_E_CAPACITY,
                    "%d queries x %d train rows need %zu bytes of partial results; the static scratch "
                    "holds %zu (use slamfe_match_shard_keys, or fewer queries / rows per call)",
                    nq, nt, need, (size_t)SLAMFE_STATIC_SCRATCH_BYTES);
    // every partial the merge reads is written by the match kernel of this call:
    // nothing in the array has to start at zero
    KCHK(c, slamfe_launch_match(c->stream, (const uint8_t*)d_query, nq, nullptr,
                                (const uint8_t*)d_train, nt, nullptr, nq, qpt, chunk, nchunks,
                                (uint32_t*)c->static_scratch, c->p.ratio, nullptr, d_keys, train_base, 1,
                                0, 0, 0));
    return SLAMFE_OK;
}

int slamfe_merge_shard_keys(slamfe_ctx* c, const uint64_t* d_keys_all, int nshards, int nq,
                            slamfe_match2* d_out)
{
    if (!c || !d_keys_all || !d_out || nshards < 1 || nq < 0) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    KCHK(c, slamfe_launch_merge_shard_keys(c->stream, d_keys_all, nshards, nq, c->p.ratio, d_out));
    return SLAMFE_OK;
}

int slamfe_map_put(slamfe_ctx* c, const int32_t* point_ids, const uint8_t* desc32, int n)
{
    if (!c || n < 0 || (n > 0 && (!point_ids || !desc32))) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    for (int i = 0; i < n; ++i)
        if (point_ids[i] < 0 || point_ids[i] >= c->p.map_capacity)
            return fail(c, SLAMFE_E_CAPACITY, "point id %d outside map_capacity %d", point_ids[i],
                        c->p.map_capacity);
    if (n == 0) return SLAMFE_OK;
    int rc = pin_reserve(c, (size_t)n * 32);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(c->h_pin, desc32, (size_t)n * 32);
    // ids are dense and usually consecutive (Init.cc:296 uses m.getPointsSize()):
    // copy runs of consecutive ids in one transfer each
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && point_ids[j] == point_ids[j - 1] + 1) ++j;
        HIPCHK(c, hipMemcpyAsync(c->map_desc + (size_t)point_ids[i] * 32, c->h_pin + (size_t)i * 32,
                                 (size_t)(j - i) * 32, hipMemcpyHostToDevice, c->stream));
        i = j;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SLAMFE_OK;
}

namespace {

// Buffers of the grid over R projected points.
int grow_projection_grid(slamfe_ctx* c, int R)
{
    int rc;
    if ((rc = grow(c, c->pj_cells, slamfe_projection_grid_bytes()))) return rc;
    if ((rc = grow(c, c->pj_sorted, (size_t)R * 16))) return rc;
    if (R > SLAMFE_PROJ_ONE_WG_POINTS) {
        if ((rc = grow(c, c->pj_cnt, slamfe_projection_grid_bytes()))) return rc;
        if ((rc = grow(c, c->pj_rank, (size_t)R * 4))) return rc;
    }
    return SLAMFE_OK;
}

// The search of Init.cc:229-259 over a visible set that is either uploaded from
// the host (uv, point_ids) or already on the device (d_uv, d_ids).
int search_projected(slamfe_ctx* c, uint64_t frame, const int32_t* kp_idx, int n_inl, const float* uv,
                     const int32_t* point_ids, const float* d_uv, const int32_t* d_ids, int R,
                     float radius, int max_hamming, int32_t* matched_point_id)
{
    if (R > SLAMFE_PROJ_MAX_POINTS)
        return fail(c, SLAMFE_E_CAPACITY, "%d visible points; at most %d", R, SLAMFE_PROJ_MAX_POINTS);
    const bool from_visible_set = d_uv != nullptr;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int slot = slot_of(c, frame);
    if (slot < 0) return fail(c, SLAMFE_E_HANDLE, "frame handle expired or unknown");
    if (n_inl == 0) return SLAMFE_OK;
    if (R == 0) { // Init.cc:233: nothing to search
        for (int i = 0; i < n_inl; ++i) matched_point_id[i] = -1;
        return SLAMFE_OK;
    }
    // the frame's keypoint count: known on the host if the frame came through a
    // host-pointer call (its results were downloaded); else one read-back
    int32_t nkp = c->slot_count_host[(size_t)slot];
    if (nkp < 0) {
        HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->kp_count + slot, sizeof(int32_t), hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        nkp = c->h_cnt[0];
        c->slot_count_host[(size_t)slot] = nkp;
    }
    if (nkp > c->max_kp) nkp = c->max_kp;
    for (int i = 0; i < n_inl; ++i)
        if (kp_idx[i] < 0 || kp_idx[i] >= nkp)
            return fail(c, SLAMFE_E_INVALID, "kp_idx[%d] = %d outside the frame's %d keypoints", i,
                        kp_idx[i], nkp);
    int rc;
    // inputs travel as one block: inlier indices, then (host-supplied visible set
    // only) uv and ids -- one transfer instead of three
    const size_t o_idx = 0, o_uv = align256((size_t)n_inl * 4), o_ids = o_uv + align256((size_t)R * 8),
                 in_bytes = d_uv ? o_uv : o_ids + align256((size_t)R * 4);
    if ((rc = grow(c, c->pj_in, in_bytes))) return rc;
    if ((rc = grow(c, c->pj_cand, (size_t)n_inl * SLAMFE_PROJ_K * 4))) return rc;
    if ((rc = grow(c, c->pj_candn, (size_t)n_inl * 4))) return rc;
    // the grid over the visible set: already built by slamfe_reproject for this radius?
    const bool have_grid = from_visible_set && c->pj_grid_ready && radius == c->pj_grid_radius;
    if (!have_grid && (rc = grow_projection_grid(c, R))) return rc;
    hipStream_t s = c->stream;
    const size_t o_out = in_bytes;
    if ((rc = pin_reserve(c, o_out + (size_t)n_inl * 4))) return rc;
    HIPCHK(c, hipStreamSynchronize(s)); // the bounce buffer may still be in use
    memcpy(c->h_pin + o_idx, kp_idx, (size_t)n_inl * 4);
    if (!d_uv) {
        memcpy(c->h_pin + o_uv, uv, (size_t)R * 8);
        memcpy(c->h_pin + o_ids, point_ids, (size_t)R * 4);
        d_uv = (const float*)((const uint8_t*)c->pj_in.p + o_uv);
        d_ids = (const int32_t*)((const uint8_t*)c->pj_in.p + o_ids);
    }
    // Host-supplied visible set: everything goes up in one transfer.  Device-resident
    // set: the kernels read the inlier indices (a few KB) from the page-locked buffer
    // itself -- one transfer command less on a call that is a chain of short steps.
    const int32_t* d_idx = (const int32_t*)(c->h_pin + o_idx);
    if (!from_visible_set) {
        HIPCHK(c, hipMemcpyAsync(c->pj_in.p, c->h_pin, in_bytes, hipMemcpyHostToDevice, s));
        d_idx = (const int32_t*)c->pj_in.p;
    }
    if (!have_grid) {
        KCHK(c, slamfe_launch_projection_grid(s, d_uv, d_ids, R, nullptr, radius, c->p.max_width, c->p.max_height,
                                              (uint32_t*)c->pj_cells.p, c->pj_sorted.p,
                                              (uint32_t*)c->pj_cnt.p, (uint32_t*)c->pj_rank.p));
        // a grid over the device-resident visible set stays valid for further searches
        // with this radius; one over a host-supplied set replaces it
        c->pj_grid_ready = from_visible_set;
        c->pj_grid_radius = radius;
    }
    KCHK(c, slamfe_launch_projection(s, c->kps + (size_t)slot * c->max_kp,
                                     c->desc + (size_t)slot * c->max_kp * 32, nkp,
                                     d_idx, n_inl, d_ids, R, c->map_desc,
                                     radius, max_hamming, c->p.max_width, c->p.max_height,
                                     (const uint32_t*)c->pj_cells.p, c->pj_sorted.p, (uint32_t*)c->pj_cand.p,
                                     (int32_t*)c->pj_candn.p,
                                     // the resolve kernel writes the answers to the page-locked
                                     // buffer itself: no transfer command after it
                                     (int32_t*)(c->h_pin + o_out)));
    HIPCHK(c, hipStreamSynchronize(s));
    memcpy(matched_point_id, c->h_pin + o_out, (size_t)n_inl * 4);
    return SLAMFE_OK;
}

// cv::invert(DECOMP_LU) on a 4x4 float matrix: elimination with partial pivoting
// on [A | I] in float, the pivot's reciprocal kept and used by the
// back-substitution.  Returns false for a singular matrix.
bool invert4x4(const float* src, float* dst)
{
    float A[4][4], B[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            A[i][j] = src[i * 4 + j];
            B[i][j] = i == j ? 1.f : 0.f;
        }
    for (int i = 0; i < 4; ++i) {
        int k = i;
        for (int j = i + 1; j < 4; ++j)
            if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
        if (std::fabs(A[k][i]) < FLT_EPSILON * 10) return false;
        if (k != i)
            for (int j = 0; j < 4; ++j) {
                std::swap(A[i][j], A[k][j]);
                std::swap(B[i][j], B[k][j]);
            }
        const float d = -1 / A[i][i];
        for (int j = i + 1; j < 4; ++j) {
            const float alpha = A[j][i] * d;
            for (int q = i + 1; q < 4; ++q) A[j][q] += alpha * A[i][q];
            for (int q = 0; q < 4; ++q) B[j][q] += alpha * B[i][q];
        }
        A[i][i] = -d;
    }
    for (int i = 3; i >= 0; --i)
        for (int j = 0; j < 4; ++j) {
            float sum = B[i][j];
            for (int q = i + 1; q < 4; ++q) sum -= A[i][q] * B[q][j];
            dst[i * 4 + j] = B[i][j] = sum * A[i][i];
        }
    return true;
}

} // namespace

int slamfe_search_by_projection(slamfe_ctx* c, uint64_t frame, const int32_t* kp_idx, int n_inl,
                                const float* uv, const int32_t* point_ids, int R, float radius,
                                int max_hamming, int32_t* matched_point_id)
{
    if (!c || n_inl < 0 || R < 0 || (n_inl > 0 && (!kp_idx || !matched_point_id)) ||
        (R > 0 && (!uv || !point_ids)))
        return SLAMFE_E_INVALID;
    for (int j = 0; j < R; ++j)
        if (point_ids[j] < 0 || point_ids[j] >= c->p.map_capacity)
            return fail(c, SLAMFE_E_INVALID, "point id %d outside map_capacity %d", point_ids[j],
                        c->p.map_capacity);
    return search_projected(c, frame, kp_idx, n_inl, uv, point_ids, nullptr, nullptr, R, radius,
                            max_hamming, matched_point_id);
}

int slamfe_map_put_xyz(slamfe_ctx* c, const int32_t* point_ids, const float* xyz, int n)
{
    if (!c || n < 0 || (n > 0 && (!point_ids || !xyz))) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    for (int i = 0; i < n; ++i)
        if (point_ids[i] < 0 || point_ids[i] >= c->p.map_capacity)
            return fail(c, SLAMFE_E_CAPACITY, "point id %d outside map_capacity %d", point_ids[i],
                        c->p.map_capacity);
    if (n == 0) return SLAMFE_OK;
    int rc = pin_reserve(c, (size_t)n * 12);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(c->h_pin, xyz, (size_t)n * 12);
    int i = 0;
    while (i < n) { // runs of consecutive ids go in one transfer each
        int j = i + 1;
        while (j < n && point_ids[j] == point_ids[j - 1] + 1) ++j;
        HIPCHK(c, hipMemcpyAsync(c->map_xyz + (size_t)point_ids[i] * 3, c->h_pin + (size_t)i * 12,
                                 (size_t)(j - i) * 12, hipMemcpyHostToDevice, c->stream));
        i = j;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->vis_count = -1;
    c->pj_grid_ready = false;
    return SLAMFE_OK;
}

int slamfe_reproject(slamfe_ctx* c, const float* pose, const float* K, float th, float W, float H,
                     int npoints, float* uv, int32_t* point_idx, int cap, int* n_visible)
{
    if (!c || !pose || !K || npoints < 0) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    if (npoints > c->p.map_capacity)
        return fail(c, SLAMFE_E_CAPACITY, "%d points; map_capacity is %d", npoints, c->p.map_capacity);
    c->vis_count = -1;
    c->pj_grid_ready = false;
    float kf[16], M[16];
    if (!invert4x4(pose, kf)) return fail(c, SLAMFE_E_INVALID, "the keyframe pose is singular");
    // K4x4 * kf with K4x4 = [K 0; 0 1]; four float products summed left to right
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 4; ++q) {
            float k4[4] = {r < 3 ? K[r * 3] : 0.f, r < 3 ? K[r * 3 + 1] : 0.f, r < 3 ? K[r * 3 + 2] : 0.f,
                           r < 3 ? 0.f : 1.f};
            M[r * 4 + q] = k4[0] * kf[q] + k4[1] * kf[4 + q] + k4[2] * kf[8 + q] + k4[3] * kf[12 + q];
        }
    if (npoints == 0) {
        c->vis_count = 0;
        if (n_visible) *n_visible = 0;
        return SLAMFE_OK;
    }
    const size_t nwaves = ((size_t)npoints + 255) / 256 * 4;
    int rc;
    if ((rc = grow(c, c->rp_uv_all, (size_t)npoints * 8))) return rc;
    if ((rc = grow(c, c->rp_mask, nwaves * 8))) return rc;
    if ((rc = grow(c, c->rp_off, nwaves * 4))) return rc;
    if ((rc = grow(c, c->rp_total, 4))) return rc;
    if ((rc = grow(c, c->vis_uv, (size_t)npoints * 8))) return rc;
    if ((rc = grow(c, c->vis_idx, (size_t)npoints * 4))) return rc;
    hipStream_t s = c->stream;
    // The grid the search will walk (Init.cc:229-259 follows the reprojection loop
    // directly) is queued right behind the reprojection kernels, sized for all
    // npoints and reading the number of visible points where the scan left it, so
    // that the search call itself is candidates + resolve only.
    const bool grid_ahead = npoints <= SLAMFE_PROJ_MAX_POINTS && c->pj_grid_radius >= 0.f;
    if (grid_ahead && (rc = grow_projection_grid(c, npoints))) return rc;
    if (!c->ev_count) HIPCHK(c, hipEventCreateWithFlags(&c->ev_count, hipEventDisableTiming));
    KCHK(c, slamfe_launch_reproject(s, c->map_xyz, npoints, M, th, W, H, (float*)c->rp_uv_all.p,
                                    (uint64_t*)c->rp_mask.p, (int32_t*)c->rp_off.p,
                                    (int32_t*)c->rp_total.p, (float*)c->vis_uv.p, (int32_t*)c->vis_idx.p));
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->rp_total.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(c->ev_count, s));
    if (grid_ahead)
        KCHK(c, slamfe_launch_projection_grid(s, (const float*)c->vis_uv.p, (const int32_t*)c->vis_idx.p,
                                              npoints, (const int32_t*)c->rp_total.p, c->pj_grid_radius,
                                              c->p.max_width, c->p.max_height, (uint32_t*)c->pj_cells.p,
                                              c->pj_sorted.p, (uint32_t*)c->pj_cnt.p, (uint32_t*)c->pj_rank.p));
    // the host waits for the count only; the grid is built while it goes on
    HIPCHK(c, hipEventSynchronize(c->ev_count));
    const int R = c->h_cnt[0];
    c->vis_count = R;
    c->pj_grid_ready = grid_ahead && R > 0;
    if (n_visible) *n_visible = R;
    if (!uv && !point_idx) return SLAMFE_OK;
    if (R > cap) return fail(c, SLAMFE_E_CAPACITY, "%d visible points exceed the caller's capacity %d", R, cap);
    if (R == 0) return SLAMFE_OK;
    if ((rc = pin_reserve(c, (size_t)R * 12))) return rc;
    if (uv) HIPCHK(c, hipMemcpyAsync(c->h_pin, c->vis_uv.p, (size_t)R * 8, hipMemcpyDeviceToHost, s));
    if (point_idx)
        HIPCHK(c, hipMemcpyAsync(c->h_pin + (size_t)R * 8, c->vis_idx.p, (size_t)R * 4,
                                 hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (uv) memcpy(uv, c->h_pin, (size_t)R * 8);
    if (point_idx) memcpy(point_idx, c->h_pin + (size_t)R * 8, (size_t)R * 4);
    return SLAMFE_OK;
}

int slamfe_triangulate(slamfe_ctx* c, const float* P1, const float* P2, const float* pts1,
                       const float* pts2, int n, float* xyz)
{
    if (!c || !P1 || !P2 || n < 0 || (n > 0 && (!pts1 || !pts2 || !xyz))) return SLAMFE_E_INVALID;
    if (n == 0) return SLAMFE_OK;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    int rc;
    const size_t pb = (size_t)n * 8, ob = (size_t)n * 12;
    if ((rc = grow(c, c->tri_in, 2 * pb))) return rc;
    if ((rc = grow(c, c->tri_out, ob))) return rc;
    if ((rc = pin_reserve(c, 2 * pb + ob))) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipStreamSynchronize(s));
    memcpy(c->h_pin, pts1, pb);
    memcpy(c->h_pin + pb, pts2, pb);
    HIPCHK(c, hipMemcpyAsync(c->tri_in.p, c->h_pin, 2 * pb, hipMemcpyHostToDevice, s));
    KCHK(c, slamfe_launch_triangulate(s, P1, P2, (const float*)c->tri_in.p,
                                      (const float*)((const uint8_t*)c->tri_in.p + pb), n,
                                      (float*)c->tri_out.p));
    HIPCHK(c, hipMemcpyAsync(c->h_pin + 2 * pb, c->tri_out.p, ob, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    memcpy(xyz, c->h_pin + 2 * pb, ob);
    return SLAMFE_OK;
}

int slamfe_search_by_projection_visible(slamfe_ctx* c, uint64_t frame, const int32_t* kp_idx, int n_inl,
                                        float radius, int max_hamming, int32_t* matched_point_id)
{
    if (!c || n_inl < 0 || (n_inl > 0 && (!kp_idx || !matched_point_id))) return SLAMFE_E_INVALID;
    if (c->vis_count < 0)
        return fail(c, SLAMFE_E_INVALID, "no visible set on the device: call slamfe_reproject first");
    return search_projected(c, frame, kp_idx, n_inl, nullptr, nullptr, (const float*)c->vis_uv.p,
                            (const int32_t*)c->vis_idx.p, c->vis_count, radius, max_hamming,
                            matched_point_id);
}

int slamfe_debug_probe(slamfe_ctx* c, int stage)
{
    if (!c || stage < 0 || stage > SLAMFE_STAGE_MATCH) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    while (stage && c->probe_ev.size() < 128) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->probe_ev.push_back(e);
    }
    c->probe_stage = stage;
    c->probe_used = 0;
    return SLAMFE_OK;
}

int slamfe_debug_probe_read(slamfe_ctx* c, float* avg_ms, int* n)
{
    if (!c || !avg_ms) return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sum = 0;
    const int cnt = (int)(c->probe_used / 2);
    for (int i = 0; i < cnt; ++i) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->probe_ev[2 * (size_t)i], c->probe_ev[2 * (size_t)i + 1]));
        sum += ms;
    }
    *avg_ms = cnt ? (float)(sum / cnt) : 0.f;
    if (n) *n = cnt;
    return SLAMFE_OK;
}

int slamfe_debug_time_stage(slamfe_ctx* c, int stage, int nframes, int reps, float* avg_ms)
{
    if (!c || !avg_ms || reps < 1 || c->gw == 0 || nframes < 1 || nframes > c->p.max_batch)
        return SLAMFE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->p.device_id));
    const slamfe_dev_geom& g = c->geom;
    hipStream_t s = c->stream;
    // the slots of the most recent extraction of nframes frames
    int slot_base = (int)((c->next_seq - (uint64_t)nframes) % (uint64_t)c->nslots);
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) {
        switch (stage) {
        case SLAMFE_STAGE_PYRAMID:
            KCHK(c, slamfe_launch_pyramid(s, g, nframes, c->pyr, c->taps, c->cur_l0));
            break;
        case SLAMFE_STAGE_FAST:
            KCHK(c, slamfe_launch_fast_blur(s, g, nframes, c->pyr, c->blur, c->p.fast_threshold,
                                            c->p.edge_threshold, c->cand, c->cand_cnt, c->cur_l0));
            break;
        case SLAMFE_STAGE_SELECT:
            KCHK(c, slamfe_launch_select(s, g, nframes, c->cand, c->cand_cnt, c->band_hist, c->band_off,
                                         c->level_kept, c->level_cutoff, slot_base, c->nslots,
                                         c->kp_key, c->kp_level, c->kp_count, c->max_kp, c->sort_lds, false));
            break;
        case SLAMFE_STAGE_ORIENT:
            KCHK(c, slamfe_launch_orient(s, g, nframes, c->pyr, slot_base, c->nslots, c->kp_key,
                                         c->kp_level, c->kp_count, c->kp_mom, c->kps, c->kp_rot,
                                         c->max_kp, c->p.mode, c->p.patch_size, c->cur_l0));
            break;
        case SLAMFE_STAGE_DESCRIBE:
            KCHK(c, slamfe_launch_describe(s, g, nframes, c->blur, slot_base, c->nslots, c->kp_key,
                                           c->kp_level, c->kp_count, c->kp_rot, c->desc, c->max_kp));
            break;
        case SLAMFE_STAGE_MATCH: {
            int np = nframes > 1 ? nframes - 1 : 1, qpt, chunk, nchunks;
            slamfe_match_plan(c->max_kp, c->max_kp, np, c->p.match_kernel, &qpt, &chunk, &nchunks);
            int rc = grow(c, c->match_part, (size_t)np * nchunks * c->max_kp * 8);
            if (rc) return rc;
            rc = grow(c, c->match_o, (size_t)nframes * c->max_kp * sizeof(slamfe_match2));
            if (rc) return rc;
            // frame i against frame i+1 (the last one against itself)
            KCHK(c, slamfe_launch_match(s, c->desc, 0, c->kp_count, c->desc, 0, c->kp_count, c->max_kp,
                                        qpt, chunk, nchunks,
                                        (uint32_t*)c->match_part.p, c->p.ratio,
                                        (slamfe_match2*)c->match_o.p, nullptr, 0, np, slot_base,
                                        nframes > 1 ? slot_base + 1 : slot_base, c->nslots));
            break;
        }
        default:
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return fail(c, SLAMFE_E_INVALID, "unknown stage %d", stage);
        }
    }
    HIPCHK(c, hipEventRecord(e1, s));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)reps;
    return SLAMFE_OK;
}

int slamfe_debug_read_level(slamfe_ctx* c, int frame_in_batch, int level, int blurred, uint8_t* out,
                            int out_capacity, int* w, int* h)
{
    if (!c || !out || c->gw == 0) return SLAMFE_E_INVALID;
    if (frame_in_batch < 0 || frame_in_batch >= c->p.max_batch || level < 0 ||
        level >= c->geom.nlevels)
        return fail(c, SLAMFE_E_INVALID, "no such frame / level");
    HIPCHK(c, hipSetDevice(c->p.device_id));
    const slamfe_dev_level& L = c->geom.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if ((size_t)L.w * L.h > (size_t)out_capacity)
        return fail(c, SLAMFE_E_CAPACITY, "level needs %d bytes", L.w * L.h);
    const uint8_t* src = (blurred ? c->blur : c->pyr) + (size_t)frame_in_batch * c->geom.frame_bytes + L.off;
    if (!blurred && level == 0 && c->cur_l0.base) // possibly the caller's frames, used in place
        src = c->cur_l0.base + (size_t)frame_in_batch * c->cur_l0.frame_stride;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!blurred) {
        HIPCHK(c, hipMemcpy2D(out, (size_t)L.w, src, (size_t)L.pitch, (size_t)L.w, (size_t)L.h,
                              hipMemcpyDeviceToHost));
        return SLAMFE_OK;
    }
    // the blurred level is tiled: 16 x 8 pixels per 128-byte tile, tiles row-major
    const size_t rows8 = (size_t)((L.h + 7) & ~7);
    std::vector<uint8_t> tiled((size_t)L.pitch * rows8);
    HIPCHK(c, hipMemcpy(tiled.data(), src, tiled.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x)
            out[(size_t)y * L.w + x] =
                tiled[((size_t)(y >> 3) * (size_t)(L.pitch >> 4) + (size_t)(x >> 4)) * 128 + (size_t)(y & 7) * 16 + (size_t)(x & 15)];
    return SLAMFE_OK;
}

} // extern "C"
