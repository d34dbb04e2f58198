// slamfe_k_select.hip -- keypoint selection (retain-best) and ordered compaction
#include "slamfe_device.h"

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

// k_band_hist + k_level_cut as one launch: one workgroup of 1024 threads per frame
// (frames of at most SELECT_PLAN_MAX_TILES tiles; larger ones take the two kernels).  Pass 1 builds the eight level
// histograms in LDS from every candidate list of the frame; a wavefront per
// level then finds the level's cut; pass 2 counts every tile's kept candidates
// into its band; a wavefront per level turns the band counts into offsets.
// Outputs are those of k_level_cut (band_off, level_kept, level_cutoff).
#define SELECT_PLAN_THREADS 1024
#define SELECT_PLAN_MAX_TILES 8192
#define SELECT_PLAN_MAX_BANDS 1024

// Visit the keys of one tile's list: four 16-byte loads in flight at a time.
template <class F>
__device__ __forceinline__ void tile_keys(const uint32_t* __restrict__ list, uint32_t n, F fn)
{
    for (uint32_t k0 = 0; k0 < n; k0 += 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = k0 + 4u * u < n ? *reinterpret_cast<const uint4*>(list + k0 + 4 * u) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t b = k0 + 4u * u;
            if (b + 0 < n) fn(v[u].x);
            if (b + 1 < n) fn(v[u].y);
            if (b + 2 < n) fn(v[u].z);
            if (b + 3 < n) fn(v[u].w);
        }
    }
}

__global__ __launch_bounds__(SELECT_PLAN_THREADS) void k_select_plan(slamfe_dev_geom g,
                                                                      const uint32_t* __restrict__ cand,
                                                                      const uint32_t* __restrict__ cand_cnt,
                                                                      uint32_t* __restrict__ band_off,
                                                                      uint32_t* __restrict__ level_kept,
                                                                      int32_t* __restrict__ level_cutoff)
{
    __shared__ uint32_t hist[SLAMFE_MAX_LEVELS][256];
    __shared__ uint32_t bkept[SELECT_PLAN_MAX_BANDS];
    __shared__ int32_t cut_s[SLAMFE_MAX_LEVELS];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < SLAMFE_MAX_LEVELS * 256; i += SELECT_PLAN_THREADS) (&hist[0][0])[i] = 0;
    for (int i = tid; i < g.total_bands; i += SELECT_PLAN_THREADS) bkept[i] = 0;
    const size_t tbase = (size_t)f * g.total_tiles;
    // the counts of this thread's tiles (tile = tid, tid + 1024, ...), fetched together
    constexpr int TPT = SELECT_PLAN_MAX_TILES / SELECT_PLAN_THREADS;
    uint32_t cnt[TPT];
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int t = tid + u * SELECT_PLAN_THREADS;
        cnt[u] = t < g.total_tiles ? cand_cnt[tbase + t] : 0u;
    }
    __syncthreads();
    // ---- pass 1: level histograms
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int t = tid + u * SELECT_PLAN_THREADS;
        if (cnt[u] == 0) continue;
        uint32_t* h = hist[find_level_by_tile(g, t)];
        tile_keys(cand + (tbase + t) * SLAMFE_TILE_CAND, cnt[u], [&](uint32_t k) { atomicAdd(&h[SLAMFE_KEY_S(k)], 1u); });
    }
    __syncthreads();
    // ---- the cut of every level: wavefront = level, lane l holds scores 255 - 4l .. 252 - 4l
    // so that a prefix over the lanes is a suffix over the scores
    if (wv < g.nlevels) {
        const int l = wv, quota = g.lv[l].quota;
        const uint32_t* h = hist[l];
        const int s0 = 255 - 4 * lane; // the lane's highest score
        const uint32_t b0 = h[s0], b1 = h[s0 - 1], b2 = h[s0 - 2], b3 = h[s0 - 3];
        uint32_t incl = b0 + b1 + b2 + b3;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        const uint32_t before = incl - (b0 + b1 + b2 + b3); // candidates with a score above the lane's
        // suffix[s] for the lane's four scores
        const uint32_t f0 = before + b0, f1 = f0 + b1, f2 = f1 + b2, f3 = f2 + b3;
        // number of scores s with suffix[s] >= quota (suffix grows as s falls)
        const uint32_t q = (uint32_t)max(quota, 0);
        const int mine = (f0 >= q) + (f1 >= q) + (f2 >= q) + (f3 >= q);
        const int c = wave_sum(mine);
        int cut;
        uint32_t kept;
        if (total <= (uint32_t)max(quota, 0) || (quota > 0 && total <= (uint32_t)quota)) {
            cut = 0;
            kept = total;
        } else if (quota <= 0) {
            cut = 256;
            kept = 0;
        } else {
            cut = c - 1;
            // suffix[cut]: held by the lane whose scores include cut
            const int owner = (255 - cut) >> 2, which = (255 - cut) & 3;
            const uint32_t mysuf = which == 0 ? f0 : which == 1 ? f1 : which == 2 ? f2 : f3;
            kept = (uint32_t)__shfl((int)mysuf, owner);
        }
        if (lane == 0) {
            cut_s[l] = cut;
            level_kept[f * SLAMFE_MAX_LEVELS + l] = kept;
            level_cutoff[f * SLAMFE_MAX_LEVELS + l] = cut;
        }
    }
    __syncthreads();
    // ---- pass 2: kept candidates of every tile, summed per band
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int t = tid + u * SELECT_PLAN_THREADS;
        if (cnt[u] == 0) continue;
        const int l = find_level_by_tile(g, t);
        const int cut = cut_s[l];
        uint32_t k = 0;
        tile_keys(cand + (tbase + t) * SLAMFE_TILE_CAND, cnt[u], [&](uint32_t key) { k += (int)SLAMFE_KEY_S(key) >= cut; });
        if (k) atomicAdd(&bkept[g.lv[l].band_base + (t - g.lv[l].tile_base) / g.lv[l].tiles_x], k);
    }
    __syncthreads();
    // ---- band offsets inside each level: wavefront = level, four bands per lane and step
    if (wv < g.nlevels) {
        const int l = wv, nb = g.lv[l].tiles_y, bb = g.lv[l].band_base;
        uint32_t carry = 0;
        for (int b0 = 0; b0 < nb; b0 += 256) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = b0 + 4 * lane + k;
                v[k] = b < nb ? bkept[bb + b] : 0u;
            }
            const uint32_t sum = v[0] + v[1] + v[2] + v[3];
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += o;
            }
            uint32_t off = carry + incl - sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = b0 + 4 * lane + k;
                if (b < nb) band_off[(size_t)f * g.total_bands + bb + b] = off;
                off += v[k];
            }
            carry += (uint32_t)__shfl((int)incl, 63);
        }
    }
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

int slamfe_launch_select(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint32_t* cand,
                         const uint32_t* cand_cnt, uint32_t* band_hist, uint32_t* band_off,
                         uint32_t* level_kept, int32_t* level_cutoff, int slot_base, int nslots,
                         uint32_t* kp_key, uint32_t* kp_level, int32_t* kp_count, int max_kp,
                         size_t sort_lds_bytes, bool few)
{
    (void)few;
    if (g.total_tiles <= SELECT_PLAN_MAX_TILES && g.total_bands <= SELECT_PLAN_MAX_BANDS) {
        // histograms, cuts and band offsets in one launch, one workgroup per frame
        // (frames up to about 1920 x 1080; measured on MI355X, 256 frames of 640 x 480:
        // 0.054 ms against 0.091 ms for the three-launch form, which moves a
        // histogram per tile row through memory)
        hipLaunchKernelGGL(k_select_plan, dim3(nframes), dim3(SELECT_PLAN_THREADS), 0, s, g, cand, cand_cnt,
                           band_off, level_kept, level_cutoff);
        LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_band_hist, frame_grid((g.total_bands + 3) / 4, nframes), dim3(256), 0, s, g,
                           cand, cand_cnt, band_hist, nframes);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_level_cut, frame_grid(g.nlevels, nframes), dim3(256), 0, s, g, band_hist,
                           band_off, level_kept, level_cutoff, nframes);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_select_write, frame_grid((g.total_bands + 3) / 4, nframes), dim3(256), sort_lds_bytes,
                       s, g, cand, cand_cnt, band_off, level_kept, level_cutoff, slot_base, nslots,
                       kp_key, kp_level, kp_count, max_kp, nframes);
    LAUNCH_CHECK();
    return 0;
}

// The sort buffer of k_select_write may need more than the default 64 KiB of
// dynamic LDS (4K-wide frames).
int slamfe_select_init(size_t sort_lds_bytes)
{
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_select_write),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sort_lds_bytes > 65536 ? sort_lds_bytes : 65536));
}
