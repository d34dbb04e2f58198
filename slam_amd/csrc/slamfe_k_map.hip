// slamfe_k_map.hip -- search by projection, map-point reprojection, triangulation
#include "slamfe_device.h"

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

// The consumed-point bitset of k_proj_resolve may need more than the default
// 64 KiB of dynamic LDS (very large local maps).
int slamfe_projection_init(void)
{
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_resolve),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SLAMFE_PROJ_MAX_POINTS / 8 +
                                        (2 * (PROJ_CHUNK * PROJ_STRIDE + PROJ_CHUNK) + PROJ_CHUNK +
                                         2 * PROJ_CLAIM) * 4 + 64);
}

// Raise the dynamic-LDS limits of the two kernels that may need more than the
// default 64 KiB.
int slamfe_select_init(size_t sort_lds_bytes);
int slamfe_kernels_init(size_t sort_lds_bytes)
{
    int e = slamfe_select_init(sort_lds_bytes);
    if (e) return e;
    return slamfe_projection_init();
}
