// slamfe_k_map.hip -- search by projection, map-point reprojection, triangulation
#include "slamfe_device.h"

// ---------------------------------------------------------------------------
// K9  search by projection (Init.cc:229-259).
//
// Pass 0: bucket the visible points into a uniform grid whose cells are at
// least one search radius wide: count per cell, prefix sum, scatter of
// {u, v, position, point id} records into cell order -- one workgroup for the
// sets the reference has (up to about 20 000 points), three small launches for
// larger ones.  The grid depends on the visible set and the radius only, so it
// can be built as soon as the set exists, ahead of the search call.
// Pass 1 (parallel, one wavefront per inlier): test the points of the 3 x 3
// cells around the inlier's keypoint -- pixel distance gate first (Init.cc:389-391
// arithmetic: float differences, double squares and sqrt, narrowed to float),
// then Hamming distance against the device-resident map descriptor row -- and
// keep the PROJ_K best candidates as keys (hamming << 20 | position), sorted.
// The keys carry the point's position in the caller's (ascending) list, so the
// order of the records inside a cell does not matter.
// Pass 2 (one wavefront, in inlier order): take the first candidate not yet
// consumed; that is exactly the reference's "minimum over the remaining
// points, then erase".  An inlier whose list overflowed and ran dry is
// re-scanned against all points by the whole wave.
// ---------------------------------------------------------------------------
#define PROJ_K 32
#define PROJ_GRID_MAX 128   /* cells per axis */

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

// Grid over the projected points.  A coordinate's cell is
// clamp(floor(x / cell), 0, n - 1): the clamp is monotone and never widens a
// difference, so two positions no further apart than one cell width (>= the
// radius with a margin far above the rounding of the division) land in the same
// or in adjacent cells, whatever their values -- points outside the image, at
// infinity or NaN included (NaN goes to cell 0 and never passes the gate).
struct slamfe_proj_grid {
    double inv_cell;
    int gx, gy;
};

__device__ __forceinline__ int proj_cell_coord(float x, double inv_cell, int n)
{
    const double c = floor((double)x * inv_cell);
    return (int)fmin(fmax(c, 0.0), (double)(n - 1)); // fmax(NaN, 0) = 0
}

// One workgroup builds the whole grid in LDS: for the visible sets the
// reference has, the work is a few loads and LDS atomics per point, too little
// for a multi-kernel count / scan / scatter.  Its time is the LDS atomics
// (points cluster, so many lanes hit one counter and are served one after the
// other) and the memory latency of the one CU that runs it.  So a thread
// fetches its points in one batch of independent loads and keeps them in
// registers (R <= 1024 * PROJ_GRID_KEEP), and the count pass's atomics return
// the point's rank inside its cell, which makes the scatter pass atomic-free:
// position = cell start + rank.
#define PROJ_GRID_KEEP 20
#define PROJ_GRID_ONE_WG (1024 * PROJ_GRID_KEEP)
static_assert(PROJ_GRID_ONE_WG == SLAMFE_PROJ_ONE_WG_POINTS, "one-workgroup limit of the grid build");

// exclusive prefix sum over ncells counters by one workgroup of 1024 threads:
// cell_start[c] = number of points in cells before c, cell_start[ncells] = total.
// `starts`, if not null, receives the same values (an LDS copy for the caller).
__device__ __forceinline__ void proj_scan_cells(const uint32_t* cnt, int ncells, uint32_t* __restrict__ cell_start,
                                                uint32_t* starts, uint32_t* wtot)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (ncells + 1023) / 1024;
    const int c0 = min(tid * per, ncells), c1 = min(c0 + per, ncells);
    uint32_t sum = 0;
    for (int c = c0; c < c1; ++c) sum += cnt[c];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    uint32_t before = incl - sum;
    for (int k = 0; k < wv; ++k) before += wtot[k];
    for (int c = c0; c < c1; ++c) {
        const uint32_t n = cnt[c];
        if (starts) starts[c] = before;
        cell_start[c] = before;
        before += n;
    }
    if (tid == 1023) cell_start[ncells] = before; // == R
}

__global__ __launch_bounds__(1024) void k_proj_grid(const float2* __restrict__ uv,
                                                     const int32_t* __restrict__ point_ids, int R,
                                                     const int32_t* __restrict__ d_R, slamfe_proj_grid G,
                                                     uint32_t* __restrict__ cell_start,
                                                     uint4* __restrict__ sorted)
{
    extern __shared__ uint32_t glds[]; // ncells counters (later: cell starts), then 16 wave totals
    if (d_R) R = min(R, *d_R); // the number of points may be known on the device only
    const int ncells = G.gx * G.gy;
    uint32_t* cnt = glds;
    uint32_t* wtot = glds + ncells;
    const int tid = threadIdx.x;
    for (int c = tid; c < ncells; c += 1024) cnt[c] = 0;
    __syncthreads();
    float2 kp[PROJ_GRID_KEEP];
    uint32_t kcr[PROJ_GRID_KEEP]; // cell << 16 | rank inside the cell (both below 2^16 here)
#pragma unroll
    for (int u = 0; u < PROJ_GRID_KEEP; ++u) {
        const int j = u * 1024 + tid;
        kp[u] = j < R ? uv[j] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PROJ_GRID_KEEP; ++u) {
        const int j = u * 1024 + tid;
        kcr[u] = 0;
        if (j < R) {
            const uint32_t cell = (uint32_t)(proj_cell_coord(kp[u].y, G.inv_cell, G.gy) * G.gx +
                                             proj_cell_coord(kp[u].x, G.inv_cell, G.gx));
            kcr[u] = (cell << 16) | atomicAdd(&cnt[cell], 1u);
        }
    }
    __syncthreads();
    proj_scan_cells(cnt, ncells, cell_start, cnt, wtot); // cnt: from here on the cells' starts
    __syncthreads();
    int32_t id[PROJ_GRID_KEEP];
#pragma unroll
    for (int u = 0; u < PROJ_GRID_KEEP; ++u) {
        const int j = u * 1024 + tid;
        id[u] = j < R ? point_ids[j] : 0;
    }
#pragma unroll
    for (int u = 0; u < PROJ_GRID_KEEP; ++u) {
        const int j = u * 1024 + tid;
        if (j < R)
            sorted[cnt[kcr[u] >> 16] + (kcr[u] & 0xffffu)] =
                make_uint4(__float_as_uint(kp[u].x), __float_as_uint(kp[u].y), (uint32_t)j, (uint32_t)id[u]);
    }
}

// Larger visible sets: the same three steps as launches of their own, so that the
// whole chip counts and scatters.  Count: one lane per point, the atomic on the
// cell's counter (in memory, zeroed by the caller) returns the point's rank in its
// cell.  Scan: one workgroup.  Scatter: position = cell start + rank, no atomics.
__global__ __launch_bounds__(256) void k_proj_grid_count(const float2* __restrict__ uv, int R,
                                                          const int32_t* __restrict__ d_R,
                                                          slamfe_proj_grid G, uint32_t* __restrict__ cell_cnt,
                                                          uint32_t* __restrict__ rank)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (d_R) R = min(R, *d_R);
    if (j >= R) return;
    const float2 p = uv[j];
    const int cell = proj_cell_coord(p.y, G.inv_cell, G.gy) * G.gx + proj_cell_coord(p.x, G.inv_cell, G.gx);
    rank[j] = atomicAdd(&cell_cnt[cell], 1u);
}

__global__ __launch_bounds__(1024) void k_proj_grid_scan(const uint32_t* __restrict__ cell_cnt, int ncells,
                                                          uint32_t* __restrict__ cell_start)
{
    __shared__ uint32_t wtot[16];
    proj_scan_cells(cell_cnt, ncells, cell_start, nullptr, wtot);
}

__global__ __launch_bounds__(256) void k_proj_grid_scatter(const float2* __restrict__ uv,
                                                            const int32_t* __restrict__ point_ids, int R,
                                                            const int32_t* __restrict__ d_R,
                                                            slamfe_proj_grid G,
                                                            const uint32_t* __restrict__ cell_start,
                                                            const uint32_t* __restrict__ rank,
                                                            uint4* __restrict__ sorted)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (d_R) R = min(R, *d_R);
    if (j >= R) return;
    const float2 p = uv[j];
    const int cell = proj_cell_coord(p.y, G.inv_cell, G.gy) * G.gx + proj_cell_coord(p.x, G.inv_cell, G.gx);
    sorted[cell_start[cell] + rank[j]] =
        make_uint4(__float_as_uint(p.x), __float_as_uint(p.y), (uint32_t)j, (uint32_t)point_ids[j]);
}

// Bitonic sort of 64 keys held one per lane, ascending by lane.
__device__ __forceinline__ uint32_t proj_wave_sort64(uint32_t v)
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

// The points of the 3 x 3 cells around (kx, ky) as one index space: the three
// cells of a grid row are neighbours in cell order, so a window is three ranges
// of `sorted`.
struct proj_window {
    uint32_t beg[3], len[3], total;
};

__device__ __forceinline__ proj_window proj_window_of(float kx, float ky, const slamfe_proj_grid& G,
                                                      const uint32_t* __restrict__ cell_start)
{
    const int cx = proj_cell_coord(kx, G.inv_cell, G.gx), cy = proj_cell_coord(ky, G.inv_cell, G.gy);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.gx - 1);
    proj_window w;
    w.total = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int yy = cy - 1 + d;
        const bool in = yy >= 0 && yy < G.gy;
        const int row = in ? yy : cy;
        // (the keypoint is the wavefront's: make the range bounds scalar)
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)cell_start[row * G.gx + x0]);
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)cell_start[row * G.gx + x1 + 1]);
        w.beg[d] = b;
        w.len[d] = in ? e - b : 0u;
        w.total += w.len[d];
    }
    return w;
}

__device__ __forceinline__ uint32_t proj_window_at(const proj_window& w, uint32_t t)
{
    return t < w.len[0] ? w.beg[0] + t
                        : (t < w.len[0] + w.len[1] ? w.beg[1] + (t - w.len[0])
                                                   : w.beg[2] + (t - w.len[0] - w.len[1]));
}

__device__ __forceinline__ int proj_hamming(const uint32_t* __restrict__ row, const uint4& q0, const uint4& q1)
{
    const uint4* mp = reinterpret_cast<const uint4*>(row);
    const uint4 a = mp[0], b = mp[1];
    return __popc(a.x ^ q0.x) + __popc(a.y ^ q0.y) + __popc(a.z ^ q0.z) + __popc(a.w ^ q0.w) +
           __popc(b.x ^ q1.x) + __popc(b.y ^ q1.y) + __popc(b.z ^ q1.z) + __popc(b.w ^ q1.w);
}

// Compare-exchange network that sorts a bitonic sequence of 64 keys (one per
// lane) into ascending order.
__device__ __forceinline__ uint32_t proj_wave_merge64(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, j);
        v = (lane & j) == 0 ? min(v, o) : max(v, o);
    }
    return v;
}

// One wavefront per inlier, four per workgroup, no workgroup barriers.  The
// window is walked PROJ_UNROLL x 64 records at a time so that the record loads
// of a step, and then its descriptor-row loads, are in flight together (the
// kernel is a chain of dependent loads).  The wave keeps the 64 smallest keys seen
// so far sorted in registers, one per lane.  Keys that pass both gates and are
// smaller than the current PROJ_K-th smallest wait in a 64-entry LDS buffer; a
// full buffer is sorted (register bitonic network), laid against the kept keys in
// opposite order -- the lane-wise minimum of an ascending and a descending run
// holds the 64 smallest of both and is bitonic -- and merged.  Any number of
// candidates is ranked this way; dense maps cost a few more merges.
#define PROJ_UNROLL 4
__global__ __launch_bounds__(256) void k_proj_candidates(
    const slamfe_keypoint* __restrict__ kps, const uint32_t* __restrict__ fdesc, int nkp,
    const int32_t* __restrict__ kp_idx, int n_inl, slamfe_proj_grid G,
    const uint32_t* __restrict__ cell_start, const uint4* __restrict__ sorted,
    const uint32_t* __restrict__ map_desc, double gate_dd, int max_hamming,
    uint32_t* __restrict__ cand_out, int32_t* __restrict__ cand_n)
{
    __shared__ uint32_t pending[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv);
    if (i >= n_inl) return; // whole wavefronts leave; no workgroup barrier below
    uint32_t* my = pending[wv];
    const int idx = kp_idx[i];
    if (idx < 0 || idx >= nkp) { // rejected by the host before the launch; never taken
        if (lane == 0) cand_n[i] = 0;
        return;
    }
    const float kx = kps[idx].x, ky = kps[idx].y;
    const uint4* qp = reinterpret_cast<const uint4*>(fdesc + (size_t)idx * 8);
    const uint4 q0 = qp[0], q1 = qp[1];
    const proj_window w = proj_window_of(kx, ky, G, cell_start);
    uint32_t best = 0xffffffffu; // the 64 smallest keys so far, ascending over the lanes
    uint32_t kth = 0xffffffffu;  // the PROJ_K-th smallest so far: larger keys cannot matter
    uint32_t npend = 0, ntotal = 0; // keys waiting in LDS / keys that passed both gates (uniform)
    auto flush = [&]() {
        wave_lds_fence();
        uint32_t v = (uint32_t)lane < npend ? my[lane] : 0xffffffffu;
        v = proj_wave_sort64(v);
        v = (uint32_t)__shfl((int)v, 63 - lane); // descending
        best = proj_wave_merge64(min(best, v));
        kth = (uint32_t)__builtin_amdgcn_readlane((int)best, PROJ_K - 1);
        npend = 0;
        wave_lds_fence();
    };
    for (uint32_t t0 = 0; t0 < w.total; t0 += 64 * PROJ_UNROLL) { // uniform
        uint4 e[PROJ_UNROLL];
        bool in[PROJ_UNROLL];
#pragma unroll
        for (int u = 0; u < PROJ_UNROLL; ++u) {
            const uint32_t t = t0 + (uint32_t)(u * 64 + lane);
            in[u] = t < w.total;
            e[u] = make_uint4(0, 0, 0, 0);
            if (in[u]) e[u] = sorted[proj_window_at(w, t)];
        }
        int h[PROJ_UNROLL];
#pragma unroll
        for (int u = 0; u < PROJ_UNROLL; ++u) {
            h[u] = 1 << 10;
            if (in[u] && proj_gate(__uint_as_float(e[u].x), __uint_as_float(e[u].y), kx, ky, gate_dd))
                h[u] = proj_hamming(map_desc + (size_t)e[u].w * 8, q0, q1);
        }
#pragma unroll
        for (int u = 0; u < PROJ_UNROLL; ++u) {
            const bool pass = h[u] <= max_hamming;
            ntotal += (uint32_t)__popcll(__ballot(pass));
            const uint32_t key = ((uint32_t)h[u] << 20) | e[u].z;
            const bool keep = pass && key < kth;
            const unsigned long long m = __ballot(keep);
            const uint32_t c = (uint32_t)__popcll(m);
            if (c == 0) continue;         // uniform
            if (npend + c > 64) flush();  // uniform; the keys of this step are compared with the old kth: harmless
            if (keep) my[npend + lane_rank(m)] = key;
            npend += c;
        }
    }
    if (npend) flush();
    if (lane < PROJ_K && best != 0xffffffffu) cand_out[(size_t)i * PROJ_K + lane] = best;
    // cand_n: number of stored candidates; bit 30 set if more passed the gates
    if (lane == 0)
        cand_n[i] = (int32_t)(ntotal < PROJ_K ? ntotal : (PROJ_K | (ntotal > PROJ_K ? (1 << 30) : 0)));
}

#define PROJ_CHUNK 64 /* inliers per staged chunk */
#define PROJ_STRIDE (PROJ_K + 1) /* LDS words per staged list: odd, so lanes hit different banks */
#define PROJ_CLAIM 4096          /* entries of a proposal table */

__global__ __launch_bounds__(256) void k_proj_resolve(
    const slamfe_keypoint* __restrict__ kps, const uint32_t* __restrict__ fdesc, int nkp,
    const int32_t* __restrict__ kp_idx, int n_inl, const int32_t* __restrict__ point_ids, int R,
    slamfe_proj_grid G, const uint32_t* __restrict__ cell_start, const uint4* __restrict__ sorted,
    const uint32_t* __restrict__ map_desc, double gate_dd, int max_hamming,
    const uint32_t* __restrict__ cand, const int32_t* __restrict__ cand_n,
    int32_t* __restrict__ matched)
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
        // 16-byte pieces, all of a thread's loads issued before the first store: the
        // copy is one memory latency, not one per piece
        const uint4* src = reinterpret_cast<const uint4*>(cand + (size_t)c0 * PROJ_K);
        constexpr int PIECES = PROJ_CHUNK * PROJ_K / 4; // 512
        uint4 v[(PIECES + 191) / 192];
#pragma unroll
        for (int u = 0; u < (PIECES + 191) / 192; ++u) {
            const int e = t + u * nt;
            v[u] = make_uint4(0, 0, 0, 0);
            if (u * nt < PIECES && e < nc * (PROJ_K / 4)) v[u] = src[e];
        }
#pragma unroll
        for (int u = 0; u < (PIECES + 191) / 192; ++u) {
            const int e = t + u * nt;
            if (u * nt < PIECES && e < nc * (PROJ_K / 4)) {
                uint32_t* d = st + (e / (PROJ_K / 4)) * PROJ_STRIDE + (e % (PROJ_K / 4)) * 4;
                d[0] = v[u].x;
                d[1] = v[u].y;
                d[2] = v[u].z;
                d[3] = v[u].w;
            }
        }
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
            // every round finishes at least the first open lane, so 64 rounds always
            // suffice; the bound makes the exit unconditional and keeps the round
            // number inside its 7-bit field of the claim key
            static_assert(PROJ_CHUNK + 1 < 128, "the claim key holds the round number in 7 bits");
            while (__ballot(!done) && round < 127) {
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
                // entered in a hash table with atomicMax on (round, 63 - lane, point)
                // -- 7 + 6 + 19 bits: a chunk can take 64 rounds (one lane finishing per
                // round), positions stay below SLAMFE_PROJ_MAX_POINTS = 2^19 --:
                // the current round beats stale entries, the lowest lane wins its
                // slot, and a lane that does not find its own entry stops.  Two
                // different points in one slot stop the later lane needlessly, which
                // only shortens this round's prefix.
                ++round; // 1 .. 64 (7 bits in the claim key)
                bool stop = dry;
                if (prop >= 0) {
                    const uint32_t mine = (round << 25) | ((uint32_t)(63 - lane) << 19) | (uint32_t)prop;
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
                    // inlier k has to look at the points around it again (the 3 x 3 cells
                    // of its keypoint): the whole wave does it
                    const int i = c0 + k;
                    const int idx = kp_idx[i];
                    uint32_t best = 0xffffffffu;
                    if (idx >= 0 && idx < nkp) {
                        const float kx = kps[idx].x, ky = kps[idx].y;
                        const uint4* qp = reinterpret_cast<const uint4*>(fdesc + (size_t)idx * 8);
                        const uint4 q0 = qp[0], q1 = qp[1];
                        const proj_window w = proj_window_of(kx, ky, G, cell_start);
                        for (uint32_t t = (uint32_t)lane; t < w.total; t += 64) {
                            const uint4 e = sorted[proj_window_at(w, t)];
                            const uint32_t j = e.z;
                            if ((vt[j >> 5] >> (j & 31)) & 1u) continue;
                            if (!proj_gate(__uint_as_float(e.x), __uint_as_float(e.y), kx, ky, gate_dd)) continue;
                            const int h = proj_hamming(map_desc + (size_t)e.w * 8, q0, q1);
                            if (h <= max_hamming) best = min(best, ((uint32_t)h << 20) | j);
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

// Cell width and grid size for a search radius on frames of at most max_w x max_h
// pixels: cells a little wider than the radius (the margin of 1e-4 dwarfs the
// float rounding of the gate and of the cell division), at most PROJ_GRID_MAX
// cells per axis.
static slamfe_proj_grid proj_grid_for(float radius, int max_w, int max_h)
{
    double r = radius > 0.f ? (double)radius : 0.0; // negative / NaN radius: nothing passes anyway
    const double maxdim = (double)(max_w > max_h ? max_w : max_h);
    double cell = r * 1.0001 + 1e-3;
    if (cell < maxdim / PROJ_GRID_MAX) cell = maxdim / PROJ_GRID_MAX;
    slamfe_proj_grid G;
    G.inv_cell = 1.0 / cell; // 0 for an infinite radius: one cell
    auto cells = [&](int dim) {
        double n = floor((double)dim * G.inv_cell) + 1.0;
        return (int)(n < 1.0 ? 1.0 : (n > PROJ_GRID_MAX ? (double)PROJ_GRID_MAX : n));
    };
    G.gx = cells(max_w);
    G.gy = cells(max_h);
    return G;
}

size_t slamfe_projection_grid_bytes(void) { return ((size_t)PROJ_GRID_MAX * PROJ_GRID_MAX + 1) * sizeof(uint32_t); }

// Pass 0 of the search: the grid over the projected points (uv, point_ids in device
// memory) for a search radius.  There are R points, or min(R, *d_R) if d_R is not
// null (a count that is known on the device only: the grid can then be queued
// behind the kernels that make the points, without a host round trip).  cell_cnt
// (as many bytes as cell_start) and rank (R words) are used only where R exceeds
// what the one-workgroup kernel holds.
int slamfe_launch_projection_grid(hipStream_t s, const float* uv, const int32_t* point_ids, int R,
                                  const int32_t* d_R, float radius, int max_w, int max_h,
                                  uint32_t* cell_start, void* sorted, uint32_t* cell_cnt, uint32_t* rank)
{
    if (R <= 0) return 0;
    const slamfe_proj_grid G = proj_grid_for(radius, max_w, max_h);
    const int ncells = G.gx * G.gy;
    if (R <= PROJ_GRID_ONE_WG) {
        hipLaunchKernelGGL(k_proj_grid, dim3(1), dim3(1024), ((size_t)ncells + 16) * sizeof(uint32_t), s,
                           reinterpret_cast<const float2*>(uv), point_ids, R, d_R, G, cell_start,
                           reinterpret_cast<uint4*>(sorted));
        LAUNCH_CHECK();
        return 0;
    }
    hipError_t e = hipMemsetAsync(cell_cnt, 0, (size_t)ncells * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_proj_grid_count, dim3((R + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(uv), R, d_R, G, cell_cnt, rank);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_proj_grid_scan, dim3(1), dim3(1024), 0, s, cell_cnt, ncells, cell_start);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_proj_grid_scatter, dim3((R + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(uv), point_ids, R, d_R, G, cell_start, rank,
                       reinterpret_cast<uint4*>(sorted));
    LAUNCH_CHECK();
    return 0;
}

// Passes 1 and 2 over a grid built by slamfe_launch_projection_grid for the same
// radius and frame-size limits.
int slamfe_launch_projection(hipStream_t s, const slamfe_keypoint* kps, const uint8_t* fdesc, int nkp,
                             const int32_t* kp_idx, int n_inl, const int32_t* point_ids, int R,
                             const uint8_t* map_desc, float radius, int max_hamming, int max_w, int max_h,
                             const uint32_t* cell_start, const void* sorted, uint32_t* cand, int32_t* cand_n,
                             int32_t* matched)
{
    if (n_inl <= 0) return 0;
    const double gate_dd = slamfe_proj_gate_threshold(radius);
    if (max_hamming > 256) max_hamming = 256; // no two rows are further apart
    const slamfe_proj_grid G = proj_grid_for(radius, max_w, max_h);
    hipLaunchKernelGGL(k_proj_candidates, dim3((n_inl + 3) / 4), dim3(256), 0, s, kps,
                       reinterpret_cast<const uint32_t*>(fdesc), nkp, kp_idx, n_inl, G, cell_start,
                       reinterpret_cast<const uint4*>(sorted), reinterpret_cast<const uint32_t*>(map_desc),
                       gate_dd, max_hamming, cand, cand_n);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(256),
                       (2 * (PROJ_CHUNK * PROJ_STRIDE + PROJ_CHUNK) + PROJ_CHUNK + 2 * PROJ_CLAIM) * 4 +
                           ((size_t)R + 31) / 32 * 4 + 4,
                       s, kps,
                       reinterpret_cast<const uint32_t*>(fdesc), nkp, kp_idx, n_inl, point_ids, R, G,
                       cell_start, reinterpret_cast<const uint4*>(sorted),
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
    // the grid kernel's cell counters: up to PROJ_GRID_MAX^2 + 16 words
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_grid),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (PROJ_GRID_MAX * PROJ_GRID_MAX + 16) * 4);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_resolve),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SLAMFE_PROJ_MAX_POINTS / 8 +
                                        (2 * (PROJ_CHUNK * PROJ_STRIDE + PROJ_CHUNK) + PROJ_CHUNK +
                                         2 * PROJ_CLAIM) * 4 + 64);
}

// Raise the dynamic-LDS limits of the two kernels that may need more than the
// default 64 KiB.
int slamfe_select_init(size_t sort_lds_bytes);
int slamfe_image_init(void);
int slamfe_kernels_init(size_t sort_lds_bytes)
{
    int e = slamfe_select_init(sort_lds_bytes);
    if (e) return e;
    if ((e = slamfe_image_init())) return e;
    return slamfe_projection_init();
}
