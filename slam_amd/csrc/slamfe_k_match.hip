// slamfe_k_match.hip -- brute-force Hamming k=2: matrix-core and vector-ALU kernels, merges
#include "slamfe_device.h"

// ---------------------------------------------------------------------------
// K8  brute-force Hamming k=2 (BFMatcher::knnMatch, Init.cc:131) + ratio test
//     (Init.cc:146-147).
//
// Two kernels give identical results: k_match_mfma (matrix cores, further down)
// and k_match (vector ALU): a lane owns QPT query rows in registers (8 dwords
// each; QPT = 1 is what the plan uses), train rows are staged through LDS, one
// comparison is 8 v_xor + 8 accumulating v_bcnt; the running top-2 is kept as
// packed keys (distance << 23 | train row), for which integer min *is* the
// (distance, index) order of knnMatch.  The train set is cut into chunks;
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

// ---------------------------------------------------------------------------
// K8m  the same brute-force Hamming k=2 on the matrix cores, in FP4.
//
// With train bits a in {0, 1} and query bits mapped to b' = 1 - 2b in {+1, -1},
//   sum_k a_k * b'_k = popcount(a) - 2 a.b = hamming(a, b) - popcount(b),
// so a chain of matrix instructions over the 256 bit positions gives the exact
// distances of 32 train rows x 32 queries, minus a per-query constant that does
// not change the order.  gfx950's v_mfma_f32_32x32x64_f8f6f4 takes 4-bit floats
// (E2M1: 0, 0.5 .. 6) and covers 64 bit positions per instruction in the time the
// 8-bit integer instruction covers 32: a bit is a nibble, train bits are stored as
// {0, +4} and query bits as {+4, -4}, so the chain (4 instructions) delivers
// 16 * (distance - popcount(b)) in f32, exactly (every partial sum is a multiple
// of 16 below 2^14).
//
// The chain's initial accumulator holds 16 * 256 + (bias + row in tile) / 1024
// per result register, so the result IS the key: distance in the integer part,
// row in the fraction (23 significant bits, all exact in f32).  Keys are positive
// floats, whose order is the order of their bit patterns: the running top-2 is
// kept with the integer v_min_u32 / v_med3_u32 on the raw registers.  In the
// result tile the query is the lane (column) and the 16 registers are train rows,
// so the top-2 of a query stays in its lane: 1.5 vector operations per comparison
// (v_min3, v_med3, v_min per two keys) instead of 19.  Row numbers inside the keys are relative to the current tile
// (older entries are moved down by 32 rows = 1/32 per tile, one v_sub_f32 each,
// instead of re-basing 16 constants) and made absolute at the end.
//
// A workgroup = 256 queries (4 waves x 2 blocks of 32, expanded to nibbles in 32
// VGPRs per lane) x one chunk of train rows.  Train rows are read packed (32
// bytes), expanded to nibbles in LDS (one dword -> 16 bytes per thread and tile
// through a 256-entry byte -> 8 nibbles table, rows padded to 144 bytes so that
// the 16-byte fragment reads are conflict-free), in three buffers of one stage
// (two tiles) each.  A stage is one block of straight-line code: the fragment
// reads of both tiles are issued first, then the chains of tile t run while the
// vector ALU updates the top-2 from tile t-1 (two accumulator sets) and expands
// the next stage's rows.
//
// Measured on MI355X (tools/ubench_mfma_fp4.hip, tools/time_match.py,
// tools/match_stamps.py): the bare chain runs at 8.1 P bit-operations/s, and with
// the vector work of the top-2 update beside it (about 10 operations per matrix
// instruction) at 5.4: this kernel is bound by vector issue, not by the matrix
// pipe.  The 8-bit integer form it replaces (8 instructions per chain, 4.9 P
// bare) was bound by the matrix pipe and ran 2000 x 1M in 0.31 ms.
// ---------------------------------------------------------------------------
#define MM_TILE 32
#define MM_SUB 2 /* tiles per LDS stage (one workgroup barrier per stage) */
#define MM_ROW 144
#define MM_BIAS 8128u /* >= 32 * (tiles per chunk - 1), < 8192 - 31: chunks of at most 8160 rows */
#define MM_FRAC 1024.0f /* key = 16 * (distance + 256 - popcount(query)) + (MM_BIAS + row) / MM_FRAC */
#define MM_NONE 0x7f800000u /* +inf: "no key"; stays +inf under the per-tile subtraction */
typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef float mm_v16f __attribute__((ext_vector_type(16)));

// Median of three, which the compiler turns into one v_med3_u32.  (Not inline
// assembly: the compiler's hazard handling does not see a read of an MFMA result
// inside an asm statement, and such a read issued too soon after the chain
// returns the accumulator without the last products.)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    return max(min(a, b), min(max(a, b), c));
}

// the 8 bits of a byte -> 8 nibbles of 0 / 1 (bit i in nibble i)
__device__ __forceinline__ uint32_t bits8_to_nibbles(uint32_t x)
{
    uint32_t t = (x | (x << 12)) & 0x000f000fu;
    t = (t | (t << 6)) & 0x03030303u;
    return (t | (t << 3)) & 0x11111111u;
}

// Where the finished match rows / shard keys go when a workgroup sees the whole
// train set (one chunk): the kernel then emits them itself and no merge runs.
struct slamfe_match_direct {
    int on;
    float ratio;
    slamfe_match2* out;
    uint64_t* keys;
    int64_t train_base;
};

#ifdef SLAMFE_STAMPS
// Diagnostic build only: shader-clock and wall-clock stamps of wave 0 of 64 sampled
// workgroups of k_match_mfma (kernel entry, stage loop entry and exit, kernel
// exit; then the stage count), read back with slamfe_debug_match_stamps().
__device__ unsigned long long g_mm_stamps[64 * 16];
#define MM_STAMP(i)                                                                              \
    do {                                                                                         \
        if ((blockIdx.x & 31u) == 7u && (blockIdx.x >> 5) < 64u && threadIdx.x == 0) {            \
            g_mm_stamps[(blockIdx.x >> 5) * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();        \
            g_mm_stamps[(blockIdx.x >> 5) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                        \
    } while (0)
extern "C" int slamfe_debug_match_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mm_stamps), (size_t)n * 8);
}
// ... and of every workgroup (up to 4096): wall-clock entry and exit, the
// hardware id register and the XCC id (where it ran)
__device__ unsigned long long g_mm_where[4096 * 4];
extern "C" int slamfe_debug_match_where(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mm_where), (size_t)n * 8);
}
#else
#define MM_STAMP(i)
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_match_mfma(const uint32_t* __restrict__ query, int nq_imm,
                                                    const int32_t* __restrict__ nq_dev,
                                                    const uint32_t* __restrict__ train, int nt_imm,
                                                    const int32_t* __restrict__ nt_dev,
                                                    uint32_t* __restrict__ part, slamfe_pair_slots ps,
                                                    slamfe_match_geo mg, slamfe_match_direct direct)
{
    __shared__ __attribute__((aligned(16))) uint8_t sA[3][MM_SUB][MM_TILE * MM_ROW];
    MM_STAMP(0);
#ifdef SLAMFE_STAMPS
    if (blockIdx.x < 4096u && threadIdx.x == 0) {
        g_mm_where[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        g_mm_where[blockIdx.x * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_mm_where[blockIdx.x * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    const int pair = (int)(blockIdx.x % (unsigned)mg.npairs);
    const int local = (int)(blockIdx.x / (unsigned)mg.npairs);
    if (ps.nslots) {
        int sq = (ps.q_base + pair) % ps.nslots, st = (ps.t_base + pair) % ps.nslots;
        query += (size_t)sq * mg.cap * 8;
        train += (size_t)st * mg.cap * 8;
        nq_dev += sq;
        nt_dev += st;
        part += (size_t)pair * mg.nchunks * mg.cap * 2;
        if (direct.out) direct.out += (size_t)pair * mg.cap;
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
            if (h == 0 && qi < nq) {
                if (direct.on)
                    match_emit(SLAMFE_MKEY_NONE, SLAMFE_MKEY_NONE, qi, direct.ratio, direct.out, direct.keys, direct.train_base);
                else
                    reinterpret_cast<uint2*>(part)[(size_t)chunk * mg.part_sc + (size_t)qi * mg.part_sq] =
                        make_uint2(SLAMFE_MKEY_NONE, SLAMFE_MKEY_NONE);
            }
        }
        return;
    }

    // byte -> eight nibbles of 0 / 6 (FP4 +4.0): the expansion of the train rows is a
    // table lookup (4 LDS reads per dword instead of two dozen vector operations)
    __shared__ uint32_t lut[256];
    lut[tid] = bits8_to_nibbles((uint32_t)tid) * 6u;
    // ... and byte -> eight nibbles of 6 (+4.0, bit clear) / 0xe (-4.0, bit set) for
    // the query rows: 32 table reads per lane instead of a few hundred vector
    // operations at the start of every workgroup
    __shared__ uint32_t qlut[256];
    qlut[tid] = 0x66666666u | (bits8_to_nibbles((uint32_t)tid) << 3);
    __syncthreads();

    // query fragments: the instruction's 64 bit positions are two dwords of the
    // row, one per lane half; lane (r, h) holds dword 2 kb + h of its query row
    // as 32 nibbles of 6 (+4.0, bit clear) / 0xe (-4.0, bit set).  (Which bit sits
    // in which nibble does not matter as long as train rows use the same map.)
    mm_v4i B[2][4];
    int pq[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int qi = min(qfirst + 32 * blk, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t qw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int pc = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) pc += __popc(qw[w]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const uint32_t mine = h ? qw[2 * kb + 1] : qw[2 * kb];
#pragma unroll
            for (int n = 0; n < 4; ++n) B[blk][kb][n] = (int)qlut[(mine >> (8 * n)) & 0xffu];
        }
        pq[blk] = pc;
    }

    // a stage = MM_SUB tiles; the thread's piece of tile u of a stage: dword
    // (tid & 7) of row (tid >> 3)
    auto load_bits = [&](int st, int u) -> uint32_t {
        // (no branch and no select: past the chunk the row is clamped, and what the
        // chains make of such a row is never looked at -- the top-2 update masks the
        // rows beyond a short last tile and a missing tile is not computed.  The value
        // is first used a stage later, so nothing in this stage waits for the load.)
        const int row = j0 + (st * MM_SUB + u) * MM_TILE + (tid >> 3);
        return train[(size_t)min(row, j1 - 1) * 8 + (tid & 7)];
    };
    auto store_tile = [&](int buf, int u, uint32_t bits) {
        const uint4 e = make_uint4(lut[bits & 0xffu], lut[(bits >> 8) & 0xffu], lut[(bits >> 16) & 0xffu],
                                   lut[bits >> 24]);
        *reinterpret_cast<uint4*>(&sA[buf][u][(tid >> 3) * MM_ROW + (tid & 7) * 16]) = e;
    };
    const int ntiles = (j1 - j0 + MM_TILE - 1) / MM_TILE;
    const int nstages = (ntiles + MM_SUB - 1) / MM_SUB;
    // Three stage buffers: the rows of stage s+1 are fetched from memory during
    // stage s-1 and written to LDS in the middle of stage s, so that a wave reaches
    // the barrier at the end of a stage with nothing of its own pending.
    uint32_t nbits[MM_SUB];
#pragma unroll
    for (int u = 0; u < MM_SUB; ++u) store_tile(0, u, load_bits(0, u));
#pragma unroll
    for (int u = 0; u < MM_SUB; ++u) nbits[u] = load_bits(1, u);
    __syncthreads();

    // result register j of lane half h is train row (j & 3) + 8 (j >> 2) + 4 h of the tile
    // and the accumulators start from 16 * 256 + (MM_BIAS + that row) / MM_FRAC, so
    // that the chain delivers the finished keys
    mm_v16f cj;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        cj[j] = 4096.0f + (float)(MM_BIAS + (uint32_t)((j & 3) + 8 * (j >> 2) + 4 * h)) * (1.0f / MM_FRAC);
    uint32_t b1[2] = {MM_NONE, MM_NONE}, b2[2] = {MM_NONE, MM_NONE};

    // One tile's fragment reads (4 x 16 bytes per lane) and chains (two query
    // blocks).  The reads of a tile are issued a whole tile ahead of the chains
    // that consume them, so their LDS latency hides under the previous chains.
    auto fetch = [&](int buf, int u, mm_v4i* a) {
        const uint8_t* base = &sA[buf][u][r * MM_ROW + h * 16];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) a[kb] = *reinterpret_cast<const mm_v4i*>(base + kb * 32);
    };
    // (FP4 operands are 4 registers; the builtin's type has 8, the upper 4 are not read)
#define MM_WIDE(x) __builtin_shufflevector(x, x, 0, 1, 2, 3, -1, -1, -1, -1)
    auto compute = [&](const mm_v4i* a, mm_v16f& acc0, mm_v16f& acc1) {
        acc0 = cj;
        acc1 = cj;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(MM_WIDE(a[kb]), MM_WIDE(B[0][kb]), acc0, 4, 4, 0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(MM_WIDE(a[kb]), MM_WIDE(B[1][kb]), acc1, 4, 4, 0, 0, 0, 0);
        }
    };
    // ... and the top-2 update from a finished tile's keys.  The two are issued
    // for different tiles: the update of tile t runs on the vector ALU while the
    // matrix core works on tile t+1 (two accumulator sets).
    auto down = [](uint32_t k) { return __float_as_uint(__uint_as_float(k) - (float)MM_TILE * (1.0f / MM_FRAC)); };
    auto extract = [&](const mm_v16f& acc0, const mm_v16f& acc1, int rows) {
        // entries of earlier tiles move down by one tile
        b1[0] = down(b1[0]); b2[0] = down(b2[0]); b1[1] = down(b1[1]); b2[1] = down(b2[1]);
        // Two keys x, y at a time: with b1 <= b2 the second smallest of (x, y, b1, b2)
        // is min(med3(x, y, b1), b2) and the smallest min3(x, y, b1): 3 operations
        // per pair of comparisons.
        // (min3 is associated differently from the min(x, y) inside the median so that
        // the two do not share a subexpression: sharing it keeps the compiler from
        // forming v_med3 and v_min3.)
        auto push2 = [](uint32_t x, uint32_t y, uint32_t& lo, uint32_t& hi) {
            const uint32_t m = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(x), __uint_as_float(y), __uint_as_float(lo)));
            lo = min(x, min(y, lo));
            hi = min(m, hi);
        };
        if (rows >= MM_TILE) { // uniform; only the last tile of a chunk can be short
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                push2(__float_as_uint(acc0[j]), __float_as_uint(acc0[j + 1]), b1[0], b2[0]);
                push2(__float_as_uint(acc1[j]), __float_as_uint(acc1[j + 1]), b1[1], b2[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const bool ok0 = (j & 3) + 8 * (j >> 2) + 4 * h < rows;
                const bool ok1 = ((j + 1) & 3) + 8 * ((j + 1) >> 2) + 4 * h < rows;
                push2(ok0 ? __float_as_uint(acc0[j]) : MM_NONE, ok1 ? __float_as_uint(acc0[j + 1]) : MM_NONE, b1[0], b2[0]);
                push2(ok0 ? __float_as_uint(acc1[j]) : MM_NONE, ok1 ? __float_as_uint(acc1[j + 1]) : MM_NONE, b1[1], b2[1]);
            }
        }
    };
    static_assert(MM_SUB == 2, "the stage loop below is written for two tiles per stage");
    const int last_rows = j1 - (j0 + (ntiles - 1) * MM_TILE); // valid rows of the last tile
    mm_v16f P0, P1, Q0, Q1; // even tiles accumulate in P, odd tiles in Q
#pragma unroll
    for (int j = 0; j < 16; ++j) Q0[j] = Q1[j] = __uint_as_float(MM_NONE);
    // All stages but the last are straight-line code (both tiles full, a next stage
    // to load): fragment reads, the two tiles' chains, the updates of the tiles
    // before them and the expansion of the next stage's rows share one block that
    // the compiler interleaves.  The last stage handles a short or missing tile.
    mm_v4i F[4], G[4];
    int cur = 0; // buffer of the stage being computed
    MM_STAMP(1);
    for (int st = 0; st + 1 < nstages; ++st) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        fetch(cur, 0, F);
        // the second tile's fragments are on their way while the first tile's chains run
        fetch(cur, 1, G);
        __builtin_amdgcn_sched_barrier(0); // keep the compiler from sinking the reads to their uses
        compute(F, P0, P1);
        // the next stage's rows (in registers since the previous stage) go to their
        // buffer, which was last read two stages ago; then the loads for the stage
        // after it start
#pragma unroll
        for (int u = 0; u < MM_SUB; ++u) store_tile(nxt, u, nbits[u]);
#pragma unroll
        for (int u = 0; u < MM_SUB; ++u) nbits[u] = load_bits(st + 2, u);
        // (in the first stage Q holds "no key" in every register and the update
        // changes nothing)
        extract(Q0, Q1, MM_TILE);
        // everything above is issued before the second tile's chains: the stores
        // are out in mid-stage, not in front of the barrier
        __builtin_amdgcn_sched_barrier(0);
        compute(G, Q0, Q1);
        extract(P0, P1, MM_TILE);
        __syncthreads();
        cur = nxt;
    }
    {
        const int st = nstages - 1;
        fetch(cur, 0, F);
        fetch(cur, 1, G);
        __builtin_amdgcn_sched_barrier(0);
        compute(F, P0, P1);
        extract(Q0, Q1, MM_TILE);
        if (st * MM_SUB + 1 < ntiles) { // uniform
            compute(G, Q0, Q1);
            extract(P0, P1, MM_TILE);
            extract(Q0, Q1, last_rows);
        } else {
            extract(P0, P1, last_rows);
        }
    }

    MM_STAMP(2);
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
                if (k >= MM_NONE) return SLAMFE_MKEY_NONE;
                // key * MM_FRAC is an integer below 2^24: (distance + 256 - popcount) << 14 | bias + row
                const uint32_t ki = (uint32_t)(__uint_as_float(k) * MM_FRAC);
                const uint32_t d = (ki >> 14) - 256u + (uint32_t)pq[blk];
                const uint32_t row = (ki & 16383u) - MM_BIAS + last + (uint32_t)j0;
                return (d << SLAMFE_MKEY_SHIFT) | row;
            };
            if (direct.on) // the only chunk: these are the query's final two
                match_emit(conv(m1), conv(m2), qi, direct.ratio, direct.out, direct.keys, direct.train_base);
            else
                reinterpret_cast<uint2*>(part)[(size_t)chunk * mg.part_sc + (size_t)qi * mg.part_sq] =
                    make_uint2(conv(m1), conv(m2));
        }
    }
    MM_STAMP(3);
#ifdef SLAMFE_STAMPS
    if ((blockIdx.x & 31u) == 7u && (blockIdx.x >> 5) < 64u && threadIdx.x == 0)
        g_mm_stamps[(blockIdx.x >> 5) * 16 + 8] = (unsigned long long)nstages;
    if (blockIdx.x < 4096u && threadIdx.x == 0) g_mm_where[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
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
        // cost of three tiles) and its stages hand over through a barrier, so chunks
        // are long: 4096 rows for frame pairs, shorter only when that would leave
        // fewer than ~512 workgroups (a single frame pair).  Measured on MI355X:
        // 256 frame pairs 0.230 / 0.197 / 0.179 / 0.177 ms with 512 / 1024 / 2048 /
        // 4096 rows.
        // A large train set is cut into about 4096-row chunks, their number chosen so
        // that the workgroups fill whole rounds of the chip's 512 places (two
        // workgroups per CU): 2000 x 1M ran 6.13 T cmp/s with 8128-row chunks (992
        // workgroups: the last round is a third empty, and the two workgroups of a CU
        // do not run at the same speed), 6.33-6.39 with 3904-3936 rows (4 rounds),
        // 5.9 with 992 rows (the query expansion weighs more).
        q = 64;
        if (nt_max > 8192) {
            const long long places = 512;
            const long long rounds = (have * ((nt_max + 4095) / 4096) + places - 1) / places;
            long long nc_want = rounds * places / have;
            if (nc_want < 1) nc_want = 1;
            long long r = ((nt_max + nc_want - 1) / nc_want + 31) / 32 * 32;
            rows = (int)(r < 256 ? 256 : (r > 8128 ? 8128 : r));
        } else {
            static const int lengths[] = {4096, 2048, 1024, 512, 256};
            rows = 256;
            for (int r : lengths) {
                rows = r;
                if (have * ((nt_max + r - 1) / r) >= 512) break;
            }
        }
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
    // the matrix-core kernel keeps tile-relative row numbers in the low MM_SHIFT
    // bits of its keys: a chunk may not span more tiles than MM_BIAS allows
    if (qpt == 64 && ((chunk_rows + MM_TILE - 1) / MM_TILE - 1) * MM_TILE > (int)MM_BIAS)
        return (int)hipErrorInvalidValue;
    // one chunk (frame pairs): the matrix-core kernel writes the rows itself
    slamfe_match_direct direct = {qpt == 64 && nchunks == 1, ratio, d_out, d_keys, train_base};
    if (qpt == 64)
        hipLaunchKernelGGL(k_match_mfma, grid, dim3(256), 0, s, q32, nq, d_nq, t32, nt, d_nt, part, ps, mg, direct);
    else if (qpt == 1)
        hipLaunchKernelGGL(k_match<1>, grid, dim3(256), 0, s, q32, nq, d_nq, t32, nt, d_nt, part, ps, mg);
    else
        return (int)hipErrorInvalidValue;
    LAUNCH_CHECK();
    if (direct.on) return 0;
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
