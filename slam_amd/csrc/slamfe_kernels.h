/* slamfe_kernels.h -- launchers of the HIP kernels (internal to libslamfe). */
#ifndef SLAMFE_KERNELS_H
#define SLAMFE_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slamfe.h"
#include "slamfe_dev.h"

/* Every launcher returns 0 or a hipError_t value. */
int slamfe_launch_ingest(hipStream_t s, const uint8_t* d_in, int w, int h, int stride,
                         size_t frame_stride, int channels, int nframes, uint8_t* pyr,
                         uint32_t frame_bytes, int pitch);
int slamfe_launch_ingest_scaled(hipStream_t s, const uint8_t* d_in, int stride, const uint32_t* xt,
                                const uint32_t* yt, int dw, int dh, uint8_t* gray, int gpitch,
                                uint8_t* bgr_out, int bpitch);
/* Level 0 of a batch may live outside the pyramid buffer (the caller's gray frames,
 * used in place when their layout is that of level 0): frame f's level 0 starts at
 * l0.base + f * l0.frame_stride, rows of lv[0].pitch bytes.  Inside the pyramid:
 * base = pyr + lv[0].off, frame_stride = frame_bytes. */
typedef struct slamfe_level0 {
    const uint8_t* base;
    size_t frame_stride;
} slamfe_level0;

/* Where a call of a few frames wants its results besides the frame slots: pinned
 * host arrays laid out [frame of the call][max_kp rows] (all null: nowhere). */
typedef struct slamfe_host_rows {
    int32_t* cnt;
    uint8_t* kps;
    uint8_t* desc;
} slamfe_host_rows;
int slamfe_launch_pyramid(hipStream_t s, const slamfe_dev_geom& g, int nframes, uint8_t* pyr,
                          const uint32_t* taps, slamfe_level0 l0);
/* Ingest + all pyramid levels in one launch (a few frames per call); d_in = NULL:
 * level 0 is already in pyr.  Needs g.pt_x > 0. */
int slamfe_launch_pyramid_fused(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* d_in,
                                int stride, size_t frame_stride, int channels, uint8_t* pyr,
                                const uint32_t* taps);
int slamfe_launch_fast_blur(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                            uint8_t* blur, int thr, int edge, uint32_t* cand, uint32_t* cand_cnt,
                            slamfe_level0 l0);
int slamfe_launch_select(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint32_t* cand,
                         const uint32_t* cand_cnt, uint32_t* band_hist, uint32_t* band_off,
                         uint32_t* level_kept, int32_t* level_cutoff, int slot_base, int nslots,
                         uint32_t* kp_key, uint32_t* kp_level, int32_t* kp_count, int max_kp,
                         size_t sort_lds_bytes, bool few /* one-launch plan kernel */);
int slamfe_launch_orient(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                         int slot_base, int nslots, const uint32_t* kp_key, const uint32_t* kp_level,
                         const int32_t* kp_count, int32_t* kp_mom, slamfe_keypoint* kps, float* kp_rot,
                         int max_kp, int mode, int patch_size, slamfe_level0 l0);
int slamfe_launch_describe(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* blur,
                           int slot_base, int nslots, const uint32_t* kp_key,
                           const uint32_t* kp_level, const int32_t* kp_count, const float* kp_rot,
                           uint8_t* desc, int max_kp);

/* k_orient + k_finish + k_describe as one launch (a few frames per call). */
int slamfe_launch_orient_describe(hipStream_t s, const slamfe_dev_geom& g, int nframes, const uint8_t* pyr,
                                  const uint8_t* blur, int slot_base, int nslots, const uint32_t* kp_key,
                                  const uint32_t* kp_level, const int32_t* kp_count, int32_t* kp_mom,
                                  slamfe_keypoint* kps, float* kp_rot, uint8_t* desc, int max_kp, int mode,
                                  int patch_size, slamfe_host_rows host, slamfe_level0 l0);

/* Counts and (with_rows) keypoint records + descriptor rows of nframes consecutive
 * slots into host-pinned arrays laid out [frame][max_kp]. */
int slamfe_launch_export(hipStream_t s, const int32_t* kp_count, const slamfe_keypoint* kps, const uint8_t* desc,
                         int slot_base, int nslots, int max_kp, int nframes, int with_rows, int32_t* h_cnt,
                         uint8_t* h_kps, uint8_t* h_desc);

/* Choose queries per lane and the train chunking for a match of at most
 * nq_max x nt_max rows (npairs independent pairs in one launch); the partial
 * buffer needs npairs * nchunks * cap * 8 bytes. */
int slamfe_match_plan(int nq_max, int nt_max, int npairs, int kernel, int* qpt, int* chunk_rows,
                      int* nchunks);
int slamfe_launch_match(hipStream_t s, const uint8_t* d_query, int nq, const int32_t* d_nq,
                        const uint8_t* d_train, int nt, const int32_t* d_nt, int cap, int qpt,
                        int chunk_rows, int nchunks, uint32_t* part, float ratio,
                        slamfe_match2* d_out, uint64_t* d_keys, int64_t train_base, int npairs,
                        int q_slot_base, int t_slot_base, int nslots);
int slamfe_launch_merge_shard_keys(hipStream_t s, const uint64_t* d_keys, int nshards, int nq,
                                   float ratio, slamfe_match2* d_out);

#define SLAMFE_PROJ_K 32
#define SLAMFE_PROJ_MAX_POINTS (1 << 19)
// the grid over the projected points (pass 0 of the search); cell_cnt and rank are
// needed only above SLAMFE_PROJ_ONE_WG_POINTS points
#define SLAMFE_PROJ_ONE_WG_POINTS 20480
int slamfe_launch_projection_grid(hipStream_t s, const float* uv, const int32_t* point_ids, int R,
                                  const int32_t* d_R /* or null: min(R, *d_R) points */,
                                  float radius, int max_w, int max_h, uint32_t* cell_start /* grid_bytes */,
                                  void* sorted /* R * 16 bytes */, uint32_t* cell_cnt /* grid_bytes */,
                                  uint32_t* rank /* R * 4 bytes */);
// candidates + resolve over that grid (same radius, max_w, max_h)
int slamfe_launch_projection(hipStream_t s, const slamfe_keypoint* kps, const uint8_t* fdesc, int nkp,
                             const int32_t* kp_idx, int n_inl, const int32_t* point_ids, int R,
                             const uint8_t* map_desc, float radius, int max_hamming, int max_w, int max_h,
                             const uint32_t* cell_start, const void* sorted, uint32_t* cand, int32_t* cand_n,
                             int32_t* matched);
size_t slamfe_projection_grid_bytes(void);
int slamfe_launch_reproject(hipStream_t s, const float* xyz, int npoints, const float* M16, float th,
                            float W, float H, float* uv_all, uint64_t* wave_mask, int32_t* wave_off,
                            int32_t* total, float* vis_uv, int32_t* vis_idx);
int slamfe_launch_triangulate(hipStream_t s, const float* P1, const float* P2, const float* d_pts1,
                              const float* d_pts2, int n, float* d_xyz);
int slamfe_kernels_init(size_t sort_lds_bytes);

#endif
