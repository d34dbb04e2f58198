/* slamfe.h -- C ABI of the MI355X feature front-end (libslamfe.so).
 *
 * The library replaces the four OpenCV call sites of the reference's tracking
 * thread and nothing else:
 *
 *   Init.cc:64-66   cv::FastFeatureDetector::create / cv::ORB::create /
 *                   new cv::BFMatcher            -> slamfe_create
 *   Init.cc:75-78   cvtColor + detect + compute  -> slamfe_extract
 *   Init.cc:131     knnMatch(desc1, desc2, 2)    -> slamfe_match_knn2
 *   Init.cc:144-153 Lowe ratio loop              -> slamfe_match2.ratio_pass
 *   Init.cc:229-259 search by projection         -> slamfe_search_by_projection
 *   Init.cc:296     Point(..., desc2.row(idx))   -> slamfe_map_put
 *
 * Conventions
 *   - Plain C, no exceptions: every call returns SLAMFE_OK (0) or a negative
 *     SLAMFE_E_* code; slamfe_last_error() gives the text for the last failure
 *     on that context.  (The reference lets cv::Exception terminate the
 *     process; a host that wants that behaviour aborts on a non-zero return.)
 *   - Outputs are caller-allocated; the library copies (it never aliases a
 *     caller buffer after the call returns, unlike the shared cv::Mat buffers
 *     of Init.cc:296 / Init.cc:323).
 *   - A context owns one GPU, one HIP stream and all device buffers.  It is not
 *     re-entrant: use it from one host thread at a time (the reference calls
 *     the front-end only from the processFrames thread, Init.cc:83).
 *   - Pointers named d_* are device pointers on the context's GPU; all others
 *     are host pointers.
 *   - Stream ordering.  Every call runs on the context's stream (params.stream,
 *     or a stream the context creates).  Calls that take only host pointers are
 *     synchronous: they return after their results have been copied out.  Calls
 *     documented as asynchronous only queue work on that stream.  For every d_*
 *     argument the caller orders its own accesses against that stream:
 *       inputs  (d_imgs, d_query, d_train, d_keys_all): what the caller wrote
 *               must be complete before the call is made -- written on the
 *               context's stream itself, or followed by a synchronisation of the
 *               writing stream (or an event the context's stream waits on) -- and
 *               must not be overwritten until the queued work has run;
 *       outputs (d_out, d_keys): must not be touched by other streams from the
 *               call until the queued work has run (a fill of d_out queued on
 *               another stream and not waited for races with the library's
 *               writes); they may be read after slamfe_sync, or by work queued
 *               later on the context's stream.
 *     The library never synchronises with other streams on its own.
 *   - Keypoint order and capacity.  In SLAMFE_MODE_ORB keypoints come in the
 *     canonical order (level, y, x); cv::ORB's own order after retainBest is
 *     implementation-defined (std::nth_element), so only the keypoint *set* is
 *     comparable with an OpenCV build, and train indices / tie-breaks downstream
 *     follow the canonical order.  SLAMFE_MODE_PROVIDED keeps FAST's raster order,
 *     as the reference's calls do.  retainBest keeps every tie at the cut-off
 *     score, so a frame can yield more than max_keypoints keypoints: the
 *     host-pointer calls then return SLAMFE_E_CAPACITY; on the resident path the
 *     frame is truncated to its first max_keypoints keypoints in that order (the
 *     device-side count keeps the untruncated number, see
 *     slamfe_frame_device_ptrs) and the match calls use the truncated frame.
 */
#ifndef SLAMFE_H
#define SLAMFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMFE_OK 0
#define SLAMFE_E_INVALID (-1)  /* bad argument */
#define SLAMFE_E_CAPACITY (-2) /* caller or context capacity too small; see call */
#define SLAMFE_E_DEVICE (-3)   /* HIP runtime error */
#define SLAMFE_E_NOMEM (-4)
#define SLAMFE_E_HANDLE (-5)   /* frame handle expired or unknown */

/* Operating mode of slamfe_extract.
 * SLAMFE_MODE_ORB       full ORB: image pyramid, per-level FAST + retain-best
 *                       quota, intensity-centroid orientation, blur, rBRIEF
 *                       (what cv::ORB::detectAndCompute does).
 * SLAMFE_MODE_PROVIDED  what the reference literally runs (Init.cc:76 then
 *                       Init.cc:78): FAST on the full-resolution image, then
 *                       ORB::compute on those keypoints -- one level, no quota,
 *                       angle fixed at -1 degree, 31-px border filter. */
#define SLAMFE_MODE_ORB 0
#define SLAMFE_MODE_PROVIDED 1

/* The two brute-force Hamming kernels behind knnMatch (Init.cc:131); results are
 * identical.  POPCOUNT: v_xor / v_bcnt on the vector ALU.  MATRIX: the bit
 * vectors expanded to 4-bit floating-point operands (a bit is a nibble) and
 * multiplied on the matrix cores, exactly; about four times as fast as POPCOUNT
 * on large jobs.  AUTO: MATRIX from 128 queries up, else POPCOUNT. */
#define SLAMFE_MATCH_AUTO 0
#define SLAMFE_MATCH_POPCOUNT 1
#define SLAMFE_MATCH_MATRIX 2

/* Layout-identical to cv::KeyPoint (28 bytes).  Consumers in the reference
 * read only pt (Init.cc:150-151, 241, 280-283; optimizer.cc:97-98). */
typedef struct slamfe_keypoint {
    float x, y;     /* pt, in full-resolution pixels */
    float size;     /* 7 (FAST) or 31 * level scale (ORB) */
    float angle;    /* degrees in [0,360), or -1 */
    float response; /* FAST score */
    int32_t octave; /* pyramid level */
    int32_t class_id; /* -1 */
} slamfe_keypoint;

/* One row of a k=2 match result: what knnMatch (Init.cc:131) returns for one
 * query plus the ratio test of Init.cc:146-147.  Order rule: (distance, train
 * index) ascending.  idx = -1 where fewer than 1 / 2 train rows exist. */
typedef struct slamfe_match2 {
    int32_t idx0, idx1;
    uint16_t d0, d1;     /* Hamming distances, 0..256 */
    uint8_t ratio_pass;  /* nfound == 2 && (float)d0 < ratio * (float)d1 */
    uint8_t nfound;      /* min(2, train rows) */
    uint8_t reserved[2];
} slamfe_match2;

typedef struct slamfe_params {
    int32_t struct_size;     /* sizeof(slamfe_params), set by slamfe_default_params */
    int32_t mode;            /* SLAMFE_MODE_* */
    int32_t fast_threshold;  /* thold: 30 in main.cc:11 */
    int32_t nfeatures;       /* 10000 in main.cc:11 (only used in SLAMFE_MODE_ORB) */
    float scale_factor;      /* 1.2f, Init.cc:65 */
    int32_t nlevels;         /* 8, Init.cc:65 */
    int32_t edge_threshold;  /* 31, Init.cc:65 */
    int32_t patch_size;      /* 31, Init.cc:65 (only 31 is implemented) */
    float ratio;             /* 0.75f, Init.cc:146 */
    int32_t device_id;       /* HIP device ordinal */
    int32_t max_width;       /* largest frame the context accepts (<= 4096) */
    int32_t max_height;
    int32_t max_batch;       /* frames per slamfe_extract_batch / _resident call */
    int32_t max_keypoints;   /* per-frame keypoint capacity on the device */
    int32_t frame_slots;     /* device-resident frame results kept (ring) */
    int32_t map_capacity;    /* rows of the device-resident map descriptor table */
    void* stream;            /* hipStream_t to run on, or NULL to create one */
    int32_t match_kernel;    /* SLAMFE_MATCH_*: which brute-force kernel knnMatch (Init.cc:131) uses */
    int32_t reserved0;       /* 0 */
} slamfe_params;

typedef struct slamfe_ctx slamfe_ctx;

/* Fill *p with the reference's constants (Init.cc:64-66, main.cc:11). */
int slamfe_default_params(slamfe_params* p);

/* Replaces the object construction of Init.cc:64-66.  On a device failure
 * (SLAMFE_E_DEVICE: no GPU, out of device memory) *out may still receive a
 * context: its only valid uses are slamfe_last_error and slamfe_destroy.  On
 * SLAMFE_E_INVALID *out is NULL. */
int slamfe_create(const slamfe_params* p, slamfe_ctx** out);
/* Releases what slamfe_create made (the reference never frees the objects of
 * Init.cc:64-66; its process simply ends). */
void slamfe_destroy(slamfe_ctx* ctx);
/* Text of the last failure on this context: what the cv::Exception thrown
 * from Init.cc:76 / 78 / 131 would have carried. */
const char* slamfe_last_error(const slamfe_ctx* ctx);

/* Replaces Init::extractKeyPoints (Init.cc:74-79).
 * img: h rows of `stride` bytes, 1 channel (gray) or 3 (BGR, converted as
 * cv::COLOR_BGR2GRAY does).  kps/desc: caller buffers for `cap` keypoints and
 * cap*32 descriptor bytes (row-major, as a continuous CV_8U Mat).
 * Keypoint order: (level, y, x) ascending; in SLAMFE_MODE_PROVIDED that is
 * FAST's own row-major order.  *frame_handle names the device-resident copy.
 * Returns SLAMFE_E_CAPACITY with *n_out = needed count if cap (or the
 * context's max_keypoints) is too small. */
int slamfe_extract(slamfe_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                   slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                   uint64_t* frame_handle);

/* Init.cc:75-78 as slamfe_extract, without the copy into caller arrays: *kps and
 * *desc point at the *n_out keypoint records and 32-byte descriptor rows in the
 * context's own page-locked result arrays (host memory, read-only for the
 * caller), valid until the next extraction or download call on this context.  A
 * host layer that fills containers of its own (std::vector<cv::KeyPoint>, a
 * cv::Mat of n rows) copies exactly n records and rows from there, instead of
 * offering arrays sized for the worst case.  Fails with SLAMFE_E_CAPACITY (and
 * *n_out = needed count) only if the context's max_keypoints is too small. */
int slamfe_extract_view(slamfe_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                        const slamfe_keypoint** kps, const uint8_t** desc, int* n_out,
                        uint64_t* frame_handle);

/* Init.cc:75-78 for n <= max_batch frames of one size (the reference takes one
 * frame per loop iteration, Init.cc:124-130; a batch is n iterations); frame i uses imgs[i],
 * kps + i*cap, desc + i*cap*32, n_out[i], frame_handles[i].  kps and desc may
 * be NULL to leave the results on the device only. */
int slamfe_extract_batch(slamfe_ctx* ctx, const uint8_t* const* imgs, int n, int w, int h,
                         int stride, int channels, slamfe_keypoint* kps, uint8_t* desc, int cap,
                         int* n_out, uint64_t* frame_handles);

/* Replaces cv::resize(frame, frame, cv::Size(), fx, fy) (Init.cc:103 and
 * Init.cc:126; the reference uses 0.99) followed by Init::extractKeyPoints on
 * the resized frame (Init.cc:113, Init.cc:130).  bgr: the decoded 8-bit BGR
 * frame.  The resize is OpenCV's INTER_LINEAR (11-bit fixed point), output size
 * (cvRound(w*fx), cvRound(h*fy)) returned in *out_w, *out_h.  resized_bgr, if not
 * NULL, receives the resized BGR frame (out_h rows of out_w*3 bytes), which the
 * reference still needs on the host (Init.cc:294, Init.cc:320-321).  Factors
 * must lie in (0.5, 1].  Everything else as slamfe_extract. */
int slamfe_extract_scaled_bgr(slamfe_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, double fx,
                              double fy, uint8_t* resized_bgr, int* out_w, int* out_h,
                              slamfe_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                              uint64_t* frame_handle);

/* Init.cc:75-78 for frames that already sit in device memory (frame i at
 * d_imgs + i*frame_stride_bytes).  Asynchronous on the context's stream; no
 * host synchronisation.  Results stay device-resident under the handles.
 * d_imgs must be complete when the call is made and stay unchanged until the
 * queued work has run (see "Stream ordering" above). */
int slamfe_extract_resident(slamfe_ctx* ctx, const void* d_imgs, int n, int w, int h, int stride,
                            size_t frame_stride_bytes, int channels, uint64_t* frame_handles);

/* Copy a device-resident frame result (the keypoints / descriptors of
 * Init.cc:76-78) to the host; synchronises. */
int slamfe_frame_download(slamfe_ctx* ctx, uint64_t frame_handle, slamfe_keypoint* kps,
                          uint8_t* desc, int cap, int* n_out);

/* Replaces matcher->knnMatch(desc1, desc2, matches, 2) + the ratio loop
 * (Init.cc:131-153).  query = desc1 (previous frame), train = desc2.
 * out[i] for i < nq. */
int slamfe_match_knn2(slamfe_ctx* ctx, const uint8_t* query, int nq, const uint8_t* train, int nt,
                      slamfe_match2* out);

/* Init.cc:131-153 on two device-resident frames: rows to out[0 .. *nq_out), at
 * most cap.  out may be NULL: then only *nq_out (the query frame's keypoint
 * count) is returned and no matching is done; callers that want the rows on the
 * device use slamfe_match_resident with a buffer of their own. */
int slamfe_match_frames(slamfe_ctx* ctx, uint64_t query_frame, uint64_t train_frame,
                        slamfe_match2* out, int cap, int* nq_out);

/* Init.cc:131-153 as slamfe_match_frames, without the copy into a caller array:
 * *rows points at the *nq_out match rows in the context's page-locked bounce
 * buffer (host memory, read-only for the caller), valid until the next call on
 * this context.  This is the call behind the host layer's knnMatch on two frame
 * handles: the previous and the current frame are already on the device
 * (Init.cc:322-323 hands descriptors2 on to descriptors1), so nothing is uploaded. */
int slamfe_match_frames_view(slamfe_ctx* ctx, uint64_t query_frame, uint64_t train_frame,
                             const slamfe_match2** rows, int* nq_out);

/* Asynchronous form of Init.cc:131-153: both frames device-resident, result written to
 * d_out[0..max_keypoints) (device memory), row counts read on the device.  Rows
 * beyond the query frame's count are left untouched.  d_out must not be written
 * by another stream while the queued work runs (see "Stream ordering" above). */
int slamfe_match_resident(slamfe_ctx* ctx, uint64_t query_frame, uint64_t train_frame,
                          slamfe_match2* d_out);

/* Batched asynchronous form: pair i matches frame first_query + i against
 * frame first_train + i (handles of consecutive extractions are consecutive),
 * writing rows to d_out + i * max_keypoints.  One launch for all pairs: this
 * is the per-frame loop of Init.cc:130-131 for a pipelined batch. */
int slamfe_match_resident_batch(slamfe_ctx* ctx, uint64_t first_query, uint64_t first_train,
                                int npairs, slamfe_match2* d_out);

/* Sharded brute-force matching (relocalisation against a map split over
 * several GPUs).  d_query: nq x 32 B, d_train: this shard's nt x 32 B whose
 * first row has global index train_base.  Writes, per query, the shard's two
 * best candidates as keys (distance << 32 | global index), ascending;
 * missing candidates are UINT64_MAX.  Integer min on keys is the (distance,
 * index) order of Init.cc:131, so merging shards reproduces the single-GPU
 * result exactly.  Asynchronous. */
int slamfe_match_shard_keys(slamfe_ctx* ctx, const void* d_query, int nq, const void* d_train,
                            int nt, int64_t train_base, uint64_t* d_keys);

/* No counterpart in the reference.  slamfe_match_shard_keys for callers that
 * cannot allocate device memory -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc: same arguments, same keys, but the
 * per-chunk partial results live in a fixed 4 MiB device array of the library
 * (one per GPU, held until the process ends) instead of a buffer that grows on
 * demand, and the call makes no hipMalloc / hipFree.  The job must satisfy
 * chunks(nt) * nq * 8 <= 4 MiB, where a chunk is about 4096 train rows under
 * the matrix-core kernel (2048 to 4096: the number of chunks is rounded so that
 * the workgroups fill the chip evenly) and 128 to 512 rows under the popcount
 * kernel (2000 queries x 1,000,000 rows on the matrix cores need 3.9 MiB, about
 * the most that fits); beyond that it returns
 * SLAMFE_E_CAPACITY and launches nothing.  All contexts on one GPU share the
 * array: two such calls that overlap on one GPU (different streams) race.
 * Asynchronous on the context's stream. */
int slamfe_match_shard_keys_static(slamfe_ctx* ctx, const void* d_query, int nq, const void* d_train,
                                   int nt, int64_t train_base, uint64_t* d_keys);

/* Second half of the sharded form of Init.cc:131-153: merge nshards key sets (laid out [shard][nq][2], e.g. the output of an
 * all-gather) into match rows + ratio flag.  Asynchronous. */
int slamfe_merge_shard_keys(slamfe_ctx* ctx, const uint64_t* d_keys_all, int nshards, int nq,
                            slamfe_match2* d_out);

/* Device mirror of Point::desc (keyframe.h:32): row point_ids[i] of the map
 * descriptor table := desc32[i*32 .. +32).  Called where the reference creates
 * a Point from descriptors2.row(idx) (Init.cc:296).  Ids are dense
 * (m.getPointsSize()), < map_capacity. */
int slamfe_map_put(slamfe_ctx* ctx, const int32_t* point_ids, const uint8_t* desc32, int n);

/* Replaces the inner search of Init.cc:229-259.
 * frame: handle of the current frame (descriptors2 / keypoints2).
 * kp_idx[i]: goodMatches[i].trainIdx of the i-th RANSAC inlier, in loop order.
 * uv / point_ids: rPoints / the map ids behind rPointsIdx (R visible points,
 * ascending, Init.cc:203-218).  For inlier i the best remaining visible point
 * is the one with Hamming distance <= max_hamming and pixel distance <= radius
 * that has the smallest Hamming distance (lowest position on ties); it is then
 * removed from the candidates of later inliers (Init.cc:255-256).
 * matched_point_id[i] = that point's id, or -1 (the caller then triangulates,
 * Init.cc:260). */
int slamfe_search_by_projection(slamfe_ctx* ctx, uint64_t frame, const int32_t* kp_idx, int n_inl,
                                const float* uv, const int32_t* point_ids, int R, float radius,
                                int max_hamming, int32_t* matched_point_id);

/* Mirror of the map points' coordinates: Point::Point(id, x, y, z, ...)
 * (keyframe.cc:6, created at Init.cc:296) and Point::setCoords after bundle
 * adjustment (keyframe.cc:15-25, optimizer.cc:145).  Row point_ids[i] of the
 * device table receives xyz[3*i .. 3*i+2].  Ids as for slamfe_map_put. */
int slamfe_map_put_xyz(slamfe_ctx* ctx, const int32_t* point_ids, const float* xyz, int n);

/* Replaces the reprojection loop of Init.cc:203-218 over map points 0..npoints-1
 * (the reference's map index equals Point::id, Init.cc:296): depth gate against
 * th (Init.cc:205), pt = K4x4 * pose.inv() * X (Init.cc:206-208), the two
 * homogeneous divisions (Init.cc:211, 213) and the image gate 0 < u < W,
 * 0 < v < H (Init.cc:214-215).  pose: keyframe->getPose(), 4x4 row-major float;
 * K: 3x3 row-major float (Init.cc:95-99); th: pose(2,3) (Init.cc:192).  The
 * visible points stay on the device for slamfe_search_by_projection_visible;
 * uv (rPoints, 2 floats each) and point_idx (rPointsIdx), if not NULL, receive
 * them in ascending order, at most cap; *n_visible their number.  A singular
 * pose is SLAMFE_E_INVALID.  Float arithmetic follows OpenCV's small-matrix
 * code: float products summed left to right, LU inverse in float, division by s
 * as multiplication by the float value of 1.0/s. */
int slamfe_reproject(slamfe_ctx* ctx, const float* pose, const float* K, float th, float W, float H,
                     int npoints, float* uv, int32_t* point_idx, int cap, int* n_visible);

/* Init.cc:229-259 as slamfe_search_by_projection, over the visible set the last
 * slamfe_reproject call left on the device (no host round trip of rPoints /
 * rPointsIdx); matched_point_id[i] is the matched point's id or -1. */
int slamfe_search_by_projection_visible(slamfe_ctx* ctx, uint64_t frame_handle, const int32_t* kp_idx,
                                        int n_inl, float radius, int max_hamming,
                                        int32_t* matched_point_id);

/* Replaces the Init::triangulate calls of the per-match loop (Init.cc:260, with
 * Init.cc:5-17) by one call for n point pairs: for pair i the 4x4 system of
 * Init.cc:8-11 from P1, P2 (3x4, row-major float: Init.cc:168-176, 195) and the
 * pixel positions pts1[2i..], pts2[2i..] (keypoints1[queryIdx].pt,
 * keypoints2[trainIdx].pt), its singular value decomposition (Init.cc:14) and
 * xyz[3i..] = Vt.row(3)[0..2] / Vt.row(3)[3] (Init.cc:15-16).  Results may be
 * inf / nan, which the caller tests as at Init.cc:262.  The SVD is the one-sided
 * Jacobi method OpenCV applies to small float matrices. */
int slamfe_triangulate(slamfe_ctx* ctx, const float* P1, const float* P2, const float* pts1,
                       const float* pts2, int n, float* xyz);

/* Wait for everything queued on the context's stream (no counterpart in the
 * reference, whose OpenCV calls are synchronous). */
int slamfe_sync(slamfe_ctx* ctx);

/* No counterpart in the reference: device pointers of a resident frame (valid until the slot is recycled after
 * frame_slots further frames): keypoints, 32-byte descriptor rows, and the
 * device-side count.  *d_count may exceed max_keypoints: the frame then holds
 * only its first max_keypoints keypoints in (level, y, x) order (ties at the
 * retain-best cut-off overflowed the slot); readers clamp the count. */
int slamfe_frame_device_ptrs(slamfe_ctx* ctx, uint64_t frame_handle, const slamfe_keypoint** d_kps,
                             const uint8_t** d_desc, const int32_t** d_count);

/* Measurement hook (no counterpart in the reference): re-run one stage of the
 * most recent extraction of `nframes` frames `reps` times on the context's
 * stream, bracketed by HIP events on that stream, and return the average
 * milliseconds per repetition.  Every stage is idempotent on the resident
 * data, so the frame results are unchanged afterwards.  Gray device frames whose
 * rows have level 0's pitch (width a multiple of 64, rows and frames 16-byte
 * aligned) are read in place by slamfe_extract_resident, not copied: the stages
 * that read level 0 (1, 2, 4) and slamfe_debug_read_level(level 0) read the
 * caller's frames again, which must therefore still be allocated and unchanged. */
#define SLAMFE_STAGE_PYRAMID 1
#define SLAMFE_STAGE_FAST 2 /* fused FAST + NMS + blur kernel */
#define SLAMFE_STAGE_SELECT 3
#define SLAMFE_STAGE_ORIENT 4
#define SLAMFE_STAGE_DESCRIBE 6
#define SLAMFE_STAGE_MATCH 7
int slamfe_debug_time_stage(slamfe_ctx* ctx, int stage, int nframes, int reps, float* avg_ms);

/* Measurement hook (no counterpart in the reference): the same stages timed where
 * they run.  After slamfe_debug_probe(ctx, stage) every slamfe_extract_resident
 * / _batch call (stages 1..6) or slamfe_match_resident_batch call (stage 7)
 * records a pair of HIP events around that stage on the context's stream (the
 * first 64 intervals are kept); slamfe_debug_probe_read waits for the stream and
 * returns their mean in milliseconds and their number.  stage 0 switches the
 * probe off. */
int slamfe_debug_probe(slamfe_ctx* ctx, int stage);
int slamfe_debug_probe_read(slamfe_ctx* ctx, float* avg_ms, int* n);

/* Test hook (no counterpart in the reference): copy pyramid level `level` of
 * frame `frame_in_batch` of the most recent extraction to the host, either
 * the unblurred level (blurred = 0: what FAST and the orientation read) or the
 * blurred one (what the descriptors read).  out: *h rows of *w bytes. */
int slamfe_debug_read_level(slamfe_ctx* ctx, int frame_in_batch, int level, int blurred,
                            uint8_t* out, int out_capacity, int* w, int* h);

/* Library version, for the loader's sanity check (no counterpart in the
 * reference). */
int slamfe_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAMFE_H */
