/* slamfe_oracle.h -- CPU restatement of the reference's feature front-end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (slam_amd/, the
 * library behind include/slamfe.h) may include, link or call this code; it is
 * used by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of
 * bench.py, as the checker and as the timed scalar CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in OpenCV (features2d,
 * core, imgproc), which the reference pulls in with an unpinned
 * find_package(OpenCV REQUIRED) (CMakeLists.txt:37) and which is absent from
 * the reference tree and from this machine.  The reference has no tests or
 * golden vectors for the path.  These functions restate the published OpenCV
 * 4.x algorithms at the reference's call sites (Init.cc:64-66, 75-78, 131,
 * 144-153, 229-259, 389-391); they are checked against analytical
 * known-answers and against independent slow definitions in tests/, not
 * against OpenCV output.
 */
#ifndef SLAMFE_ORACLE_H
#define SLAMFE_ORACLE_H

#include <stdint.h>
#include "slamfe.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16
#define ORC_TRIG_CONTRACT 0 /* slamfe_sincos_deg (the contract both sides implement) */
#define ORC_TRIG_LIBM 1     /* host cosf/sinf, to count the bits that would differ */

typedef struct orc_corner {
    int x, y, score;
} orc_corner;

/* cv::cvtColor(frame, frame, COLOR_BGR2GRAY), Init.cc:75. */
void orc_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray, int gstride);

/* cv::resize(frame, frame, cv::Size(), fx, fy) of Init.cc:103 / Init.cc:126 for
 * 8-bit images with cn interleaved channels: INTER_LINEAR with 11-bit fixed
 * point coefficients (OpenCV's own code path; a build that routes resize
 * through a vendor primitive library may round differently).  The output size
 * is (cvRound(sw*fx), cvRound(sh*fy)); dst rows hold dw*cn bytes. */
void orc_resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, int cn, double fx,
                          double fy, uint8_t* dst, int dstride);
void orc_scaled_size(int sw, int sh, double fx, double fy, int* dw, int* dh);

/* FAST-9/16 score of every pixel (0 where the segment test fails), the map
 * cv::FastFeatureDetector builds row by row before non-max suppression
 * (Init.cc:64, Init.cc:76). */
void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int thr, uint8_t* score,
                        int sstride);

/* detector->detect(frame, keypoints), Init.cc:76: segment test, score, strict
 * 3x3 non-max suppression; output in row-major order.  Returns the count
 * (writes at most cap entries). */
int orc_fast9(const uint8_t* img, int w, int h, int stride, int thr, int nonmax, orc_corner* out,
              int cap);

/* Pyramid geometry and per-level feature quotas of cv::ORB (Init.cc:65). */
void orc_level_geometry(int w, int h, int nlevels, float scale_factor, int* lw, int* lh,
                        float* lscale);
void orc_level_quotas(int nfeatures, int nlevels, float scale_factor, int* quota);

/* cv::resize(..., INTER_LINEAR_EXACT) for 8-bit single-channel images. */
void orc_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw,
                             int dh, int dstride);

/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) as ORB applies it to a
 * pyramid level (8-bit fixed-point separable filter). */
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride);

/* Intensity-centroid moments and angle (degrees) at an integer centre. */
void orc_ic_moments(const uint8_t* img, int stride, int cx, int cy, int* m10, int* m01);
float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy);
float orc_fast_atan2(float y, float x);
void orc_sincos_deg(float angle_deg, int trig_mode, float* s, float* c);

/* rBRIEF-256 at (cx,cy) of a blurred level for a keypoint angle in degrees. */
void orc_brief256(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg,
                  int trig_mode, uint8_t desc[32]);

/* Init::extractKeyPoints (Init.cc:74-79) on a gray or BGR frame, in either
 * mode of slamfe_params.  Returns the keypoint count (may exceed cap: then only
 * cap entries were written).  Keypoint order: (level, y, x). */
int orc_extract(const slamfe_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                int trig_mode, slamfe_keypoint* kps, uint8_t* desc, int cap);

/* Intermediate products of orc_extract for stage-by-stage parity checks:
 * level l of the unblurred and of the blurred pyramid (tightly packed,
 * lw[l] bytes per row).  Gray input only. */
void orc_build_pyramid(const uint8_t* gray, int w, int h, int stride, int nlevels,
                       float scale_factor, uint8_t** levels /* caller-allocated, lw*lh each */);

/* matcher->knnMatch(desc1, desc2, matches, 2) + the ratio loop,
 * Init.cc:131-153. */
void orc_match_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt, float ratio,
                    slamfe_match2* out);

/* The search of Init.cc:229-259 with the distance of Init.cc:389-391.
 * kp_xy: 2 floats per keypoint of the current frame; frame_desc: its
 * descriptor rows; kp_idx: the inliers' trainIdx in loop order; uv /
 * point_desc / point_ids: the R visible map points in ascending order. */
void orc_search_by_projection(const float* kp_xy, const uint8_t* frame_desc, const int32_t* kp_idx,
                              int n_inl, const float* uv, const uint8_t* point_desc,
                              const int32_t* point_ids, int R, float radius, int max_hamming,
                              int32_t* matched_point_id);

int orc_hamming256(const uint8_t* a, const uint8_t* b);

/* The reprojection loop of Init.cc:203-218: pose = keyframe->getPose() (4x4
 * row-major float), K (3x3 row-major float), xyz = getCoords() of the map points
 * in map order.  Writes the visible points (rPoints as uv pairs, rPointsIdx) in
 * ascending order and returns their number.  orc_invert4x4 returns 0 and a zero
 * matrix for a singular input, as cv::invert does. */
int orc_invert4x4(const float* src, float* dst);
void orc_matmul4x4(const float* a, const float* b, float* d);
int orc_reproject(const float* pose, const float* K, const float* xyz, int npoints, float th, float W,
                  float H, float* uv, int32_t* idx);

/* Init::triangulate (Init.cc:5-31) for n point pairs: P1, P2 3x4 row-major
 * float, pts 2 floats per point, xyz 3 floats per pair (may be inf / nan, which
 * Init.cc:262 tests).  orc_svd4_last_row: row 3 of Vt of the 4x4 SVD. */
void orc_triangulate(const float* P1, const float* P2, const float* pts1, const float* pts2, int n,
                     float* xyz);
void orc_svd4_last_row(const float* A, float* vt3);

#ifdef __cplusplus
}
#endif
#endif
