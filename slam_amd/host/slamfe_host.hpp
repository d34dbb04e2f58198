// slamfe_host.hpp -- header-only C++ host layer over the C ABI of slamfe.h.
//
// It gives the reference's tracking thread (Init::processFrames, Init.cc:87-328)
// the calls it makes today, with the same names, argument order and result
// types, so that the three call sites can be switched with a few lines:
//
//   Init.cc:64-66  detector / desc / matcher creation  -> slamfe::FrontEnd ctor
//   Init.cc:74-79  Init::extractKeyPoints(frame, keypoints, descriptors)
//                                                       -> FrontEnd::extractKeyPoints
//   Init.cc:131    matcher->knnMatch(d1, d2, matches, 2) -> FrontEnd::knnMatch (on the two
//                  frames' handles: both are on the device already; the descriptor
//                  matrices need not travel again)
//   Init.cc:229-259 the search-by-projection double loop -> FrontEnd::searchByProjection
//   Init.cc:296    new Point(..., descriptors2.row(idx), ...) -> FrontEnd::mapPut
//
// With -DSLAMFE_WITH_OPENCV the types are OpenCV's own (cv::KeyPoint, cv::Mat,
// cv::DMatch); without it, layout-compatible stand-ins are used so that the
// layer can be compiled and tested on a machine without OpenCV.
//
// Error behaviour: OpenCV reports failures by throwing cv::Exception, which the
// reference never catches.  This layer throws slamfe::Error (a
// std::runtime_error) for every non-zero status of the C ABI.
#ifndef SLAMFE_HOST_HPP
#define SLAMFE_HOST_HPP

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "slamfe.h"

#ifdef SLAMFE_WITH_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#endif

namespace slamfe {

#ifdef SLAMFE_WITH_OPENCV
using KeyPoint = cv::KeyPoint;
using DMatch = cv::DMatch;
using Point2f = cv::Point2f;
#else
struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
// same 28-byte layout as cv::KeyPoint
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
// same 16-byte layout as cv::DMatch
struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 0;
};
#endif
static_assert(sizeof(KeyPoint) == sizeof(slamfe_keypoint), "KeyPoint must match cv::KeyPoint's layout");
static_assert(sizeof(DMatch) == 16, "DMatch must match cv::DMatch's layout");

// A continuous CV_8U matrix (rows x cols), the shape of `descriptors` in the
// reference; and the view of an 8-bit image (gray or BGR).
struct Mat8 {
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    uint8_t* ptr(int r) { return data.data() + (size_t)r * cols; }
    const uint8_t* ptr(int r) const { return data.data() + (size_t)r * cols; }
    bool empty() const { return rows == 0; }
};

struct ImageView {
    const uint8_t* data;
    int width, height, stride, channels; // channels: 1 gray, 3 BGR
};

class Error : public std::runtime_error {
public:
    Error(int code_, const std::string& what_) : std::runtime_error(what_), code(code_) {}
    int code;
};

class FrontEnd {
public:
    // Same argument meaning as the object creation of Init::Init
    // (Init.cc:64-66): nfeatures, FAST threshold; the remaining ORB arguments
    // are the reference's constants (1.2f, 8 levels, edge 31, patch 31).
    // mode SLAMFE_MODE_PROVIDED reproduces the reference's literal
    // FAST-then-ORB::compute call pair; SLAMFE_MODE_ORB is full ORB.
    FrontEnd(int nfeatures, int thold, int maxWidth, int maxHeight, int mode = SLAMFE_MODE_PROVIDED,
             int device = 0, int mapCapacity = 1 << 16)
    {
        slamfe_params p;
        slamfe_default_params(&p);
        p.mode = mode;
        p.nfeatures = nfeatures;
        p.fast_threshold = thold;
        p.max_width = maxWidth;
        p.max_height = maxHeight;
        p.device_id = device;
        p.map_capacity = mapCapacity;
        int rc = slamfe_create(&p, &ctx_);
        if (rc != SLAMFE_OK) {
            std::string msg = ctx_ ? slamfe_last_error(ctx_) : "slamfe_create failed";
            if (ctx_) slamfe_destroy(ctx_);
            ctx_ = nullptr;
            throw Error(rc, msg);
        }
        capacity_ = mode == SLAMFE_MODE_ORB ? (2 * nfeatures > 4096 ? 2 * nfeatures : 4096) : 32768;
    }
    ~FrontEnd()
    {
        if (ctx_) slamfe_destroy(ctx_);
    }
    FrontEnd(const FrontEnd&) = delete;
    FrontEnd& operator=(const FrontEnd&) = delete;

    // Init::extractKeyPoints(frame, keypoints, descriptors), Init.cc:74-79.
    // Returns the handle of the device-resident copy of this frame's result.
    // The library hands out its page-locked result arrays; exactly n records and
    // n rows are copied from there (no buffers sized for the worst case, nothing
    // value-initialised first).
    uint64_t extractKeyPoints(const ImageView& frame, std::vector<KeyPoint>& keypoints, Mat8& descriptors)
    {
        const uint8_t* rows = nullptr;
        int n = 0;
        uint64_t handle = extractView(frame, keypoints, rows, n);
        descriptors.rows = n;
        descriptors.cols = 32;
        descriptors.data.assign(rows, rows + (size_t)n * 32);
        return handle;
    }

    // cv::resize(frame, frame, cv::Size(), fx, fy) (Init.cc:103, Init.cc:126) followed
    // by Init::extractKeyPoints on the resized frame (Init.cc:113, Init.cc:130).
    // resized receives the resized BGR frame, which the caller keeps using on
    // the host (Init.cc:294).
    uint64_t resizeAndExtract(const ImageView& frame, double fx, double fy, std::vector<uint8_t>& resized,
                              int& outWidth, int& outHeight, std::vector<KeyPoint>& keypoints,
                              Mat8& descriptors)
    {
        if (frame.channels != 3) throw Error(SLAMFE_E_INVALID, "resizeAndExtract takes a BGR frame");
        // staging arrays of the layer, allocated once and never initialised
        if (!stageKps_) {
            stageKps_.reset(new slamfe_keypoint[(size_t)capacity_]);
            stageDesc_.reset(new uint8_t[(size_t)capacity_ * 32]);
        }
        if (resized.size() < (size_t)frame.width * frame.height * 3)
            resized.resize((size_t)frame.width * frame.height * 3);
        int n = 0;
        uint64_t handle = 0;
        check(slamfe_extract_scaled_bgr(ctx_, frame.data, frame.width, frame.height, frame.stride, fx, fy,
                                        resized.data(), &outWidth, &outHeight, stageKps_.get(),
                                        stageDesc_.get(), capacity_, &n, &handle));
        resized.resize((size_t)outWidth * outHeight * 3);
        const KeyPoint* k = reinterpret_cast<const KeyPoint*>(stageKps_.get());
        keypoints.assign(k, k + n);
        descriptors.rows = n;
        descriptors.cols = 32;
        descriptors.data.assign(stageDesc_.get(), stageDesc_.get() + (size_t)n * 32);
        return handle;
    }

    // Mirror of Point::getCoords() by Point::id (keyframe.cc:6, 15-25, optimizer.cc:145).
    void mapPutXyz(int pointId, float x, float y, float z)
    {
        const float xyz[3] = {x, y, z};
        check(slamfe_map_put_xyz(ctx_, &pointId, xyz, 1));
    }
    void mapPutXyz(const std::vector<int>& pointIds, const std::vector<float>& xyz)
    {
        if (xyz.size() != pointIds.size() * 3) throw Error(SLAMFE_E_INVALID, "three coordinates per point id");
        check(slamfe_map_put_xyz(ctx_, pointIds.data(), xyz.data(), (int)pointIds.size()));
    }

    // The reprojection loop of Init.cc:203-218 over map points 0..npoints-1:
    // fills rPoints (u, v pairs) and rPointsIdx in ascending order; the visible
    // set also stays on the device for searchByProjectionVisible.
    // pose16: keyframe->getPose() row-major, K9: the 3x3 intrinsics row-major.
    int reproject(const float* pose16, const float* K9, float th, float W, float H, int npoints,
                  std::vector<float>& rPointsUv, std::vector<int>& rPointsIdx)
    {
        rPointsUv.resize((size_t)npoints * 2);
        rPointsIdx.resize((size_t)npoints);
        int n = 0;
        check(slamfe_reproject(ctx_, pose16, K9, th, W, H, npoints, rPointsUv.data(), rPointsIdx.data(),
                               npoints, &n));
        rPointsUv.resize((size_t)n * 2);
        rPointsIdx.resize((size_t)n);
        return n;
    }

    // Init.cc:229-259 over the visible set of the last reproject() call.
    std::vector<int> searchByProjectionVisible(uint64_t frameHandle, const std::vector<int>& inlierTrainIdx,
                                               float radius = 32.f, int maxHamming = 128)
    {
        std::vector<int> hit(inlierTrainIdx.size(), -1);
        check(slamfe_search_by_projection_visible(ctx_, frameHandle, inlierTrainIdx.data(),
                                                  (int)inlierTrainIdx.size(), radius, maxHamming, hit.data()));
        return hit;
    }

    // Init::triangulate (Init.cc:5-31) for all the pairs of the per-match loop at
    // once (Init.cc:260): P1, P2 3x4 row-major float, pts 2 floats per point;
    // xyz gets 3 floats per pair (test them with std::isfinite as Init.cc:262 does).
    void triangulate(const float* P1, const float* P2, const std::vector<float>& pts1,
                     const std::vector<float>& pts2, std::vector<float>& xyz)
    {
        if (pts1.size() != pts2.size() || pts1.size() % 2) throw Error(SLAMFE_E_INVALID, "two floats per point, one point of each view per pair");
        const int n = (int)(pts1.size() / 2);
        xyz.resize((size_t)n * 3);
        check(slamfe_triangulate(ctx_, P1, P2, pts1.data(), pts2.data(), n, xyz.data()));
    }

    // matcher->knnMatch(descriptors1, descriptors2, matches, 2), Init.cc:131, on the
    // device-resident copies of the two frames (the handles extractKeyPoints
    // returned for the previous and the current frame): nothing is uploaded.
    // matches[i] holds up to k = 2 DMatch in (distance, trainIdx) order.  Inner
    // vectors that matches already holds are reused, so a caller that does not
    // clear it between frames pays no allocation after the first frame.
    void knnMatch(uint64_t queryFrame, uint64_t trainFrame, std::vector<std::vector<DMatch>>& matches, int k = 2)
    {
        if (k != 2) throw Error(SLAMFE_E_INVALID, "only k = 2 is implemented (the reference's call)");
        const slamfe_match2* rows = nullptr;
        int nq = 0;
        check(slamfe_match_frames_view(ctx_, queryFrame, trainFrame, &rows, &nq));
        fillMatches(rows, nq, matches);
    }

    // Init.cc:131-153 in one step: knnMatch on the two frames' handles followed by
    // the reference's ratio loop, which stops at the first query with fewer than two
    // neighbours (Init.cc:145) and keeps matches[i][0] where
    // matches[i][0].distance < 0.75f * matches[i][1].distance (Init.cc:146-147; the
    // library evaluates the same float comparison with the context's ratio).  Fills
    // goodMatches, pMatches1 and pMatches2 as Init.cc:148-151 do, without building
    // the vector of vectors.  Returns the number of queries (matches.size()).
    int knnMatchRatio(uint64_t queryFrame, uint64_t trainFrame, const std::vector<KeyPoint>& keypoints1,
                      const std::vector<KeyPoint>& keypoints2, std::vector<DMatch>& goodMatches,
                      std::vector<Point2f>& pMatches1, std::vector<Point2f>& pMatches2)
    {
        const slamfe_match2* rows = nullptr;
        int nq = 0;
        check(slamfe_match_frames_view(ctx_, queryFrame, trainFrame, &rows, &nq));
        goodMatches.clear();
        pMatches1.clear();
        pMatches2.clear();
        for (int i = 0; i < nq; ++i) {
            const slamfe_match2& m = rows[i];
            if (m.nfound < 2) break;
            if (!m.ratio_pass) continue;
            goodMatches.push_back(make(i, m.idx0, m.d0));
            pMatches1.push_back(keypoints1[(size_t)i].pt);
            pMatches2.push_back(keypoints2[(size_t)m.idx0].pt);
        }
        return nq;
    }

    // The same call on descriptor matrices in host memory (both are uploaded): for
    // descriptors that did not come from extractKeyPoints.
    void knnMatch(const Mat8& query, const Mat8& train, std::vector<std::vector<DMatch>>& matches, int k = 2)
    {
        knnMatchHost(query.data.data(), query.rows, train.data.data(), train.rows, matches, k);
    }

    // new Point(id, ..., descriptors2.row(idx), ...), Init.cc:296: mirror the
    // descriptor row of map point `id` on the device.
    void mapPut(int id, const uint8_t* desc32)
    {
        int32_t pid = id;
        check(slamfe_map_put(ctx_, &pid, desc32, 1));
    }

    // The inner search of Init.cc:229-259.  inlierTrainIdx[i] is
    // goodMatches[i].trainIdx of the i-th entry with mask[i] != 0, in loop order;
    // rPoints / rPointIds are the reprojected visible points of Init.cc:203-218
    // (ids = points[rPointsIdx[j]]->id).  Result[i] = id of the map point the
    // reference would attach keypoint i to, or -1 (then it triangulates).
    std::vector<int> searchByProjection(uint64_t frameHandle, const std::vector<int>& inlierTrainIdx,
                                        const std::vector<Point2f>& rPoints, const std::vector<int>& rPointIds,
                                        float radius = 32.f, int maxHamming = 128)
    {
        std::vector<int> out(inlierTrainIdx.size(), -1);
        if (inlierTrainIdx.empty()) return out;
        static_assert(sizeof(Point2f) == 8, "Point2f is two floats");
        check(slamfe_search_by_projection(ctx_, frameHandle, inlierTrainIdx.data(), (int)inlierTrainIdx.size(),
                                          reinterpret_cast<const float*>(rPoints.data()), rPointIds.data(),
                                          (int)rPoints.size(), radius, maxHamming, out.data()));
        return out;
    }

    slamfe_ctx* ctx() { return ctx_; }

#ifdef SLAMFE_WITH_OPENCV
    // Drop-in signatures for the reference's own types.
    uint64_t extractKeyPoints(const cv::Mat& frame, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors)
    {
        CV_Assert(frame.depth() == CV_8U && (frame.channels() == 1 || frame.channels() == 3));
        ImageView v{frame.data, frame.cols, frame.rows, (int)frame.step, frame.channels()};
        const uint8_t* rows = nullptr;
        int n = 0;
        uint64_t h = extractView(v, keypoints, rows, n);
        // a fresh buffer every frame: Mat::create alone would keep the old one when the
        // keypoint count repeats, and descriptors1 / Point::desc still refer to it
        // (Init.cc:296, Init.cc:323)
        descriptors.release();
        descriptors.create(n, 32, CV_8U);
        if (n) std::memcpy(descriptors.data, rows, (size_t)n * 32);
        return h;
    }
    void knnMatch(const cv::Mat& query, const cv::Mat& train, std::vector<std::vector<cv::DMatch>>& matches, int k)
    {
        CV_Assert(query.isContinuous() && train.isContinuous() && query.cols == 32 && train.cols == 32);
        knnMatchHost(query.data, query.rows, train.data, train.rows, matches, k);
    }
#endif

private:
    // slamfe_extract_view + the copy of exactly n keypoint records
    uint64_t extractView(const ImageView& frame, std::vector<KeyPoint>& keypoints, const uint8_t*& rows, int& n)
    {
        const slamfe_keypoint* kp = nullptr;
        uint64_t handle = 0;
        check(slamfe_extract_view(ctx_, frame.data, frame.width, frame.height, frame.stride, frame.channels, &kp,
                                  &rows, &n, &handle));
        const KeyPoint* k = reinterpret_cast<const KeyPoint*>(kp);
        keypoints.assign(k, k + n);
        return handle;
    }
    void knnMatchHost(const uint8_t* query, int nq, const uint8_t* train, int nt,
                      std::vector<std::vector<DMatch>>& matches, int k)
    {
        if (k != 2) throw Error(SLAMFE_E_INVALID, "only k = 2 is implemented (the reference's call)");
        matchRows_.resize((size_t)nq);
        check(slamfe_match_knn2(ctx_, query, nq, train, nt, matchRows_.data()));
        fillMatches(matchRows_.data(), nq, matches);
    }
    // rows -> vector<vector<DMatch>>; existing inner vectors keep their storage
    static void fillMatches(const slamfe_match2* rows, int nq, std::vector<std::vector<DMatch>>& matches)
    {
        matches.resize((size_t)nq);
        for (int i = 0; i < nq; ++i) {
            const slamfe_match2& m = rows[i];
            std::vector<DMatch>& v = matches[(size_t)i];
            v.resize(m.nfound);
            if (m.nfound > 0) v[0] = make(i, m.idx0, m.d0);
            if (m.nfound > 1) v[1] = make(i, m.idx1, m.d1);
        }
    }
    static DMatch make(int q, int t, int d)
    {
        DMatch m;
        m.queryIdx = q;
        m.trainIdx = t;
        m.imgIdx = 0;
        m.distance = (float)d;
        return m;
    }
    void check(int rc)
    {
        if (rc != SLAMFE_OK) throw Error(rc, slamfe_last_error(ctx_));
    }
    slamfe_ctx* ctx_ = nullptr;
    int capacity_ = 0;
    std::unique_ptr<slamfe_keypoint[]> stageKps_;
    std::unique_ptr<uint8_t[]> stageDesc_;
    std::vector<slamfe_match2> matchRows_;
};

} // namespace slamfe
#endif
