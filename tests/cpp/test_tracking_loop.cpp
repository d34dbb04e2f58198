// Replays the reference's tracking loop (Init::processFrames, Init.cc:118-328)
// over a stream of synthetic frames through the C++ host layer, frame after
// frame, and runs the same loop on the CPU checker; every per-frame product
// must be identical: keypoints, descriptors, knnMatch rows and the ratio test
// (Init.cc:130-153), the reprojected map points (Init.cc:203-218), the search by
// projection (Init.cc:229-259) and the triangulated points (Init.cc:260).
//
// What the reference computes with OpenCV's RANSAC / recoverPose (Init.cc:156-166)
// is replaced by fixed inputs: every ratio-test pass is an inlier, and the camera
// translates parallel to the image plane exactly as the synthetic stream pans.
// The map grows as at Init.cc:296 (mapPut + mapPutXyz for every third
// unmatched inlier), all coordinates are rewritten every 10 frames as the bundle
// adjustment does (optimizer.cc:141-146), and the context keeps only four frame
// slots, so the handle ring wraps many times.
// usage: test_tracking_loop frames.raw width height nframes pan_x pan_y
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "slamfe_host.hpp"
#include "slamfe_oracle.h"

#define REQUIRE(c)                                                                     \
    do {                                                                               \
        if (!(c)) {                                                                    \
            std::fprintf(stderr, "FAILED %s (line %d, frame %d)\n", #c, __LINE__, fr); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char** argv)
{
    if (argc != 7) return 2;
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
    const float panx = (float)std::atof(argv[5]), pany = (float)std::atof(argv[6]);
    std::vector<uint8_t> all((size_t)W * H * N);
    {
        FILE* f = std::fopen(argv[1], "rb");
        if (!f || std::fread(all.data(), 1, all.size(), f) != all.size()) {
            std::fprintf(stderr, "cannot read %s\n", argv[1]);
            return 2;
        }
        std::fclose(f);
    }
    int fr = 0;
    const int thold = 30, nfeatures = 10000; // main.cc:11
    const float fx = 984.0f, cx = W / 2.0f, cy = H / 2.0f, z0 = 8.0f; // Init.cc:90-92; all points at one depth
    const float K[9] = {fx, 0, cx, 0, fx, cy, 0, 0, 1};
    try {
        const int mapCapacity = 1 << 15;
        slamfe::FrontEnd fe(nfeatures, thold, W, H, SLAMFE_MODE_PROVIDED, 0, mapCapacity);
        slamfe_params p;
        slamfe_default_params(&p);
        p.mode = SLAMFE_MODE_PROVIDED;
        p.fast_threshold = thold;
        p.nfeatures = nfeatures;
        const int cap = W * H / 4 + 16;
        std::vector<slamfe_keypoint> ok1((size_t)cap), ok2((size_t)cap);
        std::vector<uint8_t> od1((size_t)cap * 32), od2((size_t)cap * 32);

        std::vector<slamfe::KeyPoint> keypoints1, keypoints2;
        slamfe::Mat8 descriptors1, descriptors2;
        uint64_t h1 = fe.extractKeyPoints({all.data(), W, H, W, 1}, keypoints1, descriptors1); // Init.cc:113
        int on1 = orc_extract(&p, all.data(), W, H, W, 1, ORC_TRIG_CONTRACT, ok1.data(), od1.data(), cap);
        REQUIRE((int)keypoints1.size() == on1 && on1 > 50);

        // the map as the host holds it (Point::getCoords / Point::desc by id) -- the
        // checker's inputs; the library keeps its own mirror
        std::vector<float> mapXyz;
        std::vector<uint8_t> mapDesc;
        int npoints = 0, total_hits = 0, total_new = 0, total_good = 0;
        std::vector<std::vector<slamfe::DMatch>> matches;
        for (fr = 1; fr < N; ++fr) {
            const uint8_t* img = all.data() + (size_t)fr * W * H;
            uint64_t h2 = fe.extractKeyPoints({img, W, H, W, 1}, keypoints2, descriptors2); // Init.cc:130
            int on2 = orc_extract(&p, img, W, H, W, 1, ORC_TRIG_CONTRACT, ok2.data(), od2.data(), cap);
            REQUIRE((int)keypoints2.size() == on2);
            REQUIRE(std::memcmp(keypoints2.data(), ok2.data(), (size_t)on2 * sizeof(slamfe_keypoint)) == 0);
            REQUIRE(std::memcmp(descriptors2.data.data(), od2.data(), (size_t)on2 * 32) == 0);

            fe.knnMatch(h1, h2, matches, 2); // Init.cc:131 (matches is not cleared: its inner vectors are reused)
            std::vector<slamfe_match2> om((size_t)on1);
            orc_match_knn2(od1.data(), on1, od2.data(), on2, 0.75f, om.data());
            std::vector<slamfe::DMatch> goodMatches;
            for (size_t i = 0; i < matches.size(); ++i) { // Init.cc:144-153
                if (matches[i].size() < 2) break;
                const float ratio = 0.75;
                REQUIRE(matches[i][0].trainIdx == om[i].idx0 && matches[i][1].trainIdx == om[i].idx1);
                REQUIRE(matches[i][0].distance == (float)om[i].d0 && matches[i][1].distance == (float)om[i].d1);
                bool pass = matches[i][0].distance < ratio * matches[i][1].distance;
                REQUIRE(pass == (om[i].ratio_pass != 0));
                if (pass) goodMatches.push_back(matches[i][0]);
            }
            total_good += (int)goodMatches.size();

            // keyframe pose: the camera has moved with the pan (Init.cc:187-192)
            const float tx = panx * fr * z0 / fx, ty = pany * fr * z0 / fx;
            const float pose[16] = {1, 0, 0, tx, 0, 1, 0, ty, 0, 0, 1, 0, 0, 0, 0, 1};
            const float th = pose[11];

            // Init.cc:203-218 on the device and on the checker
            std::vector<float> rUv, oUv((size_t)npoints * 2 + 2);
            std::vector<int> rIdx;
            std::vector<int32_t> oIdx((size_t)npoints + 1);
            int R = fe.reproject(pose, K, th, (float)W, (float)H, npoints, rUv, rIdx);
            int oR = orc_reproject(pose, K, mapXyz.data(), npoints, th, (float)W, (float)H, oUv.data(), oIdx.data());
            REQUIRE(R == oR);
            REQUIRE(R == 0 || std::memcmp(rUv.data(), oUv.data(), (size_t)R * 8) == 0);
            REQUIRE(R == 0 || std::memcmp(rIdx.data(), oIdx.data(), (size_t)R * 4) == 0);

            // Init.cc:229-259, every good match an inlier, in loop order
            std::vector<int> inl;
            for (auto& m : goodMatches) inl.push_back(m.trainIdx);
            std::vector<int> hit = fe.searchByProjectionVisible(h2, inl);
            std::vector<float> kxy((size_t)on2 * 2);
            for (int i = 0; i < on2; ++i) {
                kxy[2 * (size_t)i] = ok2[(size_t)i].x;
                kxy[2 * (size_t)i + 1] = ok2[(size_t)i].y;
            }
            std::vector<uint8_t> vdesc((size_t)R * 32 + 32);
            for (int j = 0; j < R; ++j) std::memcpy(&vdesc[(size_t)j * 32], &mapDesc[(size_t)oIdx[(size_t)j] * 32], 32);
            std::vector<int32_t> want(inl.size() + 1);
            orc_search_by_projection(kxy.data(), od2.data(), inl.data(), (int)inl.size(), oUv.data(), vdesc.data(),
                                     oIdx.data(), R, 32.f, 128, want.data());
            for (size_t i = 0; i < inl.size(); ++i) REQUIRE(hit[i] == want[i]);
            // the host-array form of the same search gives the same answer
            std::vector<slamfe::Point2f> rPoints;
            for (int j = 0; j < R; ++j) rPoints.emplace_back(rUv[2 * (size_t)j], rUv[2 * (size_t)j + 1]);
            std::vector<int> hit2 = fe.searchByProjection(h2, inl, rPoints, rIdx);
            for (size_t i = 0; i < inl.size(); ++i) REQUIRE(hit2[i] == want[i]);

            // Init.cc:260: triangulate the inliers the search left unmatched
            const float P1[12] = {fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0};
            const float dtx = panx * z0 / fx, dty = pany * z0 / fx;
            const float P2[12] = {fx, 0, cx, -fx * dtx, 0, fx, cy, -fx * dty, 0, 0, 1, 0};
            std::vector<float> t1, t2, xyz, oxyz;
            std::vector<int> unmatched;
            for (size_t i = 0; i < inl.size(); ++i) {
                if (hit[i] >= 0) {
                    ++total_hits;
                    continue;
                }
                unmatched.push_back((int)i);
                t1.push_back(keypoints1[(size_t)goodMatches[i].queryIdx].pt.x);
                t1.push_back(keypoints1[(size_t)goodMatches[i].queryIdx].pt.y);
                t2.push_back(keypoints2[(size_t)inl[i]].pt.x);
                t2.push_back(keypoints2[(size_t)inl[i]].pt.y);
            }
            fe.triangulate(P1, P2, t1, t2, xyz);
            oxyz.resize(xyz.size());
            orc_triangulate(P1, P2, t1.data(), t2.data(), (int)unmatched.size(), oxyz.data());
            REQUIRE(xyz.empty() || std::memcmp(xyz.data(), oxyz.data(), xyz.size() * sizeof(float)) == 0);

            // Init.cc:293-299: new map points (every third unmatched inlier), placed on
            // the ray of their keypoint at the common depth
            std::vector<int> newIds;
            std::vector<float> newXyz;
            for (size_t u = 0; u < unmatched.size(); u += 3) {
                if (npoints >= mapCapacity) break;
                const int idx = inl[(size_t)unmatched[u]];
                const float X[3] = {(keypoints2[(size_t)idx].pt.x - cx) / fx * z0 + tx,
                                    (keypoints2[(size_t)idx].pt.y - cy) / fx * z0 + ty, z0};
                const int id = npoints++; // m.getPointsSize()
                fe.mapPut(id, descriptors2.ptr(idx));
                newIds.push_back(id);
                newXyz.insert(newXyz.end(), X, X + 3);
                mapXyz.insert(mapXyz.end(), X, X + 3);
                mapDesc.insert(mapDesc.end(), descriptors2.ptr(idx), descriptors2.ptr(idx) + 32);
                ++total_new;
            }
            if (!newIds.empty()) fe.mapPutXyz(newIds, newXyz);

            // every 10 frames the bundle adjustment moves every point (optimizer.cc:141-146)
            if (fr % 10 == 0 && npoints > 0) {
                std::vector<int> ids((size_t)npoints);
                for (int i = 0; i < npoints; ++i) {
                    ids[(size_t)i] = i;
                    mapXyz[3 * (size_t)i] += 0.002f * (float)((i * 7 + fr) % 5 - 2);
                    mapXyz[3 * (size_t)i + 1] -= 0.0015f * (float)((i * 3 + fr) % 7 - 3);
                    mapXyz[3 * (size_t)i + 2] += 0.01f * (float)((i + fr) % 3 - 1);
                }
                fe.mapPutXyz(ids, mapXyz);
            }

            keypoints1 = keypoints2; // Init.cc:322-323
            descriptors1 = descriptors2;
            h1 = h2;
            ok1 = ok2;
            od1 = od2;
            on1 = on2;
        }
        fr = N;
        REQUIRE(total_hits > 5 * N && total_new > 5 * N && npoints > 100);
        std::printf("tracking loop ok: %d frames, %d good matches, %d map points, %d re-observed, %d created\n", N,
                    total_good, npoints, total_hits, total_new);
    } catch (const slamfe::Error& e) {
        std::fprintf(stderr, "slamfe::Error %d at frame %d: %s\n", e.code, fr, e.what());
        return 3;
    }
    return 0;
}
