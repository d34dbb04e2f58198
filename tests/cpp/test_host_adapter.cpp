// Drives the C++ host layer the way Init::processFrames drives OpenCV
// (Init.cc:113, 130-153, 229-259) on two synthetic frames and checks every
// result against the CPU checker.  Exit code 0 = identical.
// usage: test_host_adapter frameA.raw frameB.raw width height
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "slamfe_host.hpp"
#include "slamfe_oracle.h"

static std::vector<uint8_t> read_file(const char* path, size_t n)
{
    std::vector<uint8_t> v(n);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), 1, n, f) != n) {
        std::fprintf(stderr, "cannot read %s\n", path);
        std::exit(2);
    }
    std::fclose(f);
    return v;
}

#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) {                                                       \
            std::fprintf(stderr, "FAILED %s (line %d)\n", #c, __LINE__);  \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
    std::vector<uint8_t> fa = read_file(argv[1], (size_t)W * H), fb = read_file(argv[2], (size_t)W * H);
    const int thold = 30, nfeatures = 10000; // main.cc:11
    try {
        slamfe::FrontEnd fe(nfeatures, thold, W, H, SLAMFE_MODE_PROVIDED);
        std::vector<slamfe::KeyPoint> keypoints1, keypoints2;
        slamfe::Mat8 descriptors1, descriptors2;
        uint64_t h1 = fe.extractKeyPoints({fa.data(), W, H, W, 1}, keypoints1, descriptors1); // Init.cc:113
        uint64_t h2 = fe.extractKeyPoints({fb.data(), W, H, W, 1}, keypoints2, descriptors2); // Init.cc:130
        std::vector<std::vector<slamfe::DMatch>> matches, matchesHost;
        fe.knnMatch(h1, h2, matches, 2); // Init.cc:131 on the resident frames
        fe.knnMatch(descriptors1, descriptors2, matchesHost, 2); // ... and on host matrices: the same rows
        REQUIRE(matches.size() == matchesHost.size());
        for (size_t i = 0; i < matches.size(); ++i) {
            REQUIRE(matches[i].size() == matchesHost[i].size());
            REQUIRE(matches[i].empty() ||
                    std::memcmp(matches[i].data(), matchesHost[i].data(), matches[i].size() * sizeof(slamfe::DMatch)) == 0);
        }
        // a second call into the same container reuses its inner vectors and gives the same rows
        const slamfe::DMatch* inner0 = matches.empty() ? nullptr : matches[0].data();
        fe.knnMatch(h1, h2, matches, 2);
        REQUIRE(matches.empty() || matches[0].data() == inner0);

        // the checker on the same inputs
        slamfe_params p;
        slamfe_default_params(&p);
        p.mode = SLAMFE_MODE_PROVIDED;
        p.fast_threshold = thold;
        p.nfeatures = nfeatures;
        const int cap = W * H / 4 + 16;
        std::vector<slamfe_keypoint> k1((size_t)cap), k2((size_t)cap);
        std::vector<uint8_t> d1((size_t)cap * 32), d2((size_t)cap * 32);
        int n1 = orc_extract(&p, fa.data(), W, H, W, 1, ORC_TRIG_CONTRACT, k1.data(), d1.data(), cap);
        int n2 = orc_extract(&p, fb.data(), W, H, W, 1, ORC_TRIG_CONTRACT, k2.data(), d2.data(), cap);
        REQUIRE((int)keypoints1.size() == n1 && (int)keypoints2.size() == n2 && n1 > 100);
        REQUIRE(std::memcmp(keypoints1.data(), k1.data(), (size_t)n1 * sizeof(slamfe_keypoint)) == 0);
        REQUIRE(std::memcmp(keypoints2.data(), k2.data(), (size_t)n2 * sizeof(slamfe_keypoint)) == 0);
        REQUIRE(std::memcmp(descriptors1.data.data(), d1.data(), (size_t)n1 * 32) == 0);
        REQUIRE(std::memcmp(descriptors2.data.data(), d2.data(), (size_t)n2 * 32) == 0);
        std::vector<slamfe_match2> om((size_t)n1);
        orc_match_knn2(d1.data(), n1, d2.data(), n2, 0.75f, om.data());

        // Lowe's ratio test exactly as Init.cc:144-153
        std::vector<slamfe::DMatch> goodMatches;
        int oracle_good = 0;
        for (size_t i = 0; i < matches.size(); ++i) {
            if (matches[i].size() < 2) break;
            const float ratio = 0.75;
            REQUIRE(matches[i][0].trainIdx == om[i].idx0 && matches[i][1].trainIdx == om[i].idx1);
            REQUIRE(matches[i][0].distance == (float)om[i].d0 && matches[i][1].distance == (float)om[i].d1);
            REQUIRE(matches[i][0].queryIdx == (int)i && matches[i][0].imgIdx == 0);
            bool pass = matches[i][0].distance < ratio * matches[i][1].distance;
            REQUIRE(pass == (om[i].ratio_pass != 0));
            if (pass) goodMatches.push_back(matches[i][0]);
            oracle_good += om[i].ratio_pass;
        }
        REQUIRE((int)goodMatches.size() == oracle_good && oracle_good > 50);
        {   // Init.cc:131-153 in one step gives the same three vectors
            std::vector<slamfe::DMatch> good2;
            std::vector<slamfe::Point2f> p1, p2;
            REQUIRE(fe.knnMatchRatio(h1, h2, keypoints1, keypoints2, good2, p1, p2) == n1);
            REQUIRE(good2.size() == goodMatches.size() && p1.size() == good2.size() && p2.size() == good2.size());
            for (size_t i = 0; i < good2.size(); ++i) {
                REQUIRE(std::memcmp(&good2[i], &goodMatches[i], sizeof(slamfe::DMatch)) == 0);
                REQUIRE(p1[i].x == keypoints1[(size_t)good2[i].queryIdx].pt.x && p1[i].y == keypoints1[(size_t)good2[i].queryIdx].pt.y);
                REQUIRE(p2[i].x == keypoints2[(size_t)good2[i].trainIdx].pt.x && p2[i].y == keypoints2[(size_t)good2[i].trainIdx].pt.y);
            }
        }

        // a small map: every third good match becomes a map point made from
        // frame 2's descriptor row (Init.cc:296), projected near its keypoint
        std::vector<slamfe::Point2f> rPoints;
        std::vector<int> rIds;
        std::vector<uint8_t> pdesc;
        std::vector<float> uv;
        int id = 0;
        for (size_t i = 0; i < goodMatches.size(); i += 3) {
            int t = goodMatches[i].trainIdx;
            fe.mapPut(id, descriptors2.ptr(t));
            rPoints.emplace_back(keypoints2[(size_t)t].pt.x + 1.5f, keypoints2[(size_t)t].pt.y - 2.25f);
            uv.push_back(rPoints.back().x);
            uv.push_back(rPoints.back().y);
            pdesc.insert(pdesc.end(), descriptors2.ptr(t), descriptors2.ptr(t) + 32);
            rIds.push_back(id++);
        }
        std::vector<int> inl;
        for (auto& m : goodMatches) inl.push_back(m.trainIdx);
        std::vector<int> got = fe.searchByProjection(h2, inl, rPoints, rIds);
        std::vector<float> kxy((size_t)n2 * 2);
        for (int i = 0; i < n2; ++i) {
            kxy[2 * (size_t)i] = k2[(size_t)i].x;
            kxy[2 * (size_t)i + 1] = k2[(size_t)i].y;
        }
        std::vector<int32_t> want(inl.size());
        orc_search_by_projection(kxy.data(), d2.data(), inl.data(), (int)inl.size(), uv.data(), pdesc.data(),
                                 rIds.data(), (int)rIds.size(), 32.f, 128, want.data());
        int hits = 0;
        for (size_t i = 0; i < inl.size(); ++i) {
            REQUIRE(got[i] == want[i]);
            hits += got[i] >= 0;
        }
        REQUIRE(hits >= (int)rIds.size() / 2);

        // the same search with the reprojection done by the library (Init.cc:203-218):
        // map points placed on the rays of their pixels under an identity pose
        const float f = 984.0f, K[9] = {f, 0, W / 2.0f, 0, f, H / 2.0f, 0, 0, 1};
        const float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::vector<float> xyz;
        for (size_t j = 0; j < rIds.size(); ++j) {
            float z = 2.0f + (float)(j % 7);
            xyz.push_back((uv[2 * j] - K[2]) / f * z);
            xyz.push_back((uv[2 * j + 1] - K[5]) / f * z);
            xyz.push_back(z);
        }
        fe.mapPutXyz(rIds, xyz);
        std::vector<float> ruv, ouv(uv.size());
        std::vector<int> ridx;
        std::vector<int32_t> oidx(rIds.size());
        int R = fe.reproject(pose, K, pose[11], (float)W, (float)H, (int)rIds.size(), ruv, ridx);
        int oR = orc_reproject(pose, K, xyz.data(), (int)rIds.size(), pose[11], (float)W, (float)H, ouv.data(),
                               oidx.data());
        REQUIRE(R == oR && R > (int)rIds.size() / 2);
        REQUIRE(std::memcmp(ruv.data(), ouv.data(), (size_t)R * 8) == 0);
        REQUIRE(std::memcmp(ridx.data(), oidx.data(), (size_t)R * 4) == 0);
        std::vector<int> got2 = fe.searchByProjectionVisible(h2, inl);
        std::vector<uint8_t> vdesc;
        for (int j = 0; j < R; ++j) vdesc.insert(vdesc.end(), pdesc.begin() + 32 * oidx[(size_t)j], pdesc.begin() + 32 * oidx[(size_t)j] + 32);
        orc_search_by_projection(kxy.data(), d2.data(), inl.data(), (int)inl.size(), ouv.data(), vdesc.data(),
                                 oidx.data(), R, 32.f, 128, want.data());
        for (size_t i = 0; i < inl.size(); ++i) REQUIRE(got2[i] == want[i]);

        // cv::resize(frame, frame, Size(), 0.99, 0.99) + extractKeyPoints (Init.cc:103, 113)
        std::vector<uint8_t> bgr((size_t)W * H * 3), resized, oresized((size_t)W * H * 3);
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            bgr[3 * i] = fa[i];
            bgr[3 * i + 1] = fb[i];
            bgr[3 * i + 2] = (uint8_t)((fa[i] + fb[i]) / 2);
        }
        int rw = 0, rh = 0, ow = 0, oh = 0;
        std::vector<slamfe::KeyPoint> keypoints3;
        slamfe::Mat8 descriptors3;
        fe.resizeAndExtract({bgr.data(), W, H, W * 3, 3}, 0.99, 0.99, resized, rw, rh, keypoints3, descriptors3);
        orc_scaled_size(W, H, 0.99, 0.99, &ow, &oh);
        REQUIRE(rw == ow && rh == oh);
        orc_resize_linear_8u(bgr.data(), W, H, W * 3, 3, 0.99, 0.99, oresized.data(), ow * 3);
        REQUIRE(std::memcmp(resized.data(), oresized.data(), (size_t)ow * oh * 3) == 0);
        int n3 = orc_extract(&p, oresized.data(), ow, oh, ow * 3, 3, ORC_TRIG_CONTRACT, k1.data(), d1.data(), cap);
        REQUIRE((int)keypoints3.size() == n3 && n3 > 100);
        REQUIRE(std::memcmp(keypoints3.data(), k1.data(), (size_t)n3 * sizeof(slamfe_keypoint)) == 0);
        REQUIRE(std::memcmp(descriptors3.data.data(), d1.data(), (size_t)n3 * 32) == 0);

        // Init::triangulate for all good matches at once (Init.cc:260, 5-17)
        const float P1[12] = {f, 0, W / 2.0f, 0, 0, f, H / 2.0f, 0, 0, 0, 1, 0};
        const float P2[12] = {f, 0, W / 2.0f, -98.4f, 0, f, H / 2.0f, 9.84f, 0, 0, 1, 0.02f};
        std::vector<float> t1, t2, txyz, oxyz;
        for (auto& m : goodMatches) {
            t1.push_back(keypoints1[(size_t)m.queryIdx].pt.x);
            t1.push_back(keypoints1[(size_t)m.queryIdx].pt.y);
            t2.push_back(keypoints2[(size_t)m.trainIdx].pt.x);
            t2.push_back(keypoints2[(size_t)m.trainIdx].pt.y);
        }
        fe.triangulate(P1, P2, t1, t2, txyz);
        oxyz.resize(txyz.size());
        orc_triangulate(P1, P2, t1.data(), t2.data(), (int)goodMatches.size(), oxyz.data());
        REQUIRE(txyz.size() == goodMatches.size() * 3);
        REQUIRE(std::memcmp(txyz.data(), oxyz.data(), txyz.size() * sizeof(float)) == 0);
        std::printf("host adapter ok: %d/%d keypoints, %d good matches, %d map hits, %d reprojected, %d keypoints after resize, %d triangulated\n",
                    n1, n2, oracle_good, hits, R, n3, (int)goodMatches.size());
    } catch (const slamfe::Error& e) {
        std::fprintf(stderr, "slamfe::Error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
