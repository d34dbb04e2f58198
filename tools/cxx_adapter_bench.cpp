// cxx_adapter_bench.cpp -- latency of the C++ host layer (slamfe_host.hpp), the one
// the reference's tracking thread binds (Init.cc:74-79, Init.cc:130-153, Init.cc:322-323):
// extractKeyPoints into a std::vector<KeyPoint> and a descriptor matrix, then
// knnMatch on the two frames' handles into a vector<vector<DMatch>>, with the
// reference's swap of the frames at the end of the iteration.  Timed beside the C
// caller of tools/single_call_bench.c, which hands plain arrays to the C ABI.
// usage: cxx_adapter_bench frameA.raw frameB.raw width height mode fast_threshold nfeatures reps
// prints one JSON object with median and mean milliseconds.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slamfe_host.hpp"

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

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

static void stats(std::vector<double>& t, double* median, double* mean)
{
    double s = 0;
    for (double x : t) s += x;
    std::sort(t.begin(), t.end());
    *median = t[t.size() / 2];
    *mean = s / (double)t.size();
}

int main(int argc, char** argv)
{
    if (argc != 9) return 2;
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), mode = std::atoi(argv[5]), thr = std::atoi(argv[6]),
              nfeat = std::atoi(argv[7]), reps = std::atoi(argv[8]);
    std::vector<uint8_t> a = read_file(argv[1], (size_t)w * h), b = read_file(argv[2], (size_t)w * h);
    try {
        slamfe::FrontEnd fe(nfeat, thr, w, h, mode, 0, 0);
        std::vector<slamfe::KeyPoint> keypoints1, keypoints2;
        slamfe::Mat8 descriptors1, descriptors2;
        std::vector<std::vector<slamfe::DMatch>> matches;
        std::vector<slamfe::DMatch> goodMatches;
        std::vector<slamfe::Point2f> pMatches1, pMatches2;
        std::vector<double> t((size_t)reps);
        uint64_t h1 = 0, h2 = 0;
        const slamfe::ImageView va{a.data(), w, h, w, 1}, vb{b.data(), w, h, w, 1};
        for (int i = 0; i < 10; ++i) h1 = fe.extractKeyPoints(va, keypoints1, descriptors1);
        for (int i = 0; i < reps; ++i) {
            double t0 = now_ms();
            h1 = fe.extractKeyPoints(va, keypoints1, descriptors1);
            t[(size_t)i] = now_ms() - t0;
        }
        double ext_med, ext_mean, loop_med, loop_mean, clr_med, clr_mean, ratio_med, ratio_mean, up_med, up_mean;
        stats(t, &ext_med, &ext_mean);
        // the patched loop: Init.cc:128-131 and the hand-over of Init.cc:322-323
        for (int i = 0; i < reps + 5; ++i) {
            double t0 = now_ms();
            keypoints2.clear();
            h2 = fe.extractKeyPoints(vb, keypoints2, descriptors2);
            fe.knnMatch(h1, h2, matches, 2);
            keypoints1 = keypoints2;
            std::swap(descriptors1, descriptors2); // cv::Mat assignment shares the buffer: no copy either
            h1 = h2;
            if (i >= 5) t[(size_t)(i - 5)] = now_ms() - t0;
        }
        stats(t, &loop_med, &loop_mean);
        const size_t nmatch = matches.size();
        // the same with the reference's matches.clear() left in (Init.cc:129): 2 k inner vectors a frame
        for (int i = 0; i < reps + 5; ++i) {
            double t0 = now_ms();
            keypoints2.clear();
            matches.clear();
            h2 = fe.extractKeyPoints(vb, keypoints2, descriptors2);
            fe.knnMatch(h1, h2, matches, 2);
            keypoints1 = keypoints2;
            std::swap(descriptors1, descriptors2);
            h1 = h2;
            if (i >= 5) t[(size_t)(i - 5)] = now_ms() - t0;
        }
        stats(t, &clr_med, &clr_mean);
        // knnMatch + the ratio loop of Init.cc:144-153 in one step
        for (int i = 0; i < reps + 5; ++i) {
            double t0 = now_ms();
            keypoints2.clear();
            h2 = fe.extractKeyPoints(vb, keypoints2, descriptors2);
            fe.knnMatchRatio(h1, h2, keypoints1, keypoints2, goodMatches, pMatches1, pMatches2);
            keypoints1 = keypoints2;
            std::swap(descriptors1, descriptors2);
            h1 = h2;
            if (i >= 5) t[(size_t)(i - 5)] = now_ms() - t0;
        }
        stats(t, &ratio_med, &ratio_mean);
        // knnMatch on the host matrices (both uploaded again): the binding before the handles
        fe.extractKeyPoints(va, keypoints1, descriptors1);
        for (int i = 0; i < reps + 5; ++i) {
            double t0 = now_ms();
            keypoints2.clear();
            fe.extractKeyPoints(vb, keypoints2, descriptors2);
            fe.knnMatch(descriptors1, descriptors2, matches, 2);
            keypoints1 = keypoints2;
            std::swap(descriptors1, descriptors2);
            if (i >= 5) t[(size_t)(i - 5)] = now_ms() - t0;
        }
        stats(t, &up_med, &up_mean);
        std::printf("{\"extract_ms\": %.4f, \"extract_mean_ms\": %.4f, \"extract_plus_match_ms\": %.4f, "
                    "\"extract_plus_match_mean_ms\": %.4f, \"with_matches_clear_ms\": %.4f, "
                    "\"extract_plus_match_ratio_ms\": %.4f, \"host_matrix_match_ms\": %.4f, "
                    "\"keypoints\": %d, \"match_rows\": %d, \"good_matches\": %d}\n",
                    ext_med, ext_mean, loop_med, loop_mean, clr_med, ratio_med, up_med, (int)keypoints2.size(),
                    (int)nmatch, (int)goodMatches.size());
    } catch (const slamfe::Error& e) {
        std::fprintf(stderr, "slamfe::Error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
