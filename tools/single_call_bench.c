/* single_call_bench.c -- latency of the host-pointer calls as a C host sees them
 * (the reference's tracking thread is C++ and calls once per frame, Init.cc:130-131):
 * slamfe_extract on one frame, and the loop "extract the new frame, match the
 * previous one against it".  Frames come from raw gray files written by the caller.
 * usage: single_call_bench frameA.raw frameB.raw width height mode fast_threshold nfeatures reps
 * prints one JSON object with median and mean milliseconds. */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "slamfe.h"

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int cmp_double(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : x > y;
}

static uint8_t* read_file(const char* path, size_t n)
{
    uint8_t* buf = (uint8_t*)malloc(n);
    FILE* f = fopen(path, "rb");
    if (!buf || !f || fread(buf, 1, n, f) != n) {
        fprintf(stderr, "cannot read %s\n", path);
        exit(2);
    }
    fclose(f);
    return buf;
}

static void stats(double* t, int n, double* median, double* mean)
{
    double s = 0;
    int i;
    for (i = 0; i < n; ++i) s += t[i];
    qsort(t, (size_t)n, sizeof *t, cmp_double);
    *median = t[n / 2];
    *mean = s / n;
}

int main(int argc, char** argv)
{
    if (argc != 9) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]), mode = atoi(argv[5]), thr = atoi(argv[6]),
              nfeat = atoi(argv[7]), reps = atoi(argv[8]);
    uint8_t* a = read_file(argv[1], (size_t)w * h);
    uint8_t* b = read_file(argv[2], (size_t)w * h);
    slamfe_params p;
    slamfe_ctx* ctx = NULL;
    slamfe_default_params(&p);
    p.mode = mode;
    p.fast_threshold = thr;
    p.nfeatures = nfeat;
    p.max_width = w;
    p.max_height = h;
    p.map_capacity = 0;
    if (slamfe_create(&p, &ctx) != SLAMFE_OK) {
        fprintf(stderr, "slamfe_create: %s\n", ctx ? slamfe_last_error(ctx) : "failed");
        return 3;
    }
    const int cap = mode == SLAMFE_MODE_ORB ? (2 * nfeat > 4096 ? 2 * nfeat : 4096) : 32768;
    slamfe_keypoint* kps = (slamfe_keypoint*)malloc((size_t)cap * sizeof *kps);
    uint8_t* desc = (uint8_t*)malloc((size_t)cap * 32);
    slamfe_match2* rows = (slamfe_match2*)malloc((size_t)cap * sizeof *rows);
    double* t = (double*)malloc((size_t)reps * sizeof *t);
    int n = 0, nq = 0, i, rc = 0;
    uint64_t prev = 0, cur = 0;
    for (i = 0; i < 10 && !rc; ++i) rc = slamfe_extract(ctx, a, w, h, w, 1, kps, desc, cap, &n, &prev);
    for (i = 0; i < reps && !rc; ++i) {
        double t0 = now_ms();
        rc = slamfe_extract(ctx, a, w, h, w, 1, kps, desc, cap, &n, &prev);
        t[i] = now_ms() - t0;
    }
    double ext_med, ext_mean, loop_med, loop_mean;
    stats(t, reps, &ext_med, &ext_mean);
    for (i = 0; i < reps + 5 && !rc; ++i) { /* Init.cc:130-131 */
        double t0 = now_ms();
        rc = slamfe_extract(ctx, b, w, h, w, 1, kps, desc, cap, &n, &cur);
        if (!rc) rc = slamfe_match_frames(ctx, prev, cur, rows, cap, &nq);
        prev = cur;
        if (i >= 5) t[i - 5] = now_ms() - t0;
    }
    if (rc) {
        fprintf(stderr, "call failed (%d): %s\n", rc, slamfe_last_error(ctx));
        return 4;
    }
    stats(t, reps, &loop_med, &loop_mean);
    printf("{\"extract_ms\": %.4f, \"extract_mean_ms\": %.4f, \"extract_plus_match_ms\": %.4f, "
           "\"extract_plus_match_mean_ms\": %.4f, \"keypoints\": %d, \"match_rows\": %d}\n",
           ext_med, ext_mean, loop_med, loop_mean, n, nq);
    slamfe_destroy(ctx);
    return 0;
}
