/* abi_batch_check.c -- plain C caller of the batch / resident entry points of
 * include/slamfe.h (the header is a C header; this file is built with gcc -std=c99).
 * Self-consistency only, no checker involved: the batch call, the per-frame
 * call, the download of a resident frame and the host / resident match calls
 * must agree bit for bit, and the error paths must answer as documented.
 * Also the program that tools/asan_host_check.sh runs under AddressSanitizer. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slamfe.h"

#define W 320
#define H 240
#define NF 5

#define CHECK(cond)                                                                         \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            fprintf(stderr, "abi_batch_check: line %d: %s failed (%s)\n", __LINE__, #cond, \
                    ctx ? slamfe_last_error(ctx) : "-");                                    \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd(void)
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

/* blocky noise plus rectangles: plenty of corners on every pyramid level */
static void make_frame(uint8_t* img, int shift)
{
    static uint8_t cells[(H / 8 + 2) * (W / 8 + 8)];
    int x, y, i;
    if (shift == 0)
        for (i = 0; i < (int)sizeof cells; ++i) cells[i] = (uint8_t)(rnd() & 255u);
    for (y = 0; y < H; ++y)
        for (x = 0; x < W; ++x) {
            int c = cells[(y / 8) * (W / 8 + 8) + (x + shift) / 8];
            img[y * W + x] = (uint8_t)(64 + c / 2 + (int)(rnd() & 3u));
        }
}

int main(void)
{
    slamfe_ctx* ctx = NULL;
    slamfe_params p;
    static uint8_t frames[NF][W * H];
    const uint8_t* ptrs[NF];
    int i, j, cap, rc;

    CHECK(slamfe_default_params(&p) == SLAMFE_OK);
    CHECK(p.struct_size == (int)sizeof(slamfe_params));
    p.mode = SLAMFE_MODE_ORB;
    p.fast_threshold = 20;
    p.nfeatures = 500;
    p.max_width = W;
    p.max_height = H;
    p.max_batch = NF;
    p.frame_slots = 2 * NF;
    p.map_capacity = 0;
    CHECK(slamfe_create(&p, &ctx) == SLAMFE_OK);
    for (i = 0; i < NF; ++i) {
        make_frame(frames[i], 3 * i);
        ptrs[i] = frames[i];
    }

    cap = 600;
    slamfe_keypoint* kps = (slamfe_keypoint*)malloc((size_t)NF * cap * sizeof *kps);
    uint8_t* desc = (uint8_t*)malloc((size_t)NF * cap * 32);
    slamfe_keypoint* kp1 = (slamfe_keypoint*)malloc((size_t)cap * sizeof *kp1);
    uint8_t* de1 = (uint8_t*)malloc((size_t)cap * 32);
    int n[NF], n2[NF], n1 = 0;
    uint64_t h[NF], h2[NF], h1 = 0;
    CHECK(kps && desc && kp1 && de1);

    /* batch call with results */
    CHECK(slamfe_extract_batch(ctx, ptrs, NF, W, H, W, 1, kps, desc, cap, n, h) == SLAMFE_OK);
    for (i = 0; i < NF; ++i) CHECK(n[i] > 200 && n[i] <= 500);

    /* the same frames one by one, and the resident copies of the batch */
    for (i = 0; i < NF; ++i) {
        CHECK(slamfe_extract(ctx, frames[i], W, H, W, 1, kp1, de1, cap, &n1, &h1) == SLAMFE_OK);
        CHECK(n1 == n[i]);
        CHECK(memcmp(kp1, kps + (size_t)i * cap, (size_t)n1 * sizeof *kp1) == 0);
        CHECK(memcmp(de1, desc + (size_t)i * cap * 32, (size_t)n1 * 32) == 0);
    }
    /* the ring holds 2 * NF frames: the batch's handles are still alive after NF more */
    for (i = 0; i < NF; ++i) {
        n1 = -1;
        CHECK(slamfe_frame_download(ctx, h[i], kp1, de1, cap, &n1) == SLAMFE_OK);
        CHECK(n1 == n[i]);
        CHECK(memcmp(kp1, kps + (size_t)i * cap, (size_t)n1 * sizeof *kp1) == 0);
        CHECK(memcmp(de1, desc + (size_t)i * cap * 32, (size_t)n1 * 32) == 0);
    }

    /* counts only: results stay on the device */
    CHECK(slamfe_extract_batch(ctx, ptrs, NF, W, H, W, 1, NULL, NULL, 0, n2, h2) == SLAMFE_OK);
    for (i = 0; i < NF; ++i) CHECK(n2[i] == n[i]);
    /* ... which pushed the first batch out of the ring */
    CHECK(slamfe_frame_download(ctx, h[0], kp1, de1, cap, &n1) == SLAMFE_E_HANDLE);

    /* resident match against the host-pointer match of the same descriptors */
    {
        slamfe_match2* ma = (slamfe_match2*)malloc((size_t)cap * sizeof *ma);
        slamfe_match2* mb = (slamfe_match2*)malloc((size_t)cap * sizeof *mb);
        int nq = -1, good = 0;
        CHECK(ma && mb);
        for (i = 0; i + 1 < NF; ++i) {
            CHECK(slamfe_match_frames(ctx, h2[i + 1], h2[i], ma, cap, &nq) == SLAMFE_OK);
            CHECK(nq == n[i + 1]);
            CHECK(slamfe_match_knn2(ctx, desc + (size_t)(i + 1) * cap * 32, n[i + 1],
                                    desc + (size_t)i * cap * 32, n[i], mb) == SLAMFE_OK);
            for (j = 0; j < nq; ++j) {
                CHECK(ma[j].idx0 == mb[j].idx0 && ma[j].idx1 == mb[j].idx1 && ma[j].d0 == mb[j].d0 &&
                      ma[j].d1 == mb[j].d1 && ma[j].ratio_pass == mb[j].ratio_pass &&
                      ma[j].nfound == 2);
                good += ma[j].ratio_pass;
            }
        }
        CHECK(good > 100);
        /* too small a result buffer: the needed count comes back */
        nq = -1;
        CHECK(slamfe_match_frames(ctx, h2[1], h2[0], ma, 10, &nq) == SLAMFE_E_CAPACITY);
        CHECK(nq == n[1]);
        /* one train row: a single neighbour, no ratio test */
        CHECK(slamfe_match_knn2(ctx, desc, 3, desc, 1, mb) == SLAMFE_OK);
        CHECK(mb[0].nfound == 1 && mb[0].idx0 == 0 && mb[0].idx1 == -1 && mb[0].d0 == 0 &&
              mb[0].ratio_pass == 0);
        free(ma);
        free(mb);
    }

    /* error paths */
    rc = slamfe_extract(ctx, frames[0], W, H, W, 1, kp1, de1, 50, &n1, &h1);
    CHECK(rc == SLAMFE_E_CAPACITY && n1 == n[0]);
    CHECK(strlen(slamfe_last_error(ctx)) > 0);
    CHECK(slamfe_extract(ctx, frames[0], W + 1, H, W + 1, 1, kp1, de1, cap, &n1, &h1) == SLAMFE_E_INVALID);
    CHECK(slamfe_extract(ctx, frames[0], W, H, W - 1, 1, kp1, de1, cap, &n1, &h1) == SLAMFE_E_INVALID);
    CHECK(slamfe_extract(ctx, frames[0], W, H, W, 2, kp1, de1, cap, &n1, &h1) == SLAMFE_E_INVALID);
    CHECK(slamfe_extract(ctx, NULL, W, H, W, 1, kp1, de1, cap, &n1, &h1) == SLAMFE_E_INVALID);
    CHECK(slamfe_extract_batch(ctx, ptrs, NF + 1, W, H, W, 1, kps, desc, cap, n, h) == SLAMFE_E_CAPACITY ||
          slamfe_extract_batch(ctx, ptrs, NF + 1, W, H, W, 1, kps, desc, cap, n, h) == SLAMFE_E_INVALID);
    CHECK(slamfe_frame_download(ctx, 0xdeadbeefULL, kp1, de1, cap, &n1) == SLAMFE_E_HANDLE);
    CHECK(slamfe_sync(ctx) == SLAMFE_OK);

    free(kps);
    free(desc);
    free(kp1);
    free(de1);
    slamfe_destroy(ctx);
    printf("abi batch check ok: %d %d %d %d %d keypoints\n", n[0], n[1], n[2], n[3], n[4]);
    return 0;
}
