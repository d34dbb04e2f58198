/* slamfe_dev.h -- structures shared by the host library and the HIP kernels. */
#ifndef SLAMFE_DEV_H
#define SLAMFE_DEV_H

#include <stdint.h>

#define SLAMFE_MAX_LEVELS 8
#define SLAMFE_MAX_DIM 4096

/* FAST / blur tiles: 64 x 16 output pixels per workgroup of 256 threads.
 * After 3x3 non-max suppression no two survivors are 8-neighbours, so a tile
 * holds at most (64/2) * (16/2) = 256 of them. */
#define SLAMFE_TILE_W 64
#define SLAMFE_TILE_H 16
#define SLAMFE_TILE_CAND 256
/* The FAST + blur kernel works on groups of vertically adjacent tiles (one
 * workgroup per group): the 8 halo rows a tile needs are then shared. */
#define SLAMFE_FB_TILES 2

/* One pyramid level of one frame.  All frames of a batch share the geometry;
 * frame f, level l starts at pyr + f * frame_bytes + off. */
typedef struct slamfe_dev_level {
    int32_t w, h;        /* level size in pixels */
    int32_t pitch;       /* bytes per row (multiple of 64) */
    int32_t quota;       /* retain-best quota (INT32_MAX: keep all) */
    uint32_t off;        /* byte offset of the level inside a frame's pyramid */
    int32_t tiles_x, tiles_y;
    int32_t tile_base;   /* index of the level's first tile in the frame's tile list */
    int32_t band_base;   /* index of the level's first tile row ("band") */
    float scale;         /* level scale factor (1, 1.2, 1.44, ...) */
    uint32_t tap_x, tap_y; /* offsets (in uint32 entries) of the level's resize taps */
    int32_t group_base;  /* index of the level's first tile group (SLAMFE_FB_TILES tile rows each) */
} slamfe_dev_level;

typedef struct slamfe_dev_geom {
    int32_t nlevels;
    int32_t total_tiles;
    int32_t total_bands;
    int32_t total_groups;
    uint32_t frame_bytes; /* bytes of one frame's pyramid */
    /* one-launch pyramid (k_pyramid_fused): level-0 tiles of SLAMFE_PT_W x SLAMFE_PT_H
     * pixels; pt_x = 0 if the geometry does not fit its LDS budget */
    int32_t pt_x, pt_y;        /* tiles per axis */
    uint32_t ptab_x, ptab_y;   /* offsets (uint32 entries) of its range tables in the tap buffer */
    uint32_t pt_lds;           /* dynamic LDS bytes */
    uint32_t pt_off_taps, pt_off_a, pt_off_b; /* byte offsets inside that LDS: tap cache, the two level buffers */
    slamfe_dev_level lv[SLAMFE_MAX_LEVELS];
} slamfe_dev_geom;

#ifndef SLAMFE_PT_W
#define SLAMFE_PT_W 128
#endif
#ifndef SLAMFE_PT_H
#define SLAMFE_PT_H 32
#endif
#define SLAMFE_PT_LDS_MAX (96 * 1024)

/* Candidate / keypoint key: y in bits 31..20, x in bits 19..8, score in 7..0,
 * so ascending keys are in (y, x) order. */
#define SLAMFE_KEY(x, y, s) (((uint32_t)(y) << 20) | ((uint32_t)(x) << 8) | (uint32_t)(s))
#define SLAMFE_KEY_X(k) (((k) >> 8) & 0xfffu)
#define SLAMFE_KEY_Y(k) ((k) >> 20)
#define SLAMFE_KEY_S(k) ((k) & 0xffu)

/* Match keys inside one call: distance in bits 31..23, train row in 22..0. */
#define SLAMFE_MKEY_SHIFT 23
#define SLAMFE_MKEY_NONE 0xffffffffu
#define SLAMFE_MAX_TRAIN (1 << SLAMFE_MKEY_SHIFT)

#endif
