// Threshold sweep of the blob detector (SimpleBlobDetector's 17 binarisations, util_cylinder.py:1857-1864): constants
// and bookkeeping shared by sweep.hip (component lists) and region.hip (border following, blobs, grouping).
#pragma once
#include "cpe_dev.h"

namespace cpe {

constexpr int NTHR = 17;          // thresholds 50, 60, ..., 210 (SimpleBlobDetector defaults, util_cylinder.py:1836)
constexpr int NBK = NTHR + 1;     // grey-level buckets: 0: v <= 50, b: 50 + 10 (b - 1) < v <= 50 + 10 b, 17: v > 210
__host__ __device__ inline int sw_level(int v) { return v <= 50 ? 0 : (((v - 41) / 10) < NTHR ? (v - 41) / 10 : NTHR); }

// per-frame int counters (SW_STRIDE ints per frame)
constexpr int SW_STRIDE = 192;
enum {
    SW_NH = 8,                    // + k: dark components away from the rectangle border at threshold k (length of hl[k])
    SW_NL = SW_NH + NTHR,         // + k: bright components at threshold k (length of bl[k])
    SW_NB = SW_NL + NTHR,         // + k: blobs of threshold k
    SW_ND = SW_NB + NTHR,         // + k: border distances stored for threshold k
    SW_PN = SW_ND + NTHR,         // entries in use of the sweep's event pool
    SW_BN = SW_PN + 1,            // + type * NTHR + k: batches of type (0: unions, 1: open local components) at threshold k
    SW_HT = SW_BN + 2 * NTHR,     // holes binned by tile (total)
    SW_NT = SW_HT + 1,            // + k: holes of threshold k whose border is followed (length of tl[k])
    SW_NC = SW_NT + NTHR,         // + k: border-point chunks in use
    SW_NA = SW_NC + NTHR          // + k: blobs of threshold k that came from hole borders (they are listed first)
};
static_assert(SW_NA + NTHR <= SW_STRIDE, "sweep counters");

constexpr int SW_TILE = 64;       // side of the LDS-resident tiles of the sweep
__host__ __device__ inline int sw_tiles_x(int w) { return (w + SW_TILE - 1) / SW_TILE; }
__host__ __device__ inline int sw_tiles_y(int h) { return (h + SW_TILE - 1) / SW_TILE; }

struct SweepBuffers {
    const uint8_t *cl;            // CLAHE'd L channel [n][h][w]
    int *G;                       // global union-find over pixel indices (touched only at tile-border pixels and local roots)
    int *acc;                     // per-root accumulator (same sparsity)
    uint8_t *tch;                 // dark: root of a component that reaches the working rectangle's border (epoch-marked)
    int *sw;                      // counters [n][SW_STRIDE]
    int2 *pool; int pool_cap;     // event pool per frame (unions to apply / open local components), pool_cap entries
    int2 *bh; int bcap;           // batch headers [n][2][NTHR][bcap]: (first pool entry, entries)
    int *hb_off;                  // holes binned by the tile of their west pixel: offsets [n][tiles + 1] (+ cursors [n][tiles])
    int2 *hb_ent; int hb_cap;     // ... entries (west pixel inside its tile | threshold << 12, pixels of the hole)
    int dbg;                      // timing aid (CPE_SW_DBG): stop the tile pass early
    int2 *hl, *bl, *tl;           // results [n][NTHR][sweep_cap]: holes / bright components / holes whose border is followed
};

// component lists of all 17 thresholds: hl[k] = (first pixel, pixels) of every enclosed dark component (4-connected),
// tl[k] the ones worth following (3 < pixels < 5000), bl[k] = (first pixel, pixels of the holes it encloses, each capped at
// 5000) of every bright component (8-connected).  sweep_dark first; sweep_bright needs its hole lists.
int sweep_dark(const SweepBuffers &B, int n, int h, int w, FrameState *st, hipStream_t s);
int sweep_bright(const SweepBuffers &B, int n, int h, int w, FrameState *st, hipStream_t s);

}  // namespace cpe
