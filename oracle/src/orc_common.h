/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference hot path
 * cv3vpl-lab/cylinder-pose-estimation:
 *     python_grid_detection_cylinder.py::detect_grid  ->  utils/util_cylinder.py
 *     utils/fitSingleCylinder.m -> chooseIdx.m / fitCylinderWPts3.m / getDistPts3ToLine.m ...
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (cylinder-pose-estimation_amd/) never links,
 * imports or calls it.
 *
 * Parity status: the Hessian part of the pre-process (skimage/scipy) and the
 * cv2-free host logic are PINNED by golden vectors generated from the real
 * reference functions (tests/golden/, tools/gen_golden.py).  Everything that
 * restates OpenCV / MATLAB-toolbox arithmetic ([ext] in SURVEY.md) is
 * "parity unpinned": neither library exists in this image.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_API __attribute__((visibility("default")))

typedef struct { int x, y; } orc_pt;

/* growable int point list */
typedef struct {
    orc_pt *p;
    int n, cap;
} orc_ptvec;

static inline void orc_ptvec_push(orc_ptvec *v, int x, int y)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->p = (orc_pt *)realloc(v->p, (size_t)v->cap * sizeof(orc_pt));
    }
    v->p[v->n].x = x;
    v->p[v->n].y = y;
    v->n++;
}

static inline int orc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* BORDER_REFLECT_101 index (OpenCV default border), valid for |overshoot| < n */
static inline int orc_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

#ifdef __cplusplus
}
#endif
#endif
