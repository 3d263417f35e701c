/*
 * ORACLE (test infrastructure) -- cv2.findContours / moments / contourArea / boundingRect /
 * convexHull / minEnclosingCircle as OpenCV 4.5.5 computes them ([ext], parity unpinned: OpenCV
 * is absent from this image; restated from its published algorithm -- Suzuki & Abe border
 * following in the legacy C scanner, Green's-theorem polygon moments, Welzl-type circle in f32).
 * Call sites: util_cylinder.py:161, 1817-1825, 1883-1896, 1914, 1968-1980, 2019.
 *
 * The scanner is the LITERAL sequential algorithm (raster scan, nbd = 2 marks, -126 "right
 * bound" marks, lnbd test for RETR_EXTERNAL), including its quirks with 1-px-thin walls.
 * Contours are returned in OpenCV's order: most recently found first.
 */
#include "orc_common.h"

typedef struct {
    orc_ptvec pts;
    int is_hole;
    int ox, oy; /* origin (start pixel) */
} orc_contour;

typedef struct {
    orc_contour *c;
    int n, cap;
} orc_contours;

ORC_API void orc_contours_free(orc_contours *cs)
{
    if (!cs) return;
    for (int i = 0; i < cs->n; i++) free(cs->c[i].pts.p);
    free(cs->c);
    free(cs);
}

ORC_API int orc_contours_count(const orc_contours *cs) { return cs->n; }
ORC_API int orc_contour_size(const orc_contours *cs, int i) { return cs->c[i].pts.n; }
ORC_API int orc_contour_is_hole(const orc_contours *cs, int i) { return cs->c[i].is_hole; }
ORC_API const orc_pt *orc_contour_points(const orc_contours *cs, int i) { return cs->c[i].pts.p; }

static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* icvFetchContour: follow one border starting at (x0,y0) on the padded int8 image */
static void fetch_contour(int8_t *img, int step, int x0, int y0, int is_hole, int method_simple, orc_ptvec *out,
                          int offx, int offy)
{
    const int8_t nbd = 2;
    int deltas[16];
    for (int k = 0; k < 8; k++) deltas[k] = deltas[k + 8] = DY[k] * step + DX[k];
    int8_t *i0 = img + (size_t)y0 * step + x0, *i1, *i3, *i4 = 0;
    int s, s_end, prev_s;
    int px = x0, py = y0;
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) { /* single pixel */
        *i0 = (int8_t)(nbd | -128);
        orc_ptvec_push(out, px + offx, py + offy);
        return;
    }
    i3 = i0;
    prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        s = s < 15 ? s : 15;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)(nbd | -128);
        else if (*i3 == 1) *i3 = nbd;
        if (s != prev_s || !method_simple) {
            orc_ptvec_push(out, px + offx, py + offy);
            prev_s = s;
        }
        px += DX[s];
        py += DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* mode: 0 = RETR_EXTERNAL, 1 = RETR_LIST.  method: 1 = CHAIN_APPROX_NONE, 2 = CHAIN_APPROX_SIMPLE */
ORC_API orc_contours *orc_find_contours(const uint8_t *src, int h, int w, int mode, int method)
{
    int step = w + 2, H = h + 2;
    int8_t *img = (int8_t *)calloc((size_t)step * H, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = src[(size_t)y * w + x] ? 1 : 0;
    orc_contours *cs = (orc_contours *)calloc(1, sizeof(*cs));
    int lx, ly;
    for (int y = 1; y < H - 1; y++) {
        int8_t *row = img + (size_t)y * step;
        int prev = 0;
        lx = 0; ly = y;
        for (int x = 1; x < step - 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) goto resume;
                if (prev & -2) lx = x - 1;
                is_hole = 1;
            }
            if (mode == 0 && (is_hole || img[(size_t)ly * step + lx] > 0)) goto resume;
            {
                if (cs->n == cs->cap) {
                    cs->cap = cs->cap ? cs->cap * 2 : 64;
                    cs->c = (orc_contour *)realloc(cs->c, (size_t)cs->cap * sizeof(orc_contour));
                }
                orc_contour *c = &cs->c[cs->n++];
                memset(c, 0, sizeof(*c));
                c->is_hole = is_hole;
                c->ox = x - is_hole - 1;
                c->oy = y - 1;
                fetch_contour(img, step, x - is_hole, y, is_hole, method == 2, &c->pts, -1, -1);
                p = row[x]; /* the trace may have re-marked the current pixel */
            }
        resume:
            prev = p;
            if (prev & -2) lx = x;
        }
    }
    free(img);
    /* OpenCV returns the most recently found contour first */
    for (int i = 0, j = cs->n - 1; i < j; i++, j--) {
        orc_contour t = cs->c[i]; cs->c[i] = cs->c[j]; cs->c[j] = t;
    }
    return cs;
}

/* cv2.moments(contour): m00, m10, m01 (contourMoments) */
ORC_API void orc_contour_moments(const orc_pt *p, int n, double *m00, double *m10, double *m01)
{
    *m00 = *m10 = *m01 = 0;
    if (n == 0) return;
    double a00 = 0, a10 = 0, a01 = 0;
    double xi_1 = p[n - 1].x, yi_1 = p[n - 1].y;
    for (int i = 0; i < n; i++) {
        double xi = p[i].x, yi = p[i].y;
        double dxy = xi_1 * yi - xi * yi_1;
        double xii_1 = xi_1 + xi, yii_1 = yi_1 + yi;
        a00 += dxy;
        a10 += dxy * xii_1;
        a01 += dxy * yii_1;
        xi_1 = xi; yi_1 = yi;
    }
    if (fabs(a00) > 1.1920928955078125e-07 /* FLT_EPSILON */) {
        double db1_2, db1_6;
        if (a00 > 0) { db1_2 = 0.5; db1_6 = 0.16666666666666666666666666666667; }
        else { db1_2 = -0.5; db1_6 = -0.16666666666666666666666666666667; }
        *m00 = a00 * db1_2;
        *m10 = a10 * db1_6;
        *m01 = a01 * db1_6;
    }
}

/* cv2.contourArea(contour) (oriented = false) */
ORC_API double orc_contour_area(const orc_pt *p, int n)
{
    if (n == 0) return 0;
    double a00 = 0;
    double px = p[n - 1].x, py = p[n - 1].y;
    for (int i = 0; i < n; i++) {
        double x = p[i].x, y = p[i].y;
        a00 += px * y - py * x;
        px = x; py = y;
    }
    a00 *= 0.5;
    return fabs(a00);
}

/* cv2.boundingRect(points) -> x, y, w, h */
ORC_API void orc_bounding_rect(const orc_pt *p, int n, int *r)
{
    if (n == 0) { r[0] = r[1] = r[2] = r[3] = 0; return; }
    int x0 = p[0].x, x1 = p[0].x, y0 = p[0].y, y1 = p[0].y;
    for (int i = 1; i < n; i++) {
        if (p[i].x < x0) x0 = p[i].x;
        if (p[i].x > x1) x1 = p[i].x;
        if (p[i].y < y0) y0 = p[i].y;
        if (p[i].y > y1) y1 = p[i].y;
    }
    r[0] = x0; r[1] = y0; r[2] = x1 - x0 + 1; r[3] = y1 - y0 + 1;
}

static int pt_cmp(const void *a, const void *b)
{
    const orc_pt *p = (const orc_pt *)a, *q = (const orc_pt *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}
static long long cross(orc_pt o, orc_pt a, orc_pt b)
{
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

/* cv2.convexHull(points): strict hull vertices (no collinear points), as a closed polygon.
 * Only the vertex SET and cyclic order matter downstream (filled with drawContours, :1896). */
ORC_API int orc_convex_hull(const orc_pt *pin, int n, orc_pt *hull)
{
    orc_pt *p = (orc_pt *)malloc((size_t)n * sizeof(orc_pt));
    memcpy(p, pin, (size_t)n * sizeof(orc_pt));
    qsort(p, n, sizeof(orc_pt), pt_cmp);
    int m = 0;
    for (int i = 0; i < n; i++)
        if (m == 0 || pt_cmp(&p[i], &p[m - 1]) != 0) p[m++] = p[i];
    n = m;
    if (n <= 2) {
        for (int i = 0; i < n; i++) hull[i] = p[i];
        free(p);
        return n;
    }
    int k = 0;
    for (int i = 0; i < n; i++) {
        while (k >= 2 && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--;
        hull[k++] = p[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; i--) {
        while (k >= t && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--;
        hull[k++] = p[i];
    }
    free(p);
    return k - 1;
}

/* ---- cv2.minEnclosingCircle (f32, sequential over the contour points; OpenCV 4.5.5 shapedescr.cpp) */
static const float MEC_EPS = 1.0e-4f;
static float normf2(float x, float y) { return (float)sqrt((double)x * x + (double)y * y); }

static void circle3(const float *px, const float *py, float *cx_, float *cy_, float *r_)
{
    float v1x = px[1] - px[0], v1y = py[1] - py[0];
    float v2x = px[2] - px[0], v2y = py[2] - py[0];
    float m1x = (px[0] + px[1]) / 2.0f, m1y = (py[0] + py[1]) / 2.0f;
    float c1 = m1x * v1x + m1y * v1y;
    float m2x = (px[0] + px[2]) / 2.0f, m2y = (py[0] + py[2]) / 2.0f;
    float c2 = m2x * v2x + m2y * v2y;
    float det = v1x * v2y - v1y * v2x;
    if (fabsf(det) <= MEC_EPS) {
        float d1 = (px[0] - px[1]) * (px[0] - px[1]) + (py[0] - py[1]) * (py[0] - py[1]);
        float d2 = (px[0] - px[2]) * (px[0] - px[2]) + (py[0] - py[2]) * (py[0] - py[2]);
        float d3 = (px[1] - px[2]) * (px[1] - px[2]) + (py[1] - py[2]) * (py[1] - py[2]);
        float mx = d1 > d2 ? d1 : d2;
        mx = mx > d3 ? mx : d3;
        *r_ = sqrtf(mx) * 0.5f + MEC_EPS;
        if (d1 >= d2 && d1 >= d3) { *cx_ = (px[0] + px[1]) * 0.5f; *cy_ = (py[0] + py[1]) * 0.5f; }
        else if (d2 >= d1 && d2 >= d3) { *cx_ = (px[0] + px[2]) * 0.5f; *cy_ = (py[0] + py[2]) * 0.5f; }
        else { *cx_ = (px[1] + px[2]) * 0.5f; *cy_ = (py[1] + py[2]) * 0.5f; }
        return;
    }
    float cx = (c1 * v2y - c2 * v1y) / det;
    float cy = (v1x * c2 - v2x * c1) / det;
    *cx_ = cx; *cy_ = cy;
    cx -= px[0]; cy -= py[0];
    *r_ = (float)sqrt((double)(cx * cx + cy * cy)) + MEC_EPS;
}

static void third_point(const orc_pt *p, int i, int j, float *cx, float *cy, float *r)
{
    *cx = (float)(p[j].x + p[i].x) / 2.0f;
    *cy = (float)(p[j].y + p[i].y) / 2.0f;
    *r = normf2((float)(p[j].x - p[i].x), (float)(p[j].y - p[i].y)) / 2.0f + MEC_EPS;
    for (int k = 0; k < j; k++) {
        float dx = *cx - (float)p[k].x, dy = *cy - (float)p[k].y;
        if (normf2(dx, dy) < *r) continue;
        float fx[3] = {(float)p[i].x, (float)p[j].x, (float)p[k].x}, fy[3] = {(float)p[i].y, (float)p[j].y, (float)p[k].y};
        float ncx, ncy, nr = 0;
        circle3(fx, fy, &ncx, &ncy, &nr);
        if (nr > 0) { *r = nr; *cx = ncx; *cy = ncy; }
    }
}

static void second_point(const orc_pt *p, int i, float *cx, float *cy, float *r)
{
    *cx = (float)(p[0].x + p[i].x) / 2.0f;
    *cy = (float)(p[0].y + p[i].y) / 2.0f;
    *r = normf2((float)(p[0].x - p[i].x), (float)(p[0].y - p[i].y)) / 2.0f + MEC_EPS;
    for (int j = 1; j < i; j++) {
        float dx = *cx - (float)p[j].x, dy = *cy - (float)p[j].y;
        if (normf2(dx, dy) < *r) continue;
        float ncx, ncy, nr = 0;
        third_point(p, i, j, &ncx, &ncy, &nr);
        if (nr > 0) { *r = nr; *cx = ncx; *cy = ncy; }
    }
}

ORC_API void orc_min_enclosing_circle(const orc_pt *p, int n, float *cx, float *cy, float *r)
{
    *cx = *cy = *r = 0;
    if (n == 0) return;
    if (n == 1) { *cx = (float)p[0].x; *cy = (float)p[0].y; *r = MEC_EPS; return; }
    if (n == 2) {
        *cx = ((float)p[0].x + (float)p[1].x) / 2.0f;
        *cy = ((float)p[0].y + (float)p[1].y) / 2.0f;
        double dx = p[0].x - p[1].x, dy = p[0].y - p[1].y;
        *r = (float)(sqrt(dx * dx + dy * dy) / 2.0) + MEC_EPS;
        return;
    }
    *cx = (float)(p[0].x + p[1].x) / 2.0f;
    *cy = (float)(p[0].y + p[1].y) / 2.0f;
    *r = normf2((float)(p[0].x - p[1].x), (float)(p[0].y - p[1].y)) / 2.0f + MEC_EPS;
    for (int i = 2; i < n; i++) {
        float dx = (float)p[i].x - *cx, dy = (float)p[i].y - *cy;
        float d = normf2(dx, dy);
        if (d < *r) continue;
        float ncx, ncy, nr = 0;
        second_point(p, i, &ncx, &ncy, &nr);
        if (nr > 0) { *r = nr; *cx = ncx; *cy = ncy; }
    }
}
