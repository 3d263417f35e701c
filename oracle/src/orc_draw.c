/*
 * ORACLE (test infrastructure) -- the OpenCV 4.5.5 rasterisers the hot path reaches ([ext], parity
 * unpinned; restated from drawing.cpp):
 *   cv2.circle(.., thickness=-1)                      util_cylinder.py:1877   -> Circle (midpoint)
 *   cv2.drawContours(mask,[hull],-1,255,thickness=-1) :1896  -> CollectPolyEdges (outline by Line) +
 *                                                              FillEdgeCollection (scan-line fill)
 *   cv2.ellipse(mask, c, axes, 0, 0, 360, 0, -1)      :1992  -> ellipse2Poly (delta from the axes,
 *                                                              f32 sine table) + FillConvexPoly (+ Line2)
 * All on single-channel u8 images.
 */
#include "orc_common.h"

#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

static inline void put(uint8_t *img, int h, int w, int x, int y, uint8_t c)
{
    if (x >= 0 && x < w && y >= 0 && y < h) img[(size_t)y * w + x] = c;
}
static inline void hline(uint8_t *img, int h, int w, int y, int x1, int x2, uint8_t c)
{
    if (y < 0 || y >= h) return;
    if (x1 < 0) x1 = 0;
    if (x2 >= w) x2 = w - 1;
    for (int x = x1; x <= x2; x++) img[(size_t)y * w + x] = c;
}

/* Line(): 8-connected Bresenham of LineIterator(leftToRight = true).  Pixels outside are skipped
 * (OpenCV clips the segment first; for an 8-connected Bresenham the visible pixels are the same
 * whenever the clipped ends stay on the ideal line's raster -- true for the hull outlines here,
 * whose vertices all lie inside the image). */
ORC_API void orc_line(uint8_t *img, int h, int w, int x1, int y1, int x2, int y2, uint8_t c)
{
    int dx = x2 - x1, dy = y2 - y1;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
    int sx = 1, sy = dy < 0 ? -1 : 1;
    int ady = dy < 0 ? -dy : dy;
    int x = x1, y = y1;
    if (ady > dx) { /* steep: major axis y */
        int err = ady - (dx + dx), plus = ady + ady, minus = -(dx + dx);
        for (int i = 0; i <= ady; i++) {
            put(img, h, w, x, y, c);
            int m = err < 0;
            err += minus + (m ? plus : 0);
            y += sy;
            if (m) x += sx;
        }
    } else {
        int err = dx - (ady + ady), plus = dx + dx, minus = -(ady + ady);
        for (int i = 0; i <= dx; i++) {
            put(img, h, w, x, y, c);
            int m = err < 0;
            err += minus + (m ? plus : 0);
            x += sx;
            if (m) y += sy;
        }
    }
}

/* Line2(): fixed-point DDA between sub-pixel end points (16.16) */
static void line2(uint8_t *img, int h, int w, int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y, uint8_t c)
{
    int64_t dx = p2x - p1x, dy = p2y - p1y;
    int64_t j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
    int64_t i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
    int64_t x_step, y_step;
    int ecount;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        p1x ^= p2x & j; p2x ^= p1x & j; p1x ^= p2x & j;
        p1y ^= p2y & j; p2y ^= p1y & j; p1y ^= p2y & j;
        x_step = XY_ONE;
        y_step = dy * (1 << XY_SHIFT) / (ax | 1);
        ecount = (int)((p2x - p1x) >> XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        p1x ^= p2x & i; p2x ^= p1x & i; p1x ^= p2x & i;
        p1y ^= p2y & i; p2y ^= p1y & i; p1y ^= p2y & i;
        x_step = dx * (1 << XY_SHIFT) / (ay | 1);
        y_step = XY_ONE;
        ecount = (int)((p2y - p1y) >> XY_SHIFT);
    }
    p1x += (XY_ONE >> 1);
    p1y += (XY_ONE >> 1);
    put(img, h, w, (int)((p2x + (XY_ONE >> 1)) >> XY_SHIFT), (int)((p2y + (XY_ONE >> 1)) >> XY_SHIFT), c);
    if (ax > ay) {
        p1x >>= XY_SHIFT;
        while (ecount >= 0) {
            put(img, h, w, (int)p1x, (int)(p1y >> XY_SHIFT), c);
            p1x++;
            p1y += y_step;
            ecount--;
        }
    } else {
        p1y >>= XY_SHIFT;
        while (ecount >= 0) {
            put(img, h, w, (int)(p1x >> XY_SHIFT), (int)p1y, c);
            p1x += x_step;
            p1y++;
            ecount--;
        }
    }
}

/* Circle(img, centre, radius, colour, fill = 1) */
ORC_API void orc_circle_fill(uint8_t *img, int h, int w, int cx, int cy, int radius, uint8_t c)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        hline(img, h, w, y11, x11, x12, c);
        hline(img, h, w, y12, x11, x12, c);
        hline(img, h, w, y21, x21, x22, c);
        hline(img, h, w, y22, x21, x22, c);
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* drawContours(img, [poly], -1, c, thickness=-1): integer vertices */
typedef struct { int y0, y1; int64_t x, dx; } poly_edge;
static int edge_cmp(const void *a, const void *b)
{
    const poly_edge *e1 = (const poly_edge *)a, *e2 = (const poly_edge *)b;
    if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
    if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
    if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
    return 0;
}

ORC_API void orc_fill_poly(uint8_t *img, int h, int w, const orc_pt *v, int n, uint8_t c)
{
    if (n == 0) return;
    poly_edge *edges = (poly_edge *)malloc((size_t)(n + 1) * sizeof(poly_edge));
    int ne = 0;
    int64_t p0x = (int64_t)v[n - 1].x << XY_SHIFT, p0y = v[n - 1].y;
    for (int i = 0; i < n; i++) {
        int64_t p1x = (int64_t)v[i].x << XY_SHIFT, p1y = v[i].y;
        orc_line(img, h, w, (int)((p0x + (XY_ONE >> 1)) >> XY_SHIFT), (int)p0y, (int)((p1x + (XY_ONE >> 1)) >> XY_SHIFT),
                 (int)p1y, c);
        if (p0y != p1y) {
            poly_edge e;
            if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = p0x; }
            else { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = p1x; }
            e.dx = (p1x - p0x) / (p1y - p0y);
            edges[ne++] = e;
        }
        p0x = p1x; p0y = p1y;
    }
    if (ne >= 2) {
        qsort(edges, ne, sizeof(poly_edge), edge_cmp);
        int y_max = edges[0].y1;
        for (int i = 1; i < ne; i++)
            if (edges[i].y1 > y_max) y_max = edges[i].y1;
        if (y_max > h) y_max = h;
        /* even-odd scan-line fill over the active edges, x kept per edge in 16.16 */
        int64_t *ax = (int64_t *)malloc((size_t)ne * sizeof(int64_t));
        for (int y = edges[0].y0; y < y_max; y++) {
            int na = 0;
            for (int i = 0; i < ne; i++)
                if (edges[i].y0 <= y && y < edges[i].y1) ax[na++] = edges[i].x + (int64_t)(y - edges[i].y0) * edges[i].dx;
            for (int a = 1; a < na; a++) { /* sort */
                int64_t k = ax[a];
                int b = a - 1;
                while (b >= 0 && ax[b] > k) { ax[b + 1] = ax[b]; b--; }
                ax[b + 1] = k;
            }
            if (y < 0) continue;
            for (int a = 0; a + 1 < na; a += 2) {
                int x1 = (int)((ax[a] + XY_ONE - 1) >> XY_SHIFT), x2 = (int)(ax[a + 1] >> XY_SHIFT);
                if (x1 < w && x2 >= 0) hline(img, h, w, y, x1, x2, c);
            }
        }
        free(ax);
    }
    free(edges);
}

/* FillConvexPoly with shift = XY_SHIFT (vertices in 16.16), line_type = 8 */
static void fill_convex_poly_fx(uint8_t *img, int h, int w, const int64_t *vx, const int64_t *vy, int npts, uint8_t c)
{
    const int shift = XY_SHIFT;
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    int delta = 1 << shift >> 1;
    int i, y, imin = 0, edges = npts;
    int64_t xmin, xmax, ymin, ymax;
    int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
    int64_t p0x = vx[npts - 1], p0y = vy[npts - 1];
    xmin = xmax = vx[0];
    ymin = ymax = vy[0];
    for (i = 0; i < npts; i++) {
        int64_t px = vx[i], py = vy[i];
        if (py < ymin) { ymin = py; imin = i; }
        if (py > ymax) ymax = py;
        if (px > xmax) xmax = px;
        if (px < xmin) xmin = px;
        line2(img, h, w, p0x, p0y, px, py, c);
        p0x = px; p0y = py;
    }
    xmin = (xmin + delta) >> shift;
    xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift;
    ymax = (ymax + delta) >> shift;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    if (ymax > h - 1) ymax = h - 1;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y = (int)ymin;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)((vy[idx] + delta) >> shift);
                    if (ty > y) {
                        int64_t xs = vx[idx0], xe = vx[idx];
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < w) hline(img, h, w, y, xx1, xx2, c);
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

static int cv_round(double v) { return (int)lrint(v); }

/* cv2.ellipse(img, (cx,cy), (a,b), 0, 0, 360, colour, thickness=-1): filled, axis-aligned */
ORC_API void orc_ellipse_fill(uint8_t *img, int h, int w, int cx, int cy, int a, int b, uint8_t c)
{
    int64_t ccx = (int64_t)cx << XY_SHIFT, ccy = (int64_t)cy << XY_SHIFT;
    int64_t aw = llabs((int64_t)a << XY_SHIFT), ah = llabs((int64_t)b << XY_SHIFT);
    int delta = (int)(((aw > ah ? aw : ah) + (XY_ONE >> 1)) >> XY_SHIFT);
    delta = delta < 3 ? 90 : delta < 10 ? 30 : delta < 15 ? 18 : 5;
    int64_t vx[400], vy[400];
    int nv = 0;
    int64_t prevx = -1, prevy = -1;
    int have_prev = 0;
    for (int i = 0; i < 360 + delta; i += delta) {
        int ang = i > 360 ? 360 : i;
        /* SinTable is a float table of sin(deg); cos(t) = SinTable[450 - t] */
        double sx = (double)(float)sin((450 - ang) * 0.017453292519943295769236907684886);
        double sy = (double)(float)sin(ang * 0.017453292519943295769236907684886);
        /* exact table entries at multiples of 90 */
        if ((450 - ang) % 90 == 0) { int q = ((450 - ang) / 90) % 4; sx = q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 0 : -1; }
        if (ang % 90 == 0) { int q = (ang / 90) % 4; sy = q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 0 : -1; }
        double x = (double)aw * sx, y = (double)ah * sy;
        /* angle = 0: alpha = 1, beta = 0 */
        double ptx = (double)ccx + x, pty = (double)ccy + y;
        int64_t qx = (int64_t)cv_round(ptx / XY_ONE) << XY_SHIFT, qy = (int64_t)cv_round(pty / XY_ONE) << XY_SHIFT;
        qx += cv_round(ptx - (double)qx);
        qy += cv_round(pty - (double)qy);
        if (!have_prev || qx != prevx || qy != prevy) {
            vx[nv] = qx; vy[nv] = qy; nv++;
            prevx = qx; prevy = qy; have_prev = 1;
        }
    }
    if (nv == 1) { vx[0] = vx[1] = ccx; vy[0] = vy[1] = ccy; nv = 2; }
    fill_convex_poly_fx(img, h, w, vx, vy, nv, c);
}
