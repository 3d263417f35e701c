/*
 * ORACLE (test infrastructure) -- detect_grid end to end for one grey frame
 *   python_grid_detection_cylinder.py:68-112 and util_cylinder.color_and_expand_lines (:2014-2060)
 * Output: centre point + table (x, y, col, row) sorted by (col,row), i.e. what make_json serialises
 * (util_cylinder.py:1674-1727) and makePyGridPts.m:39-41 decodes to an N x 4 matrix.
 */
#include "orc_common.h"

#include "orc_lines.h"

void orc_preprocess(const uint8_t *gray, int h, int w, uint8_t *blurred, uint8_t *mask, double *b_out);
int orc_extract_joints(const uint8_t *binary, int h, int w, uint8_t *hmask, uint8_t *vmask, int *cent, int cap);
int orc_detect_largest_blob(const uint8_t *gray, int h, int w, double clip, uint8_t *mask, int *rect, uint8_t *cl_out,
                            int *nkp_out);
int orc_mask_roi_around_center(const uint8_t *hmask, const uint8_t *vmask, const uint8_t *mask_contour,
                               const uint8_t *gray, int h, int w, uint8_t *roi_h, uint8_t *roi_v, int *r0, int *spot);
void orc_expand_line_roi(const uint8_t *mask_roi, const uint8_t *mask_contour, int h, int w, int kernel_size,
                         uint8_t *out, int *dbg);
int orc_connected_components(const uint8_t *mask, int h, int w, int32_t *labels);
void orc_blur7(const uint8_t *src, int h, int w, uint8_t *dst);
void orc_group_points(const int *cent, int n, const int32_t *labels, int lh, int lw, int x_off, int y_off,
                      orc_lineset *out);
void orc_fit_lines(orc_lineset *ls, int is_row);
void orc_remove_label(orc_lineset *rows, orc_lineset *cols);
void orc_intersections(orc_lineset *rows, orc_lineset *cols, const int *rect);
void orc_clean_and_relabel(orc_lineset *rows, orc_lineset *cols);
int orc_subpixel_refine(const uint8_t *gray, int h, int w, orc_lineset *rows, orc_lineset *cols, int window, double step);
int orc_index_points(const orc_lineset *rows, const orc_lineset *cols, const uint8_t *gauss7, int h, int w, int r0,
                     double *center, double *xy, int *id, int cap);

typedef struct {
    /* optional intermediate images (h*w each) for stage-by-stage parity tests; any may be NULL */
    uint8_t *binary, *hmask, *vmask, *mask_contour, *roi_h, *roi_v, *exp_h, *exp_v;
    int *joints;      /* cap_joints x 2 : all joints (contour order) */
    int cap_joints;
    int n_joints;     /* out */
    int n_cyl_joints; /* out: joints inside the bounding rect */
    int rect[4];      /* out */
    int r0;           /* out: circle_radius0 */
    int spot[4];      /* out: ellipse cx, cy, a, b */
    int n_rows, n_cols; /* out: lines after clean_and_relabel */
    int n_keypoints;    /* out */
    orc_lineset *rows_out, *cols_out; /* optional: rows_updated / cols_updated after clean_and_relabel (:2044) */
} orc_detect_debug;

/* returns status: 0 ok, 1 no region, 2 no spot, 3 no rows/cols, 4 empty */
ORC_API int orc_detect_grid_ex(const uint8_t *gray, int h, int w, int subpixel, int sp_window, double sp_step, double *center,
                               double *xy, int *id, int cap, int *n_out, orc_detect_debug *dbg);

ORC_API int orc_detect_grid(const uint8_t *gray, int h, int w, double *center, double *xy, int *id, int cap,
                            int *n_out, orc_detect_debug *dbg)
{
    return orc_detect_grid_ex(gray, h, w, 0, 7, 1.0, center, xy, id, cap, n_out, dbg);
}

static int detect_grid_impl(const uint8_t *gray, const uint8_t *bgr, int h, int w, int subpixel, int sp_window, double sp_step, double *center,
                            double *xy, int *id, int cap, int *n_out, orc_detect_debug *dbg);

/* subpixel != 0: modify_grayscale_Cline between remove_label and the intersections (the call commented out at :2040) */
ORC_API int orc_detect_grid_ex(const uint8_t *gray, int h, int w, int subpixel, int sp_window, double sp_step, double *center,
                               double *xy, int *id, int cap, int *n_out, orc_detect_debug *dbg)
{
    return detect_grid_impl(gray, NULL, h, w, subpixel, sp_window, sp_step, center, xy, id, cap, n_out, dbg);
}

void orc_bgr2gray(const uint8_t *bgr, size_t npx, uint8_t *gray);
void orc_lab_l_bgr(const uint8_t *bgr, int h, int w, uint8_t *L);
int orc_detect_largest_blob_l(const uint8_t *gray, const uint8_t *l_in, int h, int w, double clip, uint8_t *mask, int *rect,
                              uint8_t *cl_out, int *nkp_out);

/* detect_grid on a true-colour frame (H x W x 3 BGR, the CLI's cv2.imread): load_and_preprocess_image and
 * mask_roi_around_center work on BGR2GRAY of it (util_cylinder.py:1781-1789, :1957); detect_largest_blob takes the L channel
 * of BGR2LAB of the COLOUR image (:1840); indexing_data blurs the colour image 7x7 channel by channel and converts the
 * result to grey (:1433-1435). */
ORC_API int orc_detect_grid_bgr(const uint8_t *bgr, int h, int w, double *center, double *xy, int *id, int cap, int *n_out,
                                orc_detect_debug *dbg)
{
    uint8_t *gray = (uint8_t *)malloc((size_t)h * w);
    orc_bgr2gray(bgr, (size_t)h * w, gray);
    const int st = detect_grid_impl(gray, bgr, h, w, 0, 7, 1.0, center, xy, id, cap, n_out, dbg);
    free(gray);
    return st;
}

static int detect_grid_impl(const uint8_t *gray, const uint8_t *bgr, int h, int w, int subpixel, int sp_window, double sp_step, double *center,
                            double *xy, int *id, int cap, int *n_out, orc_detect_debug *dbg)
{
    size_t N = (size_t)h * w;
    uint8_t *blurred = (uint8_t *)malloc(N), *binary = (uint8_t *)malloc(N);
    uint8_t *hmask = (uint8_t *)malloc(N), *vmask = (uint8_t *)malloc(N), *mc = (uint8_t *)malloc(N);
    uint8_t *roi_h = (uint8_t *)malloc(N), *roi_v = (uint8_t *)malloc(N);
    uint8_t *exp_h = (uint8_t *)malloc(N), *exp_v = (uint8_t *)malloc(N), *g7 = (uint8_t *)malloc(N);
    int capj = 1 << 16;
    int *cent = (int *)malloc((size_t)capj * 2 * sizeof(int)), *cyl = (int *)malloc((size_t)capj * 2 * sizeof(int));
    orc_lineset *rows = (orc_lineset *)malloc(sizeof(orc_lineset)), *cols = (orc_lineset *)malloc(sizeof(orc_lineset));
    int32_t *lab_h = NULL, *lab_v = NULL;
    uint8_t *crop = NULL;
    int st = 0;
    *n_out = 0;
    orc_capacity_overflow = 0;

    /* 1 */ orc_preprocess(gray, h, w, blurred, binary, NULL);
    /* 2 */ int nj = orc_extract_joints(binary, h, w, hmask, vmask, cent, capj);
    if (nj > capj) nj = capj;
    int rect[4] = {0, 0, 0, 0}, r0 = 0, spot[4] = {0, 0, 0, 0}, nkp = 0;
    uint8_t *lplane = NULL;
    if (bgr) { lplane = (uint8_t *)malloc(N); orc_lab_l_bgr(bgr, h, w, lplane); }
    /* 3 */ st = orc_detect_largest_blob_l(gray, lplane, h, w, 4.5, mc, rect, NULL, &nkp);
    free(lplane);
    int ncyl = 0;
    if (st == 0) {
        /* 4: keep joints inside boundingRect(max_contour) (half-open, :1918) */
        for (int i = 0; i < nj; i++) {
            int cx = cent[2 * i], cy = cent[2 * i + 1];
            if (rect[0] <= cx && cx < rect[0] + rect[2] && rect[1] <= cy && cy < rect[1] + rect[3]) {
                cyl[2 * ncyl] = cx; cyl[2 * ncyl + 1] = cy; ncyl++;
            }
        }
        if (ncyl > CPE_MAXJ) { orc_capacity_overflow = 1; ncyl = CPE_MAXJ; }   /* include/cpe.h: capacity of the joint table */
        /* 5 */ st = orc_mask_roi_around_center(hmask, vmask, mc, gray, h, w, roi_h, roi_v, &r0, spot);
    }
    if (st == 0) {
        /* 6 */
        int ks = 91 + r0;
        orc_expand_line_roi(roi_h, mc, h, w, ks, exp_h, NULL);
        orc_expand_line_roi(roi_v, mc, h, w, ks, exp_v, NULL);
        int x0 = rect[0], y0 = rect[1], cw = rect[2], ch = rect[3];
        /* numpy slicing clips the crop at the image border */
        if (x0 + cw > w) cw = w - x0;
        if (y0 + ch > h) ch = h - y0;
        crop = (uint8_t *)malloc((size_t)cw * ch);
        lab_h = (int32_t *)malloc((size_t)cw * ch * sizeof(int32_t));
        lab_v = (int32_t *)malloc((size_t)cw * ch * sizeof(int32_t));
        for (int y = 0; y < ch; y++) memcpy(crop + (size_t)y * cw, exp_h + (size_t)(y0 + y) * w + x0, (size_t)cw);
        orc_connected_components(crop, ch, cw, lab_h);
        for (int y = 0; y < ch; y++) memcpy(crop + (size_t)y * cw, exp_v + (size_t)(y0 + y) * w + x0, (size_t)cw);
        orc_connected_components(crop, ch, cw, lab_v);
        orc_group_points(cyl, ncyl, lab_h, ch, cw, x0, y0, rows);
        orc_group_points(cyl, ncyl, lab_v, ch, cw, x0, y0, cols);
        orc_fit_lines(cols, 0);
        orc_fit_lines(rows, 1);
        orc_remove_label(rows, cols);
        if (subpixel) st = orc_subpixel_refine(gray, h, w, rows, cols, sp_window, sp_step);
        if (st == 0) {
            orc_intersections(rows, cols, rect);
            orc_clean_and_relabel(rows, cols);
            if (bgr) {      /* :1433-1435: GaussianBlur((7,7)) of the colour image, channel by channel, then BGR2GRAY */
                uint8_t *pl = (uint8_t *)malloc(N), *bl = (uint8_t *)malloc(3 * N);
                for (int c = 0; c < 3; c++) {
                    for (size_t i = 0; i < N; i++) pl[i] = bgr[3 * i + c];
                    orc_blur7(pl, h, w, g7);
                    for (size_t i = 0; i < N; i++) bl[3 * i + c] = g7[i];
                }
                orc_bgr2gray(bl, N, g7);
                free(pl); free(bl);
            } else orc_blur7(gray, h, w, g7);
            int n = orc_index_points(rows, cols, g7, h, w, r0, center, xy, id, cap);
            if (n < 0) st = -n;
            else if (n > CPE_MAXP) orc_capacity_overflow = 1;   /* more grid points than a table of the boundary holds */
            else *n_out = n;
        }
    }
    if (dbg) {
        if (dbg->binary) memcpy(dbg->binary, binary, N);
        if (dbg->hmask) memcpy(dbg->hmask, hmask, N);
        if (dbg->vmask) memcpy(dbg->vmask, vmask, N);
        if (dbg->mask_contour) memcpy(dbg->mask_contour, mc, N);
        if (dbg->roi_h && st != 1 && st != 2) memcpy(dbg->roi_h, roi_h, N);
        if (dbg->roi_v && st != 1 && st != 2) memcpy(dbg->roi_v, roi_v, N);
        if (dbg->exp_h && st != 1 && st != 2) memcpy(dbg->exp_h, exp_h, N);
        if (dbg->exp_v && st != 1 && st != 2) memcpy(dbg->exp_v, exp_v, N);
        if (dbg->joints) memcpy(dbg->joints, cent, (size_t)(nj < dbg->cap_joints ? nj : dbg->cap_joints) * 2 * sizeof(int));
        dbg->n_joints = nj;
        dbg->n_cyl_joints = ncyl;
        memcpy(dbg->rect, rect, sizeof(rect));
        dbg->r0 = r0;
        memcpy(dbg->spot, spot, sizeof(spot));
        dbg->n_rows = (st == 0 || (st >= 3 && st != 7)) ? rows->nlines : 0;
        dbg->n_cols = (st == 0 || (st >= 3 && st != 7)) ? cols->nlines : 0;
        dbg->n_keypoints = nkp;
        const int have = (st == 0 || (st >= 3 && st != 7));
        if (dbg->rows_out) { if (have) memcpy(dbg->rows_out, rows, sizeof(orc_lineset)); else dbg->rows_out->nlines = 0; }
        if (dbg->cols_out) { if (have) memcpy(dbg->cols_out, cols, sizeof(orc_lineset)); else dbg->cols_out->nlines = 0; }
    }
    free(blurred); free(binary); free(hmask); free(vmask); free(mc); free(roi_h); free(roi_v); free(exp_h); free(exp_v);
    free(g7); free(cent); free(cyl); free(rows); free(cols); free(lab_h); free(lab_v); free(crop);
    if (orc_capacity_overflow) { st = ORC_ST_OVERFLOW; *n_out = 0; }   /* as the library: any exceeded capacity overrides the status */
    return st;
}
