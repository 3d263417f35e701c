/*
 * ORACLE (test infrastructure) -- stage a-3: detect_largest_blob (util_cylinder.py:1830-1899)
 *   BGR2LAB L channel (LUT, grey-replicated input) -> CLAHE(4.5, 4x4) -> SimpleBlobDetector
 *   (minArea 10, circularity/convexity/inertia off) -> filled discs -> largest external contour ->
 *   convex hull -> filled hull mask + bounding rect.
 * [ext] OpenCV 4.5.5 restated (color_lab.cpp RGB2Lab_b, clahe.cpp, blobdetector.cpp) -- parity unpinned.
 */
#include "orc_common.h"

typedef struct orc_contours orc_contours;
orc_contours *orc_find_contours(const uint8_t *src, int h, int w, int mode, int method);
void orc_contours_free(orc_contours *cs);
int orc_contours_count(const orc_contours *cs);
int orc_contour_size(const orc_contours *cs, int i);
int orc_contour_is_hole(const orc_contours *cs, int i);
const orc_pt *orc_contour_points(const orc_contours *cs, int i);
void orc_contour_moments(const orc_pt *p, int n, double *m00, double *m10, double *m01);
double orc_contour_area(const orc_pt *p, int n);
void orc_bounding_rect(const orc_pt *p, int n, int *r);
int orc_convex_hull(const orc_pt *pin, int n, orc_pt *hull);
void orc_circle_fill(uint8_t *img, int h, int w, int cx, int cy, int radius, uint8_t c);
void orc_fill_poly(uint8_t *img, int h, int w, const orc_pt *v, int n, uint8_t c);

/* tools/gen_lab_lut.py */
static const uint8_t LAB_L[256] = { 0, 1, 1, 2, 2, 3, 5, 5, 6, 7, 7, 8, 9, 9, 10, 11, 12, 12, 14, 15, 16, 17, 18, 19, 21, 23, 24, 25, 27, 27, 28, 30, 31, 33, 34, 35, 36, 38, 39, 40, 41, 42, 43, 45, 46, 47, 48, 50, 51, 52, 53, 54, 55, 57, 58, 59, 60, 61, 62, 63, 65, 66, 67, 68, 69, 70, 71, 73, 74, 75, 76, 77, 78, 79, 80, 82, 82, 83, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 97, 98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 119, 119, 121, 122, 123, 124, 125, 126, 127, 128, 129, 130, 131, 132, 133, 134, 135, 136, 137, 138, 139, 140, 141, 142, 143, 144, 145, 146, 147, 148, 149, 150, 151, 152, 153, 154, 155, 156, 156, 157, 158, 159, 160, 161, 162, 163, 164, 165, 166, 167, 168, 169, 170, 171, 172, 173, 174, 175, 176, 177, 178, 179, 180, 180, 181, 182, 183, 184, 185, 186, 187, 188, 189, 190, 191, 192, 193, 194, 195, 196, 196, 197, 198, 199, 200, 201, 202, 203, 204, 205, 206, 207, 208, 208, 209, 210, 211, 212, 213, 214, 215, 216, 217, 218, 219, 219, 220, 221, 222, 223, 224, 225, 226, 227, 228, 228, 229, 230, 231, 232, 233, 234, 235, 236, 237, 237, 238, 239, 240, 241, 242, 243, 244, 245, 245, 246, 247, 248, 249, 250, 251, 252, 253, 253, 254, 255 };

/* true-colour frames (util_cylinder.py:1840 on a 3-channel image), [ext] OpenCV 4.5.5 RGB2Lab_b: every channel through the
 * sRGB gamma table (ORC_GAMMA, 11 bits), Y = the XYZ row {0.212671, 0.715160, 0.072169} in 12-bit fixed point
 * (871 / 2929 / 296 for R / G / B, sum 4096), L through the cube-root table: ORC_LY[Y] = (296 * cbrt_tab[Y] - 1336934 + 2^14) >> 15.
 * Both tables by tools/gen_lab_lut.py colour; R = G = B = v gives LAB_L[v]. */
static const uint16_t ORC_GAMMA[256] = { 0, 1, 1, 2, 2, 3, 4, 4, 5, 6, 6, 7, 8, 8, 9, 10, 11, 11, 12, 13, 14, 15, 16, 17, 19, 20, 21, 22, 24, 25, 26, 28, 29, 31, 33, 34, 36, 38, 40, 41, 43, 45, 47, 49, 51, 54, 56, 58, 60, 63, 65, 68, 70, 73, 75, 78, 81, 83, 86, 89, 92, 95, 98, 101, 105, 108, 111, 115, 118, 121, 125, 129, 132, 136, 140, 144, 147, 151, 155, 160, 164, 168, 172, 176, 181, 185, 190, 194, 199, 204, 209, 213, 218, 223, 228, 233, 239, 244, 249, 255, 260, 265, 271, 277, 282, 288, 294, 300, 306, 312, 318, 324, 331, 337, 343, 350, 356, 363, 370, 376, 383, 390, 397, 404, 411, 418, 426, 433, 440, 448, 455, 463, 471, 478, 486, 494, 502, 510, 518, 527, 535, 543, 552, 560, 569, 578, 586, 595, 604, 613, 622, 631, 641, 650, 659, 669, 678, 688, 698, 707, 717, 727, 737, 747, 757, 768, 778, 788, 799, 809, 820, 831, 842, 852, 863, 875, 886, 897, 908, 920, 931, 943, 954, 966, 978, 990, 1002, 1014, 1026, 1038, 1050, 1063, 1075, 1088, 1101, 1113, 1126, 1139, 1152, 1165, 1178, 1192, 1205, 1218, 1232, 1245, 1259, 1273, 1287, 1301, 1315, 1329, 1343, 1357, 1372, 1386, 1401, 1415, 1430, 1445, 1460, 1475, 1490, 1505, 1521, 1536, 1551, 1567, 1583, 1598, 1614, 1630, 1646, 1662, 1678, 1695, 1711, 1728, 1744, 1761, 1778, 1794, 1811, 1828, 1846, 1863, 1880, 1897, 1915, 1933, 1950, 1968, 1986, 2004, 2022, 2040 };
static const uint8_t ORC_LY[2041] = { 0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 27, 28, 29, 30, 31, 32, 33, 33, 34, 35, 36, 36, 37, 38, 38, 39, 40, 40, 41, 42, 42, 43, 43, 44, 45, 45, 46, 46, 47, 47, 48, 48, 49, 50, 50, 51, 51, 52, 52, 53, 53, 54, 54, 54, 55, 55, 56, 56, 57, 57, 58, 58, 58, 59, 59, 60, 60, 61, 61, 61, 62, 62, 63, 63, 63, 64, 64, 65, 65, 65, 66, 66, 66, 67, 67, 68, 68, 68, 69, 69, 69, 70, 70, 70, 71, 71, 71, 72, 72, 72, 73, 73, 73, 74, 74, 74, 75, 75, 75, 76, 76, 76, 77, 77, 77, 77, 78, 78, 78, 79, 79, 79, 80, 80, 80, 80, 81, 81, 81, 82, 82, 82, 82, 83, 83, 83, 83, 84, 84, 84, 85, 85, 85, 85, 86, 86, 86, 86, 87, 87, 87, 87, 88, 88, 88, 88, 89, 89, 89, 89, 90, 90, 90, 90, 91, 91, 91, 91, 92, 92, 92, 92, 93, 93, 93, 93, 94, 94, 94, 94, 95, 95, 95, 95, 95, 96, 96, 96, 96, 97, 97, 97, 97, 97, 98, 98, 98, 98, 99, 99, 99, 99, 99, 100, 100, 100, 100, 101, 101, 101, 101, 101, 102, 102, 102, 102, 102, 103, 103, 103, 103, 103, 104, 104, 104, 104, 104, 105, 105, 105, 105, 105, 106, 106, 106, 106, 106, 107, 107, 107, 107, 107, 108, 108, 108, 108, 108, 109, 109, 109, 109, 109, 109, 110, 110, 110, 110, 110, 111, 111, 111, 111, 111, 112, 112, 112, 112, 112, 112, 113, 113, 113, 113, 113, 114, 114, 114, 114, 114, 114, 115, 115, 115, 115, 115, 115, 116, 116, 116, 116, 116, 116, 117, 117, 117, 117, 117, 117, 118, 118, 118, 118, 118, 119, 119, 119, 119, 119, 119, 119, 120, 120, 120, 120, 120, 120, 121, 121, 121, 121, 121, 121, 122, 122, 122, 122, 122, 122, 123, 123, 123, 123, 123, 123, 124, 124, 124, 124, 124, 124, 124, 125, 125, 125, 125, 125, 125, 126, 126, 126, 126, 126, 126, 126, 127, 127, 127, 127, 127, 127, 128, 128, 128, 128, 128, 128, 128, 129, 129, 129, 129, 129, 129, 129, 130, 130, 130, 130, 130, 130, 130, 131, 131, 131, 131, 131, 131, 131, 132, 132, 132, 132, 132, 132, 132, 133, 133, 133, 133, 133, 133, 133, 134, 134, 134, 134, 134, 134, 134, 135, 135, 135, 135, 135, 135, 135, 135, 136, 136, 136, 136, 136, 136, 136, 137, 137, 137, 137, 137, 137, 137, 138, 138, 138, 138, 138, 138, 138, 138, 139, 139, 139, 139, 139, 139, 139, 139, 140, 140, 140, 140, 140, 140, 140, 141, 141, 141, 141, 141, 141, 141, 141, 142, 142, 142, 142, 142, 142, 142, 142, 143, 143, 143, 143, 143, 143, 143, 143, 144, 144, 144, 144, 144, 144, 144, 144, 145, 145, 145, 145, 145, 145, 145, 145, 146, 146, 146, 146, 146, 146, 146, 146, 147, 147, 147, 147, 147, 147, 147, 147, 147, 148, 148, 148, 148, 148, 148, 148, 148, 149, 149, 149, 149, 149, 149, 149, 149, 149, 150, 150, 150, 150, 150, 150, 150, 150, 151, 151, 151, 151, 151, 151, 151, 151, 151, 152, 152, 152, 152, 152, 152, 152, 152, 152, 153, 153, 153, 153, 153, 153, 153, 153, 154, 154, 154, 154, 154, 154, 154, 154, 154, 155, 155, 155, 155, 155, 155, 155, 155, 155, 156, 156, 156, 156, 156, 156, 156, 156, 156, 156, 157, 157, 157, 157, 157, 157, 157, 157, 157, 158, 158, 158, 158, 158, 158, 158, 158, 158, 159, 159, 159, 159, 159, 159, 159, 159, 159, 159, 160, 160, 160, 160, 160, 160, 160, 160, 160, 161, 161, 161, 161, 161, 161, 161, 161, 161, 161, 162, 162, 162, 162, 162, 162, 162, 162, 162, 163, 163, 163, 163, 163, 163, 163, 163, 163, 163, 164, 164, 164, 164, 164, 164, 164, 164, 164, 164, 165, 165, 165, 165, 165, 165, 165, 165, 165, 165, 166, 166, 166, 166, 166, 166, 166, 166, 166, 166, 167, 167, 167, 167, 167, 167, 167, 167, 167, 167, 168, 168, 168, 168, 168, 168, 168, 168, 168, 168, 168, 169, 169, 169, 169, 169, 169, 169, 169, 169, 169, 170, 170, 170, 170, 170, 170, 170, 170, 170, 170, 170, 171, 171, 171, 171, 171, 171, 171, 171, 171, 171, 172, 172, 172, 172, 172, 172, 172, 172, 172, 172, 172, 173, 173, 173, 173, 173, 173, 173, 173, 173, 173, 173, 174, 174, 174, 174, 174, 174, 174, 174, 174, 174, 174, 175, 175, 175, 175, 175, 175, 175, 175, 175, 175, 176, 176, 176, 176, 176, 176, 176, 176, 176, 176, 176, 176, 177, 177, 177, 177, 177, 177, 177, 177, 177, 177, 177, 178, 178, 178, 178, 178, 178, 178, 178, 178, 178, 178, 179, 179, 179, 179, 179, 179, 179, 179, 179, 179, 179, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 181, 181, 181, 181, 181, 181, 181, 181, 181, 181, 181, 181, 182, 182, 182, 182, 182, 182, 182, 182, 182, 182, 182, 183, 183, 183, 183, 183, 183, 183, 183, 183, 183, 183, 183, 184, 184, 184, 184, 184, 184, 184, 184, 184, 184, 184, 184, 185, 185, 185, 185, 185, 185, 185, 185, 185, 185, 185, 185, 186, 186, 186, 186, 186, 186, 186, 186, 186, 186, 186, 186, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 187, 188, 188, 188, 188, 188, 188, 188, 188, 188, 188, 188, 188, 189, 189, 189, 189, 189, 189, 189, 189, 189, 189, 189, 189, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 190, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 191, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 194, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 195, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 196, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 197, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 198, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 199, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 201, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 202, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 203, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 204, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 205, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 207, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 208, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 209, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 210, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 211, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 212, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 213, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 214, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 215, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 216, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 217, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 218, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 219, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 220, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 221, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 222, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 223, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 224, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 225, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 226, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 227, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 228, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 229, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 230, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 231, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 232, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 233, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 234, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 235, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 236, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 237, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 238, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 239, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 240, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 241, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 242, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 243, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 244, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 245, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 246, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 247, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 248, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 249, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 251, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 252, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 253, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 254, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255 };
ORC_API void orc_lab_l_bgr(const uint8_t *bgr, int h, int w, uint8_t *L)
{
    for (size_t i = 0; i < (size_t)h * w; i++) {
        const int B = ORC_GAMMA[bgr[3 * i]], G = ORC_GAMMA[bgr[3 * i + 1]], R = ORC_GAMMA[bgr[3 * i + 2]];
        L[i] = ORC_LY[(R * 871 + G * 2929 + B * 296 + (1 << 11)) >> 12];
    }
}

ORC_API void orc_lab_l(const uint8_t *gray, int h, int w, uint8_t *L)
{
    for (size_t i = 0; i < (size_t)h * w; i++) L[i] = LAB_L[gray[i]];
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* cv2.createCLAHE(clipLimit, (tx,ty)).apply(src) */
ORC_API void orc_clahe(const uint8_t *src, int h, int w, double clip, int tilesX, int tilesY, uint8_t *dst)
{
    int eh = h, ew = w;
    const uint8_t *lsrc = src;
    uint8_t *ext = NULL;
    if (w % tilesX != 0 || h % tilesY != 0) {
        eh = h + (tilesY - (h % tilesY));
        ew = w + (tilesX - (w % tilesX));
        ext = (uint8_t *)malloc((size_t)eh * ew);
        for (int y = 0; y < eh; y++)
            for (int x = 0; x < ew; x++)
                ext[(size_t)y * ew + x] = src[(size_t)orc_reflect101(y, h) * w + orc_reflect101(x, w)];
        lsrc = ext;
    }
    int tw = ew / tilesX, th = eh / tilesY;
    int total = tw * th;
    int clipLimit = 0;
    if (clip > 0.0) {
        clipLimit = (int)(clip * total / 256);
        if (clipLimit < 1) clipLimit = 1;
    }
    float lutScale = (float)(255) / total;
    uint8_t *lut = (uint8_t *)malloc((size_t)tilesX * tilesY * 256);
    for (int ty = 0; ty < tilesY; ty++)
        for (int tx = 0; tx < tilesX; tx++) {
            int hist[256] = {0};
            for (int y = 0; y < th; y++)
                for (int x = 0; x < tw; x++) hist[lsrc[(size_t)(ty * th + y) * ew + tx * tw + x]]++;
            if (clipLimit > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; i++)
                    if (hist[i] > clipLimit) { clipped += hist[i] - clipLimit; hist[i] = clipLimit; }
                int batch = clipped / 256, residual = clipped - batch * 256;
                for (int i = 0; i < 256; i++) hist[i] += batch;
                if (residual != 0) {
                    int stepr = 256 / residual;
                    if (stepr < 1) stepr = 1;
                    for (int i = 0; i < 256 && residual > 0; i += stepr, residual--) hist[i]++;
                }
            }
            int sum = 0;
            uint8_t *tl = lut + (size_t)(ty * tilesX + tx) * 256;
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                tl[i] = sat_u8(cv_round_f((float)sum * lutScale));
            }
        }
    float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; y++) {
        float tyf = y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tilesY - 1) ty2 = tilesY - 1;
        for (int x = 0; x < w; x++) {
            float txf = x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tilesX - 1) tx2 = tilesX - 1;
            int v = src[(size_t)y * w + x];
            const uint8_t *p1 = lut + (size_t)(ty1 * tilesX) * 256, *p2 = lut + (size_t)(ty2 * tilesX) * 256;
            float a = (float)p1[tx1 * 256 + v] * xa1, b = (float)p1[tx2 * 256 + v] * xa;
            float c = (float)p2[tx1 * 256 + v] * xa1, d = (float)p2[tx2 * 256 + v] * xa;
            float res = (a + b) * ya1 + (c + d) * ya;
            dst[(size_t)y * w + x] = sat_u8(cv_round_f(res));
        }
    }
    free(lut);
    free(ext);
}

typedef struct { double x, y, radius; } blob_center;
typedef struct { blob_center *c; int n, cap; } center_group;

static int dbl_cmp(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* SimpleBlobDetector::findBlobs on one binarised image */
static int find_blobs(const uint8_t *bin, int h, int w, double minArea, double maxArea, blob_center **out)
{
    orc_contours *cs = orc_find_contours(bin, h, w, 1 /* RETR_LIST */, 1 /* CHAIN_APPROX_NONE */);
    int nc = orc_contours_count(cs), n = 0;
    blob_center *res = (blob_center *)malloc((size_t)(nc + 1) * sizeof(blob_center));
    for (int ci = 0; ci < nc; ci++) {
        const orc_pt *p = orc_contour_points(cs, ci);
        int np = orc_contour_size(cs, ci);
        double m00, m10, m01;
        orc_contour_moments(p, np, &m00, &m10, &m01);
        if (m00 < minArea || m00 >= maxArea) continue;
        if (m00 == 0.0) continue;
        double cx = m10 / m00, cy = m01 / m00;
        int ix = (int)lrint(cx), iy = (int)lrint(cy);
        if (bin[(size_t)iy * w + ix] != 0) continue; /* blobColor = 0 */
        double *d = (double *)malloc((size_t)np * sizeof(double));
        for (int k = 0; k < np; k++) {
            double dx = cx - p[k].x, dy = cy - p[k].y;
            d[k] = sqrt(dx * dx + dy * dy);
        }
        qsort(d, np, sizeof(double), dbl_cmp);
        res[n].x = cx; res[n].y = cy;
        res[n].radius = (d[(np - 1) / 2] + d[np / 2]) / 2.;
        n++;
        free(d);
    }
    orc_contours_free(cs);
    *out = res;
    return n;
}

/* SimpleBlobDetector(minArea=10, defaults otherwise).detect(gray) -> key points (x, y, size) as f32 */
ORC_API int orc_simple_blob_detector(const uint8_t *gray, int h, int w, float *kp /* cap x 3 */, int cap,
                                     int *stats /* optional 17: blobs per threshold */)
{
    const double minDist = 10.0;
    center_group *groups = NULL;
    int ng = 0, gcap = 0;
    uint8_t *bin = (uint8_t *)malloc((size_t)h * w);
    int ti = 0;
    for (double thresh = 50; thresh < 220; thresh += 10, ti++) {
        int it = (int)floor(thresh);
        for (size_t i = 0; i < (size_t)h * w; i++) bin[i] = gray[i] > it ? 255 : 0;
        blob_center *cur;
        int ncur = find_blobs(bin, h, w, 10.0f, 5000.0f, &cur);
        if (stats) stats[ti] = ncur;
        int ng0 = ng; /* groups created at this threshold are appended afterwards */
        center_group *newg = NULL;
        int nnew = 0, newcap = 0;
        for (int i = 0; i < ncur; i++) {
            int isNew = 1;
            for (int j = 0; j < ng0; j++) {
                center_group *g = &groups[j];
                blob_center *mid = &g->c[g->n / 2];
                double dx = mid->x - cur[i].x, dy = mid->y - cur[i].y;
                double dist = sqrt(dx * dx + dy * dy);
                isNew = dist >= minDist && dist >= mid->radius && dist >= cur[i].radius;
                if (!isNew) {
                    if (g->n == g->cap) { g->cap *= 2; g->c = (blob_center *)realloc(g->c, (size_t)g->cap * sizeof(blob_center)); }
                    g->c[g->n++] = cur[i];
                    int k = g->n - 1;
                    while (k > 0 && cur[i].radius < g->c[k - 1].radius) { g->c[k] = g->c[k - 1]; k--; }
                    g->c[k] = cur[i];
                    break;
                }
            }
            if (isNew) {
                if (nnew == newcap) { newcap = newcap ? newcap * 2 : 64; newg = (center_group *)realloc(newg, (size_t)newcap * sizeof(center_group)); }
                newg[nnew].cap = 4; newg[nnew].n = 1;
                newg[nnew].c = (blob_center *)malloc(4 * sizeof(blob_center));
                newg[nnew].c[0] = cur[i];
                nnew++;
            }
        }
        if (ng + nnew > gcap) { gcap = (ng + nnew) * 2 + 16; groups = (center_group *)realloc(groups, (size_t)gcap * sizeof(center_group)); }
        for (int i = 0; i < nnew; i++) groups[ng++] = newg[i];
        free(newg);
        free(cur);
    }
    free(bin);
    int nk = 0;
    for (int i = 0; i < ng; i++) {
        center_group *g = &groups[i];
        if (g->n >= 2) {
            double sx = 0, sy = 0, nrm = 0;
            for (int j = 0; j < g->n; j++) { sx += 1.0 * g->c[j].x; sy += 1.0 * g->c[j].y; nrm += 1.0; }
            sx *= (1. / nrm); sy *= (1. / nrm);
            if (nk < cap) {
                kp[3 * nk] = (float)sx; kp[3 * nk + 1] = (float)sy;
                kp[3 * nk + 2] = (float)(g->c[g->n / 2].radius) * 2.0f;
            }
            nk++;
        }
        free(g->c);
    }
    free(groups);
    return nk;
}

/* detect_largest_blob: gray -> mask_contour (filled hull, 0/255), rect[4] = boundingRect(max_contour).
 * returns 0 ok, 1 = no contour (cv2.convexHull(None) raises in the reference). */
int orc_detect_largest_blob_l(const uint8_t *gray, const uint8_t *l_in, int h, int w, double clip, uint8_t *mask, int *rect,
                              uint8_t *cl_out, int *nkp_out);
ORC_API int orc_detect_largest_blob(const uint8_t *gray, int h, int w, double clip, uint8_t *mask, int *rect,
                                    uint8_t *cl_out /* optional: CLAHE'd L */, int *nkp_out)
{
    return orc_detect_largest_blob_l(gray, NULL, h, w, clip, mask, rect, cl_out, nkp_out);
}

/* l_in: the L channel of a true-colour frame (orc_lab_l_bgr), or NULL: L of the grey-replicated frame */
int orc_detect_largest_blob_l(const uint8_t *gray, const uint8_t *l_in, int h, int w, double clip, uint8_t *mask, int *rect,
                              uint8_t *cl_out, int *nkp_out)
{
    uint8_t *L = (uint8_t *)malloc((size_t)h * w), *cl = (uint8_t *)malloc((size_t)h * w);
    if (l_in) memcpy(L, l_in, (size_t)h * w);
    else orc_lab_l(gray, h, w, L);
    orc_clahe(L, h, w, clip, 4, 4, cl);
    if (cl_out) memcpy(cl_out, cl, (size_t)h * w);
    int cap = 65536;
    float *kp = (float *)malloc((size_t)cap * 3 * sizeof(float));
    int nk = orc_simple_blob_detector(cl, h, w, kp, cap, NULL);
    if (nk > cap) nk = cap;
    if (nkp_out) *nkp_out = nk;
    uint8_t *ext = (uint8_t *)calloc((size_t)h * w, 1);
    for (int i = 0; i < nk; i++) {
        float radius = kp[3 * i + 2] / 2;              /* python float division of an f32-valued float */
        int er = (int)((double)radius + 4);
        orc_circle_fill(ext, h, w, (int)kp[3 * i], (int)kp[3 * i + 1], er, 255);
    }
    orc_contours *cs = orc_find_contours(ext, h, w, 0, 2);
    int nc = orc_contours_count(cs), best = -1;
    double max_area = 0;
    for (int i = 0; i < nc; i++) {
        double a = orc_contour_area(orc_contour_points(cs, i), orc_contour_size(cs, i));
        if (a > max_area) { max_area = a; best = i; }
    }
    memset(mask, 0, (size_t)h * w);
    int st = 1;
    if (best >= 0) {
        const orc_pt *p = orc_contour_points(cs, best);
        int np = orc_contour_size(cs, best);
        orc_pt *hull = (orc_pt *)malloc((size_t)(np + 1) * sizeof(orc_pt));
        int nh = orc_convex_hull(p, np, hull);
        orc_fill_poly(mask, h, w, hull, nh, 255);
        orc_bounding_rect(p, np, rect);
        free(hull);
        st = 0;
    }
    orc_contours_free(cs);
    free(ext); free(kp); free(L); free(cl);
    return st;
}
