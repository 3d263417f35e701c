/*
 * ORACLE (test infrastructure) -- the line set shared by orc_lines.c, orc_plane.c and orc_detect.c, and the capacities of
 * the line stage.  The reference's lists are unbounded (util_cylinder.py:376-389, 1106-1151); the C-ABI library bounds them
 * (include/cpe.h: CPE_MAXJ joints inside the region rectangle, CPE_MAXL label groups per direction, CPE_MAXLP joints per
 * group, CPE_MAXP grid points) and reports CPE_ST_OVERFLOW beyond.  The oracle reads the same constants and reports the
 * same status, so that the two sides agree on every frame: a frame beyond the capacities is a documented deviation from
 * the reference on both, not a silent truncation on one.
 */
#ifndef ORC_LINES_H
#define ORC_LINES_H
#include "../../include/cpe.h"

#define ORC_MAXL CPE_MAXL   /* lines per direction */
#define ORC_MAXLP CPE_MAXLP /* points per line */
#define ORC_ST_OVERFLOW CPE_ST_OVERFLOW

typedef struct {
    int nlines;
    int npts[ORC_MAXL];
    double pts[ORC_MAXL][ORC_MAXLP][2];
    double eq[ORC_MAXL][6];
    int has_eq[ORC_MAXL]; /* clean_and_relabel keeps an equation only if it is not [0,0,0,0] (always kept: 6 long) */
    int label[ORC_MAXL];  /* bookkeeping: original component label / running number */
} orc_lineset;

/* set where a list would have to drop an entry; orc_detect_grid_ex / orc_detect_grid_plane clear it on entry and turn it
 * into ORC_ST_OVERFLOW on exit (single-threaded test infrastructure: one flag per process) */
extern int orc_capacity_overflow;
#endif
