/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Coarse-to-fine tile recursion of image3.cpp:676-865 (calcDisparityMap), depth-first and
 * in the reference's visiting order, with the preview (image3.cpp:747-748) and the file
 * outputs (image3.cpp:822-861) left to the caller: each leaf is returned as a record. */
#include <stdlib.h>
#include "oracle.h"

typedef struct {
    const void* img1; const void* img2;
    int dtype, width, height; size_t stride;      /* the second raster (image3.cpp:684-685 takes its size) */
    int width1, height1; size_t stride1;          /* the first raster */
    int min_size, precision;
    orc_leaf* leaves; int max_leaves, n_leaves, n_calls, failed;
} walk_t;

static void walk(walk_t* w, orc_rect bbox1, orc_rect bbox2)
{
    if (w->failed) return;
    int src_width = w->width, src_height = w->height;          /* image3.cpp:684-685 */
    int width = bbox1.width > bbox2.width ? bbox1.width : bbox2.width;
    int height = bbox1.height > bbox2.height ? bbox1.height : bbox2.height;

    orc_offset off;
    orc_find_offset(w->img1, w->width1, w->height1, w->stride1, w->img2, w->width, w->height, w->stride,
                    w->dtype, bbox1, bbox2, width, height, w->precision, &off, NULL, NULL);
    w->n_calls++;
    float max = off.peak;

    int total_offset_x = (bbox1.minx - bbox2.minx) + off.dx;    /* image3.cpp:696-699 */
    int total_offset_y = (bbox1.miny - bbox2.miny) + off.dy;

    orc_rect overlap, overlap2;                                 /* image3.cpp:701-708 */
    overlap.minx = bbox1.minx > total_offset_x ? bbox1.minx : total_offset_x;
    overlap.miny = bbox1.miny > total_offset_y ? bbox1.miny : total_offset_y;
    int maxx = (bbox1.minx + bbox1.width < total_offset_x + src_width) ? bbox1.minx + bbox1.width : total_offset_x + src_width;
    int maxy = (bbox1.miny + bbox1.height < total_offset_y + src_height) ? bbox1.miny + bbox1.height : total_offset_y + src_height;
    overlap.width = maxx - overlap.minx;
    overlap.height = maxy - overlap.miny;
    if (overlap.width <= 0 || overlap.height <= 0) { w->failed = 1; return; }  /* error("no overlap") */

    overlap2.minx = overlap.minx - total_offset_x;              /* image3.cpp:714-740 */
    if (overlap2.minx < 0) { overlap.minx -= overlap2.minx; overlap.width -= overlap2.minx; overlap2.minx = 0; }
    overlap2.miny = overlap.miny - total_offset_y;
    if (overlap2.miny < 0) { overlap.miny -= overlap2.miny; overlap.height -= overlap2.miny; overlap2.miny = 0; }
    maxx = src_width - (overlap2.minx + overlap.width);
    if (maxx < 0) overlap.width -= maxx;
    maxy = src_height - (overlap2.miny + overlap.height);
    if (maxy < 0) overlap.height -= maxy;
    overlap2.width = overlap.width;
    overlap2.height = overlap.height;

    if (overlap.width > 2 * w->min_size && overlap.height > 2 * w->min_size && max > 0.75) {  /* :751 */
        int dim = overlap.width < overlap.height ? overlap.width : overlap.height;
        { unsigned char r = 0; while (dim >>= 1) r++; dim = 1 << r; }
        dim = (dim / 2) + (dim % 1);
        orc_rect r1, r2;
        r1.width = r1.height = r2.width = r2.height = dim;
        for (int j = 0; j < overlap.height; j += dim) {          /* image3.cpp:764-774 */
            int jj = j + dim > overlap.height ? overlap.height - dim : j;
            for (int i = 0; i < overlap.width; i += dim) {
                int ii = i + dim > overlap.width ? overlap.width - dim : i;
                r1.minx = overlap.minx + ii; r1.miny = overlap.miny + jj;
                r2.minx = overlap2.minx + ii; r2.miny = overlap2.miny + jj;
                walk(w, r1, r2);
            }
        }
    } else {                                                     /* image3.cpp:812-861 */
        if (w->n_leaves < w->max_leaves) {
            orc_leaf* l = &w->leaves[w->n_leaves];
            l->bbox1 = bbox1;
            l->total_dx = total_offset_x; l->total_dy = total_offset_y;
            l->u1 = overlap.minx + overlap.width / 2;
            l->v1 = overlap.miny + overlap.height / 2;
            l->u2 = overlap2.minx + overlap2.width / 2;
            l->v2 = overlap2.miny + overlap2.height / 2;
            l->peak = max;
        }
        w->n_leaves++;
    }
}

int orc_disparity_map(const void* img1, const void* img2, int dtype, int width, int height,
                      size_t stride_bytes, int min_size, int precision,
                      orc_leaf* leaves, int max_leaves, int* n_find_offset_calls)
{
    return orc_disparity_map2(img1, width, height, stride_bytes, img2, width, height, stride_bytes, dtype, min_size, precision,
                              leaves, max_leaves, n_find_offset_calls);
}

/* Rasters of different size: the reference only warns (image3.cpp:995-998), starts from the whole of each raster
 * (image3.cpp:1001-1006) and clips overlaps against the second raster's extent (image3.cpp:684-685). */
int orc_disparity_map2(const void* img1, int width1, int height1, size_t stride1_bytes,
                       const void* img2, int width2, int height2, size_t stride2_bytes, int dtype,
                       int min_size, int precision, orc_leaf* leaves, int max_leaves, int* n_find_offset_calls)
{
    walk_t w = { img1, img2, dtype, width2, height2, stride2_bytes, width1, height1, stride1_bytes, min_size, precision,
                 leaves, max_leaves, 0, 0, 0 };
    orc_rect full1 = { 0, 0, width1, height1 }, full2 = { 0, 0, width2, height2 };
    walk(&w, full1, full2);
    if (n_find_offset_calls) *n_find_offset_calls = w.n_calls;
    return w.failed ? -1 : w.n_leaves;
}
