// Host-side coarse-to-fine driver: calcDisparityMap of image3.cpp:676-865, restructured from the
// reference's depth-first recursion into level-synchronous batches (what its own TODO at
// image3.cpp:866-870 asks for): all tiles of one recursion level are independent, so each level
// becomes one mis_xcorr_tiles call per tile size, and all leaves are triangulated in one
// mis_triangulate call.  Leaves are returned in the reference's depth-first visiting order, so
// an XYZ file written from them has the reference's line order.
//
// Plain host C++ above the C ABI: no HIP here.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mi_stereo.h"
#include "mis_host.h"

namespace {

// A tile of the current level and its place in the tree of all tiles evaluated so far.  The children of a tile are
// appended to the tree together, in the reference's loop order, so the depth-first visiting order of the reference's
// recursion is a walk over (first_child, n_children) at the end: no per-tile path, no sort.
struct Node {
    mis_rect b1, b2;
    int tree;                     // index into the tree
};
struct TreeNode {
    int first_child = -1, n_children = 0;
    int leaf = -1;                // index into the leaves (a tile is a leaf or has children)
};

int pow2_floor(int v) { int r = 0; while (v >>= 1) r++; return 1 << r; }

}  // namespace

int mis_disparity_map_with(const MisTileEval& eval, int image_width, int image_height, int min_size,
                           mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles)
{
    return mis_disparity_map_with2(eval, image_width, image_height, image_width, image_height, min_size, leaves, max_leaves, n_leaves, n_tiles);
}

int mis_disparity_map_with2(const MisTileEval& eval, int width1, int height1, int width2, int height2, int min_size,
                            mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles)
{
    if (!n_leaves || (max_leaves > 0 && !leaves) || min_size < 1) return MIS_ERR_ARG;
    if (width1 < 1 || height1 < 1 || width2 < 1 || height2 < 1) return MIS_ERR_ARG;
    *n_leaves = 0;
    if (n_tiles) *n_tiles = 0;
    try {      // the containers below may throw (std::bad_alloc); nothing may leave the C ABI
    const int src_width = width2, src_height = height2;                 // the SECOND raster's size, image3.cpp:684-685
    std::vector<Node> level(1);
    level[0].b1 = mis_rect{ 0, 0, width1, height1 };                     // image3.cpp:1001-1006
    level[0].b2 = mis_rect{ 0, 0, width2, height2 };
    level[0].tree = 0;
    std::vector<TreeNode> tree(1);
    std::vector<mis_leaf> out;
    int tiles = 0;
    std::vector<mis_rect> r1, r2;
    std::vector<mis_offset> o;

    while (!level.empty()) {
        // group the level's tiles by transform size (max of the two boxes, image3.cpp:687-688)
        std::map<std::pair<int, int>, std::vector<int>> by_size;
        for (int i = 0; i < (int)level.size(); i++) {
            const Node& n = level[i];
            by_size[{ std::max(n.b1.width, n.b2.width), std::max(n.b1.height, n.b2.height) }].push_back(i);
        }
        std::vector<mis_offset> off(level.size());
        for (auto& kv : by_size) {
            const int w = kv.first.first, h = kv.first.second;
            r1.clear(); r2.clear();
            for (int i : kv.second) { r1.push_back(level[i].b1); r2.push_back(level[i].b2); }
            o.resize(r1.size());
            int rc = eval(r1.data(), r2.data(), (int)r1.size(), w, h, o.data());
            if (rc) return rc;
            for (size_t k = 0; k < kv.second.size(); k++) off[kv.second[k]] = o[k];
            tiles += (int)r1.size();
        }

        std::vector<Node> next;
        next.reserve(level.size() * 4);
        for (int i = 0; i < (int)level.size(); i++) {
            const Node& n = level[i];
            const mis_rect bbox1 = n.b1, bbox2 = n.b2;
            const float max = off[i].peak;
            const int total_offset_x = (bbox1.minx - bbox2.minx) + off[i].dx;   // image3.cpp:696-699
            const int total_offset_y = (bbox1.miny - bbox2.miny) + off[i].dy;
            mis_rect overlap, overlap2;                                          // image3.cpp:701-708
            overlap.minx = std::max(bbox1.minx, total_offset_x);
            overlap.miny = std::max(bbox1.miny, total_offset_y);
            int maxx = std::min(bbox1.minx + bbox1.width, total_offset_x + src_width);
            int maxy = std::min(bbox1.miny + bbox1.height, total_offset_y + src_height);
            overlap.width = maxx - overlap.minx;
            overlap.height = maxy - overlap.miny;
            if (overlap.width <= 0 || overlap.height <= 0) return MIS_ERR_STATE;   // error("no overlap"), :710-712
            overlap2.minx = overlap.minx - total_offset_x;                       // image3.cpp:714-740
            if (overlap2.minx < 0) { overlap.minx -= overlap2.minx; overlap.width -= overlap2.minx; overlap2.minx = 0; }
            overlap2.miny = overlap.miny - total_offset_y;
            if (overlap2.miny < 0) { overlap.miny -= overlap2.miny; overlap.height -= overlap2.miny; overlap2.miny = 0; }
            maxx = src_width - (overlap2.minx + overlap.width);
            if (maxx < 0) overlap.width -= maxx;
            maxy = src_height - (overlap2.miny + overlap.height);
            if (maxy < 0) overlap.height -= maxy;
            overlap2.width = overlap.width;
            overlap2.height = overlap.height;

            if (overlap.width > 2 * min_size && overlap.height > 2 * min_size && max > 0.75) {   // :751
                const int dim = pow2_floor(std::min(overlap.width, overlap.height)) / 2;          // :754-761
                tree[n.tree].first_child = (int)tree.size();
                for (int j = 0; j < overlap.height; j += dim) {                                  // :764-774
                    const int jj = j + dim > overlap.height ? overlap.height - dim : j;
                    for (int ii0 = 0; ii0 < overlap.width; ii0 += dim) {
                        const int ii = ii0 + dim > overlap.width ? overlap.width - dim : ii0;
                        Node c;
                        c.b1 = mis_rect{ overlap.minx + ii, overlap.miny + jj, dim, dim };
                        c.b2 = mis_rect{ overlap2.minx + ii, overlap2.miny + jj, dim, dim };
                        c.tree = (int)tree.size();
                        tree.emplace_back();
                        next.push_back(c);
                    }
                }
                tree[n.tree].n_children = (int)tree.size() - tree[n.tree].first_child;
            } else {                                                                              // :812-861
                mis_leaf lf;
                lf.bbox1 = bbox1;
                lf.total_dx = total_offset_x; lf.total_dy = total_offset_y;
                lf.u1 = overlap.minx + overlap.width / 2;
                lf.v1 = overlap.miny + overlap.height / 2;
                lf.u2 = overlap2.minx + overlap2.width / 2;
                lf.v2 = overlap2.miny + overlap2.height / 2;
                lf.peak = max;
                tree[n.tree].leaf = (int)out.size();
                out.push_back(lf);
            }
        }
        level.swap(next);
    }
    *n_leaves = (int)out.size();
    if (n_tiles) *n_tiles = tiles;
    // the leaves in the reference's depth-first visiting order: children pushed in reverse, so the first one is visited first
    std::vector<int> stack(1, 0);
    int emitted = 0;
    while (!stack.empty() && emitted < max_leaves) {
        const TreeNode t = tree[stack.back()];
        stack.pop_back();
        if (t.leaf >= 0) { leaves[emitted++] = out[t.leaf]; continue; }
        for (int c = t.n_children - 1; c >= 0; c--) stack.push_back(t.first_child + c);
    }
    return MIS_OK;
    } catch (...) {
        return MIS_ERR_STATE;
    }
}

extern "C" {

int mis_disparity_map(mis_ctx* ctx, int image_width, int image_height, int min_size,
                      mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles)
{
    if (!ctx) return MIS_ERR_ARG;
    try {
        const MisTileEval eval = [ctx](const mis_rect* r1, const mis_rect* r2, int n, int w, int h, mis_offset* out) {
            return mis_xcorr_tiles(ctx, r1, r2, n, w, h, out);
        };
        return mis_disparity_map_with(eval, image_width, image_height, min_size, leaves, max_leaves, n_leaves, n_tiles);
    } catch (...) {
        return MIS_ERR_STATE;
    }
}

int mis_disparity_map2(mis_ctx* ctx, int width1, int height1, int width2, int height2, int min_size,
                       mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles)
{
    if (!ctx) return MIS_ERR_ARG;
    try {
        const MisTileEval eval = [ctx](const mis_rect* r1, const mis_rect* r2, int n, int w, int h, mis_offset* out) {
            return mis_xcorr_tiles(ctx, r1, r2, n, w, h, out);
        };
        return mis_disparity_map_with2(eval, width1, height1, width2, height2, min_size, leaves, max_leaves, n_leaves, n_tiles);
    } catch (...) {
        return MIS_ERR_STATE;
    }
}

// XYZ text output of image3.cpp:842-849: "%f %f %f " + one "%f " per band + newline.
// bands: n x n_bands values sampled at (u1,v1) of each leaf (image3.cpp:838-841), may be NULL
// when n_bands == 0.  append != 0 keeps an existing file (the reference writes as it goes).
int mis_write_xyz_ex(const char* path, const float* xyz, int n, const float* bands, int n_bands, int flags)
{
    if (!path || n < 0 || (n > 0 && !xyz) || n_bands < 0 || (n_bands > 0 && !bands)) return MIS_ERR_ARG;
    FILE* f = fopen(path, (flags & MIS_XYZ_APPEND) ? "a" : "w");
    if (!f) return MIS_ERR_STATE;
    try {
        for (int i = 0; i < n; i++) {
            std::string b;
            for (int k = 0; k < n_bands; k++) {
                char tmp[64];
                snprintf(tmp, sizeof tmp, "%f ", bands[(size_t)i * n_bands + k]);
                b += tmp;
            }
            // image3.cpp:830-846: the reference's buffer holds 10*n_bands bytes including the terminator
            if ((flags & MIS_XYZ_REF_TRUNCATE) && n_bands > 0 && b.size() > (size_t)(10 * n_bands - 1)) b.resize(10 * n_bands - 1);
            fprintf(f, "%f %f %f %s\n", xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], b.c_str());
        }
    } catch (...) {
        fclose(f);
        return MIS_ERR_STATE;
    }
    return fclose(f) == 0 ? MIS_OK : MIS_ERR_STATE;
}

int mis_write_xyz(const char* path, const float* xyz, int n, const float* bands, int n_bands, int append)
{
    return mis_write_xyz_ex(path, xyz, n, bands, n_bands, append ? MIS_XYZ_APPEND : 0);
}

// The three bands the reference writes into its "dispersion" GeoTIFF for every leaf
// (image3.cpp:853-860): sqrtf(tx*tx + ty*ty) (integer products, then sqrtf), tx, ty, each
// stretched over the leaf's bbox1.  out: 3 planes of width*height floats, band-interleaved by
// plane (INTERLEAVE=BAND, image3.cpp:882-890); pixels no leaf covers keep their value.
int mis_rasterize_leaves(const mis_leaf* leaves, int n, int width, int height, float* out)
{
    if (n < 0 || (n > 0 && !leaves) || !out || width <= 0 || height <= 0) return MIS_ERR_ARG;
    const size_t plane = (size_t)width * height;
    for (int i = 0; i < n; i++) {
        const mis_leaf& l = leaves[i];
        const float disparity = sqrtf((float)(l.total_dx * l.total_dx + l.total_dy * l.total_dy));
        const int x0 = std::max(l.bbox1.minx, 0), y0 = std::max(l.bbox1.miny, 0);
        const int x1 = std::min(l.bbox1.minx + l.bbox1.width, width), y1 = std::min(l.bbox1.miny + l.bbox1.height, height);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const size_t o = (size_t)y * width + x;
                out[o] = disparity;
                out[plane + o] = (float)l.total_dx;
                out[2 * plane + o] = (float)l.total_dy;
            }
    }
    return MIS_OK;
}

}  // extern "C"
