// Internal declarations shared by the host-side translation units (plain C++ above the C entry points).
#pragma once

#include <functional>

#include "mi_stereo.h"

// Batched findOffset of one recursion level and tile size: (r1, r2, n, tile_w, tile_h, out) -> status.
typedef std::function<int(const mis_rect*, const mis_rect*, int, int, int, mis_offset*)> MisTileEval;

// calcDisparityMap (image3.cpp:676-865), level-synchronous, on any evaluator of tile batches (one context,
// or a group of devices that shares a level's tiles out).  Arguments as mis_disparity_map.
int mis_disparity_map_with2(const MisTileEval& eval, int width1, int height1, int width2, int height2, int min_size,
                            mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles);
int mis_disparity_map_with(const MisTileEval& eval, int image_width, int image_height, int min_size,
                           mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles);
