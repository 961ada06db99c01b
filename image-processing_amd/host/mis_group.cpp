// Several GPUs driven by one host process: a group of contexts, one per listed device, and the dense
// correlation of one host-resident pair split into row strips over them.  Plain C++ above the library's
// own C entry points (mis_upload_pair, mis_ncc_dense_rows): one host thread per device uploads the
// strip's rows plus the window halo, computes the strip and copies it into the caller's rasters.
// The reference has no counterpart (single process, image3.cpp:974-975).
#include "mi_stereo.h"
#include "mis_host.h"

#include <cstdarg>
#include <cstdio>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct mis_group {
    std::vector<mis_ctx*> ctx;
    std::string err;
};

// message of the last failed mis_create_group on this thread (there is no group to hold it)
static thread_local std::string g_group_create_err;

static int group_fail(mis_group* g, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf; else g_group_create_err = buf;
    return code;
}

// runs work(i) for every slot of the group, slot 0 on the calling thread; false when a thread could not be started
template <class F>
static bool group_parallel(int n_dev, F&& work)
{
    std::vector<std::thread> th;
    bool spawned = true;
    try {
        th.reserve(n_dev);
        for (int i = 1; i < n_dev; i++) th.emplace_back(work, i);
    } catch (...) {
        spawned = false;                          // the threads that did start are joined below
    }
    if (spawned) work(0);
    for (std::thread& t : th) t.join();
    return spawned;
}

extern "C" {

int mis_create_group(const int* device_ids, int n_dev, mis_group** out)
{
    if (!out) return group_fail(nullptr, MIS_ERR_ARG, "group: out is NULL");
    *out = nullptr;
    if (!device_ids || n_dev < 1 || n_dev > 64) return group_fail(nullptr, MIS_ERR_ARG, "group: 1..64 devices expected, got %d", n_dev);
    mis_group* g = new (std::nothrow) mis_group();
    if (!g) return group_fail(nullptr, MIS_ERR_DEVICE, "group: out of host memory");
    try {
        for (int i = 0; i < n_dev; i++) {
            mis_ctx* c = nullptr;
            const int rc = mis_create(device_ids[i], &c);
            if (rc) {
                group_fail(nullptr, rc, "group: device %d: %s", device_ids[i], mis_last_error(nullptr));
                for (mis_ctx* d : g->ctx) mis_destroy(d);
                delete g;
                return rc;
            }
            g->ctx.push_back(c);
        }
    } catch (...) {
        for (mis_ctx* d : g->ctx) mis_destroy(d);
        delete g;
        return group_fail(nullptr, MIS_ERR_DEVICE, "group: out of host memory");
    }
    *out = g;
    return MIS_OK;
}

int mis_destroy_group(mis_group* g)
{
    if (!g) return MIS_OK;
    int rc = MIS_OK;
    for (mis_ctx* c : g->ctx) {
        const int r = mis_destroy(c);
        if (!rc) rc = r;
    }
    delete g;
    return rc;
}

int mis_group_size(const mis_group* g) { return g ? (int)g->ctx.size() : 0; }

const char* mis_group_last_error(const mis_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }

int mis_group_ncc_dense(mis_group* g, const void* img1, const void* img2, int dtype,
                        int width, int height, size_t row_stride_bytes, int window, int dmin, int dmax,
                        float* disp, float* peak, int16_t* idisp, float* peak2)
{
    if (!g) return MIS_ERR_ARG;
    g->err.clear();
    if (!img1 || !img2 || !disp || !peak) return group_fail(g, MIS_ERR_ARG, "group dense NCC: NULL raster");
    if (width <= 0 || height <= 0) return group_fail(g, MIS_ERR_ARG, "group dense NCC: raster %d x %d", width, height);
    if (window < 3 || window > 31 || !(window & 1)) return group_fail(g, MIS_ERR_ARG, "group dense NCC: window %d not odd in 3..31", window);
    const int n_dev = (int)g->ctx.size(), r = window / 2;
    const int base = height / n_dev, rem = height % n_dev;
    std::vector<int> rcs(n_dev, MIS_OK);
    auto work = [&](int i) {
        // output rows [r0, r0 + n) of the frame; rows [lo, hi) of the rasters go to the device
        const int r0 = i * base + (i < rem ? i : rem), n = base + (i < rem ? 1 : 0);
        if (n == 0) return;
        const int lo = r0 - r > 0 ? r0 - r : 0, hi = r0 + n + r < height ? r0 + n + r : height;
        mis_ctx* c = g->ctx[i];
        const char* p1 = (const char*)img1 + (size_t)lo * row_stride_bytes;
        const char* p2 = (const char*)img2 + (size_t)lo * row_stride_bytes;
        int rc = mis_upload_pair(c, p1, p2, dtype, width, hi - lo, row_stride_bytes);
        // the strip's rasters start at frame row lo: addressed from there, the device's row k is the frame's row lo + k
        const size_t o = (size_t)lo * width;
        if (!rc) rc = mis_ncc_dense_rows(c, window, dmin, dmax, r0 - lo, n, disp + o, peak + o,
                                         idisp ? idisp + o : nullptr, peak2 ? peak2 + o : nullptr);
        rcs[i] = rc;
    };
    if (!group_parallel(n_dev, work)) return group_fail(g, MIS_ERR_DEVICE, "group dense NCC: could not start a host thread");
    for (int i = 0; i < n_dev; i++)
        if (rcs[i]) return group_fail(g, rcs[i], "group dense NCC: device slot %d: %s", i, mis_last_error(g->ctx[i]));
    return MIS_OK;
}

int mis_group_upload_pair(mis_group* g, const void* img1, const void* img2, int dtype,
                          int width, int height, size_t row_stride_bytes)
{
    if (!g) return MIS_ERR_ARG;
    g->err.clear();
    const int n_dev = (int)g->ctx.size();
    std::vector<int> rcs(n_dev, MIS_OK);
    auto work = [&](int i) { rcs[i] = mis_upload_pair(g->ctx[i], img1, img2, dtype, width, height, row_stride_bytes); };
    if (!group_parallel(n_dev, work)) return group_fail(g, MIS_ERR_DEVICE, "group upload: could not start a host thread");
    for (int i = 0; i < n_dev; i++)
        if (rcs[i]) return group_fail(g, rcs[i], "group upload: device slot %d: %s", i, mis_last_error(g->ctx[i]));
    return MIS_OK;
}

int mis_group_xcorr_tiles(mis_group* g, const mis_rect* r1, const mis_rect* r2, int n,
                          int tile_w, int tile_h, mis_offset* out)
{
    if (!g) return MIS_ERR_ARG;
    g->err.clear();
    if (n < 0 || (n > 0 && (!r1 || !r2 || !out))) return group_fail(g, MIS_ERR_ARG, "group tile correlation: bad tile buffers");
    if (n == 0) return MIS_OK;
    // contiguous shares of the batch, the remainder tiles to the first devices; a share of a few tiles is not worth
    // a device of its own (the first levels of the recursion hold 1, 4, 16 tiles)
    const int n_all = (int)g->ctx.size();
    const int n_dev = n < 4 * n_all ? (n + 3) / 4 < n_all ? (n + 3) / 4 : n_all : n_all;
    const int base = n / n_dev, rem = n % n_dev;
    std::vector<int> rcs(n_dev, MIS_OK);
    auto work = [&](int i) {
        const int t0 = i * base + (i < rem ? i : rem), nt = base + (i < rem ? 1 : 0);
        if (nt > 0) rcs[i] = mis_xcorr_tiles(g->ctx[i], r1 + t0, r2 + t0, nt, tile_w, tile_h, out + t0);
    };
    if (!group_parallel(n_dev, work)) return group_fail(g, MIS_ERR_DEVICE, "group tile correlation: could not start a host thread");
    for (int i = 0; i < n_dev; i++)
        if (rcs[i]) return group_fail(g, rcs[i], "group tile correlation: device slot %d: %s", i, mis_last_error(g->ctx[i]));
    return MIS_OK;
}

int mis_group_disparity_map(mis_group* g, int image_width, int image_height, int min_size,
                            mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles)
{
    if (!g) return MIS_ERR_ARG;
    g->err.clear();
    try {
        const MisTileEval eval = [g](const mis_rect* r1, const mis_rect* r2, int n, int w, int h, mis_offset* out) {
            return mis_group_xcorr_tiles(g, r1, r2, n, w, h, out);
        };
        const int rc = mis_disparity_map_with(eval, image_width, image_height, min_size, leaves, max_leaves, n_leaves, n_tiles);
        // a failed tile batch left its own message (the evaluator's); anything else comes from the driver's checks
        if (rc && g->err.empty()) group_fail(g, rc, rc == MIS_ERR_STATE ? "group disparity map: no overlap" : "group disparity map: bad argument");
        return rc;
    } catch (...) {
        return group_fail(g, MIS_ERR_STATE, "group disparity map: out of host memory");
    }
}

}  // extern "C"
