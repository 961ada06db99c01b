// Several GPUs driven by one host process: a group of contexts, one per listed device, and the dense
// correlation of one host-resident pair split into row strips over them.  Plain C++ above the library's
// own C entry points (mis_upload_pair, mis_ncc_dense_rows): one host thread per device uploads the
// strip's rows plus the window halo, computes the strip and copies it into the caller's rasters.
// The reference has no counterpart (single process, image3.cpp:974-975).
#include "mi_stereo.h"

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

static std::string g_group_create_err;

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
    {
        std::vector<std::thread> th;
        bool spawned = true;
        try {
            th.reserve(n_dev);
            for (int i = 1; i < n_dev; i++) th.emplace_back(work, i);
        } catch (...) {
            spawned = false;                      // the threads that did start are joined below
        }
        if (spawned) work(0);
        for (std::thread& t : th) t.join();
        if (!spawned) return group_fail(g, MIS_ERR_DEVICE, "group dense NCC: could not start a host thread");
    }
    for (int i = 0; i < n_dev; i++)
        if (rcs[i]) return group_fail(g, rcs[i], "group dense NCC: device slot %d: %s", i, mis_last_error(g->ctx[i]));
    return MIS_OK;
}

}  // extern "C"
