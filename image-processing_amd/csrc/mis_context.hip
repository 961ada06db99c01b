// Context, error reporting and image-pair management of libmistereo.so.
//
// The context replaces the reference's process globals (image3.cpp:340-344) and its per-call
// allocate / plan / free cycle (image3.cpp:357-372, 421-425): device buffers are owned here and
// reused across calls.

#include "mis_common.h"

#include <string.h>
#include <thread>

#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <mutex>

static std::string g_create_err;
static std::mutex g_create_mu;

int mis_fail(mis_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else { std::lock_guard<std::mutex> g(g_create_mu); g_create_err = buf; }
    return code;
}

int mis_reserve(mis_ctx* ctx, void** buf, size_t* cap, size_t bytes)
{
    if (*cap >= bytes && *buf) return MIS_OK;
    if (*buf) { MIS_HIP(ctx, hipFree(*buf)); *buf = nullptr; *cap = 0; }
    MIS_HIP(ctx, hipMalloc(buf, bytes));
    *cap = bytes;
    return MIS_OK;
}

hipEvent_t mis_prof_mark(mis_ctx* ctx)
{
    if (!ctx->profile) return nullptr;
    if (ctx->prof_used == ctx->prof_events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->prof_events.push_back(e);
    }
    hipEvent_t e = ctx->prof_events[ctx->prof_used++];
    (void)hipEventRecord(e, ctx->stream);
    return e;
}

// The runtime's own path for pageable destinations moves about 10 GB/s (one staging thread); page-locked
// staging plus a few copying threads is several times that.
static void host_copy_parallel(void* dst, const void* src, size_t bytes)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : (nt > 8 ? 8 : nt);
    if (bytes < ((size_t)4 << 20)) nt = 1;
    if (nt == 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
    for (unsigned i = 0; i < nt; i++) {
        const size_t o = (size_t)i * per;
        if (o >= bytes) break;
        const size_t m = bytes - o < per ? bytes - o : per;
        // no exception may leave the library: a part whose thread cannot be started is copied here
        try {
            th.emplace_back([=] { memcpy((char*)dst + o, (const char*)src + o, m); });
        } catch (...) {
            memcpy((char*)dst + o, (const char*)src + o, m);
        }
    }
    for (auto& t : th) t.join();
}

// ranges page-locked by mis_pin_host: the caller's pointer -> the page-aligned base that was registered
static std::mutex g_pin_mu;
static std::map<void*, void*> g_pinned;

// true when [p, p + bytes) is page-locked host memory the device can copy to directly (mis_pin_host, hipHostRegister
// or hipHostMalloc by the caller); pageable memory makes hipPointerGetAttributes fail, which is not an error here
bool mis_host_pinned(const void* p, size_t bytes)
{
    if (!p || !bytes) return false;
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipPointerGetAttributes(&a1, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
}

int mis_wait_upload(mis_ctx* ctx, int row_hi, hipStream_t s)
{
    for (int k = 0; k < ctx->up_n; k++)
        if (ctx->up_rows[k] >= row_hi || k == ctx->up_n - 1) {
            MIS_HIP(ctx, hipStreamWaitEvent(s, ctx->up_done[k], 0));
            break;
        }
    return MIS_OK;
}

int mis_copy_back(mis_ctx* ctx, const MisCopyJob* jobs, int njobs)
{
    if (!ctx->copy_stream) MIS_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    {
        // every destination page-locked: the device writes straight into the caller's rasters, no staging, no host copy
        bool all_pinned = njobs > 0;
        for (int j = 0; j < njobs && all_pinned; j++) all_pinned = mis_host_pinned(jobs[j].dst, jobs[j].bytes);
        if (all_pinned) {
            for (int j = 0; j < njobs; j++) {
                if (jobs[j].after) MIS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, jobs[j].after, 0));
                MIS_HIP(ctx, hipMemcpyAsync(jobs[j].dst, jobs[j].src, jobs[j].bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
            }
            return MIS_OK;            // the caller synchronises the copy stream
        }
    }
    if (!ctx->pinned) {
        MIS_HIP(ctx, hipHostMalloc(&ctx->pinned, 2 * MIS_STAGE_BYTES, hipHostMallocDefault));
        for (int i = 0; i < 2; i++) MIS_HIP(ctx, hipEventCreateWithFlags(&ctx->slot_ready[i], hipEventDisableTiming));
    }
    hipStream_t cs = ctx->copy_stream;
    // pieces in flight: (host destination, slot, bytes)
    struct Piece { void* dst; int slot; size_t bytes; };
    Piece pending = { nullptr, 0, 0 };
    int slot = 0;
    auto drain = [&](const Piece& p) -> int {
        if (!p.dst) return MIS_OK;
        MIS_HIP(ctx, hipEventSynchronize(ctx->slot_ready[p.slot]));
        host_copy_parallel(p.dst, (char*)ctx->pinned + (size_t)p.slot * MIS_STAGE_BYTES, p.bytes);
        return MIS_OK;
    };
    for (int j = 0; j < njobs; j++) {
        if (jobs[j].after) MIS_HIP(ctx, hipStreamWaitEvent(cs, jobs[j].after, 0));
        for (size_t o = 0; o < jobs[j].bytes; o += MIS_STAGE_BYTES) {
            const size_t m = jobs[j].bytes - o < MIS_STAGE_BYTES ? jobs[j].bytes - o : MIS_STAGE_BYTES;
            // the slot about to be refilled was drained one piece ago (two slots, one piece pending)
            MIS_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + (size_t)slot * MIS_STAGE_BYTES, (const char*)jobs[j].src + o, m,
                                        hipMemcpyDeviceToHost, cs));
            MIS_HIP(ctx, hipEventRecord(ctx->slot_ready[slot], cs));
            int rc = drain(pending);                      // host copy of the previous piece while this one travels
            if (rc) return rc;
            pending = { (char*)jobs[j].dst + o, slot, m };
            slot ^= 1;
        }
    }
    return drain(pending);
}

extern "C" {

int mis_alloc_host(mis_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!out || !bytes) return mis_fail(ctx, MIS_ERR_ARG, "host allocation: empty request");
    *out = nullptr;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MIS_HIP(ctx, hipHostMalloc(out, bytes, hipHostMallocPortable));
    return MIS_OK;
}

int mis_free_host(mis_ctx* ctx, void* ptr)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ptr) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MIS_HIP(ctx, hipHostFree(ptr));
    return MIS_OK;
}

int mis_pin_host(mis_ctx* ctx, void* ptr, size_t bytes)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ptr || !bytes) return mis_fail(ctx, MIS_ERR_ARG, "pin: empty range");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    if (mis_host_pinned(ptr, bytes)) return MIS_OK;
    // whole pages: the range is widened to page boundaries (the neighbouring bytes of those pages are locked with it)
    const uintptr_t page = 4096, lo = (uintptr_t)ptr & ~(page - 1), hi = ((uintptr_t)ptr + bytes + page - 1) & ~(page - 1);
    MIS_HIP(ctx, hipHostRegister((void*)lo, hi - lo, hipHostRegisterDefault));
    std::lock_guard<std::mutex> g(g_pin_mu);
    g_pinned[ptr] = (void*)lo;
    return MIS_OK;
}

int mis_unpin_host(mis_ctx* ctx, void* ptr)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ptr) return mis_fail(ctx, MIS_ERR_ARG, "unpin: NULL");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        auto it = g_pinned.find(ptr);
        if (it == g_pinned.end()) return mis_fail(ctx, MIS_ERR_STATE, "unpin: %p was not pinned by mis_pin_host", ptr);
        base = it->second;
        g_pinned.erase(it);
    }
    MIS_HIP(ctx, hipHostUnregister(base));
    return MIS_OK;
}

int mis_profile_enable(mis_ctx* ctx, int on)
{
    if (!ctx) return MIS_ERR_ARG;
    ctx->profile = on != 0;
    ctx->prof_used = 0;
    return MIS_OK;
}

int mis_profile_read(mis_ctx* ctx, double* tables_ms, double* main_ms, double* refine_ms, int* calls)
{
    if (!ctx) return MIS_ERR_ARG;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double sum[3] = { 0, 0, 0 };
    int n = 0;
    for (size_t i = 0; i + 3 < ctx->prof_used && i + 3 < ctx->prof_events.size(); i += 4) {
        for (int k = 0; k < 3; k++) {
            float ms = 0;
            MIS_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i + k], ctx->prof_events[i + k + 1]));
            sum[k] += ms;
        }
        n++;
    }
    ctx->prof_used = 0;
    if (tables_ms) *tables_ms = sum[0];
    if (main_ms) *main_ms = sum[1];
    if (refine_ms) *refine_ms = sum[2];
    if (calls) *calls = n;
    return MIS_OK;
}

const char* mis_version(void) { return "libmistereo gfx950 " __DATE__; }

int mis_create(int device_id, mis_ctx** out)
{
    if (!out) return mis_fail(nullptr, MIS_ERR_ARG, "mis_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return mis_fail(nullptr, MIS_ERR_DEVICE, "mis_create: no HIP device (%s); there is no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "count 0");
    if (device_id < 0 || device_id >= n)
        return mis_fail(nullptr, MIS_ERR_ARG, "mis_create: device %d out of range (have %d)", device_id, n);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return mis_fail(nullptr, MIS_ERR_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return mis_fail(nullptr, MIS_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return mis_fail(nullptr, MIS_ERR_UNSUPPORTED, "mis_create: device %d is %s; this library is built for gfx950 only",
                        device_id, prop.gcnArchName);
    mis_ctx* c = new mis_ctx();
    c->device = device_id;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return mis_fail(nullptr, MIS_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    if (mis_tri_init_device(c) != MIS_OK) {
        const std::string msg = c->err;
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return mis_fail(nullptr, MIS_ERR_DEVICE, "mis_create: %s", msg.c_str());
    }
    *out = c;
    return MIS_OK;
}

int mis_destroy(mis_ctx* ctx)
{
    if (!ctx) return MIS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    mis_ncc_release(ctx);
    mis_ncc16_release(ctx);
    mis_xcorr_release(ctx);
    if (ctx->tri_buf) (void)hipFree(ctx->tri_buf);
    if (ctx->out_buf) (void)hipFree(ctx->out_buf);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->up_stream) { (void)hipStreamSynchronize(ctx->up_stream); (void)hipStreamDestroy(ctx->up_stream); }
    if (ctx->up_start) (void)hipEventDestroy(ctx->up_start);
    for (hipEvent_t e : ctx->up_done) if (e) (void)hipEventDestroy(e);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (hipEvent_t e : ctx->slot_ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->strip_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) if (ctx->owned_img[i]) (void)hipFree(ctx->owned_img[i]);
    for (void* p : ctx->owned_bands) (void)hipFree(p);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MIS_OK;
}

const char* mis_last_error(const mis_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> g(g_create_mu);
    return g_create_err.c_str();
}

int mis_set_stream(mis_ctx* ctx, void* hip_stream)
{
    if (!ctx) return MIS_ERR_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MIS_OK;
}

int mis_synchronize(mis_ctx* ctx)
{
    if (!ctx) return MIS_ERR_ARG;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->up_n) MIS_HIP(ctx, hipStreamSynchronize(ctx->up_stream));      // a pair left in flight by mis_upload_pair
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIS_OK;
}

static void drop_extra_bands(mis_ctx* ctx)
{
    ctx->band_img[0].clear(); ctx->band_img[1].clear();
    for (void* p : ctx->owned_bands) (void)hipFree(p);
    ctx->owned_bands.clear();
}

static int check_pair_args(mis_ctx* ctx, const void* a, const void* b, int dtype, int w, int h, size_t stride)
{
    if (!ctx) return MIS_ERR_ARG;
    int bpp = mis_pixel_bytes(dtype);
    if (!a || !b) return mis_fail(ctx, MIS_ERR_ARG, "image pointer is NULL");
    if (!bpp) return mis_fail(ctx, MIS_ERR_ARG, "unknown pixel type %d", dtype);
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) return mis_fail(ctx, MIS_ERR_ARG, "bad raster size %d x %d", w, h);
    if (stride < (size_t)w * bpp) return mis_fail(ctx, MIS_ERR_ARG, "row stride %zu smaller than a row (%zu)", stride, (size_t)w * bpp);
    if (stride % bpp) return mis_fail(ctx, MIS_ERR_ARG, "row stride %zu not a multiple of the pixel size", stride);
    return MIS_OK;
}

int mis_bind_pair_device(mis_ctx* ctx, const void* d_img1, const void* d_img2, int dtype,
                         int width, int height, size_t row_stride_bytes)
{
    int rc = check_pair_args(ctx, d_img1, d_img2, dtype, width, height, row_stride_bytes);
    if (rc) return rc;
    drop_extra_bands(ctx);
    ctx->img[0] = d_img1; ctx->img[1] = d_img2;
    ctx->dtype = dtype; ctx->width = width; ctx->height = height; ctx->stride = row_stride_bytes;
    ctx->pair_serial++;
    ctx->up_n = 0;
    return MIS_OK;
}

int mis_invalidate_pair(mis_ctx* ctx)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "invalidate: no image pair bound");
    ctx->pair_serial++;             // every table derived from the rasters is rebuilt by the next call that needs it
    return MIS_OK;
}

int mis_upload_pair(mis_ctx* ctx, const void* img1, const void* img2, int dtype,
                    int width, int height, size_t row_stride_bytes)
{
    int rc = check_pair_args(ctx, img1, img2, dtype, width, height, row_stride_bytes);
    if (rc) return rc;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    size_t rowb = (size_t)width * mis_pixel_bytes(dtype);
    size_t bytes = rowb * height;
    // an earlier pair still on its way into the context's rasters: it must have landed before they are rewritten or freed
    if (ctx->up_n) { MIS_HIP(ctx, hipStreamSynchronize(ctx->up_stream)); ctx->up_n = 0; }
    if (ctx->owned_img_bytes < bytes) {
        // the bound pair may be these very buffers: unbind before they are freed, so that a failed
        // allocation below leaves the context without a pair instead of with dangling pointers
        if (ctx->img[0] == ctx->owned_img[0] || ctx->img[1] == ctx->owned_img[1]) {
            drop_extra_bands(ctx);
            ctx->img[0] = ctx->img[1] = nullptr;
            ctx->pair_serial++;
        }
        for (int i = 0; i < 2; i++) {
            if (ctx->owned_img[i]) { MIS_HIP(ctx, hipFree(ctx->owned_img[i])); ctx->owned_img[i] = nullptr; }
        }
        ctx->owned_img_bytes = 0;
        for (int i = 0; i < 2; i++) MIS_HIP(ctx, hipMalloc(&ctx->owned_img[i], bytes));
        ctx->owned_img_bytes = bytes;
    }
    const void* src[2] = { img1, img2 };
    if (row_stride_bytes == rowb && height >= 64 && mis_host_pinned(img1, bytes) && mis_host_pinned(img2, bytes)) {
        // Page-locked rasters: the pair travels by DMA in row chunks on a stream of its own and the call returns at
        // once; the entry points that read the pair wait (on the device) for the chunks that hold their rows, so the
        // first strip of a dense correlation starts while the rest of the pair is still on its way.
        constexpr int NCH = 8;
        if (!ctx->up_stream) {
            MIS_HIP(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
            MIS_HIP(ctx, hipEventCreateWithFlags(&ctx->up_start, hipEventDisableTiming));
            for (int k = 0; k < NCH; k++) MIS_HIP(ctx, hipEventCreateWithFlags(&ctx->up_done[k], hipEventDisableTiming));
        }
        // the buffers may still be read by work queued on the context's stream
        MIS_HIP(ctx, hipEventRecord(ctx->up_start, ctx->stream));
        MIS_HIP(ctx, hipStreamWaitEvent(ctx->up_stream, ctx->up_start, 0));
        rc = mis_bind_pair_device(ctx, ctx->owned_img[0], ctx->owned_img[1], dtype, width, height, rowb);
        if (rc) return rc;
        for (int k = 0; k < NCH; k++) {
            const int y0 = (int)((long long)height * k / NCH), y1 = (int)((long long)height * (k + 1) / NCH);
            for (int i = 0; i < 2; i++)
                MIS_HIP(ctx, hipMemcpyAsync((char*)ctx->owned_img[i] + (size_t)y0 * rowb, (const char*)src[i] + (size_t)y0 * rowb,
                                            (size_t)(y1 - y0) * rowb, hipMemcpyHostToDevice, ctx->up_stream));
            MIS_HIP(ctx, hipEventRecord(ctx->up_done[k], ctx->up_stream));
            ctx->up_rows[k] = y1;
        }
        ctx->up_n = NCH;
        return MIS_OK;
    }
    for (int i = 0; i < 2; i++) {
        if (row_stride_bytes == rowb)       // one contiguous block: much faster than a 2-D copy from pageable memory
            MIS_HIP(ctx, hipMemcpyAsync(ctx->owned_img[i], src[i], bytes, hipMemcpyHostToDevice, ctx->stream));
        else
            MIS_HIP(ctx, hipMemcpy2DAsync(ctx->owned_img[i], rowb, src[i], row_stride_bytes, rowb, height,
                                          hipMemcpyHostToDevice, ctx->stream));
    }
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return mis_bind_pair_device(ctx, ctx->owned_img[0], ctx->owned_img[1], dtype, width, height, rowb);
}

int mis_upload_pair2(mis_ctx* ctx, const void* img1, int width1, int height1, size_t row_stride1_bytes,
                     const void* img2, int width2, int height2, size_t row_stride2_bytes, int dtype)
{
    int rc = check_pair_args(ctx, img1, img2, dtype, width1, height1, row_stride1_bytes);
    if (rc) return rc;
    rc = check_pair_args(ctx, img1, img2, dtype, width2, height2, row_stride2_bytes);
    if (rc) return rc;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const int W = width1 > width2 ? width1 : width2, H = height1 > height2 ? height1 : height2;
    const size_t bpp = (size_t)mis_pixel_bytes(dtype), rowb = (size_t)W * bpp, bytes = rowb * H;
    if (ctx->up_n) { MIS_HIP(ctx, hipStreamSynchronize(ctx->up_stream)); ctx->up_n = 0; }   // an earlier pair still on its way into these rasters
    if (ctx->owned_img_bytes < bytes) {
        if (ctx->img[0] == ctx->owned_img[0] || ctx->img[1] == ctx->owned_img[1]) {
            drop_extra_bands(ctx);
            ctx->img[0] = ctx->img[1] = nullptr;
            ctx->pair_serial++;
        }
        for (int i = 0; i < 2; i++) {
            if (ctx->owned_img[i]) { MIS_HIP(ctx, hipFree(ctx->owned_img[i])); ctx->owned_img[i] = nullptr; }
        }
        ctx->owned_img_bytes = 0;
        for (int i = 0; i < 2; i++) MIS_HIP(ctx, hipMalloc(&ctx->owned_img[i], bytes));
        ctx->owned_img_bytes = bytes;
    }
    const void* src[2] = { img1, img2 };
    const int ws[2] = { width1, width2 }, hs[2] = { height1, height2 };
    const size_t strides[2] = { row_stride1_bytes, row_stride2_bytes };
    for (int i = 0; i < 2; i++) {
        if (ws[i] != W || hs[i] != H) MIS_HIP(ctx, hipMemsetAsync(ctx->owned_img[i], 0, bytes, ctx->stream));
        MIS_HIP(ctx, hipMemcpy2DAsync(ctx->owned_img[i], rowb, src[i], strides[i], (size_t)ws[i] * bpp, hs[i],
                                      hipMemcpyHostToDevice, ctx->stream));
    }
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return mis_bind_pair_device(ctx, ctx->owned_img[0], ctx->owned_img[1], dtype, W, H, rowb);
}

int mis_set_extra_bands_device(mis_ctx* ctx, int n_extra, const void* const* d_img1, const void* const* d_img2)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "extra bands: bind the first band with mis_upload_pair / mis_bind_pair_device first");
    if (n_extra < 0 || n_extra > 15 || (n_extra > 0 && (!d_img1 || !d_img2))) return mis_fail(ctx, MIS_ERR_ARG, "extra bands: bad arguments");
    for (int i = 0; i < n_extra; i++)
        if (!d_img1[i] || !d_img2[i]) return mis_fail(ctx, MIS_ERR_ARG, "extra bands: band %d is NULL", i + 2);
    drop_extra_bands(ctx);
    for (int i = 0; i < n_extra; i++) { ctx->band_img[0].push_back(d_img1[i]); ctx->band_img[1].push_back(d_img2[i]); }
    return MIS_OK;
}

int mis_upload_extra_bands(mis_ctx* ctx, int n_extra, const void* const* img1, const void* const* img2, size_t row_stride_bytes)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "extra bands: bind the first band with mis_upload_pair / mis_bind_pair_device first");
    if (n_extra < 0 || n_extra > 15 || (n_extra > 0 && (!img1 || !img2))) return mis_fail(ctx, MIS_ERR_ARG, "extra bands: bad arguments");
    const size_t rowb = (size_t)ctx->width * mis_pixel_bytes(ctx->dtype);
    if (row_stride_bytes < rowb) return mis_fail(ctx, MIS_ERR_ARG, "extra bands: row stride smaller than a row");
    if (ctx->stride != rowb) return mis_fail(ctx, MIS_ERR_STATE, "extra bands: the bound pair must be tightly packed (upload it with mis_upload_pair)");
    for (int i = 0; i < n_extra; i++)
        if (!img1[i] || !img2[i]) return mis_fail(ctx, MIS_ERR_ARG, "extra bands: band %d is NULL", i + 2);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    drop_extra_bands(ctx);
    // every failure below leaves the context without extra bands (never with lists of unequal length)
    hipError_t e = hipSuccess;
    for (int i = 0; i < n_extra && e == hipSuccess; i++) {
        const void* src[2] = { img1[i], img2[i] };
        for (int s = 0; s < 2 && e == hipSuccess; s++) {
            void* d = nullptr;
            e = hipMalloc(&d, rowb * ctx->height);
            if (e != hipSuccess) break;
            ctx->owned_bands.push_back(d);
            if (row_stride_bytes == rowb)
                e = hipMemcpyAsync(d, src[s], rowb * ctx->height, hipMemcpyHostToDevice, ctx->stream);
            else
                e = hipMemcpy2DAsync(d, rowb, src[s], row_stride_bytes, rowb, ctx->height, hipMemcpyHostToDevice, ctx->stream);
            ctx->band_img[s].push_back(d);
        }
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) {
        drop_extra_bands(ctx);
        return mis_fail(ctx, MIS_ERR_DEVICE, "extra bands: upload failed: %s", hipGetErrorString(e));
    }
    return MIS_OK;
}

}  // extern "C"
