// Internal declarations shared by the translation units of libmistereo.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "mi_stereo.h"

struct mis_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;   // created by mis_create
    hipStream_t stream = nullptr;       // stream work is enqueued on (own_stream by default)
    std::string err;

    // bound image pair
    const void* img[2] = { nullptr, nullptr };   // device pointers
    void* owned_img[2] = { nullptr, nullptr };   // set when mis_upload_pair allocated them
    size_t owned_img_bytes = 0;
    int dtype = -1, width = 0, height = 0;
    size_t stride = 0;                            // bytes
    uint64_t pair_serial = 0;                     // bumped on every bind/upload
    // further bands of the same two rasters (tile correlation only): device pointers, same
    // type / size / stride as img[]; dropped whenever a new pair is bound
    std::vector<const void*> band_img[2];
    std::vector<void*> owned_bands;               // set when mis_upload_extra_bands allocated them

    // dense-NCC workspace (see ncc_kernels.hip)
    struct NccWs* ncc = nullptr;
    // 16-bit dense-NCC workspace (see ncc16_kernels.hip)
    struct Ncc16Ws* ncc16 = nullptr;
    // FFT-correlation workspace (see xcorr_kernels.hip)
    struct XcorrWs* xc = nullptr;
    // optional per-call device timing of the dense-NCC kernels (mis_profile_*)
    bool profile = false;
    std::vector<hipEvent_t> prof_events;   // per call: start, tables done, main kernel done, end
    size_t prof_used = 0;                  // events handed out since the last read

    // triangulation scratch
    void* tri_buf = nullptr; size_t tri_bytes = 0;

    // host-buffer entry points: device-side result rasters (kept between calls) and the stream and
    // events that copy finished row strips back while later strips are still being computed
    void* out_buf = nullptr; size_t out_bytes = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t strip_done[8] = {};
    void* pinned = nullptr;               // two staging slots of MIS_STAGE_BYTES each (page-locked)
    hipEvent_t slot_ready[2] = {};
    // mis_upload_pair from page-locked rasters: the pair travels in row chunks on up_stream while the call returns;
    // chunk k (rows below up_rows[k] of both rasters) has arrived when up_done[k] completes.  up_n == 0: nothing in flight.
    hipStream_t up_stream = nullptr;
    hipEvent_t up_done[8] = {}, up_start = nullptr;
    int up_rows[8] = {};
    int up_n = 0;
};
// orders `s` behind the arrival of rows [0, row_hi) of a pair that mis_upload_pair left in flight (no-op otherwise)
int mis_wait_upload(mis_ctx* ctx, int row_hi, hipStream_t s);
#define MIS_STAGE_BYTES ((size_t)32 << 20)

// One piece of a device -> pageable host copy.
struct MisCopyJob { void* dst; const void* src; size_t bytes; hipEvent_t after; };
// Copies the jobs in order on ctx->copy_stream through the page-locked staging slots: the transfer
// of one piece overlaps the host-side copy (several threads) of the previous one.  A job starts once
// its `after` event (may be null) has completed.  Returns when everything has arrived.
int mis_copy_back(mis_ctx* ctx, const MisCopyJob* jobs, int njobs);
// true when [p, p + bytes) is page-locked host memory the device can copy to and from directly
bool mis_host_pinned(const void* p, size_t bytes);

int mis_fail(mis_ctx* ctx, int code, const char* fmt, ...);
#define MIS_HIP(ctx, call)                                                             \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (void)hipGetLastError(); /* reported here: a later launch check must not see it again */ \
            return mis_fail((ctx), MIS_ERR_DEVICE, "%s failed: %s (%s:%d)", #call,     \
                            hipGetErrorString(e_), __FILE__, __LINE__);                \
        }                                                                              \
    } while (0)

// grows *buf to at least `bytes` (device memory); contents are not preserved
int mis_reserve(mis_ctx* ctx, void** buf, size_t* cap, size_t bytes);

// next event of the profiling pool, recorded on the context's stream (nullptr: profiling off)
hipEvent_t mis_prof_mark(mis_ctx* ctx);

// looks up, once per device, the address of the triangulation's statically allocated counter slots
int mis_tri_init_device(mis_ctx* ctx);

void mis_ncc_release(mis_ctx* ctx);
void mis_ncc16_release(mis_ctx* ctx);
void mis_xcorr_release(mis_ctx* ctx);

static inline int mis_pixel_bytes(int dtype)
{
    return dtype == MIS_U8 ? 1 : dtype == MIS_U16 ? 2 : dtype == MIS_F32 ? 4 : 0;
}
