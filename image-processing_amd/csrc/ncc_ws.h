// Workspace of the 8-bit dense NCC and the entry points its translation units share, gfx950.
//   csrc/ncc_kernels.hip         tables, vertical packs, vector argmax kernels, border, the call's orchestration
//   csrc/ncc_mfma_kernels.hip    argmax on the matrix cores (ncc_mfma.h) and its launches
//   csrc/ncc_mfma2_kernels.hip   the same file once more: the kernels that also write the second-peak plane
//   csrc/ncc_refine_kernels.hip  key merge, exact scores, subpixel: the refinement both argmax paths feed
// (several units so that their many template instantiations compile side by side)
#pragma once

#include "ncc_shared.h"

struct NccWs {
    // per-pixel tables, pitch P elements per row, H rows
    uint32_t* Lpack = nullptr;  // half2( L'(x,y), -L'(x,y-W) )      (0 outside the raster)
    float* negc = nullptr;      // -round(S_L/N)
    float* negg = nullptr;      // -(S_L - N*round(S_L/N))/N
    float* iL = nullptr;        // 1/sqrt(sum (L-mean)^2), 0 when the window is flat or does not fit
    uint32_t* Rpack = nullptr;  // half2( R'(x,y), R'(x,y-W) )
    float* Rsr = nullptr;       // S_R (window sum of R')
    float* Rir = nullptr;       // 1/sqrt(sum (R-mean)^2), 0 when flat, NaN where the window does not fit
    uint32_t* VL = nullptr;     // [ceil(H/4)][pitch]: 4 vertically adjacent pixels, (p-128) as int8 each
    uint32_t* VR = nullptr;
    float* keys = nullptr;      // [planes][H][W] best key per pixel and chunk of 64 disparities (vector kernels) or range of 96 right columns (matrix-core kernels)
    size_t keys_bytes = 0;
    size_t cap_px = 0;          // capacity in pixels (pitch*rows) of the arrays above
    int pitch = 0, rows = 0;
    uint64_t pair_serial = 0;   // tables valid for this pair ...
    int window = 0;             // ... this window ...
    int built_lo = 0, built_hi = 0;   // ... and these raster rows [built_lo, built_hi) (multiples of NCC_STATS_BAND, or H)
    // Second stream of the fast path: the small kernels nothing waits for until the refinement
    // (vertical packs, frame border, tail disparities) run beside the argmax kernel instead of in
    // front of it.  ev_fork orders the side stream behind the main one, ev_tables behind the tables,
    // ev_join the refinement behind the side stream.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_tables = nullptr, ev_join = nullptr, ev_main = nullptr;
    bool forked = false;        // the side stream carries work of the current call
    // The tables are built by the first call after a bind and reused by later calls, which may come
    // on another stream (mis_set_stream): ev_built marks the end of the building call on the stream
    // it ran on, and a reusing call on a different stream waits for it.
    hipEvent_t ev_built = nullptr;
    hipStream_t built_stream = nullptr;
    bool building = false;      // the current call builds the tables
};

// side stream of the current call, ordered behind the main stream / main stream waits for the side stream's work
int ncc_fork(mis_ctx* ctx, NccWs* w);
int ncc_join(mis_ctx* ctx, NccWs* w);

// Matrix-core argmax and refinement of rows [a.row0, a.row1) (tables built, window 2 r + 1 up to 31 x 31); d_peak2 may be null.
int ncc_mfma_launch(mis_ctx* ctx, int r, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2);

// Refinement on ctx->stream: merges nplanes key planes (mf: the matrix-core kernels' planes, else the vector kernels'),
// keys2: the second-peak planes or null (then d_peak2 is null too).
int ncc_refine_launch(mis_ctx* ctx, int r, bool mf, const NccArgs& a, int nplanes, const float* keys2,
                      float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2);
