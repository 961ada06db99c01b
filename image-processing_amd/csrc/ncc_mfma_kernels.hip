// Dense NCC argmax on the matrix cores (8-bit pairs): the launches of ncc_mfma.h's kernels; see ncc_ws.h for the split.
#include "ncc_ws.h"
#include "ncc_mfma.h"

// Matrix-core path: one key plane per range of 32*NCC_MF_T right columns, no tail kernel (the tiles on the
// edge of the disparity range are masked per element).
template <int R, bool WANT2>
static int ncc_launch_mfma(mis_ctx* ctx, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    NccWs* w = ctx->ncc;
    const int nd = a.dmax - a.dmin + 1;
    const int nranges = NccMfCfg<R>::ranges(nd);
    const int nblocks = (a.W - 2 * R + 63) / 64;
    const size_t plane_px = (size_t)a.H * a.W;
    int rc = mis_reserve(ctx, (void**)&w->keys, &w->keys_bytes, (size_t)nranges * plane_px * sizeof(float) * (WANT2 ? 2 : 1));
    if (rc) return rc;
    float* keys2 = WANT2 ? w->keys + (size_t)nranges * plane_px : nullptr;      // second-peak planes behind the first ones
    // rows per wavefront: every band re-runs the 2R warm-up rows (a fifth of the cost of a scored row each), so
    // taller is cheaper as long as the grid still holds several rounds of wavefronts for the chip's 2048 slots
    a.band_rows = 256;
    while (a.band_rows > 32 && (long long)nblocks * nranges * ((a.row1 - a.row0 + a.band_rows - 1) / a.band_rows) < 4 * 2048)
        a.band_rows /= 2;
    a.nbands = (a.row1 - a.row0 + a.band_rows - 1) / a.band_rows;
    mis_prof_mark(ctx);                            // tables done
    {
        // ranges whose tiles all lie inside the disparity range: diagonals T*p - (NB-1) .. T*p + T - 1 within [1, gF]
        const int gF = nd >= 32 ? (nd - 32) / 32 : -1;
        int p_lo = (NCC_MF_NB + NCC_MF_T - 1) / NCC_MF_T;
        int p_hi = gF - NCC_MF_T + 1 >= 0 ? (gF - NCC_MF_T + 1) / NCC_MF_T : -1;
        if (p_hi >= nranges) p_hi = nranges - 1;
        if (p_hi < p_lo) { p_lo = nranges; p_hi = nranges - 1; }      // none
        const int n_in = p_hi - p_lo + 1, n_edge = nranges - n_in;
        // the edge ranges go to the side stream (behind the tables of both images): the two launches share the
        // machine instead of running one after the other, each with a tail of its own
        hipStream_t se = ctx->stream;
        if (n_edge > 0 && n_in > 0) {
            rc = ncc_fork(ctx, w);
            if (rc) return rc;
            MIS_HIP(ctx, hipEventRecord(w->ev_main, ctx->stream));
            MIS_HIP(ctx, hipStreamWaitEvent(w->side, w->ev_main, 0));
            se = w->side;
        }
        if (n_edge > 0)
            hipLaunchKernelGGL((ncc_mfma_argmax_kernel<R, true, WANT2>), dim3((unsigned)(nblocks * a.nbands * n_edge)), dim3(64), 0, se,
                               a, nblocks, n_edge, n_in > 0 ? p_lo : nranges, p_hi + 1, w->Lpack, w->negc, w->negg, w->Rpack, w->Rsr, w->Rir, w->keys, keys2);
        if (n_in > 0)
            hipLaunchKernelGGL((ncc_mfma_argmax_kernel<R, false, WANT2>), dim3((unsigned)(nblocks * a.nbands * n_in)), dim3(64), 0, ctx->stream,
                               a, nblocks, n_in, p_lo, 0, w->Lpack, w->negc, w->negg, w->Rpack, w->Rsr, w->Rir, w->keys, keys2);
    }
    rc = ncc_join(ctx, w);                         // vertical packs, border, edge ranges
    if (rc) return rc;
    mis_prof_mark(ctx);
    rc = ncc_refine_launch(ctx, R, true, a, nranges, keys2, d_disp, d_peak, d_idisp, d_peak2);
    if (rc) return rc;
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

int ncc_mfma_launch(mis_ctx* ctx, int r, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
#define NCC_MCASE(RR) case RR: return d_peak2 ? ncc_launch_mfma<RR, true>(ctx, a, d_disp, d_peak, d_idisp, d_peak2) \
                                              : ncc_launch_mfma<RR, false>(ctx, a, d_disp, d_peak, d_idisp, nullptr);
    switch (r) {
        NCC_MCASE(1) NCC_MCASE(2) NCC_MCASE(3) NCC_MCASE(4) NCC_MCASE(5) NCC_MCASE(6) NCC_MCASE(7) NCC_MCASE(8)
        NCC_MCASE(9) NCC_MCASE(10) NCC_MCASE(11) NCC_MCASE(12) NCC_MCASE(13) NCC_MCASE(14) NCC_MCASE(15)
    }
#undef NCC_MCASE
    static_assert(NCC_MF_MAX_R == 15, "one case per window the matrix-core kernels take");
    return mis_fail(ctx, MIS_ERR_ARG, "dense NCC on the matrix cores: unsupported window %d", 2 * r + 1);
}
