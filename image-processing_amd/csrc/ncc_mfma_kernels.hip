// Dense NCC argmax on the matrix cores (8-bit pairs): the launches of ncc_mfma.h's kernels; see ncc_ws.h for the split.
#include "ncc_ws.h"
#include "ncc_mfma.h"

#include <stdlib.h>

// ---- edge ranges with compile-time tiles ----
// tiles (t, c) of range r: diagonal g = T r + t - c of the 32 x 32 tile grid; present while 0 <= g <= G, masked per element
// unless every element lies inside the disparity range (the kernel's own rule)
static void ncc_mf_tile_masks(int r, int nd, uint32_t& any_bits, uint32_t& part_bits)
{
    const int G = (nd + 30) >> 5;
    any_bits = part_bits = 0u;
    for (int t = 0; t < NCC_MF_T; t++)
        for (int c = 0; c < NCC_MF_NB; c++) {
            const int g = NCC_MF_T * r + t - c;
            const bool any = g >= 0 && g <= G, full = g >= 1 && 32 * g + 31 <= nd - 1;
            any_bits |= (uint32_t)any << (t * NCC_MF_NB + c);
            part_bits |= (uint32_t)(any && !full) << (t * NCC_MF_NB + c);
        }
}
// The patterns that have kernels: the first range of any search of 96 disparities or more; the last ranges of searches
// of 32 G + 1 disparities (+-64, +-128, +-256, +-512, ...: only diagonal G is masked): G = 1 (mod 3) ends in one range,
// G = 2 (mod 3) in two (the last one holds a single masked tile); and the last ranges of the general case, where the two
// last diagonals are masked, for G = 0, 1, 2 (mod 3).  1911 of the 2048 search sizes the API takes (all from 96 on but
// a few multiples of 32) find all their edge ranges here; the others run the kernel that tests its tiles at run time.
#define NCC_MF_PATTERNS(X) X(0x3Du, 0x09u) X(0x2Fu, 0x24u) X(0x3Fu, 0x10u) X(0x02u, 0x02u) X(0x0Bu, 0x0Bu) X(0x2Fu, 0x2Du) X(0x3Fu, 0x34u)
static bool ncc_mf_has_pattern(uint32_t any_bits, uint32_t part_bits)
{
#define NCC_MF_HAS(A, B) if (any_bits == A && part_bits == B) return true;
    NCC_MF_PATTERNS(NCC_MF_HAS)
#undef NCC_MF_HAS
    return false;
}
template <int R, bool WANT2>
static void ncc_mf_launch_pattern(uint32_t any_bits, uint32_t part_bits, dim3 grid, hipStream_t s, NccArgs a, int nblocks, int range,
                                  const uint32_t* Lpack, const float* negc, const float* negg, const uint32_t* Rpack,
                                  const float* Rsr, const float* Rir, float* keys, float* keys2)
{
#define NCC_MF_GO(A, B) if (any_bits == A && part_bits == B) { \
        hipLaunchKernelGGL((ncc_mfma_argmax_kernel<R, false, WANT2, A, B>), grid, dim3(64), 0, s, \
                           a, nblocks, 1, range, 0, Lpack, negc, negg, Rpack, Rsr, Rir, keys, keys2); return; }
    NCC_MF_PATTERNS(NCC_MF_GO)
#undef NCC_MF_GO
}
static const int spec_env = [] { const char* e = getenv("MISTEREO_NCC_EDGE_SPEC"); return e ? atoi(e) : 1; }();     // 0: the run-time edge kernel (A/B switch)

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
        // edge ranges whose tiles match a compile-time pattern run as one launch each, without the run-time tile tests
        bool spec = n_edge > 0 && spec_env != 0;
        uint32_t any_r[8], part_r[8]; int edge_r[8]; int ne = 0;
        for (int r = 0; spec && r < nranges; r++) {
            if (n_in > 0 && r >= p_lo && r <= p_hi) continue;
            if (ne == 8) { spec = false; break; }
            ncc_mf_tile_masks(r, nd, any_r[ne], part_r[ne]);
            if (any_r[ne] == 0u) continue;         // no candidates at all in this range: nothing to launch
            spec = ncc_mf_has_pattern(any_r[ne], part_r[ne]);
            edge_r[ne++] = r;
        }
        if (spec) {
            for (int i = 0; i < ne; i++)
                ncc_mf_launch_pattern<R, WANT2>(any_r[i], part_r[i], dim3((unsigned)(nblocks * a.nbands)), se,
                                                a, nblocks, edge_r[i], w->Lpack, w->negc, w->negg, w->Rpack, w->Rsr, w->Rir, w->keys, keys2);
        } else
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

// This file is compiled twice (the second time through ncc_mfma2_kernels.hip, which sets NCC_MF_WANT2): the kernels with
// and without the second-peak plane are two translation units that compile side by side.
#ifndef NCC_MF_WANT2
#define NCC_MF_WANT2 0
#endif
#if NCC_MF_WANT2
int ncc_mfma_launch_peak2(mis_ctx* ctx, int r, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
#define NCC_MCASE(RR) case RR: return ncc_launch_mfma<RR, true>(ctx, a, d_disp, d_peak, d_idisp, d_peak2);
#else
int ncc_mfma_launch_peak2(mis_ctx* ctx, int r, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2);
int ncc_mfma_launch(mis_ctx* ctx, int r, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    if (d_peak2) return ncc_mfma_launch_peak2(ctx, r, a, d_disp, d_peak, d_idisp, d_peak2);
#define NCC_MCASE(RR) case RR: return ncc_launch_mfma<RR, false>(ctx, a, d_disp, d_peak, d_idisp, nullptr);
#endif
    switch (r) {
        NCC_MCASE(1) NCC_MCASE(2) NCC_MCASE(3) NCC_MCASE(4) NCC_MCASE(5) NCC_MCASE(6) NCC_MCASE(7) NCC_MCASE(8)
        NCC_MCASE(9) NCC_MCASE(10) NCC_MCASE(11) NCC_MCASE(12) NCC_MCASE(13) NCC_MCASE(14) NCC_MCASE(15)
    }
#undef NCC_MCASE
    static_assert(NCC_MF_MAX_R == 15, "one case per window the matrix-core kernels take");
    return mis_fail(ctx, MIS_ERR_ARG, "dense NCC on the matrix cores: unsupported window %d", 2 * r + 1);
}
