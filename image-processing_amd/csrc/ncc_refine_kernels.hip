// Refinement of the dense NCC (8-bit pairs), gfx950: merges the key planes of an argmax path, recomputes the exact
// scores of the winner and its two neighbours, subpixel vertex, optional second peak.  Shared by the matrix-core
// kernels (ncc_mfma.h) and the vector kernels (ncc_kernels.hip); see ncc_ws.h for the split.
#include "ncc_ws.h"
#include "ncc_mfma.h"          // NCC_MF_T: right-column tiles per key plane of the matrix-core kernels

// One workgroup = BW consecutive pixels of 4 consecutive rows (4 BW threads, a wavefront is
// 64 pixels of one row).  Neighbouring pixels need not agree on their disparity (occlusions,
// mismatches), so per-pixel gathers from global memory would touch whole cache lines for a few
// bytes each; instead the workgroup stages, once, every quad row its windows touch: the left
// columns of the run and the right columns of the run for the WHOLE search range, and each pixel
// then reads its 2R+1 left and 2R+3 right columns from LDS.
// BW = 64 or 128: runs of 64 pixels stage 1.7 times the columns per pixel but put twice the workgroups on a CU, whose
// phases (keys, staging, sums, tables) then overlap better -- as long as their LDS images still fit (ncc_refine_bw).
#ifndef NCC_REFINE_KEY_BATCH
#define NCC_REFINE_KEY_BATCH 5
#endif
template <int R, int BW>
struct NccRefineCfg {
    static constexpr int NQ = (2 * R + 7) / 4 + 1;       // quad rows the windows of 4 consecutive rows can touch
    static constexpr int LC = (BW + 2 * R + 3) & ~3;     // left columns of the run, row stride a multiple of 4
    static int right_cols(int nd) { return (BW + (nd - 1) + 2 * R + 3 + 3) & ~3; }
    static size_t lds_bytes(int nd) { return (size_t)NQ * (LC + right_cols(nd)) * sizeof(uint32_t); }
};
// the run width for a search range: 64 while six workgroups of that width fit the CU's 160 KB (+-256 under windows up to
// 25 x 25, +-384 under 21 x 21), else 128 (measured at 8192 x 4096, 21 x 21: +-64 0.85 / 0.95 ms, +-256 0.87 / 1.02, +-384
// 1.13 / 1.14, +-512 1.48 / 1.19, +-1000 3.07 / 1.98; 31 x 31 +-256, five workgroups: 1.61 / 1.55)
template <int R>
static int ncc_refine_bw(int nd) { return 6 * NccRefineCfg<R, 64>::lds_bytes(nd) <= 160 * 1024 ? 64 : 128; }

// WANT2: the second-peak output (best score further than one disparity from the winner).  Candidates:
// the winner's chunk contributes its second key plane (lanes further than one lane from its winner),
// every other chunk its best key -- except a neighbouring chunk whose best lane touches the winner
// across the chunk boundary (winner in lane 63, neighbour's best in lane 0, or the mirror case): that
// chunk contributes its second plane, which also leaves out the neighbour's lane 1 (or 62), i.e. the
// disparity two away from the winner, whose exact score is then computed here like the winner's.
// Key values carry 26 bits of the score (relative error below 8e-6).
// MF: key planes of the matrix-core kernels (a range of 32*NCC_MF_T right columns per plane, 25-bit score, column
// index in the range below) instead of the vector kernel's (a chunk of 64 disparities per plane, 26-bit score,
// 63 - lane below)
template <int R, bool WANT2, bool MF, int BW>
__global__ __launch_bounds__(4 * BW) void ncc_refine_kernel(
    NccArgs a, const uint32_t* __restrict__ VL, const uint32_t* __restrict__ VR,
    const float* __restrict__ gNegc, const float* __restrict__ gNegg, const float* __restrict__ gIL,
    const float* __restrict__ gRsr, const float* __restrict__ gRir, const float* __restrict__ gKeys, int nchunk,
    float* __restrict__ oDisp, float* __restrict__ oPeak, int16_t* __restrict__ oIdisp,
    const float* __restrict__ gKeys2, float* __restrict__ oPeak2)
{
    constexpr int range_tiles = MF ? NCC_MF_T : 0;
    using Cfg = NccRefineCfg<R, BW>;
    constexpr int NQ = Cfg::NQ, LC = Cfg::LC, NT = 4 * BW;
    extern __shared__ __attribute__((aligned(16))) uint32_t ncc_refine_lds[];
    const int NC = (BW + (a.dmax - a.dmin) + 2 * R + 3 + 3) & ~3;     // == Cfg::right_cols(nd)
    uint32_t* tL = ncc_refine_lds;               // [NQ][LC]
    uint32_t* tR = ncc_refine_lds + NQ * LC;     // [NQ][NC]

    const int tid = threadIdx.x;
    // a wavefront is 64 pixels of ONE row: the row index in a scalar register keeps the window-row masks,
    // the quad tests and the row addresses below on the scalar unit
    const int lx = tid & (BW - 1), ly = __builtin_amdgcn_readfirstlane(tid / BW);
    const int x0 = R + blockIdx.x * BW;
    const int y0 = a.row0 + 4 * blockIdx.y;
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = (x < a.W - R) && (y < a.row1);
    const int P = a.pitch;
    const size_t o = (size_t)y * a.out_pitch + x;
    // merge the chunk keys: the best truncated score wins; a second chunk whose truncated score is
    // EQUAL is remembered (keys carry 26 bits: two candidates closer than 8e-6 relative look alike),
    // and the two are told apart further down by their exact scores
    int ds = MIS_INVALID_DISP, alt = MIS_INVALID_DISP;
    float sec = -1.0e30f;                          // WANT2: best second-peak candidate so far (argmax-kernel scale)
    int xd = 0;                                    // WANT2: +2 / -2 when the disparity two away needs its exact score
    const size_t plane = (size_t)a.H * a.out_pitch;
    constexpr int NKB = MF ? 6 : NCC_REFINE_KEY_BATCH;  // planes per round trip (independent loads; +-256 is 6 ranges or 9 chunks)
    float kfirst[NKB];                             // WANT2 on matrix-core planes: the first NKB keys, kept for the second-peak pass
#pragma unroll
    for (int i = 0; i < NKB; i++) kfirst[i] = 0.f;
    if (inside) {
        float best = -1.0e30f;                     // below any score, above the bias of masked lanes
        for (int c0 = 0; c0 < nchunk; c0 += NKB) {
            float k4[NKB];
#pragma unroll
            for (int i = 0; i < NKB; i++) {
                // planes in ascending order of disparity (the matrix-core kernel's planes ascend in right column)
                const int pc = min(c0 + i, nchunk - 1);
                k4[i] = gKeys[(size_t)(range_tiles ? nchunk - 1 - pc : pc) * plane + o];
            }
#pragma unroll
            for (int i = 0; i < NKB; i++) {
                const uint32_t kb = __builtin_bit_cast(uint32_t, k4[i]);
                const float v = __builtin_bit_cast(float, kb & (range_tiles ? 0xFFFFFF80u : 0xFFFFFFC0u));
                const int dc = range_tiles ? a.dmax + ((x - R) & 63) - 32 * range_tiles * (nchunk - 1 - (c0 + i)) - (int)(kb & 127u)
                                           : a.dmin + (c0 + i) * 64 + 63 - (int)(kb & 63u);
                if (c0 + i < nchunk) {
                    if (v > best) { best = v; ds = dc; alt = MIS_INVALID_DISP; }
                    else if (v == best && alt == MIS_INVALID_DISP) alt = dc;
                }
                if (WANT2 && MF && c0 == 0) kfirst[i] = k4[i];
            }
        }
        if (oIdisp) oIdisp[o] = (int16_t)ds;       // rewritten below where the exact scores pick the other candidate
        if (ds == MIS_INVALID_DISP) {
            oDisp[o] = (float)MIS_INVALID_DISP; oPeak[o] = 0.f;
            if (WANT2) oPeak2[o] = -1.f;
        }
    }
    const bool valid = inside && ds != MIS_INVALID_DISP;

    const int qb = (y0 - R) >> 2;                 // first staged quad row
    const int nq_rows = (a.H + 3) >> 2;
    const int cbase = x0 - a.dmax - R - 1;        // raster column of right-tile column 0
    // 16 bytes per thread and load; workgroups whose columns all lie inside the pitch (all but those
    // at the left and right edge of the raster) skip the per-element clamps
    const bool interior = (cbase >= 0) && (cbase + NC <= P) && (x0 - R + LC <= P);
    if (interior) {
        // a thread owns one 16-byte column group and requests it for ALL quad rows before the first one is written to
        // LDS: one round trip to memory per workgroup instead of one per quad row
        for (int c4 = tid; c4 < (LC + NC) / 4; c4 += NT) {
            const bool left = c4 < LC / 4;
            const int c = left ? 4 * c4 : 4 * c4 - LC;
            const uint32_t* src = left ? VL + (x0 - R) + c : VR + cbase + c;      // 4-byte aligned
            uint32_t* dst = left ? tL + c : tR + c;
            const int dstep = left ? LC : NC;
            uint4 v[NQ];
#pragma unroll
            for (int qi = 0; qi < NQ; qi++) __builtin_memcpy(&v[qi], src + (size_t)min(qb + qi, nq_rows - 1) * P, 16);
#pragma unroll
            for (int qi = 0; qi < NQ; qi++) *(uint4*)(dst + qi * dstep) = v[qi];
        }
    } else {
        // at the raster's left and right edge: one column per thread, clamped into the pitch, again all quad rows
        // requested before the first LDS write
        for (int c = tid; c < LC + NC; c += NT) {
            const bool left = c < LC;
            const int cc = left ? c : c - LC;
            const uint32_t* src = left ? VL + min(x0 - R + cc, P - 1) : VR + min(max(cbase + cc, 0), P - 1);
            uint32_t* dst = left ? tL + cc : tR + cc;
            const int dstep = left ? LC : NC;
            uint32_t v[NQ];
#pragma unroll
            for (int qi = 0; qi < NQ; qi++) v[qi] = src[(size_t)min(qb + qi, nq_rows - 1) * P];
#pragma unroll
            for (int qi = 0; qi < NQ; qi++) dst[qi * dstep] = v[qi];
        }
    }
    __syncthreads();
    if (!valid) return;

    const uint32_t* pl = tL + lx;                            // column x-R
    // exact window correlation with the right window centred at column uc (one candidate)
    auto exact_sum = [&](int uc) -> int {
        int S = 0;
        const uint32_t* pc = tR + (uc - R - cbase);
#pragma unroll
        for (int qi = 0; qi < NQ; qi++) {
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int row = 4 * (qb + qi) + k;
                if (row >= y - R && row <= y + R) m |= 0xFFu << (8 * k);
            }
            if (m == 0) continue;                 // wave-uniform: a wavefront is one row
            const uint32_t* plq = pl + qi * LC;
            const uint32_t* pcq = pc + qi * NC;
#pragma unroll
            for (int i = 0; i <= 2 * R; i++) S = __builtin_amdgcn_sdot4((int)(plq[i] & m), (int)pcq[i], S, false);
        }
        return S;
    };
    // (loads its own table values: the rare passes that use it must not keep registers alive across the main loop)
    auto exact_score = [&](int S, int uc) -> float {
        const float ncx = gNegc[(size_t)y * P + (x - R)], ngx = gNegg[(size_t)y * P + (x - R)];
        const float srx = gRsr[(size_t)y * P + uc], irx = gRir[(size_t)y * P + uc];
        const float tt = __builtin_fmaf(ncx, srx, (float)S);
        const float cov = __builtin_fmaf(ngx, srx, tt);
        return __builtin_fmaf(cov, irx, 0.f);                                // NaN where the window does not fit
    };
    // Exact scores of disparities dcur-1, dcur, dcur+1.
    float il = 0.f;
    auto triple = [&](int dcur, float& sm, float& s0, float& sp) {
        const int u = x - dcur;                       // window centre in the right image for the candidate
        int Sm = 0, S0 = 0, Sp = 0;                   // disparities dcur-1, dcur, dcur+1 <-> columns u+1, u, u-1
        const uint32_t* pr = tR + (u - R - 1 - cbase);           // column u-R-1
#pragma unroll
        for (int qi = 0; qi < NQ; qi++) {
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int row = 4 * (qb + qi) + k;
                if (row >= y - R && row <= y + R) m |= 0xFFu << (8 * k);
            }
            if (m == 0) continue;                     // wave-uniform: a wavefront is one row
            const uint32_t* plq = pl + qi * LC;
            const uint32_t* prq = pr + qi * NC;
            // the row's LDS address goes through a register of its own: folded into one base for all quad
            // rows, the column offsets outgrow the 8-bit fields of ds_read2 and every read costs an add
            typedef const __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
            uint32_t a32 = (uint32_t)(uintptr_t)(lds_u32_ptr)prq;
            asm volatile("" : "+v"(a32));
            const lds_u32_ptr prq3 = (lds_u32_ptr)(uintptr_t)a32;
            uint32_t rc = prq3[0], rb = prq3[1];
            if (m == 0xFFFFFFFFu) {                   // all four rows of the quad lie in the window: no masking
#pragma unroll
                for (int i = 0; i <= 2 * R; i++) {
                    const uint32_t ra = prq3[i + 2];
                    const uint32_t l = plq[i];
                    Sm = __builtin_amdgcn_sdot4((int)l, (int)ra, Sm, false);
                    S0 = __builtin_amdgcn_sdot4((int)l, (int)rb, S0, false);
                    Sp = __builtin_amdgcn_sdot4((int)l, (int)rc, Sp, false);
                    rc = rb; rb = ra;
                }
            } else {
#pragma unroll
                for (int i = 0; i <= 2 * R; i++) {
                    const uint32_t ra = prq3[i + 2];
                    const uint32_t l = plq[i] & m;
                    Sm = __builtin_amdgcn_sdot4((int)l, (int)ra, Sm, false);
                    S0 = __builtin_amdgcn_sdot4((int)l, (int)rb, S0, false);
                    Sp = __builtin_amdgcn_sdot4((int)l, (int)rc, Sp, false);
                    rc = rb; rb = ra;
                }
            }
        }
        const float nc = gNegc[(size_t)y * P + (x - R)], ng = gNegg[(size_t)y * P + (x - R)];
        il = gIL[(size_t)y * P + (x - R)];
        // the right-image tables of the three disparities: six loads issued together (clamped columns), the
        // range tests applied afterwards; loads inside the tests would be three more exposed round trips
        float srv3[3], irv3[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int col = min(max(u + 1 - k, 0), a.W - 1);
            srv3[k] = gRsr[(size_t)y * P + col];
            irv3[k] = gRir[(size_t)y * P + col];
        }
        auto score = [&](int S, int k, int col, int d) -> float {
            if (d < a.dmin || d > a.dmax || col < 0 || col >= a.W) return __builtin_nanf("");
            const float tt = __builtin_fmaf(nc, srv3[k], (float)S);
            const float cov = __builtin_fmaf(ng, srv3[k], tt);
            return __builtin_fmaf(cov, irv3[k], 0.f);                        // NaN where the window does not fit
        };
        sm = score(Sm, 0, u + 1, dcur - 1); s0 = score(S0, 1, u, dcur); sp = score(Sp, 2, u - 1, dcur + 1);
    };
    float sm, s0, sp;
    triple(ds, sm, s0, sp);
    // Two chunks delivered EQUAL truncated scores: the exact centre scores decide, the lower disparity keeps
    // a true tie.  Rare (repeated image content); only wavefronts that hold such a pixel take the second pass.
    const bool has = alt != MIS_INVALID_DISP;
    if (__any(has)) {
        float tm, t0, tp;
        triple(has ? alt : ds, tm, t0, tp);
        if (has && (t0 > s0 || (t0 == s0 && alt < ds))) {
            sm = tm; s0 = t0; sp = tp; ds = alt;
            if (oIdisp) oIdisp[o] = (int16_t)ds;
        }
    }
    if constexpr (WANT2 && MF) {
        // plane p holds right columns 96 p + j of the run that starts at disparity dmax + ((x - R) & 63): a higher
        // plane or j is a lower disparity.  The winner's plane gives its second key; a neighbouring plane whose best
        // is the winner's neighbour across the plane boundary (j = 95 before j = 0) gives its second key too, which
        // also leaves out the column two away from the winner: that one's exact score is computed below.
        constexpr int RW = 32 * range_tiles;
        const int tt = a.dmax + ((x - R) & 63) - ds, ps = tt / RW, js = tt % RW;
        // plane c was key number nchunk - 1 - c of the merge above
        if (nchunk <= NKB) {
            int cs = -1;                               // the neighbouring plane whose second key is needed, if any
#pragma unroll
            for (int i = 0; i < NKB; i++) {
                const int c = nchunk - 1 - i;
                if (c < 0) continue;
                const uint32_t kb = __builtin_bit_cast(uint32_t, kfirst[i]);
                const int jr = (int)(kb & 127u);
                if (c == ps) continue;
                if (c == ps - 1 && js == 0 && jr == RW - 1) { cs = c; xd = 2; continue; }
                if (c == ps + 1 && js == RW - 1 && jr == 0) { cs = c; xd = -2; continue; }
                const float v = __builtin_bit_cast(float, kb & 0xFFFFFF80u);
                if (v > sec) sec = v;
            }
            const float v0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, gKeys2[(size_t)ps * plane + o]) & 0xFFFFFF80u);
            const float v1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, gKeys2[(size_t)(cs >= 0 ? cs : ps) * plane + o]) & 0xFFFFFF80u);
            if (v0 > sec) sec = v0;
            if (v1 > sec) sec = v1;
        } else
        for (int c = 0; c < nchunk; c++) {
            const uint32_t kb = __builtin_bit_cast(uint32_t, gKeys[(size_t)c * plane + o]);
            const int jr = (int)(kb & 127u);
            bool second_plane = (c == ps);
            if (c == ps - 1 && js == 0 && jr == RW - 1) { second_plane = true; xd = 2; }
            if (c == ps + 1 && js == RW - 1 && jr == 0) { second_plane = true; xd = -2; }
            const uint32_t vb = second_plane ? __builtin_bit_cast(uint32_t, gKeys2[(size_t)c * plane + o]) : kb;
            const float v = __builtin_bit_cast(float, vb & 0xFFFFFF80u);
            if (v > sec) sec = v;
        }
    } else if (WANT2) {
        const int bc = (ds - a.dmin) >> 6, lw = (ds - a.dmin) & 63;
        for (int c = 0; c < nchunk; c++) {
            const uint32_t kb = __builtin_bit_cast(uint32_t, gKeys[(size_t)c * plane + o]);
            const int l1 = 63 - (int)(kb & 63u);
            bool second_plane = (c == bc);
            if (c == bc + 1 && lw == 63 && l1 == 0) { second_plane = true; xd = 2; }
            if (c == bc - 1 && lw == 0 && l1 == 63) { second_plane = true; xd = -2; }
            const uint32_t vb = second_plane ? __builtin_bit_cast(uint32_t, gKeys2[(size_t)c * plane + o]) : kb;
            const float v = __builtin_bit_cast(float, vb & 0xFFFFFFC0u);
            if (v > sec) sec = v;
        }
    }
    float delta = 0.f;
    if (sm == sm && sp == sp) {
        const float den = 2.f * (sm - 2.f * s0 + sp);
        if (den != 0.f) delta = fminf(0.5f, fmaxf(-0.5f, (sm - sp) / den));
    }
    oDisp[o] = (float)ds + delta;
    oPeak[o] = s0 * il;
    if constexpr (WANT2) {
        // exact score of the disparity two away from the winner, where a chunk boundary hid it (rare: the
        // extra pass runs only in wavefronts that hold such a pixel)
        const int d2 = ds + xd;
        const bool need = xd != 0 && d2 >= a.dmin && d2 <= a.dmax;
        if (__any(need)) {
            const int u2 = need ? x - d2 : x - ds;
            const int Sx = exact_sum(u2);
            if (need && u2 >= 0 && u2 < a.W) {
                const float sx = exact_score(Sx, u2);
                if (sx > sec) sec = sx;
            }
        }
        oPeak2[o] = (sec > -1.0e30f) ? sec * il : -1.f;
    }
}

template <int R, bool WANT2, bool MF, int BW>
static int ncc_refine_go(mis_ctx* ctx, const NccArgs& a, int nplanes, const float* keys2,
                         float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    NccWs* w = ctx->ncc;
    const int nd = a.dmax - a.dmin + 1;
    const size_t lds = NccRefineCfg<R, BW>::lds_bytes(nd);
    if (lds > 64 * 1024)
        MIS_HIP(ctx, hipFuncSetAttribute((const void*)ncc_refine_kernel<R, WANT2, MF, BW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ncc_refine_kernel<R, WANT2, MF, BW>), dim3((a.W - 2 * R + BW - 1) / BW, (a.row1 - a.row0 + 3) / 4), dim3(4 * BW),
                       lds, ctx->stream,
                       a, w->VL, w->VR, w->negc, w->negg, w->iL, w->Rsr, w->Rir, w->keys, nplanes, d_disp, d_peak, d_idisp,
                       keys2, d_peak2);
    return MIS_OK;
}

template <int R>
static int ncc_refine_r(mis_ctx* ctx, bool mf, const NccArgs& a, int nplanes, const float* keys2,
                        float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    const bool want2 = keys2 != nullptr;
    if (!mf)                                       // the vector kernels' planes: one width
        return want2 ? ncc_refine_go<R, true, false, 128>(ctx, a, nplanes, keys2, d_disp, d_peak, d_idisp, d_peak2)
                     : ncc_refine_go<R, false, false, 128>(ctx, a, nplanes, nullptr, d_disp, d_peak, d_idisp, nullptr);
    if (ncc_refine_bw<R>(a.dmax - a.dmin + 1) == 64)
        return want2 ? ncc_refine_go<R, true, true, 64>(ctx, a, nplanes, keys2, d_disp, d_peak, d_idisp, d_peak2)
                     : ncc_refine_go<R, false, true, 64>(ctx, a, nplanes, nullptr, d_disp, d_peak, d_idisp, nullptr);
    return want2 ? ncc_refine_go<R, true, true, 128>(ctx, a, nplanes, keys2, d_disp, d_peak, d_idisp, d_peak2)
                 : ncc_refine_go<R, false, true, 128>(ctx, a, nplanes, nullptr, d_disp, d_peak, d_idisp, nullptr);
}

int ncc_refine_launch(mis_ctx* ctx, int r, bool mf, const NccArgs& a, int nplanes, const float* keys2,
                      float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
#define NCC_RCASE(RR) case RR: return ncc_refine_r<RR>(ctx, mf, a, nplanes, keys2, d_disp, d_peak, d_idisp, d_peak2);
    switch (r) {
        NCC_RCASE(1) NCC_RCASE(2) NCC_RCASE(3) NCC_RCASE(4) NCC_RCASE(5) NCC_RCASE(6) NCC_RCASE(7) NCC_RCASE(8)
        NCC_RCASE(9) NCC_RCASE(10) NCC_RCASE(11) NCC_RCASE(12) NCC_RCASE(13) NCC_RCASE(14) NCC_RCASE(15)
    }
#undef NCC_RCASE
    return mis_fail(ctx, MIS_ERR_ARG, "dense NCC refinement: unsupported window %d", 2 * r + 1);
}
