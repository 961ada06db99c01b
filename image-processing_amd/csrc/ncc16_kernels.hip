// Dense windowed NCC for 16-bit rasters, gfx950: the same lane <-> disparity walk and batched
// argmax as the 8-bit path (ncc_kernels.hip), with exact 64-bit integer sums instead of the
// f16/f32 trick: a product of two 16-bit pixels needs 31 bits and a 31x31 window sum 41.
//   column sums: 64-bit integers updated with v_mad_i64_i32 (one instruction multiplies two 32-bit
//   operands and adds the 64-bit accumulator: half the instructions of a convert + v_fma_f64 pair).
//   The left operand is stored premultiplied by N = W*W, so the sums are N*S_LR directly.
//   Every accumulator carries the bit pattern of the double 1.5*2^52 as an additive constant: read
//   as a double it IS 1.5*2^52 + sum, exactly (|sum| < 2^51), so the difference of two accumulators
//   in v_add_f64 is their exact integer difference and the window sum slides along the strip in
//   binary64 without any conversion instruction.
//   num = N*S_LR - S_L*S_R exactly (binary64, integers below 2^53),
//   score = float(num) * (1/sqrt(N var_L)) * (1/sqrt(N var_R))
// Pixels are shifted by 32768 into int16 range first.  Exact like the oracle.  Definition: see mi_stereo.h (dense windowed NCC).
#include "ncc_shared.h"

#include <math.h>

struct Ncc16Ws {
    int32_t *L32 = nullptr, *R32 = nullptr;     // shifted rasters, pitch P
    int32_t *SL = nullptr, *SR = nullptr;       // window sums of the shifted pixels
    float *iL = nullptr, *iR = nullptr;         // 1/sqrt(N*sum sq - sum^2); iR NaN where the window does not fit
    int32_t* NL = nullptr;                      // N * (left pixel - 32768): wave-uniform operand of the column sums
    double* SLd = nullptr;                      // left window sums as binary64 (wave-uniform operand of the numerator)
    int32_t* hs1 = nullptr; long long* hs2 = nullptr;
    float* keys = nullptr; size_t keys_bytes = 0;
    size_t cap_px = 0;
    int pitch = 0, rows = 0, window = 0;
    uint64_t pair_serial = 0;
    int built_lo = 0, built_hi = 0;             // raster rows whose tables exist (multiples of NCC16_STATS_BAND, or H)
    hipEvent_t ev_built = nullptr;              // end of the table build on built_stream (see ncc_kernels.hip)
    hipStream_t built_stream = nullptr;
};

void mis_ncc16_release(mis_ctx* ctx)
{
    Ncc16Ws* w = ctx->ncc16;
    if (!w) return;
    void* bufs[] = { w->L32, w->R32, w->SL, w->SR, w->iL, w->iR, w->hs1, w->hs2, w->keys, w->NL, w->SLd };
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (w->ev_built) (void)hipEventDestroy(w->ev_built);
    delete w;
    ctx->ncc16 = nullptr;
}

__global__ __launch_bounds__(256) void ncc16_hsum_kernel(const uint16_t* __restrict__ img, size_t stride_px,
                                                         int W, int H, int r, int pitch, int yfirst, int32_t* __restrict__ p32,
                                                         int32_t* __restrict__ pn, int32_t* __restrict__ hs1, long long* __restrict__ hs2)
{
    __shared__ int tile[256 + 2 * 16];
    const int y = yfirst + blockIdx.y, x0 = blockIdx.x * 256;
    const uint16_t* row = img + (size_t)y * stride_px;
    for (int i = threadIdx.x; i < 256 + 2 * r; i += 256) {
        int x = x0 - r + i;
        tile[i] = (x >= 0 && x < W) ? (int)row[x] - 32768 : 0;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= pitch) return;
    int s1 = 0; long long s2 = 0;
    for (int i = 0; i <= 2 * r; i++) { int v = tile[threadIdx.x + i]; s1 += v; s2 += (long long)v * v; }
    const size_t o = (size_t)y * pitch + x;
    p32[o] = tile[threadIdx.x + r];
    if (pn) pn[o] = (2 * r + 1) * (2 * r + 1) * tile[threadIdx.x + r];       // |N p'| <= 961 * 32768 < 2^31
    hs1[o] = s1; hs2[o] = s2;
}

#define NCC16_STATS_BAND 128
template <bool LEFT>
__global__ __launch_bounds__(256) void ncc16_stats_kernel(int W, int H, int r, int pitch, int band0,
                                                          const int32_t* __restrict__ hs1, const long long* __restrict__ hs2,
                                                          int32_t* __restrict__ osum, float* __restrict__ oinv, double* __restrict__ osumd)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= pitch) return;
    const int y0 = (band0 + blockIdx.y) * NCC16_STATS_BAND, y1 = min(y0 + NCC16_STATS_BAND, H);
    const long long N = (long long)(2 * r + 1) * (2 * r + 1);
    const bool colok = (x >= r && x < W - r);
    long long s1 = 0, s2 = 0;
    bool have = false;
    for (int y = y0; y < y1; y++) {
        const size_t o = (size_t)y * pitch + x;
        int vs = 0;
        float vi = LEFT ? 0.f : (colok ? 0.f : __builtin_nanf(""));
        if (colok && y >= r && y < H - r) {
            if (!have) {
                s1 = 0; s2 = 0;
                for (int j = -r; j <= r; j++) { s1 += hs1[(size_t)(y + j) * pitch + x]; s2 += hs2[(size_t)(y + j) * pitch + x]; }
                have = true;
            } else {
                s1 += hs1[(size_t)(y + r) * pitch + x] - hs1[(size_t)(y - r - 1) * pitch + x];
                s2 += hs2[(size_t)(y + r) * pitch + x] - hs2[(size_t)(y - r - 1) * pitch + x];
            }
            const long long var = N * s2 - s1 * s1;
            vs = (int)s1;
            vi = var > 0 ? (float)(1.0 / sqrt((double)var)) : 0.f;
        }
        if (LEFT) { if (x >= r) { osum[o - r] = vs; oinv[o - r] = vi; osumd[o - r] = (double)vs; } }     // left tables shifted by r (see ncc_kernels.hip)
        else { osum[o] = vs; oinv[o] = vi; }
    }
}

template <int R, int WT>
struct Ncc16Cfg {
    static constexpr int NC = WT + 2 * R;
    static constexpr int NCP = (NC + 3) & ~3;
    static constexpr int SEG = NCP + 63;                           // staged right-image columns: lane l reads elements 63-l .. 63-l+NCP-1
    static constexpr int LENP = ((SEG + 3 + 47) / 64) * 64 + 16;
};

// One wavefront per workgroup = (strip of WT columns, band of rows, chunk of 64 disparities).
// LDS: 4 arrays (entering right row, leaving right row, S_R, 1/sigma_R) x 4 shifted copies.
// bit pattern of the double 1.5 * 2^52: an int64 accumulator that starts from it reads, as a double, 1.5 * 2^52 + sum
#define NCC16_BIAS_BITS 0x4338000000000000LL
#define NCC16_BIAS 6755399441055744.0

// WANT2: second key plane (best key among the lanes further than one lane from the column's winner),
// as in the 8-bit kernel.
template <int R, int WT, bool WANT2>
__global__ __launch_bounds__(64, 3) void ncc16_argmax_kernel(
    NccArgs a, const int32_t* __restrict__ gNL, const int32_t* __restrict__ gR,
    const double* __restrict__ gSL, const int32_t* __restrict__ gSR, const float* __restrict__ gIR,
    float* __restrict__ oKeys, float* __restrict__ oKeys2)
{
    using Cfg = Ncc16Cfg<R, WT>;
    constexpr int WN = 2 * R + 1, NC = Cfg::NC, NCP = Cfg::NCP, SEG = Cfg::SEG, LENP = Cfg::LENP;
    __shared__ __attribute__((aligned(16))) int sArr[16 * LENP];

    const int lane = threadIdx.x;
    int chunk, strip, band;
    {
        const int total = a.nstrips * a.nbands * a.nchunk;
        const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
        const int q = total >> 3, rem = total & 7;
        const int wid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
        chunk = wid % a.nchunk;
        strip = (wid / a.nchunk) % a.nstrips;
        band = wid / (a.nchunk * a.nstrips);
    }
    const int x0 = R + strip * WT;
    const int yb0 = a.row0 + band * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.row1);
    if (yb0 >= yb1) return;
    const int T = 2 * R + (yb1 - yb0);
    const int dtop = a.dmin + chunk * 64 + 63;
    const int d = dtop - 63 + lane;
    const int loff = 63 - lane;
    const int xr0 = x0 - R - dtop;
    const float bias = (d <= a.dmax) ? 0.f : -3.0e38f;
    const uint32_t tag = 63u - (uint32_t)lane;
    const int cA = loff & 3, offB = loff + R, cB = offB & 3;
    const int P = a.pitch;

    int mycol;
    {
        float qi[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            NccTree tr;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) tr.push(jj, (float)(b * 16 + jj));
            qi[b] = tr.q;
        }
        mycol = (int)ncc_reduce_quads(qi[0], qi[1], qi[2], qi[3], lane);
    }

    long long C[NC];              // N * (column sum of products) + NCC16_BIAS_BITS
#pragma unroll
    for (int k = 0; k < NC; k++) C[k] = NCC16_BIAS_BITS;

    // The next row is fetched into registers while the current row is processed and written to LDS
    // at the end of the iteration (a wavefront's LDS operations execute in order), so the global
    // memory latency is hidden behind a whole row of arithmetic.  Every load is unconditional (column
    // clamped into the raster, 32-bit lane offset on a wave-uniform row pointer) and the conditions
    // are applied when the values are published: a load inside a conditional block is a branch and
    // a wait of its own.
    constexpr int NSTG = (SEG + 63) / 64;
    int nvn[NSTG], nvo[NSTG], nvs[NSTG]; float nvi[NSTG];
    unsigned xcl[NSTG]; bool xin[NSTG];
#pragma unroll
    for (int q = 0; q < NSTG; q++) {
        const int i = lane + 64 * q;
        const unsigned xr = (unsigned)(xr0 + i);
        xin[q] = (xr < (unsigned)a.W && i < SEG);               // also false for negative columns
        xcl[q] = min(xr, (unsigned)(a.W - 1)) * 4u;            // byte offset: scalar row base + 32-bit lane offset addressing
    }
    auto fetch = [&](int ynew, int yout, bool stats) {
        const int yold = ynew - WN;
        // one buffer descriptor per table row: scalar base, 32-bit lane offset (no 64-bit address arithmetic per load)
        auto rowbuf = [&](const void* base, int row) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (size_t)row * P * 4), 0, P * 4, 0x00027000);
        };
        const __amdgpu_buffer_rsrc_t rn = rowbuf(gR, ynew), ro = rowbuf(gR, yold >= 0 ? yold : 0);
        const __amdgpu_buffer_rsrc_t rs = rowbuf(gSR, stats ? yout : 0), ri = rowbuf(gIR, stats ? yout : 0);
#pragma unroll
        for (int q = 0; q < NSTG; q++) {
            nvn[q] = __builtin_amdgcn_raw_buffer_load_b32(rn, xcl[q], 0, 0);
            nvo[q] = __builtin_amdgcn_raw_buffer_load_b32(ro, xcl[q], 0, 0);
            nvs[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, xcl[q], 0, 0);
            nvi[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, xcl[q], 0, 0));
        }
    };
    // warm: nothing leaves the window yet (the leaving row counts as 0); stats: the row carries window statistics
    auto publish = [&](bool stats, bool old_ok) {
#pragma unroll
        for (int q = 0; q < NSTG; q++) {
            const int i = lane + 64 * q;
            if (i >= SEG) continue;
            const int vn = xin[q] ? nvn[q] : 0;
            const int vo = (xin[q] && old_ok) ? -nvo[q] : 0;     // the leaving row is staged negated (32768 fits an int)
            const int vs = (xin[q] && stats) ? nvs[q] : 0;
            const float vi = stats ? (xin[q] ? nvi[q] : __builtin_nanf("")) : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int pos = i + ((4 - c) & 3);
                sArr[(0 * 4 + c) * LENP + pos] = vn;
                sArr[(1 * 4 + c) * LENP + pos] = vo;
                sArr[(2 * 4 + c) * LENP + pos] = vs;
                sArr[(3 * 4 + c) * LENP + pos] = __builtin_bit_cast(int, vi);
            }
        }
    };
    fetch(yb0 - R, yb0 - 2 * R, false);
    publish(false, false);

    // requests written above this fence are issued before the arithmetic below it (the compiler would
    // otherwise sink every load to its first use, i.e. wait for it on the spot)
#define NCC16_ISSUED() asm volatile("" ::: "memory")

    for (int t = 0; t < T; t++) {
        const int ynew = yb0 - R + t;
        const bool score = (t >= 2 * R);
        const int yout = ynew - R;
        const bool have_next = (t + 1 < T);
        fetch(have_next ? ynew + 1 : ynew, have_next ? yout + 1 : yout, t + 1 >= 2 * R);     // unconditional; dropped after the last row
        NCC16_ISSUED();

        // ---- column sums, software pipelined: the operands of 4 columns are requested while the
        // previous 4 are consumed ----
        {
            const int4* bn4 = (const int4*)(sArr + (0 * 4 + cA) * LENP + loff + ((4 - cA) & 3));
            const int4* bo4 = (const int4*)(sArr + (1 * 4 + cA) * LENP + loff + ((4 - cA) & 3));
            const int32_t* lnew = gNL + (size_t)ynew * P + (x0 - R);
            const int yold = ynew - WN;
            const int32_t* lold = gNL + (size_t)(yold >= 0 ? yold : 0) * P + (x0 - R);
            constexpr int NGA = NCP / 4;
            int4 bn[2], bo[2];
            int an[2][4], ao[2][4];
            auto loadA = [&](int g, int set) {
                bn[set] = bn4[g]; bo[set] = bo4[g];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * g + e < NC) { an[set][e] = lnew[4 * g + e]; ao[set][e] = lold[4 * g + e]; }
            };
            loadA(0, 0);
            NCC16_ISSUED();
#pragma unroll
            for (int g = 0; g < NGA; g++) {
                const int set = g & 1;
                __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): LDS and scalar loads of this group
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NGA) loadA(g + 1, set ^ 1);
                NCC16_ISSUED();
                __builtin_amdgcn_sched_barrier(0);
                const int bnv[4] = { bn[set].x, bn[set].y, bn[set].z, bn[set].w };
                const int bov[4] = { bo[set].x, bo[set].y, bo[set].z, bo[set].w };
                // the entering row of all four columns first, then the leaving row: two multiply-adds into
                // the same accumulator back to back would wait for each other
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * g + e < NC) C[4 * g + e] += (long long)an[set][e] * (long long)bnv[e];      // v_mad_i64_i32
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * g + e < NC) C[4 * g + e] += (long long)ao[set][e] * (long long)bov[e];      // bo: negated leaving row, 0 while the window fills
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        float red = 0.f, red2 = 0.f;
        if (score) {
            const int4* sr4 = (const int4*)(sArr + (2 * 4 + cB) * LENP + offB + ((4 - cB) & 3));
            const float4* ir4 = (const float4*)(sArr + (3 * 4 + cB) * LENP + offB + ((4 - cB) & 3));
            const double* slrow = gSL + (size_t)yout * P + (x0 - R);
            constexpr int NGB = WT / 4;
            int4 s4[2]; float4 i4[2]; double sl[2][4];
            auto loadB = [&](int g, int set) {
                s4[set] = sr4[g]; i4[set] = ir4[g];
#pragma unroll
                for (int e = 0; e < 4; e++) sl[set][e] = slrow[4 * g + e];
            };
            loadB(0, 0);
            NCC16_ISSUED();
            // window sum of the first output column: 2R+1 accumulators added as integers (each carries the
            // bias constant once), all but one bias removed, the last one by the binary64 subtraction
            double S;
            {
                long long si = C[0];
#pragma unroll
                for (int k = 1; k <= 2 * R; k++) si += C[k];
                si -= (long long)((unsigned long long)NCC16_BIAS_BITS * (unsigned long long)(2 * R));    // wraps like the sum did
                S = __builtin_bit_cast(double, si) - NCC16_BIAS;
            }
            float q[4] = { -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f };
            float q2[4] = { -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f };
            NccTree tr, tr2;
            float kk[WANT2 ? 16 : 1];                           // keys of the current block of 16 columns
#pragma unroll
            for (int g = 0; g < NGB; g++) {
                const int set = g & 1;
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NGB) loadB(g + 1, set ^ 1);
                NCC16_ISSUED();
                __builtin_amdgcn_sched_barrier(0);
                const int srv[4] = { s4[set].x, s4[set].y, s4[set].z, s4[set].w };
                const float irv[4] = { i4[set].x, i4[set].y, i4[set].z, i4[set].w };
                float sc[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int j = 4 * g + e;
                    // N*S_LR - S_L*S_R, exact: integers below 2^53
                    const double num = __builtin_fma(-sl[set][e], (double)srv[e], S);
                    sc[e] = __builtin_fmaf((float)num, irv[e], bias);
                    // slide: the difference of two biased accumulators read as doubles is their exact integer difference
                    if (j + 1 < WT)
                        S += __builtin_bit_cast(double, C[j + 1 + 2 * R]) - __builtin_bit_cast(double, C[j]);
                }
                if constexpr (WANT2) {
                    tr.push4_keep((4 * g) & 15, sc[0], sc[1], sc[2], sc[3], tag);
#pragma unroll
                    for (int e = 0; e < 4; e++) kk[((4 * g) & 15) + e] = sc[e];
                } else {
                    tr.push4((4 * g) & 15, sc[0], sc[1], sc[2], sc[3], tag);
                }
                if ((g & 3) == 3) {
                    q[g >> 2] = tr.q;
                    if constexpr (WANT2) {
                        ncc_second_pass16(kk, ncc_quad_full(tr.q), -3.0e38f, tr2);
                        q2[g >> 2] = tr2.q;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            red = ncc_reduce_quads(q[0], q[1], q[2], q[3], lane);
            if constexpr (WANT2) red2 = ncc_reduce_quads(q2[0], q2[1], q2[2], q2[3], lane);
        }
        publish(t + 1 >= 2 * R, t + 1 >= WN && ynew + 1 - WN >= 0);
        // the store comes after publish: vector memory operations complete in order, and publish
        // waits for the prefetched row
        if (score) {
            const int x = x0 + mycol;
            if (mycol < WT && x < a.W - R) {
                oKeys[((size_t)chunk * a.H + yout) * a.out_pitch + x] = red;
                if constexpr (WANT2) oKeys2[((size_t)chunk * a.H + yout) * a.out_pitch + x] = red2;
            }
        }
    }
#undef NCC16_ISSUED
}

// Merge of the chunk keys, exact scores of the winner and its two neighbours by direct window sums.
// One workgroup = NCC16_REF_BW consecutive pixels of 4 consecutive rows.  Neighbouring pixels need
// not agree on their disparity, so instead of per-pixel gathers the workgroup walks the window rows
// in groups of NCC16_REF_G: each group's left columns and right columns for the WHOLE search range
// are staged in LDS (int32), every pixel adds the rows of the group that lie in its window.
#define NCC16_REF_BW 128
#define NCC16_REF_G 8
template <int R>
struct Ncc16RefineCfg {
    static constexpr int LC = (NCC16_REF_BW + 2 * R + 3) & ~3;      // row strides are multiples of 4: 16-byte LDS writes
    static int right_cols(int nd) { return (NCC16_REF_BW + (nd - 1) + 2 * R + 3 + 3) & ~3; }
    static size_t lds_bytes(int nd) { return (size_t)NCC16_REF_G * (LC + right_cols(nd)) * sizeof(int32_t); }
};

// WANT2: second-peak output from the two key planes, with the exact score of the disparity two away
// from the winner where a chunk boundary hid it (see ncc_refine_kernel in ncc_refine_kernels.hip); that
// candidate's window sum rides along with the other three.
// (six wavefronts per SIMD = three workgroups per CU: worth the few registers it spills, 2.84 -> 2.65 ms on a 16384 x 1024 band)
template <int R, bool WANT2>
__global__ __launch_bounds__(512, 6) void ncc16_refine_kernel(
    NccArgs a, const int32_t* __restrict__ gL, const int32_t* __restrict__ gR,
    const int32_t* __restrict__ gSL, const float* __restrict__ gIL, const int32_t* __restrict__ gSR,
    const float* __restrict__ gIR, const float* __restrict__ gKeys, int nchunk,
    float* __restrict__ oDisp, float* __restrict__ oPeak, int16_t* __restrict__ oIdisp,
    const float* __restrict__ gKeys2, float* __restrict__ oPeak2)
{
    constexpr int LC = Ncc16RefineCfg<R>::LC, G = NCC16_REF_G;
    extern __shared__ __attribute__((aligned(16))) int32_t ncc16_refine_lds[];
    const int NC = (NCC16_REF_BW + (a.dmax - a.dmin) + 2 * R + 3 + 3) & ~3;      // == Ncc16RefineCfg<R>::right_cols(nd)
    int32_t* tL = ncc16_refine_lds;               // [G][LC]
    int32_t* tR = ncc16_refine_lds + G * LC;      // [G][NC]
    const int tid = threadIdx.x;
    const int lx = tid & (NCC16_REF_BW - 1), ly = __builtin_amdgcn_readfirstlane(tid / NCC16_REF_BW);   // a wavefront is 64 pixels of one row
    const int x0 = R + blockIdx.x * NCC16_REF_BW;
    const int y0 = a.row0 + 4 * blockIdx.y;
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = (x < a.W - R) && (y < a.row1);
    const int P = a.pitch;
    const size_t o = (size_t)y * a.out_pitch + x;
    int ds = MIS_INVALID_DISP;
    if (inside) {
        float best = -1.0e30f;
        const size_t plane = (size_t)a.H * a.out_pitch;
        for (int c0 = 0; c0 < nchunk; c0 += 6) {   // six planes per round trip: the loads are independent
            float k6[6];
#pragma unroll
            for (int i = 0; i < 6; i++) k6[i] = gKeys[(size_t)min(c0 + i, nchunk - 1) * plane + o];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const uint32_t kb = __builtin_bit_cast(uint32_t, k6[i]);
                const float v = __builtin_bit_cast(float, kb & 0xFFFFFFC0u);
                if (c0 + i < nchunk && v > best) { best = v; ds = a.dmin + (c0 + i) * 64 + 63 - (int)(kb & 63u); }
            }
        }
        if (oIdisp) oIdisp[o] = (int16_t)ds;
        if (ds == MIS_INVALID_DISP) { oDisp[o] = (float)MIS_INVALID_DISP; oPeak[o] = 0.f; if (WANT2) oPeak2[o] = -1.f; }
    }
    const bool valid = inside && ds != MIS_INVALID_DISP;
    float sec = -1.0e30f;                         // WANT2: best second-peak candidate (argmax-kernel scale)
    int xd = 0;                                   // WANT2: +2 / -2 when the disparity two away needs its exact score
    if (WANT2 && valid) {
        const size_t plane = (size_t)a.H * a.out_pitch;
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
    const int u = valid ? x - ds : x0 - a.dmax;   // window centre in the right image (any in-range value when idle)
    const int cbase = x0 - a.dmax - R - 1;        // raster column of right-tile column 0
    long long Sm = 0, S0 = 0, Sp = 0;             // disparities ds-1, ds, ds+1 <-> right columns u+1, u, u-1 (exact 64-bit sums)
    long long Sx = 0;                             // WANT2: disparity ds + xd <-> right column u - xd
    const int d2 = ds + xd;
    const bool need = WANT2 && valid && xd != 0 && d2 >= a.dmin && d2 <= a.dmax;
    const int u2 = need ? x - d2 : u;
    const int32_t* pl = tL + lx;                              // column x-R
    const int32_t* pr = tR + (u - R - 1 - cbase);             // column u-R-1
    const int32_t* px = tR + (u2 - R - cbase);                // column u2-R

    const int ylo = y0 - R, yhi = min(y0 + 3, a.row1 - 1) + R;   // window rows of the 4 output rows
    // workgroups whose columns all lie inside the pitch (all but those at the left and right edge of the raster) stage
    // 16 bytes per thread and load, without per-element clamps
    const bool interior = (cbase >= 0) && (cbase + NC <= P) && (x0 - R + LC <= P);
    for (int yg = ylo; yg <= yhi; yg += G) {
        __syncthreads();                           // the previous group has been consumed
        if (interior) {
            // a thread owns one 16-byte column group and requests it for ALL rows of the group before the first one is
            // written to LDS: one round trip to memory per group instead of two per row
            for (int c4 = tid; c4 < (LC + NC) / 4; c4 += 512) {
                const bool left = c4 < LC / 4;
                const int c = left ? 4 * c4 : 4 * c4 - LC;
                const int32_t* src = left ? gL + (x0 - R) + c : gR + cbase + c;       // 4-byte aligned
                int32_t* dst = left ? tL + c : tR + c;
                const int dstep = left ? LC : NC;
                uint4 v[G];
#pragma unroll
                for (int j = 0; j < G; j++) __builtin_memcpy(&v[j], src + (size_t)min(yg + j, a.H - 1) * P, 16);
#pragma unroll
                for (int j = 0; j < G; j++) *(uint4*)(dst + j * dstep) = v[j];
            }
        } else {
            // at the raster's left and right edge: one column per thread, clamped into the pitch, again all rows of the
            // group requested before the first LDS write
            for (int c = tid; c < LC + NC; c += 512) {
                const bool left = c < LC;
                const int cc = left ? c : c - LC;
                const int32_t* src = left ? gL + min(x0 - R + cc, P - 1) : gR + min(max(cbase + cc, 0), P - 1);
                int32_t* dst = left ? tL + cc : tR + cc;
                const int dstep = left ? LC : NC;
                int32_t v[G];
#pragma unroll
                for (int j = 0; j < G; j++) v[j] = src[(size_t)min(yg + j, a.H - 1) * P];
#pragma unroll
                for (int j = 0; j < G; j++) dst[j * dstep] = v[j];
            }
        }
        __syncthreads();
        if (!valid) continue;
#pragma unroll 1
        for (int j = 0; j < G; j++) {
            const int row = yg + j;
            if (row < y - R || row > y + R) continue;          // wave-uniform: a wavefront is one row
            const int32_t* plj = pl + j * LC;
            const int32_t* prj = pr + j * NC;
            int rc = prj[0], rb = prj[1];
            const int32_t* pxj = px + j * NC;
#pragma unroll
            for (int i = 0; i <= 2 * R; i++) {
                const int ra = prj[i + 2];
                const int l = plj[i];
                Sm += (long long)l * (long long)ra;          // v_mad_i64_i32: multiply and 64-bit accumulate in one instruction
                S0 += (long long)l * (long long)rb;
                Sp += (long long)l * (long long)rc;
                if constexpr (WANT2) Sx += (long long)l * (long long)pxj[i];
                rc = rb; rb = ra;
            }
        }
    }
    if (!valid) return;

    constexpr long long NN = (long long)((2 * R + 1) * (2 * R + 1));
    const long long sl = (long long)gSL[(size_t)y * P + (x - R)];
    const float il = gIL[(size_t)y * P + (x - R)];
    // table values of the three candidates: loaded together (clamped columns), range tests afterwards
    int srv[3]; float irv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int col = min(max(u + 1 - k, 0), a.W - 1);
        srv[k] = gSR[(size_t)y * P + col];
        irv[k] = gIR[(size_t)y * P + col];
    }
    auto score = [&](long long S, int k, int col, int d) -> float {
        if (d < a.dmin || d > a.dmax || col < 0 || col >= a.W) return __builtin_nanf("");
        const long long num = NN * S - sl * (long long)srv[k];          // exact, below 2^53 in magnitude
        return __builtin_fmaf((float)(double)num, irv[k], 0.f);         // rounded once to binary32, like the argmax kernel's conversion
    };
    const float sm = score(Sm, 0, u + 1, ds - 1), s0 = score(S0, 1, u, ds), sp = score(Sp, 2, u - 1, ds + 1);
    float delta = 0.f;
    if (sm == sm && sp == sp) {
        const float den = 2.f * (sm - 2.f * s0 + sp);
        if (den != 0.f) delta = fminf(0.5f, fmaxf(-0.5f, (sm - sp) / den));
    }
    oDisp[o] = (float)ds + delta;
    oPeak[o] = s0 * il;
    if constexpr (WANT2) {
        if (need && u2 >= 0 && u2 < a.W) {
            const long long num = NN * Sx - sl * (long long)gSR[(size_t)y * P + u2];
            const float sx = __builtin_fmaf((float)(double)num, gIR[(size_t)y * P + u2], 0.f);     // NaN where the window does not fit
            if (sx > sec) sec = sx;
        }
        oPeak2[o] = (sec > -1.0e30f) ? sec * il : -1.f;
    }
}

// tables for raster rows [need_lo, need_hi) (the rows the requested output rows read); rows built by
// earlier calls on the same pair are kept (see ncc_prepare in ncc_kernels.hip)
static int ncc16_prepare(mis_ctx* ctx, int window, int need_lo, int need_hi)
{
    Ncc16Ws* w = ctx->ncc16;
    if (!w) { w = new Ncc16Ws(); ctx->ncc16 = w; }
    const int W = ctx->width, H = ctx->height, r = window / 2;
    const int pitch = ((W + 64 + 63) / 64) * 64 + 64;
    const size_t npx = (size_t)pitch * H;
    if (w->cap_px < npx) {
        void** b4[] = { (void**)&w->L32, (void**)&w->R32, (void**)&w->SL, (void**)&w->SR, (void**)&w->iL, (void**)&w->iR,
                        (void**)&w->hs1, (void**)&w->NL };
        for (void** b : b4) { if (*b) { MIS_HIP(ctx, hipFree(*b)); *b = nullptr; } }
        void** b8[] = { (void**)&w->hs2, (void**)&w->SLd };
        for (void** b : b8) { if (*b) { MIS_HIP(ctx, hipFree(*b)); *b = nullptr; } }
        w->cap_px = 0;
        for (void** b : b4) MIS_HIP(ctx, hipMalloc(b, npx * 4));
        for (void** b : b8) MIS_HIP(ctx, hipMalloc(b, npx * 8));
        w->cap_px = npx;
        w->pair_serial = 0;
    }
    need_lo = need_lo < 0 ? 0 : need_lo / NCC16_STATS_BAND * NCC16_STATS_BAND;
    need_hi = (need_hi + NCC16_STATS_BAND - 1) / NCC16_STATS_BAND * NCC16_STATS_BAND;
    if (need_hi > H) need_hi = H;
    const bool same = (w->pair_serial == ctx->pair_serial && w->window == window && w->pitch == pitch && w->rows == H);
    int lo[2], hi[2], nparts = 0;
    if (same && need_lo >= w->built_lo && need_hi <= w->built_hi) {
        if (w->built_stream != ctx->stream && w->ev_built) MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, w->ev_built, 0));
        return MIS_OK;
    }
    if (same && need_lo <= w->built_hi && need_hi >= w->built_lo) {
        if (need_lo < w->built_lo) { lo[nparts] = need_lo; hi[nparts] = w->built_lo; nparts++; }
        if (need_hi > w->built_hi) { lo[nparts] = w->built_hi; hi[nparts] = need_hi; nparts++; }
        if (w->built_stream != ctx->stream && w->ev_built) MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, w->ev_built, 0));
        w->built_lo = need_lo < w->built_lo ? need_lo : w->built_lo;
        w->built_hi = need_hi > w->built_hi ? need_hi : w->built_hi;
    } else {
        lo[0] = need_lo; hi[0] = need_hi; nparts = 1;
        w->built_lo = need_lo; w->built_hi = need_hi;
    }
    w->pitch = pitch; w->rows = H;
    w->pair_serial = 0;               // the cache claims nothing until every launch below has been accepted
    hipStream_t s = ctx->stream;
    {   // a pair still on its way from page-locked host rasters: the rows the table kernels read must have arrived
        int rcu = mis_wait_upload(ctx, need_hi + r < H ? need_hi + r : H, s);
        if (rcu) return rcu;
    }
    const size_t stride_px = ctx->stride / 2;
    const dim3 block(256);
    for (int k = 0; k < nparts; k++) {
        // the statistics of a band need the horizontal sums of r rows above and below it (hs1 / hs2 are
        // scratch shared by the two images, rewritten per image)
        const int h0 = lo[k] - r > 0 ? lo[k] - r : 0, h1 = hi[k] + r < H ? hi[k] + r : H;
        const dim3 grid((pitch + 255) / 256, h1 - h0);
        const dim3 sgrid((pitch + 255) / 256, (hi[k] - lo[k] + NCC16_STATS_BAND - 1) / NCC16_STATS_BAND);
        const int band0 = lo[k] / NCC16_STATS_BAND;
        hipLaunchKernelGGL(ncc16_hsum_kernel, grid, block, 0, s, (const uint16_t*)ctx->img[0], stride_px, W, H, r, pitch, h0, w->L32, w->NL, w->hs1, w->hs2);
        hipLaunchKernelGGL(ncc16_stats_kernel<true>, sgrid, block, 0, s, W, H, r, pitch, band0, w->hs1, w->hs2, w->SL, w->iL, w->SLd);
        hipLaunchKernelGGL(ncc16_hsum_kernel, grid, block, 0, s, (const uint16_t*)ctx->img[1], stride_px, W, H, r, pitch, h0, w->R32, (int32_t*)nullptr, w->hs1, w->hs2);
        hipLaunchKernelGGL(ncc16_stats_kernel<false>, sgrid, block, 0, s, W, H, r, pitch, band0, w->hs1, w->hs2, w->SR, w->iR, (double*)nullptr);
    }
    MIS_HIP(ctx, hipGetLastError());
    if (!w->ev_built) MIS_HIP(ctx, hipEventCreateWithFlags(&w->ev_built, hipEventDisableTiming));
    MIS_HIP(ctx, hipEventRecord(w->ev_built, ctx->stream));
    w->built_stream = ctx->stream;
    w->pair_serial = ctx->pair_serial;
    w->window = window;
    return MIS_OK;
}

template <int R, bool WANT2>
static int ncc16_launch(mis_ctx* ctx, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    constexpr int WT = 32;
    Ncc16Ws* w = ctx->ncc16;
    const size_t plane_px = (size_t)a.H * a.W;
    int rc = mis_reserve(ctx, (void**)&w->keys, &w->keys_bytes, (size_t)a.nchunk * plane_px * sizeof(float) * (WANT2 ? 2 : 1));
    if (rc) return rc;
    float* keys2 = WANT2 ? w->keys + (size_t)a.nchunk * plane_px : nullptr;
    a.nstrips = (a.W - 2 * R + WT - 1) / WT;
    a.nbands = (a.row1 - a.row0 + a.band_rows - 1) / a.band_rows;
    hipLaunchKernelGGL((ncc16_argmax_kernel<R, WT, WANT2>), dim3((unsigned)(a.nstrips * a.nbands * a.nchunk)), dim3(64), 0, ctx->stream,
                       a, w->NL, w->R32, w->SLd, w->SR, w->iR, w->keys, keys2);
    mis_prof_mark(ctx);
    {
        const size_t lds = Ncc16RefineCfg<R>::lds_bytes(a.dmax - a.dmin + 1);
        if (lds > 64 * 1024)
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)ncc16_refine_kernel<R, WANT2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((ncc16_refine_kernel<R, WANT2>), dim3((a.W - 2 * R + NCC16_REF_BW - 1) / NCC16_REF_BW, (a.row1 - a.row0 + 3) / 4),
                           dim3(512), lds, ctx->stream,
                           a, w->L32, w->R32, w->SL, w->iL, w->SR, w->iR, w->keys, a.nchunk, d_disp, d_peak, d_idisp, keys2, d_peak2);
    }
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

// called by the dense-NCC entry points for 16-bit pairs; a: geometry already filled in
int mis_ncc16_run(mis_ctx* ctx, int window, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    const int r = window / 2;
    int rc = ncc16_prepare(ctx, window, a.row0 - r, a.row1 + r);      // a.row0 / a.row1: output rows, clipped to [r, H-r)
    if (rc) return rc;
    a.pitch = ctx->ncc16->pitch;
    mis_prof_mark(ctx);
#define NCC16_CASE(RR) case RR: rc = d_peak2 ? ncc16_launch<RR, true>(ctx, a, d_disp, d_peak, d_idisp, d_peak2) \
                                 : ncc16_launch<RR, false>(ctx, a, d_disp, d_peak, d_idisp, nullptr); break;
    switch (window / 2) {
        NCC16_CASE(1) NCC16_CASE(2) NCC16_CASE(3) NCC16_CASE(4) NCC16_CASE(5) NCC16_CASE(6) NCC16_CASE(7) NCC16_CASE(8)
        NCC16_CASE(9) NCC16_CASE(10) NCC16_CASE(11) NCC16_CASE(12) NCC16_CASE(13) NCC16_CASE(14) NCC16_CASE(15)
        default: rc = mis_fail(ctx, MIS_ERR_ARG, "dense NCC: unsupported window %d", window);
    }
#undef NCC16_CASE
    mis_prof_mark(ctx);
    return rc;
}
