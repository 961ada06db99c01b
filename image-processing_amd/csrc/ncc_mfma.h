// Dense NCC argmax on the matrix cores (8-bit pairs), gfx950.
//
// The window correlation of a left pixel x with the right pixels x' = x - d,
//     S(x, x') = sum over window rows j and taps i of L'(x+i, y+j) * R'(x'+i, y+j),
// is a product of two Hankel matrices: A[x'][k] = R'(x'+tap(k), row(k)), B[k][x] = L'(x+tap(k), row(k)).
// Walking down the rows, S(y) = S(y-1) + (row y+r) - (row y-r-1), and both rows ride one product with the
// accumulator as C operand: k runs over (tap, {entering row, leaving row}) pairs, which is exactly the
// layout of the packed row pairs the tables hold (half2(p(y), -+p(y-W)) per pixel; the left image carries
// the minus sign).  So one v_mfma_f32_32x32x16_f16 per 8 taps slides 1024 window sums one row down; all
// values are integers below 2^24, exact in f32.
//
//   wavefront   <-> 64 left columns (two 32-column blocks) x one range of 32*T right columns x a band of rows
//   accumulators<-> T x 2 tiles of 32 x 32 window sums; column x on the lane, right column x' in the registers
//   cov         =   S - S_L S_R / N as a second product on the accumulators (v_mfma_f32_32x32x2_f32, an exact
//                   fmaf chain: k = 0 carries -round(S_L/N), k = 1 the remainder, as in the vector kernel)
//   score       =   cov / sigma_R (one multiply; 1/sigma_L is common to a column and does not move the argmax)
//   argmax      :   a lane owns its column, so the maximum over x' is a running v_max3 over registers; keys
//                   carry 26 bits of the score and the register id below, the two half-waves (which hold
//                   interleaved groups of 4 rows) meet once per row through one v_permlane32_swap
// The operands come from a wave-private LDS image of the current packed rows in 4 copies shifted by one
// pixel each (ds_read_b128 of 4 consecutive pixels = 8 consecutive k needs 16-byte alignment).
//
// Output: one key plane per range of right columns: upper 25 bits of the best score, below them the
// right column's index j in the range (d = dmax + ((x - r) & 63) - 32*T*range - j).
#pragma once

#include "ncc_shared.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define NCC_MF_T 3           // 32-column tiles of right columns per wavefront
#define NCC_MF_NB 2          // 32-column blocks of left columns per wavefront
#define NCC_MF_LOW (-3.0e38f)

template <int R>
struct NccMfCfg {
    static constexpr int KS = (2 * R + 1 + 7) / 8;              // products of 8 taps (16 k) per row step
    static constexpr int SEGR = 32 * NCC_MF_T + 8 * KS;         // staged right pixels
    static constexpr int SEGL = 32 * NCC_MF_NB + 8 * KS;        // staged left pixels
    static constexpr int QR = SEGR / 4, QL = SEGL / 4;          // 16-byte positions per copy
    static constexpr int SEGM = SEGR > SEGL ? SEGR : SEGL;
    static constexpr int CS = ((SEGM + 47) / 64) * 64 + 16;     // copy stride in words: >= SEGM, == 16 (mod 64): the four copies sit on disjoint bank quarters
    static int ranges(int nd) { return (((nd + 30) >> 5) + NCC_MF_NB + NCC_MF_T - 1) / NCC_MF_T; }
};

// v_max_f32 semantics (a quiet NaN loses) without the canonicalisation the compiler puts in front of fmaxf
__device__ __forceinline__ float ncc_max3(float a, float b, float c)
{
    float o;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}

template <int R>
__global__ __launch_bounds__(64, 2) void ncc_mfma_argmax_kernel(
    NccArgs a, int nblocks, int nranges,
    const uint32_t* __restrict__ gLpack, const float* __restrict__ gNegc, const float* __restrict__ gNegg,
    const uint32_t* __restrict__ gRpack, const float* __restrict__ gRsr, const float* __restrict__ gRir,
    float* __restrict__ oKeys)
{
    using Cfg = NccMfCfg<R>;
    constexpr int KS = Cfg::KS, T = NCC_MF_T, NB = NCC_MF_NB, CS = Cfg::CS, QR = Cfg::QR, QL = Cfg::QL;
    constexpr int WN = 2 * R + 1;
    static_assert(NB == 2, "the half-wave merge pairs two column blocks");
    static_assert(16 * T <= 64, "register tags are inline constants");
    __shared__ __attribute__((aligned(16))) float sR[4 * CS];
    __shared__ __attribute__((aligned(16))) float sL[4 * CS];
    __shared__ __attribute__((aligned(16))) float sSr[32 * T];
    __shared__ __attribute__((aligned(16))) float sIr[32 * T];

    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    // Workgroups are dealt to the 8 XCDs round-robin; remap so that every XCD works through a contiguous
    // run of work items, range-fastest: its wavefronts then share left rows and overlapping right columns
    // in the L2 (speed only, nothing depends on it).
    int range, block, band;
    {
        const int total = nblocks * a.nbands * nranges;
        const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
        const int q = total >> 3, rem = total & 7;
        const int wid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
        range = wid % nranges;
        block = (wid / nranges) % nblocks;
        band = wid / (nranges * nblocks);
    }
    const int X0 = R + 64 * block;
    const int yb0 = a.row0 + band * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.row1);
    if (yb0 >= yb1) return;
    const int nsteps = 2 * R + (yb1 - yb0);
    const int nd = a.dmax - a.dmin + 1;
    const int G = (nd + 30) >> 5;                   // diagonals g = 0..G of 32x32 tiles touch the disparity range
    const int XR0 = X0 - a.dmax + 32 * T * range;   // raster column of the range's first right pixel
    const int P = a.pitch;

    // tile (t, c): right columns XR0 + 32t + m, left columns X0 + 32c + n; d = dmax - 32g + (n - m), g = T*range + t - c
    auto tile_g = [&](int t, int c) { return T * range + t - c; };
    auto tile_any = [&](int t, int c) { const int g = tile_g(t, c); return g >= 0 && g <= G; };
    auto tile_full = [&](int t, int c) { const int g = tile_g(t, c); return g >= 1 && 32 * g + 31 <= nd - 1; };

    // ---- staging: global -> registers (a row ahead) -> LDS ----
    // position s of the 16-byte grid: copy s / Q, slot s % Q; copy c, slot q holds pixels 4q+c .. 4q+c+3
    int colR[2], dstR[2], colL[2], dstL[2];
    bool okR[2], okL[2];
#pragma unroll
    for (int r2 = 0; r2 < 2; r2++) {
        const int s = lane + 64 * r2;
        okR[r2] = s < 4 * QR;
        okL[r2] = s < 4 * QL;
        const int cR = okR[r2] ? s / QR : 0, qR = okR[r2] ? s % QR : 0;
        const int cL = okL[r2] ? s / QL : 0, qL = okL[r2] ? s % QL : 0;
        colR[r2] = XR0 - R + 4 * qR + cR;      // tap 0 of the window centred at XR0 is column XR0 - R
        dstR[r2] = cR * CS + 4 * qR;
        colL[r2] = X0 - R + 4 * qL + cL; dstL[r2] = cL * CS + 4 * qL;
    }
    const bool okT = lane < 8 * T;
    const int colT = XR0 + 4 * (okT ? lane : 0);
    // clamped start columns: the loads are unconditional and inside the row; vectors that straddle the
    // row's ends are put right at publish time (only wavefronts at the raster's left and right edge)
    auto clamp4 = [&](int col) { return min(max(col, 0), P - 4); };
    const int ccR0 = clamp4(colR[0]), ccR1 = clamp4(colR[1]), ccT = clamp4(colT);
    // element k of the vector wanted at column col, from the vector loaded at column cc; fill outside [0, lim)
    auto fix4 = [&](u32x4_t v, int col, int cc, int lim, uint32_t fill) -> u32x4_t {
        u32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = col + k, i = x - cc;
            const uint32_t e = i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
            o[k] = (x < 0 || x >= lim) ? fill : e;
        }
        return o;
    };
    const int nlcol = 64 * block + 32 * 0 + n;      // left tables are stored shifted by R columns: entry x-R describes column x
    u32x4_t fr[2], fl[2], fsr, fir;
    float fnl[NB];
    auto fetch = [&](int ynew, int yout, bool stats) {
        const uint32_t* rp = gRpack + (size_t)ynew * P;
        const uint32_t* lp = gLpack + (size_t)ynew * P;
        const float* sp = gRsr + (size_t)(stats ? yout : 0) * P;
        const float* ip = gRir + (size_t)(stats ? yout : 0) * P;
        const float* np = (h ? gNegg : gNegc) + (size_t)(stats ? yout : 0) * P;
        __builtin_memcpy(&fr[0], rp + ccR0, 16);
        __builtin_memcpy(&fr[1], rp + ccR1, 16);
        __builtin_memcpy(&fl[0], lp + colL[0], 16);
        __builtin_memcpy(&fl[1], lp + colL[1], 16);
        __builtin_memcpy(&fsr, sp + ccT, 16);
        __builtin_memcpy(&fir, ip + ccT, 16);
#pragma unroll
        for (int c = 0; c < NB; c++) fnl[c] = np[nlcol + 32 * c];
    };
    const uint32_t qnan = 0x7FC00000u;
    auto publish = [&](bool stats, bool warm) {
        if (warm) {                                  // nothing leaves the window yet: drop the leaving-row halves of the left row
#pragma unroll
            for (int e = 0; e < 4; e++) { fl[0][e] &= 0x0000FFFFu; fl[1][e] &= 0x0000FFFFu; }
        }
        if (ccR0 != colR[0]) fr[0] = fix4(fr[0], colR[0], ccR0, P, 0u);
        if (ccR1 != colR[1]) fr[1] = fix4(fr[1], colR[1], ccR1, P, 0u);
        if (okR[0]) *(u32x4_t*)(sR + dstR[0]) = fr[0];
        if (okR[1]) *(u32x4_t*)(sR + dstR[1]) = fr[1];
        if (okL[0]) *(u32x4_t*)(sL + dstL[0]) = fl[0];
        if (okL[1]) *(u32x4_t*)(sL + dstL[1]) = fl[1];
        if (stats) {
            if (ccT != colT) { fsr = fix4(fsr, colT, ccT, P, 0u); fir = fix4(fir, colT, ccT, P, qnan); }
            if (okT) { *(u32x4_t*)(sSr + 4 * lane) = fsr; *(u32x4_t*)(sIr + 4 * lane) = fir; }
        }
    };

    // ---- per-lane read positions ----
    const int c4 = n & 3;
    const int oLane = c4 * CS + (n - c4) + 4 * h;   // pixel n + 4h of copy c4, 16-byte aligned
    const uint32_t keep_h0 = h ? 0u : 0xFFFFFFFFu;  // taps past the window in the upper half-wave's share of a product

    f32x16_t acc[T][NB];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int c = 0; c < NB; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][c][i] = 0.f;

    fetch(yb0 - R, yb0 - 2 * R, false);
    publish(false, true);
    float nl[NB] = { 0.f, 0.f };

    const uint32_t kmask = 0xFFFFFFC0u;
    for (int st = 0; st < nsteps; st++) {
        const int ynew = yb0 - R + st;
        const int yout = ynew - R;
        const bool score = (st >= 2 * R);
        const bool have_next = (st + 1 < nsteps);
        if (have_next) fetch(ynew + 1, yout + 1, st + 1 >= 2 * R);

        // left operands of this row: B[k][x], 8 k per lane and product
        u32x4_t bop[NB][KS];
#pragma unroll
        for (int c = 0; c < NB; c++)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                u32x4_t v = *(const u32x4_t*)(sL + oLane + 32 * c + 8 * kk);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int tap0 = 8 * kk + e, tap1 = tap0 + 4;        // this element's tap in the lower / upper half-wave
                    if (tap0 > 2 * R) v[e] = 0u;
                    else if (tap1 > 2 * R) v[e] &= keep_h0;
                }
                bop[c][kk] = v;
            }

        float best[NB] = { NCC_MF_LOW, NCC_MF_LOW };
#pragma unroll
        for (int t = 0; t < T; t++) {
            if (!tile_any(t, 0) && !tile_any(t, 1)) continue;
            u32x4_t aop[KS];
#pragma unroll
            for (int kk = 0; kk < KS; kk++) aop[kk] = *(const u32x4_t*)(sR + oLane + 32 * t + 8 * kk);
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (!tile_any(t, c)) continue;
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, aop[kk]),
                                                                      __builtin_bit_cast(half8_t, bop[c][kk]), acc[t][c], 0, 0, 0);
            }
            if (!score) continue;
            const float sra = sSr[32 * t + n];
            f32x4_t irv[4];
#pragma unroll
            for (int b = 0; b < 4; b++) irv[b] = *(const f32x4_t*)(sIr + 32 * t + 8 * b + 4 * h);
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (!tile_any(t, c)) continue;
                const f32x16_t cov = __builtin_amdgcn_mfma_f32_32x32x2f32(sra, nl[c], acc[t][c], 0, 0, 0);
                float key[16];
                if (tile_full(t, c)) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float sc = cov[i] * irv[i >> 2][i & 3];
                        key[i] = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, sc) & kmask) | (uint32_t)(16 * t + i));
                    }
                } else {
                    // a tile on the edge of the disparity range: 0 <= 32g + (m - n) <= nd - 1 decides per element
                    const int vb = 32 * tile_g(t, c) + 4 * h - n;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float sc = cov[i] * irv[i >> 2][i & 3];
                        const float k = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, sc) & kmask) | (uint32_t)(16 * t + i));
                        const bool in = (unsigned)(vb + 8 * (i >> 2) + (i & 3)) <= (unsigned)(nd - 1);
                        key[i] = in ? k : NCC_MF_LOW;
                    }
                }
                float b0 = best[c];
#pragma unroll
                for (int i = 0; i < 16; i += 2) b0 = ncc_max3(b0, key[i], key[i + 1]);
                best[c] = b0;
            }
        }

        // single-buffered LDS image: every read of this row was issued above and LDS executes a wavefront's
        // operations in order, so the next row may be written now
        if (have_next) publish(st + 1 >= 2 * R, st + 1 < WN);
#pragma unroll
        for (int c = 0; c < NB; c++) nl[c] = fnl[c];

        if (score) {
            // the two half-waves hold different right columns of the same left column: bring them together
            // (lanes 0-31 then own block 0, lanes 32-63 block 1), compare on the 26-bit scores, the larger
            // right column (lower disparity) keeps a tie
            float k0 = best[0], k1 = best[1];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(k0), "+v"(k1));   // k0: lower half-wave's keys, k1: upper's
            auto jof = [&](float k, int hh) {
                const uint32_t tg = __builtin_bit_cast(uint32_t, k) & 63u;
                return (int)(32u * (tg >> 4) + 8u * ((tg >> 2) & 3u) + (tg & 3u)) + 4 * hh;
            };
            const int j0 = jof(k0, 0), j1 = jof(k1, 1);
            const float s0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, k0) & kmask);
            const float s1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, k1) & kmask);
            const bool up = (s1 > s0) || (s1 == s0 && j1 > j0);
            const uint32_t sb = __builtin_bit_cast(uint32_t, up ? s1 : s0);
            const uint32_t out = (sb & 0xFFFFFF80u) | (uint32_t)(up ? j1 : j0);
            const int x = X0 + lane;
            if (x < a.W - R) oKeys[((size_t)range * a.H + yout) * a.out_pitch + x] = __builtin_bit_cast(float, out);
        }
    }
}
