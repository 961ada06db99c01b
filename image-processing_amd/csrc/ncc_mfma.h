// Dense NCC argmax on the matrix cores (8-bit pairs, windows up to 21 x 21), gfx950.
//
// The window correlation of a left pixel x with the right pixels x' = x - d,
//     S(x, x') = sum over window rows j and taps i of L'(x+i, y+j) * R'(x'+i, y+j),
// is a product of two Hankel matrices: A[x'][k] = R'(x'+tap(k), row(k)), B[k][x] = L'(x+tap(k), row(k)).
// Walking down the rows, S(y) = S(y-1) + (row y+r) - (row y-r-1), and both rows ride one product with the
// accumulator as C operand: k runs over (tap, {entering row, leaving row}) pairs, which is exactly the
// layout of the packed row pairs the tables hold (half2(p(y), -+p(y-W)) per pixel; the left image carries
// the minus sign).  So one v_mfma_f32_32x32x16_f16 per 8 taps slides 1024 window sums one row down.
//
// The integer part of the mean correction rides the same products.  The score needs
//     cov = S - S_L S_R / N = (S + c S_R) + g S_R,   c = -round(S_L / N) (integer), g = -(S_L - N round(S_L/N)) / N,
// and T = S + c S_R slides too: T(y) - T(y-1) = [S(y) - S(y-1)] + c(y) S_R(y) - c(y-1) S_R(y-1).  With
// S_R = 256 s1 + s0 (s1 = floor(S_R / 256), 0 <= s0 < 256) the two terms are four more k: A carries
// (s1, s0) of this and of the previous row, B carries (256 c, c) and -(256 c, c); they sit in the two taps the
// products have to spare behind the window's last tap.  Every value is an integer, exact in f16 as an operand,
// and every partial sum stays below 2^24 (|S| and |c S_R| <= 128^2 * 441), so the accumulators hold T exactly
// in whatever order the matrix unit adds.  Windows wider than 21 do not keep that bound for T (|S| and |c S_R| reach
// 128^2 * 961 each, their partial sums 2^25): there only S slides on the matrix cores -- its partial sums stay below
// 32 * 31 * 128^2 < 2^24 for windows up to 31 x 31 -- and T = fma(c, S_R, S) is one more vector instruction per
// register, exact because T itself is an integer below 2^24 (|T| <= |cov| + |S_R| / 2).
//
//   wavefront   <-> 64 left columns (two 32-column blocks) x one range of 32*T right columns x a band of rows
//   accumulators<-> T x 2 tiles of 32 x 32 values of T(x, x'); column x on the lane, right column x' in the registers
//   score       =   (T + g S_R) / sigma_R, evaluated as fma(T, 1/sigma_R, g * (S_R / sigma_R)): the second
//                   operand pair comes per right column from LDS (two broadcast reads per 4 registers).
//                   1/sigma_L is common to a column and does not move the argmax; the refinement kernel
//                   recomputes the winners' scores exactly.
//   argmax      :   a lane owns its column, so the maximum over x' is a running v_max3 over registers; keys
//                   carry 26 bits of the score and the register id below, the two half-waves (which hold
//                   interleaved groups of 4 rows) meet once per row through one v_permlane32_swap
// The operands come from a wave-private LDS image of the current packed rows in 4 copies shifted by one
// pixel each (ds_read_b128 of 4 consecutive pixels = 8 consecutive k needs 16-byte alignment).
//
// Output: one key plane per range of right columns: upper 25 bits of the best score, below them the
// right column's index j in the range (d = dmax + ((x - r) & 63) - 32*T*range - j); with the second-peak
// output, a second plane per range: the best score among the range's right columns further than one from j.
#pragma once

#include "ncc_shared.h"

#include <type_traits>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define NCC_MF_T 3           // 32-column tiles of right columns per wavefront
#define NCC_MF_NB 2          // 32-column blocks of left columns per wavefront
#define NCC_MF_LOW (-3.0e38f)
#define NCC_MF_MAX_R 15      // widest window whose window sums S keep their partial sums below 2^24
#define NCC_MF_CORR_MAX_R 10 // widest window whose corrected sums T = S + c S_R do too: the correction rides the products

template <int R>
struct NccMfCfg {
    static constexpr bool CORR = R <= NCC_MF_CORR_MAX_R;        // the integer part of the mean correction rides the products
    static constexpr int KS = (2 * R + 1 + (CORR ? 2 : 0) + 7) / 8;   // products of 8 taps (16 k) per row step: the window's taps (and two spare ones)
    static constexpr int SEGR = 32 * NCC_MF_T + 8 * KS;         // staged right pixels
    static constexpr int SEGL = 32 * NCC_MF_NB + 8 * KS;        // staged left pixels
    static constexpr int QR = SEGR / 4, QL = SEGL / 4;          // 16-byte positions per copy
    static constexpr int SEGM = SEGR > SEGL ? SEGR : SEGL;
    static constexpr int CS = ((SEGM + 47) / 64) * 64 + 16;     // copy stride in words: >= SEGM, == 16 (mod 64): the four copies sit on disjoint bank quarters
    static int ranges(int nd) { return (((nd + 30) >> 5) + NCC_MF_NB + NCC_MF_T - 1) / NCC_MF_T; }
};

// EDGE = false: the ranges [seg0, seg0 + nranges), whose tiles all lie inside the disparity range: no tests.
// EDGE = true: the other ranges, [0, seg0) and [seg1, seg1 + nranges - seg0): tiles outside the disparity range are
// skipped, tiles on its edge masked per element.  Two kernels rather than two loops in one kernel: a kernel's
// register allocation covers all its loops, and with both loops present the 96 accumulators spill.
// WANT2: a second key plane per range: the best score among the range's right columns further than one column from
// its winner (the second peak's candidate; the refinement merges the ranges).  Right columns fall into three
// classes by index mod 3; the winner and its two neighbours sit in three different classes, so the best two keys
// of every class are enough: a class whose best is one of the three contributes its second best.
// ANYM / PARTM (bit t*NB + c): a launch for ONE range whose tiles are known when the kernel is compiled -- ANYM the
// tiles that hold candidates at all, PARTM those that have to be masked per element (EDGE = false, range = seg0).  The
// first range of every search and the last ranges of the usual +-2^k searches get such kernels (NCC_MF_PATTERNS): the
// run-time tests of the EDGE kernel cost it a quarter of its speed, and those ranges are a third of a +-256 search.
template <int R, bool EDGE, bool WANT2, uint32_t ANYM = 0u, uint32_t PARTM = 0u>
__global__ __launch_bounds__(64, 2) void ncc_mfma_argmax_kernel(
    NccArgs a, int nblocks, int nranges, int seg0, int seg1,
    const uint32_t* __restrict__ gLpack, const float* __restrict__ gNegc, const float* __restrict__ gNegg,
    const uint32_t* __restrict__ gRpack, const float* __restrict__ gRsr, const float* __restrict__ gRir,
    float* __restrict__ oKeys, float* __restrict__ oKeys2)
{
    using Cfg = NccMfCfg<R>;
    constexpr int KS = Cfg::KS, T = NCC_MF_T, NB = NCC_MF_NB, CS = Cfg::CS, QR = Cfg::QR, QL = Cfg::QL;
    constexpr int WN = 2 * R + 1;
    constexpr bool CORR = Cfg::CORR;
    static_assert(R <= NCC_MF_MAX_R, "partial sums must stay below 2^24");
    static_assert(32 * (2 * R + 1) * 128 * 128 < (1 << 24), "partial sums of S: window, entering and leaving row");
    static_assert(NB == 2, "the half-wave merge pairs two column blocks");
    static_assert(16 * T <= 64, "register tags are inline constants");
    static_assert(T == 3, "the scored rows place their memory operations by tile and column block");
    constexpr bool SPEC = ANYM != 0u;             // tiles fixed at compile time
    static_assert(!SPEC || !EDGE, "a launch with compile-time tiles maps its range like the interior kernel");
    static_assert((PARTM & ~ANYM) == 0u && ANYM < (1u << (T * NB)), "tile masks");
    constexpr uint32_t ANYC = SPEC ? ANYM : (1u << (T * NB)) - 1u;
    // the k-th present tile of a row, counted over (t, c) in order: where the row's memory operations go
    constexpr int NPRESENT = __builtin_popcount(ANYC);
    // the scored rows spread their loads and the key store over the row (see fetch_part); with four products per tile the
    // kernels have no registers to spare for the longer live ranges and keep the burst at the top of the row
    constexpr bool SPREAD = KS <= 3;
    static_assert(!CORR || 8 * KS - 2 > 2 * R, "two spare taps behind the window");
    static_assert(8 * KS > 2 * R, "the products cover the window's taps");
    __shared__ __attribute__((aligned(16))) float sR[4 * CS];
    __shared__ __attribute__((aligned(16))) float sL[4 * CS];
    __shared__ __attribute__((aligned(16))) float sIr[32 * T];          // 1 / sigma_R of the row being scored
    __shared__ __attribute__((aligned(16))) float sQ[32 * T];           // S_R / sigma_R
    __shared__ __attribute__((aligned(16))) uint32_t sC[2][CORR ? 32 * T : 4];   // CORR: half2(s1, s0) of S_R, rows of even / odd parity
    __shared__ __attribute__((aligned(16))) float sSr[CORR ? 4 : 32 * T];        // otherwise: S_R of the row being scored

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
        range = EDGE ? (range < seg0 ? range : seg1 + (range - seg0)) : seg0 + range;
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

    // tile (t, c): right columns XR0 + 32t + m, left columns X0 + 32c + n; d = dmax - 32g + (n - m), g = T*range + t - c.
    // A tile is skipped when no element lies in the disparity range (g < 0 or g > G), masked per element when
    // some do not (only in the first and last ranges).
    uint32_t any_bits = 0, part_bits = 0;           // bit t*NB + c
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int g = T * range + t - c;
            const bool any = g >= 0 && g <= G, full = g >= 1 && 32 * g + 31 <= nd - 1;
            any_bits |= (uint32_t)any << (t * NB + c);
            part_bits |= (uint32_t)(any && !full) << (t * NB + c);
        }
    any_bits = SPEC ? ANYM : __builtin_amdgcn_readfirstlane(any_bits);
    part_bits = SPEC ? PARTM : __builtin_amdgcn_readfirstlane(part_bits);

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
        colR[r2] = XR0 - R + 4 * qR + cR;           // tap 0 of the window centred at XR0 is column XR0 - R
        dstR[r2] = cR * CS + 4 * qR;
        colL[r2] = X0 - R + 4 * qL + cL; dstL[r2] = cL * CS + 4 * qL;
    }
    const bool okT = lane < 8 * T;
    const int colT = XR0 + 4 * (okT ? lane : 0);
    // clamped start columns: the loads are unconditional and inside the row; vectors that straddle the
    // row's ends are put right at publish time (only wavefronts at the raster's left and right edge)
    auto clamp4 = [&](int col) { return min(max(col, 0), P - 4); };
    const int ccR0 = clamp4(colR[0]), ccR1 = clamp4(colR[1]), ccT = clamp4(colT);
    const bool fixup = (ccR0 != colR[0]) || (ccR1 != colR[1]) || (ccT != colT);
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
    const int nlcol = 64 * block + n;               // left tables are stored shifted by R columns: entry x-R describes column x
    u32x4_t fr[2], fl[2], fsr, fir;
    float fnc[NB], fng[NB];
    // every load is issued unconditionally (a load inside a conditional block is waited for at its end);
    // rows are clamped into the band instead: the statistics fetched for warm-up rows and after the last row
    // are finite and never scored
    const int ylast = yb1 - 1 + R;
    // The ten loads of a row in four parts -- 0: right rows, 1: left rows, 2: right statistics, 3: left statistics.  The scored
    // rows issue them (and the key store) one part per scoring block instead of as one burst at the top of the row: a burst
    // waits for room in the CU's memory pipeline with nothing else to issue (0.4 ms of the 21 x 21 frame).
    auto fetch_part = [&](int part, int ynew) {
        const int yn = min(ynew, ylast), yo = min(max(yn - R, yb0), yb1 - 1);
        if (part == 0) {
            const uint32_t* rp = gRpack + (size_t)yn * P;
            __builtin_memcpy(&fr[0], rp + ccR0, 16);
            __builtin_memcpy(&fr[1], rp + ccR1, 16);
        } else if (part == 1) {
            const uint32_t* lp = gLpack + (size_t)yn * P;
            __builtin_memcpy(&fl[0], lp + colL[0], 16);
            __builtin_memcpy(&fl[1], lp + colL[1], 16);
        } else if (part == 2) {
            __builtin_memcpy(&fsr, gRsr + (size_t)yo * P + ccT, 16);
            __builtin_memcpy(&fir, gRir + (size_t)yo * P + ccT, 16);
        } else {
            const float* cp = gNegc + (size_t)yo * P;
            const float* gp = gNegg + (size_t)yo * P;
#pragma unroll
            for (int c = 0; c < NB; c++) { fnc[c] = cp[nlcol + 32 * c]; fng[c] = gp[nlcol + 32 * c]; }
        }
    };
    auto fetch = [&](int ynew) {
#pragma unroll
        for (int part = 0; part < 4; part++) fetch_part(part, ynew);
    };
    const uint32_t qnan = 0x7FC00000u;
    // row: the output row the statistics belong to (its parity picks the slot of the (s1, s0) pairs)
    auto publish = [&](bool warm, int row) {
        if (warm) {                                  // nothing leaves the window yet: drop the leaving-row halves of the left row
#pragma unroll
            for (int e = 0; e < 4; e++) { fl[0][e] &= 0x0000FFFFu; fl[1][e] &= 0x0000FFFFu; }
        }
        if (fixup) {
            fr[0] = fix4(fr[0], colR[0], ccR0, P, 0u);
            fr[1] = fix4(fr[1], colR[1], ccR1, P, 0u);
            fsr = fix4(fsr, colT, ccT, P, 0u);
            fir = fix4(fir, colT, ccT, P, qnan);
        }
        if (okR[0]) *(u32x4_t*)(sR + dstR[0]) = fr[0];
        if (okR[1]) *(u32x4_t*)(sR + dstR[1]) = fr[1];
        if (okL[0]) *(u32x4_t*)(sL + dstL[0]) = fl[0];
        if (okL[1]) *(u32x4_t*)(sL + dstL[1]) = fl[1];
        if (okT) {
            const f32x4_t srv = __builtin_bit_cast(f32x4_t, fsr), irv = __builtin_bit_cast(f32x4_t, fir);
            const f32x4_t q4 = srv * irv;
            *(f32x4_t*)(sIr + 4 * lane) = irv;
            *(f32x4_t*)(sQ + 4 * lane) = q4;
            if constexpr (CORR) {
                const f32x4_t s1v = __builtin_elementwise_floor(srv * (1.0f / 256.0f));
                const f32x4_t s0v = srv - 256.0f * s1v;      // exact: integers below 2^17
                uint32_t pk[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    half2_t p;
                    p.x = (_Float16)s1v[e]; p.y = (_Float16)s0v[e];
                    pk[e] = __builtin_bit_cast(uint32_t, p);
                }
                const u32x4_t cv = { pk[0], pk[1], pk[2], pk[3] };
                *(u32x4_t*)(&sC[row & 1][4 * lane]) = cv;
            } else {
                *(f32x4_t*)(sSr + 4 * lane) = srv;
            }
        }
    };

    // ---- per-lane read positions ----
    const int c4 = n & 3;
    const int oLane = c4 * CS + (n - c4) + 4 * h;   // pixel n + 4h of copy c4, 16-byte aligned
    const uint32_t keep_h0 = h ? 0u : 0xFFFFFFFFu;  // taps past the window in the upper half-wave's share of a product
    const uint32_t kmask = 0xFFFFFFC0u;

    f32x16_t acc[T][NB];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int c = 0; c < NB; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][c][i] = 0.f;
    // (s1, s0) slots start finite: they are multiplied by zero until the first scored row
    if constexpr (CORR)
        for (int i = lane; i < 2 * 32 * T; i += 64) (&sC[0][0])[i] = 0u;

    // left operands of the current row: B[k][x], 8 k per lane and product.  The last product's two spare taps
    // (upper half-wave, elements 2 and 3) carry the correction pairs cur = (256 c, c) and prv = -(256 c, c) of the
    // previous row.
    auto load_b = [&](u32x4_t (&bop)[NB][KS], const uint32_t (&cur)[NB], const uint32_t (&prv)[NB]) {
#pragma unroll
        for (int c = 0; c < NB; c++)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                u32x4_t v = *(const u32x4_t*)(sL + oLane + 32 * c + 8 * kk);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int tap0 = 8 * kk + e, tap1 = tap0 + 4;        // this element's tap in the lower / upper half-wave
                    if (CORR && kk == KS - 1 && e >= 2) {
                        const uint32_t lo = tap0 > 2 * R ? 0u : v[e];
                        v[e] = h ? (e == 2 ? cur[c] : prv[c]) : lo;
                    } else if (tap0 > 2 * R) v[e] = 0u;
                    else if (tap1 > 2 * R) v[e] &= keep_h0;
                }
                bop[c][kk] = v;
            }
    };
    // right operands of tile t: A[x'][k]; the spare taps carry (s1, s0) of the row being scored and of the row before
    auto load_a = [&](int t, u32x4_t (&aop)[KS], int row) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++) aop[kk] = *(const u32x4_t*)(sR + oLane + 32 * t + 8 * kk);
        if constexpr (!CORR) return;
        uint32_t ccur = sC[row & 1][32 * t + n], cprv = sC[(row & 1) ^ 1][32 * t + n];
        asm volatile("" : "+v"(ccur), "+v"(cprv));   // both reads in every lane: inside a branch on h they would wait for all earlier reads
        aop[KS - 1][2] = h ? ccur : aop[KS - 1][2];
        aop[KS - 1][3] = h ? cprv : aop[KS - 1][3];
    };
    auto slide = [&](int t, int c, const u32x4_t (&aop)[KS], const u32x4_t (&bop)[NB][KS]) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
            acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, aop[kk]),
                                                              __builtin_bit_cast(half8_t, bop[c][kk]), acc[t][c], 0, 0, 0);
    };
    // (256 c, c) as a pair of halves; c is an integer of magnitude <= 128
    auto cpair = [&](float c) {
        half2_t p;
        p.x = (_Float16)(256.0f * c); p.y = (_Float16)c;
        return __builtin_bit_cast(uint32_t, p);
    };

    // the half-waves' keys of a row -> one key per left column -> key plane
    // the key of a row is stored while the NEXT row is scored (store_keys): right behind the row's loads it would queue with them
    uint32_t key_out = 0u, key2_out = 0u;
    auto store_keys = [&](int yrow) {
        const int x = X0 + lane;
        if (x < a.W - R) {
            oKeys[((size_t)range * a.H + yrow) * a.out_pitch + x] = __builtin_bit_cast(float, key_out);
            if (WANT2) oKeys2[((size_t)range * a.H + yrow) * a.out_pitch + x] = __builtin_bit_cast(float, key2_out);
        }
    };
    auto emit = [&](float b0, float b1) {
        // the two half-waves hold different right columns of the same left column: bring them together
        // (lanes 0-31 then own block 0, lanes 32-63 block 1), compare on the 26-bit scores, the larger
        // right column (lower disparity) keeps a tie
        float k0 = b0, k1 = b1;
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
        key_out = (sb & 0xFFFFFF80u) | (uint32_t)(up ? j1 : j0);
        return up ? j1 : j0;
    };
    // second plane: m1 / m2 = best / second-best key per class of the lane's share of right columns, [block][label];
    // a register's label is its right column's index mod 3 in the lower half-wave, one less than it (mod 3) in the upper
    auto emit2 = [&](float (&m1)[NB][3], float (&m2)[NB][3], int jstar) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(m1[0][q]), "+v"(m1[1][q]));   // [0]: the lower half-wave's, [1]: the upper's
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(m2[0][q]), "+v"(m2[1][q]));
        }
        float k2 = NCC_MF_LOW;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float a1 = m1[0][q], a2 = m2[0][q], b1 = m1[1][(q + 2) % 3], b2 = m2[1][(q + 2) % 3];
            const bool bw = b1 > a1;
            const float hi = bw ? b1 : a1, lo = bw ? a1 : b1;
            const uint32_t tg = __builtin_bit_cast(uint32_t, hi) & 63u;
            const int jm = (int)(32u * (tg >> 4) + 8u * ((tg >> 2) & 3u) + (tg & 3u)) + (bw ? 4 : 0);
            float nx = lo > a2 ? lo : a2;
            nx = nx > b2 ? nx : b2;
            const float cand = ((unsigned)(jm - jstar + 1) <= 2u) ? nx : hi;
            k2 = cand > k2 ? cand : k2;
        }
        key2_out = __builtin_bit_cast(uint32_t, k2) & 0xFFFFFF80u;
    };

    // masked tiles of a launch with compile-time tiles: 1 where the element's disparity lies in the range, NaN elsewhere
    // (the tests do not depend on the row; the factors stay in registers where accumulators and factors together fit)
    constexpr int NPART = __builtin_popcount(SPEC ? PARTM : 0u);
    constexpr int NPART_HELD = (NPART >= 1 && NPRESENT + NPART <= 7) ? NPART : 0;      // 16 registers per tile, present or masked: at most 112
    float nanm[NPART_HELD > 0 ? NPART_HELD : 1][16];
    if constexpr (NPART_HELD > 0) {
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (!((PARTM >> (t * NB + c)) & 1u)) continue;
                const int kp = __builtin_popcount(PARTM & ((1u << (t * NB + c)) - 1u));
                const int vb = 32 * (T * range + t - c) + 4 * h - n;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const bool in = (unsigned)(vb + 8 * (i >> 2) + (i & 3)) <= (unsigned)(nd - 1);
                    nanm[kp < NPART_HELD ? kp : 0][i] = in ? 1.0f : __builtin_bit_cast(float, qnan);
                }
            }
    }

    // ---- warm-up rows: the window fills, nothing is scored, no correction ----
    const uint32_t zero2[NB] = { 0u, 0u };
    fetch(yb0 - R);
    publish(true, yb0 - 1);
    for (int st = 0; st < 2 * R; st++) {
        fetch(yb0 - R + st + 1);
        u32x4_t bop[NB][KS];
        load_b(bop, zero2, zero2);
#pragma unroll
        for (int t = 0; t < T; t++) {
            if (SPEC ? !((ANYM >> (t * NB)) & ((1u << NB) - 1u)) : (EDGE && !((any_bits >> (t * NB)) & ((1u << NB) - 1u)))) continue;
            u32x4_t aop[KS];
            load_a(t, aop, 0);
#pragma unroll
            for (int c = 0; c < NB; c++)
                if (SPEC ? ((ANYM >> (t * NB + c)) & 1u) != 0u : (!EDGE || ((any_bits >> (t * NB + c)) & 1u))) slide(t, c, aop, bop);
        }
        // the row published by the last warm-up step is the first scored one: its statistics go to its slot
        publish(st + 1 < WN, st + 1 == 2 * R ? yb0 : yb0 - 1);
    }

    // ---- scored rows ----
    {
    uint32_t cprev[NB] = { 0u, 0u };                 // -(256 c, c) of the previous scored row (nothing to take back at the first one)
    for (int st = 2 * R; st < nsteps; st++) {
        uint32_t ab = any_bits, pb = part_bits;
        if (EDGE) asm volatile("" : "+s"(ab), "+s"(pb));        // tested bit by bit where needed, not as a dozen loop-invariant lane masks
        const int ynew = yb0 - R + st, yout = ynew - R;
        const float ng[NB] = { fng[0], fng[1] };     // fetched with this row's operands during the previous step
        const float nc[NB] = { fnc[0], fnc[1] };
        uint32_t ccur[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) ccur[c] = CORR ? cpair(fnc[c]) : 0u;
        if constexpr (SPREAD && NPRESENT >= 5) fetch_part(0, ynew + 1);
        else fetch(ynew + 1);
        u32x4_t bop[NB][KS];
        load_b(bop, ccur, cprev);
#pragma unroll
        for (int c = 0; c < NB; c++) cprev[c] = ccur[c] ^ 0x80008000u;
        float best[NB] = { NCC_MF_LOW, NCC_MF_LOW };
        float m1[NB][3], m2[NB][3];
        if (WANT2) {
#pragma unroll
            for (int c = 0; c < NB; c++)
#pragma unroll
                for (int q = 0; q < 3; q++) m1[c][q] = m2[c][q] = NCC_MF_LOW;
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            if (SPREAD && EDGE && t >= 1) {                // (EDGE: all six tiles may be present)
                // tiles may be skipped here: the rest of the row's memory operations go between the tiles
                __builtin_amdgcn_sched_barrier(0);
                if (t == 1) { fetch_part(1, ynew + 1); fetch_part(2, ynew + 1); }
                if (t == 2) { fetch_part(3, ynew + 1); if (st > 2 * R) store_keys(yout - 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SPEC ? !((ANYM >> (t * NB)) & ((1u << NB) - 1u)) : (EDGE && !((ab >> (t * NB)) & ((1u << NB) - 1u)))) continue;
            u32x4_t aop[KS];
            load_a(t, aop, yout);
            f32x4_t irv[4], qv[4], srv[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                irv[b] = *(const f32x4_t*)(sIr + 32 * t + 8 * b + 4 * h);
                qv[b] = *(const f32x4_t*)(sQ + 32 * t + 8 * b + 4 * h);
                if constexpr (!CORR) srv[b] = *(const f32x4_t*)(sSr + 32 * t + 8 * b + 4 * h);
            }
#pragma unroll
            for (int c = 0; c < NB; c++)
                if (SPEC ? ((ANYM >> (t * NB + c)) & 1u) != 0u : (!EDGE || ((ab >> (t * NB + c)) & 1u))) slide(t, c, aop, bop);
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (SPEC && !((ANYM >> (t * NB + c)) & 1u)) continue;
                // the k-th present block of the row (k = t * NB + c where all six are)
                const int kblk = __builtin_popcount(ANYC & ((1u << (t * NB + c)) - 1u));
                if (SPREAD && !EDGE && NPRESENT >= 5 && kblk >= 1 && kblk <= 4) {
                    // one part of the next row's loads, or the previous row's key store, in front of this block's statements
                    __builtin_amdgcn_sched_barrier(0);
                    if (kblk <= 3) fetch_part(kblk, ynew + 1);
                    else if (st > 2 * R) store_keys(yout - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (EDGE && !((ab >> (t * NB + c)) & 1u)) continue;
                float ir[16];
#pragma unroll
                for (int i = 0; i < 16; i++) ir[i] = irv[i >> 2][i & 3];
                if constexpr (SPEC && NPART_HELD > 0) {
                    // masked tiles of a launch with compile-time tiles: the per-element factors (1 or NaN) were laid down
                    // in registers before the rows, one multiplication per element
                    if ((PARTM >> (t * NB + c)) & 1u) {
                        const int kp = __builtin_popcount(PARTM & ((1u << (t * NB + c)) - 1u));
#pragma unroll
                        for (int i = 0; i < 16; i++) ir[i] *= nanm[kp < NPART_HELD ? kp : 0][i];
                    }
                } else
                if (SPEC ? ((PARTM >> (t * NB + c)) & 1u) != 0u : (EDGE && ((pb >> (t * NB + c)) & 1u))) {
                    // a tile on the edge of the disparity range: 0 <= 32g + (m - n) <= nd - 1 decides per element;
                    // elements outside get a NaN factor and drop out of the maximum (the empty statement keeps
                    // the tests inside this rarely taken branch)
                    int vb = 32 * (T * range + t - c) + 4 * h - n;
                    asm volatile("" : "+v"(vb));
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const bool in = (unsigned)(vb + 8 * (i >> 2) + (i & 3)) <= (unsigned)(nd - 1);
                        ir[i] = in ? ir[i] : __builtin_bit_cast(float, qnan);
                    }
                }
                // The matrix unit's results may not be read by a vector instruction before the product has drained (wait
                // states that software must insert).  The compiler counts them for instructions it emits itself, not for
                // the text of an asm statement, and its scheduler moves the products right in front of the statements that
                // read their accumulators.  So one instruction of the compiler's own reads the tile first (it gets the
                // wait states), and every statement below takes its result as an operand, which orders them behind it.
                const float drained = __builtin_fmaf(acc[t][c][0], 0.f, 0.f);
                // one statement per 4 registers (the compiler pads every asm statement with a wait state of its own):
                // p = g * (S_R / sigma_R), score = fma(T, 1 / sigma_R, p), key, running maximum
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float k0, k1, k2, k3;
                    if constexpr (!CORR) {
                        // windows beyond 21 x 21: the accumulators hold S; T = fma(c, S_R, S) (exact), score = T / sigma_R + g S_R / sigma_R
                        asm("v_fma_f32 %0, %8, %10, %4\n\t"
                            "v_fma_f32 %1, %8, %11, %5\n\t"
                            "v_fma_f32 %2, %8, %12, %6\n\t"
                            "v_fma_f32 %3, %8, %13, %7\n\t"
                            "v_mul_f32 %0, %0, %18\n\t"
                            "v_mul_f32 %1, %1, %19\n\t"
                            "v_mul_f32 %2, %2, %20\n\t"
                            "v_mul_f32 %3, %3, %21\n\t"
                            "v_fmac_f32 %0, %9, %14\n\t"
                            "v_fmac_f32 %1, %9, %15\n\t"
                            "v_fmac_f32 %2, %9, %16\n\t"
                            "v_fmac_f32 %3, %9, %17\n\t"
                            "v_and_or_b32 %0, %0, %22, %23\n\t"
                            "v_and_or_b32 %1, %1, %22, %24\n\t"
                            "v_and_or_b32 %2, %2, %22, %25\n\t"
                            "v_and_or_b32 %3, %3, %22, %26"
                            : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
                            : "v"(acc[t][c][4 * j]), "v"(acc[t][c][4 * j + 1]), "v"(acc[t][c][4 * j + 2]), "v"(acc[t][c][4 * j + 3]),
                              "v"(nc[c]), "v"(ng[c]), "v"(srv[j][0]), "v"(srv[j][1]), "v"(srv[j][2]), "v"(srv[j][3]),
                              "v"(qv[j][0]), "v"(qv[j][1]), "v"(qv[j][2]), "v"(qv[j][3]),
                              "v"(ir[4 * j]), "v"(ir[4 * j + 1]), "v"(ir[4 * j + 2]), "v"(ir[4 * j + 3]), "s"(kmask),
                              "n"(16 * t + 4 * j), "n"(16 * t + 4 * j + 1), "n"(16 * t + 4 * j + 2), "n"(16 * t + 4 * j + 3), "v"(drained));
                        if constexpr (WANT2) {
                            const int L = (2 * t + 2 * j) % 3;
                            float &A1 = m1[c][L], &A2 = m2[c][L], &B1 = m1[c][(L + 1) % 3], &B2 = m2[c][(L + 1) % 3],
                                  &C1 = m1[c][(L + 2) % 3], &C2 = m2[c][(L + 2) % 3];
                            asm("v_med3_f32 %1, %0, %1, %6\n\t"
                                "v_med3_f32 %3, %2, %3, %7\n\t"
                                "v_med3_f32 %5, %4, %5, %8\n\t"
                                "v_max_f32 %0, %0, %6\n\t"
                                "v_max_f32 %2, %2, %7\n\t"
                                "v_max_f32 %4, %4, %8\n\t"
                                "v_med3_f32 %1, %0, %1, %9\n\t"
                                "v_max_f32 %0, %0, %9"
                                : "+v"(A1), "+v"(A2), "+v"(B1), "+v"(B2), "+v"(C1), "+v"(C2)
                                : "v"(k0), "v"(k1), "v"(k2), "v"(k3));
                        } else {
                            asm("v_max3_f32 %0, %0, %1, %2\n\t"
                                "v_max3_f32 %0, %0, %3, %4"
                                : "+v"(best[c]) : "v"(k0), "v"(k1), "v"(k2), "v"(k3));
                        }
                        continue;
                    }
                    if constexpr (WANT2) {
                        // the four registers' labels are L, L+1, L+2, L (mod 3): best two per class by v_med3 / v_max
                        const int L = (2 * t + 2 * j) % 3;
                        float &A1 = m1[c][L], &A2 = m2[c][L], &B1 = m1[c][(L + 1) % 3], &B2 = m2[c][(L + 1) % 3],
                              &C1 = m1[c][(L + 2) % 3], &C2 = m2[c][(L + 2) % 3];
                        asm("v_mul_f32 %6, %14, %15\n\t"
                            "v_mul_f32 %7, %14, %16\n\t"
                            "v_mul_f32 %8, %14, %17\n\t"
                            "v_mul_f32 %9, %14, %18\n\t"
                            "v_fmac_f32 %6, %10, %19\n\t"
                            "v_fmac_f32 %7, %11, %20\n\t"
                            "v_fmac_f32 %8, %12, %21\n\t"
                            "v_fmac_f32 %9, %13, %22\n\t"
                            "v_and_or_b32 %6, %6, %23, %24\n\t"
                            "v_and_or_b32 %7, %7, %23, %25\n\t"
                            "v_and_or_b32 %8, %8, %23, %26\n\t"
                            "v_and_or_b32 %9, %9, %23, %27\n\t"
                            "v_med3_f32 %1, %0, %1, %6\n\t"
                            "v_med3_f32 %3, %2, %3, %7\n\t"
                            "v_med3_f32 %5, %4, %5, %8\n\t"
                            "v_max_f32 %0, %0, %6\n\t"
                            "v_max_f32 %2, %2, %7\n\t"
                            "v_max_f32 %4, %4, %8\n\t"
                            "v_med3_f32 %1, %0, %1, %9\n\t"
                            "v_max_f32 %0, %0, %9"
                            : "+v"(A1), "+v"(A2), "+v"(B1), "+v"(B2), "+v"(C1), "+v"(C2), "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
                            : "v"(acc[t][c][4 * j]), "v"(acc[t][c][4 * j + 1]), "v"(acc[t][c][4 * j + 2]), "v"(acc[t][c][4 * j + 3]),
                              "v"(ng[c]), "v"(qv[j][0]), "v"(qv[j][1]), "v"(qv[j][2]), "v"(qv[j][3]),
                              "v"(ir[4 * j]), "v"(ir[4 * j + 1]), "v"(ir[4 * j + 2]), "v"(ir[4 * j + 3]), "s"(kmask),
                              "n"(16 * t + 4 * j), "n"(16 * t + 4 * j + 1), "n"(16 * t + 4 * j + 2), "n"(16 * t + 4 * j + 3), "v"(drained));
                        continue;
                    }
                    asm("v_mul_f32 %1, %9, %10\n\t"
                        "v_mul_f32 %2, %9, %11\n\t"
                        "v_mul_f32 %3, %9, %12\n\t"
                        "v_mul_f32 %4, %9, %13\n\t"
                        "v_fmac_f32 %1, %5, %14\n\t"
                        "v_fmac_f32 %2, %6, %15\n\t"
                        "v_fmac_f32 %3, %7, %16\n\t"
                        "v_fmac_f32 %4, %8, %17\n\t"
                        "v_and_or_b32 %1, %1, %18, %19\n\t"
                        "v_and_or_b32 %2, %2, %18, %20\n\t"
                        "v_and_or_b32 %3, %3, %18, %21\n\t"
                        "v_and_or_b32 %4, %4, %18, %22\n\t"
                        "v_max3_f32 %0, %0, %1, %2\n\t"
                        "v_max3_f32 %0, %0, %3, %4"
                        : "+v"(best[c]), "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
                        : "v"(acc[t][c][4 * j]), "v"(acc[t][c][4 * j + 1]), "v"(acc[t][c][4 * j + 2]), "v"(acc[t][c][4 * j + 3]),
                          "v"(ng[c]), "v"(qv[j][0]), "v"(qv[j][1]), "v"(qv[j][2]), "v"(qv[j][3]),
                          "v"(ir[4 * j]), "v"(ir[4 * j + 1]), "v"(ir[4 * j + 2]), "v"(ir[4 * j + 3]), "s"(kmask),
                          "n"(16 * t + 4 * j), "n"(16 * t + 4 * j + 1), "n"(16 * t + 4 * j + 2), "n"(16 * t + 4 * j + 3), "v"(drained));
                }
            }
        }

        // single-buffered LDS image: every read of this row was issued above and LDS executes a wavefront's
        // operations in order, so the next row may be written now
        publish(false, yout + 1);
        if constexpr (WANT2) {
#pragma unroll
            for (int c = 0; c < NB; c++) {
                const float m01 = m1[c][0] > m1[c][1] ? m1[c][0] : m1[c][1];
                best[c] = m01 > m1[c][2] ? m01 : m1[c][2];
            }
            const int jstar = emit(best[0], best[1]);
            emit2(m1, m2, jstar);
        } else {
            emit(best[0], best[1]);
        }
        if constexpr (!(SPREAD && (EDGE || NPRESENT >= 5))) store_keys(yout);
    }
    if (SPREAD && (EDGE || NPRESENT >= 5) && nsteps > 2 * R) store_keys(yb1 - 1);         // the last row's keys
    }
}
