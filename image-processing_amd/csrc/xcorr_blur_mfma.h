// The 151-tap Gaussian of the difference-of-Gaussian peak pick (image3.cpp:481-483) on the matrix cores, gfx950.
//
// A 1-D filter along a line is a product with a banded Toeplitz matrix: out[n] = sum_j g[j] * pad[n + j], i.e.
// Out(32 lines x 32 outputs) = Pad(32 lines x 192 samples) * G(192 x 32), G[k][n] = g[k - n] inside the band.
// v_mfma_f32_32x32x16_f16 takes f16 operands, and the surfaces need more than f16's 11 bits, so every operand is
// split in two halves (value = hi + lo, lo = the f16 nearest to the remainder: 22 bits together) and a product
// is three matrix products, hi*hi + hi*lo + lo*hi, accumulated in f32 (the lo*lo term is below 2^-22 of the
// result).  The taps are scaled by 128 for the split so that their low halves stay normal f16 numbers.
// The sums are not accumulated in the tap order of the vector kernels (which follow the filter loops of the
// reference's library); the results differ from theirs by about 3e-8 of the surface's range (measured on 512 x 512 tiles).
//
//   rows pass    one workgroup = 32 rows of a tile: normalise (image3.cpp:414-416), the padded rows
//                (BORDER_REFLECT_101) as f16 halves in LDS, 32 x 32 output tiles per wavefront
//   columns pass one workgroup = 32 columns of a tile over all rows: the row-filtered columns transposed into LDS
//                (a column is a "line" there), the same product, then DoG = surface - blur and the first maximum
// Both need tile sides that are multiples of 32 (no odd-size rule applies) and at most 1024 (LDS).
#pragma once

typedef _Float16 xb_half8 __attribute__((ext_vector_type(8)));
typedef float xb_f32x16 __attribute__((ext_vector_type(16)));

#define XB_KSTEPS 12            // 16-sample steps: 32 outputs + 150 more samples = 182 <= 192
#define XB_TAPSCALE 128.0f

// line stride in f16 elements: room for L + 160 samples (75 + L + 75 reflected ones, zero padded to the end of
// the last 16-sample step), 16-byte aligned lines whose starts fall on different bank groups
static inline int xb_line_stride(int L) { int s = L + 160; while (s % 16 != 8) s++; return s; }
static inline size_t xb_lds_bytes(int L) { return (size_t)2 * 32 * xb_line_stride(L) * sizeof(_Float16); }
static inline bool xb_fits(int w, int h)
{
    return w % 32 == 0 && h % 32 == 0 && w >= 32 && h >= 32 && xb_lds_bytes(w) <= 160 * 1024 && xb_lds_bytes(h) <= 160 * 1024;
}

// The Toeplitz operand: lane l holds, for output n = l & 31, the taps of samples 16 s + 8 (l >> 5) + e (e = 0..7) of step s.
// Built once per context into a table of 2 * XB_KSTEPS * 64 vectors ([step][hi, lo][lane]), 24 coalesced loads per lane
// in the filter kernels.
__global__ void xb_build_taps(const float* __restrict__ gauss, xb_half8* __restrict__ table)
{
    const int lane = threadIdx.x, n = lane & 31, kg = lane >> 5;
    for (int s = 0; s < XB_KSTEPS; s++) {
        xb_half8 h, l;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int gi = 16 * s + 8 * kg + e - n;
            const float v = (gi >= 0 && gi < XC_GAUSS_K) ? gauss[gi] * XB_TAPSCALE : 0.f;
            const _Float16 a = (_Float16)v;
            h[e] = a;
            l[e] = (_Float16)(v - (float)a);
        }
        table[(2 * s) * 64 + lane] = h;
        table[(2 * s + 1) * 64 + lane] = l;
    }
}
__device__ __forceinline__ void xb_taps(const xb_half8* __restrict__ table, int lane, xb_half8 (&bh)[XB_KSTEPS], xb_half8 (&bl)[XB_KSTEPS])
{
#pragma unroll
    for (int s = 0; s < XB_KSTEPS; s++) {
        bh[s] = table[(2 * s) * 64 + lane];
        bl[s] = table[(2 * s + 1) * 64 + lane];
    }
}

// 32 lines x 32 outputs starting at sample `first` of every line (lines of this lane: lane & 31)
__device__ __forceinline__ xb_f32x16 xb_block(const _Float16* hi, const _Float16* lo, int SW, int first, int lane,
                                              const xb_half8 (&bh)[XB_KSTEPS], const xb_half8 (&bl)[XB_KSTEPS])
{
    xb_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const int base = (lane & 31) * SW + first + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < XB_KSTEPS; s++) {
        const xb_half8 ah = *(const xb_half8*)(hi + base + 16 * s);
        const xb_half8 al = *(const xb_half8*)(lo + base + 16 * s);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc, 0, 0, 0);
    }
    return acc;
}

// stage 4 on the matrix cores: grid (h / 32, tiles), 512 threads.  Same outputs as xc_norm_rowblur: the normalised
// surface back into `surf`, the row-filtered one into `tmp`.
__global__ __launch_bounds__(512) void xc_norm_rowblur_mf(int w, int h, int SW, const unsigned* __restrict__ minmax,
                                                          const xb_half8* __restrict__ taps,
                                                          float* __restrict__ surf, float* __restrict__ tmp, int pre)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char xb_smem[];
    _Float16* hi = (_Float16*)xb_smem;
    _Float16* lo = hi + 32 * SW;
    const int t = blockIdx.y, y0 = 32 * blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float mn = __uint_as_float(minmax[2 * t]), mx = __uint_as_float(minmax[2 * t + 1]);
    const float range = mx - mn;
    xb_half8 bh[XB_KSTEPS], bl[XB_KSTEPS];
    if (32 * wv < w) xb_taps(taps, lane, bh, bl);   // wavefronts without an output block (narrow tiles) only stage
    float* sblk = surf + ((size_t)t * h + y0) * w;  // the block's 32 rows are contiguous
    // 16 bytes per thread and load, eight loads in flight before the first store (the stores go to the array
    // the loads read, so nothing is reordered across them by the compiler)
    const int total = 32 * w;
    for (int f0 = 4 * tid; f0 < total; f0 += 8 * 2048) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *(const float4*)(sblk + min(f0 + 2048 * k, total - 4));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int f = f0 + 2048 * k;
            if (f >= total) continue;
            float o[4] = { v[k].x, v[k].y, v[k].z, v[k].w };
            if (!pre) {                                  // pre: the surface already holds the product of the normalised band surfaces
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (float)(((double)o[e] - (double)mn) / (double)range);
                *(float4*)(sblk + f) = make_float4(o[0], o[1], o[2], o[3]);   // the normalised surface, kept for stage 5
            }
            const int r = f / w, x = f - r * w;         // w is a multiple of 4: the four samples share a row
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const _Float16 a = (_Float16)o[e];
                hi[r * SW + XC_GAUSS_R + x + e] = a;
                lo[r * SW + XC_GAUSS_R + x + e] = (_Float16)(o[e] - (float)a);
            }
        }
    }
    __syncthreads();
    // the 75 reflected samples on either side, zeros up to the end of the last step
    for (int idx = tid; idx < 32 * 160; idx += 512) {
        const int r = idx / 160, j = idx % 160;
        const int i = j < XC_GAUSS_R ? j : w + j;   // j = 75..159 -> samples w + 75 .. w + 159
        _Float16 vh = (_Float16)0.f, vl = (_Float16)0.f;
        if (i < w + 2 * XC_GAUSS_R) {
            const int sx = reflect101(i - XC_GAUSS_R, w);
            vh = hi[r * SW + XC_GAUSS_R + sx];
            vl = lo[r * SW + XC_GAUSS_R + sx];
        }
        hi[r * SW + i] = vh;
        lo[r * SW + i] = vl;
    }
    __syncthreads();
    float* tblk = tmp + ((size_t)t * h + y0) * w;
    const int n = lane & 31, kg = lane >> 5;
    for (int b = wv; b < w / 32; b += 8) {
        const xb_f32x16 acc = xb_block(hi, lo, SW, 32 * b, lane, bh, bl);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int row = 8 * (i >> 2) + 4 * kg + (i & 3);
            tblk[(size_t)row * w + 32 * b + n] = acc[i] * (1.0f / XB_TAPSCALE);
        }
    }
}

// stage 5 on the matrix cores: grid (w / 32, tiles), 512 threads.  Same outputs as xc_colblur_argmax.
__global__ __launch_bounds__(512) void xc_colblur_argmax_mf(int w, int h, int SW, const xb_half8* __restrict__ taps,
                                                            const float* __restrict__ surf, const float* __restrict__ tmp,
                                                            float* __restrict__ dog_out, unsigned long long* __restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char xb_smem[];
    _Float16* hi = (_Float16*)xb_smem;
    _Float16* lo = hi + 32 * SW;
    const int t = blockIdx.y, x0 = 32 * blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    xb_half8 bh[XB_KSTEPS], bl[XB_KSTEPS];
    if (32 * wv < h) xb_taps(taps, lane, bh, bl);   // wavefronts without an output block (short tiles) only stage
    const float* tt = tmp + (size_t)t * h * w;
    // the columns as lines: sample i of line c is row reflect(i - 75) of column x0 + c; 16 rows per pass, sixteen
    // passes' loads in flight before the first store
    const int c = tid & 31, ri = tid >> 5;
    const int npass = (h + 160) / 16;               // h is a multiple of 32
    for (int q0 = 0; q0 < npass; q0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = min(16 * (q0 + k) + ri, h + 2 * XC_GAUSS_R - 1);
            v[k] = tt[(size_t)reflect101(i - XC_GAUSS_R, h) * w + x0 + c];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = 16 * (q0 + k) + ri;
            if (q0 + k < npass) {
                const float o = i < h + 2 * XC_GAUSS_R ? v[k] : 0.f;
                const _Float16 a = (_Float16)o;
                hi[c * SW + i] = a;
                lo[c * SW + i] = (_Float16)(o - (float)a);
            }
        }
    }
    __syncthreads();
    const int n = lane & 31, kg = lane >> 5;
    unsigned long long best = 0;
    for (int b = wv; b < h / 32; b += 8) {
        const xb_f32x16 acc = xb_block(hi, lo, SW, 32 * b, lane, bh, bl);
        const int y = 32 * b + n;
        const size_t orow = (size_t)t * h * w + (size_t)y * w;
#pragma unroll
        for (int i4 = 0; i4 < 4; i4++) {
            const int x = x0 + 8 * i4 + 4 * kg;
            const float4 sf = *(const float4*)(surf + orow + x);
            const float s4[4] = { sf.x, sf.y, sf.z, sf.w };
            float d4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                d4[e] = __fsub_rn(s4[e], acc[4 * i4 + e] * (1.0f / XB_TAPSCALE));
                if (d4[e] > -1.0f) {
                    const unsigned long long k = argmax_key(d4[e], (unsigned)(y * w + x + e));
                    best = k > best ? k : best;
                }
            }
            if (dog_out) *(float4*)(dog_out + orow + x) = make_float4(d4[0], d4[1], d4[2], d4[3]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    tile_commit_key(keys, t, best);
}

// ---------------------------------------------------------------------------------------------------------------------
// Lines longer than the LDS image above holds (tiles of 2048 and up), and even sides that are not multiples of 32 (the
// whole rasters of the recursion's first level): the same products on SEGMENTS of XB_SEG outputs of a line, each with its
// 75 + 75 neighbours (reflected at the line's ends only).  A segment's neighbours belong to other workgroups, so the rows
// pass cannot normalise the surface in place as xc_norm_rowblur_mf does: it reads the raw surface, normalises on the fly
// (a product with the reciprocal of the range in double: the filter's operands keep 22 bits anyway) and writes only the
// row-filtered plane; the columns pass, which reads every sample of the surface exactly once for the DoG, normalises it
// there as the reference does (a division in double, narrowed) and writes it back.
// Partial blocks (rows, columns or outputs beyond the tile) are clamped on the way in and masked on the way out.
// ---------------------------------------------------------------------------------------------------------------------
#define XB_SEG 256
#define XB_SEG_STRIDE 424       // xb_line_stride(XB_SEG): 256 + 160 samples, rounded up to 8 mod 16
static inline size_t xb_seg_lds_bytes() { return (size_t)2 * 32 * XB_SEG_STRIDE * sizeof(_Float16); }
static inline bool xb_seg_fits(int w, int h) { return w % 4 == 0 && h % 2 == 0 && w >= 32 && h >= 32; }

// rows pass: grid (ceil(h / 32), tiles, ceil(w / XB_SEG)), 512 threads
__global__ __launch_bounds__(512) void xc_rowblur_seg_mf(int w, int h, const unsigned* __restrict__ minmax,
                                                         const xb_half8* __restrict__ taps,
                                                         const float* __restrict__ surf, float* __restrict__ tmp, int pre)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char xb_smem[];
    constexpr int SW = XB_SEG_STRIDE;
    constexpr int LW = XB_SEG + 160;
    _Float16* hi = (_Float16*)xb_smem;
    _Float16* lo = hi + 32 * SW;
    const int t = blockIdx.y, y0 = 32 * blockIdx.x, x0 = XB_SEG * blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float mn = __uint_as_float(minmax[2 * t]), mx = __uint_as_float(minmax[2 * t + 1]);
    const double dmn = (double)mn, rrange = 1.0 / (double)(mx - mn);
    xb_half8 bh[XB_KSTEPS], bl[XB_KSTEPS];
    xb_taps(taps, lane, bh, bl);
    const float* st = surf + (size_t)t * h * w;
    const int nout = min(XB_SEG, w - x0);           // outputs of this segment
    const int nvalid = nout + 2 * XC_GAUSS_R;       // samples of the line image that exist; zeros behind them
    for (int idx0 = tid; idx0 < 32 * LW; idx0 += 8 * 512) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = min(idx0 + 512 * k, 32 * LW - 1);
            const int r = idx / LW, i = idx - r * LW;
            const int sx = reflect101(x0 + min(i, nvalid - 1) - XC_GAUSS_R, w);
            v[k] = st[(size_t)min(y0 + r, h - 1) * w + sx];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = idx0 + 512 * k;
            if (idx >= 32 * LW) continue;
            const int r = idx / LW, i = idx - r * LW;
            float o = v[k];
            if (!pre) o = (float)(((double)o - dmn) * rrange);
            if (i >= nvalid) o = 0.f;
            const _Float16 a = (_Float16)o;
            hi[r * SW + i] = a;
            lo[r * SW + i] = (_Float16)(o - (float)a);
        }
    }
    __syncthreads();
    float* tblk = tmp + (size_t)t * h * w;
    const int n = lane & 31, kg = lane >> 5;
    for (int b = wv; 32 * b < nout; b += 8) {
        const xb_f32x16 acc = xb_block(hi, lo, SW, 32 * b, lane, bh, bl);
        const int x = x0 + 32 * b + n;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int row = y0 + 8 * (i >> 2) + 4 * kg + (i & 3);
            if (row < h && x < w) tblk[(size_t)row * w + x] = acc[i] * (1.0f / XB_TAPSCALE);
        }
    }
}

// columns pass: grid (ceil(w / 32), tiles, ceil(h / XB_SEG)), 512 threads; w a multiple of 4
__global__ __launch_bounds__(512) void xc_colblur_seg_mf(int w, int h, const unsigned* __restrict__ minmax,
                                                         const xb_half8* __restrict__ taps, float* __restrict__ surf,
                                                         const float* __restrict__ tmp, float* __restrict__ dog_out,
                                                         unsigned long long* __restrict__ keys, int pre)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char xb_smem[];
    constexpr int SW = XB_SEG_STRIDE;
    constexpr int LW = XB_SEG + 160;
    _Float16* hi = (_Float16*)xb_smem;
    _Float16* lo = hi + 32 * SW;
    const int t = blockIdx.y, x0 = 32 * blockIdx.x, y0 = XB_SEG * blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float mn = __uint_as_float(minmax[2 * t]), mx = __uint_as_float(minmax[2 * t + 1]);
    const double dmn = (double)mn, drange = (double)(mx - mn);
    xb_half8 bh[XB_KSTEPS], bl[XB_KSTEPS];
    xb_taps(taps, lane, bh, bl);
    const float* tt = tmp + (size_t)t * h * w;
    const int nout = min(XB_SEG, h - y0);
    const int nvalid = nout + 2 * XC_GAUSS_R;
    // the columns as lines: sample i of line c is row reflect(y0 + i - 75) of column x0 + c; 16 samples per pass
    const int c = tid & 31, ri = tid >> 5;
    const int xc = min(x0 + c, w - 1);
    for (int q0 = 0; q0 < LW / 16; q0 += 13) {      // LW / 16 = 26 passes, thirteen in flight
        float v[13];
#pragma unroll
        for (int k = 0; k < 13; k++) {
            const int i = min(16 * (q0 + k) + ri, nvalid - 1);
            v[k] = tt[(size_t)reflect101(y0 + i - XC_GAUSS_R, h) * w + xc];
        }
#pragma unroll
        for (int k = 0; k < 13; k++) {
            const int i = 16 * (q0 + k) + ri;
            const float o = i < nvalid ? v[k] : 0.f;
            const _Float16 a = (_Float16)o;
            hi[c * SW + i] = a;
            lo[c * SW + i] = (_Float16)(o - (float)a);
        }
    }
    __syncthreads();
    const int n = lane & 31, kg = lane >> 5;
    float* st = surf + (size_t)t * h * w;
    unsigned long long best = 0;
    for (int b = wv; 32 * b < nout; b += 8) {
        const xb_f32x16 acc = xb_block(hi, lo, SW, 32 * b, lane, bh, bl);
        const int y = y0 + 32 * b + n;
        const size_t orow = (size_t)min(y, h - 1) * w;
#pragma unroll
        for (int i4 = 0; i4 < 4; i4++) {
            const int x = x0 + 8 * i4 + 4 * kg;                  // x .. x + 3 inside the tile or all outside (w % 4 == 0)
            const bool in = y < h && x < w;
            const float4 sf = *(const float4*)(st + orow + min(x, w - 4));
            float s4[4] = { sf.x, sf.y, sf.z, sf.w };
            float d4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (!pre) s4[e] = (float)(((double)s4[e] - dmn) / drange);
                d4[e] = __fsub_rn(s4[e], acc[4 * i4 + e] * (1.0f / XB_TAPSCALE));
                if (in && d4[e] > -1.0f) {
                    const unsigned long long k = argmax_key(d4[e], (unsigned)(y * w + x + e));
                    best = k > best ? k : best;
                }
            }
            if (in) {
                if (!pre) *(float4*)(st + orow + x) = make_float4(s4[0], s4[1], s4[2], s4[3]);
                if (dog_out) *(float4*)(dog_out + (size_t)t * h * w + orow + x) = make_float4(d4[0], d4[1], d4[2], d4[3]);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    tile_commit_key(keys, t, best);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tiles of 32 or 64 samples a side (the last levels of the recursion, thousands of tiles each): with BORDER_REFLECT_101 the
// 151 taps fold onto the L samples of a line, so the filter along a line is a product with a fixed L x L matrix,
// F[n][k] = sum of the taps g[j] with reflect101(n + j - 75, L) == k, and the whole blur of a tile is F_h * X * F_w^T: two
// small products, no padded lines, no LDS.  One wavefront owns the 32 columns `bx` of a tile: rows pass Y = X F_w^T for its
// columns (the A operand is read straight from the surface: 8 consecutive floats of a row per lane and step), columns pass
// Z = F_h Y with Y taken from the accumulators as the B operand.  A lane's accumulators of a 32 x 32 block hold rows
// 8 (i >> 2) + 4 (lane >> 5) + (i & 3) of column lane & 31, the B operand of step s wants eight K indices per lane: the
// columns pass simply sums over K in the order the accumulators have (K index of step s, element e of a lane:
// 32 (s >> 1) + 8 (2 (s & 1) + (e >> 2)) + 4 (lane >> 5) + (e & 3)), and the table of F_h is laid out to match.
// Same splitting into f16 halves, same scaling and the same outputs as the two kernels above together (normalised surface,
// here into the plane that holds the row-filtered surface there, DoG, first maximum).
// EXPERIMENTAL, off unless MISTEREO_XC_SMALL=1.  Equal to the oracle within the tests' bars and 5x faster than the two
// general kernels on 9520 tiles of 64 x 64 (0.17 against 0.95 ms), bit-reproducible alone on one stream -- but with two
// wavefronts per tile (64 samples wide) and other kernels running beside it (the second half of a batch on the side stream, or
// other contexts on the device) a few tiles per thousand came back with a DoG peak off by 1e-4..1e-2, the same call repeated a
// moment later gave the right value, and which shapes were hit moved with every change of the instruction order (acquire /
// release fences around the surface, agent-scope loads, wait states around the products, operands held in their registers:
// each moved the rate, one combination brought it to zero in 250 000 tiles, none explained it).  The strongest lead: with the
// surface read by volatile loads (which go past the caches) the rate on two streams fell from 40 % of the calls to none in
// 40 000 tiles, and to 2 tiles in 170 000 beside two other contexts, a release fence at the end of the kernel that writes the
// surface changing nothing: the wavefronts here read a tile on another XCD than the one that wrote it and seem to meet
// samples of an earlier call in that XCD's caches.  The two general kernels have never shown this under the same load (their
// workgroups read a tile from the XCD that wrote it: block index and tile go together in both kernels).
// Table of a length L (NB = L / 32 blocks, KS = L / 16 steps), vectors of 8 halves per lane:
//   rows operand     [(s * NB + bx) * 2 + hl][lane]   F[32 bx + (lane & 31)][16 s + 8 (lane >> 5) + e] * 128
//   columns operand  [(by * KS + s) * 2 + hl][lane]   F[32 by + (lane & 31)][K index above] * 128, after the rows operand
// ---------------------------------------------------------------------------------------------------------------------
static inline size_t xb_small_vectors(int L) { return (size_t)2 * (L / 16) * (L / 32) * 2 * 64; }
static inline bool xb_small_fits(int w, int h) { return (w == 32 || w == 64) && (h == 32 || h == 64); }

// taps: the 151 float taps; out: xb_small_vectors(L) vectors of 8 halves (as raw 16-bit patterns via _Float16 on the host)
static void xb_build_small(const float* taps, int L, _Float16* out)
{
    const int NB = L / 32, KS = L / 16;
    std::vector<double> F((size_t)L * L, 0.0);
    for (int n = 0; n < L; n++)
        for (int j = 0; j < XC_GAUSS_K; j++) {
            int p = n + j - XC_GAUSS_R;
            while ((unsigned)p >= (unsigned)L) p = p < 0 ? -p : 2 * L - 2 - p;
            F[(size_t)n * L + p] += (double)taps[j];
        }
    auto put = [&](size_t vec, int lane, int e, double v) {
        const float f = (float)v * XB_TAPSCALE;
        const _Float16 a = (_Float16)f;
        out[((vec * 2 + 0) * 64 + lane) * 8 + e] = a;
        out[((vec * 2 + 1) * 64 + lane) * 8 + e] = (_Float16)(f - (float)a);
    };
    const size_t cols0 = (size_t)KS * NB;          // first vector pair of the columns operand
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 31, kg = lane >> 5;
        for (int s = 0; s < KS; s++)
            for (int b = 0; b < NB; b++)
                for (int e = 0; e < 8; e++) {
                    put((size_t)s * NB + b, lane, e, F[(size_t)(32 * b + n) * L + 16 * s + 8 * kg + e]);
                    const int k = 32 * (s >> 1) + 8 * (2 * (s & 1) + (e >> 2)) + 4 * kg + (e & 3);
                    put(cols0 + (size_t)b * KS + s, lane, e, F[(size_t)(32 * b + n) * L + k]);
                }
    }
}

// Eight idle cycles between an operand's last conversion to f16 and its first matrix product (xc_blur_small_mf feeds
// conversion results straight into the products; the kernels above go through LDS); part of the search described at that kernel.
#define XB_SETTLE(h, l) asm volatile("s_nop 7" : "+v"(h), "+v"(l))
__device__ __forceinline__ void xb_split8(const float (&v)[8], xb_half8& h, xb_half8& l)
{
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const _Float16 a = (_Float16)v[e];
        h[e] = a;
        l[e] = (_Float16)(v[e] - (float)a);
    }
}

// stages 4 and 5 of tiles of 32 NBW x 32 NBH samples: grid ceil(tiles * NBW / 4), 256 threads (four independent
// wavefronts).  tw: the table of the width, th: the table of the height.  The normalised surface goes to a plane of its
// own (`norm_out`, tile by tile like `surf`; not written when `pre`), NOT back into `surf`: a wavefront of a tile reads
// samples that another wavefront of the tile would overwrite.
template <int NBW, int NBH>
__global__ __launch_bounds__(256) void xc_blur_small_mf(int nt, const unsigned* __restrict__ minmax,
                                                        const xb_half8* __restrict__ tw, const xb_half8* __restrict__ th,
                                                        const float* __restrict__ surf, float* __restrict__ norm_out,
                                                        float* __restrict__ dog_out,
                                                        unsigned long long* __restrict__ keys, int pre)
{
    constexpr int W = 32 * NBW, H = 32 * NBH, KSW = 2 * NBW, KSH = 2 * NBH;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, kg = lane >> 5;
    const int wid = blockIdx.x * 4 + (tid >> 6);
    if (wid >= nt * NBW) return;                    // whole wavefronts; no barrier below
    const int t = wid / NBW, bx = wid % NBW;
    const float* st = surf + (size_t)t * W * H;
    const float mn = __uint_as_float(minmax[2 * t]), mx = __uint_as_float(minmax[2 * t + 1]);
    const double dmn = (double)mn, drange = (double)(mx - mn);
    // the whole tile as A operands, normalised as the reference does (image3.cpp:414-416, in double, narrowed)
    xb_half8 ah[NBH][KSW], al[NBH][KSW];
    {
        float4 raw[NBH][KSW][2];
#pragma unroll
        for (int by = 0; by < NBH; by++)
#pragma unroll
            for (int s = 0; s < KSW; s++) {
                const float4* p = (const float4*)(st + (size_t)(32 * by + n) * W + 16 * s + 8 * kg);
                raw[by][s][0] = p[0]; raw[by][s][1] = p[1];
            }
#pragma unroll
        for (int by = 0; by < NBH; by++)
#pragma unroll
            for (int s = 0; s < KSW; s++) {
                float v[8] = { raw[by][s][0].x, raw[by][s][0].y, raw[by][s][0].z, raw[by][s][0].w,
                               raw[by][s][1].x, raw[by][s][1].y, raw[by][s][1].z, raw[by][s][1].w };
                if (!pre) {
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = (float)(((double)v[e] - dmn) / drange);
                }
                xb_split8(v, ah[by][s], al[by][s]);
                XB_SETTLE(ah[by][s], al[by][s]);
            }
    }
    // this wavefront's own columns in the accumulators' layout
    float own[NBH][16];
#pragma unroll
    for (int by = 0; by < NBH; by++)
#pragma unroll
        for (int i = 0; i < 16; i++)
            own[by][i] = st[(size_t)(32 * by + 8 * (i >> 2) + 4 * kg + (i & 3)) * W + 32 * bx + n];
    // rows pass
    xb_f32x16 Y[NBH];
    xb_half8 bh[KSW], bl[KSW];
    {
#pragma unroll
        for (int s = 0; s < KSW; s++) {
            bh[s] = tw[((size_t)(s * NBW + bx) * 2 + 0) * 64 + lane];
            bl[s] = tw[((size_t)(s * NBW + bx) * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int by = 0; by < NBH; by++) {
#pragma unroll
            for (int i = 0; i < 16; i++) Y[by][i] = 0.f;
#pragma unroll
            for (int s = 0; s < KSW; s++) {
                Y[by] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[by][s], bh[s], Y[by], 0, 0, 0);
                Y[by] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[by][s], bl[s], Y[by], 0, 0, 0);
                Y[by] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[by][s], bh[s], Y[by], 0, 0, 0);
            }
        }
    }
    // the row-filtered columns as B operands of the columns pass
    xb_half8 yh[KSH], yl[KSH];
#pragma unroll
    for (int s = 0; s < KSH; s++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = Y[s >> 1][8 * (s & 1) + e] * (1.0f / XB_TAPSCALE);
        xb_split8(v, yh[s], yl[s]);
        XB_SETTLE(yh[s], yl[s]);
    }
    const xb_half8* tc = th + (size_t)KSH * NBH * 2 * 64;
    unsigned long long best = 0;
#pragma unroll
    for (int by = 0; by < NBH; by++) {
        xb_f32x16 Z;
#pragma unroll
        for (int i = 0; i < 16; i++) Z[i] = 0.f;
        xb_half8 fh[KSH], fl[KSH];
#pragma unroll
        for (int s = 0; s < KSH; s++) {
            fh[s] = tc[((size_t)(by * KSH + s) * 2 + 0) * 64 + lane];
            fl[s] = tc[((size_t)(by * KSH + s) * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < KSH; s++) {
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[s], yh[s], Z, 0, 0, 0);
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s], yl[s], Z, 0, 0, 0);
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s], yh[s], Z, 0, 0, 0);
        }
        // the A operand's lines are the output rows: a lane's accumulators are rows of ITS column x again
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int y = 32 * by + 8 * (i >> 2) + 4 * kg + (i & 3), x = 32 * bx + n;
            float xn = own[by][i];
            if (!pre) {
                xn = (float)(((double)xn - dmn) / drange);
                norm_out[(size_t)t * W * H + (size_t)y * W + x] = xn;
            }
            const float d = __fsub_rn(xn, Z[i] * (1.0f / XB_TAPSCALE));
            if (dog_out) dog_out[(size_t)t * W * H + (size_t)y * W + x] = d;
            if (d > -1.0f) {
                const unsigned long long k = argmax_key(d, (unsigned)(y * W + x));
                best = k > best ? k : best;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    // one atomic per wavefront: the wavefronts of this workgroup belong to different tiles (tile_commit_key combines a
    // workgroup through LDS, and the slots of wavefronts that left above would hold whatever the last kernel wrote there)
    if (lane == 0 && best) atomicMax(&keys[t], best);
}
