// Batched FFT tile cross-correlation with difference-of-Gaussian peak pick, gfx950.
//
// Replaces findOffset + runFFT (image3.cpp:318-335, 346-517), CalcFFT of image-cross-cor.cpp
// (104-176) and CalcFFT of image-fft.cpp (45-105).  Stages per tile of w x h pixels (N = w*h):
//   1 load both tiles (zero outside the bounding box) as ONE complex raster a + i b,
//     forward row FFTs                                                               xc_rows_fwd
//   2 forward column FFTs, A*conj(B) from the joint spectrum, inverse column FFT,     xc_cols
//     fused
//   3 inverse row FFT, |c|, tile min / max                                           xc_rows_inv_abs
//   4 min-max normalisation to [0,1] and the row pass of the 151-tap Gaussian         xc_norm_rowblur[_mf]
//   5 column pass, difference of Gaussians, first maximum                             xc_colblur_argmax[_mf]
//     (_mf: the filter as a banded Toeplitz product on the matrix cores, xcorr_blur_mfma.h, for tile sides that
//     are multiples of 32 up to 1024; the vector kernels follow the tap order of the reference's filter loops)
//   6 wrap the peak position to a signed offset                                       xc_finish
// Every 1-D transform lives in LDS.  Forward transforms are decimation-in-frequency (natural
// order in, bit-reversed out), inverse ones decimation-in-time (bit-reversed in, natural out),
// so the spectrum stays in bit-reversed order in both dimensions between stages 1 and 3 and no
// reordering pass exists: the pointwise product does not care about the order.
#include "mis_common.h"

#include <math.h>
#include <string.h>
#include <map>
#include <utility>
#include <vector>

#define XC_GAUSS_K 151          // cv::Size(151,151) at image3.cpp:481
#define XC_GAUSS_R 75
#define XC_GAUSS_SIGMA 99.0

struct XcorrWs {
    std::map<int, float2*> twiddle;     // per FFT length n: n/2 values exp(-2 pi i k / n)
    std::map<int, float2*> chirp;       // per non-power-of-two length n: n chirp values, then the M filter values
    std::map<int, float2*> circle;      // per length n with prime factors <= 13: n values exp(-2 pi i t / n)
    float* gauss = nullptr;             // 151 taps
    void* taps_mf = nullptr;            // the taps as matrix-core operands (xcorr_blur_mfma.h), built on first use
    std::map<int, void*> taps_small;    // per tile side 32 / 64: the folded filter matrix as matrix-core operands
    hipStream_t side = nullptr;         // second stream: the two halves of a batch run side by side
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* buf = nullptr; size_t buf_bytes = 0;      // spectra / surfaces of one batch of tiles
    void* meta = nullptr; size_t meta_bytes = 0;    // per-tile min/max, argmax keys, rects
    void* io = nullptr; size_t io_bytes = 0;        // rectangles in, offsets out of a host call (mis_xcorr_tiles): kept between
                                                    // calls, the recursion makes one call per tile size and level
    void* io_host = nullptr; size_t io_host_bytes = 0;   // the same, page-locked, on the host: one copy up, one copy down
    const float* last_surface = nullptr;            // normalised surfaces of the last batch (inspection)
    const float* last_abs = nullptr;                // |c| surfaces of the last batch (before normalisation)
};

void mis_xcorr_release(mis_ctx* ctx)
{
    XcorrWs* w = ctx->xc;
    if (!w) return;
    for (auto& kv : w->twiddle) (void)hipFree(kv.second);
    for (auto& kv : w->chirp) (void)hipFree(kv.second);
    for (auto& kv : w->circle) (void)hipFree(kv.second);
    if (w->gauss) (void)hipFree(w->gauss);
    if (w->taps_mf) (void)hipFree(w->taps_mf);
    for (auto& kv : w->taps_small) (void)hipFree(kv.second);
    if (w->side) {
        (void)hipStreamSynchronize(w->side);
        (void)hipEventDestroy(w->ev_fork); (void)hipEventDestroy(w->ev_join);
        (void)hipStreamDestroy(w->side);
    }
    if (w->buf) (void)hipFree(w->buf);
    if (w->meta) (void)hipFree(w->meta);
    if (w->io) (void)hipFree(w->io);
    if (w->io_host) (void)hipHostFree(w->io_host);
    delete w;
    ctx->xc = nullptr;
}

static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }
static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static int get_twiddle(mis_ctx* ctx, int n, const float2** out)
{
    XcorrWs* w = ctx->xc;
    auto it = w->twiddle.find(n);
    if (it != w->twiddle.end()) { *out = it->second; return MIS_OK; }
    std::vector<float2> h((size_t)(n / 2 > 0 ? n / 2 : 1));
    for (int k = 0; k < n / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        h[k] = make_float2((float)cos(a), (float)sin(a));
    }
    float2* d = nullptr;
    MIS_HIP(ctx, hipMalloc((void**)&d, h.size() * sizeof(float2)));
    MIS_HIP(ctx, hipMemcpyAsync(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    w->twiddle[n] = d;
    *out = d;
    return MIS_OK;
}

// OpenCV 2.4 getGaussianKernel(151, 99, CV_32F): taps exp(-x^2/(2 sigma^2)) evaluated in double,
// stored as float, normalised by the double sum of the stored floats
static int ensure_ws(mis_ctx* ctx)
{
    if (!ctx->xc) ctx->xc = new XcorrWs();
    XcorrWs* w = ctx->xc;
    if (!w->gauss) {
        float k[XC_GAUSS_K];
        double scale2x = -0.5 / (XC_GAUSS_SIGMA * XC_GAUSS_SIGMA), sum = 0;
        for (int i = 0; i < XC_GAUSS_K; i++) {
            double x = i - (XC_GAUSS_K - 1) * 0.5;
            k[i] = (float)exp(scale2x * x * x);
            sum += k[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < XC_GAUSS_K; i++) k[i] = (float)(k[i] * sum);
        MIS_HIP(ctx, hipMalloc((void**)&w->gauss, sizeof k));
        MIS_HIP(ctx, hipMemcpyAsync(w->gauss, k, sizeof k, hipMemcpyHostToDevice, ctx->stream));
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------
// In-LDS radix-2 transforms on `cnt` independent sequences of length n = 2^logn laid out back to
// back (sequence s occupies x[s*n .. s*n+n)).  All threads of the workgroup take part; the caller
// has synchronised before the call, and the routines end with a barrier.
// tw: n/2 values exp(-2 pi i k / n).
// NT: threads of the workgroup, a template parameter here and a literal in the kernels below
// (every kernel is launched with exactly its __launch_bounds__): blockDim.x is a load from the
// dispatch packet, a memory round trip the compiler repeats in every pass.
// ---------------------------------------------------------------------------------------------
// (fused multiply-adds are allowed in the transforms: FFTW's own codelets use them where the
// machine has them, and no particular rounding of the spectrum is part of the reference's behaviour)
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
#pragma clang fp contract(fast)
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// LDS index swizzle of the padded layouts: one unused element after every eight.  The later passes
// address LDS with lane strides of 8 or 64 elements (64 or 512 bytes), which map whole groups of
// lanes onto the same banks; with the pad the strides become 9 and 72 elements and a wavefront's
// accesses spread over all banks.  Arrays hold n + n/8 elements (n a multiple of 8).
template <bool PAD>
__device__ __forceinline__ int xp(int i) { return PAD ? i + (i >> 3) : i; }

// Keeps every element of v[] behind this point: loads issued above cannot be moved (back) into the
// conditional blocks below, where the compiler would wait for each one separately.
template <int U>
__device__ __forceinline__ void pin_loaded(float4 (&v)[U])
{
#pragma unroll
    for (int u = 0; u < U; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
}

// Contiguous global -> LDS copy of n float2 (n even, 16-byte aligned on both sides) by an NT-thread
// workgroup: 16-byte loads, U of them in flight per thread (clamped addresses, tested stores).
template <int NT, int U, bool PAD = false>
__device__ __forceinline__ void stage_contiguous(float2* dst, const float2* __restrict__ src, int n)
{
    if (n & 1) {                                     // degenerate sizes only
        for (int i = threadIdx.x; i < n; i += NT) dst[i] = src[i];
        return;
    }
    const int n4 = n >> 1;
    const float4* s4 = (const float4*)src;
    float4* d4 = (float4*)dst;
    for (int base = 0; base < n4; base += NT * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = s4[min(base + (int)threadIdx.x + NT * u, n4 - 1)];
        pin_loaded(v);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + (int)threadIdx.x + NT * u;
            if (i < n4) {
                if (PAD) {                           // the pad breaks the 16-byte alignment of every second pair
                    dst[xp<PAD>(2 * i)] = make_float2(v[u].x, v[u].y);
                    dst[xp<PAD>(2 * i + 1)] = make_float2(v[u].z, v[u].w);
                } else d4[i] = v[u];
            }
        }
    }
}

// Up to three consecutive radix-2 stages are done in registers by the thread that owns the 2^S
// elements they connect (indices base + m * 2^l_lo, m = 0 .. 2^S - 1): the arithmetic is exactly
// that of S radix-2 passes, with a third of the LDS traffic and barriers.
// Stage with half = 2^l: butterfly (i, i + half), twiddle index (i mod half) << (logn - 1 - l).
template <int S, int NT, bool PAD>
__device__ __forceinline__ void fft_dif_pass(float2* x, int logn, int cnt, int l_hi, const float2* __restrict__ tw)
{
    constexpr int M = 1 << S;
    const int l_lo = l_hi - S + 1;
    const int n = 1 << logn;
    const int gps = n >> S, lgps = logn - S;       // groups per sequence
    const int groups = cnt * gps;
    for (int g = threadIdx.x; g < groups; g += NT) {
        const int seq = g >> lgps, gi = g & (gps - 1);       // gps is a power of two
        const int low = gi & ((1 << l_lo) - 1), high = gi >> l_lo;
        const int i0 = (high << (l_hi + 1)) + low;            // index of element m = 0 inside the sequence
        const int b0 = seq * n + i0;
        float2 v[M];
#pragma unroll
        for (int m = 0; m < M; m++) v[m] = x[xp<PAD>(b0 + (m << l_lo))];
#pragma unroll
        for (int st = 0; st < S; st++) {
            const int l = l_hi - st;
            const int hm = M >> (st + 1);                      // partner distance in units of m
#pragma unroll
            for (int m = 0; m < M; m++) {
                if (m & hm) continue;
                const int k = (i0 + (m << l_lo)) & ((1 << l) - 1);
                const float2 w = tw[k << (logn - 1 - l)];
                const float2 a = v[m], c = v[m + hm];
                v[m] = make_float2(a.x + c.x, a.y + c.y);
                v[m + hm] = cmul(make_float2(a.x - c.x, a.y - c.y), w);
            }
        }
#pragma unroll
        for (int m = 0; m < M; m++) x[xp<PAD>(b0 + (m << l_lo))] = v[m];
    }
    __syncthreads();
}

// Radix-8 passes: three radix-2 stages whose twiddles factor into fixed eighth roots of unity inside the
// 8-point butterfly and ONE power of w = exp(-2 pi i low / 2^(l_hi+1)) per element outside it (element m of the
// group gets w^bitrev3(m): after the butterfly in the forward transform, before it in the inverse one).  Seven
// complex multiplications per 8 elements instead of twelve, none in the pass whose groups start at low = 0.
// w, w^2, w^4 are table values (the twiddles the three radix-2 stages use for k = low), the other powers products.
#define XC_RSQRT2 0.70710678118654752440f
template <bool INV>
__device__ __forceinline__ void fft_powers(const float2* __restrict__ tw, int low, int sh, float2 (&p)[8])
{
    float2 w1 = tw[low << sh], w2 = tw[low << (sh + 1)], w4 = tw[low << (sh + 2)];
    if (INV) { w1.y = -w1.y; w2.y = -w2.y; w4.y = -w4.y; }
    const float2 w3 = cmul(w1, w2);
    p[1] = w4; p[2] = w2; p[3] = cmul(w2, w4); p[4] = w1; p[5] = cmul(w1, w4); p[6] = w3; p[7] = cmul(w3, w4);
}

template <int NT, bool PAD>
__device__ __forceinline__ void fft_dif_pass8(float2* x, int logn, int cnt, int l_hi, const float2* __restrict__ tw)
{
    const int l_lo = l_hi - 2;
    const int n = 1 << logn;
    const int gps = n >> 3, lgps = logn - 3;
    const int groups = cnt * gps;
    const int sh = logn - 1 - l_hi;
    for (int g = threadIdx.x; g < groups; g += NT) {
        const int seq = g >> lgps, gi = g & (gps - 1);       // gps is a power of two
        const int low = gi & ((1 << l_lo) - 1), high = gi >> l_lo;
        const int b0 = seq * n + (high << (l_hi + 1)) + low;
        float2 v[8];
#pragma unroll
        for (int m = 0; m < 8; m++) v[m] = x[xp<PAD>(b0 + (m << l_lo))];
        // stage (m, m + 4): difference times exp(-2 pi i m / 8)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const float2 a = v[m], c = v[m + 4];
            v[m] = make_float2(a.x + c.x, a.y + c.y);
            const float dx = a.x - c.x, dy = a.y - c.y;
            v[m + 4] = m == 0 ? make_float2(dx, dy)
                     : m == 1 ? make_float2((dx + dy) * XC_RSQRT2, (dy - dx) * XC_RSQRT2)
                     : m == 2 ? make_float2(dy, -dx)
                              : make_float2((dy - dx) * XC_RSQRT2, (-dx - dy) * XC_RSQRT2);
        }
        // stage (m, m + 2): difference times exp(-2 pi i (m & 1) / 4)
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (m & 2) continue;
            const float2 a = v[m], c = v[m + 2];
            v[m] = make_float2(a.x + c.x, a.y + c.y);
            const float dx = a.x - c.x, dy = a.y - c.y;
            v[m + 2] = (m & 1) ? make_float2(dy, -dx) : make_float2(dx, dy);
        }
        // stage (m, m + 1)
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
            const float2 a = v[m], c = v[m + 1];
            v[m] = make_float2(a.x + c.x, a.y + c.y);
            v[m + 1] = make_float2(a.x - c.x, a.y - c.y);
        }
        if (l_lo > 0) {
            float2 p[8];
            fft_powers<false>(tw, low, sh, p);
#pragma unroll
            for (int m = 1; m < 8; m++) v[m] = cmul(v[m], p[m]);
        }
#pragma unroll
        for (int m = 0; m < 8; m++) x[xp<PAD>(b0 + (m << l_lo))] = v[m];
    }
    __syncthreads();
}

template <int NT, bool PAD>
__device__ __forceinline__ void fft_dit_pass8(float2* x, int logn, int cnt, int l_lo, const float2* __restrict__ tw)
{
    const int l_hi = l_lo + 2;
    const int n = 1 << logn;
    const int gps = n >> 3, lgps = logn - 3;
    const int groups = cnt * gps;
    const int sh = logn - 1 - l_hi;
    for (int g = threadIdx.x; g < groups; g += NT) {
        const int seq = g >> lgps, gi = g & (gps - 1);       // gps is a power of two
        const int low = gi & ((1 << l_lo) - 1), high = gi >> l_lo;
        const int b0 = seq * n + (high << (l_hi + 1)) + low;
        float2 v[8];
#pragma unroll
        for (int m = 0; m < 8; m++) v[m] = x[xp<PAD>(b0 + (m << l_lo))];
        if (l_lo > 0) {
            float2 p[8];
            fft_powers<true>(tw, low, sh, p);
#pragma unroll
            for (int m = 1; m < 8; m++) v[m] = cmul(v[m], p[m]);
        }
        // stage (m, m + 1)
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
            const float2 a = v[m], c = v[m + 1];
            v[m] = make_float2(a.x + c.x, a.y + c.y);
            v[m + 1] = make_float2(a.x - c.x, a.y - c.y);
        }
        // stage (m, m + 2): second operand times exp(+2 pi i (m & 1) / 4)
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (m & 2) continue;
            const float2 a = v[m], c = v[m + 2];
            const float2 t = (m & 1) ? make_float2(-c.y, c.x) : c;
            v[m] = make_float2(a.x + t.x, a.y + t.y);
            v[m + 2] = make_float2(a.x - t.x, a.y - t.y);
        }
        // stage (m, m + 4): second operand times exp(+2 pi i m / 8)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const float2 a = v[m], c = v[m + 4];
            const float2 t = m == 0 ? c
                           : m == 1 ? make_float2((c.x - c.y) * XC_RSQRT2, (c.x + c.y) * XC_RSQRT2)
                           : m == 2 ? make_float2(-c.y, c.x)
                                    : make_float2((-c.x - c.y) * XC_RSQRT2, (c.x - c.y) * XC_RSQRT2);
            v[m] = make_float2(a.x + t.x, a.y + t.y);
            v[m + 4] = make_float2(a.x - t.x, a.y - t.y);
        }
#pragma unroll
        for (int m = 0; m < 8; m++) x[xp<PAD>(b0 + (m << l_lo))] = v[m];
    }
    __syncthreads();
}

// forward (sign -1), decimation in frequency: natural order in, bit-reversed order out
template <int NT, bool PAD = false>
__device__ void fft_dif_fwd(float2* x, int logn, int cnt, const float2* __restrict__ tw)
{
    int l = logn - 1;
    while (l >= 0) {
        if (l >= 2) { fft_dif_pass8<NT, PAD>(x, logn, cnt, l, tw); l -= 3; }
        else if (l == 1) { fft_dif_pass<2, NT, PAD>(x, logn, cnt, l, tw); l -= 2; }
        else { fft_dif_pass<1, NT, PAD>(x, logn, cnt, l, tw); l -= 1; }
    }
}

template <int S, int NT, bool PAD>
__device__ __forceinline__ void fft_dit_pass(float2* x, int logn, int cnt, int l_lo, const float2* __restrict__ tw)
{
    constexpr int M = 1 << S;
    const int l_hi = l_lo + S - 1;
    const int n = 1 << logn;
    const int gps = n >> S, lgps = logn - S;
    const int groups = cnt * gps;
    for (int g = threadIdx.x; g < groups; g += NT) {
        const int seq = g >> lgps, gi = g & (gps - 1);       // gps is a power of two
        const int low = gi & ((1 << l_lo) - 1), high = gi >> l_lo;
        const int i0 = (high << (l_hi + 1)) + low;
        const int b0 = seq * n + i0;
        float2 v[M];
#pragma unroll
        for (int m = 0; m < M; m++) v[m] = x[xp<PAD>(b0 + (m << l_lo))];
#pragma unroll
        for (int st = 0; st < S; st++) {
            const int l = l_lo + st;
            const int hm = 1 << st;
#pragma unroll
            for (int m = 0; m < M; m++) {
                if (m & hm) continue;
                const int k = (i0 + (m << l_lo)) & ((1 << l) - 1);
                float2 w = tw[k << (logn - 1 - l)];
                w.y = -w.y;
                const float2 a = v[m];
                const float2 t = cmul(v[m + hm], w);
                v[m] = make_float2(a.x + t.x, a.y + t.y);
                v[m + hm] = make_float2(a.x - t.x, a.y - t.y);
            }
        }
#pragma unroll
        for (int m = 0; m < M; m++) x[xp<PAD>(b0 + (m << l_lo))] = v[m];
    }
    __syncthreads();
}

// backward (sign +1), decimation in time: bit-reversed order in, natural order out, unnormalised
template <int NT, bool PAD = false>
__device__ void fft_dit_inv(float2* x, int logn, int cnt, const float2* __restrict__ tw)
{
    int l = 0;
    while (l < logn) {
        const int left = logn - l;
        if (left >= 3) { fft_dit_pass8<NT, PAD>(x, logn, cnt, l, tw); l += 3; }
        else if (left == 2) { fft_dit_pass<2, NT, PAD>(x, logn, cnt, l, tw); l += 2; }
        else { fft_dit_pass<1, NT, PAD>(x, logn, cnt, l, tw); l += 1; }
    }
}

__device__ __forceinline__ int bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// ---------------------------------------------------------------------------------------------
struct XcImage { const void* img[2]; int dtype, iw, ih; size_t stride; };

__device__ __forceinline__ float load_px(const void* img, int dtype, size_t stride, int x, int y)
{
    const unsigned char* row = (const unsigned char*)img + (size_t)y * stride;
    if (dtype == MIS_U8) return (float)row[x];
    if (dtype == MIS_U16) return (float)((const uint16_t*)row)[x];
    return ((const float*)row)[x];
}

// Pixels of the rows of a tile pair -> LDS as complex values, image 1 in the real part and image 2 in
// the imaginary part (both inputs are real, so ONE complex transform carries both spectra: with
// Z = FFT(a + i b), A[k] = (Z[k] + conj(Z[-k])) / 2 and B[k] = (Z[k] - conj(Z[-k])) / (2i)).
// U pixel pairs in flight per thread: coordinates are clamped into the raster so that every load is
// legal and unconditional, pixels outside the bbox or the raster are replaced by 0 afterwards.  w = 2^logw.
template <typename T, int NT, int U, bool PAD>
__device__ __forceinline__ void stage_pixels(float2* x, const XcImage& im, const mis_rect b1, const mis_rect b2,
                                             int row0, int w, int logw, int n)
{
    for (int base = 0; base < n; base += NT * U) {
        float f[U], g[U];
        bool ok1[U], ok2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + (int)threadIdx.x + NT * u;
            const int ry = row0 + (i >> logw), rx = i & (w - 1);
            {
                const int sx = b1.minx + rx, sy = b1.miny + ry;
                ok1[u] = rx < b1.width && ry < b1.height && sx >= 0 && sx < im.iw && sy >= 0 && sy < im.ih;
                const int cx = min(max(sx, 0), im.iw - 1), cy = min(max(sy, 0), im.ih - 1);
                f[u] = (float)((const T*)((const unsigned char*)im.img[0] + (size_t)cy * im.stride))[cx];
            }
            {
                const int sx = b2.minx + rx, sy = b2.miny + ry;
                ok2[u] = rx < b2.width && ry < b2.height && sx >= 0 && sx < im.iw && sy >= 0 && sy < im.ih;
                const int cx = min(max(sx, 0), im.iw - 1), cy = min(max(sy, 0), im.ih - 1);
                g[u] = (float)((const T*)((const unsigned char*)im.img[1] + (size_t)cy * im.stride))[cx];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" : "+v"(f[u]), "+v"(g[u]));
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + (int)threadIdx.x + NT * u;
            if (i < n) x[xp<PAD>(i)] = make_float2(ok1[u] ? f[u] : 0.f, ok2[u] ? g[u] : 0.f);
        }
    }
}

// stage 1: grid (rows/RB, tiles).  RB rows of one tile pair per workgroup.
// spec layout: [tile][h][w] float2 (image 1 + i * image 2, columns in bit-reversed order)
// PAD: padded LDS layout (xp), chosen by the host whenever the padded array fits.
template <bool PAD>
__global__ __launch_bounds__(256) void xc_rows_fwd(XcImage im, const mis_rect* __restrict__ r1,
                                                   const mis_rect* __restrict__ r2, int tile0,
                                                   int w, int h, int logw, int RB,
                                                   const float2* __restrict__ tw, float2* __restrict__ spec,
                                                   unsigned* __restrict__ minmax, unsigned long long* __restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* x = (float2*)smem_raw;
    const int t = blockIdx.y, row0 = blockIdx.x * RB;
    const mis_rect b1 = r1[tile0 + t], b2 = r2[tile0 + t];
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // the tile's extrema (and, first band only, its peak key) start over
        minmax[2 * t] = 0x7F800000u;                // +inf
        minmax[2 * t + 1] = 0u;
        if (keys) keys[t] = 0ull;
    }
    // image3.cpp:322-328: the buffer is zeroed, then the bbox region is read into it
    if (im.dtype == MIS_U8) stage_pixels<uint8_t, 256, 4, PAD>(x, im, b1, b2, row0, w, logw, RB * w);
    else if (im.dtype == MIS_U16) stage_pixels<uint16_t, 256, 4, PAD>(x, im, b1, b2, row0, w, logw, RB * w);
    else stage_pixels<float, 256, 4, PAD>(x, im, b1, b2, row0, w, logw, RB * w);
    float2* stw = x + (((size_t)xp<PAD>(RB * w) + 1) & ~(size_t)1);   // twiddles in LDS: every butterfly reads one
    stage_contiguous<256, 2>(stw, tw, w / 2);
    __syncthreads();
    fft_dif_fwd<256, PAD>(x, logw, RB, stw);
    float2* out = spec + ((size_t)t * h + row0) * w;
    for (int i = threadIdx.x; i < RB * w; i += 256) out[i] = x[xp<PAD>(i)];
}

// stage 2: grid (w/(2 CB), tiles).  Forward column transforms of the joint spectrum Z, the product
// A conj(B) of the two spectra it carries, inverse column transforms, fused.
//   A conj(B) [k] = Im(Z[k] Z[-k]) / 2 + i (|Z[k]|^2 - |Z[-k]|^2) / 4        (-k: both indices negated)
// so a workgroup needs the columns of frequency kx together with those of -kx.  Column positions are
// bit-reversed (position p holds kx = rev(p)): an aligned block of CB positions holds the
// frequencies kxbase + (jr << sh), jr = rev(j), sh = log2(w / CB), kxbase = rev(block index); its
// partner frequencies -kx lie in the block with kxbase' = 2^sh - kxbase at position j' = CB-1-j
// (kxbase = 0 is its own partner with j' = rev((CB - jr) mod CB), kxbase = 2^(sh-1) with j' = CB-1-j).
// A workgroup therefore takes two blocks: {b, 2^sh - b}, or the two self-paired ones {0, 2^(sh-1)}.
// LDS: X[2 CB][h] (column-major so that every column is one contiguous sequence); rows come out of
// the forward transform bit-reversed as well: row position r holds ky = rev(r), partner rev(-ky).
template <bool PAD>
__global__ __launch_bounds__(512) void xc_cols(int w, int h, int logw, int logh, int CB, float scale2,
                                               const float2* __restrict__ tw, float2* __restrict__ spec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* A = (float2*)smem_raw;
    const int nB = CB * h;
    const int t = blockIdx.y, g = blockIdx.x;
    const int logcb = __ffs(CB) - 1, sh = logw - logcb;                 // sh >= 1: the host keeps CB <= w / 2
    const int nblk = 1 << sh;
    const int bq[2] = { g == 0 ? 0 : g, g == 0 ? nblk >> 1 : nblk - g };
    int c0q[2];
#pragma unroll
    for (int q = 0; q < 2; q++) c0q[q] = (sh ? (int)(__brev((unsigned)bq[q]) >> (32 - sh)) : 0) << logcb;
    float2* sa = spec + (size_t)t * h * w;
    if (CB >= 2) {
        // 16-byte loads (two adjacent columns of a row), four per block in flight per thread
        const int hc = CB >> 1, lhc = __ffs(hc) - 1, n4 = hc * h;
        for (int base = 0; base < n4; base += 512 * 4) {
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = min(base + (int)threadIdx.x + 512 * u, n4 - 1);
                const size_t o = (size_t)(i >> lhc) * w + 2 * (i & (hc - 1));
                va[u] = *(const float4*)(sa + o + c0q[0]);
                vb[u] = *(const float4*)(sa + o + c0q[1]);
            }
            pin_loaded(va); pin_loaded(vb);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + (int)threadIdx.x + 512 * u;
                if (i < n4) {
                    const int r = i >> lhc, c = 2 * (i & (hc - 1));
                    A[xp<PAD>(c * h + r)] = make_float2(va[u].x, va[u].y); A[xp<PAD>((c + 1) * h + r)] = make_float2(va[u].z, va[u].w);
                    A[xp<PAD>(nB + c * h + r)] = make_float2(vb[u].x, vb[u].y); A[xp<PAD>(nB + (c + 1) * h + r)] = make_float2(vb[u].z, vb[u].w);
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < h; i += 512) {
            A[xp<PAD>(i)] = sa[(size_t)i * w + c0q[0]];
            A[xp<PAD>(nB + i)] = sa[(size_t)i * w + c0q[1]];
        }
    }
    float2* stw = A + (((size_t)xp<PAD>(2 * nB) + 1) & ~(size_t)1);
    stage_contiguous<512, 1>(stw, tw, h / 2);
    __syncthreads();
    fft_dif_fwd<512, PAD>(A, logh, 2 * CB, stw);
    // Product.  Every pair (k, -k) is handled once, by the thread of its element in the first block
    // (two different blocks), or of its lower index (a self-paired block): P[-k] = conj(P[k]).
    // runFFT divides each spectrum by sqrt(N) (image3.cpp:331-334): scale2 = 1 / N; product image3.cpp:392-394.
    const int ntot = (g == 0 ? 2 : 1) * nB;
    for (int e = threadIdx.x; e < ntot; e += 512) {
        const int q = e >= nB ? 1 : 0, ee = e - q * nB;
        const int j = ee >> logh, r = ee & (h - 1);
        const int rp = logh ? (int)(__brev((unsigned)((h - (int)(__brev((unsigned)r) >> (32 - logh))) & (h - 1))) >> (32 - logh)) : 0;
        int qp, jp;
        if (g != 0) { qp = 1; jp = CB - 1 - j; }
        else if (q == 1) { qp = 1; jp = CB - 1 - j; }
        else {
            qp = 0;
            const int jr = logcb ? (int)(__brev((unsigned)j) >> (32 - logcb)) : 0;
            jp = logcb ? (int)(__brev((unsigned)((CB - jr) & (CB - 1))) >> (32 - logcb)) : 0;
        }
        const int ep = qp * nB + jp * h + rp;
        if (g == 0 && ep < e) continue;                      // the partner's thread writes both
        const float2 zk = A[xp<PAD>(e)], zm = A[xp<PAD>(ep)];
        const float re = 0.5f * (zk.x * zm.y + zk.y * zm.x) * scale2;
        const float im = 0.25f * ((zk.x * zk.x + zk.y * zk.y) - (zm.x * zm.x + zm.y * zm.y)) * scale2;
        A[xp<PAD>(e)] = make_float2(re, im);
        if (ep != e) A[xp<PAD>(ep)] = make_float2(re, -im);
    }
    __syncthreads();
    if (g != 0) {
        // The correlation is real, so its spectrum is Hermitian, P[-ky,-kx] = conj(P[ky,kx]), and the inverse column
        // transform of a column -kx is the conjugate of column kx's, row by row: only the first block is
        // transformed, the partner block (column j <-> CB-1-j) is written as its conjugate.
        fft_dit_inv<512, PAD>(A, logh, CB, stw);
        for (int i = threadIdx.x; i < CB * h; i += 512) {
            const int r = i >> logcb, c = i & (CB - 1);
            const float2 v = A[xp<PAD>(c * h + r)];
            sa[(size_t)r * w + c0q[0] + c] = v;
            sa[(size_t)r * w + c0q[1] + (CB - 1 - c)] = make_float2(v.x, -v.y);
        }
        return;
    }
    fft_dit_inv<512, PAD>(A, logh, 2 * CB, stw);      // the two self-paired blocks
    for (int i = threadIdx.x; i < 2 * CB * h; i += 512) {
        const int q = i >= nB ? 1 : 0, ii = i - q * nB;
        const int r = ii >> logcb, c = ii & (CB - 1);
        sa[(size_t)r * w + c0q[q] + c] = A[xp<PAD>(q * nB + c * h + r)];
    }
}

__device__ __forceinline__ float wave_min_f(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Per-tile extrema and argmax keys are combined with atomics on one address per tile.  Same-address
// atomics are serialised in the memory system (of the order of 10 ns each) and the workgroups of a
// tile run at the same time, so a tile's atomics form a chain: with one atomic per wavefront that
// chain was 60 % of the inverse row kernel's time.  The wavefronts of a workgroup are therefore
// combined through LDS first.  (Testing the current value with a plain load before the atomic
// costs more than it saves: that load is a full memory round trip in front of every atomic.)
// tile extrema: the workgroup's wavefronts are combined through LDS, then one atomic pair
__device__ __forceinline__ void tile_commit_minmax(unsigned* minmax, int t, float lo, float hi)   // wave-reduced lo, hi
{
    __shared__ float s_lo[16], s_hi[16];
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; i++) { lo = fminf(lo, s_lo[i]); hi = fmaxf(hi, s_hi[i]); }
        if (lo <= hi) {
            atomicMin(&minmax[2 * t], __float_as_uint(lo));
            atomicMax(&minmax[2 * t + 1], __float_as_uint(hi));
        }
    }
}
__device__ __forceinline__ void tile_commit_key(unsigned long long* keys, int t, unsigned long long best)   // wave-reduced
{
    __shared__ unsigned long long s_k[16];
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_k[wv] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; i++) best = s_k[i] > best ? s_k[i] : best;
        if (best) atomicMax(&keys[t], best);
    }
}

// stage 3: grid (rows/RB, tiles).  Inverse row transform, divide by sqrt(N) (image3.cpp:401-405),
// magnitude, tile minimum and maximum (image3.cpp:402-411).  minmax[tile] = {min bits, max bits}
// of non-negative floats, whose unsigned bit patterns order like the values.
// The correlation of two real rasters is real: after the inverse column transforms every row's
// spectrum is Hermitian, so two rows ride through ONE complex inverse transform, z = Q[y1] + i Q[y2],
// whose real part is row y1 and whose imaginary part is row y2 (the reference's own imaginary parts
// are rounding noise around 1e-7 of the peak, far below the tolerances of the surface).  RB odd
// (RB == 1: rows of 2048 and more) keeps one row per transform and takes the complex magnitude.
template <bool PAD>
__global__ __launch_bounds__(256) void xc_rows_inv_abs(int w, int h, int logw, int RB, float scale,
                                                       const float2* __restrict__ tw,
                                                       const float2* __restrict__ spec,
                                                       float* __restrict__ surf, unsigned* __restrict__ minmax)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* x = (float2*)smem_raw;
    const int t = blockIdx.y, row0 = blockIdx.x * RB;
    const float2* in = spec + ((size_t)t * h + row0) * w;
    float* out = surf + ((size_t)t * h + row0) * w;
    float lo = INFINITY, hi = 0.f;
    if (RB & 1) {
        stage_contiguous<256, 4, PAD>(x, in, RB * w);
        float2* stw = x + (((size_t)xp<PAD>(RB * w) + 1) & ~(size_t)1);
        stage_contiguous<256, 2>(stw, tw, w / 2);
        __syncthreads();
        fft_dit_inv<256, PAD>(x, logw, RB, stw);
        for (int i = threadIdx.x; i < RB * w; i += 256) {
            const float2 v = x[xp<PAD>(i)];
            // cabs() of the complex float promoted to double (image3.cpp:407-410)
            const float m = (float)hypot((double)(v.x * scale), (double)(v.y * scale));
            out[i] = m;
            lo = fminf(lo, m); hi = fmaxf(hi, m);
        }
    } else {
        // pairs of rows (2p, 2p+1) -> sequence p; 16-byte loads, four pairs in flight per thread
        const int half = RB >> 1, n4 = half * (w >> 1);                 // float4 = two complex values
        const float4* in4 = (const float4*)in;
        for (int base = 0; base < n4; base += 256 * 4) {
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = min(base + (int)threadIdx.x + 256 * u, n4 - 1);
                const int p = i >> (logw - 1), c = i & ((w >> 1) - 1);       // w is a power of two here
                va[u] = in4[(size_t)(2 * p) * (w >> 1) + c];
                vb[u] = in4[(size_t)(2 * p + 1) * (w >> 1) + c];
            }
            pin_loaded(va); pin_loaded(vb);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + (int)threadIdx.x + 256 * u;
                if (i < n4) {
                    const int p = i >> (logw - 1), c = i & ((w >> 1) - 1);       // w is a power of two here
                    x[xp<PAD>(p * w + 2 * c)] = make_float2(va[u].x - vb[u].y, va[u].y + vb[u].x);
                    x[xp<PAD>(p * w + 2 * c + 1)] = make_float2(va[u].z - vb[u].w, va[u].w + vb[u].z);
                }
            }
        }
        float2* stw = x + (((size_t)xp<PAD>(RB * w) + 1) & ~(size_t)1);
        stage_contiguous<256, 2>(stw, tw, w / 2);
        __syncthreads();
        fft_dit_inv<256, PAD>(x, logw, half, stw);
        for (int i = threadIdx.x; i < half * w; i += 256) {
            const int p = i >> logw, c = i & (w - 1);
            const float2 v = x[xp<PAD>(i)];
            const float m1 = fabsf(v.x * scale), m2 = fabsf(v.y * scale);
            out[(size_t)(2 * p) * w + c] = m1;
            out[(size_t)(2 * p + 1) * w + c] = m2;
            lo = fminf(lo, fminf(m1, m2)); hi = fmaxf(hi, fmaxf(m1, m2));
        }
    }
    lo = wave_min_f(lo); hi = wave_max_f(hi);
    tile_commit_minmax(minmax, t, lo, hi);
}

// OpenCV borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// stage 4: grid (h, tiles): one row per workgroup of NT threads (four outputs per thread: NT = w / 4 up to 256, so
// that no wavefront idles through the filter loop).  out = (|c| - min) / (max - min)
// (image3.cpp:414-416, evaluated in double and narrowed), columns >= 2*(w/2) and rows >= 2*(h/2)
// zeroed for the blur input (image3.cpp:471-478); row pass of the Gaussian, taps accumulated left
// to right with separate multiply and add.
template <int NT>
__global__ __launch_bounds__(NT) void xc_norm_rowblur(int w, int h, const unsigned* __restrict__ minmax,
                                                       const float* __restrict__ gauss,
                                                       float* __restrict__ surf, float* __restrict__ tmp, int pre)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* g = (float*)smem_raw;                    // 151 taps (152 slots)
    float* prow = g + 152;                          // w + 150 reflected samples, 16-byte aligned, zero padded to w + 160
    float* nrow = prow + ((w + 163) & ~3);          // w normalised samples
    const int t = blockIdx.y, y = blockIdx.x;
    const float mn = __uint_as_float(minmax[2 * t]), mx = __uint_as_float(minmax[2 * t + 1]);
    const float range = mx - mn;
    float* srow = surf + ((size_t)t * h + y) * w;
    const int w2 = (w / 2) * 2, h2 = (h / 2) * 2;
    for (int i = threadIdx.x; i < XC_GAUSS_K; i += NT) g[i] = gauss[i];
    for (int x = threadIdx.x; x < w; x += NT) {
        const float m = srow[x];
        // pre: the surface already holds the product of the normalised band surfaces
        const float o = pre ? m : (float)(((double)m - (double)mn) / (double)range);
        srow[x] = o;                                 // the normalised surface, kept for stage 5
        nrow[x] = o;
    }
    __syncthreads();
    const int kw = (w == 1) ? 1 : XC_GAUSS_K, rx = kw / 2;
    for (int i = threadIdx.x; i < ((w + 163) & ~3); i += NT) {
        float v = 0.f;
        if (i < w + kw - 1) {
            const int sx = reflect101(i - rx, w);
            v = (sx < w2 && y < h2) ? nrow[sx] : 0.f;
        }
        prow[i] = v;
    }
    __syncthreads();
    float* trow = tmp + ((size_t)t * h + y) * w;
    if (kw == 1) {
        for (int x = threadIdx.x; x < w; x += NT) trow[x] = prow[x];
        return;
    }
    // four adjacent outputs per thread: the taps slide through registers, so every padded sample and
    // every coefficient is read from LDS once per four outputs (16-byte reads); each output still
    // accumulates its taps left to right with separate multiply and add
    for (int x0 = 4 * threadIdx.x; x0 < w; x0 += 4 * NT) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        float4 v = *(const float4*)(prow + x0);
#pragma unroll 2
        for (int k = 0; k + 4 <= XC_GAUSS_K; k += 4) {
            const float4 n = *(const float4*)(prow + x0 + k + 4);
            const float4 c = *(const float4*)(g + k);
            a0 = __fadd_rn(a0, __fmul_rn(v.x, c.x)); a1 = __fadd_rn(a1, __fmul_rn(v.y, c.x));
            a2 = __fadd_rn(a2, __fmul_rn(v.z, c.x)); a3 = __fadd_rn(a3, __fmul_rn(v.w, c.x));
            a0 = __fadd_rn(a0, __fmul_rn(v.y, c.y)); a1 = __fadd_rn(a1, __fmul_rn(v.z, c.y));
            a2 = __fadd_rn(a2, __fmul_rn(v.w, c.y)); a3 = __fadd_rn(a3, __fmul_rn(n.x, c.y));
            a0 = __fadd_rn(a0, __fmul_rn(v.z, c.z)); a1 = __fadd_rn(a1, __fmul_rn(v.w, c.z));
            a2 = __fadd_rn(a2, __fmul_rn(n.x, c.z)); a3 = __fadd_rn(a3, __fmul_rn(n.y, c.z));
            a0 = __fadd_rn(a0, __fmul_rn(v.w, c.w)); a1 = __fadd_rn(a1, __fmul_rn(n.x, c.w));
            a2 = __fadd_rn(a2, __fmul_rn(n.y, c.w)); a3 = __fadd_rn(a3, __fmul_rn(n.z, c.w));
            v = n;
        }
        {   // taps 148, 149, 150 (151 = 37 * 4 + 3); v holds samples x0+148 .. x0+151
            const float4 n = *(const float4*)(prow + x0 + 152);
            const float c0 = g[148], c1 = g[149], c2 = g[150];
            a0 = __fadd_rn(a0, __fmul_rn(v.x, c0)); a1 = __fadd_rn(a1, __fmul_rn(v.y, c0));
            a2 = __fadd_rn(a2, __fmul_rn(v.z, c0)); a3 = __fadd_rn(a3, __fmul_rn(v.w, c0));
            a0 = __fadd_rn(a0, __fmul_rn(v.y, c1)); a1 = __fadd_rn(a1, __fmul_rn(v.z, c1));
            a2 = __fadd_rn(a2, __fmul_rn(v.w, c1)); a3 = __fadd_rn(a3, __fmul_rn(n.x, c1));
            a0 = __fadd_rn(a0, __fmul_rn(v.z, c2)); a1 = __fadd_rn(a1, __fmul_rn(v.w, c2));
            a2 = __fadd_rn(a2, __fmul_rn(n.x, c2)); a3 = __fadd_rn(a3, __fmul_rn(n.y, c2));
        }
        if (x0 < w) trow[x0] = a0;
        if (x0 + 1 < w) trow[x0 + 1] = a1;
        if (x0 + 2 < w) trow[x0 + 2] = a2;
        if (x0 + 3 < w) trow[x0 + 3] = a3;
    }
}

__device__ __forceinline__ unsigned long long argmax_key(float v, unsigned idx)
{
    unsigned b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // order-preserving map to unsigned
    return ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// stage 5: grid (w/64, h/RB5, tiles), 256 threads = 64 columns x 4 row groups.  Column pass of
// the Gaussian (centre tap first, then symmetric pairs outward), DoG = out - blur
// (image3.cpp:483), first strict maximum in row-major order above the initial -1
// (image3.cpp:486-501) as a 64-bit key: ordered value bits, then inverted linear index.
#define XC_RB5 32
__global__ __launch_bounds__(256) void xc_colblur_argmax(int w, int h, const float* __restrict__ gauss,
                                                         const float* __restrict__ surf,
                                                         const float* __restrict__ tmp,
                                                         float* __restrict__ dog_out,
                                                         unsigned long long* __restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float col[(((XC_RB5 + 2 * XC_GAUSS_R) * 16 + 255) / 256) * 1024];
    __shared__ float g[XC_GAUSS_K];
    const int t = blockIdx.z, x0 = blockIdx.x * 64, y0 = blockIdx.y * XC_RB5;
    const int kh = (h == 1) ? 1 : XC_GAUSS_K, ry = kh / 2;
    const int w2 = (w / 2) * 2, h2 = (h / 2) * 2;
    const float* tt = tmp + (size_t)t * h * w;
    for (int i = threadIdx.x; i < XC_GAUSS_K; i += 256) g[i] = gauss[i];
    const int nrows = XC_RB5 + 2 * ry;
    if ((w & 3) == 0 && x0 + 64 <= w && h >= 256) {
        // 16-byte loads, all of a thread's requests in flight before the first is stored.  The row
        // index is reflected without a loop (one reflection suffices for h >= 256) and clamped rather
        // than tested: any branch between two loads makes the compiler wait for the first one, and
        // the staging then is a chain of memory round trips several times longer than the arithmetic.
        constexpr int NV = ((XC_RB5 + 2 * XC_GAUSS_R) * 16 + 255) / 256;
        float4 v[NV];
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const int i = threadIdx.x + 256 * q, r = min(i >> 4, nrows - 1), c4 = i & 15;
            int sy = y0 - ry + r;
            sy = sy < 0 ? -sy : sy;
            sy = sy >= h ? 2 * h - 2 - sy : sy;
            v[q] = *(const float4*)(tt + (size_t)sy * w + x0 + 4 * c4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NV; q++)                 // unconditional (col is padded to NV * 256 float4): a test here pulls the load back in
            *(float4*)(col + 4 * (threadIdx.x + 256 * q)) = v[q];
    } else {
        for (int i = threadIdx.x; i < nrows * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            const int sy = reflect101(y0 - ry + r, h);
            col[i] = (x0 + c < w) ? tt[(size_t)sy * w + x0 + c] : 0.f;
        }
    }
    __syncthreads();
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int x = x0 + c;
    unsigned long long best = 0;
    // eight consecutive rows per thread: the symmetric tap pairs slide through registers (one new
    // sample on each side per tap for all eight rows); every output still starts from the centre tap
    // and adds the pairs outward with separate multiply and add
    constexpr int NR = XC_RB5 / 4;
    const int j0 = rg * NR;
    // the thread's samples of the normalised surface, requested now (clamped coordinates, tested at
    // the end) so that they travel during the blur instead of one exposed round trip each after it
    float sf[NR];
#pragma unroll
    for (int i = 0; i < NR; i++)
        sf[i] = surf[(size_t)t * h * w + (size_t)min(y0 + j0 + i, h - 1) * w + min(x, w - 1)];
    float acc[NR];
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < NR; i++) acc[i] = col[(j0 + i) * 64 + c];
    } else {
        float up[NR], dn[NR];            // samples at row + k and row - k
        const float gc = g[XC_GAUSS_R];
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const float ctr = col[(j0 + i + ry) * 64 + c];
            acc[i] = __fmul_rn(gc, ctr);
            up[i] = ctr; dn[i] = ctr;
        }
#pragma unroll
        for (int k = 1; k <= XC_GAUSS_R; k++) {      // fully unrolled: the window shifts become register renaming
            // shift the windows: up[i] <- sample(row_i + k), dn[i] <- sample(row_i - k)
#pragma unroll
            for (int i = 0; i < NR - 1; i++) up[i] = up[i + 1];
            up[NR - 1] = col[(j0 + NR - 1 + ry + k) * 64 + c];
#pragma unroll
            for (int i = NR - 1; i > 0; i--) dn[i] = dn[i - 1];
            dn[0] = col[(j0 + ry - k) * 64 + c];
            const float gk = g[XC_GAUSS_R + k];
#pragma unroll
            for (int i = 0; i < NR; i++) acc[i] = __fadd_rn(acc[i], __fmul_rn(gk, __fadd_rn(up[i], dn[i])));
        }
    }
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const int y = y0 + j0 + i;
        if (x >= w || y >= h) continue;
        const size_t o = (size_t)t * h * w + (size_t)y * w + x;
        const float imout = (x < w2 && y < h2) ? sf[i] : 0.f;
        const float dog = __fsub_rn(imout, acc[i]);
        if (dog_out) dog_out[o] = dog;
        if (x < w2 && y < h2 && dog > -1.0f) {
            const unsigned long long k = argmax_key(dog, (unsigned)(y * w + x));
            best = k > best ? k : best;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    tile_commit_key(keys, t, best);
}

// stage 6: one thread per tile (image3.cpp:507-516)
__global__ void xc_finish(int n, int w, int h, const unsigned long long* __restrict__ keys,
                          const unsigned* __restrict__ minmax, const float* __restrict__ surf,
                          mis_offset* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    mis_offset r;
    const unsigned long long k = keys[t];
    const bool degenerate = minmax[2 * t] == minmax[2 * t + 1];       // max == min: 0/0 surface
    if (k == 0 || degenerate) {
        r.dx = 0; r.dy = 0; r.peak = -1.f; r.dog_peak = -1.f;          // nothing beat the initial -1
    } else {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu);
        unsigned b = (unsigned)(k >> 32);
        b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
        int col = (int)(idx % (unsigned)w), row = (int)(idx / (unsigned)w);
        r.peak = surf[(size_t)t * h * w + idx];
        r.dog_peak = __uint_as_float(b);
        if (col > w / 2) col -= w;
        if (row > h / 2) row -= h;
        r.dx = col; r.dy = row;
    }
    out[t] = r;
}

__global__ void xc_init_meta(int n, unsigned* minmax, unsigned long long* keys)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    minmax[2 * t] = 0x7F800000u;   // +inf
    minmax[2 * t + 1] = 0u;
    keys[t] = 0ull;
}

// ---------------------------------------------------------------------------------------------
// General sizes (the first level of image3.cpp's recursion correlates the two whole rasters, whose
// sides are whatever the photographs are; FFTW takes any length, image3.cpp:365-372).
// A 1-D DFT of a length n that is not a power of two is computed as a chirp-z (Bluestein)
// convolution of length M = 2^k >= 2n-1 in LDS:
//     X[k] = c[k] * sum_j (x[j] c[j]) conj(c[k-j]),   c[j] = exp(-i pi j^2 / n)
// with the transform of the conj(c) filter precomputed in double (stored in the bit-reversed order
// the DIF transform produces).  The 2-D transform is rows, transpose, rows, so that every 1-D
// transform reads a contiguous sequence.  Slower than the power-of-two path (two transposes, natural
// order throughout); it runs once per pair.  Lengths whose prime factors are all <= 13 skip the chirp:
// a mixed-radix transform of the length itself (xcorr_smooth_dft.h).
// ---------------------------------------------------------------------------------------------
#include "xcorr_smooth_dft.h"
struct XgDft { int n, M, logM, pow2, smooth; const float2* chirp; const float2* filt; const float2* tw; XsPlan xs; };

// In-place DFT of sequences of length d.n.  Sequence r lives at data + (r / per_batch) * batch_stride
// + (r % per_batch) * n.  inverse != 0: unnormalised backward transform (conjugate, forward, conjugate).
template <int NT, bool PAD>
__global__ __launch_bounds__(NT) void xg_dft_rows(float2* __restrict__ data, XgDft d, int per_batch,
                                                   size_t batch_stride, int inverse)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* x = (float2*)smem_raw;
    const int r = blockIdx.x;
    float2* row = data + (size_t)(r / per_batch) * batch_stride + (size_t)(r % per_batch) * d.n;
    const float sgn = inverse ? -1.f : 1.f;
    if (d.pow2) {
        for (int i = threadIdx.x; i < d.n; i += NT) { float2 v = row[i]; v.y *= sgn; x[xp<PAD>(i)] = v; }
        __syncthreads();
        fft_dif_fwd<NT, PAD>(x, d.logM, 1, d.tw);
        for (int i = threadIdx.x; i < d.n; i += NT) {
            float2 v = x[xp<PAD>(i)]; v.y *= sgn;
            row[d.logM ? bitrev(i, d.logM) : 0] = v;
        }
        return;
    }
    // Global operands are fetched U at a time with clamped indices and tested afterwards: a workgroup
    // owns a CU's whole LDS here, so nothing else hides a load that is waited for on the spot.
    constexpr int U = 8;
    for (int base = threadIdx.x; base < d.M; base += NT * U) {
        float2 v[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(base + NT * u, d.n - 1);
            v[u] = row[i]; c[u] = d.chirp[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(c[u].x), "+v"(c[u].y));
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + NT * u;
            if (i < d.M) {
                float2 t = v[u]; t.y *= sgn;
                x[xp<PAD>(i)] = i < d.n ? cmul(t, c[u]) : make_float2(0.f, 0.f);
            }
        }
    }
    __syncthreads();
    fft_dif_fwd<NT, PAD>(x, d.logM, 1, d.tw);
    for (int base = threadIdx.x; base < d.M; base += NT * U) {
        float2 f[U];
#pragma unroll
        for (int u = 0; u < U; u++) f[u] = d.filt[min(base + NT * u, d.M - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" : "+v"(f[u].x), "+v"(f[u].y));
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + NT * u;
            if (i < d.M) x[xp<PAD>(i)] = cmul(x[xp<PAD>(i)], f[u]);
        }
    }
    __syncthreads();
    fft_dit_inv<NT, PAD>(x, d.logM, 1, d.tw);
    const float inv_m = 1.0f / (float)d.M;
    for (int base = threadIdx.x; base < d.n; base += NT * U) {
        float2 c[U];
#pragma unroll
        for (int u = 0; u < U; u++) c[u] = d.chirp[min(base + NT * u, d.n - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" : "+v"(c[u].x), "+v"(c[u].y));
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + NT * u;
            if (i < d.n) {
                const float2 v = cmul(x[xp<PAD>(i)], c[u]);
                row[i] = make_float2(v.x * inv_m, sgn * v.y * inv_m);
            }
        }
    }
}

// out[b][c][r] = in[b][r][c]; grid (ceil(cols/32), ceil(rows/32), batches), 256 threads
__global__ __launch_bounds__(256) void xg_transpose(const float2* __restrict__ in, float2* __restrict__ out,
                                                    int rows, int cols, size_t in_stride, size_t out_stride)
{
    __shared__ float2 tile[32][33];
    const float2* src = in + (size_t)blockIdx.z * in_stride;
    float2* dst = out + (size_t)blockIdx.z * out_stride;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = src[(size_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) dst[(size_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

// both tiles as ONE complex raster z = a + i b, zero outside the bounding boxes (image3.cpp:322-328); tile t at spec + t * 2N
__global__ __launch_bounds__(256) void xg_load_tiles(XcImage im, const mis_rect* __restrict__ r1,
                                                     const mis_rect* __restrict__ r2, int tile0, int w, int h,
                                                     float2* __restrict__ spec)
{
    const int t = blockIdx.z, ry = blockIdx.y;
    const int rx = blockIdx.x * 256 + threadIdx.x;
    if (rx >= w) return;
    float v[2];
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const mis_rect bb = side ? r2[tile0 + t] : r1[tile0 + t];
        v[side] = 0.f;
        if (rx < bb.width && ry < bb.height) {
            const int sx = bb.minx + rx, sy = bb.miny + ry;
            if (sx >= 0 && sx < im.iw && sy >= 0 && sy < im.ih) v[side] = load_px(im.img[side], im.dtype, im.stride, sx, sy);
        }
    }
    spec[((size_t)t * 2 * h + ry) * w + rx] = make_float2(v[0], v[1]);
}

// (A s) conj(B s) of the two real tiles (image3.cpp:331-334, 392-394) from the transform Z of a + i b, natural order,
// stored transposed (rows = kx): A[k] = (Z[k] + conj(Z[-k])) / 2, B[k] = (Z[k] - conj(Z[-k])) / (2i).  Reads the first
// plane of a tile (spec + t * 2N) and writes the second (every element is also some other element's mirror).
__global__ __launch_bounds__(256) void xg_product(float2* __restrict__ spec, int w, int h, float scale)
{
    const int ky = blockIdx.x * 256 + threadIdx.x, kx = blockIdx.y;
    if (ky >= h) return;
    const size_t N = (size_t)w * h;
    float2* sz = spec + (size_t)blockIdx.z * 2 * N;
    const int mx = kx ? w - kx : 0, my = ky ? h - ky : 0;
    const float2 zk = sz[(size_t)kx * h + ky], zm = sz[(size_t)mx * h + my];
    const float hs = 0.5f * scale;
    const float2 a = make_float2((zk.x + zm.x) * hs, (zk.y - zm.y) * hs);
    const float2 b = make_float2((zk.y + zm.y) * hs, (zm.x - zk.x) * hs);
    sz[N + (size_t)kx * h + ky] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// |c| / sqrt(N) and the tile extrema (image3.cpp:401-411), as in xc_rows_inv_abs
#define XG_ABS_PER_THREAD 16     // elements per thread: fewer workgroups, fewer same-address atomics per tile
__global__ __launch_bounds__(256) void xg_abs_minmax(const float2* __restrict__ spec, size_t N, float scale,
                                                     float* __restrict__ surf, unsigned* __restrict__ minmax)
{
    const size_t i0 = (size_t)blockIdx.x * (256 * XG_ABS_PER_THREAD) + threadIdx.x;
    const int t = blockIdx.y;
    const float2* sp = spec + (size_t)t * 2 * N;
    float lo = INFINITY, hi = 0.f;
    float2 v[XG_ABS_PER_THREAD];
#pragma unroll
    for (int k = 0; k < XG_ABS_PER_THREAD; k++) {      // clamped addresses: every load unconditional and in flight together
        const size_t i = i0 + (size_t)256 * k;
        v[k] = sp[i < N ? i : N - 1];
    }
#pragma unroll
    for (int k = 0; k < XG_ABS_PER_THREAD; k++) {
        const size_t i = i0 + (size_t)256 * k;
        const float m = (float)hypot((double)(v[k].x * scale), (double)(v[k].y * scale));
        if (i < N) {
            surf[(size_t)t * N + i] = m;
            lo = fminf(lo, m); hi = fmaxf(hi, m);
        }
    }
    lo = wave_min_f(lo); hi = wave_max_f(hi);
    tile_commit_minmax(minmax, t, lo, hi);
}

// raster of any pixel type -> complex (image-fft.cpp:66-71)
__global__ __launch_bounds__(256) void xg_load_raster(const void* __restrict__ in, int dtype, size_t n,
                                                      float2* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v;
    if (dtype == MIS_U8) v = (float)((const uint8_t*)in)[i];
    else if (dtype == MIS_U16) v = (float)((const uint16_t*)in)[i];
    else v = ((const float*)in)[i];
    out[i] = make_float2(v, 0.f);
}

// division by sqrt(N) held in double (image-fft.cpp:84-87)
__global__ __launch_bounds__(256) void xg_divide(float2* __restrict__ x, size_t n, double norm)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 v = x[i];
    x[i] = make_float2((float)((double)v.x / norm), (float)((double)v.y / norm));
}

// Band loop of image3.cpp:382-418 with n_bands not forced to 1: prod = prod * norm(band), the
// product evaluated in double and narrowed (image3.cpp:414-416); first != 0 starts from 1.0.
__global__ __launch_bounds__(256) void xc_band_product(size_t N, const unsigned* __restrict__ minmax,
                                                       const float* __restrict__ surf, float* __restrict__ prod, int first)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= N) return;
    const float mn = __uint_as_float(minmax[2 * t]), mx = __uint_as_float(minmax[2 * t + 1]);
    const double v = ((double)surf[(size_t)t * N + i] - (double)mn) / (double)(mx - mn);   // 0/0 -> NaN
    const float prev = first ? 1.0f : prod[(size_t)t * N + i];
    prod[(size_t)t * N + i] = (float)((double)prev * v);
}

__global__ void xc_reset_minmax(int n, unsigned* minmax)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    minmax[2 * t] = 0x7F800000u;
    minmax[2 * t + 1] = 0u;
}

// findCPPMax (image2.cpp:216-308, image-find-cpp.cpp:68-158): band product (left to right, float),
// first strict maximum above -1 in row-major order outside the central mask rectangle.
__global__ __launch_bounds__(256) void xc_masked_peak(const float* __restrict__ bands, int cols, int rows, int nb,
                                                      int rows2, int cols2, int r0, int r1, int c0, int c1,
                                                      unsigned long long* __restrict__ key)
{
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    unsigned long long best = 0;
    if (col < cols2 && row < rows2 && !(row >= r0 && row < r1 && col >= c0 && col < c1)) {
        const size_t px = (size_t)rows * cols, o = (size_t)row * cols + col;
        float v = bands[o];
        for (int b = 1; b < nb; b++) v = __fmul_rn(v, bands[(size_t)b * px + o]);
        if (v > -1.0f) best = argmax_key(v, (unsigned)o);
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(key, best);
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct XcPlan { int w, h, logw, logh, RBw, CB; bool general; };

// descriptor of a 1-D DFT of any length 1..16384 (powers of two) or 1..8192 (other lengths)
static int get_dft(mis_ctx* ctx, int n, XgDft* d)
{
    XcorrWs* w = ctx->xc;
    d->n = n; d->chirp = nullptr; d->filt = nullptr; d->smooth = 0;
    if (is_pow2(n)) {
        d->pow2 = 1; d->M = n; d->logM = ilog2(n);
        return get_twiddle(ctx, n, &d->tw);
    }
    d->pow2 = 0;
    // MISTEREO_XG_CHIRP=1 keeps the chirp-z transform for every length (comparison runs)
    static const bool chirp_only = [] { const char* e = getenv("MISTEREO_XG_CHIRP"); return e && e[0] == '1'; }();
    if (!chirp_only && xs_factor(n, &d->xs)) {
        auto ci = w->circle.find(n);
        if (ci == w->circle.end()) {
            std::vector<float2> hc((size_t)n);
            for (int t = 0; t < n; t++) {
                const double a = -2.0 * M_PI * (double)t / (double)n;
                hc[t] = make_float2((float)cos(a), (float)sin(a));
            }
            float2* dc = nullptr;
            MIS_HIP(ctx, hipMalloc((void**)&dc, hc.size() * sizeof(float2)));
            MIS_HIP(ctx, hipMemcpyAsync(dc, hc.data(), hc.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
            MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ci = w->circle.emplace(n, dc).first;
        }
        d->smooth = 1; d->xs.circle = ci->second; d->M = n; d->logM = 0; d->tw = nullptr;
        return MIS_OK;
    }
    d->M = 1; while (d->M < 2 * n - 1) d->M <<= 1;
    d->logM = ilog2(d->M);
    int rc = get_twiddle(ctx, d->M, &d->tw); if (rc) return rc;
    auto it = w->chirp.find(n);
    if (it == w->chirp.end()) {
        const int M = d->M, logM = d->logM;
        std::vector<double> cr(n), ci(n), fr(M, 0.0), fi(M, 0.0);
        for (int j = 0; j < n; j++) {
            const long long q = ((long long)j * j) % (2LL * n);          // j^2 mod 2n keeps the angle small
            const double a = -M_PI * (double)q / (double)n;
            cr[j] = cos(a); ci[j] = sin(a);
        }
        for (int j = 0; j < n; j++) {                                     // filter conj(c), wrapped
            fr[j] = cr[j]; fi[j] = -ci[j];
            if (j) { fr[M - j] = cr[j]; fi[M - j] = -ci[j]; }
        }
        // forward transform of the filter in double: iterative radix-2, natural order out
        for (int i = 0, j = 0; i < M; i++) {
            if (i < j) { std::swap(fr[i], fr[j]); std::swap(fi[i], fi[j]); }
            int m = M >> 1;
            while (m >= 1 && (j & m)) { j ^= m; m >>= 1; }
            j |= m;
        }
        for (int len = 2; len <= M; len <<= 1) {
            const double ang = -2.0 * M_PI / (double)len;
            for (int i = 0; i < M; i += len)
                for (int k = 0; k < len / 2; k++) {
                    const double wr = cos(ang * k), wi = sin(ang * k);
                    const int a0 = i + k, a1 = a0 + len / 2;
                    const double tr = fr[a1] * wr - fi[a1] * wi, ti = fr[a1] * wi + fi[a1] * wr;
                    fr[a1] = fr[a0] - tr; fi[a1] = fi[a0] - ti;
                    fr[a0] += tr; fi[a0] += ti;
                }
        }
        std::vector<float2> hbuf((size_t)n + M);
        for (int j = 0; j < n; j++) hbuf[j] = make_float2((float)cr[j], (float)ci[j]);
        for (int pos = 0; pos < M; pos++) {                               // the DIF output order
            unsigned rv = 0;
            for (int b = 0; b < logM; b++) if (pos & (1 << b)) rv |= 1u << (logM - 1 - b);
            hbuf[(size_t)n + pos] = make_float2((float)fr[rv], (float)fi[rv]);
        }
        float2* dbuf = nullptr;
        MIS_HIP(ctx, hipMalloc((void**)&dbuf, hbuf.size() * sizeof(float2)));
        MIS_HIP(ctx, hipMemcpyAsync(dbuf, hbuf.data(), hbuf.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        it = w->chirp.emplace(n, dbuf).first;
    }
    d->chirp = it->second;
    d->filt = it->second + n;
    return MIS_OK;
}

static int launch_dft(mis_ctx* ctx, float2* data, const XgDft& d, int nseq, int per_batch, size_t batch_stride, int inverse)
{
    if (d.smooth) {
        const size_t shs = ((size_t)d.n + d.n / 8 + 1) * sizeof(float2);
        const dim3 g((unsigned)nseq);
#define XS_LAUNCH(NTV)                                                                                              \
    do {                                                                                                            \
        if (shs > 64 * 1024)                                                                                        \
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)xs_dft_rows<NTV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shs)); \
        hipLaunchKernelGGL((xs_dft_rows<NTV, true>), g, dim3(NTV), shs, ctx->stream, data, d.xs, per_batch, batch_stride, inverse);       \
    } while (0)
        if (d.n > 4096) XS_LAUNCH(1024); else if (d.n > 2048) XS_LAUNCH(512); else XS_LAUNCH(256);
#undef XS_LAUNCH
        return MIS_OK;
    }
    // padded LDS layout (see xp) whenever it fits
    const bool pad = (d.M % 8 == 0) && ((size_t)d.M + d.M / 8) * sizeof(float2) <= 160 * 1024;
    const size_t sh = ((size_t)d.M + (pad ? d.M / 8 : 0)) * sizeof(float2);
    const dim3 grid((unsigned)nseq);
#define XG_LAUNCH(NTV, PADV)                                                                                        \
    do {                                                                                                            \
        if (sh > 64 * 1024)                                                                                         \
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)xg_dft_rows<NTV, PADV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((xg_dft_rows<NTV, PADV>), grid, dim3(NTV), sh, ctx->stream, data, d, per_batch, batch_stride, inverse);       \
    } while (0)
    // a long sequence owns most of a CU's LDS, so its own workgroup has to bring the wavefronts that hide latency
    if (d.M >= 8192) { if (pad) XG_LAUNCH(1024, true); else XG_LAUNCH(1024, false); }
    else if (d.M >= 2048) { if (pad) XG_LAUNCH(512, true); else XG_LAUNCH(512, false); }
    else             { if (pad) XG_LAUNCH(256, true); else XG_LAUNCH(256, false); }
#undef XG_LAUNCH
    return MIS_OK;
}

static int make_plan(mis_ctx* ctx, int w, int h, XcPlan* p)
{
    p->w = w; p->h = h;
    p->general = !(is_pow2(w) && is_pow2(h) && w >= 16 && h >= 16 && w <= 8192 && h <= 8192);
    if (p->general) {
        const int lim_w = is_pow2(w) ? 16384 : 8192, lim_h = is_pow2(h) ? 16384 : 8192;
        if (w < 1 || h < 1 || w > lim_w || h > lim_h)
            return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "tile size %d x %d: sides up to 8192 (16384 for powers of two) are supported", w, h);
        p->logw = p->logh = 0; p->RBw = p->CB = 1;
        return MIS_OK;
    }
    p->logw = ilog2(w); p->logh = ilog2(h);
    p->RBw = w >= 2048 ? 1 : 2048 / w;          // rows per workgroup: 2048 complex values in LDS
    if (p->RBw > h) p->RBw = h;
    p->CB = h >= 4096 ? 1 : 4096 / h;           // columns per block, two blocks per workgroup: 2 x 4096 complex values
    if (p->CB > 16) p->CB = 16;
    if (p->CB > w / 2) p->CB = w / 2;           // a workgroup takes two blocks of columns (a frequency block and its negative)
    return MIS_OK;
}

// correlates tiles [tile0, tile0+nt) of the rect lists (device pointers); results to d_out[tile0..]
// peak_stages false: stop after stage 3 (|c| surface and its extrema)
#include "xcorr_blur_mfma.h"

// stages 4 and 5 of a batch: the Gaussian on the matrix cores where the tile shape allows it (MISTEREO_XC_MFMA=0
// keeps the vector kernels, for comparisons), else the vector kernels
static bool blur_on_matrix_cores(int w, int h)
{
    static const int mf_env = [] { const char* e = getenv("MISTEREO_XC_MFMA"); return e ? atoi(e) : 1; }();
    return mf_env != 0 && xb_fits(w, h);
}
// the matrix-core operand table, built on the context's stream before any stage that reads it is enqueued
// MISTEREO_XC_SMALL=1: tiles of 32 / 64 samples a side take the folded filter (xc_blur_small_mf) instead of the two general
// kernels.  OFF by default: its results for tiles 64 samples wide were not reproducible while other kernels ran beside it
// (see the kernel's header), and a result that depends on what else the device is doing is not a result.
static bool blur_small(int w, int h)
{
    static const int env = [] { const char* e = getenv("MISTEREO_XC_SMALL"); return e ? atoi(e) : 0; }();
    return env != 0 && blur_on_matrix_cores(w, h) && xb_small_fits(w, h);
}
// even sides the two LDS-resident kernels do not hold (2048 and up, or not multiples of 32) take the segmented pair;
// MISTEREO_XC_SEG=0 leaves them to the vector kernels
static bool blur_seg(int w, int h)
{
    static const int mf_env = [] { const char* e = getenv("MISTEREO_XC_MFMA"); return e ? atoi(e) : 1; }();
    static const int env = [] { const char* e = getenv("MISTEREO_XC_SEG"); return e ? atoi(e) : 1; }();
    // MISTEREO_XC_SEG_FROM=n: square-ish tiles of n and up take the segmented pair although the LDS-resident one holds them
    // (measured: no gain at 1024, 512, 256 or 128: the default keeps the LDS-resident kernels wherever they fit)
    static const int from = [] { const char* e = getenv("MISTEREO_XC_SEG_FROM"); return e ? atoi(e) : 2048; }();
    if (mf_env == 0 || env == 0 || !xb_seg_fits(w, h)) return false;
    return !xb_fits(w, h) || (w >= from && h >= from);
}
static int ensure_small_taps(mis_ctx* ctx, int L)
{
    XcorrWs* ws = ctx->xc;
    if (ws->taps_small.count(L)) return MIS_OK;
    float k[XC_GAUSS_K];
    MIS_HIP(ctx, hipMemcpyAsync(k, ws->gauss, sizeof k, hipMemcpyDeviceToHost, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<_Float16> host(xb_small_vectors(L) * 8);
    xb_build_small(k, L, host.data());
    void* d = nullptr;
    MIS_HIP(ctx, hipMalloc(&d, host.size() * sizeof(_Float16)));
    MIS_HIP(ctx, hipMemcpyAsync(d, host.data(), host.size() * sizeof(_Float16), hipMemcpyHostToDevice, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ws->taps_small[L] = d;
    return MIS_OK;
}
static int ensure_taps(mis_ctx* ctx, int w, int h)
{
    XcorrWs* ws = ctx->xc;
    if (blur_small(w, h)) {
        int rc = ensure_small_taps(ctx, w); if (rc) return rc;
        return ensure_small_taps(ctx, h);
    }
    if (!(blur_on_matrix_cores(w, h) || blur_seg(w, h)) || ws->taps_mf) return MIS_OK;
    MIS_HIP(ctx, hipMalloc(&ws->taps_mf, (size_t)2 * XB_KSTEPS * 64 * sizeof(xb_half8)));
    hipLaunchKernelGGL(xb_build_taps, dim3(1), dim3(64), 0, ctx->stream, ws->gauss, (xb_half8*)ws->taps_mf);
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}
// *norm_surf: the plane that holds the normalised surfaces afterwards (what xc_finish reads the peak from)
static int launch_blur_stages(mis_ctx* ctx, hipStream_t s, int w, int h, int nt, const unsigned* minmax, float* fsurf, float* tmp, bool pre,
                              float* dog_dbg, unsigned long long* keys, const float** norm_surf)
{
    XcorrWs* ws = ctx->xc;
    *norm_surf = fsurf;
    if (blur_small(w, h)) {
        const xb_half8* tw = (const xb_half8*)ws->taps_small[w];
        const xb_half8* th = (const xb_half8*)ws->taps_small[h];
        const int nbw = w / 32;
        const dim3 g((unsigned)(((size_t)nt * nbw + 3) / 4));
#define XB_SMALL(NBW, NBH) hipLaunchKernelGGL((xc_blur_small_mf<NBW, NBH>), g, dim3(256), 0, s, nt, minmax, tw, th, fsurf, tmp, dog_dbg, keys, pre ? 1 : 0)
        if (w == 32 && h == 32) XB_SMALL(1, 1); else if (w == 64 && h == 32) XB_SMALL(2, 1);
        else if (w == 32 && h == 64) XB_SMALL(1, 2); else XB_SMALL(2, 2);
#undef XB_SMALL
        if (!pre) *norm_surf = tmp;
        return MIS_OK;
    }
    if (blur_seg(w, h)) {
        const xb_half8* taps = (const xb_half8*)ws->taps_mf;
        const size_t sh = xb_seg_lds_bytes();
        hipLaunchKernelGGL(xc_rowblur_seg_mf, dim3((h + 31) / 32, nt, (w + XB_SEG - 1) / XB_SEG), dim3(512), sh, s,
                           w, h, minmax, taps, fsurf, tmp, pre ? 1 : 0);
        hipLaunchKernelGGL(xc_colblur_seg_mf, dim3((w + 31) / 32, nt, (h + XB_SEG - 1) / XB_SEG), dim3(512), sh, s,
                           w, h, minmax, taps, fsurf, tmp, dog_dbg, keys, pre ? 1 : 0);
        return MIS_OK;
    }
    if (blur_on_matrix_cores(w, h)) {
        const xb_half8* taps = (const xb_half8*)ws->taps_mf;
        const size_t sha = xb_lds_bytes(w), shb = xb_lds_bytes(h);
        if (sha > 64 * 1024)
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)xc_norm_rowblur_mf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha));
        if (shb > 64 * 1024)
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)xc_colblur_argmax_mf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        hipLaunchKernelGGL(xc_norm_rowblur_mf, dim3(h / 32, nt), dim3(512), sha, s, w, h, xb_line_stride(w), minmax, taps, fsurf, tmp, pre ? 1 : 0);
        hipLaunchKernelGGL(xc_colblur_argmax_mf, dim3(w / 32, nt), dim3(512), shb, s, w, h, xb_line_stride(h), taps, fsurf, tmp, dog_dbg, keys);
        return MIS_OK;
    }
    {
        const size_t sh = (size_t)(152 + 2 * w + 168) * sizeof(float);
        if (w >= 1024 || (w & 3)) {
            if (sh > 64 * 1024)
                MIS_HIP(ctx, hipFuncSetAttribute((const void*)xc_norm_rowblur<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
            hipLaunchKernelGGL(xc_norm_rowblur<256>, dim3(h, nt), dim3(256), sh, s, w, h, minmax, ws->gauss, fsurf, tmp, pre);
        } else if (w >= 512) {
            hipLaunchKernelGGL(xc_norm_rowblur<128>, dim3(h, nt), dim3(128), sh, s, w, h, minmax, ws->gauss, fsurf, tmp, pre);
        } else {
            hipLaunchKernelGGL(xc_norm_rowblur<64>, dim3(h, nt), dim3(64), sh, s, w, h, minmax, ws->gauss, fsurf, tmp, pre);
        }
    }
    hipLaunchKernelGGL(xc_colblur_argmax, dim3((w + 63) / 64, (h + XC_RB5 - 1) / XC_RB5, nt), dim3(256), 0, s,
                       w, h, ws->gauss, fsurf, tmp, dog_dbg, keys);
    return MIS_OK;
}

static int run_batch(mis_ctx* ctx, const XcPlan& p, const mis_rect* d_r1, const mis_rect* d_r2,
                     int tile0, int nt, mis_offset* d_out, float* dog_dbg, bool peak_stages = true)
{
    XcorrWs* ws = ctx->xc;
    const int w = p.w, h = p.h;
    const size_t N = (size_t)w * h;
    const float2 *tww = nullptr, *twh = nullptr;
    int rc = get_twiddle(ctx, w, &tww); if (rc) return rc;
    rc = get_twiddle(ctx, h, &twh); if (rc) return rc;

    float2* spec = (float2*)ws->buf;                                   // [nt][h][w]: joint spectrum of the tile pair
    float* surf = (float*)((char*)ws->buf + (size_t)nt * N * sizeof(float2));   // [nt][h][w]
    float* tmp = surf + (size_t)nt * N;                                // [nt][h][w]
    const int nbands = 1 + (int)ctx->band_img[0].size();
    float* prod = tmp + (size_t)nt * N;                                // [nt][h][w], only with several bands
    float* fsurf = nbands > 1 ? prod : surf;                           // what stages 4-6 work on
    unsigned* minmax = (unsigned*)ws->meta;
    unsigned long long* keys = (unsigned long long*)((char*)ws->meta + (((size_t)nt * 8 + 255) & ~(size_t)255));
    const float scale = (float)(1.0 / sqrt((double)N));
    ws->last_surface = fsurf;
    ws->last_abs = surf;
    rc = ensure_taps(ctx, w, h); if (rc) return rc;

    // the stages for tiles [toff, toff + ntk) of the batch on stream s (the arrays keep the layout of the whole batch)
    auto part = [&](int toff, int ntk, hipStream_t s) -> int {
        float2* spec_p = spec + (size_t)toff * N;
        float* surf_p = surf + (size_t)toff * N;
        float* tmp_p = tmp + (size_t)toff * N;
        float* prod_p = prod + (size_t)toff * N;
        float* fsurf_p = nbands > 1 ? prod_p : surf_p;
        unsigned* minmax_p = minmax + 2 * toff;
        unsigned long long* keys_p = keys + toff;
        size_t sh_rows = 0; bool pad_rows = false;      // LDS bytes and layout of the two row kernels
        for (int band = 0; band < nbands; band++) {
            XcImage im;
            im.img[0] = band ? ctx->band_img[0][band - 1] : ctx->img[0];
            im.img[1] = band ? ctx->band_img[1][band - 1] : ctx->img[1];
            im.dtype = ctx->dtype; im.iw = ctx->width; im.ih = ctx->height; im.stride = ctx->stride;
            {
                // padded LDS layout whenever it fits (it does for every tile size but the largest)
                const size_t nx = (size_t)p.RBw * w, nxp = ((nx + nx / 8 + 1) & ~(size_t)1);
                const bool pad = (nx % 8 == 0) && (nxp + w / 2) * sizeof(float2) <= 160 * 1024;
                sh_rows = ((pad ? nxp : nx) + w / 2) * sizeof(float2);
                pad_rows = pad;
                auto kf = pad ? xc_rows_fwd<true> : xc_rows_fwd<false>;
                auto ki = pad ? xc_rows_inv_abs<true> : xc_rows_inv_abs<false>;
                if (sh_rows > 64 * 1024) {
                    MIS_HIP(ctx, hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_rows));
                    MIS_HIP(ctx, hipFuncSetAttribute((const void*)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_rows));
                }
                hipLaunchKernelGGL(kf, dim3(h / p.RBw, ntk), dim3(256), sh_rows, s, im, d_r1, d_r2, tile0 + toff,
                                   w, h, p.logw, p.RBw, tww, spec_p, minmax_p, band == 0 ? keys_p : (unsigned long long*)nullptr);
            }
            {
                const size_t nx = (size_t)2 * p.CB * h, nxp = ((nx + nx / 8 + 1) & ~(size_t)1);
                const bool pad = (nx % 8 == 0) && (nxp + h / 2) * sizeof(float2) <= 160 * 1024;
                const size_t sh = ((pad ? nxp : nx) + h / 2) * sizeof(float2);
                auto kc = pad ? xc_cols<true> : xc_cols<false>;
                if (sh > 64 * 1024)
                    MIS_HIP(ctx, hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                hipLaunchKernelGGL(kc, dim3(w / (2 * p.CB), ntk), dim3(512), sh, s, w, h, p.logw, p.logh, p.CB, scale * scale, twh, spec_p);
            }
            {
                auto ki = pad_rows ? xc_rows_inv_abs<true> : xc_rows_inv_abs<false>;
                hipLaunchKernelGGL(ki, dim3(h / p.RBw, ntk), dim3(256), sh_rows, s, w, h, p.logw, p.RBw, scale,
                                   tww, spec_p, surf_p, minmax_p);
            }
            if (nbands > 1)
                hipLaunchKernelGGL(xc_band_product, dim3((unsigned)((N + 255) / 256), ntk), dim3(256), 0, s, N, minmax_p, surf_p, prod_p, band == 0);
        }
        if (!peak_stages) { MIS_HIP(ctx, hipGetLastError()); return MIS_OK; }
        const float* norm_p = nullptr;
        int rcp = launch_blur_stages(ctx, s, w, h, ntk, minmax_p, fsurf_p, tmp_p, nbands > 1, dog_dbg ? dog_dbg + (size_t)toff * N : nullptr, keys_p, &norm_p);
        if (rcp) return rcp;
        if (toff == 0) ws->last_surface = norm_p;
        hipLaunchKernelGGL(xc_finish, dim3((ntk + 63) / 64), dim3(64), 0, s, ntk, w, h, keys_p, minmax_p, norm_p, d_out + tile0 + toff);
        MIS_HIP(ctx, hipGetLastError());
        return MIS_OK;
    };
    // A batch is a chain of dependent launches, each with a tail and a start-up of its own: two halves on two streams
    // overlap them.  Small batches stay whole (a half must still fill the machine).
    if (nt < 64) return part(0, nt, ctx->stream);
    if (!ws->side) {
        MIS_HIP(ctx, hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming));
    }
    MIS_HIP(ctx, hipEventRecord(ws->ev_fork, ctx->stream));
    MIS_HIP(ctx, hipStreamWaitEvent(ws->side, ws->ev_fork, 0));
    const int half = nt / 2;
    rc = part(0, half, ctx->stream);
    const int rc2 = part(half, nt - half, ws->side);
    // joined whatever happened: nothing of this call stays behind on the side stream
    MIS_HIP(ctx, hipEventRecord(ws->ev_join, ws->side));
    MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ws->ev_join, 0));
    return rc ? rc : rc2;
}

// general-size variant of run_batch: stages 1-3 by row DFTs and transposes, stages 4-6 shared
static int run_batch_general(mis_ctx* ctx, const XcPlan& p, const mis_rect* d_r1, const mis_rect* d_r2,
                             int tile0, int nt, mis_offset* d_out, float* dog_dbg, bool peak_stages)
{
    XcorrWs* ws = ctx->xc;
    const int w = p.w, h = p.h;
    const size_t N = (size_t)w * h;
    XgDft dw, dh;
    int rc = get_dft(ctx, w, &dw); if (rc) return rc;
    rc = get_dft(ctx, h, &dh); if (rc) return rc;
    float2* spec = (float2*)ws->buf;                                   // [nt][2][h][w]
    float* surf = (float*)((char*)ws->buf + (size_t)nt * 2 * N * sizeof(float2));   // [nt][h][w]
    float* tmp = surf + (size_t)nt * N;                                // [nt][h][w]
    float2* specT = (float2*)(tmp + (size_t)nt * N);                   // [nt][2][w][h]
    const int nbands = 1 + (int)ctx->band_img[0].size();
    float* prod = (float*)(specT + (size_t)nt * 2 * N);                // [nt][h][w], only with several bands
    float* fsurf = nbands > 1 ? prod : surf;
    unsigned* minmax = (unsigned*)ws->meta;
    unsigned long long* keys = (unsigned long long*)((char*)ws->meta + (((size_t)nt * 8 + 255) & ~(size_t)255));
    hipStream_t s = ctx->stream;
    const float scale = (float)(1.0 / sqrt((double)N));
    ws->last_surface = fsurf;
    ws->last_abs = surf;
    hipLaunchKernelGGL(xc_init_meta, dim3((nt + 255) / 256), dim3(256), 0, s, nt, minmax, keys);
    const dim3 tgrid((w + 31) / 32, (h + 31) / 32, 1), tgridT((h + 31) / 32, (w + 31) / 32, 1);
    for (int band = 0; band < nbands; band++) {
        XcImage im;
        im.img[0] = band ? ctx->band_img[0][band - 1] : ctx->img[0];
        im.img[1] = band ? ctx->band_img[1][band - 1] : ctx->img[1];
        im.dtype = ctx->dtype; im.iw = ctx->width; im.ih = ctx->height; im.stride = ctx->stride;
        if (band) hipLaunchKernelGGL(xc_reset_minmax, dim3((nt + 255) / 256), dim3(256), 0, s, nt, minmax);
        hipLaunchKernelGGL(xg_load_tiles, dim3((w + 255) / 256, h, nt), dim3(256), 0, s, im, d_r1, d_r2, tile0, w, h, spec);
        rc = launch_dft(ctx, spec, dw, nt * h, h, 2 * N, 0); if (rc) return rc;
        hipLaunchKernelGGL(xg_transpose, dim3(tgrid.x, tgrid.y, nt), dim3(256), 0, s, spec, specT, h, w, 2 * N, 2 * N);
        rc = launch_dft(ctx, specT, dh, nt * w, w, 2 * N, 0); if (rc) return rc;
        hipLaunchKernelGGL(xg_product, dim3((h + 255) / 256, w, nt), dim3(256), 0, s, specT, w, h, scale);
        rc = launch_dft(ctx, specT + N, dh, nt * w, w, 2 * N, 1); if (rc) return rc;
        hipLaunchKernelGGL(xg_transpose, dim3(tgridT.x, tgridT.y, nt), dim3(256), 0, s, specT + N, spec, w, h, 2 * N, 2 * N);
        rc = launch_dft(ctx, spec, dw, nt * h, h, 2 * N, 1); if (rc) return rc;
        hipLaunchKernelGGL(xg_abs_minmax, dim3((unsigned)((N + 256 * XG_ABS_PER_THREAD - 1) / (256 * XG_ABS_PER_THREAD)), nt), dim3(256), 0, s, spec, N, scale, surf, minmax);
        if (nbands > 1)
            hipLaunchKernelGGL(xc_band_product, dim3((unsigned)((N + 255) / 256), nt), dim3(256), 0, s, N, minmax, surf, prod, band == 0);
    }
    if (!peak_stages) { MIS_HIP(ctx, hipGetLastError()); return MIS_OK; }
    rc = ensure_taps(ctx, w, h); if (rc) return rc;
    const float* norm_p = nullptr;
    rc = launch_blur_stages(ctx, ctx->stream, w, h, nt, minmax, fsurf, tmp, nbands > 1, dog_dbg, keys, &norm_p);
    if (rc) return rc;
    ws->last_surface = norm_p;
    hipLaunchKernelGGL(xc_finish, dim3((nt + 63) / 64), dim3(64), 0, s, nt, w, h, keys, minmax, norm_p, d_out + tile0);
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

static size_t bytes_per_tile(const XcPlan& p)
{
    // power-of-two tiles: one joint spectrum; general sizes: two spectra and their transposes
    return (size_t)p.w * p.h * ((p.general ? 4 : 1) * sizeof(float2) + 4 * sizeof(float));
}

static int xcorr_run(mis_ctx* ctx, const mis_rect* d_r1, const mis_rect* d_r2, int n, int tile_w, int tile_h,
                     mis_offset* d_out, float* dog_dbg, bool peak_stages = true)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "tile correlation: no image pair bound");
    if (n < 0 || (n > 0 && (!d_r1 || !d_r2 || (!d_out && peak_stages)))) return mis_fail(ctx, MIS_ERR_ARG, "tile correlation: NULL buffer");
    if (n == 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mis_wait_upload(ctx, ctx->height, ctx->stream);      // tiles reach anywhere in a pair that may still be on its way
    if (rc) return rc;
    rc = ensure_ws(ctx); if (rc) return rc;
    XcPlan p;
    rc = make_plan(ctx, tile_w, tile_h, &p); if (rc) return rc;
    // batches bounded by a 2 GiB workspace (one tile minimum)
    const size_t per = bytes_per_tile(p);
    int nb = (int)((2048ull << 20) / per);
    if (nb < 1) nb = 1;
    if (nb > n) nb = n;
    // tiles ride in grid.z (limit 65535); the general-size kernels put both images of a tile there
    const int zmax = p.general ? 32767 : 65535;
    if (nb > zmax) nb = zmax;
    XcorrWs* ws = ctx->xc;
    rc = mis_reserve(ctx, &ws->buf, &ws->buf_bytes, per * nb); if (rc) return rc;
    rc = mis_reserve(ctx, &ws->meta, &ws->meta_bytes, (size_t)nb * 16 + 512); if (rc) return rc;
    for (int t0 = 0; t0 < n; t0 += nb) {
        const int nt = (n - t0 < nb) ? n - t0 : nb;
        rc = p.general ? run_batch_general(ctx, p, d_r1, d_r2, t0, nt, d_out, dog_dbg, peak_stages)
                       : run_batch(ctx, p, d_r1, d_r2, t0, nt, d_out, dog_dbg, peak_stages);
        if (rc) return rc;
    }
    return MIS_OK;
}

// forward transform of a batch of rasters with natural-order output (image-fft.cpp:45-105)
__global__ __launch_bounds__(256) void xc_fft_rows_only(const void* __restrict__ in, int dtype, int w, int h,
                                                        int logw, int RB, const float2* __restrict__ tw,
                                                        float2* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* x = (float2*)smem_raw;
    const int t = blockIdx.y, row0 = blockIdx.x * RB;
    const int bpp = dtype == MIS_U8 ? 1 : dtype == MIS_U16 ? 2 : 4;
    const unsigned char* base = (const unsigned char*)in + (size_t)t * h * w * bpp;
    for (int i = threadIdx.x; i < RB * w; i += 256)
        x[i] = make_float2(load_px(base, dtype, (size_t)w * bpp, i % w, row0 + i / w), 0.f);
    __syncthreads();
    fft_dif_fwd<256>(x, logw, RB, tw);
    float2* o = out + ((size_t)t * h + row0) * w;
    for (int i = threadIdx.x; i < RB * w; i += 256) o[i] = x[i];     // columns bit-reversed
}

__global__ __launch_bounds__(512) void xc_fft_cols_only(int w, int h, int logw, int logh, int CB, double norm,
                                                        const float2* __restrict__ tw,
                                                        const float2* __restrict__ in, float2* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* A = (float2*)smem_raw;
    const int t = blockIdx.y, c0 = blockIdx.x * CB;
    const float2* src = in + (size_t)t * h * w;
    float2* dst = out + (size_t)t * h * w;
    for (int i = threadIdx.x; i < CB * h; i += 512) {
        const int r = i / CB, c = i % CB;
        A[c * h + r] = src[(size_t)r * w + c0 + c];
    }
    __syncthreads();
    fft_dif_fwd<512>(A, logh, CB, tw);
    // undo both bit reversals on the way out; divide by sqrt(N) held in double (image-fft.cpp:84-87)
    for (int i = threadIdx.x; i < CB * h; i += 512) {
        const int r = i / CB, c = i % CB;
        const float2 v = A[c * h + r];
        const int rr = bitrev(r, logh), cc = bitrev(c0 + c, logw);
        dst[(size_t)rr * w + cc] = make_float2((float)((double)v.x / norm), (float)((double)v.y / norm));
    }
}

// image-cross-cor.cpp:166-170: floor(norm * 255) as a byte, then fft2shift (even sizes: a swap of
// quadrants, i.e. a circular shift by half the size in both directions)
__global__ __launch_bounds__(256) void xc_quantise_shift(int w, int h, const unsigned* __restrict__ minmax,
                                                         const float* __restrict__ surf, uint8_t* __restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float mn = __uint_as_float(minmax[0]), mx = __uint_as_float(minmax[1]);
    const double v = floor((((double)surf[(size_t)y * w + x] - (double)mn) / (double)(mx - mn)) * 255.0);
    const uint8_t b = (mx == mn || !(v >= 0)) ? 0 : (uint8_t)v;
    // swap of the two halves of length floor(n/2); an odd last row / column stays (image-cross-cor.cpp:52-80)
    const int hw = w / 2, hh = h / 2;
    const int xs = x < hw ? x + hw : (x < 2 * hw ? x - hw : x);
    const int ys = y < hh ? y + hh : (y < 2 * hh ? y - hh : y);
    out[(size_t)ys * w + xs] = b;
}

extern "C" {

int mis_xcorr_tiles_device(mis_ctx* ctx, const mis_rect* d_r1, const mis_rect* d_r2, int n,
                           int tile_w, int tile_h, mis_offset* d_out)
{
    return xcorr_run(ctx, d_r1, d_r2, n, tile_w, tile_h, d_out, nullptr);
}

int mis_xcorr_tiles(mis_ctx* ctx, const mis_rect* r1, const mis_rect* r2, int n,
                    int tile_w, int tile_h, mis_offset* out)
{
    if (!ctx) return MIS_ERR_ARG;
    if (n < 0 || (n > 0 && (!r1 || !r2 || !out))) return mis_fail(ctx, MIS_ERR_ARG, "tile correlation: NULL buffer");
    if (n == 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_ws(ctx); if (rc) return rc;
    XcorrWs* ws = ctx->xc;
    const size_t rb = (size_t)n * sizeof(mis_rect), ob = (size_t)n * sizeof(mis_offset);
    size_t cap = 64 * 1024;                       // grows by doubling: every level of the recursion brings more tiles
    while (cap < 2 * rb + ob) cap *= 2;
    rc = mis_reserve(ctx, &ws->io, &ws->io_bytes, cap); if (rc) return rc;
    if (ws->io_host_bytes < cap) {
        if (ws->io_host) { (void)hipHostFree(ws->io_host); ws->io_host = nullptr; ws->io_host_bytes = 0; }
        MIS_HIP(ctx, hipHostMalloc(&ws->io_host, cap, hipHostMallocDefault));
        ws->io_host_bytes = cap;
    }
    mis_rect* d1 = (mis_rect*)ws->io; mis_rect* d2 = d1 + n; mis_offset* dout = (mis_offset*)(d2 + n);
    // both rectangle lists through the page-locked buffer in ONE copy, the offsets back in one: pageable copies of a few
    // kilobytes cost a staging step each, and the recursion makes a call per tile size and level
    char* hb = (char*)ws->io_host;
    memcpy(hb, r1, rb); memcpy(hb + rb, r2, rb);
    hipError_t e = hipMemcpyAsync(d1, hb, 2 * rb, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "tile correlation: upload failed: %s", hipGetErrorString(e));
    if (!rc) rc = xcorr_run(ctx, d1, d2, n, tile_w, tile_h, dout, nullptr);
    if (!rc) {
        e = hipMemcpyAsync(hb + 2 * rb, dout, ob, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "tile correlation: download failed: %s", hipGetErrorString(e));
        else memcpy(out, hb + 2 * rb, ob);
    }
    (void)hipStreamSynchronize(ctx->stream);      // the staging buffer is free for the next call, also after a failure
    return rc;
}

int mis_xcorr_tile_surfaces(mis_ctx* ctx, mis_rect r1, mis_rect r2, int tile_w, int tile_h,
                            float* surface, float* dog)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "tile correlation: no image pair bound");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)tile_w * tile_h;
    void* d = nullptr;
    MIS_HIP(ctx, hipMalloc(&d, 2 * sizeof(mis_rect) + sizeof(mis_offset) + N * sizeof(float)));
    mis_rect* d1 = (mis_rect*)d; mis_rect* d2 = d1 + 1; mis_offset* dout = (mis_offset*)(d2 + 1);
    float* ddog = (float*)(dout + 1);
    int rc = MIS_OK;
    hipError_t e = hipMemcpyAsync(d1, &r1, sizeof r1, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, &r2, sizeof r2, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "upload failed: %s", hipGetErrorString(e));
    if (!rc) rc = xcorr_run(ctx, d1, d2, 1, tile_w, tile_h, dout, ddog);
    if (!rc) {
        const float* surf = ctx->xc->last_surface;
        if (surface) e = hipMemcpyAsync(surface, surf, N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && dog) e = hipMemcpyAsync(dog, ddog, N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "download failed: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

int mis_fft2d_forward(mis_ctx* ctx, const void* in, int dtype, int w, int h, int batch, float* out_complex)
{
    if (!ctx) return MIS_ERR_ARG;
    const int bpp = mis_pixel_bytes(dtype);
    if (!in || !out_complex || !bpp || batch < 1) return mis_fail(ctx, MIS_ERR_ARG, "fft2d_forward: bad arguments");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_ws(ctx); if (rc) return rc;
    XcPlan p;
    rc = make_plan(ctx, w, h, &p); if (rc) return rc;
    const size_t N = (size_t)w * h;
    if (p.general) {
        XgDft dw, dh;
        rc = get_dft(ctx, w, &dw); if (rc) return rc;
        rc = get_dft(ctx, h, &dh); if (rc) return rc;
        XcorrWs* ws = ctx->xc;
        rc = mis_reserve(ctx, &ws->buf, &ws->buf_bytes, N * (bpp + 2 * sizeof(float2))); if (rc) return rc;
        hipStream_t s = ctx->stream;
        float2* t1 = (float2*)ws->buf;
        float2* t2 = t1 + N;
        void* din = (void*)(t2 + N);
        for (int b = 0; b < batch; b++) {
            MIS_HIP(ctx, hipMemcpyAsync(din, (const char*)in + (size_t)b * N * bpp, N * bpp, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(xg_load_raster, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, din, dtype, N, t1);
            rc = launch_dft(ctx, t1, dw, h, h, 0, 0); if (rc) return rc;
            hipLaunchKernelGGL(xg_transpose, dim3((w + 31) / 32, (h + 31) / 32, 1), dim3(256), 0, s, t1, t2, h, w, N, N);
            rc = launch_dft(ctx, t2, dh, w, w, 0, 0); if (rc) return rc;
            hipLaunchKernelGGL(xg_transpose, dim3((h + 31) / 32, (w + 31) / 32, 1), dim3(256), 0, s, t2, t1, w, h, N, N);
            hipLaunchKernelGGL(xg_divide, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, t1, N, sqrt((double)N));
            MIS_HIP(ctx, hipGetLastError());
            MIS_HIP(ctx, hipMemcpyAsync(out_complex + (size_t)b * N * 2, t1, N * sizeof(float2), hipMemcpyDeviceToHost, s));
            MIS_HIP(ctx, hipStreamSynchronize(s));
        }
        return MIS_OK;
    }
    const float2 *tww = nullptr, *twh = nullptr;
    rc = get_twiddle(ctx, w, &tww); if (rc) return rc;
    rc = get_twiddle(ctx, h, &twh); if (rc) return rc;
    int CB = 8192 / h; if (CB > 16) CB = 16; if (CB > w) CB = w; if (CB < 1) CB = 1;   // one array of <= 8192 values in LDS
    // rasters are processed in groups that fit a 2 GiB workspace
    const size_t per = N * (bpp + 2 * sizeof(float2));
    int nb = (int)((2048ull << 20) / per); if (nb < 1) nb = 1; if (nb > batch) nb = batch; if (nb > 65535) nb = 65535;
    XcorrWs* ws = ctx->xc;
    rc = mis_reserve(ctx, &ws->buf, &ws->buf_bytes, per * nb); if (rc) return rc;
    hipStream_t s = ctx->stream;
    for (int b0 = 0; b0 < batch; b0 += nb) {
        const int nt = batch - b0 < nb ? batch - b0 : nb;
        float2* t1 = (float2*)ws->buf;
        float2* t2 = t1 + (size_t)nt * N;
        void* din = (void*)(t2 + (size_t)nt * N);
        MIS_HIP(ctx, hipMemcpyAsync(din, (const char*)in + (size_t)b0 * N * bpp, (size_t)nt * N * bpp, hipMemcpyHostToDevice, s));
        const size_t sh1 = (size_t)p.RBw * w * sizeof(float2);
        if (sh1 > 64 * 1024)
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)xc_fft_rows_only, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1));
        hipLaunchKernelGGL(xc_fft_rows_only, dim3(h / p.RBw, nt), dim3(256), sh1, s, din, dtype, w, h, p.logw, p.RBw, tww, t1);
        const size_t sh2 = (size_t)CB * h * sizeof(float2);
        if (sh2 > 64 * 1024)
            MIS_HIP(ctx, hipFuncSetAttribute((const void*)xc_fft_cols_only, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2));
        hipLaunchKernelGGL(xc_fft_cols_only, dim3(w / CB, nt), dim3(512), sh2, s, w, h, p.logw, p.logh, CB,
                           sqrt((double)N), twh, t1, t2);
        MIS_HIP(ctx, hipGetLastError());
        MIS_HIP(ctx, hipMemcpyAsync(out_complex + (size_t)b0 * N * 2, t2, (size_t)nt * N * sizeof(float2), hipMemcpyDeviceToHost, s));
        MIS_HIP(ctx, hipStreamSynchronize(s));
    }
    return MIS_OK;
}

int mis_masked_peak(mis_ctx* ctx, const float* bands, int width, int height, int n_bands, float mask_pct,
                    int* offset_x, int* offset_y, float* peak)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!bands || width < 1 || height < 1 || n_bands < 1 || !offset_x || !offset_y || !peak)
        return mis_fail(ctx, MIS_ERR_ARG, "masked peak: bad arguments");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_ws(ctx); if (rc) return rc;
    XcorrWs* ws = ctx->xc;
    const size_t N = (size_t)width * height;
    rc = mis_reserve(ctx, &ws->buf, &ws->buf_bytes, N * n_bands * sizeof(float)); if (rc) return rc;
    rc = mis_reserve(ctx, &ws->meta, &ws->meta_bytes, 512); if (rc) return rc;
    hipStream_t s = ctx->stream;
    // image2.cpp:248-263; the mask bounds are float expressions truncated on assignment
    const int pp_row = height / 2, pp_col = width / 2;
    if (mask_pct > 99) mask_pct = 99;
    if (mask_pct <= 1) mask_pct = 0;
    const int r0 = (int)((float)pp_row - ((float)pp_row * mask_pct / 100.0f));
    const int r1 = (int)((float)pp_row + ((float)pp_row * mask_pct / 100.0f));
    const int c0 = (int)((float)pp_col - ((float)pp_col * mask_pct / 100.0f));
    const int c1 = (int)((float)pp_col + ((float)pp_col * mask_pct / 100.0f));
    unsigned long long* key = (unsigned long long*)ws->meta;
    MIS_HIP(ctx, hipMemcpyAsync(ws->buf, bands, N * n_bands * sizeof(float), hipMemcpyHostToDevice, s));
    MIS_HIP(ctx, hipMemsetAsync(key, 0, sizeof *key, s));
    hipLaunchKernelGGL(xc_masked_peak, dim3((width + 255) / 256, height), dim3(256), 0, s, (const float*)ws->buf,
                       width, height, n_bands, pp_row * 2, pp_col * 2, r0, r1, c0, c1, key);
    MIS_HIP(ctx, hipGetLastError());
    unsigned long long k = 0;
    MIS_HIP(ctx, hipMemcpyAsync(&k, key, sizeof k, hipMemcpyDeviceToHost, s));
    MIS_HIP(ctx, hipStreamSynchronize(s));
    int max_col = 0, max_row = 0;
    float mx = -1.0f;                              // nothing above the initial -1 (image2.cpp:266-267)
    if (k) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu);
        unsigned b = (unsigned)(k >> 32);
        b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
        memcpy(&mx, &b, 4);
        max_col = (int)(idx % (unsigned)width); max_row = (int)(idx / (unsigned)width);
    }
    *offset_x = max_col - pp_col;                  // image2.cpp:301-302
    *offset_y = max_row - pp_row;
    *peak = mx;
    return MIS_OK;
}

int mis_xcorr_surface_u8(mis_ctx* ctx, uint8_t* out)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "correlation surface: no image pair bound");
    if (!out) return mis_fail(ctx, MIS_ERR_ARG, "correlation surface: out is NULL");
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const int w = ctx->width, h = ctx->height;
    const size_t N = (size_t)w * h;
    void* d = nullptr;
    MIS_HIP(ctx, hipMalloc(&d, 2 * sizeof(mis_rect) + sizeof(mis_offset) + N));
    mis_rect* d1 = (mis_rect*)d; mis_rect* d2 = d1 + 1; mis_offset* dout = (mis_offset*)(d2 + 1);
    uint8_t* d8 = (uint8_t*)(dout + 1);
    const mis_rect full = { 0, 0, w, h };
    int rc = MIS_OK;
    hipError_t e = hipMemcpyAsync(d1, &full, sizeof full, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, &full, sizeof full, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "upload failed: %s", hipGetErrorString(e));
    if (!rc) rc = xcorr_run(ctx, d1, d2, 1, w, h, nullptr, nullptr, false);   // stages 1-3 only
    if (!rc) {
        XcorrWs* ws = ctx->xc;
        const float* surf = ctx->xc->last_abs;
        hipStream_t s = ctx->stream;
        hipLaunchKernelGGL(xc_quantise_shift, dim3((w + 255) / 256, h), dim3(256), 0, s, w, h,
                           (const unsigned*)ws->meta, surf, d8);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, d8, N, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "correlation surface failed: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
