// Ray triangulation of matched points, gfx950: calculateXYZ2 (image3.cpp:102-132) and
// calculateXYZ (image3.cpp:148-264, FAST_LINE variant), one thread per point, strict float32.
//
// This file is compiled with -ffp-contract=off: the reference rounds every product and sum
// separately (scaleAdd, image3.cpp:110,116), a fused multiply-add would move the 1.0-step search.
// The camera inverse is hoisted out of the per-point code (the reference recomputes it for every
// point, image3.cpp:169-172); it is evaluated on the host with the arithmetic OpenCV 2.4 uses for
// 3x3 CV_32F matrices (cofactors and dot products in double, results narrowed to float).
#include "mis_common.h"

#include <math.h>
#include <mutex>

struct TriCam {
    float Rp[9];   // R^-1
    float tp[3];   // -R^-1 t
    float fx, fy, cx, cy;
};

static void inv3(const float* S, float* D)
{
#define Sf(r, c) ((double)S[(r) * 3 + (c)])
    double d = Sf(0,0) * (Sf(1,1) * Sf(2,2) - Sf(1,2) * Sf(2,1))
             - Sf(0,1) * (Sf(1,0) * Sf(2,2) - Sf(1,2) * Sf(2,0))
             + Sf(0,2) * (Sf(1,0) * Sf(2,1) - Sf(1,1) * Sf(2,0));
    if (d == 0.) { for (int i = 0; i < 9; i++) D[i] = 0.f; return; }
    d = 1. / d;
    D[0] = (float)((Sf(1,1) * Sf(2,2) - Sf(1,2) * Sf(2,1)) * d);
    D[1] = (float)((Sf(0,2) * Sf(2,1) - Sf(0,1) * Sf(2,2)) * d);
    D[2] = (float)((Sf(0,1) * Sf(1,2) - Sf(0,2) * Sf(1,1)) * d);
    D[3] = (float)((Sf(1,2) * Sf(2,0) - Sf(1,0) * Sf(2,2)) * d);
    D[4] = (float)((Sf(0,0) * Sf(2,2) - Sf(0,2) * Sf(2,0)) * d);
    D[5] = (float)((Sf(0,2) * Sf(1,0) - Sf(0,0) * Sf(1,2)) * d);
    D[6] = (float)((Sf(1,0) * Sf(2,1) - Sf(1,1) * Sf(2,0)) * d);
    D[7] = (float)((Sf(0,1) * Sf(2,0) - Sf(0,0) * Sf(2,1)) * d);
    D[8] = (float)((Sf(0,0) * Sf(1,1) - Sf(0,1) * Sf(1,0)) * d);
#undef Sf
}

static TriCam make_cam(const mis_camera* c)
{
    TriCam t;
    inv3(c->R, t.Rp);                                     // image3.cpp:169,171
    for (int i = 0; i < 3; i++) {                         // image3.cpp:170,172: t' = -R' t
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)t.Rp[i * 3 + k] * (double)c->T[k];
        t.tp[i] = (float)(s * -1.0);
    }
    t.fx = c->A[0]; t.fy = c->A[4]; t.cx = c->A[2]; t.cy = c->A[5];   // image3.cpp:175-183
    return t;
}

// image3.cpp:102-132
__device__ __forceinline__ float xyz2(const float* ang1, const float* t1p, const float* ang2, const float* t2p,
                                      float z1, float& X, float& Y, float& Z)
{
    float w1x = z1 * ang1[0] + t1p[0];
    float w1y = z1 * ang1[1] + t1p[1];
    float w1z = z1 * ang1[2] + t1p[2];
    float z2 = (w1z - t2p[2]) / ang2[2];
    float w2x = z2 * ang2[0] + t2p[0];
    float w2y = z2 * ang2[1] + t2p[1];
    float dx = w1x - w2x, dy = w1y - w2y;
    float err = sqrtf(dx * dx + dy * dy);   // correctly rounded (the __fsqrt_rn intrinsic is not on gfx950)
    X = w1x; Y = w1y; Z = w1z;
    return err;
}

// rotate the pixel ray into world coordinates: gemm with double accumulation (image3.cpp:185-199).
// The reference's integer pixel coordinates are converted to float where they meet the principal point
// (usual arithmetic conversion of `u1 - cx1`); they arrive here converted, which also admits a
// fractional column.
__device__ __forceinline__ void ray(const TriCam& c, float u, float v, float* ang)
{
    float a0 = (u - c.cx) / c.fx;
    float a1 = -(v - c.cy) / c.fy;
    float a2 = -1.0f;
    for (int i = 0; i < 3; i++) {
        double s = 0.0;                          // the accumulator starts at +0 (sign of a zero result)
        s += (double)c.Rp[i * 3] * (double)a0;
        s += (double)c.Rp[i * 3 + 1] * (double)a1;
        s += (double)c.Rp[i * 3 + 2] * (double)a2;
        ang[i] = (float)s;
    }
}

// ---- the same search, two steps of z per iteration --------------------------------------------
// The two loops of image3.cpp:220-246 evaluate xyz2 at z0, z0+1, ... about a hundred steps past the
// minimum, so a point costs 200+ evaluations.  The fast path below produces the same bits with a
// third of the instructions:
//  * two consecutive z are evaluated together with packed float32 instructions (each lane of a
//    packed multiply / add / fma rounds exactly like the scalar one);
//  * the quotient (w1z - t2z) / ang2z uses the division sequence the compiler emits for a float
//    division (reciprocal estimate, one Newton step, two residual corrections of the quotient) with
//    the part that depends on the divisor alone hoisted out of the loops.  That sequence skips the
//    operand scaling and the special-case fix-up of the general one, which only act on zero, huge,
//    tiny or non-finite operands: tri_fast_ok() admits the points and cameras whose operands stay
//    away from those ranges, every other point runs the
//    loops as written, with general arithmetic;
//  * the square root is not evaluated per step.  err = sqrtf(s) is only ever compared with the
//    smallest error so far, E; sqrtf rounds to nearest, so sqrtf(s) < E holds exactly when s lies
//    below the square of the midpoint between E and the float below it.  That bound is kept as a
//    float (the largest one below the squared midpoint, which itself is never a float) and is
//    recomputed, in double, whenever E changes;
//  * X, Y, Z of the best step are recomputed from its z at the end (they depend on z alone).
typedef float tri_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float tri_below(float v) { return __uint_as_float(__float_as_uint(v) - 1u); }   // v > 0

// largest float s with sqrtf(s) < E  (NaN when no s qualifies through a NaN E; -1 for E == 0)
__device__ float tri_bound(float E)
{
    if (E == 0.0f) return -1.0f;
    const double B = 0.5 * ((double)tri_below(E) + (double)E);
    const double T = B * B;                                  // exact: B has 25 significant bits
    float f = (float)T;
    if ((double)f >= T) f = tri_below(f);
    return f;
}

struct TriFast;
__device__ double tri_error_bound(const TriFast& r, const float* t1p, const float* t2p, float z0, float lim);

struct TriFast {
    float a1[3], a2[3];      // ray directions
    float nb, y;             // -ang2z, refined reciprocal of ang2z
};

__device__ __forceinline__ tri_f2 tri_s2(const TriFast& r, const float* t1p, const float* t2p, tri_f2 z)
{
    const tri_f2 w1x = z * r.a1[0] + t1p[0];
    const tri_f2 w1y = z * r.a1[1] + t1p[1];
    const tri_f2 w1z = z * r.a1[2] + t1p[2];
    const tri_f2 a = w1z - t2p[2];
    const tri_f2 nb = { r.nb, r.nb }, y = { r.y, r.y };
    const tri_f2 q0 = a * y;
    const tri_f2 r0 = __builtin_elementwise_fma(nb, q0, a);
    const tri_f2 q1 = __builtin_elementwise_fma(r0, y, q0);
    const tri_f2 r1 = __builtin_elementwise_fma(nb, q1, a);
    const tri_f2 z2 = __builtin_elementwise_fma(r1, y, q1);
    const tri_f2 w2x = z2 * r.a2[0] + t2p[0];
    const tri_f2 w2y = z2 * r.a2[1] + t2p[1];
    const tri_f2 dx = w1x - w2x, dy = w1y - w2y;
    return dx * dx + dy * dy;
}

// Early end of a search loop.  In exact arithmetic the error of image3.cpp:102-132 is the length of a
// vector that is linear in z (the second ray's parameter follows from equal heights), i.e. the
// square root of a quadratic: convex, so once it has risen from one step to the next it keeps
// rising in that direction.  The float evaluation differs from the exact one by at most
// eps + 4u*err (u = 2^-24; eps from the rounding of every product, sum and the quotient, with the
// magnitudes bounded over the whole range a loop can visit).  If two consecutive steps a, b of a
// loop satisfy err_b - err_a > 2 (eps + 4u*err_b), the exact error rose, so every later step c has
// float error >= exact_c - (..) >= exact_b - (..) > err_a >= smallest error so far: none of the
// remaining steps (the reference runs about a hundred of them before it gives up) can change the
// result, and the loop ends there.  The test is done on the sums of squares without a square root:
// with D = s_b - s_a - 16u*s_b it is D > 0 and D^2 > 16 eps^2 s_b.  The value returned carries a
// factor 1.25 of slack on top of the rounded-up constants.
__device__ double tri_error_bound(const TriFast& r, const float* t1p, const float* t2p, float z0, float lim)
{
    const double u = 5.9604644775390625e-08;
    const double zm = fmax(fabs((double)z0), fabs((double)lim)) + 210.0;      // every z a loop can reach
    const double m1x = zm * fabs((double)r.a1[0]) + fabs((double)t1p[0]);
    const double m1y = zm * fabs((double)r.a1[1]) + fabs((double)t1p[1]);
    const double m1z = zm * fabs((double)r.a1[2]) + fabs((double)t1p[2]);
    const double e1x = 2.5 * u * m1x, e1y = 2.5 * u * m1y, e1z = 2.5 * u * m1z;   // product and sum of w1
    const double an = m1z + fabs((double)t2p[2]);                             // |w1z - t2z|
    const double ia = 1.0 / fabs((double)r.a2[2]);
    const double z2m = an * ia * 1.000001;                                    // |z2|
    const double ez2 = (e1z + 1.5 * u * an) * ia + 1.5 * u * z2m;             // difference, then the rounded quotient
    const double m2x = z2m * fabs((double)r.a2[0]) + fabs((double)t2p[0]);
    const double m2y = z2m * fabs((double)r.a2[1]) + fabs((double)t2p[1]);
    const double e2x = fabs((double)r.a2[0]) * ez2 + 2.5 * u * m2x;
    const double e2y = fabs((double)r.a2[1]) * ez2 + 2.5 * u * m2y;
    const double edx = e1x + e2x + 1.5 * u * (m1x + m2x);
    const double edy = e1y + e2y + 1.5 * u * (m1y + m2y);
    return 1.25 * sqrt(edx * edx + edy * edy);
}

// cameras whose heights keep the quotient's operands in the plain range of the division sequence
static bool tri_cams_fast(const TriCam& c1, const TriCam& c2)
{
    const float lo = 9.5367431640625e-07f, hi = 1048576.0f;        // 2^-20, 2^20
    const float t1 = fabsf(c1.tp[2]), t2 = fabsf(c2.tp[2]);
    return t1 <= hi && t2 >= lo && t2 <= hi;                        // false for NaN
}
__device__ __forceinline__ bool tri_fast_ok(float ang1z, float ang2z, float z0)
{
    const float lo = 9.5367431640625e-07f, hi = 1048576.0f;
    const float b = fabsf(ang2z);
    // z0: anything below the 2^25 at which the caller's own special case takes over keeps |w1z - t2z| below 2^46
    return b >= lo && b <= hi && fabsf(ang1z) <= hi && fabsf(z0) < 33554432.0f;   // false for NaN
}

// image3.cpp:148-264 for one point
__device__ void triangulate_point(const TriCam& c1, const TriCam& c2, bool cams_fast,
                                       float u1, float v1, float u2, float v2, float& X, float& Y, float& Z)
{
    TriFast r;
    ray(c1, u1, v1, r.a1);
    ray(c2, u2, v2, r.a2);
    const float* t1p = c1.tp;
    const float* t2p = c2.tp;
    float z0 = 0.0f;
    const float error0 = xyz2(r.a1, t1p, r.a2, t2p, z0, X, Y, Z);
    float z1 = 1000.0f;
    const float error1 = xyz2(r.a1, t1p, r.a2, t2p, z1, X, Y, Z);
    z0 = z0 + ((z1 - z0) / (error1 - error0) * (0 - error0));          // image3.cpp:216
    // Degenerate starts (parallel or nearly parallel rays: the secant step above divides by a zero or
    // tiny difference).  NaN: neither loop runs.  +-infinity: every step evaluates to infinity or NaN,
    // which never compares below the smallest error so far, so the z = 1000 result stands.
    // |z0| >= 2^25: z +- 1.0f rounds back to z, so every step of a loop is the same evaluation and only
    // the first can change anything.
    if (!(fabsf(z0) <= 3.0e38f)) return;
    if (fabsf(z0) >= 33554432.0f) {
        float X1, Y1, Z1, min_error = error0;
        if (z0 < t1p[2]) {
            const float e = xyz2(r.a1, t1p, r.a2, t2p, z0, X1, Y1, Z1);
            if (e < min_error) { min_error = e; X = X1; Y = Y1; Z = Z1; }
        }
        if (z0 > 0) {
            const float e = xyz2(r.a1, t1p, r.a2, t2p, z0, X1, Y1, Z1);
            if (e < min_error) { min_error = e; X = X1; Y = Y1; Z = Z1; }
        }
        return;
    }
    if (!(cams_fast && tri_fast_ok(r.a1[2], r.a2[2], z0))) {
        // same search, general arithmetic
        float X1, Y1, Z1, min_error = error0;
        int counter = 0;
        for (z1 = z0; z1 < t1p[2]; z1 += 1.0f) {
            float e = xyz2(r.a1, t1p, r.a2, t2p, z1, X1, Y1, Z1);
            if (e < min_error) { counter = 0; min_error = e; X = X1; Y = Y1; Z = Z1; }
            else if (counter++ > 100) break;
        }
        counter = 0;
        for (z1 = z0; z1 > 0; z1 -= 1.0f) {
            float e = xyz2(r.a1, t1p, r.a2, t2p, z1, X1, Y1, Z1);
            if (e < min_error) { counter = 0; min_error = e; X = X1; Y = Y1; Z = Z1; }
            else if (counter++ > 100) break;
        }
        return;
    }
    {
        const float b = r.a2[2];
        const float rc = __builtin_amdgcn_rcpf(b);
        const float e0 = fmaf(-b, rc, 1.0f);
        r.y = fmaf(e0, rc, rc);
        r.nb = -b;
    }
    float zbest = 1000.0f;              // X, Y, Z hold the z = 1000 evaluation until a step improves on error0
    float bound = tri_bound(error0);
    const float lim = t1p[2];
    // early end (see tri_error_bound): float test with 0.1 % slack on either side, far above the few
    // ulp the test's own roundings can contribute; K = 16 eps^2
    const double eps = tri_error_bound(r, t1p, t2p, z0, lim);
    const float K = (float)(16.0 * eps * eps * 1.002);
    auto rising = [&](float sa, float sb) -> bool {
        const float D = fmaf(-17.0f * 5.9604644775390625e-08f, sb, (sb - sa) * 0.999f);
        return D > 0.0f && D * D * 0.999f > K * sb;
    };
    for (int dir = 0; dir < 2; dir++) {
        const float step = dir ? -1.0f : 1.0f;
        int counter = 0;
        float smin = __builtin_inff();                         // smallest sum of squares among this loop's completed steps
        tri_f2 z = { z0, z0 + step };
        for (;;) {
            if (dir ? !(z.x > 0) : !(z.x < lim)) break;
            const tri_f2 s = tri_s2(r, t1p, t2p, z);
            if (s.x <= bound) { counter = 0; zbest = z.x; bound = tri_bound(sqrtf(s.x)); }
            else if (counter++ > 100) break;
            if (dir ? !(z.y > 0) : !(z.y < lim)) break;
            if (s.y <= bound) { counter = 0; zbest = z.y; bound = tri_bound(sqrtf(s.y)); }
            else if (counter++ > 100) break;
            smin = fminf(smin, s.x);                           // NaN sums are ignored by fminf and fail the test below
            if (rising(smin, s.y)) break;
            smin = fminf(smin, s.y);
            z = (z + step) + step;
        }
    }
    X = zbest * r.a1[0] + t1p[0];
    Y = zbest * r.a1[1] + t1p[1];
    Z = zbest * r.a1[2] + t1p[2];
}

__global__ __launch_bounds__(256) void tri_points_kernel(TriCam c1, TriCam c2, bool cams_fast,
                                                         const int4* __restrict__ pts, int n, float* __restrict__ xyz)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 p = pts[i];
    float X, Y, Z;
    triangulate_point(c1, c2, cams_fast, (float)p.x, (float)p.y, (float)p.z, (float)p.w, X, Y, Z);
    xyz[3 * (size_t)i] = X; xyz[3 * (size_t)i + 1] = Y; xyz[3 * (size_t)i + 2] = Z;
}

#define TRI_COUNT_SLOTS 64
__global__ __launch_bounds__(256) void tri_disparity_kernel(TriCam c1, TriCam c2, bool cams_fast,
                                                            const float* __restrict__ disp,
                                                            const float* __restrict__ peak, float min_peak,
                                                            bool subpixel, int W, int row0, int frame_row0,
                                                            float* __restrict__ xyz,
                                                            unsigned long long* __restrict__ count)
{
    // grid.y rows starting at row row0 of the rasters; raster row k is row k + frame_row0 of the frame the cameras
    // describe (a strip of a frame held as a raster of its own)
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = row0 + blockIdx.y;
    bool ok = false;
    float X = NAN, Y = NAN, Z = NAN;
    if (x < W) {
        size_t o = (size_t)y * W + x;
        float d = disp[o];
        ok = (d != (float)MIS_INVALID_DISP) && (peak[o] >= min_peak);
        if (ok) {
            // the reference hands integer pixels to calculateXYZ (image3.cpp:812-824): the disparity is rounded;
            // subpixel: the fractional column x - d goes into the same arithmetic
            const float u2 = subpixel ? (float)x - d : (float)(x - (int)rintf(d));
            const float v = (float)(y + frame_row0);
            triangulate_point(c1, c2, cams_fast, (float)x, v, u2, v, X, Y, Z);
        }
        xyz[3 * o] = X; xyz[3 * o + 1] = Y; xyz[3 * o + 2] = Z;
    }
    // One atomic per workgroup, spread over TRI_COUNT_SLOTS counters in separate 128-byte lines: a
    // single device-wide counter serialises about a million same-address atomics per frame in the
    // memory system and ends up longer than the arithmetic.
    __shared__ unsigned s_cnt[4];
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) % TRI_COUNT_SLOTS;
        if (n) atomicAdd(count + slot * 16, (unsigned long long)n);
    }
}

// Sum of the spread counters -> one device word (the count of the entry point that allocates nothing).
__global__ __launch_bounds__(64) void tri_count_sum_kernel(const unsigned long long* __restrict__ count, long long* __restrict__ out)
{
    unsigned long long v = count[threadIdx.x * 16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (threadIdx.x == 0) *out = (long long)v;
}

// Counter slots of mis_triangulate_disparity_noalloc: device memory that exists from the moment the
// library is loaded, one copy per GPU, held until the process ends (8 KB).  In host code the name
// below is only a handle for hipGetSymbolAddress; kernels and memsets get the address looked up by
// mis_tri_init_device for the device they run on.
static __device__ __align__(256) unsigned long long scratch[TRI_COUNT_SLOTS * 16];
#define TRI_MAX_DEVICES 64
#define TRI_MAX_GRID_ROWS 65535
static unsigned long long* g_tri_scratch[TRI_MAX_DEVICES];      // per device, filled by mis_tri_init_device
static std::mutex g_tri_mu;

// called by mis_create with ctx->device current
int mis_tri_init_device(mis_ctx* ctx)
{
    if (ctx->device < 0 || ctx->device >= TRI_MAX_DEVICES) return MIS_OK;     // the no-alloc entry point reports it
    std::lock_guard<std::mutex> g(g_tri_mu);
    if (!g_tri_scratch[ctx->device]) {
        void* p = nullptr;
        MIS_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_tri_scratch[ctx->device] = (unsigned long long*)p;
    }
    return MIS_OK;
}

static int tri_check(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!c1 || !c2) return mis_fail(ctx, MIS_ERR_ARG, "triangulate: camera is NULL");
    return MIS_OK;
}

extern "C" {

int mis_triangulate_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                           const int32_t* d_pts, int n, float* d_xyz)
{
    int rc = tri_check(ctx, c1, c2);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_pts || !d_xyz))) return mis_fail(ctx, MIS_ERR_ARG, "triangulate: bad point buffers");
    if (n == 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    TriCam a = make_cam(c1), b = make_cam(c2);
    hipLaunchKernelGGL(tri_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a, b,
                       tri_cams_fast(a, b), (const int4*)d_pts, n, d_xyz);
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

int mis_triangulate(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                    const int32_t* pts, int n, float* xyz)
{
    int rc = tri_check(ctx, c1, c2);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!pts || !xyz))) return mis_fail(ctx, MIS_ERR_ARG, "triangulate: bad point buffers");
    if (n == 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    size_t need = (size_t)n * 16 + (size_t)n * 12;
    rc = mis_reserve(ctx, &ctx->tri_buf, &ctx->tri_bytes, need);
    if (rc) return rc;
    int32_t* dp = (int32_t*)ctx->tri_buf;
    float* dx = (float*)((char*)ctx->tri_buf + (size_t)n * 16);
    MIS_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    rc = mis_triangulate_device(ctx, c1, c2, dp, n, dx);
    if (rc) return rc;
    MIS_HIP(ctx, hipMemcpyAsync(xyz, dx, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIS_OK;
}

int mis_triangulate_disparity_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                     const float* d_disp, const float* d_peak, float min_peak,
                                     float* d_xyz, long long* n_matched)
{
    return mis_triangulate_disparity_ex_device(ctx, c1, c2, d_disp, d_peak, min_peak, 0, d_xyz, n_matched);
}

int mis_triangulate_disparity_ex_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                        const float* d_disp, const float* d_peak, float min_peak, int flags,
                                        float* d_xyz, long long* n_matched)
{
    int rc = tri_check(ctx, c1, c2);
    if (rc) return rc;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "triangulate: no image pair bound (raster size unknown)");
    return mis_triangulate_disparity_rows_device(ctx, c1, c2, d_disp, d_peak, min_peak, flags, ctx->width, 0, ctx->height, 0,
                                                 d_xyz, n_matched);
}

int mis_triangulate_disparity_rows_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                          const float* d_disp, const float* d_peak, float min_peak, int flags,
                                          int width, int row0, int nrows, int frame_row0,
                                          float* d_xyz, long long* n_matched)
{
    int rc = tri_check(ctx, c1, c2);
    if (rc) return rc;
    if (!d_disp || !d_peak || !d_xyz) return mis_fail(ctx, MIS_ERR_ARG, "triangulate: NULL buffer");
    if (flags & ~MIS_TRI_SUBPIXEL) return mis_fail(ctx, MIS_ERR_ARG, "triangulate: unknown flags 0x%x", flags);
    if (width <= 0 || row0 < 0 || nrows < 0 || frame_row0 < 0 || (long long)row0 + nrows + frame_row0 > 0x7FFFFFFF)
        return mis_fail(ctx, MIS_ERR_ARG, "triangulate: rows [%d,%d) of a raster %d wide, frame row offset %d", row0, row0 + nrows,
                        width, frame_row0);
    if (nrows == 0) { if (n_matched) *n_matched = 0; return MIS_OK; }
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cnt_bytes = (size_t)TRI_COUNT_SLOTS * 128;
    rc = mis_reserve(ctx, &ctx->tri_buf, &ctx->tri_bytes, cnt_bytes);
    if (rc) return rc;
    unsigned long long* cnt = (unsigned long long*)ctx->tri_buf;
    MIS_HIP(ctx, hipMemsetAsync(cnt, 0, cnt_bytes, ctx->stream));
    TriCam a = make_cam(c1), b = make_cam(c2);
    for (int done = 0; done < nrows; done += TRI_MAX_GRID_ROWS) {      // a launch covers at most 65535 rows (grid.y)
        const int slab = nrows - done < TRI_MAX_GRID_ROWS ? nrows - done : TRI_MAX_GRID_ROWS;
        hipLaunchKernelGGL(tri_disparity_kernel, dim3((width + 255) / 256, slab), dim3(256), 0, ctx->stream, a, b,
                           tri_cams_fast(a, b), d_disp, d_peak, min_peak, (flags & MIS_TRI_SUBPIXEL) != 0, width, row0 + done, frame_row0,
                           d_xyz, cnt);
    }
    MIS_HIP(ctx, hipGetLastError());
    if (n_matched) {
        unsigned long long h[TRI_COUNT_SLOTS * 16];
        MIS_HIP(ctx, hipMemcpyAsync(h, cnt, cnt_bytes, hipMemcpyDeviceToHost, ctx->stream));
        MIS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long total = 0;
        for (int i = 0; i < TRI_COUNT_SLOTS; i++) total += h[i * 16];
        *n_matched = (long long)total;
    }
    return MIS_OK;
}

int mis_triangulate_disparity_noalloc(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                      const float* d_disp, const float* d_peak, float min_peak,
                                      int width, int height, float* d_xyz, long long* d_n_matched)
{
    int rc = tri_check(ctx, c1, c2);
    if (rc) return rc;
    if (!d_disp || !d_peak || !d_xyz) return mis_fail(ctx, MIS_ERR_ARG, "triangulate (no-alloc): NULL buffer");
    if (width <= 0 || height <= 0 || width > MIS_NOALLOC_MAX_WIDTH || height > MIS_NOALLOC_MAX_HEIGHT)
        return mis_fail(ctx, MIS_ERR_ARG, "triangulate (no-alloc): raster %d x %d outside the supported 1..%d x 1..%d",
                        width, height, MIS_NOALLOC_MAX_WIDTH, MIS_NOALLOC_MAX_HEIGHT);
    if (ctx->device < 0 || ctx->device >= TRI_MAX_DEVICES)
        return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "triangulate (no-alloc): device %d beyond the %d devices this entry point keeps scratch for",
                        ctx->device, TRI_MAX_DEVICES);
    unsigned long long* cnt = g_tri_scratch[ctx->device];       // looked up once per device by mis_create
    if (!cnt) return mis_fail(ctx, MIS_ERR_STATE, "triangulate (no-alloc): scratch address of device %d unknown", ctx->device);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    // the array keeps the previous call's counts
    MIS_HIP(ctx, hipMemsetAsync(cnt, 0, (size_t)TRI_COUNT_SLOTS * 128, ctx->stream));
    TriCam a = make_cam(c1), b = make_cam(c2);
    for (int done = 0; done < height; done += TRI_MAX_GRID_ROWS) {
        const int slab = height - done < TRI_MAX_GRID_ROWS ? height - done : TRI_MAX_GRID_ROWS;
        hipLaunchKernelGGL(tri_disparity_kernel, dim3((width + 255) / 256, slab), dim3(256), 0, ctx->stream, a, b,
                           tri_cams_fast(a, b), d_disp, d_peak, min_peak, false, width, done, 0, d_xyz, cnt);
    }
    if (d_n_matched)
        hipLaunchKernelGGL(tri_count_sum_kernel, dim3(1), dim3(64), 0, ctx->stream, cnt, d_n_matched);
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}

}  // extern "C"
