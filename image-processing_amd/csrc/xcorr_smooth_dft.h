// 1-D DFTs of lengths whose prime factors are all <= 13 (8000 = 2^6 5^3, 6000 = 2^4 3 5^3, 7000 = 2^3 5^3 7 ...: the
// sides of most photographs), computed directly as a mixed-radix transform in LDS instead of the chirp-z detour through a
// power of two more than twice as long (xg_dft_rows in xcorr_kernels.hip, which keeps every other length).
// Included by xcorr_kernels.hip (uses its cmul and xp).
//
// Autosort (Stockham) formulation: pass q of radix R with Ns = product of the radices before it takes, for butterfly
// j in [0, n/R), k = j mod Ns,
//     v[r]  = in[j + r n/R] * exp(-2 pi i r k / (Ns R)),   r = 0..R-1
//     V     = DFT_R(v)
//     out[(j / Ns) Ns R + k + r Ns] = V[r],
// and the last pass leaves the spectrum in natural order, whatever the order of the radices.  A thread holds all of its
// butterflies' inputs in registers (n / NT <= 8 elements), a barrier separates every read of a pass from every write, so one
// array of n elements (padded as xp does) serves as input and output of every pass; the first pass reads global memory and the last one
// writes it, both with consecutive lanes on consecutive elements.
#pragma once

#define XS_MAX_PASSES 16
struct XsPlan {
    int n, npass;
    unsigned long long radices;         // four bits per pass, the first pass in the lowest bits (a kernel argument that is
                                        // indexed at run time would be copied to private memory)
    const float2* circle;               // n values exp(-2 pi i t / n)
};

// sin and cos of 2 pi q / R as compile-time constants (Taylor series in double on an argument reduced to [-pi, pi])
constexpr double xs_sin_series(double x)
{
    double t = x, s = x;
    for (int i = 1; i < 20; i++) { t *= -x * x / (double)((2 * i) * (2 * i + 1)); s += t; }
    return s;
}
constexpr double xs_cos_series(double x)
{
    double t = 1.0, s = 1.0;
    for (int i = 1; i < 20; i++) { t *= -x * x / (double)((2 * i - 1) * (2 * i)); s += t; }
    return s;
}
template <int R>
struct XsRoots {
    float c[R], s[R];
    constexpr XsRoots() : c{}, s{}
    {
        for (int q = 0; q < R; q++) {
            const int qq = 2 * q > R ? q - R : q;
            const double a = 2.0 * 3.14159265358979323846 * (double)qq / (double)R;
            c[q] = (float)xs_cos_series(a);
            s[q] = (float)xs_sin_series(a);
        }
    }
};

__device__ __forceinline__ float2 xs_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 xs_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 xs_mulmi(float2 a) { return make_float2(a.y, -a.x); }      // a * (-i)

__device__ __forceinline__ void xs_dft4(float2& u0, float2& u1, float2& u2, float2& u3)
{
    const float2 s0 = xs_add(u0, u2), s1 = xs_sub(u0, u2), s2 = xs_add(u1, u3), s3 = xs_mulmi(xs_sub(u1, u3));
    u0 = xs_add(s0, s2); u2 = xs_sub(s0, s2);
    u1 = xs_add(s1, s3); u3 = xs_sub(s1, s3);
}

// forward DFT of R values in place (exp(-2 pi i m k / R))
template <int R>
__device__ __forceinline__ void xs_bfly(float2 (&v)[R])
{
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = xs_add(a, b); v[1] = xs_sub(a, b);
    } else if constexpr (R == 4) {
        xs_dft4(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 8) {
        // decimation in frequency: even outputs from the sums, odd ones from the differences turned by eighth roots
        constexpr float H = 0.70710678118654752440f;
        float2 a0 = xs_add(v[0], v[4]), a1 = xs_add(v[1], v[5]), a2 = xs_add(v[2], v[6]), a3 = xs_add(v[3], v[7]);
        float2 b0 = xs_sub(v[0], v[4]), b1 = xs_sub(v[1], v[5]), b2 = xs_sub(v[2], v[6]), b3 = xs_sub(v[3], v[7]);
        b1 = make_float2(H * (b1.x + b1.y), H * (b1.y - b1.x));          // * exp(-i pi / 4)
        b2 = xs_mulmi(b2);
        b3 = make_float2(H * (b3.y - b3.x), -H * (b3.x + b3.y));         // * exp(-3 i pi / 4)
        xs_dft4(a0, a1, a2, a3);
        xs_dft4(b0, b1, b2, b3);
        v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
        v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
    } else {
        // odd R: pairs (k, R - k) as sums and differences, X[m], X[R-m] = (v0 + sum a_k cos) -+ i (sum b_k sin)
        static_assert(R % 2 == 1, "radix");
        constexpr int Hn = (R - 1) / 2;
        constexpr XsRoots<R> roots{};
        float2 a[Hn], b[Hn];
#pragma unroll
        for (int k = 1; k <= Hn; k++) { a[k - 1] = xs_add(v[k], v[R - k]); b[k - 1] = xs_sub(v[k], v[R - k]); }
        const float2 v0 = v[0];
        float2 sum = v0;
#pragma unroll
        for (int k = 0; k < Hn; k++) sum = xs_add(sum, a[k]);
        v[0] = sum;
#pragma unroll
        for (int m = 1; m <= Hn; m++) {
            float2 t = v0, s = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 1; k <= Hn; k++) {
                const float c = roots.c[(m * k) % R], sn = roots.s[(m * k) % R];
                t.x = fmaf(a[k - 1].x, c, t.x); t.y = fmaf(a[k - 1].y, c, t.y);
                s.x = fmaf(b[k - 1].x, sn, s.x); s.y = fmaf(b[k - 1].y, sn, s.y);
            }
            v[m] = make_float2(t.x + s.y, t.y - s.x);
            v[R - m] = make_float2(t.x - s.y, t.y + s.x);
        }
    }
}

// One pass.  src / dst: the sequence in global memory (read by the first pass, written by the last); x: the LDS array.
template <int R, int NT, bool PAD>
__device__ __forceinline__ void xs_pass(const float2* __restrict__ src, float2* __restrict__ dst, float2* x, int n, int tid, int Ns,
                                        const float2* __restrict__ circle, bool first, bool last, float sgn)
{
    constexpr int MAXB = (8 + R - 1) / R;          // butterflies per thread: n / NT <= 8 elements
    const int nb = n / R;
    const int step = nb / Ns;                      // circle index of exp(-2 pi i / (Ns R))
    const float rns = __frcp_rn((float)Ns);
    float2 v[MAXB][R], w1[MAXB];
    int kk[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        const int j = min(tid + b * NT, nb - 1);
        int k = j - (int)((float)j * rns) * Ns;        // j mod Ns: the float quotient is off by one at most (j, Ns < 2^14)
        k = k < 0 ? k + Ns : (k >= Ns ? k - Ns : k);
        kk[b] = first ? 0 : k;
        if (!first) w1[b] = circle[kk[b] * step];
    }
    if (first) {
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            const int j = min(tid + b * NT, nb - 1);
#pragma unroll
            for (int r = 0; r < R; r++) { v[b][r] = src[j + r * nb]; v[b][r].y *= sgn; }
        }
    } else {
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            const int j = min(tid + b * NT, nb - 1);
#pragma unroll
            for (int r = 0; r < R; r++) v[b][r] = x[xp<PAD>(j + r * nb)];
        }
        __syncthreads();                           // every read of this pass before any write of it
        // exp(-2 pi i r k / (Ns R)) as powers of the one value fetched, each from the two powers of half its exponent
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            float2 w[R];
            w[1] = w1[b];
#pragma unroll
            for (int r = 2; r < R; r++) w[r] = cmul(w[r / 2], w[r - r / 2]);
#pragma unroll
            for (int r = 1; r < R; r++) v[b][r] = cmul(v[b][r], w[r]);
        }
    }
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        xs_bfly<R>(v[b]);
        const int j = tid + b * NT;
        if (j < nb) {
            const int j0 = (j - kk[b]) * R + kk[b];
            if (last) {
#pragma unroll
                for (int r = 0; r < R; r++) dst[j0 + r * Ns] = make_float2(v[b][r].x, sgn * v[b][r].y);
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) x[xp<PAD>(j0 + r * Ns)] = v[b][r];
            }
        }
    }
    if (!last) __syncthreads();
}

// In-place DFT of sequences of length p.n; sequence r lives at data + (r / per_batch) * batch_stride + (r % per_batch) * n.
// inverse != 0: unnormalised backward transform (conjugate, forward, conjugate).  NT >= n / 8.
template <int NT, bool PAD>
__global__ __launch_bounds__(NT, 8) void xs_dft_rows(float2* __restrict__ data, XsPlan p, int per_batch, size_t batch_stride,
                                                   int inverse)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_smem[];
    float2* x = (float2*)xs_smem;
    const int r = blockIdx.x;
    float2* row = data + (size_t)(r / per_batch) * batch_stride + (size_t)(r % per_batch) * p.n;
    const float sgn = inverse ? -1.f : 1.f;
    int Ns = 1;
    for (int q = 0; q < p.npass; q++) {
        const int R = (int)((p.radices >> (4 * q)) & 15ull);
        const bool first = q == 0, last = q == p.npass - 1;
        // the length and the thread index behind an opaque statement in every pass: what the passes derive from them
        // (element indices, LDS addresses of every radix) would otherwise all be computed in front of the loop and
        // held in registers across it
        int n_q = p.n, tid = (int)threadIdx.x;
        asm volatile("" : "+s"(n_q), "+v"(tid));
        switch (R) {
#define XS_CASE(RV) case RV: xs_pass<RV, NT, PAD>(row, row, x, n_q, tid, Ns, p.circle, first, last, sgn); break;
            XS_CASE(2) XS_CASE(3) XS_CASE(4) XS_CASE(5) XS_CASE(7) XS_CASE(8) XS_CASE(11) XS_CASE(13)
#undef XS_CASE
        }
        Ns *= R;
    }
}

// radices of n (8s, then 4 or 2, then the odd primes up to 13); false if another prime factor remains
static bool xs_factor(int n, XsPlan* p)
{
    p->n = n; p->npass = 0; p->radices = 0;
    static const int cand[] = {8, 4, 2, 3, 5, 7, 11, 13};
    for (int c : cand)
        while (n % c == 0 && n > 1) {
            if (p->npass == XS_MAX_PASSES) return false;
            p->radices |= (unsigned long long)c << (4 * p->npass);
            p->npass++;
            n /= c;
        }
    return n == 1 && p->npass >= 1;
}
