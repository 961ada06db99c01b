/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Dense windowed zero-mean normalised cross-correlation over a horizontal disparity range.
 * The reference has no function for this: it is the dense limit its disabled "Calculate dense
 * points" block (image3.cpp:775-811) and its TODO (image3.cpp:866-870) point at, with the
 * sign convention of the reference's correlation (image3.cpp:392-398: a peak at offset k means
 * image-2 content at p appears in image 1 at p + k) and its first-maximum tie-break
 * (image3.cpp:486-501).  PARITY UNPINNED by any reference test.
 *
 * Definition, for left pixel (x,y), W x W window, r = W/2, disparity d in [dmin,dmax]:
 *   valid pixel:  r <= x < width-r, r <= y < height-r
 *   valid d:      r <= x-d < width-r
 *   score(d) = sum((L-mL)(R_d-mR_d)) / sqrt(sum((L-mL)^2) * sum((R_d-mR_d)^2)), windows centred
 *              at (x,y) in L and (x-d,y) in R; 0 when either variance is 0.
 *   All sums are exact integers (int64); the quotient is evaluated once in double:
 *     num = N*S_LR - S_L*S_R, varL = N*S_LL - S_L^2, varR likewise, score = num/sqrt(varL*varR).
 *   best = argmax over valid d, lowest d wins ties.  peak = score(best).
 *   subpixel: delta = (s[-1]-s[+1]) / (2 (s[-1] - 2 s[0] + s[+1])), clipped to +-0.5; 0 when
 *   best-1 or best+1 is not a valid d or the denominator is 0.   disp = best + delta.
 *   peak2 = max score over valid d with |d-best| > 1; -1 when there is none.
 *   Invalid pixels (and pixels with no valid d): disp = idisp = -32768, peak = 0, peak2 = -1.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads used by the OpenMP regions started from the calling thread (1: a serial run) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}

static inline int64_t px_at(const void* img, int dtype, size_t stride, int x, int y)
{
    const unsigned char* row = (const unsigned char*)img + (size_t)y * stride;
    return dtype == ORC_U8 ? (int64_t)row[x] : (int64_t)((const uint16_t*)row)[x];
}

double orc_ncc_score(const void* L, const void* R, int dtype, int width, int height, size_t stride,
                     int window, int x, int y, int d)
{
    int r = window / 2;
    if (x < r || x >= width - r || y < r || y >= height - r) return NAN;
    int xr = x - d;
    if (xr < r || xr >= width - r) return NAN;
    int64_t N = (int64_t)window * window, sl = 0, sr = 0, sll = 0, srr = 0, slr = 0;
    for (int j = -r; j <= r; j++)
        for (int i = -r; i <= r; i++) {
            int64_t a = px_at(L, dtype, stride, x + i, y + j);
            int64_t b = px_at(R, dtype, stride, xr + i, y + j);
            sl += a; sr += b; sll += a * a; srr += b * b; slr += a * b;
        }
    int64_t num = N * slr - sl * sr, vl = N * sll - sl * sl, vr = N * srr - sr * sr;
    if (vl <= 0 || vr <= 0) return 0.0;
    return (double)num / sqrt((double)vl * (double)vr);
}

/* orc_ncc_score for n (x, y, d) triples */
void orc_ncc_scores(const void* L, const void* R, int dtype, int width, int height, size_t stride,
                    int window, int n, const int* xs, const int* ys, const int* ds, double* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        out[i] = orc_ncc_score(L, R, dtype, width, height, stride, window, xs[i], ys[i], ds[i]);
}

/* summed-area table with one guard row/column: sat[(y+1)*(w+1) + (x+1)] = sum over [0..x]x[0..y] */
static void build_sat(const void* img, int dtype, size_t stride, int w, int h, int square, int64_t* sat)
{
    size_t W1 = (size_t)w + 1;
    memset(sat, 0, sizeof(int64_t) * W1);
    for (int y = 0; y < h; y++) {
        int64_t run = 0;
        sat[(size_t)(y + 1) * W1] = 0;
        for (int x = 0; x < w; x++) {
            int64_t v = px_at(img, dtype, stride, x, y);
            run += square ? v * v : v;
            sat[(size_t)(y + 1) * W1 + x + 1] = sat[(size_t)y * W1 + x + 1] + run;
        }
    }
}

static inline int64_t box(const int64_t* sat, int w, int x0, int y0, int x1, int y1)
{
    size_t W1 = (size_t)w + 1;  /* inclusive corners */
    return sat[(size_t)(y1 + 1) * W1 + x1 + 1] - sat[(size_t)y0 * W1 + x1 + 1]
         - sat[(size_t)(y1 + 1) * W1 + x0] + sat[(size_t)y0 * W1 + x0];
}

int orc_ncc_dense(const void* L, const void* R, int dtype, int width, int height, size_t stride,
                  int window, int dmin, int dmax,
                  float* disp, float* peak, int16_t* idisp, float* peak2, double* volume)
{
    if ((dtype != ORC_U8 && dtype != ORC_U16) || window < 1 || !(window & 1) || dmax < dmin) return 1;
    const int r = window / 2, nd = dmax - dmin + 1;
    const int64_t N = (int64_t)window * window;
    const size_t npx = (size_t)width * height;

    for (size_t i = 0; i < npx; i++) {
        if (disp) disp[i] = (float)ORC_INVALID_DISP;
        if (idisp) idisp[i] = (int16_t)ORC_INVALID_DISP;
        if (peak) peak[i] = 0.0f;
        if (peak2) peak2[i] = -1.0f;
    }
    if (volume) for (size_t i = 0; i < npx * nd; i++) volume[i] = NAN;
    if (width < window || height < window) return 0;

    size_t satn = ((size_t)width + 1) * ((size_t)height + 1);
    int64_t* satL = (int64_t*)malloc(sizeof(int64_t) * satn);
    int64_t* satLL = (int64_t*)malloc(sizeof(int64_t) * satn);
    int64_t* satR = (int64_t*)malloc(sizeof(int64_t) * satn);
    int64_t* satRR = (int64_t*)malloc(sizeof(int64_t) * satn);
    build_sat(L, dtype, stride, width, height, 0, satL);
    build_sat(L, dtype, stride, width, height, 1, satLL);
    build_sat(R, dtype, stride, width, height, 0, satR);
    build_sat(R, dtype, stride, width, height, 1, satRR);

    /* Row strips in parallel; inside a strip, for every d a column-sum vector of the product
     * image slides down the rows and a horizontal pass turns it into window sums. */
#pragma omp parallel
    {
        int64_t* col = (int64_t*)malloc(sizeof(int64_t) * (size_t)width);
        double* cur = (double*)malloc(sizeof(double) * (size_t)width);   /* score(d) on this row */
#pragma omp for schedule(dynamic, 1)
        for (int ys = r; ys < height - r; ys += 32) {
            int ye = ys + 32 < height - r ? ys + 32 : height - r;
            int rows = ye - ys;
            size_t n = (size_t)rows * width;
            double* best = (double*)malloc(sizeof(double) * n);
            double* sm = (double*)malloc(sizeof(double) * n);   /* score at best-1 (NaN: none) */
            double* sp = (double*)malloc(sizeof(double) * n);   /* score at best+1 (NaN: none) */
            double* prev = (double*)malloc(sizeof(double) * n); /* score at d-1 (NaN: invalid) */
            double* second = peak2 ? (double*)malloc(sizeof(double) * n) : NULL;
            int* bd = (int*)malloc(sizeof(int) * n);
            for (size_t i = 0; i < n; i++) { best[i] = -INFINITY; sm[i] = sp[i] = prev[i] = NAN; bd[i] = dmin - 1; }

            for (int pass = 0; pass < (peak2 ? 2 : 1); pass++) {
                if (pass == 1) for (size_t i = 0; i < n; i++) second[i] = -INFINITY;
                for (int d = dmin; d <= dmax; d++) {
                    /* L column x pairs with R column x-d; usable columns: both inside the raster */
                    int xa = d > 0 ? d : 0, xb = d < 0 ? width + d : width;   /* [xa, xb) */
                    if (xb - xa < window) {
                        if (pass == 0) for (size_t i = 0; i < n; i++) prev[i] = NAN;
                        continue;
                    }
                    for (int x = xa; x < xb; x++) {
                        int64_t s = 0;
                        for (int j = ys - r; j < ys + r; j++)   /* rows ys-r .. ys+r-1; +1 row added below */
                            s += px_at(L, dtype, stride, x, j) * px_at(R, dtype, stride, x - d, j);
                        col[x] = s;
                    }
                    for (int y = ys; y < ye; y++) {
                        for (int x = xa; x < xb; x++) {
                            col[x] += px_at(L, dtype, stride, x, y + r) * px_at(R, dtype, stride, x - d, y + r);
                        }
                        double* rowbest = best + (size_t)(y - ys) * width;
                        size_t rowoff = (size_t)(y - ys) * width;
                        int x0 = (xa + r > r) ? xa + r : r;              /* first centre with both windows inside */
                        int x1 = (xb - r < width - r) ? xb - r : width - r; /* exclusive */
                        int64_t s = 0;
                        for (int x = x0 - r; x < x0 + r; x++) s += col[x];
                        for (int x = 0; x < width; x++) cur[x] = NAN;
                        for (int x = x0; x < x1; x++) {
                            s += col[x + r];
                            int64_t sl = box(satL, width, x - r, y - r, x + r, y + r);
                            int64_t sll = box(satLL, width, x - r, y - r, x + r, y + r);
                            int64_t sr = box(satR, width, x - d - r, y - r, x - d + r, y + r);
                            int64_t srr = box(satRR, width, x - d - r, y - r, x - d + r, y + r);
                            int64_t num = N * s - sl * sr, vl = N * sll - sl * sl, vr = N * srr - sr * sr;
                            cur[x] = (vl <= 0 || vr <= 0) ? 0.0 : (double)num / sqrt((double)vl * (double)vr);
                            s -= col[x - r];
                        }
                        if (pass == 0) {
                            for (int x = 0; x < width; x++) {
                                double sc = cur[x];
                                size_t i = rowoff + x;
                                if (!isnan(sc)) {
                                    if (volume) volume[((size_t)y * width + x) * nd + (d - dmin)] = sc;
                                    if (sc > rowbest[x]) {
                                        rowbest[x] = sc; bd[i] = d; sm[i] = prev[i]; sp[i] = NAN;
                                    } else if (d == bd[i] + 1) {
                                        sp[i] = sc;
                                    }
                                }
                                prev[i] = sc;
                            }
                        } else {
                            for (int x = 0; x < width; x++) {
                                double sc = cur[x];
                                size_t i = rowoff + x;
                                if (!isnan(sc) && bd[i] >= dmin && abs(d - bd[i]) > 1 && sc > second[i]) second[i] = sc;
                            }
                        }
                        for (int x = xa; x < xb; x++)
                            col[x] -= px_at(L, dtype, stride, x, y - r) * px_at(R, dtype, stride, x - d, y - r);
                    }
                }
            }
            for (int y = ys; y < ye; y++)
                for (int x = r; x < width - r; x++) {
                    size_t i = (size_t)(y - ys) * width + x, o = (size_t)y * width + x;
                    if (bd[i] < dmin) continue;
                    double delta = 0.0;
                    if (!isnan(sm[i]) && !isnan(sp[i])) {
                        double den = 2.0 * (sm[i] - 2.0 * best[i] + sp[i]);
                        if (den != 0.0) {
                            delta = (sm[i] - sp[i]) / den;
                            if (delta > 0.5) delta = 0.5;
                            if (delta < -0.5) delta = -0.5;
                        }
                    }
                    if (disp) disp[o] = (float)((double)bd[i] + delta);
                    if (idisp) idisp[o] = (int16_t)bd[i];
                    if (peak) peak[o] = (float)best[i];
                    if (peak2) peak2[o] = isinf(second[i]) ? -1.0f : (float)second[i];
                }
            free(best); free(sm); free(sp); free(prev); free(second); free(bd);
        }
        free(col); free(cur);
    }
    free(satL); free(satLL); free(satR); free(satRR);
    return 0;
}
