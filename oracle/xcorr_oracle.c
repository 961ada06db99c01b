/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Tile cross-correlation with difference-of-Gaussian peak pick: image3.cpp:318-335, 346-517;
 * whole-image correlation surface: image-cross-cor.cpp:83-176.
 *
 * Third-party arithmetic restated from its published behaviour (libraries absent here):
 *   FFTW 3 single precision (unnormalised DFT) -> fft_oracle.c;
 *   OpenCV 2.4.2 cv::GaussianBlur(151x151, sigma 99, BORDER_REFLECT_101) -> orc_gaussian_blur
 *     (kernel: getGaussianKernel; row pass: taps accumulated left to right; column pass:
 *      centre tap first, then symmetric pairs outward; from the 2.4 sources as remembered,
 *      not verifiable offline);
 *   GDAL RasterIO pixel -> CFloat32 conversion: value goes to the real part, imaginary 0. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

double orc_load_px(const void* img, int dtype, size_t stride_bytes, int x, int y);

/* OpenCV borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p - 1 + 1;
        else p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* OpenCV getGaussianKernel(n, sigma, CV_32F) for sigma > 0 */
static void gaussian_kernel(float* k, int n, double sigma)
{
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2x * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/* image3.cpp:481: cv::GaussianBlur(imout, imout_g1, Size(ksize, ksize), sigma) on CV_32F */
void orc_gaussian_blur(const float* src, float* dst, int width, int height, int ksize, double sigma)
{
    int kw = (width == 1) ? 1 : ksize;   /* GaussianBlur collapses a 1-px dimension */
    int kh = (height == 1) ? 1 : ksize;
    float* kx = (float*)malloc(sizeof(float) * kw);
    float* ky = (float*)malloc(sizeof(float) * kh);
    gaussian_kernel(kx, kw, sigma);
    gaussian_kernel(ky, kh, sigma);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)width * height);
    int rx = kw / 2, ry = kh / 2;
#pragma omp parallel
    {
        int* idx = (int*)malloc(sizeof(int) * (size_t)(width + kw));
        for (int i = 0; i < width + kw - 1; i++) idx[i] = reflect101(i - rx, width);
#pragma omp for schedule(static)
        for (int y = 0; y < height; y++) {
            const float* s = src + (size_t)y * width;
            float* t = tmp + (size_t)y * width;
            for (int x = 0; x < width; x++) {
                float acc = 0.0f;
                for (int k = 0; k < kw; k++) {
                    float prod = s[idx[x + k]] * kx[k]; /* built with -ffp-contract=off: no fma */
                    acc = acc + prod;
                }
                t[x] = acc;
            }
        }
#pragma omp for schedule(static)
        for (int y = 0; y < height; y++) {
            float* d = dst + (size_t)y * width;
            for (int x = 0; x < width; x++) {
                float c0 = ky[ry] * tmp[(size_t)y * width + x];
                float acc = c0;
                for (int k = 1; k <= ry; k++) {
                    int ya = reflect101(y + k, height), yb = reflect101(y - k, height);
                    float pair = tmp[(size_t)ya * width + x] + tmp[(size_t)yb * width + x];
                    float prod = ky[ry + k] * pair;
                    acc = acc + prod;
                }
                d[x] = acc;
            }
        }
        free(idx);
    }
    free(tmp); free(kx); free(ky);
}

/* image3.cpp:318-335 runFFT, buffer fill part: zero the w*h tile, then copy the bbox region
 * (pixels outside the raster stay zero, as a failed/partial RasterIO leaves them). */
static void load_tile(double* tile_re, const void* img, int iw, int ih, size_t stride, int dtype,
                      orc_rect b, int width, int height)
{
    memset(tile_re, 0, sizeof(double) * (size_t)width * height);
    for (int y = 0; y < b.height && y < height; y++) {
        int sy = b.miny + y;
        if (sy < 0 || sy >= ih) continue;
        for (int x = 0; x < b.width && x < width; x++) {
            int sx = b.minx + x;
            if (sx < 0 || sx >= iw) continue;
            tile_re[(size_t)y * width + x] = orc_load_px(img, dtype, stride, sx, sy);
        }
    }
}

/* Correlation core shared by image3 findOffset (image3.cpp:383-416) and image-cross-cor CalcFFT
 * (image-cross-cor.cpp:135-168): returns the normalised |c| surface, values in [0,1].
 * Returns 0 on success, 1 when max == min (the 0/0 case of image3.cpp:415). */
static int corr_surface(const double* t1, const double* t2, int width, int height, int precision,
                        float* out, double* outd, float* min_out, float* max_out)
{
    size_t n = (size_t)width * height;
    float fmin, fmax;
    if (precision == 0) {
        float* a = (float*)malloc(sizeof(float) * 2 * n);
        float* b = (float*)malloc(sizeof(float) * 2 * n);
        for (size_t i = 0; i < n; i++) { a[2*i] = (float)t1[i]; a[2*i+1] = 0; b[2*i] = (float)t2[i]; b[2*i+1] = 0; }
        orc_fft2d_f32(a, width, height, -1);
        orc_fft2d_f32(b, width, height, -1);
        /* img[i] = img[i] / norm, norm = (complex float)csqrt(px_count): complex division
         * (a+bi)/(c+0i) = ((a*c)/(c*c), (b*c)/(c*c)) in float (image3.cpp:331-334) */
        float c = (float)sqrt((double)n);
        float cc = c * c;
        for (size_t i = 0; i < 2 * n; i++) {
            float pa = a[i] * c, pb = b[i] * c;
            a[i] = pa / cc; b[i] = pb / cc;
        }
        /* img2 = img1 * conj(img2) (image3.cpp:392-394), float products, no fma */
        for (size_t i = 0; i < n; i++) {
            float ac = a[2*i] * b[2*i], bd = a[2*i+1] * b[2*i+1];
            float bc = a[2*i+1] * b[2*i], ad = a[2*i] * b[2*i+1];
            b[2*i] = ac + bd;
            b[2*i+1] = bc - ad;
        }
        orc_fft2d_f32(b, width, height, +1);   /* image3.cpp:398 */
        for (size_t i = 0; i < 2 * n; i++) { float p = b[i] * c; b[i] = p / cc; }
        /* min/max of cabs() (double hypot of the float pair), kept as float (image3.cpp:402-411) */
        fmin = fmax = (float)hypot((double)b[0], (double)b[1]);
        for (size_t i = 0; i < n; i++) {
            double m = hypot((double)b[2*i], (double)b[2*i+1]);
            if (m < (double)fmin) fmin = (float)m;
            if (m > (double)fmax) fmax = (float)m;
        }
        float range = fmax - fmin;
        for (size_t i = 0; i < n; i++) {
            double m = hypot((double)b[2*i], (double)b[2*i+1]);
            double v = 1.0 * ((m - (double)fmin) / (double)range);   /* image3.cpp:414-416 */
            out[i] = (float)v;
            if (outd) outd[i] = v;
        }
        free(a); free(b);
    } else {
        double* a = (double*)malloc(sizeof(double) * 2 * n);
        double* b = (double*)malloc(sizeof(double) * 2 * n);
        for (size_t i = 0; i < n; i++) { a[2*i] = t1[i]; a[2*i+1] = 0; b[2*i] = t2[i]; b[2*i+1] = 0; }
        orc_fft2d_f64(a, width, height, -1);
        orc_fft2d_f64(b, width, height, -1);
        double c = sqrt((double)n);
        for (size_t i = 0; i < 2 * n; i++) { a[i] /= c; b[i] /= c; }
        for (size_t i = 0; i < n; i++) {
            double re = a[2*i] * b[2*i] + a[2*i+1] * b[2*i+1];
            double im = a[2*i+1] * b[2*i] - a[2*i] * b[2*i+1];
            b[2*i] = re; b[2*i+1] = im;
        }
        orc_fft2d_f64(b, width, height, +1);
        for (size_t i = 0; i < 2 * n; i++) b[i] /= c;
        double dmin, dmax;
        dmin = dmax = hypot(b[0], b[1]);
        for (size_t i = 0; i < n; i++) {
            double m = hypot(b[2*i], b[2*i+1]);
            if (m < dmin) dmin = m;
            if (m > dmax) dmax = m;
        }
        fmin = (float)dmin; fmax = (float)dmax;
        for (size_t i = 0; i < n; i++) {
            double m = hypot(b[2*i], b[2*i+1]);
            double v = (m - dmin) / (dmax - dmin);
            out[i] = (float)v;
            if (outd) outd[i] = v;
        }
        free(a); free(b);
    }
    if (min_out) *min_out = fmin;
    if (max_out) *max_out = fmax;
    return (fmax == fmin) ? 1 : 0;
}

/* image3.cpp:346-517 findOffset with the band loop of image3.cpp:382-418 left open: the reference
 * computes n_bands and then forces it to 1 (image3.cpp:352-353); n_bands == 1 is that behaviour,
 * n_bands > 1 multiplies the normalised surfaces of the bands (out[i] = out[i] * norm, the
 * product evaluated in double and narrowed, image3.cpp:414-416). */
int orc_find_offset_bands(const void* const* imgs1, int w1, int h1, size_t stride1,
                          const void* const* imgs2, int w2, int h2, size_t stride2, int dtype, int n_bands,
                          orc_rect b1, orc_rect b2, int width, int height, int precision,
                          orc_offset* res, float* surface, float* dog_out)
{
    size_t n = (size_t)width * height;
    double* t1 = (double*)malloc(sizeof(double) * n);
    double* t2 = (double*)malloc(sizeof(double) * n);
    float* out = (float*)malloc(sizeof(float) * n);
    float* band = (float*)malloc(sizeof(float) * n);
    double* bandd = (double*)malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; i++) out[i] = 1.0f;                 /* image3.cpp:378-380 */
    for (int b = 0; b < n_bands; b++) {
        load_tile(t1, imgs1[b], w1, h1, stride1, dtype, b1, width, height);
        load_tile(t2, imgs2[b], w2, h2, stride2, dtype, b2, width, height);
        corr_surface(t1, t2, width, height, precision, band, bandd, NULL, NULL);
        for (size_t i = 0; i < n; i++) out[i] = (float)((double)out[i] * bandd[i]);
    }
    free(t1); free(t2); free(band); free(bandd);
    if (surface) memcpy(surface, out, sizeof(float) * n);

    /* image3.cpp:456-483 */
    const int pp_row = height / 2, pp_col = width / 2;
    const int height2 = pp_row * 2, width2 = pp_col * 2;
    float* imout = (float*)calloc(n, sizeof(float));
    float* g1 = (float*)malloc(sizeof(float) * n);
    for (int row = 0; row < height2; row++)
        for (int col = 0; col < width2; col++)
            imout[(size_t)row * width + col] = out[(size_t)row * width + col];
    orc_gaussian_blur(imout, g1, width, height, 151, 99.0);
    for (size_t i = 0; i < n; i++) g1[i] = imout[i] - g1[i];      /* imout_g2 = imout - imout_g1 */
    if (dog_out) memcpy(dog_out, g1, sizeof(float) * n);

    /* image3.cpp:486-512: first occurrence of the strict maximum, then wrap to signed offsets */
    float max = -1, max_out = -1;
    int max_col = 0, max_row = 0;
    for (int row = 0; row < height2; row++)
        for (int col = 0; col < width2; col++) {
            float v = g1[(size_t)row * width + col];
            if (v > max) {
                max = v;
                max_out = out[(size_t)row * width + col];
                max_col = col; max_row = row;
            }
        }
    if (max_col > pp_col) max_col -= width;
    if (max_row > pp_row) max_row -= height;
    res->dx = max_col; res->dy = max_row;
    res->peak = max_out; res->dog_peak = max;
    free(imout); free(g1); free(out);
    return 0;
}

/* image3.cpp:346-517 findOffset as the reference runs it (band 1 only) */
int orc_find_offset(const void* img1, int w1, int h1, size_t stride1,
                    const void* img2, int w2, int h2, size_t stride2, int dtype,
                    orc_rect b1, orc_rect b2, int width, int height, int precision,
                    orc_offset* res, float* surface, float* dog_out)
{
    return orc_find_offset_bands(&img1, w1, h1, stride1, &img2, w2, h2, stride2, dtype, 1,
                                 b1, b2, width, height, precision, res, surface, dog_out);
}

/* findCPPMax (image2.cpp:216-308, image-find-cpp.cpp:68-158): product of the bands of an
 * fft-shifted correlation raster, first strict maximum outside a central mask of mask_pct percent
 * of the half sizes; offsets are relative to the raster centre.  bands: n_bands planes of
 * rows*cols floats. */
float orc_masked_peak(const float* bands, int cols, int rows, int n_bands, float mask_pct,
                      int* offset_x, int* offset_y)
{
    size_t px_count = (size_t)rows * cols;
    float* img = (float*)malloc(sizeof(float) * px_count);
    memcpy(img, bands, sizeof(float) * px_count);
    for (int band = 1; band < n_bands; band++)
        for (size_t i = 0; i < px_count; i++) img[i] = img[i] * bands[(size_t)band * px_count + i];
    const int pp_row = rows / 2, pp_col = cols / 2;
    const int rows2 = pp_row * 2, cols2 = pp_col * 2;
    if (mask_pct > 99) mask_pct = 99;
    if (mask_pct <= 1) mask_pct = 0;
    /* int * float / int: float arithmetic, truncated on assignment (image2.cpp:259-263) */
    const int mask_min_row = pp_row - (pp_row * mask_pct / 100);
    const int mask_max_row = pp_row + (pp_row * mask_pct / 100);
    const int mask_min_col = pp_col - (pp_col * mask_pct / 100);
    const int mask_max_col = pp_col + (pp_col * mask_pct / 100);
    float max = -1;
    int max_col = 0, max_row = 0;
    for (int row = 0; row < rows2; row++) {
        int central = (row >= mask_min_row) && (row < mask_max_row);
        for (int col = 0; col < cols2; col++) {
            if (central && col >= mask_min_col && col < mask_max_col) continue;
            float v = img[(size_t)row * cols + col];
            if (v > max) { max = v; max_col = col; max_row = row; }
        }
    }
    *offset_x = max_col - pp_col;
    *offset_y = max_row - pp_row;
    free(img);
    return max;
}

/* image-cross-cor.cpp:104-176 for one band, both rasters the size of the output
 * (the reference neither zeroes its buffers nor handles unequal sizes correctly:
 * image-cross-cor.cpp:85-100) */
int orc_xcorr_surface_u8(const void* img1, const void* img2, int dtype, int width, int height,
                         size_t stride_bytes, int precision, uint8_t* out8)
{
    size_t n = (size_t)width * height;
    double* t1 = (double*)malloc(sizeof(double) * n);
    double* t2 = (double*)malloc(sizeof(double) * n);
    float* out = (float*)malloc(sizeof(float) * n);
    orc_rect full = { 0, 0, width, height };
    load_tile(t1, img1, width, height, stride_bytes, dtype, full, width, height);
    load_tile(t2, img2, width, height, stride_bytes, dtype, full, width, height);
    float fmin, fmax;
    double* outd = (double*)malloc(sizeof(double) * n);
    int degenerate = corr_surface(t1, t2, width, height, precision, out, outd, &fmin, &fmax);
    /* out[i] = floor(((cabs - min)/(max-min)) * 255.0) (image-cross-cor.cpp:166-168); the C
     * conversion of NaN to unsigned char is undefined, the oracle writes 0 there */
    for (size_t i = 0; i < n; i++) {
        double v = floor(outd[i] * 255.0);
        out8[i] = (degenerate || !(v >= 0)) ? 0 : (uint8_t)v;
    }
    orc_fft2shift_u8(out8, height, width);   /* image-cross-cor.cpp:170 */
    free(t1); free(t2); free(out); free(outd);
    return 0;
}
