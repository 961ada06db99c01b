/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle.h).  DFT primitives. */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define REAL double
#define SUFFIX f64
#include "fft_impl.inc"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX f32
#include "fft_impl.inc"
#undef REAL
#undef SUFFIX

double orc_load_px(const void* img, int dtype, size_t stride_bytes, int x, int y)
{
    const unsigned char* row = (const unsigned char*)img + (size_t)y * stride_bytes;
    switch (dtype) {
    case ORC_U8: return (double)row[x];
    case ORC_U16: return (double)((const uint16_t*)row)[x];
    default: return (double)((const float*)row)[x];
    }
}

/* image-fft.cpp:45-105 (CalcFFT): real raster -> complex float, forward DFT, each value
 * divided by sqrt(px_count) held as a complex double (image-fft.cpp:84-87). */
int orc_fft2d_forward(const void* img, int dtype, int width, int height, size_t stride_bytes,
                      float* out)
{
    size_t n = (size_t)width * height;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            out[2 * ((size_t)y * width + x)] = (float)orc_load_px(img, dtype, stride_bytes, x, y);
            out[2 * ((size_t)y * width + x) + 1] = 0.0f;
        }
    orc_fft2d_f32(out, width, height, -1);
    /* out[i] = out[i] / norm with norm = csqrt(px_count + 0I) as complex double: the float
     * value is promoted, divided in double, and narrowed back to float. */
    double norm = sqrt((double)n);
    for (size_t i = 0; i < 2 * n; i++) out[i] = (float)((double)out[i] / norm);
    return 0;
}

/* image3.cpp:287-315: swap left/right halves, then top/bottom halves */
void orc_fft2shift_f32(float* x, int width, int height)
{
    int m2 = height / 2, n2 = width / 2;
    for (int i = 0; i < height; i++)
        for (int k = 0; k < n2; k++) {
            float t = x[(size_t)i * width + k + n2];
            x[(size_t)i * width + k + n2] = x[(size_t)i * width + k];
            x[(size_t)i * width + k] = t;
        }
    for (int i = 0; i < m2; i++)
        for (int k = 0; k < width; k++) {
            float t = x[(size_t)(i + m2) * width + k];
            x[(size_t)(i + m2) * width + k] = x[(size_t)i * width + k];
            x[(size_t)i * width + k] = t;
        }
}

/* image-cross-cor.cpp:52-80 */
void orc_fft2shift_u8(uint8_t* x, int rows, int cols)
{
    int m2 = rows / 2, n2 = cols / 2;
    for (int i = 0; i < rows; i++)
        for (int k = 0; k < n2; k++) {
            uint8_t t = x[(size_t)i * cols + k + n2];
            x[(size_t)i * cols + k + n2] = x[(size_t)i * cols + k];
            x[(size_t)i * cols + k] = t;
        }
    for (int i = 0; i < m2; i++)
        for (int k = 0; k < cols; k++) {
            uint8_t t = x[(size_t)(i + m2) * cols + k];
            x[(size_t)(i + m2) * cols + k] = x[(size_t)i * cols + k];
            x[(size_t)i * cols + k] = t;
        }
}
