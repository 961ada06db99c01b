/*
 * CPU oracle for the stereo-correlation hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * This library restates, in plain C, the arithmetic of the reference programs
 * (klassenjs/image-processing: image3.cpp, image-cross-cor.cpp, image-fft.cpp) and
 * the dense windowed-NCC definition built on top of them.  It exists to check
 * the HIP library and to provide a CPU timing baseline.  Nothing in the product
 * path links or loads it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or sample
 * imagery, and cannot be compiled in the build environment (it needs GDAL,
 * FFTW3, OpenCV 2.4 and PROJ).  This restatement is therefore pinned only by
 * the reference's source text (cited per function) and by an independent numpy
 * restatement of the same formulas (tests/golden/make_golden.py).
 */
#ifndef MIS_ORACLE_H
#define MIS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pixel types understood by every entry point */
enum { ORC_U8 = 0, ORC_U16 = 1, ORC_F32 = 2 };

/* sentinel written to disparity outputs where no window fits */
#define ORC_INVALID_DISP (-32768)

typedef struct { int minx, miny, width, height; } orc_rect;      /* image3.cpp:77-82 */
typedef struct { int dx, dy; float peak; float dog_peak; } orc_offset;
typedef struct { float A[9]; float R[9]; float T[3]; } orc_camera; /* image3.cpp:1014-1030 */

/* ---- dense windowed zero-mean NCC (dense limit of image3.cpp:775-811, 866-870) ---- */
int orc_ncc_dense(const void* left, const void* right, int dtype,
                  int width, int height, size_t stride_bytes,
                  int window, int dmin, int dmax,
                  float* disp, float* peak, int16_t* idisp, float* peak2,
                  double* volume /* nullable: [H][W][nd] scores, NaN where invalid */);
/* one score, straight from the definition (double); returns NaN where invalid */
double orc_ncc_score(const void* left, const void* right, int dtype,
                     int width, int height, size_t stride_bytes,
                     int window, int x, int y, int d);

/* ---- FFT primitives (image-fft.cpp:65-87, image3.cpp:365-372) ---- */
/* in-place 2-D complex DFT of h rows x w cols, interleaved re/im; sign -1 forward, +1 backward,
 * unnormalised like FFTW.  Any size (power of two: radix-2; else Bluestein). */
void orc_fft2d_f64(double* data, int w, int h, int sign);
void orc_fft2d_f32(float* data, int w, int h, int sign);
/* image-fft.cpp CalcFFT: forward c2c FFT of a real raster, divided by sqrt(w*h). out: w*h complex64 */
int orc_fft2d_forward(const void* img, int dtype, int width, int height, size_t stride_bytes,
                      float* out_complex);

/* ---- tile cross-correlation with DoG peak pick (image3.cpp:318-335, 346-517) ---- */
/* precision: 0 = float FFT (as the reference), 1 = double FFT */
int orc_find_offset(const void* img1, int w1, int h1, size_t stride1,
                    const void* img2, int w2, int h2, size_t stride2, int dtype,
                    orc_rect b1, orc_rect b2, int width, int height, int precision,
                    orc_offset* out,
                    float* surface /* nullable: width*height normalised |c| (image3.cpp:414-416) */,
                    float* dog /* nullable: width*height DoG image (image3.cpp:483) */);
/* the same with the band loop of image3.cpp:382-418 open (the reference forces n_bands = 1,
 * image3.cpp:353): the normalised surfaces of the bands are multiplied */
int orc_find_offset_bands(const void* const* imgs1, int w1, int h1, size_t stride1,
                          const void* const* imgs2, int w2, int h2, size_t stride2, int dtype, int n_bands,
                          orc_rect b1, orc_rect b2, int width, int height, int precision,
                          orc_offset* out, float* surface, float* dog);
/* findCPPMax (image2.cpp:216-308, image-find-cpp.cpp:68-158): masked first maximum of the band
 * product of an fft-shifted correlation raster; returns the maximum */
float orc_masked_peak(const float* bands, int cols, int rows, int n_bands, float mask_pct,
                      int* offset_x, int* offset_y);
/* the 151-tap sigma-99 blur alone, OpenCV 2.4 GaussianBlur semantics (image3.cpp:481) */
void orc_gaussian_blur(const float* src, float* dst, int width, int height, int ksize, double sigma);
/* image-cross-cor.cpp:104-176: whole-image correlation surface -> floor(norm*255) u8, fft-shifted */
int orc_xcorr_surface_u8(const void* img1, const void* img2, int dtype, int width, int height,
                         size_t stride_bytes, int precision, uint8_t* out);
/* image3.cpp:287-315 / image-cross-cor.cpp:52-80 */
void orc_fft2shift_f32(float* x, int width, int height);
void orc_fft2shift_u8(uint8_t* x, int rows, int cols);

/* ---- triangulation (image3.cpp:102-132, 148-264) ---- */
void orc_triangulate(const orc_camera* c1, const orc_camera* c2,
                     const int32_t* u1v1u2v2, int n, float* xyz);
/* float pixel coordinates (no counterpart in the reference; equals orc_triangulate on integer values) */
void orc_triangulate_f(const orc_camera* c1, const orc_camera* c2,
                       const float* u1v1u2v2, int n, float* xyz);

/* ---- hierarchical driver (image3.cpp:676-865), depth-first like the reference ---- */
typedef struct {
    orc_rect bbox1;          /* leaf tile in image 1 */
    int total_dx, total_dy;  /* image3.cpp:698-699 */
    int u1, v1, u2, v2;      /* image3.cpp:815-818 */
    float peak;
} orc_leaf;
/* returns number of leaves (<= max_leaves written), or -1 on "no overlap" (image3.cpp:710-712) */
int orc_disparity_map(const void* img1, const void* img2, int dtype, int width, int height,
                      size_t stride_bytes, int min_size, int precision,
                      orc_leaf* leaves, int max_leaves, int* n_find_offset_calls);
int orc_disparity_map2(const void* img1, int width1, int height1, size_t stride1_bytes,
                       const void* img2, int width2, int height2, size_t stride2_bytes, int dtype,
                       int min_size, int precision, orc_leaf* leaves, int max_leaves, int* n_find_offset_calls);

int orc_num_threads(void);
void orc_set_num_threads(int n);
void orc_ncc_scores(const void* left, const void* right, int dtype, int width, int height, size_t stride_bytes,
                    int window, int n, const int* xs, const int* ys, const int* ds, double* out);

#ifdef __cplusplus
}
#endif
#endif
