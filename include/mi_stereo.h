/*
 * mi_stereo.h -- C ABI of libmistereo.so, the MI355X (gfx950) stereo-correlation library.
 *
 * Drop-in boundary for the hot path of klassenjs/image-processing.  The reference has no
 * plugin or FFI layer: its programs call static functions that reach the rasters through
 * globals.  Each entry point below names the reference function it replaces (file:line in
 * the reference tree); the host program keeps raster I/O, bookkeeping and output files.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (MIS_OK == 0), never throws;
 *     mis_last_error(ctx) returns the message of the last failure on that context
 *   - the caller owns every buffer it passes in; the context owns device scratch
 *   - calls on one context are serialised by the caller; contexts are independent
 *   - "_device" entry points take device pointers and only enqueue work on the context's
 *     stream (mis_set_stream); the others take host pointers and return when results are in
 *     host memory
 *   - there is no CPU fallback: without a usable HIP device mis_create fails
 *
 * Environment (read once per process; for comparisons, the results stay within the same bars)
 *   MISTEREO_NCC_MFMA=0   dense 8-bit NCC with windows up to 21 x 21 on the vector kernels instead of the matrix cores
 *   MISTEREO_XC_MFMA=0    the Gaussian passes of the tile correlation on the vector kernels (tap order of the
 *                         reference's filter loops) instead of the matrix cores
 */
#ifndef MI_STEREO_H
#define MI_STEREO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIS_OK 0
#define MIS_ERR_ARG 1        /* invalid argument */
#define MIS_ERR_DEVICE 2     /* HIP runtime failure */
#define MIS_ERR_UNSUPPORTED 3
#define MIS_ERR_STATE 4      /* call sequence error (e.g. no image pair bound) */

/* pixel types of rasters handed to the library (GDAL Byte / UInt16 / Float32) */
#define MIS_U8 0
#define MIS_U16 1
#define MIS_F32 2

/* value written to disparity outputs where no correlation window fits */
#define MIS_INVALID_DISP (-32768)

typedef struct mis_ctx mis_ctx;

/* image3.cpp:77-82 (struct Rect) */
typedef struct { int32_t minx, miny, width, height; } mis_rect;

/* result of one tile correlation: what findOffset returns through *offset_x, *offset_y and
 * its return value (image3.cpp:507-516), plus the DoG value at the peak (image3.cpp:495) */
typedef struct { int32_t dx, dy; float peak; float dog_peak; } mis_offset;

/* camera read from "<image>.yaml": A (intrinsics), R, T as CV_32F (image3.cpp:1014-1030) */
typedef struct { float A[9]; float R[9]; float T[3]; } mis_camera;

/* ---- context ------------------------------------------------------------------------- */
/* Replaces the per-call fftwf_malloc / fftwf_plan_dft_2d / free of image3.cpp:357-372, 421-425
 * and the process globals of image3.cpp:340-344: buffers, twiddles and the blur kernel live
 * in the context. */
int mis_create(int device_id, mis_ctx** out);
int mis_destroy(mis_ctx* ctx);
const char* mis_last_error(const mis_ctx* ctx);   /* ctx may be NULL: last mis_create error */
/* hipStream_t as void*; NULL selects the context's own stream (the default) */
int mis_set_stream(mis_ctx* ctx, void* hip_stream);
int mis_synchronize(mis_ctx* ctx);

/* ---- image pair ------------------------------------------------------------------------ */
/* Replaces the GDALOpen'ed srcDS1/srcDS2 globals and the RasterIO reads inside the hot loop
 * (image3.cpp:325-328, 979-980): the pair is copied to HBM once. */
int mis_upload_pair(mis_ctx* ctx, const void* img1, const void* img2, int dtype,
                    int width, int height, size_t row_stride_bytes);
/* When both rasters are page-locked (below) and tightly packed, mis_upload_pair returns with the copy IN FLIGHT
 * (row chunks by DMA on a stream of its own); the entry points that read the pair wait on the device for the rows
 * they need, so a dense correlation starts on the first rows while the last ones still travel.  The host must not
 * change or free such rasters before mis_synchronize (or a later mis_upload_pair followed by it) has returned. */
/* No counterpart in the reference (its rasters live in GDAL's block cache, image3.cpp:325-328).  Page-locked host
 * memory for the rasters a host hands to the entry points that take host pointers (the pair it uploads, the rasters
 * results are copied into): those entry points recognise such memory and move it by DMA straight from / to it
 * instead of through pageable copies -- about twice the transfer rate, and result strips travel while later strips
 * are computed.  Optional: every entry point works with pageable memory.
 *   mis_alloc_host / mis_free_host   memory allocated page-locked (read the raster into it: it is ordinary memory
 *                                    for the host); memory the caller allocated with hipHostMalloc works as well.
 *   mis_pin_host / mis_unpin_host    page-lock a range the caller already owns (hipHostRegister).  Not every
 *                                    system allows it (MIS_ERR_DEVICE then; carry on with pageable memory).  Unpin
 *                                    before freeing the memory. */
int mis_alloc_host(mis_ctx* ctx, size_t bytes, void** out);
int mis_free_host(mis_ctx* ctx, void* ptr);
int mis_pin_host(mis_ctx* ctx, void* ptr, size_t bytes);
int mis_unpin_host(mis_ctx* ctx, void* ptr);
/* Same, for rasters already in device memory (not copied; must stay valid while bound).
 * The dense-NCC entry points keep tables derived from the pixels (window statistics, packed copies)
 * from one call to the next and rebuild them only when a pair is bound: a caller that writes new
 * pixels into buffers that stay bound must rebind them or call mis_invalidate_pair before the next
 * call, or that call answers for the old pixels.  Writes must be ordered before the next call on
 * the context's stream by the caller.
 * Both rasters have the same size and type.  (The reference only warns when the two datasets
 * differ in size, image3.cpp:995-998, and then clips against the second one's extent,
 * image3.cpp:684-685, 705-706: this library requires equal sizes; pad the smaller raster.) */
int mis_bind_pair_device(mis_ctx* ctx, const void* d_img1, const void* d_img2, int dtype,
                         int width, int height, size_t row_stride_bytes);
/* Declares that the pixels of the bound rasters have changed (same buffers, same geometry): every
 * cached table is rebuilt by the next call that needs it.  No counterpart in the reference, which
 * re-reads its rasters from the datasets inside every findOffset call (image3.cpp:325-328). */
int mis_invalidate_pair(mis_ctx* ctx);

/* ---- dense windowed NCC ------------------------------------------------------------------
 * The dense per-pixel mode the reference leaves disabled (image3.cpp:775-811) and lists as a
 * TODO (image3.cpp:866-870): for every pixel of image 1, zero-mean normalised correlation of
 * the window x window patch against image 2 at horizontal offsets d in [dmin,dmax] (image-1
 * position = image-2 position + d, the sign of image3.cpp:696-699), first maximum wins
 * (image3.cpp:486-501), parabolic subpixel refinement.
 *   disp   float  disparity with subpixel part, MIS_INVALID_DISP where no window fits
 *   peak   float  NCC at the integer maximum (0 where invalid)
 *   idisp  int16  integer maximum (nullable)
 *   peak2  float  best NCC further than 1 px from the maximum, -1 if none (nullable; a call that asks
 *                 for it takes 13 ms instead of 10 at 8192 x 8192, 21 x 21, +-256)
 * u8 and u16 rasters; window odd, 3..31; 1 <= dmax-dmin+1 <= 2048. */
int mis_ncc_dense(mis_ctx* ctx, int window, int dmin, int dmax,
                  float* disp, float* peak, int16_t* idisp, float* peak2);
int mis_ncc_dense_device(mis_ctx* ctx, int window, int dmin, int dmax,
                         float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2);
/* Row-strip form used when a frame is split over GPUs: only output rows [row0, row0+nrows)
 * are computed and written (outputs are still addressed as full frames). */
int mis_ncc_dense_rows(mis_ctx* ctx, int window, int dmin, int dmax, int row0, int nrows,
                       float* disp, float* peak, int16_t* idisp, float* peak2);
int mis_ncc_dense_rows_device(mis_ctx* ctx, int window, int dmin, int dmax, int row0, int nrows,
                              float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2);

/* ---- several GPUs from one host process ---------------------------------------------------------
 * The reference is one process on one machine (its only concurrency is FFTW's thread pool,
 * image3.cpp:974-975); rows of the dense correlation are independent, so a frame splits into row strips.
 * A group holds one context per listed device (a device may be listed more than once).
 * mis_group_ncc_dense: dense NCC of one host-resident pair; the output rows are split into n_dev contiguous
 * strips (the remainder rows go to the first devices), device i is sent only its strip plus window/2 halo rows
 * above and below, computes its rows and copies them into the caller's host rasters (one host thread per
 * device).  Correlation windows never leave a device's rows, so the result equals mis_ncc_dense on one device
 * bit for bit and no data moves between devices.  (Hosts that run one process per GPU use
 * mis_ncc_dense_rows_device and gather with RCCL instead.) */
typedef struct mis_group mis_group;
int mis_create_group(const int* device_ids, int n_dev, mis_group** out);
int mis_destroy_group(mis_group* grp);
int mis_group_size(const mis_group* grp);
const char* mis_group_last_error(const mis_group* grp);     /* grp may be NULL: last mis_create_group error */
int mis_group_ncc_dense(mis_group* grp, const void* img1, const void* img2, int dtype,
                        int width, int height, size_t row_stride_bytes, int window, int dmin, int dmax,
                        float* disp, float* peak, int16_t* idisp, float* peak2);

/* ---- FFT tile correlation ------------------------------------------------------------------
 * Batched findOffset (image3.cpp:346-517 with runFFT image3.cpp:318-335): for each i, the
 * tile r1[i] of image 1 and r2[i] of image 2, zero-padded to tile_w x tile_h, are correlated
 * over all circular shifts; the |c| surface is min-max normalised, the 151-tap sigma-99
 * Gaussian is subtracted and the first maximum of the difference gives (dx,dy).
 * tile_w, tile_h: any size up to 8192 (powers of two up to 16384).  Powers of two in 16..8192
 * run entirely in LDS-resident radix-2 transforms; other sizes (the reference's first level is
 * the whole raster) go through chirp-z row transforms and are slower per pixel.
 * A box that reaches past the raster is zero-filled pixel by pixel where it leaves it.  This is a
 * deliberate choice, not the reference's behaviour (GDAL's RasterIO refuses such a window as a
 * whole and image3.cpp:325 ignores the error: the buffer stays zero and findOffset returns -1);
 * calcDisparityMap never produces such boxes (image3.cpp:702-740), only direct callers can. */
int mis_xcorr_tiles(mis_ctx* ctx, const mis_rect* r1, const mis_rect* r2, int n,
                    int tile_w, int tile_h, mis_offset* out);
int mis_xcorr_tiles_device(mis_ctx* ctx, const mis_rect* d_r1, const mis_rect* d_r2, int n,
                           int tile_w, int tile_h, mis_offset* d_out);
/* Debug/inspection: normalised surface (image3.cpp:414-416) and DoG image (image3.cpp:483) of
 * one tile, tile_w*tile_h floats each, host memory, either may be NULL. */
int mis_xcorr_tile_surfaces(mis_ctx* ctx, mis_rect r1, mis_rect r2, int tile_w, int tile_h,
                            float* surface, float* dog);

/* Further bands of the two rasters for mis_xcorr_tiles: the band loop of findOffset
 * (image3.cpp:382-418) multiplies the normalised correlation surfaces of all bands, but the
 * reference forces n_bands to 1 (image3.cpp:352-353), which is what a context without extra bands
 * does.  n_extra (0..15) bands 2.. of image 1 and image 2, same pixel type and size as the bound
 * pair; dropped when a new pair is bound.  The _device form keeps the caller's device pointers
 * (same row stride as the bound pair), the other copies host rasters. */
int mis_set_extra_bands_device(mis_ctx* ctx, int n_extra, const void* const* d_img1, const void* const* d_img2);
int mis_upload_extra_bands(mis_ctx* ctx, int n_extra, const void* const* img1, const void* const* img2,
                           size_t row_stride_bytes);

/* findCPPMax (image2.cpp:216-308, image-find-cpp.cpp:68-158): product of the n_bands planes of an
 * fft-shifted correlation raster (host, width*height floats each), then the first maximum above -1
 * outside a central mask of mask_pct percent of the half sizes (clamped as image2.cpp:256-257);
 * offsets are relative to the raster centre, *peak is the maximum (-1 when nothing exceeded it). */
int mis_masked_peak(mis_ctx* ctx, const float* bands, int width, int height, int n_bands, float mask_pct,
                    int* offset_x, int* offset_y, float* peak);

/* image-fft.cpp:45-105 (CalcFFT): forward 2-D DFT of `batch` real rasters of w x h pixels
 * (contiguous, row stride w), each value divided by sqrt(w*h); out: batch*w*h complex64
 * (interleaved re,im).  Host pointers.  w, h: as for mis_xcorr_tiles. */
int mis_fft2d_forward(mis_ctx* ctx, const void* in, int dtype, int w, int h, int batch,
                      float* out_complex);

/* image-cross-cor.cpp:104-186 (CalcFFT) for the bound pair: whole-image correlation surface
 * quantised floor(v*255) and fft-shifted (image-cross-cor.cpp:166-170).  out: w*h bytes, host.
 * Image sides: as for mis_xcorr_tiles; odd sizes keep the reference's fft2shift behaviour (the
 * last row / column is not moved, image-cross-cor.cpp:52-80). */
int mis_xcorr_surface_u8(mis_ctx* ctx, uint8_t* out);

/* ---- triangulation -------------------------------------------------------------------------
 * calculateXYZ (image3.cpp:148-264, wrapper 272-280) for n points; pts = n x (u1,v1,u2,v2),
 * xyz = n x (X,Y,Z), float32 throughout. */
int mis_triangulate(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                    const int32_t* pts, int n, float* xyz);
int mis_triangulate_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                           const int32_t* d_pts, int n, float* d_xyz);
/* Dense form: one point per pixel of a disparity raster produced by mis_ncc_dense:
 * (u1,v1) = (x,y), (u2,v2) = (x - round(disp), y); pixels with peak < min_peak or invalid
 * disparity get NaN.  d_xyz: height*width*3 floats.  Returns the matched count in *n_matched
 * (nullable; reading it synchronises the stream). */
int mis_triangulate_disparity_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                     const float* d_disp, const float* d_peak, float min_peak,
                                     float* d_xyz, long long* n_matched);

/* Same with flags.  MIS_TRI_SUBPIXEL: the second image's column is the fractional x - disp instead of
 * x - round(disp); everything else is calculateXYZ's arithmetic unchanged (the pixel coordinates enter
 * it only through `(u - cx) / fx`, image3.cpp:185-194).  The reference itself only ever triangulates
 * integer pixels (calculateXYZ takes ints, image3.cpp:148, 812-824), so this form has no counterpart
 * there: it equals the flag-less form bit for bit wherever disp is an integer, and is checked against
 * the same arithmetic on float coordinates elsewhere. */
#define MIS_TRI_SUBPIXEL 1
int mis_triangulate_disparity_ex_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                        const float* d_disp, const float* d_peak, float min_peak, int flags,
                                        float* d_xyz, long long* n_matched);

/* The same for rows of a frame that are held as a raster of their own (a strip of a frame split over GPUs):
 * triangulates rows [row0, row0+nrows) of d_disp / d_peak (rasters `width` wide, no pair needs to be bound), whose
 * row k is row k + frame_row0 of the frame the cameras describe, into the same rows of d_xyz (3 floats per pixel).
 * A strip triangulated this way equals the same rows of the whole frame's result bit for bit.  frame_row0 >= 0. */
int mis_triangulate_disparity_rows_device(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                          const float* d_disp, const float* d_peak, float min_peak, int flags,
                                          int width, int row0, int nrows, int frame_row0,
                                          float* d_xyz, long long* n_matched);

/* No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same computation as
 * mis_triangulate_disparity_device, but the raster size is passed in (no pair needs to be bound), the
 * matched count goes to DEVICE memory (d_n_matched, nullable) without synchronising, and the per-call
 * counter slots live in an 8 KB statically allocated device array (one copy per GPU, held until the
 * process ends) instead of context memory: the call makes no hipMalloc / hipFree and only enqueues
 * on the context's stream.  Limits: width <= MIS_NOALLOC_MAX_WIDTH, height <= MIS_NOALLOC_MAX_HEIGHT
 * (MIS_ERR_ARG beyond them).  All calls on one GPU share that array (it is cleared at the start of
 * every call): calls that overlap in time on one GPU, from any context or stream, race on the count;
 * the XYZ output does not depend on it. */
#define MIS_NOALLOC_MAX_WIDTH 65536
#define MIS_NOALLOC_MAX_HEIGHT 65536
int mis_triangulate_disparity_noalloc(mis_ctx* ctx, const mis_camera* c1, const mis_camera* c2,
                                      const float* d_disp, const float* d_peak, float min_peak,
                                      int width, int height, float* d_xyz, long long* d_n_matched);

/* ---- coarse-to-fine driver and output records (host code above the kernels) -------------------
 * One leaf of the tile recursion: what calcDisparityMap hands to its outputs (image3.cpp:812-861). */
typedef struct {
    mis_rect bbox1;              /* leaf tile in image 1 (the area written to the disparity raster) */
    int32_t total_dx, total_dy;  /* image3.cpp:698-699 */
    int32_t u1, v1, u2, v2;      /* matched pixel pair handed to calculateXYZ (image3.cpp:815-818) */
    float peak;                  /* correlation peak of the leaf's tile */
} mis_leaf;

/* calcDisparityMap (image3.cpp:676-865) for the bound pair, level-synchronous instead of
 * depth-first: every recursion level is one batched mis_xcorr_tiles call per tile size.  Leaves
 * come back in the reference's depth-first visiting order.  *n_leaves is the total even when it
 * exceeds max_leaves; *n_tiles (nullable) counts findOffset evaluations.  The first level's tile
 * is the whole raster (any size mis_xcorr_tiles takes); "no
 * overlap" (image3.cpp:710-712) returns MIS_ERR_STATE. */
int mis_disparity_map(mis_ctx* ctx, int image_width, int image_height, int min_size,
                      mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles);
/* Two rasters of DIFFERENT size.  The reference only warns about them (image3.cpp:995-998): the first level
 * correlates the whole of each raster, zero-padded to the larger width and height (image3.cpp:687-688, 322-328),
 * and overlaps are clipped against the SECOND raster's extent (image3.cpp:684-685, 705-706).
 * mis_upload_pair2 copies each raster into the top-left corner of a zeroed device raster of the common size
 * max(w1,w2) x max(h1,h2) and binds that pair (every entry point then sees rasters of the common size whose
 * pixels beyond a raster's own extent are zero, which is what the reference's zero-padded read delivers);
 * mis_disparity_map2 is mis_disparity_map with the two extents kept apart. */
int mis_upload_pair2(mis_ctx* ctx, const void* img1, int width1, int height1, size_t row_stride1_bytes,
                     const void* img2, int width2, int height2, size_t row_stride2_bytes, int dtype);
int mis_disparity_map2(mis_ctx* ctx, int width1, int height1, int width2, int height2, int min_size,
                       mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles);

/* Tile correlation over the group.  The tiles of one recursion level are independent (a child needs only its
 * parent's offset, image3.cpp:768-771) and reach anywhere in both rasters, so every device holds the whole pair
 * (mis_group_upload_pair) and a batch is dealt out in contiguous shares (small batches to fewer devices).
 * mis_group_xcorr_tiles is mis_xcorr_tiles, mis_group_disparity_map is mis_disparity_map (same leaves, same
 * order), with every level's batches shared out this way. */
int mis_group_upload_pair(mis_group* grp, const void* img1, const void* img2, int dtype,
                          int width, int height, size_t row_stride_bytes);
int mis_group_xcorr_tiles(mis_group* grp, const mis_rect* r1, const mis_rect* r2, int n,
                          int tile_w, int tile_h, mis_offset* out);
int mis_group_disparity_map(mis_group* grp, int image_width, int image_height, int min_size,
                            mis_leaf* leaves, int max_leaves, int* n_leaves, int* n_tiles);

/* XYZ text file of image3.cpp:842-849: "%f %f %f " then one "%f " per band, newline; the format
 * xyz2pcd/xyzrgb2pcd.cpp:49 and xyz2geojson (its .rb scripts) read.  bands: n x n_bands samples (nullable). */
int mis_write_xyz(const char* path, const float* xyz, int n, const float* bands, int n_bands, int append);
/* Same with flags.  MIS_XYZ_REF_TRUNCATE reproduces the reference byte for byte: it formats the band
 * values into a buffer of 10*n_bands bytes (image3.cpp:830-846), so the band text of a line is cut
 * after 10*n_bands-1 characters: "%f " of any value >= 100 needs 11, digits of the last band(s)
 * are lost and a fourth 16-bit band disappears.  Without the flag every band is written in full
 * (what xyz2pcd/xyzrgb2pcd.cpp:49 can always parse); mis_write_xyz is that form. */
#define MIS_XYZ_APPEND 1
#define MIS_XYZ_REF_TRUNCATE 2
int mis_write_xyz_ex(const char* path, const float* xyz, int n, const float* bands, int n_bands, int flags);

/* The three Float32 bands of the reference's disparity GeoTIFF (image3.cpp:853-860, 872-893):
 * sqrtf(tx*tx+ty*ty), tx, ty, each leaf stretched over its bbox1; out = 3 planes of
 * width*height floats (INTERLEAVE=BAND). */
int mis_rasterize_leaves(const mis_leaf* leaves, int n, int width, int height, float* out);

/* The TIFF image3.cpp creates through GDAL (CreateOutputDataset, image3.cpp:872-893: GTiff,
 * Float32, INTERLEAVE=BAND, no georeferencing): `bands` planes of width*height floats, one strip
 * per band, PlanarConfiguration=2, SampleFormat=IEEE float; BigTIFF when the pixels exceed 4 GiB.
 * Pixels no leaf covered are whatever the caller put in `planes` (GDAL leaves them 0). */
int mis_write_tiff_f32(const char* path, const float* planes, int width, int height, int bands);

/* Camera file of image3.cpp:1014-1030: OpenCV FileStorage YAML "<image>.yaml" with the
 * !!opencv-matrix nodes A (3x3), R (3x3) and T (3x1), any element type, converted to float as
 * Mat::assignTo(CV_32F) does.  MIS_ERR_STATE: file cannot be opened; MIS_ERR_ARG: a node is
 * missing or has the wrong size. */
int mis_load_camera_yaml(const char* path, mis_camera* cam);

/* ---- device timing (no counterpart in the reference, which has no timing code) ----------------
 * While enabled, every dense-NCC call records HIP events on the context's stream around its
 * table-building kernels, its main correlation (argmax) kernel and its refinement kernel.
 * mis_profile_read waits for the stream, returns the summed durations (milliseconds) and the
 * number of calls since the last read, and resets the pool. */
int mis_profile_enable(mis_ctx* ctx, int on);
int mis_profile_read(mis_ctx* ctx, double* tables_ms, double* main_ms, double* refine_ms, int* calls);

/* library build information: "gfx950 <git describe or date>" */
const char* mis_version(void);

#ifdef __cplusplus
}
#endif
#endif
