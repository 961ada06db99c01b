/* libmistereo_rccl.so: the gather of result rows to rank 0 when ONE pair is sharded by row strips over several
 * processes, one per GPU (RCCL over xGMI).  A library of its own: libmistereo.so links no communication library.
 * The reference has no counterpart: it is one process (image3.cpp:974-975, FFTW's thread pool is its only
 * concurrency); these entry points are what a host that shards the dense correlation of mi_stereo.h
 * (mis_ncc_dense_rows_device) over the GPUs of a node calls once per pair.
 * Plain C types; ncclComm_t and hipStream_t travel as void*; every function returns 0 on success. */
#ifndef MI_STEREO_RCCL_H
#define MI_STEREO_RCCL_H

#ifdef __cplusplus
extern "C" {
#endif

#define MIS_RCCL_OK 0
#define MIS_RCCL_ERR_ARG 1
#define MIS_RCCL_ERR_DEVICE 2
#define MIS_RCCL_ERR_COMM 3
#define MIS_RCCL_ID_BYTES 128

/* Rank 0 makes an id and hands its bytes to the other ranks by whatever channel the host has (a file, MPI, a
 * socket, torch.distributed); every rank then joins with mis_rccl_comm_init (collective: all ranks must call it). */
int mis_rccl_unique_id(char id[MIS_RCCL_ID_BYTES]);
int mis_rccl_comm_init(const char id[MIS_RCCL_ID_BYTES], int rank, int world, int device_id, void** comm);
int mis_rccl_comm_destroy(void* comm);

/* The strip of output rows rank `rank` of `world` owns in a frame `height` rows high: contiguous strips, the
 * remainder rows to the first ranks (the rule of the library's own groups, mis_group_ncc_dense). */
void mis_rccl_row_strip(int height, int world, int rank, int* row0, int* nrows);

/* Gathers n_rasters float rasters (disparity, peak, ...) of a frame height x width to rank 0.  d_strip[i]: this
 * rank's raster i in DEVICE memory, whose row strip_row0 is the first row of the rank's strip (a rank that holds
 * only its strip plus halo rows passes the halo's height; a rank that holds whole-frame rasters passes its
 * strip's row0).  d_full[i] (rank 0 only): the frame's rasters in device memory; rank 0's own rows are copied
 * unless d_strip[i] + strip_row0 * width already is their place.  Transfers are posted in one group on
 * hip_stream (the stream the context works on, so the gather is ordered behind the correlation without a host
 * wait) and complete in stream order.  Collective over all ranks. */
int mis_rccl_gather_rows(void* comm, int rank, int world, int height, int width, int n_rasters,
                         const float* const* d_strip, int strip_row0, float* const* d_full, void* hip_stream);

/* Whole pairs per rank (a batch of frames): gathers every rank's n_rasters float rasters of n_floats values each to
 * rank 0.  d_src[i]: this rank's raster i; d_dst (rank 0 only): world * n_rasters device pointers, d_dst[k * n_rasters + i]
 * receives raster i of rank k (rank 0's own are copied unless d_dst[i] == d_src[i]).  One group on hip_stream;
 * collective over all ranks. */
int mis_rccl_gather_frames(void* comm, int rank, int world, size_t n_floats, int n_rasters,
                           const float* const* d_src, float* const* d_dst, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
