// The one exchange of a pair sharded by row strips over several processes (one per GPU): the gather of the result
// rows to rank 0, over RCCL.  Built as a library of its own, libmistereo_rccl.so (libmistereo.so links no
// communication library: a context is one GPU).  The reference has no counterpart (single process,
// image3.cpp:974-975).  Point-to-point transfers posted in one group: on a fully connected xGMI node every peer
// has its own link into rank 0, so the receives run side by side, straight into the rows of the full rasters.
#include "mi_stereo_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string.h>

static_assert(sizeof(ncclUniqueId) == MIS_RCCL_ID_BYTES, "the id is handed round as plain bytes");

extern "C" {

int mis_rccl_unique_id(char id[MIS_RCCL_ID_BYTES])
{
    if (!id) return MIS_RCCL_ERR_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return MIS_RCCL_ERR_COMM;
    memcpy(id, &u, sizeof u);
    return MIS_RCCL_OK;
}

int mis_rccl_comm_init(const char id[MIS_RCCL_ID_BYTES], int rank, int world, int device_id, void** comm)
{
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return MIS_RCCL_ERR_ARG;
    *comm = nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return MIS_RCCL_ERR_DEVICE;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    if (ncclCommInitRank(&c, world, u, rank) != ncclSuccess) return MIS_RCCL_ERR_COMM;
    *comm = (void*)c;
    return MIS_RCCL_OK;
}

int mis_rccl_comm_destroy(void* comm)
{
    if (!comm) return MIS_RCCL_OK;
    return ncclCommDestroy((ncclComm_t)comm) == ncclSuccess ? MIS_RCCL_OK : MIS_RCCL_ERR_COMM;
}

void mis_rccl_row_strip(int height, int world, int rank, int* row0, int* nrows)
{
    const int base = height / world, rem = height % world;
    if (row0) *row0 = rank * base + (rank < rem ? rank : rem);
    if (nrows) *nrows = base + (rank < rem ? 1 : 0);
}

int mis_rccl_gather_rows(void* comm, int rank, int world, int height, int width, int n_rasters,
                         const float* const* d_strip, int strip_row0, float* const* d_full, void* hip_stream)
{
    if (!comm || world < 1 || rank < 0 || rank >= world || height < 1 || width < 1 || n_rasters < 1 || !d_strip) return MIS_RCCL_ERR_ARG;
    if (rank == 0 && !d_full) return MIS_RCCL_ERR_ARG;
    const ncclComm_t c = (ncclComm_t)comm;
    const hipStream_t s = (hipStream_t)hip_stream;
    int my0 = 0, myn = 0;
    mis_rccl_row_strip(height, world, rank, &my0, &myn);
    if (rank == 0) {
        // rank 0's own rows: already in place when it computed into the full rasters, copied otherwise
        for (int i = 0; i < n_rasters; i++) {
            const float* src = d_strip[i] + (size_t)strip_row0 * width;
            float* dst = d_full[i] + (size_t)my0 * width;
            if (src != dst && myn > 0 &&
                hipMemcpyAsync(dst, src, (size_t)myn * width * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
                return MIS_RCCL_ERR_DEVICE;
        }
    }
    if (world == 1) return MIS_RCCL_OK;
    ncclResult_t r = ncclGroupStart();
    if (rank == 0) {
        for (int k = 1; k < world && r == ncclSuccess; k++) {
            int r0 = 0, n = 0;
            mis_rccl_row_strip(height, world, k, &r0, &n);
            for (int i = 0; i < n_rasters && r == ncclSuccess && n > 0; i++)
                r = ncclRecv(d_full[i] + (size_t)r0 * width, (size_t)n * width, ncclFloat, k, c, s);
        }
    } else {
        for (int i = 0; i < n_rasters && r == ncclSuccess && myn > 0; i++)
            r = ncclSend(d_strip[i] + (size_t)strip_row0 * width, (size_t)myn * width, ncclFloat, 0, c, s);
    }
    const ncclResult_t e = ncclGroupEnd();
    return (r == ncclSuccess && e == ncclSuccess) ? MIS_RCCL_OK : MIS_RCCL_ERR_COMM;
}

int mis_rccl_gather_frames(void* comm, int rank, int world, size_t n_floats, int n_rasters,
                           const float* const* d_src, float* const* d_dst, void* hip_stream)
{
    if (!comm || world < 1 || rank < 0 || rank >= world || n_floats < 1 || n_rasters < 1 || !d_src) return MIS_RCCL_ERR_ARG;
    if (rank == 0 && !d_dst) return MIS_RCCL_ERR_ARG;
    const ncclComm_t c = (ncclComm_t)comm;
    const hipStream_t s = (hipStream_t)hip_stream;
    if (rank == 0) {
        for (int i = 0; i < n_rasters; i++)
            if (d_dst[i] != d_src[i] &&
                hipMemcpyAsync(d_dst[i], d_src[i], n_floats * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
                return MIS_RCCL_ERR_DEVICE;
    }
    if (world == 1) return MIS_RCCL_OK;
    ncclResult_t r = ncclGroupStart();
    if (rank == 0) {
        for (int k = 1; k < world && r == ncclSuccess; k++)
            for (int i = 0; i < n_rasters && r == ncclSuccess; i++)
                r = ncclRecv(d_dst[(size_t)k * n_rasters + i], n_floats, ncclFloat, k, c, s);
    } else {
        for (int i = 0; i < n_rasters && r == ncclSuccess; i++) r = ncclSend(d_src[i], n_floats, ncclFloat, 0, c, s);
    }
    const ncclResult_t e = ncclGroupEnd();
    return (r == ncclSuccess && e == ncclSuccess) ? MIS_RCCL_OK : MIS_RCCL_ERR_COMM;
}

}  // extern "C"
