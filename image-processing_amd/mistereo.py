"""ctypes binding of libmistereo.so (include/mi_stereo.h) for tests, bench.py and Python hosts.

The library is the product; this module is plumbing.  It mirrors the reference's call sites:
``find_offset`` <-> findOffset (image3.cpp:346), ``calculate_xyz`` <-> calculateXYZ
(image3.cpp:272), ``fft2d_forward`` <-> image-fft.cpp CalcFFT, ``xcorr_surface_u8`` <->
image-cross-cor.cpp CalcFFT, ``ncc_dense`` <-> the dense mode of image3.cpp:775-811, 866-870.

There is no CPU fallback: importing works anywhere, but ``Context()`` raises unless the HIP
library is present and a gfx950 device is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MISTEREO_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("MISTEREO_LIB") or os.path.join(_HERE, "libmistereo.so")

U8, U16, F32 = 0, 1, 2
INVALID_DISP = -32768


class Rect(C.Structure):
    """struct Rect of image3.cpp:77-82."""
    _fields_ = [("minx", C.c_int32), ("miny", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class Offset(C.Structure):
    _fields_ = [("dx", C.c_int32), ("dy", C.c_int32), ("peak", C.c_float), ("dog_peak", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("A", C.c_float * 9), ("R", C.c_float * 9), ("T", C.c_float * 3)]


class Leaf(C.Structure):
    _fields_ = [("bbox1", Rect), ("total_dx", C.c_int32), ("total_dy", C.c_int32),
                ("u1", C.c_int32), ("v1", C.c_int32), ("u2", C.c_int32), ("v2", C.c_int32),
                ("peak", C.c_float)]


def _leaf_records(leaves, n):
    """The first n records of a Leaf array as tuples (minx, miny, w, h, total_dx, total_dy, u1, v1, u2, v2, peak); read
    through one numpy view of the array (a record is ten int32 and one float32) instead of one ctypes access per field."""
    if n <= 0:
        return []
    a = np.frombuffer(leaves, dtype=np.int32, count=n * 11).reshape(n, 11)
    peaks = a[:, 10].copy().view(np.float32).tolist()
    return [tuple(i) + (p,) for i, p in zip(a[:, :10].tolist(), peaks)]


class MisError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libmistereo error %d: %s" % (code, msg))
        self.code = code


EXPORTS = [
    "mis_create", "mis_destroy", "mis_last_error", "mis_set_stream", "mis_synchronize",
    "mis_upload_pair", "mis_bind_pair_device", "mis_invalidate_pair",
    "mis_ncc_dense", "mis_ncc_dense_device", "mis_ncc_dense_rows_device",
    "mis_xcorr_tiles", "mis_xcorr_tiles_device", "mis_xcorr_tile_surfaces",
    "mis_fft2d_forward", "mis_xcorr_surface_u8",
    "mis_triangulate", "mis_triangulate_device", "mis_triangulate_disparity_device",
    "mis_triangulate_disparity_noalloc", "mis_triangulate_disparity_ex_device", "mis_ncc_dense_rows",
    "mis_create_group", "mis_destroy_group", "mis_group_size", "mis_group_last_error", "mis_group_ncc_dense",
    "mis_group_upload_pair", "mis_group_xcorr_tiles", "mis_group_disparity_map", "mis_triangulate_disparity_rows_device",
    "mis_disparity_map", "mis_write_xyz", "mis_write_xyz_ex", "mis_rasterize_leaves", "mis_write_tiff_f32", "mis_load_camera_yaml",
    "mis_set_extra_bands_device", "mis_upload_extra_bands", "mis_masked_peak",
    "mis_profile_enable", "mis_profile_read", "mis_version", "mis_pin_host", "mis_unpin_host", "mis_alloc_host", "mis_free_host", "mis_upload_pair2", "mis_disparity_map2",
]

_lib = None
_rccl_lib = None
RCCL_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmistereo_rccl.so")
RCCL_EXPORTS = ["mis_rccl_unique_id", "mis_rccl_comm_init", "mis_rccl_comm_destroy", "mis_rccl_row_strip", "mis_rccl_gather_rows",
                "mis_rccl_gather_frames"]


def load_rccl_library():
    """libmistereo_rccl.so (include/mi_stereo_rccl.h): the row gather over RCCL for one process per GPU."""
    global _rccl_lib
    if _rccl_lib is None:
        if not os.path.exists(RCCL_LIB_PATH):
            raise ImportError("libmistereo_rccl.so is missing (%s): build it with `make -C image-processing_amd`" % RCCL_LIB_PATH)
        try:
            import torch  # noqa: F401  (its ROCm libraries first, see load_library)
        except ImportError:
            pass
        lib = C.CDLL(RCCL_LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        lib.mis_rccl_unique_id.argtypes = [C.c_char_p]
        lib.mis_rccl_comm_init.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
        lib.mis_rccl_comm_destroy.argtypes = [vp]
        lib.mis_rccl_row_strip.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        lib.mis_rccl_row_strip.restype = None
        lib.mis_rccl_gather_rows.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(vp), i32, C.POINTER(vp), vp]
        lib.mis_rccl_gather_frames.argtypes = [vp, i32, i32, C.c_size_t, i32, C.POINTER(vp), C.POINTER(vp), vp]
        _rccl_lib = lib
    return _rccl_lib


class RcclGather:
    """One RCCL communicator over the ranks of a job and the gather of result rows to rank 0 through it.
    id_bytes: rank 0 calls RcclGather.unique_id() and hands the 128 bytes to the other ranks."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = load_rccl_library().mis_rccl_unique_id(buf)
        if rc:
            raise MisError(rc, "mis_rccl_unique_id failed")
        return buf.raw

    def __init__(self, id_bytes, rank, world, device):
        self._lib = load_rccl_library()
        self.rank, self.world = rank, world
        self._comm = C.c_void_p()
        rc = self._lib.mis_rccl_comm_init(id_bytes, rank, world, device, C.byref(self._comm))
        if rc:
            raise MisError(rc, "mis_rccl_comm_init failed (rank %d of %d)" % (rank, world))

    def row_strip(self, height, rank=None):
        r0, n = C.c_int(0), C.c_int(0)
        self._lib.mis_rccl_row_strip(height, self.world, self.rank if rank is None else rank, C.byref(r0), C.byref(n))
        return r0.value, n.value

    def gather_rows(self, height, width, strips, strip_row0, full, stream_ptr):
        """strips: this rank's CUDA float32 rasters; full: rank 0's frame rasters (None elsewhere)."""
        n = len(strips)
        src = (C.c_void_p * n)(*[t.data_ptr() for t in strips])
        dst = (C.c_void_p * n)(*[t.data_ptr() for t in full]) if full is not None else None
        rc = self._lib.mis_rccl_gather_rows(self._comm, self.rank, self.world, height, width, n, src, strip_row0, dst,
                                            C.c_void_p(stream_ptr))
        if rc:
            raise MisError(rc, "mis_rccl_gather_rows failed (rank %d of %d)" % (self.rank, self.world))

    def gather_frames(self, srcs, dsts, stream_ptr):
        """srcs: this rank's CUDA float32 rasters; dsts (rank 0): [rank][raster] tensors of the same size."""
        n = len(srcs)
        src = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
        dst = None
        if dsts is not None:
            flat = [t.data_ptr() for per_rank in dsts for t in per_rank]
            dst = (C.c_void_p * len(flat))(*flat)
        rc = self._lib.mis_rccl_gather_frames(self._comm, self.rank, self.world, srcs[0].numel(), n, src, dst, C.c_void_p(stream_ptr))
        if rc:
            raise MisError(rc, "mis_rccl_gather_frames failed (rank %d of %d)" % (self.rank, self.world))

    def close(self):
        if self._comm:
            self._lib.mis_rccl_comm_destroy(self._comm)
            self._comm = C.c_void_p()


def load_library():
    """Load libmistereo.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmistereo.so is missing (%s): build it with `make -C image-processing_amd`; "
                              "there is no CPU fallback" % LIB_PATH)
        # PyTorch ships its own libamdhip64 with the same soname as /opt/rocm's.  Whichever is loaded
        # first serves the whole process, and torch cannot find the GPU when it is not its own: load
        # torch's first when torch is installed (tests and bench.py share device memory with it).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.mis_last_error.restype = C.c_char_p
        lib.mis_last_error.argtypes = [C.c_void_p]
        lib.mis_version.restype = C.c_char_p
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        lib.mis_create.argtypes = [i32, C.POINTER(vp)]
        lib.mis_destroy.argtypes = [vp]
        lib.mis_set_stream.argtypes = [vp, vp]
        lib.mis_synchronize.argtypes = [vp]
        lib.mis_pin_host.argtypes = [vp, vp, sz]
        lib.mis_alloc_host.argtypes = [vp, sz, C.POINTER(vp)]
        lib.mis_free_host.argtypes = [vp, vp]
        lib.mis_unpin_host.argtypes = [vp, vp]
        lib.mis_upload_pair.argtypes = [vp, vp, vp, i32, i32, i32, sz]
        lib.mis_bind_pair_device.argtypes = [vp, vp, vp, i32, i32, i32, sz]
        lib.mis_upload_pair2.argtypes = [vp, vp, i32, i32, sz, vp, i32, i32, sz, i32]
        lib.mis_disparity_map2.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32, vp, vp]
        lib.mis_ncc_dense.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
        lib.mis_ncc_dense_device.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
        lib.mis_ncc_dense_rows.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        lib.mis_create_group.argtypes = [vp, i32, C.POINTER(vp)]
        lib.mis_destroy_group.argtypes = [vp]
        lib.mis_group_size.argtypes = [vp]
        lib.mis_group_last_error.argtypes = [vp]
        lib.mis_group_last_error.restype = C.c_char_p
        lib.mis_group_ncc_dense.argtypes = [vp, vp, vp, i32, i32, i32, sz, i32, i32, i32, vp, vp, vp, vp]
        lib.mis_group_upload_pair.argtypes = [vp, vp, vp, i32, i32, i32, sz]
        lib.mis_group_xcorr_tiles.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        lib.mis_group_disparity_map.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp]
        lib.mis_ncc_dense_rows_device.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        lib.mis_xcorr_tiles.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        lib.mis_xcorr_tiles_device.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        lib.mis_xcorr_tile_surfaces.argtypes = [vp, Rect, Rect, i32, i32, vp, vp]
        lib.mis_fft2d_forward.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        lib.mis_xcorr_surface_u8.argtypes = [vp, vp]
        lib.mis_triangulate.argtypes = [vp, vp, vp, vp, i32, vp]
        lib.mis_triangulate_device.argtypes = [vp, vp, vp, vp, i32, vp]
        lib.mis_disparity_map.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp]
        lib.mis_write_xyz.argtypes = [C.c_char_p, vp, i32, vp, i32, i32]
        lib.mis_write_xyz_ex.argtypes = [C.c_char_p, vp, i32, vp, i32, i32]
        lib.mis_invalidate_pair.argtypes = [vp]
        lib.mis_rasterize_leaves.argtypes = [vp, i32, i32, i32, vp]
        lib.mis_profile_enable.argtypes = [vp, i32]
        lib.mis_profile_read.argtypes = [vp, vp, vp, vp, vp]
        lib.mis_triangulate_disparity_device.argtypes = [vp, vp, vp, vp, vp, f32, vp, vp]
        lib.mis_triangulate_disparity_ex_device.argtypes = [vp, vp, vp, vp, vp, f32, i32, vp, vp]
        lib.mis_triangulate_disparity_rows_device.argtypes = [vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, vp, vp]
        lib.mis_triangulate_disparity_noalloc.argtypes = [vp, vp, vp, vp, vp, f32, i32, i32, vp, vp]
        _lib = lib
    return _lib


def _dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.uint8:
        return U8
    if dt == np.uint16:
        return U16
    if dt == np.float32:
        return F32
    raise TypeError("unsupported pixel type %s" % dt)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dev_ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return C.c_void_p(t.data_ptr()) if t is not None else None


def make_camera(A, R, T):
    cam = Camera()
    cam.A[:] = [float(v) for v in np.asarray(A, np.float32).reshape(9)]
    cam.R[:] = [float(v) for v in np.asarray(R, np.float32).reshape(9)]
    cam.T[:] = [float(v) for v in np.asarray(T, np.float32).reshape(3)]
    return cam


XYZ_APPEND, XYZ_REF_TRUNCATE = 1, 2
TRI_SUBPIXEL = 1


def write_xyz(path, xyz, bands=None, append=False, ref_truncate=False):
    """XYZ text file in the reference's format (image3.cpp:842-849).  ref_truncate: cut the band text
    of a line after 10*n_bands-1 characters as the reference's fixed buffer does (image3.cpp:830-846)."""
    lib = load_library()
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    nb = 0
    if bands is not None:
        bands = np.ascontiguousarray(bands, np.float32).reshape(xyz.shape[0], -1)
        nb = bands.shape[1]
    rc = lib.mis_write_xyz_ex(os.fsencode(path), _np_ptr(xyz), xyz.shape[0], _np_ptr(bands) if nb else None, nb,
                              (XYZ_APPEND if append else 0) | (XYZ_REF_TRUNCATE if ref_truncate else 0))
    if rc != 0:
        raise MisError(rc, "mis_write_xyz failed for %s" % path)


def write_tiff_f32(path, planes):
    """Band-interleaved Float32 TIFF as the reference creates through GDAL (image3.cpp:872-893).
    planes: [bands, H, W] (or [H, W])."""
    lib = load_library()
    planes = np.ascontiguousarray(planes, np.float32)
    if planes.ndim == 2:
        planes = planes[None]
    b, h, w = planes.shape
    rc = lib.mis_write_tiff_f32(os.fsencode(path), _np_ptr(planes), w, h, b)
    if rc != 0:
        raise MisError(rc, "mis_write_tiff_f32 failed for %s" % path)


def load_camera_yaml(path):
    """Camera of image3.cpp:1014-1030: the A, R, T matrices of an OpenCV FileStorage YAML file."""
    lib = load_library()
    cam = Camera()
    rc = lib.mis_load_camera_yaml(os.fsencode(path), C.byref(cam))
    if rc != 0:
        raise MisError(rc, "mis_load_camera_yaml failed for %s" % path)
    return cam


def rasterize_leaves(records, width, height, fill=0.0):
    """The three bands of the reference's disparity GeoTIFF (image3.cpp:853-860) -> [3, H, W]."""
    lib = load_library()
    n = len(records)
    arr = (Leaf * max(n, 1))()
    for i, r in enumerate(records):
        arr[i].bbox1 = Rect(*r[0:4])
        arr[i].total_dx, arr[i].total_dy = r[4], r[5]
        arr[i].u1, arr[i].v1, arr[i].u2, arr[i].v2 = r[6:10]
        arr[i].peak = r[10]
    out = np.full((3, height, width), fill, np.float32)
    rc = lib.mis_rasterize_leaves(arr, n, width, height, _np_ptr(out))
    if rc != 0:
        raise MisError(rc, "mis_rasterize_leaves failed")
    return out


def _rects(rects):
    arr = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
    return arr


class Group:
    """Several GPUs driven by this process (mis_create_group): one context per listed device; a device
    may be listed more than once."""

    def __init__(self, devices):
        self._lib = load_library()
        self._h = C.c_void_p()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.mis_create_group(C.cast(ids, C.c_void_p), len(devices), C.byref(self._h))
        if rc != 0:
            raise MisError(rc, self._lib.mis_group_last_error(None).decode())

    def close(self):
        if self._h:
            self._lib.mis_destroy_group(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self._lib.mis_group_size(self._h)

    def _check(self, rc):
        if rc != 0:
            raise MisError(rc, self._lib.mis_group_last_error(self._h).decode())

    def upload_pair(self, img1, img2):
        """The whole pair to every device of the group (tile correlation reaches anywhere in both rasters)."""
        img1, img2 = np.ascontiguousarray(img1), np.ascontiguousarray(img2)
        if img1.shape != img2.shape or img1.dtype != img2.dtype or img1.ndim != 2:
            raise ValueError("image pair must be two 2-D rasters of equal shape and type")
        h, w = img1.shape
        self._check(self._lib.mis_group_upload_pair(self._h, _np_ptr(img1), _np_ptr(img2), _dtype_code(img1.dtype),
                                                    w, h, img1.strides[0]))
        self.shape = (h, w)

    def find_offsets(self, rects1, rects2, tile_w, tile_h):
        """Context.find_offsets with the batch shared out over the group's devices."""
        r1, r2 = _rects(rects1), _rects(rects2)
        if r1.shape != r2.shape:
            raise ValueError("rect lists differ in length")
        n = r1.shape[0]
        out = (Offset * max(n, 1))()
        self._check(self._lib.mis_group_xcorr_tiles(self._h, _np_ptr(r1), _np_ptr(r2), n, tile_w, tile_h, out))
        res = np.frombuffer(out, dtype=[("dx", "<i4"), ("dy", "<i4"), ("peak", "<f4"), ("dog_peak", "<f4")])
        return res[:n].copy()

    def disparity_map(self, min_size=32, max_leaves=1 << 16):
        """Context.disparity_map with every level's tile batches shared out over the group's devices."""
        h, w = self.shape
        leaves = (Leaf * max_leaves)()
        n, nt = C.c_int(0), C.c_int(0)
        self._check(self._lib.mis_group_disparity_map(self._h, w, h, min_size, leaves, max_leaves, C.byref(n), C.byref(nt)))
        if n.value > max_leaves:
            return self.disparity_map(min_size, n.value)
        return _leaf_records(leaves, min(n.value, max_leaves)), nt.value

    def ncc_dense(self, left, right, window, dmin, dmax, want_idisp=True, want_peak2=False, out=None):
        """Dense NCC of a host pair, row strips over the group's devices; same result as Context.ncc_dense.
        out: (disp, peak) float32 rasters of the caller's (page-locked ones are written by DMA)."""
        left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
        if left.shape != right.shape or left.dtype != right.dtype:
            raise ValueError("the two rasters must have one shape and type")
        h, w = left.shape
        disp = np.empty((h, w), np.float32) if out is None else out[0]
        peak = np.empty((h, w), np.float32) if out is None else out[1]
        idisp = np.empty((h, w), np.int16) if want_idisp else None
        peak2 = np.empty((h, w), np.float32) if want_peak2 else None
        rc = self._lib.mis_group_ncc_dense(self._h, _np_ptr(left), _np_ptr(right), _dtype_code(left.dtype), w, h,
                                           left.strides[0], window, dmin, dmax, _np_ptr(disp), _np_ptr(peak),
                                           _np_ptr(idisp), _np_ptr(peak2))
        if rc != 0:
            raise MisError(rc, self._lib.mis_group_last_error(self._h).decode())
        out = {"disp": disp, "peak": peak}
        if want_idisp:
            out["idisp"] = idisp
        if want_peak2:
            out["peak2"] = peak2
        return out


class Context:
    """One context per GPU and host thread; owns device scratch (mis_create / mis_destroy)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.mis_create(int(device), C.byref(self._h))
        if rc != 0:
            raise MisError(rc, self._lib.mis_last_error(None).decode())
        self.shape = None
        self._keep = None

    def close(self):
        if self._h:
            self._lib.mis_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise MisError(rc, self._lib.mis_last_error(self._h).decode())

    # ---- plumbing ----
    def set_stream(self, stream_ptr):
        self._check(self._lib.mis_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def profile(self, on=True):
        self._check(self._lib.mis_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """-> (tables_ms, main_kernel_ms, refine_ms, calls) summed since the last read."""
        a, b, c, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        self._check(self._lib.mis_profile_read(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def alloc_host(self, shape, dtype):
        """A numpy array over page-locked memory allocated by the library (free it with free_host(array))."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self._check(self._lib.mis_alloc_host(self._h, max(n, 1), C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        self._host_allocs = getattr(self, "_host_allocs", {})
        self._host_allocs[a.ctypes.data] = p.value
        return a

    def free_host(self, array):
        p = self._host_allocs.pop(array.ctypes.data)
        self._check(self._lib.mis_free_host(self._h, C.c_void_p(p)))

    def pin_host(self, array):
        """Page-lock a numpy array's memory (inputs that are uploaded, outputs that results are copied into)."""
        self._check(self._lib.mis_pin_host(self._h, _np_ptr(array), array.nbytes))

    def unpin_host(self, array):
        self._check(self._lib.mis_unpin_host(self._h, _np_ptr(array)))

    def synchronize(self):
        self._check(self._lib.mis_synchronize(self._h))

    # ---- image pair ----
    def upload_pair(self, img1, img2):
        img1 = np.ascontiguousarray(img1)
        img2 = np.ascontiguousarray(img2)
        if img1.shape != img2.shape or img1.dtype != img2.dtype or img1.ndim != 2:
            raise ValueError("image pair must be two 2-D rasters of equal shape and type")
        h, w = img1.shape
        self._check(self._lib.mis_upload_pair(self._h, _np_ptr(img1), _np_ptr(img2), _dtype_code(img1.dtype),
                                              w, h, img1.strides[0]))
        self.shape = (h, w)
        self.shapes2 = None
        self._keep = None

    def upload_pair2(self, img1, img2):
        """Two rasters of different size (the reference only warns about them, image3.cpp:995-998): each is copied into
        the top-left corner of a zeroed raster of the common size; disparity_map then keeps the two extents apart."""
        img1, img2 = np.ascontiguousarray(img1), np.ascontiguousarray(img2)
        if img1.dtype != img2.dtype or img1.ndim != 2 or img2.ndim != 2:
            raise ValueError("image pair must be two 2-D rasters of one type")
        self._check(self._lib.mis_upload_pair2(self._h, _np_ptr(img1), img1.shape[1], img1.shape[0], img1.strides[0],
                                               _np_ptr(img2), img2.shape[1], img2.shape[0], img2.strides[0], _dtype_code(img1.dtype)))
        self.shape = (max(img1.shape[0], img2.shape[0]), max(img1.shape[1], img2.shape[1]))
        self.shapes2 = (img1.shape, img2.shape)
        self._keep = None

    def bind_pair_torch(self, t1, t2):
        """Bind two 2-D CUDA tensors (uint8 / uint16-as-int16 storage is not supported: use uint8,
        torch.uint16 or float32) without copying; the tensors are kept alive by the context."""
        import torch
        if t1.shape != t2.shape or t1.dtype != t2.dtype or t1.dim() != 2 or not t1.is_cuda:
            raise ValueError("need two 2-D CUDA tensors of equal shape and type")
        code = {torch.uint8: U8, torch.uint16: U16, torch.float32: F32}[t1.dtype]
        if t1.stride(1) != 1 or t2.stride(1) != 1 or t1.stride(0) != t2.stride(0):
            raise ValueError("rows must be contiguous with equal strides")
        h, w = t1.shape
        self._check(self._lib.mis_bind_pair_device(self._h, _dev_ptr(t1), _dev_ptr(t2), code, w, h,
                                                   t1.stride(0) * t1.element_size()))
        self.shape = (h, w)
        self._keep = (t1, t2)

    def invalidate_pair(self):
        """The bound rasters' pixels changed in place: cached tables are rebuilt by the next call."""
        self._check(self._lib.mis_invalidate_pair(self._h))

    # ---- dense NCC ----
    def ncc_dense(self, window, dmin, dmax, want_idisp=True, want_peak2=False):
        h, w = self.shape
        disp = np.empty((h, w), np.float32)
        peak = np.empty((h, w), np.float32)
        idisp = np.empty((h, w), np.int16) if want_idisp else None
        peak2 = np.empty((h, w), np.float32) if want_peak2 else None
        self._check(self._lib.mis_ncc_dense(self._h, window, dmin, dmax, _np_ptr(disp), _np_ptr(peak),
                                            _np_ptr(idisp), _np_ptr(peak2)))
        out = {"disp": disp, "peak": peak}
        if want_idisp:
            out["idisp"] = idisp
        if want_peak2:
            out["peak2"] = peak2
        return out

    def ncc_dense_rows(self, window, dmin, dmax, row0, nrows, disp, peak, idisp=None, peak2=None):
        """Rows [row0, row0+nrows) into host rasters of the frame's shape (the other rows are left alone)."""
        self._check(self._lib.mis_ncc_dense_rows(self._h, window, dmin, dmax, row0, nrows, _np_ptr(disp), _np_ptr(peak),
                                                 _np_ptr(idisp), _np_ptr(peak2)))

    def ncc_dense_torch(self, window, dmin, dmax, disp, peak, idisp=None, peak2=None, row0=None, nrows=None):
        """Enqueue on the context's stream; outputs are CUDA tensors of the frame's shape."""
        if row0 is None:
            self._check(self._lib.mis_ncc_dense_device(self._h, window, dmin, dmax, _dev_ptr(disp), _dev_ptr(peak),
                                                       _dev_ptr(idisp), _dev_ptr(peak2)))
        else:
            self._check(self._lib.mis_ncc_dense_rows_device(self._h, window, dmin, dmax, row0, nrows,
                                                            _dev_ptr(disp), _dev_ptr(peak), _dev_ptr(idisp),
                                                            _dev_ptr(peak2)))

    # ---- FFT tile correlation ----
    def find_offsets(self, rects1, rects2, tile_w, tile_h):
        """Batched findOffset (image3.cpp:346): returns arrays dx, dy, peak, dog_peak."""
        r1, r2 = _rects(rects1), _rects(rects2)
        if r1.shape != r2.shape:
            raise ValueError("rect lists differ in length")
        n = r1.shape[0]
        out = (Offset * n)()
        self._check(self._lib.mis_xcorr_tiles(self._h, _np_ptr(r1), _np_ptr(r2), n, tile_w, tile_h, out))
        res = np.frombuffer(out, dtype=[("dx", "<i4"), ("dy", "<i4"), ("peak", "<f4"), ("dog_peak", "<f4")])
        return res.copy()

    def find_offsets_torch(self, rects1, rects2, tile_w, tile_h, out):
        """Batched findOffset on device memory: rects1 / rects2 int32 CUDA tensors [n, 4], out an int32
        CUDA tensor [n, 4] that receives (dx, dy, peak bits, DoG peak bits); only enqueues."""
        n = rects1.shape[0]
        self._check(self._lib.mis_xcorr_tiles_device(self._h, _dev_ptr(rects1), _dev_ptr(rects2), n, tile_w, tile_h,
                                                     _dev_ptr(out)))

    def find_offset(self, bbox1, bbox2, width, height):
        """findOffset(bbox1, bbox2, width, height, &ox, &oy) -> (peak, ox, oy)."""
        r = self.find_offsets([bbox1], [bbox2], width, height)[0]
        return float(r["peak"]), int(r["dx"]), int(r["dy"])

    def tile_surfaces(self, bbox1, bbox2, width, height):
        surf = np.empty((height, width), np.float32)
        dog = np.empty((height, width), np.float32)
        self._check(self._lib.mis_xcorr_tile_surfaces(self._h, Rect(*bbox1), Rect(*bbox2), width, height,
                                                      _np_ptr(surf), _np_ptr(dog)))
        return surf, dog

    def fft2d_forward(self, imgs):
        """image-fft.cpp CalcFFT for a batch of rasters [batch, h, w] (or one [h, w])."""
        a = np.ascontiguousarray(imgs)
        single = a.ndim == 2
        if single:
            a = a[None]
        b, h, w = a.shape
        out = np.empty((b, h, w), np.complex64)
        self._check(self._lib.mis_fft2d_forward(self._h, _np_ptr(a), _dtype_code(a.dtype), w, h, b, _np_ptr(out)))
        return out[0] if single else out

    def upload_extra_bands(self, bands1, bands2):
        """Bands 2.. of the two rasters (lists of numpy rasters shaped like the uploaded pair): the
        tile correlation multiplies the normalised surfaces of all bands (image3.cpp:382-418)."""
        bands1 = [np.ascontiguousarray(b) for b in bands1]
        bands2 = [np.ascontiguousarray(b) for b in bands2]
        n = len(bands1)
        if n != len(bands2):
            raise ValueError("both rasters need the same number of bands")
        for b in bands1 + bands2:
            if b.shape != self.shape:
                raise ValueError("extra bands must have the shape of the uploaded pair")
        p1 = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bands1])
        p2 = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bands2])
        stride = bands1[0].strides[0] if n else 0
        self._check(self._lib.mis_upload_extra_bands(self._h, n, p1, p2, C.c_size_t(stride)))

    def masked_peak(self, bands, mask_pct):
        """findCPPMax (image2.cpp:216-308): bands [n, rows, cols] float32 -> (peak, offset_x, offset_y)."""
        bands = np.ascontiguousarray(bands, np.float32)
        if bands.ndim == 2:
            bands = bands[None]
        nb, rows, cols = bands.shape
        ox, oy, pk = C.c_int(0), C.c_int(0), C.c_float(0)
        self._check(self._lib.mis_masked_peak(self._h, _np_ptr(bands), cols, rows, nb, C.c_float(mask_pct),
                                              C.byref(ox), C.byref(oy), C.byref(pk)))
        return float(pk.value), ox.value, oy.value

    def xcorr_surface_u8(self):
        h, w = self.shape
        out = np.empty((h, w), np.uint8)
        self._check(self._lib.mis_xcorr_surface_u8(self._h, _np_ptr(out)))
        return out

    # ---- coarse-to-fine driver ----
    def disparity_map(self, min_size=32, max_leaves=1 << 16):
        """calcDisparityMap (image3.cpp:676) for the bound pair -> (leaf records, findOffset count).
        Records: (minx, miny, w, h, total_dx, total_dy, u1, v1, u2, v2, peak), depth-first order."""
        h, w = self.shape
        leaves = (Leaf * max_leaves)()
        n, nt = C.c_int(0), C.c_int(0)
        two = getattr(self, "shapes2", None)
        if two is not None and self.shape == (max(two[0][0], two[1][0]), max(two[0][1], two[1][1])):
            self._check(self._lib.mis_disparity_map2(self._h, two[0][1], two[0][0], two[1][1], two[1][0], min_size, leaves,
                                                     max_leaves, C.byref(n), C.byref(nt)))
        else:
            self._check(self._lib.mis_disparity_map(self._h, w, h, min_size, leaves, max_leaves, C.byref(n), C.byref(nt)))
        if n.value > max_leaves:                    # same answer in a buffer that holds them all
            return self.disparity_map(min_size, n.value)
        return _leaf_records(leaves, min(n.value, max_leaves)), nt.value

    # ---- triangulation ----
    def calculate_xyz(self, cam1, cam2, pts):
        """calculateXYZ (image3.cpp:148-264) for rows of (u1, v1, u2, v2)."""
        p = np.ascontiguousarray(pts, np.int32).reshape(-1, 4)
        out = np.empty((p.shape[0], 3), np.float32)
        self._check(self._lib.mis_triangulate(self._h, C.byref(cam1), C.byref(cam2), _np_ptr(p), p.shape[0],
                                              _np_ptr(out)))
        return out

    def triangulate_disparity_torch(self, cam1, cam2, disp, peak, min_peak, xyz, count=True, subpixel=False):
        """subpixel: the fractional column x - disp goes into calculateXYZ's arithmetic instead of the
        rounded one (MIS_TRI_SUBPIXEL; the reference only triangulates integer pixels)."""
        n = C.c_longlong(0)
        if subpixel:
            self._check(self._lib.mis_triangulate_disparity_ex_device(self._h, C.byref(cam1), C.byref(cam2), _dev_ptr(disp),
                                                                      _dev_ptr(peak), C.c_float(min_peak), TRI_SUBPIXEL,
                                                                      _dev_ptr(xyz), C.byref(n) if count else None))
        else:
            self._check(self._lib.mis_triangulate_disparity_device(self._h, C.byref(cam1), C.byref(cam2), _dev_ptr(disp),
                                                                   _dev_ptr(peak), C.c_float(min_peak), _dev_ptr(xyz),
                                                                   C.byref(n) if count else None))
        return n.value if count else None

    def triangulate_disparity_rows_torch(self, cam1, cam2, disp, peak, min_peak, xyz, row0, nrows, frame_row0,
                                         count=True, subpixel=False):
        """Rows [row0, row0+nrows) of rasters that are a strip of a frame: raster row k is frame row k + frame_row0."""
        n = C.c_longlong(0)
        self._check(self._lib.mis_triangulate_disparity_rows_device(
            self._h, C.byref(cam1), C.byref(cam2), _dev_ptr(disp), _dev_ptr(peak), C.c_float(min_peak),
            TRI_SUBPIXEL if subpixel else 0, disp.shape[1], row0, nrows, frame_row0, _dev_ptr(xyz),
            C.byref(n) if count else None))
        return n.value if count else None

    def triangulate_disparity_noalloc_torch(self, cam1, cam2, disp, peak, min_peak, xyz, n_matched=None):
        """The entry point that allocates nothing: raster size from the tensors, the matched count is
        written to the int64 CUDA tensor n_matched (optional); nothing is synchronised."""
        h, w = disp.shape
        self._check(self._lib.mis_triangulate_disparity_noalloc(self._h, C.byref(cam1), C.byref(cam2), _dev_ptr(disp),
                                                                _dev_ptr(peak), C.c_float(min_peak), w, h,
                                                                _dev_ptr(xyz), _dev_ptr(n_matched)))
