"""ctypes binding of the C ABI in ``include/nova_hip.h`` (``libnova_hip.so``).

This is plumbing only: every call goes straight to the HIP library.  There is no CPU fallback —
if the library is missing, or the box has no HIP device, calls raise ``NovaHipError``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NOVA_HIP_LIB") or os.path.join(_HERE, "libnova_hip.so")  # NOVA_HIP_LIB: A/B builds of the same library

# option ids (include/nova_hip.h)
OPT_MAX_DEPTH, OPT_TRAVERSAL, OPT_COUNT_WORK, OPT_TILE_RANK, OPT_TILE_RANKS, OPT_TRAVERSE_KERNEL, OPT_TUNE, OPT_SORT_RAYS = \
    1, 2, 3, 4, 5, 6, 7, 8
OPT_CLAMP_TO_EDGE = 9
OPT_ALLOC_CALLS = 10  # read-only
OPT_MIS_PRETEST = 11
OPT_ASYNC_LAUNCH = 12
OPT_CALL_LANES = 13
ADDRESS_WRAP, ADDRESS_CLAMP = 0, 1
FILTER_NEAREST, FILTER_LINEAR = 0, 1

EXPORTS = [
    "nova_hip_error_string", "nova_hip_init", "nova_hip_device_count", "nova_hip_device_name",
    "nova_hip_synchronize", "nova_hip_shutdown", "nova_hip_malloc", "nova_hip_free",
    "nova_hip_host_alloc", "nova_hip_host_free", "nova_hip_memcpy_h2d", "nova_hip_memcpy_d2h", "nova_hip_memcpy_d2d", "nova_hip_fill",
    "nova_hip_image_create", "nova_hip_image_destroy", "nova_hip_image_read",
    "nova_hip_image_write", "nova_hip_image_copy", "nova_hip_image_fill",
    "nova_hip_launch_raytrace", "nova_hip_launch_accumulate", "nova_hip_launch_post_process",
    "nova_hip_render_samples", "nova_hip_trace_rays", "nova_hip_trace_rays_noalloc", "nova_hip_set_option", "nova_hip_get_option", "nova_hip_get_counters",
    "nova_hip_reset_counters", "nova_hip_bvh_max_nodes", "nova_hip_build_bvh",
    "nova_hip_build_bvh_device", "nova_hip_validate_scene_pack",
    "nova_hip_tiles_stack_bytes", "nova_hip_tiles_pack", "nova_hip_tiles_unpack",
    "nova_hip_tiles_unique_id", "nova_hip_tiles_comm_init", "nova_hip_tiles_comm_adopt",
    "nova_hip_tiles_comm_destroy", "nova_hip_tiles_comm_info", "nova_hip_tiles_allgather",
    "nova_hip_denoise_read", "nova_hip_denoise_write", "nova_hip_measure_device",
]
# error codes (include/nova_hip.h)
E_INVALID_ARGUMENT, E_UNKNOWN_BUFFER, E_BAD_SCENE, E_NO_DEVICE, E_TOO_LARGE, E_NOT_CACHED = -1, -2, -3, -4, -5, -6
E_NO_RCCL, E_NO_COMM, E_RCCL_BASE = -7, -8, -1000
UNIQUE_ID_BYTES = 128


class NovaHipError(RuntimeError):
    """Mirrors nova::Error (src/backend/cuda/types/error.hpp:10-17): carries the error code."""

    def __init__(self, code: int, what: str):
        super().__init__(f"{what} (code {code})")
        self.code = code

    def err(self) -> int:
        return self.code


class Counters(ctypes.Structure):
    _fields_ = [("rays_primary", ctypes.c_uint64), ("rays_bounce", ctypes.c_uint64),
                ("rays_shadow", ctypes.c_uint64), ("node_visits", ctypes.c_uint64),
                ("triangle_tests", ctypes.c_uint64), ("path_segments", ctypes.c_uint64),
                ("traverse_ms", ctypes.c_double), ("shade_ms", ctypes.c_double),
                ("traverse_launches", ctypes.c_uint64), ("camera_traverse_ms", ctypes.c_double),
                ("camera_traverse_launches", ctypes.c_uint64), ("rays_mis_pretested", ctypes.c_uint64),
                ("rays_shadow_skipped", ctypes.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ImageDesc(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("layers", ctypes.c_uint32), ("texel_bytes", ctypes.c_uint32),
                ("address_mode", ctypes.c_uint32), ("filter_mode", ctypes.c_uint32),
                ("normalized_coords", ctypes.c_uint32)]


_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NovaHipError(-4, f"{LIB_PATH} is missing; build it with __graft_entry__.build()")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u32, u64, i32, i64, sz = (ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32,
                                  ctypes.c_int64, ctypes.c_size_t)
    lib.nova_hip_error_string.restype = ctypes.c_char_p
    lib.nova_hip_error_string.argtypes = [ctypes.c_int]
    sig = {
        "nova_hip_init": [ctypes.c_int],
        "nova_hip_device_count": [vp],
        "nova_hip_device_name": [ctypes.c_char_p, sz],
        "nova_hip_synchronize": [],
        "nova_hip_shutdown": [],
        "nova_hip_malloc": [vp, sz],
        "nova_hip_free": [vp],
        "nova_hip_host_alloc": [vp, sz],
        "nova_hip_host_free": [vp],
        "nova_hip_memcpy_h2d": [vp, vp, sz],
        "nova_hip_memcpy_d2h": [vp, vp, sz],
        "nova_hip_memcpy_d2d": [vp, vp, sz],
        "nova_hip_fill": [vp, vp, sz, sz],
        "nova_hip_image_create": [vp, u32, u32, u32, u32, u32, u32, u32, vp],
        "nova_hip_image_destroy": [vp],
        "nova_hip_image_read": [vp, vp],
        "nova_hip_image_write": [vp, vp],
        "nova_hip_image_copy": [vp, vp],
        "nova_hip_image_fill": [vp, vp],
        "nova_hip_launch_raytrace": [vp, u32, vp, u32, u32, vp, vp, vp, vp, u32, vp, vp, u32, vp, vp],
        "nova_hip_launch_accumulate": [i32, u32] + [vp] * 9 + [u32, u32],
        "nova_hip_launch_post_process": [vp, vp, vp, u32, u32],
        "nova_hip_render_samples": [vp, vp, u32, i32, vp, u32, u32, vp, vp, vp, vp, u32, vp, vp, u32],
        "nova_hip_trace_rays": [vp, vp, vp, vp, vp, u64],
        "nova_hip_trace_rays_noalloc": [vp, vp, vp, vp, vp, u64],
        "nova_hip_set_option": [ctypes.c_int, i64],
        "nova_hip_get_option": [ctypes.c_int, vp],
        "nova_hip_get_counters": [vp],
        "nova_hip_reset_counters": [],
        "nova_hip_build_bvh": [vp, u64, vp, u64, vp, vp, vp, vp],
        "nova_hip_build_bvh_device": [vp, u64, vp, u64, vp, vp, vp, vp],
        "nova_hip_validate_scene_pack": [vp, vp, vp],
        "nova_hip_measure_device": [vp],
        "nova_hip_denoise_read": [vp, vp, vp, u32, u32, vp],
        "nova_hip_denoise_write": [vp, u32, u32],
        "nova_hip_tiles_pack": [vp, u32, u32, vp],
        "nova_hip_tiles_unpack": [vp, u32, vp],
        "nova_hip_tiles_unique_id": [vp],
        "nova_hip_tiles_comm_init": [vp, u32, u32],
        "nova_hip_tiles_comm_adopt": [vp],
        "nova_hip_tiles_comm_destroy": [],
        "nova_hip_tiles_comm_info": [vp],
        "nova_hip_tiles_allgather": [vp],
    }
    lib.nova_hip_tiles_stack_bytes.restype = u64
    lib.nova_hip_tiles_stack_bytes.argtypes = [u32, u32, u32, u32]
    lib.nova_hip_bvh_max_nodes.restype = u64
    lib.nova_hip_bvh_max_nodes.argtypes = [u64]
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int):
    if code != 0:
        raise NovaHipError(code, load().nova_hip_error_string(code).decode())


def init(device: int = 0):
    check(load().nova_hip_init(device))


def device_count() -> int:
    n = ctypes.c_int(0)
    check(load().nova_hip_device_count(ctypes.byref(n)))
    return n.value


def device_name() -> str:
    buf = ctypes.create_string_buffer(256)
    check(load().nova_hip_device_name(buf, 256))
    return buf.value.decode()


def set_option(opt: int, value: int):
    check(load().nova_hip_set_option(opt, value))


def get_option(opt: int) -> int:
    v = ctypes.c_int64(0)
    check(load().nova_hip_get_option(opt, ctypes.byref(v)))
    return v.value


def counters() -> dict:
    c = Counters()
    check(load().nova_hip_get_counters(ctypes.byref(c)))
    return c.as_dict()


class DeviceRates(ctypes.Structure):
    _fields_ = [("hbm_copy_gbs", ctypes.c_double), ("fp32_fma_tflops", ctypes.c_double)]


def measure_device() -> dict:
    """nova_hip_measure_device: the roofline's measured denominators (float4 copy GB/s, FP32 FMA TFLOP/s)."""
    r = DeviceRates()
    check(load().nova_hip_measure_device(ctypes.byref(r)))
    return {"hbm_copy_gbs": r.hbm_copy_gbs, "fp32_fma_tflops": r.fp32_fma_tflops}


def reset_counters():
    check(load().nova_hip_reset_counters())


def synchronize():
    check(load().nova_hip_synchronize())


class Buffer:
    """nova::Buffer<T> (src/backend/cuda/types/buffer.hpp:11-86) over nova_hip_malloc."""

    def __init__(self, data: Optional[np.ndarray] = None, nbytes: Optional[int] = None):
        lib = load()
        if data is not None:
            data = np.ascontiguousarray(data)
            nbytes = data.nbytes
        if not nbytes:
            raise ValueError("Cannot build an empty Buffer")  # AcceleratorException in the reference
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib.nova_hip_malloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value
        if data is not None:
            self.write(data)

    def write(self, data: np.ndarray):
        data = np.ascontiguousarray(data)
        assert data.nbytes <= self.nbytes
        check(load().nova_hip_memcpy_h2d(self.ptr, data.ctypes.data, data.nbytes))

    def read(self, dtype, count: Optional[int] = None) -> np.ndarray:
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(n, dtype)
        check(load().nova_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            check(load().nova_hip_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Image:
    """Image2DRead/Write/ReadWrite/Array (src/backend/cuda/types/image2D*.hpp) as a linear buffer."""

    def __init__(self, width: int, height: int, layers: int = 1, texel_bytes: int = 16,
                 address_mode: int = ADDRESS_CLAMP, filter_mode: int = FILTER_LINEAR,
                 normalized: bool = True, data: Optional[np.ndarray] = None):
        if width == 0 or height == 0 or layers == 0:
            raise ValueError("Cannot build an empty Image2D")
        self.width, self.height, self.layers, self.texel_bytes = width, height, layers, texel_bytes
        h = ctypes.c_void_p()
        src = None
        if data is not None:
            data = np.ascontiguousarray(data)
            assert data.nbytes == width * height * layers * texel_bytes, (data.shape, width, height)
            src = data.ctypes.data
        check(load().nova_hip_image_create(ctypes.byref(h), texel_bytes, width, height, layers,
                                           address_mode, filter_mode, 1 if normalized else 0, src))
        self.handle = h.value

    def read(self) -> np.ndarray:
        if self.texel_bytes == 16:
            out = np.empty((self.layers, self.height, self.width, 4), np.float32)
        else:
            out = np.empty((self.layers, self.height, self.width, 4), np.uint8)
        check(load().nova_hip_image_read(self.handle, out.ctypes.data))
        return out[0] if self.layers == 1 else out

    def read_into(self, host: "PinnedArray") -> np.ndarray:
        """read_image2D into page-locked host memory the caller keeps (nova_hip_host_alloc)."""
        assert host.nbytes >= self.width * self.height * self.layers * self.texel_bytes
        check(load().nova_hip_image_read(self.handle, host.ptr))
        return host.array

    def write(self, data: np.ndarray):
        data = np.ascontiguousarray(data)
        assert data.nbytes == self.width * self.height * self.layers * self.texel_bytes
        check(load().nova_hip_image_write(self.handle, data.ctypes.data))

    def copy_from(self, src: "Image"):
        check(load().nova_hip_image_copy(self.handle, src.handle))

    def fill(self, texel):
        t = np.ascontiguousarray(texel, np.float32 if self.texel_bytes == 16 else np.uint8)
        assert t.nbytes == self.texel_bytes
        check(load().nova_hip_image_fill(self.handle, t.ctypes.data))

    @property
    def data_ptr(self) -> int:
        return ImageDesc.from_address(self.handle).data

    def destroy(self):
        if getattr(self, "handle", None):
            check(load().nova_hip_image_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class PinnedArray:
    """A numpy view of page-locked host memory from nova_hip_host_alloc."""

    def __init__(self, shape, dtype):
        self.array = None
        dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(shape)) * dtype.itemsize
        p = ctypes.c_void_p()
        check(load().nova_hip_host_alloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_uint8)),
                                           shape=(self.nbytes,)).view(dtype).reshape(shape)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            check(load().nova_hip_host_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _h(img: Optional[Image]):
    return None if img is None else img.handle


def launch_raytrace(params: np.ndarray, time: int, temp_color1: Image, width: int, height: int,
                    triangles: Buffer, tri_meta: Buffer, bvh: Buffer, lights: Optional[Buffer],
                    num_lights: int, materials: Image, sky: Image, denoise_available: int,
                    albedo1: Optional[Image], normal1: Optional[Image]):
    params = np.ascontiguousarray(params)
    check(load().nova_hip_launch_raytrace(
        params.ctypes.data, time & 0xFFFFFFFF, _h(temp_color1), width, height, triangles.ptr,
        tri_meta.ptr, bvh.ptr, None if lights is None else lights.ptr, num_lights, _h(materials),
        _h(sky), denoise_available, _h(albedo1), _h(normal1)))


def launch_accumulate(sample_index: int, denoise_available: int, color1, albedo1, normal1,
                      prev_color, prev_albedo, prev_normal, color2, albedo2, normal2, width, height):
    check(load().nova_hip_launch_accumulate(
        sample_index, denoise_available, _h(color1), _h(albedo1), _h(normal1), _h(prev_color),
        _h(prev_albedo), _h(prev_normal), _h(color2), _h(albedo2), _h(normal2), width, height))


def launch_post_process(params: np.ndarray, color2: Image, pixels: Image, width: int, height: int):
    params = np.ascontiguousarray(params)
    check(load().nova_hip_launch_post_process(params.ctypes.data, _h(color2), _h(pixels), width,
                                              height))


class FrameImages(ctypes.Structure):
    """nova_hip_frame_images (include/nova_hip.h): the images Raytracer keeps per frame."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "temp_color1", "temp_color2", "prev_color", "albedo_feature1", "normal_feature1",
        "albedo_feature2", "normal_feature2", "prev_albedo_feature", "prev_normal_feature", "pixels")]


def render_samples(params: np.ndarray, times, first_sample_index: int, images: FrameImages,
                   width: int, height: int, triangles: Buffer, tri_meta: Buffer, bvh: Buffer,
                   lights: Optional[Buffer], num_lights: int, materials: Image, sky: Image,
                   denoise_available: int):
    params = np.ascontiguousarray(params)
    times = np.ascontiguousarray(np.asarray(times, np.uint64) & 0xFFFFFFFF, np.uint32)
    check(load().nova_hip_render_samples(
        params.ctypes.data, times.ctypes.data if times.size else None, times.size,
        first_sample_index, ctypes.addressof(images), width, height, triangles.ptr, tri_meta.ptr,
        bvh.ptr, None if lights is None else lights.ptr, num_lights, _h(materials), _h(sky),
        denoise_available))


def trace_rays(triangles: Buffer, bvh: Buffer, point_tmax: Buffer, dir_flags: Buffer, hits: Buffer,
               num_rays: int):
    check(load().nova_hip_trace_rays(triangles.ptr, bvh.ptr, point_tmax.ptr, dir_flags.ptr,
                                     hits.ptr, num_rays))


def trace_rays_noalloc(triangles: Buffer, bvh: Buffer, point_tmax: Buffer, dir_flags: Buffer, hits: Buffer,
                       num_rays: int):
    """nova_hip_trace_rays_noalloc: the same, for callers that cannot allocate (include/nova_hip.h)."""
    check(load().nova_hip_trace_rays_noalloc(triangles.ptr, bvh.ptr, point_tmax.ptr, dir_flags.ptr,
                                             hits.ptr, num_rays))


class DenoisePlanes(ctypes.Structure):
    _fields_ = [(n, ctypes.POINTER(ctypes.c_float)) for n in ("color", "albedo", "normal", "output")]


def denoise_read(color: Image, albedo: Image, normal: Image):
    """nova_hip_denoise_read: returns four numpy views [h, w, 3] of the library's pinned host planes
    (color, albedo, normal, output) — what the reference hands to its denoiser (raytracer.cpp:216-227)."""
    planes = DenoisePlanes()
    check(load().nova_hip_denoise_read(color.handle, albedo.handle, normal.handle, color.width, color.height,
                                       ctypes.byref(planes)))
    shape = (color.height, color.width, 3)
    return tuple(np.ctypeslib.as_array(getattr(planes, n), shape=shape) for n in ("color", "albedo", "normal", "output"))


def denoise_write(color: Image):
    """nova_hip_denoise_write: the `output` plane back into the image (pack + write_image2D, raytracer.cpp:240-241)."""
    check(load().nova_hip_denoise_write(color.handle, color.width, color.height))


def tiles_stack_bytes(width: int, height: int, ranks: int, texel_bytes: int) -> int:
    return int(load().nova_hip_tiles_stack_bytes(width, height, ranks, texel_bytes))


def tiles_pack(image: Image, rank: int, ranks: int, stack_ptr: int):
    """nova_hip_tiles_pack: `rank`'s 64x64 tiles of `image` into the contiguous stack at device pointer stack_ptr."""
    check(load().nova_hip_tiles_pack(image.handle, rank, ranks, stack_ptr))


def tiles_unpack(image: Image, ranks: int, stacks_ptr: int):
    """nova_hip_tiles_unpack: the all-gathered stacks at device pointer stacks_ptr into the whole image."""
    check(load().nova_hip_tiles_unpack(image.handle, ranks, stacks_ptr))


class CommInfo(ctypes.Structure):
    """nova_hip_comm_info (include/nova_hip.h)."""
    _fields_ = [("rank", ctypes.c_uint32), ("ranks", ctypes.c_uint32), ("device", ctypes.c_int32),
                ("rccl_version", ctypes.c_int32), ("pci_bus_id", ctypes.c_char * 32), ("rccl_path", ctypes.c_char * 256)]


def tiles_unique_id() -> bytes:
    """nova_hip_tiles_unique_id (rank 0): the 128 bytes of an ncclUniqueId, to be handed to every rank."""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    check(load().nova_hip_tiles_unique_id(buf))
    return buf.raw


def tiles_comm_init(unique_id: bytes, rank: int, ranks: int):
    """nova_hip_tiles_comm_init: ncclCommInitRank on the library's device (collective over the ranks)."""
    assert len(unique_id) == UNIQUE_ID_BYTES
    check(load().nova_hip_tiles_comm_init(ctypes.create_string_buffer(unique_id, UNIQUE_ID_BYTES), rank, ranks))


def tiles_comm_destroy():
    check(load().nova_hip_tiles_comm_destroy())


def tiles_comm_info() -> dict:
    c = CommInfo()
    check(load().nova_hip_tiles_comm_info(ctypes.byref(c)))
    return {"rank": c.rank, "ranks": c.ranks, "device": c.device, "rccl_version": c.rccl_version,
            "pci_bus_id": c.pci_bus_id.decode(), "rccl_path": c.rccl_path.decode()}


def tiles_allgather(image: Image):
    """nova_hip_tiles_allgather: pack -> ncclAllGather -> unpack on the library's stream; afterwards every rank's
    image holds every rank's tiles."""
    check(load().nova_hip_tiles_allgather(image.handle))


class BuildStats(ctypes.Structure):
    _fields_ = [("device_ms", ctypes.c_float), ("levels", ctypes.c_uint32), ("peak_bin_bytes", ctypes.c_uint64),
                ("pack_ms", ctypes.c_float)]


def build_bvh(verts: np.ndarray):
    """BVH::build + the TriangleData loop on the device (include/nova_hip.h: nova_hip_build_bvh).
    verts: float32 [n, 9] or [n, 3, 3].  Returns (bvh BVH_NODE[n_nodes], triangles float32[n, 12],
    order uint32[n], stats dict)."""
    from .scene import BVH_NODE
    lib = load()
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
    n = v.shape[0]
    cap = int(lib.nova_hip_bvh_max_nodes(n)) if n else 1
    bvh = np.zeros(cap, BVH_NODE)
    tris = np.zeros((n, 12), np.float32)
    order = np.zeros(n, np.uint32)
    n_nodes = ctypes.c_uint64(0)
    stats = BuildStats()
    check(lib.nova_hip_build_bvh(v.ctypes.data, n, bvh.ctypes.data, cap, ctypes.byref(n_nodes), tris.ctypes.data,
                                 order.ctypes.data, ctypes.byref(stats)))
    return bvh[: n_nodes.value].copy(), tris, order, {"device_ms": stats.device_ms, "levels": stats.levels,
                                                     "peak_bin_bytes": stats.peak_bin_bytes}


def build_bvh_device(verts: "Buffer", n: int):
    """nova_hip_build_bvh_device: verts is a device Buffer of n*9 floats.  Returns (bvh Buffer, triangles
    Buffer, order Buffer, n_nodes, stats dict); nothing leaves the device."""
    lib = load()
    cap = int(lib.nova_hip_bvh_max_nodes(n))
    bvh, tris, order = Buffer(nbytes=cap * 32), Buffer(nbytes=n * 48), Buffer(nbytes=n * 4)
    n_nodes = ctypes.c_uint64(0)
    stats = BuildStats()
    check(lib.nova_hip_build_bvh_device(verts.ptr, n, bvh.ptr, cap, ctypes.byref(n_nodes), tris.ptr, order.ptr,
                                        ctypes.byref(stats)))
    return bvh, tris, order, int(n_nodes.value), {"device_ms": stats.device_ms, "levels": stats.levels,
                                                  "peak_bin_bytes": stats.peak_bin_bytes, "pack_ms": stats.pack_ms}


def validate_scene_pack(bvh: "Buffer", tris: "Buffer") -> int:
    """Words in which the cached (device-made) traversal layout differs from the host packer's."""
    d = ctypes.c_uint64(0)
    check(load().nova_hip_validate_scene_pack(bvh.ptr, tris.ptr, ctypes.byref(d)))
    return int(d.value)
