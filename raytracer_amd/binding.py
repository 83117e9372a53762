"""ctypes binding of raytracer_amd/librt_hip.so (the C ABI of include/rt_hip.h).

The library is the product; this module only marshals arguments.  There is no
CPU fallback: if the shared library or a HIP device is missing the calls raise.
"""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

from .scene_io import Scene

import os

_PKG = pathlib.Path(__file__).resolve().parent
# RT_HIP_LIBRARY selects another build of the same library (A/B runs of kernel variants).
LIB_PATH = pathlib.Path(os.environ.get("RT_HIP_LIBRARY", _PKG / "librt_hip.so"))

RT_OK = 0
RNG_ROW_SERIAL, RNG_PIXEL = 0, 1
INTERSECT_WIDE8, INTERSECT_SCALAR, INTERSECT_WIDE4 = 0, 1, 2
FLAG_OUTPUT_SUM, FLAG_TIME_KERNEL, FLAG_COMPACT_TAILS, FLAG_ACCUMULATE, FLAG_NO_PRIMARY_CULL, FLAG_DENSE_TRANSFORMS = 1, 2, 4, 8, 16, 32

# Every symbol include/rt_hip.h declares.
EXPORTS = (
    "rt_abi_version", "rt_status_string", "rt_create", "rt_destroy", "rt_error_string",
    "rt_get_device_info", "rt_default_params", "rt_local_row_count", "rt_global_row",
    "rt_upload_world", "rt_reserve", "rt_load_accumulator", "rt_render", "rt_render_device", "rt_render_device_noalloc",
    "rt_render_multi", "rt_combine_device", "rt_resolve_device", "rt_resolve",
    "rt_last_kernel_ms",
)


class RenderParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
        ("band_rows", C.c_uint32), ("band_index", C.c_uint32), ("band_count", C.c_uint32),
        ("max_bounces", C.c_uint32),
        ("min_hit_distance", C.c_float), ("plane_tolerance", C.c_float),
        ("seed_constant", C.c_uint32), ("rng_mode", C.c_uint32),
        ("rows_per_order", C.c_uint32), ("lanes_per_pixel", C.c_uint32), ("batch_offset", C.c_uint32),
        ("intersect_mode", C.c_uint32), ("flags", C.c_uint32),
    ]


class World(C.Structure):
    _fields_ = [
        ("material_count", C.c_uint32), ("materials", C.c_void_p),
        ("plane_count", C.c_uint32), ("planes", C.c_void_p),
        ("sphere_count", C.c_uint32), ("spheres", C.c_void_p),
        ("rectangle_count", C.c_uint32), ("rectangles", C.c_void_p),
        ("camera", C.c_void_p),
    ]


class DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128), ("arch", C.c_char * 32),
        ("compute_units", C.c_uint32), ("clock_khz", C.c_uint32),
        ("lds_bytes_per_workgroup", C.c_uint32), ("global_memory_bytes", C.c_uint64),
        ("fp32_vector_peak_tflops", C.c_double),
    ]


class RtError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rt status {status}: {message}")
        self.status = status


_lib = None


def load_library() -> C.CDLL:
    """Loads librt_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch wheels bundle their own ROCm runtime.  A process that uses both must load torch's
    # copy first, so that librt_hip.so binds to the same libamdhip64 instead of bringing up a
    # second HIP runtime (after which torch finds no GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(LIB_PATH))
    L.rt_abi_version.restype = C.c_uint32
    L.rt_status_string.restype = C.c_char_p
    L.rt_status_string.argtypes = [C.c_int]
    L.rt_create.restype = C.c_int
    L.rt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.rt_destroy.restype = None
    L.rt_destroy.argtypes = [C.c_void_p]
    L.rt_error_string.restype = C.c_char_p
    L.rt_error_string.argtypes = [C.c_void_p]
    L.rt_get_device_info.restype = C.c_int
    L.rt_get_device_info.argtypes = [C.c_void_p, C.POINTER(DeviceInfo)]
    L.rt_default_params.restype = None
    L.rt_default_params.argtypes = [C.POINTER(RenderParams), C.c_uint32, C.c_uint32, C.c_uint32]
    L.rt_local_row_count.restype = C.c_uint32
    L.rt_local_row_count.argtypes = [C.POINTER(RenderParams)]
    L.rt_global_row.restype = C.c_uint32
    L.rt_global_row.argtypes = [C.POINTER(RenderParams), C.c_uint32]
    L.rt_upload_world.restype = C.c_int
    L.rt_upload_world.argtypes = [C.c_void_p, C.POINTER(World)]
    L.rt_reserve.restype = C.c_int
    L.rt_reserve.argtypes = [C.c_void_p, C.POINTER(RenderParams)]
    L.rt_load_accumulator.restype = C.c_int
    L.rt_load_accumulator.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p]
    L.rt_render.restype = C.c_int
    L.rt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.rt_render_device.restype = C.c_int
    L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_render_device_noalloc.restype = C.c_int
    L.rt_render_device_noalloc.argtypes = L.rt_render_device.argtypes
    L.rt_render_multi.restype = C.c_int
    L.rt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_uint64)]
    L.rt_combine_device.restype = C.c_int
    L.rt_combine_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.rt_resolve_device.restype = C.c_int
    L.rt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_resolve.restype = C.c_int
    L.rt_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    L.rt_last_kernel_ms.restype = C.c_int
    L.rt_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    if L.rt_abi_version() != 1:
        raise RuntimeError("librt_hip.so ABI version mismatch")
    _lib = L
    return L


def default_params(width: int, height: int, spp: int, **overrides) -> RenderParams:
    p = RenderParams()
    load_library().rt_default_params(C.byref(p), width, height, spp)
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"rt_render_params has no field {k!r}")
        setattr(p, k, v)
    return p


def local_row_count(params: RenderParams) -> int:
    return int(load_library().rt_local_row_count(C.byref(params)))


def global_rows(params: RenderParams) -> np.ndarray:
    L = load_library()
    return np.array([L.rt_global_row(C.byref(params), i) for i in range(local_row_count(params))], dtype=np.int64)


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None and arr.size else None


def render_multi(contexts, params: RenderParams, want_pixels=True, want_radiance=True):
    """rt_render_multi over a list of Context objects: full-size host outputs (pixels, radiance, rays)."""
    L = load_library()
    rows = params.row_end - params.row_begin
    pixels = np.zeros((rows, params.width), dtype=np.uint32) if want_pixels else None
    radiance = np.zeros((rows, params.width, 3), dtype=np.float32) if want_radiance else None
    rays = C.c_uint64(0)
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    st = L.rt_render_multi(handles, len(contexts), C.byref(params), _ptr(pixels), _ptr(radiance), C.byref(rays))
    if st != RT_OK:
        raise RtError(st, L.rt_error_string(contexts[0]._h).decode() or L.rt_status_string(st).decode())
    return pixels, radiance, int(rays.value)


class Context:
    """One GPU (rt_context)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        handle = C.c_void_p()
        st = self._lib.rt_create(device, C.byref(handle))
        if st != RT_OK:
            raise RtError(st, self._lib.rt_status_string(st).decode())
        self._h = handle
        self._scene = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int):
        if st != RT_OK:
            raise RtError(st, self._lib.rt_error_string(self._h).decode() or self._lib.rt_status_string(st).decode())

    def device_info(self) -> DeviceInfo:
        info = DeviceInfo()
        self._check(self._lib.rt_get_device_info(self._h, C.byref(info)))
        return info

    def upload_world(self, scene: Scene):
        w = World(len(scene.materials), _ptr(scene.materials), len(scene.planes), _ptr(scene.planes),
                  len(scene.spheres), _ptr(scene.spheres), len(scene.rectangles), _ptr(scene.rectangles),
                  _ptr(scene.camera))
        self._check(self._lib.rt_upload_world(self._h, C.byref(w)))
        self._scene = scene

    def reserve(self, params: RenderParams):
        """Allocates the device workspace a render with these parameters needs (FLAG_COMPACT_TAILS)."""
        self._check(self._lib.rt_reserve(self._h, C.byref(params)))

    def load_accumulator(self, params: RenderParams, radiance_sum: np.ndarray):
        """Sets the running sum that FLAG_ACCUMULATE renders add to: [local rows, W, 3] float32."""
        radiance_sum = np.ascontiguousarray(radiance_sum, dtype=np.float32)
        if radiance_sum.shape != (local_row_count(params), params.width, 3):
            raise ValueError("radiance_sum must be [local rows, width, 3]")
        self._check(self._lib.rt_load_accumulator(self._h, C.byref(params), _ptr(radiance_sum)))

    def render(self, params: RenderParams, want_pixels=True, want_radiance=True):
        """Synchronous render to host memory: (pixels [rows,W] uint32, radiance [rows,W,3] f32, rays)."""
        rows = local_row_count(params)
        pixels = np.zeros((rows, params.width), dtype=np.uint32) if want_pixels else None
        radiance = np.zeros((rows, params.width, 3), dtype=np.float32) if want_radiance else None
        rays = C.c_uint64(0)
        self._check(self._lib.rt_render(self._h, C.byref(params), _ptr(pixels), _ptr(radiance), C.byref(rays)))
        return pixels, radiance, int(rays.value)

    def render_device(self, params: RenderParams, d_pixels: int | None, d_radiance: int | None,
                      d_rays: int | None, stream: int | None = None):
        """Asynchronous render into caller-owned device buffers (raw device addresses)."""
        self._check(self._lib.rt_render_device(self._h, C.byref(params), d_pixels, d_radiance, d_rays, stream))

    def render_device_noalloc(self, params: RenderParams, d_pixels: int | None, d_radiance: int | None,
                              d_rays: int | None, stream: int | None = None):
        """render_device with the fixed per-GPU workspace: allocates nothing; refuses sizes the workspace cannot hold."""
        self._check(self._lib.rt_render_device_noalloc(self._h, C.byref(params), d_pixels, d_radiance, d_rays, stream))

    def combine_device(self, d_sums: int, part_count: int, part_stride: int, value_count: int, d_out: int,
                       stream: int | None = None):
        """Fixed-order (pairwise tree) sum of part_count radiance sums, part k at d_sums + 4 * k * part_stride."""
        self._check(self._lib.rt_combine_device(self._h, d_sums, part_count, part_stride, value_count, d_out, stream))

    def resolve_device(self, d_sum: int, pixel_count: int, total_samples: int, d_pixels: int | None,
                       d_mean: int | None, stream: int | None = None):
        self._check(self._lib.rt_resolve_device(self._h, d_sum, pixel_count, total_samples, d_pixels, d_mean, stream))

    def resolve(self, radiance_sum: np.ndarray, total_samples: int, want_pixels=True, want_mean=True):
        """Tonemap + pack of a host-resident sum [rows,W,3] f32: (pixels [rows,W] uint32, mean [rows,W,3] f32)."""
        radiance_sum = np.ascontiguousarray(radiance_sum, dtype=np.float32)
        rows, width, channels = radiance_sum.shape
        if channels != 3:
            raise ValueError("radiance_sum must be [rows, width, 3]")
        pixels = np.zeros((rows, width), dtype=np.uint32) if want_pixels else None
        mean = np.zeros((rows, width, 3), dtype=np.float32) if want_mean else None
        self._check(self._lib.rt_resolve(self._h, _ptr(radiance_sum), rows * width, total_samples, _ptr(pixels), _ptr(mean)))
        return pixels, mean

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        self._check(self._lib.rt_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)
