"""ctypes binding of oracle/librt_oracle.so - TEST INFRASTRUCTURE.

Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
LIB_PATH = ROOT / "oracle" / "librt_oracle.so"

WIDE8, SCALAR, WIDE8_AVX2, WIDE8_IEEE, WIDE4, WIDE4_IEEE, WIDE8_LANES = 0, 1, 2, 3, 4, 5, 6
ROW_SERIAL, PIXEL = 0, 1


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
        ("band_rows", C.c_uint32), ("band_index", C.c_uint32), ("band_count", C.c_uint32),
        ("max_bounces", C.c_uint32),
        ("min_hit_distance", C.c_float), ("plane_tolerance", C.c_float),
        ("seed_constant", C.c_uint32), ("rng_mode", C.c_uint32),
        ("rows_per_order", C.c_uint32), ("lanes_per_pixel", C.c_uint32), ("batch_offset", C.c_uint32),
        ("intersect_mode", C.c_uint32), ("output_sum", C.c_uint32),
    ]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "oracle"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        L = C.CDLL(str(LIB_PATH))
        L.rto_scene_create.restype = C.c_void_p
        L.rto_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_uint32, C.c_void_p]
        L.rto_scene_destroy.argtypes = [C.c_void_p]
        L.rto_default_params.argtypes = [C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_uint32]
        L.rto_local_row_count.restype = C.c_uint32
        L.rto_local_row_count.argtypes = [C.POINTER(Params)]
        L.rto_global_row.restype = C.c_uint32
        L.rto_global_row.argtypes = [C.POINTER(Params), C.c_uint32]
        L.rto_xorshift32.restype = C.c_uint32
        L.rto_xorshift32.argtypes = [C.POINTER(C.c_uint32)]
        L.rto_unilateral.restype = C.c_float
        L.rto_unilateral.argtypes = [C.POINTER(C.c_uint32)]
        L.rto_bilateral.restype = C.c_float
        L.rto_bilateral.argtypes = [C.POINTER(C.c_uint32)]
        L.rto_refract.restype = C.c_int
        L.rto_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.rto_schlick.restype = C.c_float
        L.rto_schlick.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.rto_linear_to_srgb.restype = C.c_float
        L.rto_linear_to_srgb.argtypes = [C.c_float]
        L.rto_pack_srgb.restype = C.c_uint32
        L.rto_pack_srgb.argtypes = [C.c_void_p]
        L.rto_pack_srgb_many.restype = None
        L.rto_pack_srgb_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.rto_pixel_stream_seed.restype = C.c_uint32
        L.rto_pixel_stream_seed.argtypes = [C.c_uint32] * 4
        L.rto_intersect.restype = C.c_int
        L.rto_intersect.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p]
        L.rto_raytrace_world.restype = C.c_uint32
        L.rto_raytrace_world.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_uint32), C.c_void_p]
        L.rto_render.restype = C.c_int
        L.rto_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def default_params(width, height, spp, **overrides) -> Params:
    p = Params()
    lib().rto_default_params(C.byref(p), width, height, spp)
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None and arr.size else None


def pack_srgb(radiance: np.ndarray) -> np.ndarray:
    """The reference's tonemap + pack (math_util.h:102-125, glibc powf) of a [..., 3] float32 image."""
    radiance = np.ascontiguousarray(radiance, dtype=np.float32)
    out = np.zeros(radiance.shape[:-1], dtype=np.uint32)
    lib().rto_pack_srgb_many(_ptr(radiance), out.size, _ptr(out))
    return out


class OracleScene:
    def __init__(self, scene):
        self.scene = scene
        self.handle = lib().rto_scene_create(_ptr(scene.materials), len(scene.materials),
                                             _ptr(scene.planes), len(scene.planes),
                                             _ptr(scene.spheres), len(scene.spheres),
                                             _ptr(scene.rectangles), len(scene.rectangles),
                                             _ptr(scene.camera))
        if not self.handle:
            raise RuntimeError("rto_scene_create failed")

    def __del__(self):
        if getattr(self, "handle", None):
            lib().rto_scene_destroy(self.handle)
            self.handle = None

    def render(self, params: Params, threads: int | None = None):
        """Returns (pixels uint32 [rows, W], radiance float32 [rows, W, 3], rays int)."""
        L = lib()
        rows = L.rto_local_row_count(C.byref(params))
        pixels = np.zeros((rows, params.width), dtype=np.uint32)
        radiance = np.zeros((rows, params.width, 3), dtype=np.float32)
        rays = C.c_uint64(0)
        if threads is None:
            threads = min(16, os.cpu_count() or 1)
        rc = L.rto_render(self.handle, C.byref(params), threads, _ptr(pixels), _ptr(radiance), C.byref(rays))
        if rc != 0:
            raise ValueError(f"rto_render rejected the parameters (rc={rc})")
        return pixels, radiance, int(rays.value)

    def intersect(self, mode, origin, direction, min_hit=0.001, plane_tol=0.001):
        o = np.ascontiguousarray(origin, dtype=np.float32)
        d = np.ascontiguousarray(direction, dtype=np.float32)
        n = np.zeros(3, dtype=np.float32)
        t = C.c_float(0)
        m = C.c_uint32(0)
        hit = lib().rto_intersect(self.handle, mode, min_hit, plane_tol, _ptr(o), _ptr(d), C.byref(t), C.byref(m), _ptr(n))
        return bool(hit), float(np.float32(t.value)), int(m.value), n

    def raytrace_world(self, params: Params, origin, direction, state: int):
        o = np.ascontiguousarray(origin, dtype=np.float32)
        d = np.ascontiguousarray(direction, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        st = C.c_uint32(state)
        bounces = lib().rto_raytrace_world(self.handle, C.byref(params), _ptr(o), _ptr(d), C.byref(st), _ptr(out))
        return out, int(st.value), int(bounces)
