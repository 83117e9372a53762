"""Which rectangles the scene image marks as turned about one axis (rt_layout.hpp, rect_sparsity_class), read back
through the library's host-side hook - no GPU needed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from raytracer_amd import binding as B
from raytracer_amd import scene_io
from raytracer_amd.scene_io import read_scene
from test_gpu_cull import camera, materials, rectangle


def classes(scene, imode):
    lib = B.load_library()
    lib.rt_debug_rect_classes.restype = C.c_int
    lib.rt_debug_rect_classes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    world = B.World(len(scene.materials), ptr(scene.materials), len(scene.planes), ptr(scene.planes), len(scene.spheres),
                    ptr(scene.spheres), len(scene.rectangles), ptr(scene.rectangles), ptr(scene.camera))
    out = np.full(len(scene.rectangles), 255, dtype=np.uint8)
    assert lib.rt_debug_rect_classes(C.byref(world), imode, ptr(out)) == 0
    return out


def expected_class(m):
    """Straight from the definition: 1 = zeros of a turn about y, 2 = about x, else 0."""
    if m[0, 1] == 0 and m[1, 0] == 0 and m[1, 2] == 0 and m[2, 1] == 0:
        return 1
    if m[0, 1] == 0 and m[0, 2] == 0 and m[1, 0] == 0 and m[2, 0] == 0:
        return 2
    return 0


def test_cornell_box_classes(golden_dir):
    """Walls, floor, ceiling, light and the sides of the two boxes are turned about one axis (14 rectangles); the boxes'
    tops and bottoms are turned about two (scene.h:347-469) and take the full transform.  The same in every image."""
    scene = read_scene(golden_dir / "cornell.rtsc")
    want = np.array([expected_class(r["inverse_transform"]) for r in scene.rectangles], dtype=np.uint8)
    assert want.tolist() == [2, 2, 1, 1, 1, 2, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 0, 0]
    for imode in (B.INTERSECT_WIDE8, B.INTERSECT_SCALAR, B.INTERSECT_WIDE4):
        assert classes(scene, imode).tolist() == want.tolist()


def test_out_of_range_and_late_rectangles_are_left_alone():
    rng = np.random.default_rng(2)
    wall = np.eye(4); wall[:3, 3] = (0, 0, -2)
    rects = [rectangle(wall, 1) for _ in range(70)]
    huge = rectangle(wall, 1); huge["inverse_transform"][0, 0, 3] = 3.0e13          # a translation beyond 2^40
    tiny = rectangle(wall, 1); tiny["inverse_transform"][0, 1, 1] = 1.0e-13         # a row shorter than 2^-40
    rects[3], rects[5] = huge, tiny
    scene = scene_io.Scene(materials(rng, 3), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), np.concatenate(rects), camera((0.0, 1.0, 10.0)))
    got = classes(scene, B.INTERSECT_SCALAR)                                          # slots in the caller's order
    assert got[3] == 0 and got[5] == 0
    assert (got[:3] == 1).all() and got[4] == 1 and (got[6:64] == 1).all()
    assert (got[64:] == 0).all()
