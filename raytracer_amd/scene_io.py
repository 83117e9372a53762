"""Scene records and the RTSC scene file.

The record layouts are byte-for-byte those of the reference's AoS scene types
(scene.h:13-18 Material, :20-24 Sphere, :32-36 Plane, :43-47 RectangleXY,
:163-168 Camera); they are what the C-ABI in include/rt_hip.h takes.

RTSC v1 file: 24-byte header (magic "RTSC", version, four counts), the camera,
then the material, plane, sphere and rectangle arrays.
"""
from __future__ import annotations

import dataclasses
import struct

import numpy as np

MATERIAL = np.dtype([("color", "<f4", 3), ("refractive_index", "<f4"),
                     ("emit_color", "<f4", 3), ("reflection", "<f4")])
SPHERE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("material_index", "<u4")])
PLANE = np.dtype([("normal", "<f4", 3), ("d", "<f4"), ("material_index", "<u4")])
RECTANGLE = np.dtype([("inverse_transform", "<f4", (4, 4)), ("normal", "<f4", 3),
                      ("material_index", "<u4")])
CAMERA = np.dtype([("position", "<f4", 3), ("z_axis", "<f4", 3), ("y_axis", "<f4", 3),
                   ("x_axis", "<f4", 3)])

assert MATERIAL.itemsize == 32 and SPHERE.itemsize == 20 and PLANE.itemsize == 20
assert RECTANGLE.itemsize == 80 and CAMERA.itemsize == 48

_HEADER = struct.Struct("<4sIIIII")


@dataclasses.dataclass
class Scene:
    """Host-side world: the arrays `World` points to (scene.h:177-191), minus the lane mirrors."""
    materials: np.ndarray
    planes: np.ndarray
    spheres: np.ndarray
    rectangles: np.ndarray
    camera: np.ndarray  # shape (1,) of CAMERA

    def __post_init__(self):
        self.materials = np.ascontiguousarray(self.materials, dtype=MATERIAL)
        self.planes = np.ascontiguousarray(self.planes, dtype=PLANE)
        self.spheres = np.ascontiguousarray(self.spheres, dtype=SPHERE)
        self.rectangles = np.ascontiguousarray(self.rectangles, dtype=RECTANGLE)
        self.camera = np.ascontiguousarray(self.camera, dtype=CAMERA).reshape(1)
        if len(self.materials) == 0:
            raise ValueError("a scene needs at least the sky material (index 0)")
        for name in ("planes", "spheres", "rectangles"):
            arr = getattr(self, name)
            if len(arr) and int(arr["material_index"].max()) >= len(self.materials):
                raise ValueError(f"{name}: material index out of range")


def read_scene(path) -> Scene:
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < _HEADER.size:
        raise ValueError(f"{path}: truncated RTSC header")
    magic, version, nm, npl, ns, nr = _HEADER.unpack_from(data, 0)
    if magic != b"RTSC" or version != 1:
        raise ValueError(f"{path}: not an RTSC v1 scene file")
    off = _HEADER.size
    expect = off + CAMERA.itemsize + nm * 32 + npl * 20 + ns * 20 + nr * 80
    if len(data) != expect:
        raise ValueError(f"{path}: size {len(data)} does not match header ({expect})")

    def take(dtype, count):
        nonlocal off
        arr = np.frombuffer(data, dtype=dtype, count=count, offset=off).copy()
        off += dtype.itemsize * count
        return arr

    camera = take(CAMERA, 1)
    return Scene(take(MATERIAL, nm), take(PLANE, npl), take(SPHERE, ns), take(RECTANGLE, nr), camera)


def write_scene(path, scene: Scene) -> None:
    with open(path, "wb") as f:
        f.write(_HEADER.pack(b"RTSC", 1, len(scene.materials), len(scene.planes),
                             len(scene.spheres), len(scene.rectangles)))
        f.write(scene.camera.tobytes())
        f.write(scene.materials.tobytes())
        f.write(scene.planes.tobytes())
        f.write(scene.spheres.tobytes())
        f.write(scene.rectangles.tobytes())
