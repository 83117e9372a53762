"""The primary-ray cull's decisions, checked on the CPU without a GPU: the library's host-side hook evaluates, with
the very table and functions the render kernel uses (rt_layout.hpp: cull_record, cull_box, cull_keeps), which
rectangles a wave with a given pixel tile keeps for its primary rays.  Every primary ray the reference can generate
for a pixel of the tile (main.cpp:375-388: pixel centre, jitter of a quarter pixel) that hits a rectangle
(main.cpp:96-129) must find that rectangle kept - including rays through the corners of the jitter range and
rectangles inflated by a tenth of the cull's own margin."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from raytracer_amd import binding as B
from raytracer_amd import scene_io
from raytracer_amd.scene_io import read_scene
from test_gpu_cull import camera, materials, random_rectangles, rectangle

f32 = np.float32


def reachable(scene, width, height, imode, x0, x1, row0, row1):
    lib = B.load_library()
    lib.rt_debug_cull_reachable.restype = C.c_int
    lib.rt_debug_cull_reachable.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_void_p]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    world = B.World(len(scene.materials), ptr(scene.materials), len(scene.planes), ptr(scene.planes), len(scene.spheres),
                    ptr(scene.spheres), len(scene.rectangles), ptr(scene.rectangles), ptr(scene.camera))
    out = np.zeros(len(scene.rectangles), dtype=np.uint8)
    status = lib.rt_debug_cull_reachable(C.byref(world), width, height, imode, x0, x1, row0, row1, ptr(out))
    assert status == 0
    return out.astype(bool)


def primary_directions(cam, width, height, xs, rows, jx, jy):
    """Normalised primary directions as the render forms them (main.cpp:350-388), operation by operation in float32."""
    pos, zax, yax, xax = (cam[k][0].astype(f32) for k in ("position", "z_axis", "y_axis", "x_axis"))
    aspect = f32(width) / f32(height)
    half_w, half_h = f32(1.0) * aspect * f32(0.5), f32(0.5)
    pixel_w, pixel_h = f32(0.5) / f32(width), f32(0.5) / f32(height)
    film_center = pos - zax * f32(1.0)
    film_x = (xs.astype(f32) / f32(width)) * f32(2.0) - f32(1.0)
    film_y = (rows.astype(f32) / f32(height)) * f32(-2.0) + f32(1.0)
    off_x = film_x + jx.astype(f32) * pixel_w
    off_y = film_y + jy.astype(f32) * pixel_h
    film_pos = (film_center[None, :] + (xax[None, :] * off_x[:, None]) * half_w) + (yax[None, :] * half_h) * off_y[:, None]
    d = film_pos - pos[None, :]
    length2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    return d * (f32(1.0) / np.sqrt(length2))[:, None]


def hits(scene, directions, inflate):
    """[ray, rectangle]: does the ray from the camera pass the rectangle test (in double, square inflated)?"""
    o = np.append(scene.camera["position"][0].astype(np.float64), 1.0)
    m = scene.rectangles["inverse_transform"].astype(np.float64)          # [R, 4, 4]
    op = m[:, :3, :] @ o                                                   # [R, 3]
    dp = np.einsum("rij,nj->nri", m[:, :3, :3], directions.astype(np.float64))   # [N, R, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -op[None, :, 2] / dp[:, :, 2]
        x = op[None, :, 0] + dp[:, :, 0] * t
        y = op[None, :, 1] + dp[:, :, 1] * t
    return (t > 0.0) & (np.abs(x) <= 1.0 + inflate) & (np.abs(y) <= 1.0 + inflate)


def check_tiles(scene, width, height, tile_w, tile_h, tiles, rng, imode=B.INTERSECT_WIDE8, samples=300):
    culled = total = 0
    for _ in range(tiles):
        x0 = int(rng.integers(0, (width + tile_w - 1) // tile_w)) * tile_w
        r0 = int(rng.integers(0, (height + tile_h - 1) // tile_h)) * tile_h
        x1, r1 = min(x0 + tile_w, width) - 1, min(r0 + tile_h, height) - 1
        keep = reachable(scene, width, height, imode, x0, x1, r0, r1)
        xs = rng.integers(x0, x1 + 1, samples)
        rows = rng.integers(r0, r1 + 1, samples)
        # jitter: uniform in [-1, 1], a fifth of the samples on the corners and edges of the range
        jx, jy = rng.uniform(-1, 1, samples), rng.uniform(-1, 1, samples)
        edge = rng.random(samples) < 0.2
        jx[edge] = rng.choice([-1.0, 1.0], int(edge.sum()))
        edge = rng.random(samples) < 0.2
        jy[edge] = rng.choice([-1.0, 1.0], int(edge.sum()))
        d = primary_directions(scene.camera, width, height, xs, rows, jx, jy)
        hit_any = hits(scene, d, inflate=1.0e-4).any(axis=0)              # [R]
        missed = hit_any & ~keep
        assert not missed.any(), (f"tile [{x0},{x1}]x[{r0},{r1}] of {width}x{height}: rectangles {np.flatnonzero(missed).tolist()} "
                                  f"are hit by primary rays of the tile but were culled")
        culled += int((~keep).sum())
        total += len(keep)
    return culled / max(total, 1)


@pytest.mark.parametrize("seed", range(10))
def test_no_reachable_rectangle_is_culled(seed):
    rng = np.random.default_rng(900 + seed)
    nm = 4
    kind = seed % 5
    scales = [(0.05, 5.0), (1e-3, 50.0), (0.5, 2.0), (1e-2, 1e2), (0.2, 8.0)][kind]
    rects = random_rectangles(rng, int(rng.integers(2, 50)), nm, spread=float(rng.choice([2.0, 7.0, 30.0])), scales=scales)
    pos = rng.normal(size=3) * float(rng.choice([3.0, 12.0, 60.0]))
    target = rng.normal(size=3) * 2.0
    if kind == 1:       # camera almost in the plane of the first rectangle
        m = np.linalg.inv(rects["inverse_transform"][0].astype(np.float64))
        pos = m[:3, 3] + m[:3, 0] * rng.uniform(-3, 3) + m[:3, 1] * rng.uniform(-3, 3) + m[:3, 2] * float(rng.choice([0.0, 1e-7, 1e-4]))
        target = m[:3, 3] + m[:3, 0] * rng.uniform(-1, 1)
    if kind == 3:
        pos = pos * 300.0
    scene = scene_io.Scene(materials(rng, nm), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rects, camera(pos, target))
    for tile_w, tile_h in ((8, 8), (4, 4), (8, 4), (2, 2), (1, 1)):
        width, height = int(rng.integers(16, 2000)), int(rng.integers(16, 1200))
        check_tiles(scene, width, height, tile_w, tile_h, tiles=12, rng=rng, imode=int(rng.integers(0, 3)))


def test_cornell_box_tiles(golden_dir):
    """The benchmarked case: 1920x1080, tiles of 4x4 pixels.  No hit rectangle is culled; about half of the tiles
    (the black surround of the box: 55.8 % of the pixels) keep no rectangle at all, and a tile keeps few of the 18."""
    scene = read_scene(golden_dir / "cornell.rtsc")
    rng = np.random.default_rng(1)
    fraction_culled = check_tiles(scene, 1920, 1080, 4, 4, tiles=300, rng=rng)
    assert fraction_culled > 0.8
    empty = 0
    for _ in range(400):
        x0, r0 = int(rng.integers(0, 480)) * 4, int(rng.integers(0, 270)) * 4
        empty += not reachable(scene, 1920, 1080, B.INTERSECT_WIDE8, x0, x0 + 3, r0, r0 + 3).any()
    assert 0.45 <= empty / 400 <= 0.62


def test_skewed_axes_switch_the_cull_off():
    rng = np.random.default_rng(3)
    wall = np.eye(4); wall[:3, 3] = (0, 0, -2)
    cam = camera((0.0, 1.0, 10.0))
    cam["x_axis"] = cam["x_axis"] * 1.2
    scene = scene_io.Scene(materials(rng, 3), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rectangle(wall, 1), cam)
    assert reachable(scene, 640, 360, B.INTERSECT_WIDE8, 0, 7, 0, 7).all()          # far corner: kept all the same
    scene.camera = camera((0.0, 1.0, 10.0))
    assert not reachable(scene, 640, 360, B.INTERSECT_WIDE8, 0, 7, 0, 7).any()
