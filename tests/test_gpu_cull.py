"""The per-wave primary-ray cull (rt_kernels.hpp, primary_pair_mask): a wave skips, while all of its rays are
primary rays, the rectangles that no primary ray of its pixel tile can reach.  The cull must be invisible:
pixels, linear radiance and the ray count of a render with it equal, bit for bit, those of a render with
RT_FLAG_NO_PRIMARY_CULL - and those of the CPU oracle, which tests every rectangle for every ray
(main.cpp:96-129)."""
from __future__ import annotations

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import binding as B
from raytracer_amd import scene_io
from test_gpu_parity import bits, check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def camera(pos, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """Camera(scene.h:169-174) generalised to a look-at target: z points from the target to the camera."""
    pos = np.asarray(pos, dtype=np.float64)
    z = pos - np.asarray(target, dtype=np.float64)
    z /= np.linalg.norm(z)
    x = np.cross(up, z); x /= np.linalg.norm(x)
    y = np.cross(z, x); y /= np.linalg.norm(y)
    cam = np.zeros(1, dtype=scene_io.CAMERA)
    cam["position"], cam["z_axis"], cam["y_axis"], cam["x_axis"] = pos, z, y, x
    return cam


def materials(rng, count):
    mats = np.zeros(count, dtype=scene_io.MATERIAL)
    mats["color"] = rng.uniform(0.1, 0.9, (count, 3))
    mats["reflection"] = rng.choice([0.0, 0.0, 0.5, 1.0], count)
    mats["emit_color"] = rng.uniform(0.0, 3.0, (count, 3)) * (rng.random((count, 1)) < 0.5)
    mats["emit_color"][0] = rng.uniform(0.05, 0.5, 3)        # a sky that is not black: a wrongly culled hit changes the pixel
    return mats


def rectangle(transform, material):
    """A unit square under `transform` (4x4), stored inverted as the reference does (scene.h:408-411)."""
    r = np.zeros(1, dtype=scene_io.RECTANGLE)
    r["inverse_transform"] = np.linalg.inv(transform).astype(np.float32)
    n = transform[:3, 2] / np.linalg.norm(transform[:3, 2])
    r["normal"] = n.astype(np.float32)
    r["material_index"] = material
    return r


def random_rectangles(rng, count, n_materials, spread=7.0, scales=(0.05, 5.0)):
    out = []
    for _ in range(count):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m = np.eye(4)
        m[:3, :3] = q * np.exp(rng.uniform(np.log(scales[0]), np.log(scales[1]), 3))
        m[:3, 3] = rng.uniform(-spread, spread, 3)
        out.append(rectangle(m, int(rng.integers(1, n_materials))))
    return np.concatenate(out) if out else np.zeros(0, dtype=scene_io.RECTANGLE)


def render_both_ways(ctx, p):
    """Renders with and without the cull and checks that nothing differs; returns the culled render."""
    culled = ctx.render(p)
    q = B.RenderParams.from_buffer_copy(p)
    q.flags |= B.FLAG_NO_PRIMARY_CULL
    plain = ctx.render(q)
    assert culled[2] == plain[2], f"ray count {culled[2]} with the cull, {plain[2]} without"
    differing = np.argwhere(bits(culled[1]) != bits(plain[1]))
    assert len(differing) == 0, f"radiance differs at {len(differing)} values, first at {differing[:4].tolist()}"
    assert np.array_equal(culled[0], plain[0])
    return culled


@pytest.mark.parametrize("seed", range(16))
def test_random_rectangle_worlds(ctx, seed):
    """Random worlds of 1..64 rectangles of every orientation and size (slivers, rectangles behind, around and
    edge-on to the camera), with and without spheres and planes, every intersect routine, 1..16 lanes per pixel
    (tiles from 8x8 to 2x2 pixels), ragged sizes and row bands."""
    rng = np.random.default_rng(4200 + seed)
    nm = int(rng.integers(2, 7))
    rects = random_rectangles(rng, int(rng.integers(1, 65)), nm)
    spheres = np.zeros(int(rng.integers(0, 4)) if seed % 3 == 0 else 0, dtype=scene_io.SPHERE)
    if len(spheres):
        spheres["position"] = rng.uniform(-5, 5, (len(spheres), 3))
        spheres["radius"] = rng.uniform(0.3, 1.5, len(spheres))
        spheres["material_index"] = rng.integers(1, nm, len(spheres))
    planes = np.zeros(1 if seed % 4 == 1 else 0, dtype=scene_io.PLANE)
    if len(planes):
        planes["normal"], planes["d"], planes["material_index"] = (0.0, 1.0, 0.0), 6.0, 1
    pos = rng.uniform(-1, 1, 3) * np.array([6, 4, 4]) + np.array([0, 1, 12])
    scene = scene_io.Scene(materials(rng, nm), planes, spheres, rects, camera(pos))
    ctx.upload_world(scene)
    for _ in range(4):
        lanes = int(rng.choice([1, 2, 4, 8, 16]))
        w, h = int(rng.integers(30, 420)), int(rng.integers(20, 240))
        kw = dict(lanes_per_pixel=lanes, intersect_mode=int(rng.integers(0, 3)), max_bounces=int(rng.choice([1, 3, 8])))
        if rng.random() < 0.3:
            kw.update(band_rows=int(rng.choice([1, 3, 8])), band_count=3, band_index=int(rng.integers(0, 3)))
        if rng.random() < 0.5:
            kw["flags"] = B.FLAG_COMPACT_TAILS
        spp = lanes * int(rng.choice([1, 2, 8]))
        render_both_ways(ctx, B.default_params(w, h, spp, **kw))
    # and against the oracle, which knows nothing of tiles
    check_against_oracle(ctx, scene, B.default_params(int(rng.integers(40, 100)), int(rng.integers(20, 60)), 8,
                                                      lanes_per_pixel=int(rng.choice([1, 2, 4]))))


def test_rectangle_edges_on_tile_boundaries(ctx):
    """A checkerboard whose squares' edges project exactly onto tile boundaries (and a hair inside and outside
    them), seen head-on: for every tile some rectangle ends precisely where the tile's own pixels end, so a cull
    that trusted the nominal tile bounds without their jitter and rounding margins would drop hits."""
    W, H, d = 256, 144, 10.0
    rng = np.random.default_rng(99)
    nm = 6
    rects = []
    for tile in (8, 4, 2):                                   # tile sizes of 1, 4 and 16 lanes per pixel
        for iy in range(0, H // tile, 5):
            for ix in range((iy // 5) % 2, W // tile, 7):
                # film-plane bounds of the tile (film at distance 1, width = aspect, height 1: main.cpp:358-377)
                a0, a1 = ((ix * tile) / W * 2 - 1) * (W / H) / 2, (((ix + 1) * tile) / W * 2 - 1) * (W / H) / 2
                b1, b0 = ((iy * tile) / H * -2 + 1) / 2, (((iy + 1) * tile) / H * -2 + 1) / 2
                grow = float(rng.choice([0.0, 1e-6, -1e-6, 1e-3, -1e-3]))
                m = np.eye(4)
                m[0, 0], m[1, 1] = (a1 - a0) / 2 * d * (1 + grow), (b1 - b0) / 2 * d * (1 + grow)
                m[0, 3], m[1, 3] = (a0 + a1) / 2 * d, (b0 + b1) / 2 * d
                rects.append(rectangle(m, int(rng.integers(1, nm))))
    rects = np.concatenate(rects)[:126]                      # 63 pairs: every pair has its bit in the wave's mask
    scene = scene_io.Scene(materials(rng, nm), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rects,
                           camera((0.0, 0.0, d)))
    ctx.upload_world(scene)
    for lanes in (1, 4, 16):
        render_both_ways(ctx, B.default_params(W, H, 16 * lanes, lanes_per_pixel=lanes, max_bounces=2))
    check_against_oracle(ctx, scene, B.default_params(W, H, 8, lanes_per_pixel=4, max_bounces=2))


def test_more_pairs_than_mask_bits(ctx):
    """More than 64 rectangle pairs: the pairs beyond the mask are always visited."""
    rng = np.random.default_rng(5)
    rects = random_rectangles(rng, 150, 4, spread=9.0, scales=(0.2, 1.5))
    scene = scene_io.Scene(materials(rng, 4), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rects, camera((1.0, 2.0, 15.0)))
    ctx.upload_world(scene)
    render_both_ways(ctx, B.default_params(160, 90, 8, lanes_per_pixel=2))
    check_against_oracle(ctx, scene, B.default_params(64, 36, 4))


def test_awkward_cameras_and_rectangles(ctx):
    """Cases in which the cull must stand back or take care: the camera in a rectangle's plane (o'z = 0) and
    nearly in it, a rectangle seen exactly edge-on, a camera far from the origin (the rounding margin of the ray
    set-up grows with the position's magnitude), a camera inside a closed box, axes that are not orthonormal
    (the cull is off), tiny and huge rectangles, a degenerate (rank 2) transform."""
    rng = np.random.default_rng(11)
    nm = 5
    mats = materials(rng, nm)
    none_p, none_s = np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE)

    def world(rects, cam):
        return scene_io.Scene(mats, none_p, none_s, np.concatenate(rects), cam)

    floor = np.eye(4); floor[:3, :3] = np.array([[4, 0, 0], [0, 0, 4], [0, -4, 0]], float)          # square in the plane y = 0
    wall = np.eye(4); wall[:3, :3] = np.diag([3.0, 3.0, 1.0]); wall[:3, 3] = (0, 0, -2)
    sliver = np.eye(4); sliver[:3, :3] = np.diag([1e-3, 5.0, 1.0]); sliver[:3, 3] = (1, 0, 1)
    huge = np.eye(4); huge[:3, :3] = np.diag([4e3, 4e3, 1.0]); huge[:3, 3] = (0, 0, -30)
    edge_on = np.eye(4); edge_on[:3, :3] = np.array([[0, 0, 1], [0, 2, 0], [-2, 0, 0]], float)      # in the plane x = 0
    flat = np.eye(4); flat[:3, :3] = np.diag([2.0, 2.0, 1.0]); flat[:3, 3] = (2, 1, 0)
    rank2 = rectangle(flat, 2); rank2["inverse_transform"][0, 2, :] = 0                             # third row zero: never hit
    cases = [
        (world([rectangle(floor, 1), rectangle(wall, 2)], camera((0.0, 0.0, 9.0))), "camera in the floor's plane"),
        (world([rectangle(floor, 1), rectangle(wall, 2)], camera((0.0, 1e-7, 9.0))), "camera a hair above it"),
        (world([rectangle(edge_on, 1), rectangle(wall, 3), rectangle(sliver, 2)], camera((0.0, 0.5, 8.0))), "edge-on rectangle, sliver"),
        (world([rectangle(huge, 1), rectangle(wall, 2), rank2], camera((0.3, 0.2, 12.0))), "huge rectangle, rank-2 transform"),
        (world([rectangle(wall, 1), rectangle(floor, 2)], camera((3.0e4, 1.0e4, 5.0e4), target=(3.0e4, 1.0e4, 0.0))), "camera far from the origin"),
    ]
    box = []
    for axis in range(3):
        for side in (-1.0, 1.0):
            m = np.eye(4)
            e = np.eye(3)
            m[:3, 0], m[:3, 1], m[:3, 2] = 3.0 * e[(axis + 1) % 3], 3.0 * e[(axis + 2) % 3], e[axis]
            m[:3, 3] = 3.0 * side * e[axis]
            box.append(rectangle(m, 1 + (axis * 2 + int(side > 0)) % (nm - 1)))
    cases.append((world(box, camera((0.2, 0.1, 1.5))), "camera inside a closed box"))
    skewed = camera((0.0, 1.0, 10.0))
    skewed["x_axis"] = skewed["x_axis"] * 1.2 + skewed["y_axis"] * 0.1
    cases.append((world([rectangle(wall, 1), rectangle(floor, 2)], skewed), "axes not orthonormal"))
    for scene, what in cases:
        ctx.upload_world(scene)
        for lanes, size in ((1, (97, 61)), (4, (203, 114))):
            try:
                render_both_ways(ctx, B.default_params(size[0], size[1], 8 * lanes, lanes_per_pixel=lanes, max_bounces=4))
            except AssertionError as e:
                raise AssertionError(f"{what}: {e}") from None
        check_against_oracle(ctx, scene, B.default_params(64, 36, 4, lanes_per_pixel=2, max_bounces=4))


def test_cornell_every_tile_size(ctx, cornell):
    """The Cornell box (the benchmarked scene) at the benchmarked tile size and the others, compacted or not."""
    ctx.upload_world(cornell)
    for lanes in (1, 2, 4, 8, 16, 32, 64):
        render_both_ways(ctx, B.default_params(480, 270, 8 * lanes, lanes_per_pixel=lanes))
    render_both_ways(ctx, B.default_params(960, 540, 64, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS))
