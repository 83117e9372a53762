"""Rectangles whose inverse transform holds the exact zeros of a rotation about the x or the y axis are tested
without the multiply-adds of those entries (rt_layout.hpp, rect_sparsity_class; rt_kernels.hpp, rectangle loop).
That must be invisible: pixels, linear radiance and the ray count equal, bit for bit, those of a render with
RT_FLAG_DENSE_TRANSFORMS (every rectangle through the full 3x4 transform, main.cpp:100-113) and those of the
CPU oracle, which knows no such shortcut."""
from __future__ import annotations

import numpy as np
import pytest

from raytracer_amd import binding as B
from raytracer_amd import scene_io
from test_gpu_cull import camera, materials, random_rectangles, rectangle
from test_gpu_parity import bits, check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def turned_rectangles(rng, count, n_materials, spread=6.0):
    """Unit squares scaled, turned about ONE coordinate axis (any angle, and exact quarter turns) and moved: the
    matrices a scene builder produces (scene.h:347-469), with exact zeros where the axis is left alone."""
    out = []
    for _ in range(count):
        axis = int(rng.integers(0, 3))
        kind = int(rng.integers(0, 3))
        if kind == 0:                                    # exact quarter turns: 0, +-1 entries
            c, s = [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)][int(rng.integers(0, 4))]
        else:
            angle = rng.uniform(-np.pi, np.pi)
            c, s = np.cos(angle), np.sin(angle)
        rot = np.eye(3)
        i, j = [(1, 2), (0, 2), (0, 1)][axis]
        rot[i, i], rot[i, j], rot[j, i], rot[j, j] = c, -s, s, c
        m = np.eye(4)
        m[:3, :3] = rot @ np.diag(np.exp(rng.uniform(np.log(0.2), np.log(4.0), 3)))
        m[:3, 3] = rng.uniform(-spread, spread, 3) * (rng.random(3) < 0.8)        # some coordinates exactly zero
        out.append(rectangle(m, int(rng.integers(1, n_materials))))
    return np.concatenate(out)


def render_sparse_and_dense(ctx, p):
    sparse = ctx.render(p)
    q = B.RenderParams.from_buffer_copy(p)
    q.flags |= B.FLAG_DENSE_TRANSFORMS
    dense = ctx.render(q)
    assert sparse[2] == dense[2], f"ray count {sparse[2]} with the shortened transforms, {dense[2]} with the full ones"
    differing = np.argwhere(bits(sparse[1]) != bits(dense[1]))
    assert len(differing) == 0, f"radiance differs at {len(differing)} values, first at {differing[:4].tolist()}"
    assert np.array_equal(sparse[0], dense[0])
    return sparse


@pytest.mark.parametrize("seed", range(12))
def test_turned_rectangles(ctx, seed):
    """Worlds of rectangles turned about one axis mixed with rectangles in general position, with and without spheres
    and a plane, every intersect routine; cameras on coordinate planes and axes, so that ray coordinates and
    partial sums are exact zeros of either sign."""
    rng = np.random.default_rng(7300 + seed)
    nm = int(rng.integers(3, 7))
    rects = np.concatenate([turned_rectangles(rng, int(rng.integers(2, 50)), nm), random_rectangles(rng, int(rng.integers(0, 12)), nm)])
    rects = rects[rng.permutation(len(rects))]
    spheres = np.zeros(int(rng.integers(1, 4)) if seed % 3 == 0 else 0, dtype=scene_io.SPHERE)
    if len(spheres):
        spheres["position"] = rng.uniform(-4, 4, (len(spheres), 3))
        spheres["radius"] = rng.uniform(0.3, 1.2, len(spheres))
        spheres["material_index"] = rng.integers(1, nm, len(spheres))
    planes = np.zeros(1 if seed % 4 == 1 else 0, dtype=scene_io.PLANE)
    if len(planes):
        planes["normal"], planes["d"], planes["material_index"] = (0.0, 1.0, 0.0), 7.0, 1
    pos = [(0.0, 0.0, 14.0), (0.0, 1.0, 12.0), (9.0, 0.0, 0.0), (3.0, 2.0, 11.0)][seed % 4]
    scene = scene_io.Scene(materials(rng, nm), planes, spheres, rects, camera(pos))
    ctx.upload_world(scene)
    for imode in (B.INTERSECT_WIDE8, B.INTERSECT_SCALAR, B.INTERSECT_WIDE4):
        lanes = int(rng.choice([1, 2, 4, 8]))
        kw = dict(lanes_per_pixel=lanes, intersect_mode=imode, max_bounces=int(rng.choice([2, 8])))
        if rng.random() < 0.5:
            kw["flags"] = B.FLAG_COMPACT_TAILS
        render_sparse_and_dense(ctx, B.default_params(int(rng.integers(60, 300)), int(rng.integers(40, 170)), lanes * int(rng.choice([2, 8])), **kw))
    render_sparse_and_dense(ctx, B.default_params(48, 20, 4, rng_mode=B.RNG_ROW_SERIAL, lanes_per_pixel=1))
    check_against_oracle(ctx, scene, B.default_params(int(rng.integers(40, 100)), int(rng.integers(20, 60)), 8,
                                                      lanes_per_pixel=int(rng.choice([1, 2, 4])), intersect_mode=int(rng.integers(0, 3))))


def test_more_rectangles_than_class_bits(ctx):
    """Rectangles beyond the first 64 slots use the full transform; those below keep their class."""
    rng = np.random.default_rng(17)
    rects = turned_rectangles(rng, 150, 4, spread=9.0)
    scene = scene_io.Scene(materials(rng, 4), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rects, camera((1.0, 2.0, 15.0)))
    ctx.upload_world(scene)
    render_sparse_and_dense(ctx, B.default_params(160, 90, 8, lanes_per_pixel=2))
    check_against_oracle(ctx, scene, B.default_params(64, 36, 4))


def test_cornell_box(ctx, cornell):
    """The benchmarked scene (14 of its 18 rectangles are turned about one axis), as benchmarked and otherwise."""
    ctx.upload_world(cornell)
    render_sparse_and_dense(ctx, B.default_params(960, 540, 64, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS))
    for imode in (B.INTERSECT_WIDE8, B.INTERSECT_SCALAR, B.INTERSECT_WIDE4):
        render_sparse_and_dense(ctx, B.default_params(320, 180, 16, lanes_per_pixel=2, intersect_mode=imode))
    render_sparse_and_dense(ctx, B.default_params(64, 36, 8, rng_mode=B.RNG_ROW_SERIAL, lanes_per_pixel=1))
