#!/usr/bin/env python3
"""One-off stress of the classed rectangle transforms on the GPU box: N random worlds of rectangles turned about one
axis (mixed with rectangles in general position), cameras on coordinate planes and axes, rendered with the classes
and with RT_FLAG_DENSE_TRANSFORMS | RT_FLAG_NO_PRIMARY_CULL (the plain kernel, nothing skipped); any differing bit
is reported.
    python scripts/stress_sparse.py [count] [seed]"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from raytracer_amd import binding as B          # noqa: E402
from raytracer_amd import scene_io              # noqa: E402
from test_gpu_cull import camera, materials, random_rectangles   # noqa: E402
from test_gpu_sparse import turned_rectangles   # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = B.Context(0)
    bad = 0
    for it in range(count):
        nm = int(rng.integers(3, 6))
        rects = np.concatenate([turned_rectangles(rng, int(rng.integers(1, 60)), nm, spread=float(rng.choice([2.0, 6.0, 20.0]))),
                                random_rectangles(rng, int(rng.integers(0, 8)), nm)])
        rects = rects[rng.permutation(len(rects))]
        pos = rng.normal(size=3) * float(rng.choice([3.0, 12.0, 40.0]))
        pos[rng.random(3) < 0.4] = 0.0                     # coordinates that are exact zeros
        target = np.round(rng.normal(size=3) * 2.0) * (rng.random(3) < 0.6)
        if np.linalg.norm(pos - target) < 1e-3 or abs(np.dot((pos - target) / np.linalg.norm(pos - target), [0, 1, 0])) > 0.999:
            continue
        scene = scene_io.Scene(materials(rng, nm), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rects, camera(pos, target))
        ctx.upload_world(scene)
        lanes = int(rng.choice([1, 2, 4, 8]))
        w, h = int(rng.integers(16, 300)), int(rng.integers(16, 200))
        p = B.default_params(w, h, lanes * int(rng.choice([1, 4, 8])), lanes_per_pixel=lanes, max_bounces=int(rng.choice([1, 4, 8])),
                             intersect_mode=int(rng.choice([0, 0, 2])), flags=int(rng.choice([0, B.FLAG_COMPACT_TAILS])))
        a = ctx.render(p)
        q = B.RenderParams.from_buffer_copy(p)
        q.flags |= B.FLAG_DENSE_TRANSFORMS | B.FLAG_NO_PRIMARY_CULL
        b = ctx.render(q)
        if a[2] != b[2] or not np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) or not np.array_equal(a[0], b[0]):
            bad += 1
            print("MISMATCH at world", it, "size", w, h, "lanes", lanes, "rays", a[2], b[2], flush=True)
        if it % 500 == 0:
            print("world", it, flush=True)
    print(f"{count} worlds, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
