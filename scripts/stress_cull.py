#!/usr/bin/env python3
"""One-off stress of the primary-ray cull on the GPU box: N random rectangle worlds with cameras placed to graze
rectangle planes and edges, rendered with and without the cull; any differing bit is reported.
    python scripts/stress_cull.py [count] [seed]"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from raytracer_amd import binding as B          # noqa: E402
from raytracer_amd import scene_io              # noqa: E402
from test_gpu_cull import camera, materials, random_rectangles, rectangle   # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = B.Context(0)
    bad = 0
    for it in range(count):
        nm = int(rng.integers(2, 6))
        kind = it % 5
        scales = [(0.05, 5.0), (1e-3, 50.0), (0.5, 2.0), (1e-2, 1e2), (0.2, 8.0)][kind]
        rects = random_rectangles(rng, int(rng.integers(1, 40)), nm, spread=float(rng.choice([2.0, 7.0, 30.0])), scales=scales)
        pos = rng.normal(size=3) * float(rng.choice([3.0, 12.0, 60.0]))
        target = rng.normal(size=3) * 2.0
        if kind == 1 and len(rects):          # camera almost in the plane of a rectangle, looking along it
            m = np.linalg.inv(rects["inverse_transform"][0].astype(np.float64))
            pos = m[:3, 3] + m[:3, 0] * rng.uniform(-3, 3) + m[:3, 1] * rng.uniform(-3, 3) + m[:3, 2] * float(rng.choice([0.0, 1e-7, 1e-4, -1e-5]))
            target = m[:3, 3] + m[:3, 0] * rng.uniform(-1, 1)
        if kind == 3:                          # camera far away, small field covered
            pos = pos * 1.0e3
        if np.linalg.norm(pos - target) < 1e-3 or abs(np.dot((pos - target) / np.linalg.norm(pos - target), [0, 1, 0])) > 0.999:
            continue
        scene = scene_io.Scene(materials(rng, nm), np.zeros(0, scene_io.PLANE), np.zeros(0, scene_io.SPHERE), rects, camera(pos, target))
        try:
            ctx.upload_world(scene)
        except B.RtError:
            continue
        lanes = int(rng.choice([1, 2, 4, 8]))
        w, h = int(rng.integers(16, 300)), int(rng.integers(16, 200))
        p = B.default_params(w, h, lanes * int(rng.choice([1, 4])), lanes_per_pixel=lanes, max_bounces=int(rng.choice([1, 4])),
                             intersect_mode=int(rng.integers(0, 3)))
        a = ctx.render(p)
        q = B.RenderParams.from_buffer_copy(p)
        q.flags |= B.FLAG_NO_PRIMARY_CULL
        b = ctx.render(q)
        if a[2] != b[2] or not np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) or not np.array_equal(a[0], b[0]):
            bad += 1
            print("MISMATCH at world", it, "kind", kind, "size", w, h, "lanes", lanes, "rays", a[2], b[2], flush=True)
    print(f"{count} worlds, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
