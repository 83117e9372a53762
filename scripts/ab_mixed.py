#!/usr/bin/env python3
"""On the GPU box: the Cornell box with three spheres put into it (a world that mixes spheres and rectangles),
1920x1080 at 64 spp, with the rectangles' classed transforms and with RT_FLAG_DENSE_TRANSFORMS: kernel time of each,
and whether pixels, radiance and ray count agree."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from raytracer_amd import binding as B          # noqa: E402
from raytracer_amd import scene_io              # noqa: E402

scene = scene_io.read_scene(ROOT / "tests" / "golden" / "cornell.rtsc")
spheres = np.zeros(3, dtype=scene_io.SPHERE)
spheres["position"] = [(-3.0, -5.0, -2.0), (3.5, -6.0, 1.0), (0.0, 2.0, -4.0)]
spheres["radius"] = [2.0, 1.5, 1.0]
spheres["material_index"] = [1, 2, 3]
mixed = scene_io.Scene(scene.materials, scene.planes, spheres, scene.rectangles, scene.camera)
with B.Context(0) as ctx:
    ctx.upload_world(mixed)
    results = {}
    for name, extra in (("classed", 0), ("dense", B.FLAG_DENSE_TRANSFORMS)):
        p = B.default_params(1920, 1080, 64, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS | B.FLAG_TIME_KERNEL | extra)
        times = []
        for _ in range(4):
            out = ctx.render(p)
            times.append(ctx.last_kernel_ms())
        results[name] = out
        print(f"{name:8s} kernel ms {min(times[1:]):.3f}  rays {out[2]}")
    a, b = results["classed"], results["dense"]
    print("identical:", a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))
