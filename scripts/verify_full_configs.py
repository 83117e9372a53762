#!/usr/bin/env python3
"""One-off, on the GPU box: every bench.py configuration at its FULL size and sample count, in the launch sequence
bench.py times (lanes per pixel, tail compaction, primary-ray cull), against the CPU oracle (tests/oracle_lib.py),
band of rows by band of rows: packed pixels within the 8-bit tolerance of the parity tests, linear radiance bit for
bit, ray counts equal.  The oracle needs seconds to a quarter of an hour per configuration, which is why this is a
script and not part of the GPU suite.  Appends one JSON object per (configuration, row range) to the output file.

    python scripts/verify_full_configs.py OUT.jsonl CONFIG [ROW_BEGIN ROW_END]"""
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench                                              # noqa: E402
import oracle_lib as O                                    # noqa: E402  (the checker)
from raytracer_amd import binding as B                    # noqa: E402
from raytracer_amd.host_scene import build_scene          # noqa: E402


def main():
    out, name = sys.argv[1], sys.argv[2]
    cfg = bench.CONFIGS[name]
    w, h = cfg["width"], cfg["height"]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, h)
    scene = build_scene(*cfg["scene"])
    oracle = O.OracleScene(scene)
    ctx = B.Context(0)
    ctx.upload_world(scene)
    band = max(8, 4_000_000_000 // max(1, w * cfg["spp"] * 60))      # rows per oracle call: some tens of seconds each
    rays_gpu = rays_cpu = 0
    differing_channels = differing_radiance = 0
    t0 = time.time()
    for r0 in range(lo, hi, band):
        r1 = min(hi, r0 + band)
        p = B.default_params(w, h, cfg["spp"], lanes_per_pixel=cfg["lanes"], max_bounces=cfg["bounces"], flags=B.FLAG_COMPACT_TAILS,
                             row_begin=r0, row_end=r1)
        pix, rad, rays = ctx.render(p)
        op = O.default_params(w, h, cfg["spp"], row_begin=r0, row_end=r1, max_bounces=cfg["bounces"], rng_mode=O.PIXEL,
                              lanes_per_pixel=cfg["lanes"], intersect_mode=O.WIDE8_AVX2 if not len(scene.spheres) else O.WIDE8_IEEE)
        opix, orad, orays = oracle.render(op, threads=16)
        rays_gpu += rays
        rays_cpu += orays
        differing_radiance += int((rad.view(np.uint32) != orad.view(np.uint32)).sum())
        ch = np.abs(((pix[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(int) -
                    ((opix[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(int))
        assert ch.max() <= 1
        differing_channels += int((ch != 0).sum())
        print(f"{name} rows [{r0},{r1}) of [{lo},{hi}): rays {rays} / {orays}, radiance values differing so far {differing_radiance}, "
              f"{time.time() - t0:.0f} s", flush=True)
    result = {"config": name, "width": w, "height": h, "spp": cfg["spp"], "lanes_per_pixel": cfg["lanes"], "max_bounces": cfg["bounces"],
              "rows": [lo, hi], "rays_gpu": rays_gpu, "rays_oracle": rays_cpu, "radiance_values_differing": differing_radiance,
              "pixel_channels_differing_by_one_lsb": differing_channels, "pixel_channels": (hi - lo) * w * 3,
              "oracle_seconds": round(time.time() - t0, 1)}
    with open(out, "a") as f:
        f.write(json.dumps(result) + "\n")
    print(json.dumps(result))
    return 0 if rays_gpu == rays_cpu and differing_radiance == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
