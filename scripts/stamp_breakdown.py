#!/usr/bin/env python3
"""Where the render kernel's waves spend their cycles: renders a configuration with the diagnostic build
of the library (build/librt_hip_stamps.so, compiled with -DRT_STAMPS) and prints the share of wave cycles
per region of the loop and the lane counts of the divergent blocks.

    RT_HIP_LIBRARY=$PWD/build/librt_hip_stamps.so python scripts/stamp_breakdown.py [--no-cull] [--spp=N] cornell glass ...
"""
import ctypes as C
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                                            # noqa: E402  (the configurations)
from raytracer_amd import binding as B                  # noqa: E402
from raytracer_amd.host_scene import build_scene        # noqa: E402

NAMES = ["start", "control+normalize", "planes", "sphere terms", "sphere roots", "rects", "resolve hit", "shade", "epilogue",
         "iterations", "lanes tracing", "roots bodies", "roots lanes", "start lanes", "hit lanes", "dielectric bodies", "dielectric lanes",
         "rect pairs", "rect pairs behind", "rects out of range", "rect range lanes", "rect pairs skipped", "rect pair rays"]


def cfg_pairs(cfg):
    return (len(build_scene(*cfg["scene"]).rectangles) + 1) // 2


def main():
    lib = B.load_library()
    lib.rt_debug_read_stamps.restype = C.c_int
    lib.rt_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    table = (C.c_ulonglong * len(NAMES))()
    no_cull = "--no-cull" in sys.argv
    for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["cornell"]:
        cfg = bench.CONFIGS[name]
        spp = max(cfg["lanes"] * 64, cfg["spp"] // 4)
        for a in sys.argv[1:]:
            if a.startswith("--spp="):
                spp = int(a[6:])
        with B.Context(0) as ctx:
            ctx.upload_world(build_scene(*cfg["scene"]))
            p = B.default_params(cfg["width"], cfg["height"], spp, lanes_per_pixel=cfg["lanes"], max_bounces=cfg["bounces"],
                                 flags=B.FLAG_COMPACT_TAILS | (B.FLAG_NO_PRIMARY_CULL if no_cull else 0))
            lib.rt_debug_read_stamps(None, 1)
            _, _, rays = ctx.render(p, want_radiance=False)
            lib.rt_debug_read_stamps(table, 1)
        v = dict(zip(NAMES, [int(x) for x in table]))
        cycles = sum(v[n] for n in NAMES[:9])
        it = v["iterations"]
        print(f"== {name} ({cfg['width']}x{cfg['height']} {spp} spp, {rays} rays): {it} wave-iterations, "
              f"{cycles / it:.0f} wave cycles per iteration, lanes tracing {v['lanes tracing'] / it / 64:.3f}")
        for n in NAMES[:9]:
            print(f"   {n:20s} {100 * v[n] / cycles:6.2f} %   {v[n] / it:8.1f} cycles per iteration")
        print(f"   start block: {v['start lanes'] / it:.1f} lanes per iteration; hits: {v['hit lanes'] / it:.1f} lanes per iteration")
        if v["roots bodies"]:
            print(f"   sphere roots: {v['roots bodies'] / it:.2f} bodies per iteration with {v['roots lanes'] / v['roots bodies']:.1f} lanes each")
        if v["rect pairs"]:
            print(f"   rectangle pairs: {v['rect pairs'] / it:.2f} per iteration; both planes behind every ray of the wave: "
                  f"{100 * v['rect pairs behind'] / v['rect pairs']:.2f} %; rectangles that no lane's t can improve on "
                  f"(min_hit < t < closest fails on all lanes): {100 * v['rects out of range'] / (2 * v['rect pairs']):.2f} %, "
                  f"{v['rect range lanes'] / (2 * v['rect pairs']):.1f} lanes pass it on average")
            print(f"   primary-ray cull: {v['rect pairs skipped'] / it:.3f} pairs skipped per iteration; "
                  f"rectangle pairs tested per ray: {v['rect pair rays'] / v['lanes tracing']:.3f} of {cfg_pairs(cfg)}")
        if v["dielectric bodies"]:
            print(f"   dielectric block: {v['dielectric bodies'] / it:.2f} per iteration with {v['dielectric lanes'] / v['dielectric bodies']:.1f} lanes each")


if __name__ == "__main__":
    main()
