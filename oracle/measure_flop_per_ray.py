#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - measures, with the CPU oracle's event counters, the terms of the algorithmic
flop count per ray of every benchmark configuration and writes them to profiles/flop_per_ray.json
(data that bench.py reads; bench.py itself does not run the oracle for this).

    F_ray = 17 P + 20 S + 11 S_disc + 18 S_upd + 45 R + 6 R_upd + 62 [hit] + 60 [refracting hit] + 40 / rho

P, S, R: planes, spheres, rectangles of the scene (every ray tests all of them); S_disc: spheres per ray
whose discriminant is positive (roots evaluated, main.cpp:73-77); S_upd / R_upd: closest-hit updates per
ray; [hit]: fraction of rays that hit (scatter, main.cpp:285-322); [refracting hit]: fraction of rays
whose hit refracts (math_util.h:52-93); rho: rays per sample (camera ray, main.cpp:381-388).
The per-ray averages do not depend on the resolution beyond sampling noise, so they are measured on a
reduced image (same aspect ratio, same bounce cap) with per-pixel streams.

    python oracle/measure_flop_per_ray.py
"""
import ctypes as C
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib as O                                   # noqa: E402
from raytracer_amd.host_scene import build_scene          # noqa: E402


class Statistics(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "samples", "hits", "refracting_hits", "sphere_roots",
                                           "sphere_updates", "rectangle_updates")]


CONFIGS = {
    # name: (scene arguments, width, height, spp of the measurement, max_bounces)
    "cornell": (("cornell",), 480, 270, 64, 8),
    "spheres": (("spheres",), 320, 180, 64, 8),
    "spheres4096": (("random_spheres", 4096, 1), 192, 108, 4, 8),
    "glass": (("glass_spheres", 5), 480, 270, 64, 32),
}


def measure(name):
    scene_args, w, h, spp, bounces = CONFIGS[name]
    scene = build_scene(*scene_args)
    p = O.default_params(w, h, spp, rng_mode=O.PIXEL, lanes_per_pixel=1, intersect_mode=O.WIDE8_IEEE, max_bounces=bounces)
    O.OracleScene(scene).render(p, threads=8)
    st = Statistics()
    O.lib().rto_last_statistics(C.byref(st))
    rays = st.rays
    P, S, R = len(scene.planes), len(scene.spheres), len(scene.rectangles)
    terms = {
        "planes": P, "spheres": S, "rectangles": R,
        "sphere_roots_per_ray": st.sphere_roots / rays,
        "sphere_updates_per_ray": st.sphere_updates / rays,
        "rectangle_updates_per_ray": st.rectangle_updates / rays,
        "hit_fraction": st.hits / rays,
        "refracting_hit_fraction": st.refracting_hits / rays,
        "rays_per_sample": rays / st.samples,
    }
    f = (17 * P + 20 * S + 11 * terms["sphere_roots_per_ray"] + 18 * terms["sphere_updates_per_ray"] + 45 * R +
         6 * terms["rectangle_updates_per_ray"] + 62 * terms["hit_fraction"] + 60 * terms["refracting_hit_fraction"] +
         40 / terms["rays_per_sample"])
    terms = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in terms.items()}
    return {"flop_per_ray": round(f, 1), "terms": terms,
            "measured_on": f"{' '.join(map(str, scene_args))} {w}x{h} {spp} spp, {bounces} bounces, {rays} rays (CPU oracle counters)"}


if __name__ == "__main__":
    out = {name: measure(name) for name in CONFIGS}
    out["cornell1024"] = out["cornell"]
    out["cornell4k"] = out["cornell"]      # the same scene and bounce cap at another image size and sample count
    path = ROOT / "profiles" / "flop_per_ray.json"
    path.write_text(json.dumps(out, indent=1) + "\n")
    for k, v in out.items():
        print(k, v["flop_per_ray"], v["terms"])
