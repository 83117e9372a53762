#!/usr/bin/env python3
"""Turns the output of scripts/job_profiles_r02.sh (gpurun_out/r02_*) into the committed summaries under
profiles/: one markdown file per configuration, the kernel statistics of the bench command, and
profiles/hbm_traffic.json (HBM bytes per bench step: FETCH_SIZE x 2 + WRITE_SIZE, one pass each)."""
import csv
import json
import pathlib
import re
import shutil
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"
PROF = ROOT / "profiles"
flops = json.loads((PROF / "flop_per_ray.json").read_text())

CONFIGS = {
    "cornell": ("r02_cornell", "--scene cornell --width 1920 --height 1080 --spp 256 --lanes-per-pixel 4 --compact", 887.0),
    "glass": ("r02_glass", "--scene glass_spheres --width 1920 --height 1080 --spp 512 --bounces 32 --lanes-per-pixel 8 --compact", flops["glass"]["flop_per_ray"]),
    "spheres4096": ("r02_spheres4096", "--scene random_spheres --count 4096 --width 1920 --height 1080 --spp 64 --lanes-per-pixel 4 --compact", flops["spheres4096"]["flop_per_ray"]),
    "spheres": ("r02_spheres", "--scene spheres --width 1280 --height 720 --spp 512 --lanes-per-pixel 8 --compact", flops["spheres"]["flop_per_ray"]),
}
for name, (tag, args, f) in CONFIGS.items():
    log = (OUT / tag / "stats.log").read_text()
    rays = int(re.search(r"Total computed rays: (\d+)", log).group(1))
    subprocess.run([sys.executable, str(ROOT / "scripts" / "summarize_profile.py"), str(OUT / tag), str(PROF / f"r02_kernel_{name}.md"),
                    f"Round 2: render kernels on the {name} configuration (one cold render by the native driver)", str(rays), str(f), args],
                   check=True, stdout=subprocess.DEVNULL)
    print(name, rays)

bench = OUT / "r02_bench"
shutil.copy(bench / "stats" / "stats_kernel_stats.csv", PROF / "r02_bench_stats_kernel_stats.csv")
shutil.copy(bench / "bench.json", PROF / "r02_bench.json")
shutil.copy(bench / "bench_under_rocprof.json", PROF / "r02_bench_under_rocprof.json")


def per_step(name, steps=12):
    total = 0.0
    for r in csv.DictReader(open(bench / name / f"{name}_counter_collection.csv")):
        if "rt::render_kernel" in r["Kernel_Name"] or "rt::pixel_combine_kernel" in r["Kernel_Name"]:
            total += float(r["Counter_Value"])
    return total / steps


fetch_kib, write_kib = per_step("pmc_fetch"), per_step("pmc_write")
traffic = json.loads((PROF / "hbm_traffic.json").read_text())
traffic["cornell_1920x1080_256spp"] = int(fetch_kib * 1024 * 2 + write_kib * 1024)
traffic["_how_round2"] = (f"round 2 kernels, same procedure: FETCH_SIZE {fetch_kib:.0f} KiB x 2 + WRITE_SIZE {write_kib:.0f} KiB per step "
                          "(rocprofv3 --pmc, one pass each, of `python3 bench.py --steps 10 --warmup 2`, render kernels + pixel combine)")
(PROF / "hbm_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
print("traffic per step", traffic["cornell_1920x1080_256spp"])
