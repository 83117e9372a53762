#!/usr/bin/env python3
"""Turns the output of scripts/job_profiles_r03.sh (gpurun_out/r03_*) into the committed summaries under profiles/:
one markdown file per configuration, the kernel statistics of the bench command, profiles/hbm_traffic.json (HBM
bytes per render of every configuration: FETCH_SIZE x 2 + WRITE_SIZE, one --pmc pass each, summed over the
kernels of the render) and profiles/cull_stats.json (rectangle pairs tested per ray, from the diagnostic build)."""
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
    "cornell": ("--scene cornell --width 1920 --height 1080 --spp 256 --lanes-per-pixel 4 --compact", (1920, 1080, 256)),
    "cornell1024": ("--scene cornell --width 1920 --height 1080 --spp 1024 --lanes-per-pixel 4 --compact", (1920, 1080, 1024)),
    "glass": ("--scene glass_spheres --width 1920 --height 1080 --spp 512 --bounces 32 --lanes-per-pixel 8 --compact", (1920, 1080, 512)),
    "spheres4096": ("--scene random_spheres --count 4096 --width 1920 --height 1080 --spp 64 --lanes-per-pixel 4 --compact", (1920, 1080, 64)),
    "spheres": ("--scene spheres --width 1280 --height 720 --spp 512 --lanes-per-pixel 8 --compact", (1280, 720, 512)),
}


def render_counter(directory, name):
    """Sum of a counter over the kernels of one render (the driver renders once per pass)."""
    total = 0.0
    for r in csv.DictReader(open(directory / name / f"{name}_counter_collection.csv")):
        if "rt::render_kernel" in r["Kernel_Name"] or "rt::pixel_combine_kernel" in r["Kernel_Name"]:
            total += float(r["Counter_Value"])
    return total


traffic = {"_how": "HBM bytes per render, measured: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of one render "
                   "by the native driver with the configuration's launch sequence (lanes per pixel, tail compaction, primary-ray "
                   "cull: what bench.py times), summed over the render's kernels (stage 1, stage 2, pixel combine): FETCH_SIZE "
                   "KiB x 1024 x 2 (gfx950 reports half the bytes of wide coalesced reads) + WRITE_SIZE KiB x 1024.  "
                   "`_algorithmic`: packed pixels (4 B each) + the 8-byte ray counter.  The rest is the tail compaction's workspace "
                   "(deposited stream states of 80 B written by stage 1 and read by stage 2, per-stream sums of pending pixels).",
           "_algorithmic": {}}
for name, (args, (w, h, spp)) in CONFIGS.items():
    tag = OUT / f"r03_{name}"
    log = (tag / "stats.log").read_text()
    rays = int(re.search(r"Total computed rays: (\d+)", log).group(1))
    f = flops[name]["flop_per_ray"]
    subprocess.run([sys.executable, str(ROOT / "scripts" / "summarize_profile.py"), str(tag), str(PROF / f"r03_kernel_{name}.md"),
                    f"Round 3: render kernels on the {name} configuration (one cold render by the native driver)", str(rays), str(f), args],
                   check=True, stdout=subprocess.DEVNULL)
    fetch_kib, write_kib = render_counter(tag, "pmc_fetch"), render_counter(tag, "pmc_write")
    traffic[f"{name}_{w}x{h}_{spp}spp"] = int(fetch_kib * 1024 * 2 + write_kib * 1024)
    traffic["_algorithmic"][f"{name}_{w}x{h}_{spp}spp"] = w * h * 4 + 8
    print(name, rays, "traffic", traffic[f"{name}_{w}x{h}_{spp}spp"], "ratio %.1f" % (traffic[f"{name}_{w}x{h}_{spp}spp"] / (w * h * 4)))

bench = OUT / "r03_bench"
shutil.copy(bench / "stats" / "stats_kernel_stats.csv", PROF / "r03_bench_stats_kernel_stats.csv")
shutil.copy(bench / "bench.json", PROF / "r03_bench.json")
shutil.copy(bench / "bench_under_rocprof.json", PROF / "r03_bench_under_rocprof.json")


def bench_per_step(name, steps=12):
    return render_counter(bench, name) / steps


traffic["bench_command_cornell_per_step"] = int(bench_per_step("pmc_fetch") * 1024 * 2 + bench_per_step("pmc_write") * 1024)
(PROF / "hbm_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")

text = (OUT / "r03_stamp_breakdown.txt").read_text()
shutil.copy(OUT / "r03_stamp_breakdown.txt", PROF / "r03_stamp_breakdown.txt")
pairs = {}
for block in text.split("== ")[1:]:
    m = re.search(r"rectangle pairs tested per ray: ([\d.]+) of (\d+)", block)
    if m and block.split()[0] not in pairs:
        pairs[block.split()[0]] = {"tested_per_ray": float(m.group(1)), "pairs_in_scene": int(m.group(2))}
(PROF / "cull_stats.json").write_text(json.dumps({
    "_how": "diagnostic build (-DRT_STAMPS, scripts/stamp_breakdown.py): rectangle pairs visited, weighted by the rays in flight, "
            "over the rays traced; the rest was skipped by the per-wave primary-ray cull", "rect_pairs_tested_per_ray": pairs}, indent=1) + "\n")
print("pairs", pairs)
