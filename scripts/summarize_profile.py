#!/usr/bin/env python3
"""Condenses a scripts/profile_gpu.sh output directory into one markdown summary:
kernel-trace statistics, the render kernel's counters, and the figures derived from them."""
import collections
import csv
import glob
import json
import os
import sys

src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
rays = int(sys.argv[4]) if len(sys.argv) > 4 else 1454628861
flop_per_ray = float(sys.argv[5]) if len(sys.argv) > 5 else 887
command = sys.argv[6] if len(sys.argv) > 6 else "--scene cornell --width 1920 --height 1080 --spp 256"

import re
_w, _h = re.search(r"--width (\d+)", command), re.search(r"--height (\d+)", command)
width, height = int(_w.group(1)) if _w else 1920, int(_h.group(1)) if _h else 1080

lines = [f"# {title}", "", "Command (per pass): `rocprofv3 <--kernel-trace --stats | --pmc ...> -- raytracer_amd/bin/raytracer "
         f"{command}` (scripts/profile_gpu.sh).", ""]
stats = list(csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))))
lines += ["## rocprofv3 --kernel-trace --stats", "", "| kernel | calls | total ns | average ns | % |", "|---|---|---|---|---|"]
kernel_ns = None
for r in stats:
    lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |")
    if "rt::render_kernel" in r["Name"] or "rt::pixel_combine_kernel" in r["Name"]:
        # one render = one launch of each of these kernels (a single one without tail compaction)
        kernel_ns = (kernel_ns or 0.0) + float(r["AverageNs"])
counters = collections.OrderedDict()
meta = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "rt::render_kernel" in r["Kernel_Name"] or "rt::pixel_combine_kernel" in r["Kernel_Name"]:
            counters[r["Counter_Name"]] = counters.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
lines += ["", "## render kernels (counters summed over the kernels of one render): dispatch of the last one", "", "```", json.dumps(meta), "```", "", "## render kernel: counters (one --pmc pass per group, summed over the chip)", "",
          "| counter | value |", "|---|---|"]
for k, v in counters.items():
    lines.append(f"| {k} | {v:.6g} |")
c = counters
d = []
if kernel_ns:
    ms = kernel_ns / 1e6
    d.append(f"kernel duration {ms:.3f} ms -> {rays / kernel_ns * 1e3:.0f} Mrays/s, {rays * flop_per_ray / kernel_ns / 1e3:.2f} TFLOP/s algorithmic "
             f"({rays} rays x {flop_per_ray} flop/ray)")
    if "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / kernel_ns
        d.append(f"effective clock {clk:.3f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration; the counter pass itself runs a little slower than the stats pass)")
        if "SQ_INSTS_VALU" in c:
            simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
            d.append(f"VALU wave-instructions {c['SQ_INSTS_VALU']:.4g}: {c['SQ_INSTS_VALU'] / rays * 64:.0f} per 64 rays issued, "
                     f"{simd_cycles / c['SQ_INSTS_VALU']:.2f} SIMD-cycles per VALU instruction over 1024 SIMDs")
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d.append(f"VALU lane utilisation {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64):.3f} (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))")
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        d.append(f"VALU busy {c['SQ_ACTIVE_INST_VALU'] * 4 / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} of SIMD cycles (SQ_ACTIVE_INST_VALU counts quad-cycles)")
    if "SQ_WAVES" in c and "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        d.append(f"mean resident waves per SIMD {c['SQ_WAVE_CYCLES'] * 4 / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f} (SQ_WAVE_CYCLES quad-cycles / SIMD cycles); {c['SQ_WAVES']:.0f} waves launched")
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        d.append(f"LDS bank-conflict cycles / LDS active cycles = {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.4f}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half the bytes of wide coalesced reads
        fetch = c["FETCH_SIZE"] * 1024 * 2
        write = c["WRITE_SIZE"] * 1024
        d.append(f"HBM traffic: FETCH_SIZE {c['FETCH_SIZE']:.1f} KiB (x2 gfx950 correction = {fetch / 1e6:.2f} MB), WRITE_SIZE {c['WRITE_SIZE']:.1f} KiB "
                 f"({write / 1e6:.2f} MB) = {(fetch + write) / 1e6:.2f} MB per render; algorithmic output {width}x{height}x4 B = {width * height * 4 / 1e6:.2f} MB "
                 f"(ratio {(fetch + write) / (width * height * 4):.1f}); {(fetch + write) / (kernel_ns * 1e-9) / 1e9:.2f} GB/s of ~8000")
lines += ["", "## derived", ""] + [f"- {x}" for x in d]
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(d))
