#!/usr/bin/env python3
"""Benchmark of the HIP path-tracing hot path: Mrays/s on the Cornell box at 1920x1080.

A step is one render of the whole image on every rank (scene already resident in
HBM, outputs left in HBM).  N = 1: Cornell 1920x1080 at 256 spp (the headline
single-GPU configuration).  N > 1, default --shard samples (weak scaling): every rank renders all
pixels at 256 spp with its own sample-batch indices, so the job is 1080p at 256*N
spp; the per-rank radiance sums are added onto rank 0 with one RCCL reduce and
rank 0 tonemaps.  --shard rows (strong scaling) splits the 256-spp image into
interleaved 8-row bands and gathers the packed rows with one RCCL all_gather.

Prints ONE JSON line on rank 0.  A ray is one world intersection, as in the
reference's own report (main.cpp:278,397,470).
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# FP32 operations per ray on the Cornell box (18 rectangles, no spheres or planes):
# 45 per rectangle test (two 3x4 affine applies 18+15, -oz/dz 2, p.xy 4, 6 compares),
# 62 for the scatter of a hit bounce, 40 per camera ray spread over ~2.74 rays/sample.
FLOP_PER_RAY_CORNELL = 18 * 45 + 62 + 15


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--lanes-per-pixel", type=int, default=4)
    ap.add_argument("--no-compact", action="store_true", help="single-launch render (no tail compaction)")
    ap.add_argument("--shard", choices=("samples", "rows"), default="samples")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(width, height, spp, threads):
    """Times the CPU path on this host: the reference's own code (oracle/_ref) when its harness
    was prebuilt, else this repo's AVX2 restatement of it."""
    if spp <= 0:
        return None
    threads = threads or min(16, os.cpu_count() or 1)
    sample = f"Cornell {width}x{height} at {spp} spp, one row per work order"
    harness = ROOT / "oracle" / "_ref" / "ref_harness"
    if harness.exists():
        out = subprocess.run([str(harness), "render", "cornell", str(width), str(height), str(spp), str(threads), "1", "-"],
                             check=True, stdout=subprocess.PIPE, text=True).stdout
        r = json.loads(out.strip().splitlines()[-1])
        return {"value": round(r["mrays_per_s"], 2), "unit": "Mrays/s", "cores": threads, "kind": "reference",
                "sample": sample, "seconds": round(r["ms"] / 1e3, 2)}
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    from raytracer_amd.host_scene import build_scene
    scene = O.OracleScene(build_scene("cornell"))
    p = O.default_params(width, height, spp, intersect_mode=O.WIDE8_AVX2)
    t0 = time.perf_counter()
    _, _, rays = scene.render(p, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": sample, "seconds": round(dt, 2)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from raytracer_amd import binding as B
    from raytracer_amd import distributed as D
    from raytracer_amd.host_scene import build_scene

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world_size
    distributed = world_size > 1

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)

    W, H, spp, B_lanes = args.width, args.height, args.spp, args.lanes_per_pixel
    ctx = B.Context(local_rank)
    info = ctx.device_info()
    ctx.upload_world(build_scene("cornell"))

    rows_mode = distributed and args.shard == "rows"
    base = B.default_params(W, H, spp, rng_mode=B.RNG_PIXEL, lanes_per_pixel=B_lanes,
                            flags=0 if args.no_compact else B.FLAG_COMPACT_TAILS)
    total = args.steps + args.warmup
    rays = torch.zeros(total, dtype=torch.int64, device=device)
    radiance = None
    sharded = None
    if rows_mode:
        sharded = D.RowShardedImage(base, rank, world_size, device)     # interleaved 8-row bands + all_gather buffers
        p, pixels = sharded.params, sharded.shard
    elif distributed:
        p = D.batch_params(base, rank, world_size, B.FLAG_OUTPUT_SUM)   # this rank's sample batches, un-normalised sum
        radiance = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
        pixels = torch.zeros((H, W), dtype=torch.int32, device=device)
    else:
        p = base
        pixels = torch.zeros((H, W), dtype=torch.int32, device=device)

    ctx.reserve(p)      # workspace of the compacted render: allocated here, not in a timed step
    stream = torch.cuda.current_stream(device)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(total)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(total)]

    def step(i):
        d_rays = rays.data_ptr() + 8 * i
        starts[i].record(stream)
        if radiance is not None:
            ctx.render_device(p, None, radiance.data_ptr(), d_rays, stream.cuda_stream)
        else:
            ctx.render_device(p, pixels.data_ptr(), None, d_rays, stream.cuda_stream)
        stops[i].record(stream)
        if rows_mode:
            sharded.gather()
        elif distributed:
            D.reduce_radiance_sum(radiance, world_size)
            if rank == 0:
                ctx.resolve_device(radiance.data_ptr(), W * H, spp * world_size, pixels.data_ptr(), None, stream.cuda_stream)

    def fence():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0

    timed_rays = rays[args.warmup:].sum()
    kernel_ms = sum(starts[i].elapsed_time(stops[i]) for i in range(args.warmup, total)) / max(args.steps, 1)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(timed_rays, op=dist.ReduceOp.SUM)
        km = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms = float(km.item())
    total_rays = int(timed_rays.item())
    rank_rays_per_launch = int(rays[args.warmup:].sum().item()) / max(args.steps, 1)

    if rank == 0:
        mrays = total_rays / elapsed / 1e6
        peak = info.fp32_vector_peak_tflops
        achieved = rank_rays_per_launch * FLOP_PER_RAY_CORNELL / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            try:
                key = f"cornell_{W}x{H}_{spp}spp" + ("_single_launch" if args.no_compact else "")
                traffic = json.loads(tfile.read_text()).get(key)
            except (ValueError, OSError):
                traffic = None
        if rows_mode:
            workload = f"Cornell box {W}x{H} {spp} spp, interleaved 8-row bands over {world_size} GPUs"
        elif distributed:
            workload = f"Cornell box {W}x{H} {spp * world_size} spp, {spp} spp per GPU (sample batches), RCCL reduce"
        else:
            workload = f"Cornell box {W}x{H} {spp} spp on one GPU"
        line = {
            "metric": "Mrays/s (whole node) Cornell box 1080p",
            "value": round(mrays, 1),
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if rows_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "rng_mode": "pixel", "lanes_per_pixel": B_lanes, "intersect_mode": "wide8",
                       "compact_tails": not args.no_compact,
                       "max_bounces": 8, "rays_per_step": total_rays // max(args.steps, 1),
                       "device": info.name.decode(), "compute_units": info.compute_units, "clock_mhz": info.clock_khz / 1e3},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "flop_per_ray": FLOP_PER_RAY_CORNELL, "kernel_ms": round(kernel_ms, 3),
                         "kernel": "rt::render_kernel<WIDE8, PIXEL>" if args.no_compact else
                                   "rt::render_kernel<WIDE8, PIXEL> stage 1 (dominant) + stage 2 + rt::pixel_combine_kernel, timed together"},
        }
        if world_size == 1:
            line["cpu_baseline"] = cpu_baseline(W, H, args.cpu_spp, args.cpu_threads)
        print(json.dumps(line), flush=True)

    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
