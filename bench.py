#!/usr/bin/env python3
"""Benchmark of the HIP path-tracing hot path: Mrays/s, by default on the Cornell box at 1920x1080.

A step is one render of the whole image on every rank (scene already resident in HBM, outputs left in
HBM).  N = 1: the configuration named by --config (default: Cornell 1920x1080 at 256 spp, the headline
single-GPU configuration).  N > 1, default --shard samples (weak scaling): every rank renders all pixels
at the configuration's spp with its own sample-batch indices, so the job is the same image at spp*N
samples per pixel; the per-rank radiance sums are gathered onto rank 0 with one RCCL gather, added there
in a fixed pairwise tree over the ranks (rt_combine_device) and tonemapped, which makes the image equal,
bit for bit, to one GPU rendering all the batches.  --shard rows (strong scaling) splits the image into
interleaved 8-row bands, gathers the packed rows with one RCCL all_gather and un-interleaves them.

`python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed.run) when it
was not started by a launcher; under torchrun it is one of the ranks.

Prints ONE JSON line on rank 0.  A ray is one world intersection, as in the reference's own report
(main.cpp:278,397,470).
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import socket
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# name: scene builder arguments, width, height, spp, bounce cap, lanes per pixel
CONFIGS = {
    "cornell": dict(scene=("cornell",), width=1920, height=1080, spp=256, bounces=8, lanes=4,
                    label="Cornell box 1920x1080"),
    "cornell1024": dict(scene=("cornell",), width=1920, height=1080, spp=1024, bounces=8, lanes=4,
                        label="Cornell box 1920x1080"),
    "cornell4k": dict(scene=("cornell",), width=3840, height=2160, spp=4096, bounces=8, lanes=4,
                      label="Cornell box 3840x2160"),
    "spheres": dict(scene=("spheres",), width=1280, height=720, spp=512, bounces=8, lanes=8,
                    label="spheres + plane (the reference's createScene) 1280x720"),
    "spheres4096": dict(scene=("random_spheres", 4096, 1), width=1920, height=1080, spp=64, bounces=8, lanes=4,
                        label="4096 random spheres + plane 1920x1080"),
    "glass": dict(scene=("glass_spheres", 5), width=1920, height=1080, spp=512, bounces=32, lanes=8,
                  label="glass-sphere lattice, 32 bounces, 1920x1080"),
}

# FP32 operations per ray on the Cornell box (18 rectangles, no spheres or planes): 45 per rectangle test
# (two 3x4 affine applies 18+15, -oz/dz 2, p.xy 4, 6 compares), 62 for the scatter of a bounce, 40 per
# camera ray spread over ~2.74 rays/sample.  This is the figure the roofline is priced with; the terms
# measured with the CPU oracle's counters (scatter only on the 68 % of rays that hit: 872) and the
# figures of the other configurations are in profiles/flop_per_ray.json (oracle/measure_flop_per_ray.py).
FLOP_PER_RAY_CORNELL = 18 * 45 + 62 + 15


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cornell")
    ap.add_argument("--width", type=int, default=0, help="override the configuration's image width")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--lanes-per-pixel", type=int, default=0)
    ap.add_argument("--no-compact", action="store_true", help="single-launch render (no tail compaction)")
    ap.add_argument("--no-cull", action="store_true", help="switch the per-wave primary-ray cull off (A/B measurements)")
    ap.add_argument("--shard", choices=("samples", "rows"), default="samples")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks that all render on GPU 0 and exchange through gloo + host copies: exercises the multi-rank path "
                         "where only one GPU exists; its throughput means nothing")
    ap.add_argument("--no-overlap", action="store_true", help="samples mode: gather and sum each step before the next one starts")
    ap.add_argument("--through-process-group", action="store_true",
                    help="with one rank under torch.distributed.run: run the multi-rank code path (process group, gather, fixed-order sum) anyway")
    ap.add_argument("--cpu-spp", type=int, default=-1, help="spp of the bounded CPU-baseline sample (0 = skip, -1 = sized for ~10 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every online core, as the reference does)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """Started as `python bench.py --gpus N` without a launcher: run N fresh ranks of this script under
    torch.distributed.run (one process per GPU) and pass their output through.  This process never
    touches the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(width, height, spp, threads):
    """Times the CPU path on this host: the reference's own code (oracle/_ref) when its harness was
    prebuilt, else this repo's AVX2 restatement of it.  Threads: every online core, as the reference
    starts one worker per core (platform_linux.cpp:13-15, main.cpp:450-461); the 16-thread figure (the
    CPU share of a one-GPU slice of the box) is reported beside it."""
    if spp == 0:
        return None
    online = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    harness = ROOT / "oracle" / "_ref" / "ref_harness"

    def run(n_threads, n_spp):
        if harness.exists():
            out = subprocess.run([str(harness), "render", "cornell", str(width), str(height), str(n_spp), str(n_threads), "1", "-"],
                                 check=True, stdout=subprocess.PIPE, text=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
            return r["mrays_per_s"], r["ms"] / 1e3, "reference"
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as O
        from raytracer_amd.host_scene import build_scene
        scene = O.OracleScene(build_scene("cornell"))
        p = O.default_params(width, height, n_spp, intersect_mode=O.WIDE8_AVX2)
        t0 = time.perf_counter()
        _, _, rays = scene.render(p, threads=n_threads)
        dt = time.perf_counter() - t0
        return rays / dt / 1e6, dt, "port"

    threads = threads or online
    if spp < 0:
        # size the sample for roughly 10 s of CPU work: a 2-spp probe gives the rate
        rate, _, _ = run(threads, 2)
        wanted = rate * 1e6 * 10 / (width * height * 2.74)     # samples per pixel that take 10 s at the probed rate
        spp = max(8, min(256, 1 << (max(int(wanted), 8).bit_length() - 1)))
    value, seconds, kind = run(threads, spp)
    result = {"value": round(value, 2), "unit": "Mrays/s", "cores": threads, "cores_online": online, "cpu_model": cpu_model(),
              "kind": kind, "sample": f"Cornell {width}x{height} at {spp} spp, one row per work order, best of 1",
              "seconds": round(seconds, 2)}
    if threads != 16 and online >= 16:
        v16, s16, _ = run(16, max(8, min(spp, 32)))
        result["value_16_threads"] = round(v16, 2)
        result["seconds_16_threads"] = round(s16, 2)
    return result


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    from raytracer_amd import binding as B
    from raytracer_amd import distributed as D
    from raytracer_amd.host_scene import build_scene

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world_size
    # --through-process-group (under a launcher): also a one-rank job goes through the process group, so that
    # a one-GPU box exercises the RCCL calls of the N-GPU job.
    distributed = world_size > 1 or (args.through_process_group and "RANK" in os.environ)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)   # RCCL

    cfg = CONFIGS[args.config]
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    spp, B_lanes = args.spp or cfg["spp"], args.lanes_per_pixel or cfg["lanes"]
    ctx = B.Context(local_rank)
    info = ctx.device_info()
    ctx.upload_world(build_scene(*cfg["scene"]))

    rows_mode = distributed and args.shard == "rows"
    base = B.default_params(W, H, spp, rng_mode=B.RNG_PIXEL, lanes_per_pixel=B_lanes, max_bounces=cfg["bounces"],
                            flags=(0 if args.no_compact else B.FLAG_COMPACT_TAILS) | (B.FLAG_NO_PRIMARY_CULL if args.no_cull else 0))
    total = args.steps + args.warmup
    rays = torch.zeros(total, dtype=torch.int64, device=device)
    radiance = gathered = sharded = None
    pixels = torch.zeros((H, W), dtype=torch.int32, device=device)
    if rows_mode:
        sharded = D.RowShardedImage(base, rank, world_size, device)     # interleaved 8-row bands + all_gather buffers
        p = sharded.params
    elif distributed:
        p = D.batch_params(base, rank, world_size, B.FLAG_OUTPUT_SUM)   # this rank's sample batches, un-normalised sum
        radiance = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
        gathered = D.GatheredSums(radiance, rank, world_size)
    else:
        p = base

    ctx.reserve(p)      # workspace of the compacted render: allocated here, not in a timed step
    stream = torch.cuda.current_stream(device)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(total)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(total)]

    # samples mode: the gather of step i travels (on RCCL's stream) while step i + 1 renders into the other
    # buffer; rank 0 adds and resolves step i after step i + 1 has been enqueued.  drain() completes the last one.
    overlap = distributed and not rows_mode and not args.no_overlap
    buffers = [radiance, torch.zeros_like(radiance)] if overlap else [radiance]
    combined = torch.zeros_like(radiance) if distributed and not rows_mode and rank == 0 else None   # rank 0: sum over the ranks
    in_flight = []      # at most one (handle) of a gather that has been started and not yet combined

    def combine_on_rank_0():
        if rank == 0:
            # fixed pairwise tree over the ranks, then color / sampleSize + pack (main.cpp:393)
            ctx.combine_device(gathered.parts.data_ptr(), world_size, H * W * 3, H * W * 3, combined.data_ptr(), stream.cuda_stream)
            ctx.resolve_device(combined.data_ptr(), W * H, spp * world_size, pixels.data_ptr(), None, stream.cuda_stream)

    def drain():
        while in_flight:
            gathered.finish(in_flight.pop())
            combine_on_rank_0()

    def step(i):
        d_rays = rays.data_ptr() + 8 * i
        starts[i].record(stream)
        if rows_mode:
            ctx.render_device(p, sharded.shard.data_ptr(), None, d_rays, stream.cuda_stream)
        elif distributed:
            target = buffers[i % len(buffers)]
            ctx.render_device(p, None, target.data_ptr(), d_rays, stream.cuda_stream)
        else:
            ctx.render_device(p, pixels.data_ptr(), None, d_rays, stream.cuda_stream)
        stops[i].record(stream)
        if rows_mode:
            sharded.gather()
            if rank == 0:
                sharded.assemble(out=pixels)
        elif distributed:
            drain()                                   # the previous step's sums: arrived while this step rendered
            in_flight.append(gathered.start(target))
            if not overlap:
                drain()

    def fence():
        drain()
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

    # Outside the timed region, N > 1: rank 0 renders the whole job alone and compares it with the image the
    # ranks composed (pixels and ray count), so every multi-GPU line says whether the composition was exact.
    composition = None
    if distributed and rank == 0:
        if rows_mode:
            whole = base
        else:
            whole = B.default_params(W, H, spp * world_size, rng_mode=B.RNG_PIXEL, lanes_per_pixel=B_lanes * world_size,
                                     max_bounces=cfg["bounces"]) if B_lanes * world_size <= 64 else None
        if whole is not None:
            check_pixels = torch.zeros((H, W), dtype=torch.int32, device=device)
            check_rays = torch.zeros(1, dtype=torch.int64, device=device)
            whole.flags = 0
            ctx.render_device(whole, check_pixels.data_ptr(), None, check_rays.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(device)
            composition = {"pixels_identical_to_one_gpu_render": bool(torch.equal(check_pixels, pixels)),
                           "one_gpu_rays": int(check_rays.item())}

    timed_rays = rays[args.warmup:].sum()
    kernel_ms = sum(starts[i].elapsed_time(stops[i]) for i in range(args.warmup, total)) / max(args.steps, 1)
    rank_rays_per_launch = int(timed_rays.item()) / max(args.steps, 1)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if args.rehearse_on_one_gpu:
            t, timed_rays = t.cpu(), timed_rays.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(timed_rays, op=dist.ReduceOp.SUM)
        km = torch.tensor([kernel_ms], dtype=torch.float64, device=t.device)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms = float(km.item())
    total_rays = int(timed_rays.item())

    if rank == 0:
        mrays = total_rays / elapsed / 1e6
        peak = info.fp32_vector_peak_tflops
        flop_terms = None
        flop_per_ray = float(FLOP_PER_RAY_CORNELL)
        ffile = ROOT / "profiles" / "flop_per_ray.json"
        if ffile.exists():
            try:
                entry = json.loads(ffile.read_text()).get(args.config)
                if entry:
                    flop_terms = entry["terms"]
                    if not args.config.startswith("cornell"):
                        flop_per_ray = float(entry["flop_per_ray"])
            except (ValueError, OSError, KeyError):
                pass
        elif not args.config.startswith("cornell"):
            flop_per_ray = float("nan")
        achieved = rank_rays_per_launch * flop_per_ray / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists() and args.config == "cornell":
            try:
                key = f"cornell_{W}x{H}_{spp}spp" + ("_single_launch" if args.no_compact else "")
                traffic = json.loads(tfile.read_text()).get(key)
            except (ValueError, OSError):
                traffic = None
        label = cfg["label"] if (W, H) == (cfg["width"], cfg["height"]) else f"{cfg['label'].rsplit(' ', 1)[0]} {W}x{H}"
        if rows_mode:
            workload = f"{label} {spp} spp, interleaved 8-row bands over {world_size} GPUs, RCCL all_gather"
        elif distributed:
            workload = (f"{label} {spp * world_size} spp: {spp} spp per GPU (sample batches), "
                        f"RCCL gather + fixed-order sum on rank 0")
        else:
            workload = f"{label} {spp} spp on one GPU"
        compacted = not args.no_compact and spp // B_lanes >= 8       # the library's rule (rt_api.hip)
        line = {
            "metric": ("Mrays/s (whole node) Cornell box 1080p" if (W, H) == (1920, 1080) else f"Mrays/s (whole node) Cornell box {W}x{H}")
                      if args.config.startswith("cornell") else f"Mrays/s (whole node) {args.config}",
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
            "config": {"workload": workload, "name": args.config, "rng_mode": "pixel", "lanes_per_pixel": B_lanes, "intersect_mode": "wide8",
                       "compact_tails": compacted, "max_bounces": cfg["bounces"], "rays_per_step": total_rays // max(args.steps, 1),
                       "shard": args.shard if distributed else None, "rehearsal_on_one_gpu": bool(args.rehearse_on_one_gpu),
                       "device": info.name.decode(), "compute_units": info.compute_units, "clock_mhz": info.clock_khz / 1e3},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": "profiles/hbm_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of this command, not measured by this run)" if traffic else None,
                         "flop_per_ray": flop_per_ray, "flop_terms": flop_terms, "kernel_ms": round(kernel_ms, 3),
                         "kernel": "rt::render_kernel<WIDE8, PIXEL> stage 1 (dominant) + stage 2 + rt::pixel_combine_kernel, timed together" if compacted else
                                   "rt::render_kernel<WIDE8, PIXEL>"},
        }
        if composition is not None:
            composition["rays_identical"] = composition.pop("one_gpu_rays") == total_rays // max(args.steps, 1)
            line["composition_check"] = composition
        if world_size == 1 and not distributed:
            line["cpu_baseline"] = cpu_baseline(W if args.config.startswith("cornell") else 1920, H if args.config.startswith("cornell") else 1080,
                                                args.cpu_spp, args.cpu_threads)
        print(json.dumps(line), flush=True)

    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
