#!/usr/bin/env python3
"""Benchmark of the HIP path-tracing hot path: Mrays/s, by default on the Cornell box at 1920x1080.

A step is one render of the whole job (scene already resident in HBM, outputs left in HBM).

N = 1: the configuration named by --config (default: Cornell 1920x1080 at 256 spp, the headline single-GPU
configuration).

N > 1, default (--shard samples --scaling strong): a FIXED job split over the GPUs - for the Cornell box the
1920x1080 image at 1024 samples per pixel in total (other configurations: their own sample count): every rank
renders all pixels with 1024 / N samples and its own sample-batch indices; the per-rank radiance sums are
gathered onto rank 0 with one RCCL gather, added there in a fixed pairwise tree over the ranks
(rt_combine_device) and tonemapped, which makes the image equal, bit for bit, to one GPU rendering all the
batches.  --scaling weak keeps the configuration's spp PER RANK instead (the job grows with N).  --shard rows
(strong scaling as well) splits the image into interleaved 8-row bands, gathers the packed rows with one RCCL
all_gather and un-interleaves them.  Every N > 1 line carries a composition check (rank 0 renders the whole job
alone, outside the timed region, and compares pixels and ray count) and the one-GPU time of the same job, from
which `efficiency_vs_one_gpu` is computed.

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
    "cornell": dict(scene=("cornell",), width=1920, height=1080, spp=256, bounces=8, lanes=4, strong_spp=1024,
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

# FP32 operations per ray, F_ray = 17 P + 20 S + 11 S_disc + 18 S_upd + 45 R + 6 R_upd + 62 [hit] + 60 [refracting hit]
# + 40 / rays-per-sample, with the per-ray terms counted by the CPU oracle (oracle/measure_flop_per_ray.py ->
# profiles/flop_per_ray.json; Cornell: 18 rectangles x 45 + scatter on the 68 % of rays that hit + camera = 872.3).
# This is the algorithm's work per ray as the reference does it; the fallback below prices the scatter on every ray.
FLOP_PER_RAY_CORNELL_UPPER = 18 * 45 + 62 + 15


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
    ap.add_argument("--dense-transforms", action="store_true", help="every rectangle through the full 3x4 transform, also those "
                    "turned about one axis (A/B measurements)")
    ap.add_argument("--shard", choices=("samples", "rows"), default="samples")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="samples mode with N > 1: strong = a fixed total sample count split over the ranks (default), "
                         "weak = the configuration's spp on every rank")
    ap.add_argument("--spp-total", type=int, default=0, help="strong scaling: samples per pixel of the whole job (default: the "
                                                             "configuration's strong_spp, else its spp)")
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


def cpu_baseline(config_name, cfg, width, height, spp, threads):
    """Times the CPU path on this host for the SAME scene, image size and bounce cap as the GPU job, on a bounded
    sample count (about 10 s of CPU work), best of three runs.  The reference's own code (oracle/_ref/ref_harness,
    built from its sources) renders its two named scenes and any scene file; its bounce cap is the literal 8 of
    main.cpp:275, so a configuration with another cap (glass: 32) is timed with this repo's AVX2 restatement of the
    same routine instead (kind "port"), as is everything when the harness was not prebuilt.  Threads: every online
    core, as the reference starts one worker per core (platform_linux.cpp:13-15, main.cpp:450-461); the 16-thread
    figure (the CPU share of a one-GPU slice of the box) is reported beside it when more cores are online.  The
    reference's spheres scene is also timed with the harness built at LANE_WIDTH 4 (the SSE path of simd.h:20-26),
    which is how the first benchmark configuration is worded."""
    if spp == 0:
        return None
    import tempfile
    from raytracer_amd import scene_io
    from raytracer_amd.host_scene import build_scene
    online = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    harness = ROOT / "oracle" / "_ref" / "ref_harness"
    harness_sse = ROOT / "oracle" / "_ref" / "ref_harness_lane4"
    use_reference = harness.exists() and cfg["bounces"] == 8
    scene = build_scene(*cfg["scene"])
    tmp = None
    if use_reference:
        if cfg["scene"][0] in ("cornell", "spheres"):
            scene_arg = cfg["scene"][0]                 # the reference's own builder (scene.h:193-469)
        else:
            tmp = tempfile.NamedTemporaryFile(suffix=".rtsc", delete=False)
            tmp.close()
            scene_io.write_scene(tmp.name, scene)
            scene_arg = tmp.name
    else:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as O
        oracle_scene = O.OracleScene(scene)

    def run(n_threads, n_spp, binary=harness):
        if use_reference:
            out = subprocess.run([str(binary), "render", scene_arg, str(width), str(height), str(n_spp), str(n_threads), "1", "-"],
                                 check=True, stdout=subprocess.PIPE, text=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
            return r["mrays_per_s"], r["ms"] / 1e3, r["rays"]
        p = O.default_params(width, height, n_spp, intersect_mode=O.WIDE8_AVX2, max_bounces=cfg["bounces"])
        t0 = time.perf_counter()
        _, _, rays = oracle_scene.render(p, threads=n_threads)
        dt = time.perf_counter() - t0
        return rays / dt / 1e6, dt, rays

    try:
        threads = threads or online
        runs = []
        if spp < 0:
            # size the sample for roughly 3 s of CPU work per run: a 1-spp probe gives the rate (and, where one
            # sample per pixel already takes that long, is the first of the runs)
            probe = run(threads, 1)
            spp = max(1, min(256, 1 << max(int(3.0 / max(probe[1], 1e-3)), 1).bit_length() - 1))
            if spp == 1:
                runs.append(probe)
        while len(runs) < (3 if not runs or runs[0][1] < 6.0 else 2):
            runs.append(run(threads, spp))
        value, seconds, rays = max(runs)
        result = {"value": round(value, 2), "unit": "Mrays/s", "cores": threads, "cores_online": online, "cpu_model": cpu_model(),
                  "kind": "reference" if use_reference else "port",
                  "sample": f"{cfg['label'].rsplit(' ', 1)[0].replace(', 32 bounces,', '')} {width}x{height} at {spp} spp, {cfg['bounces']} bounces, "
                            f"{rays} rays, one row per work order, AVX2 lanes, best of {len(runs)}",
                  "spp": spp, "seconds": round(seconds, 2), "runs_mrays_per_s": [round(r[0], 2) for r in runs]}
        if threads != 16 and online >= 16:
            v16, s16, _ = max(run(16, spp) for _ in range(2))
            result["value_16_threads"] = round(v16, 2)
            result["seconds_16_threads"] = round(s16, 2)
        if use_reference and harness_sse.exists() and config_name == "spheres":
            vs, ss, _ = max(run(threads, spp, harness_sse) for _ in range(3))
            result["value_sse"] = round(vs, 2)
            result["seconds_sse"] = round(ss, 2)
            result["sse_note"] = "the same harness built at LANE_WIDTH 4 (simd.h:20-26, simd_sse.h), best of 3"
        return result
    finally:
        if tmp is not None:
            os.unlink(tmp.name)


def plan_job(args, cfg, world_size, distributed):
    """Image size, samples per pixel of the whole job and of one rank, lanes per pixel, and the scaling mode."""
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    lanes = args.lanes_per_pixel or cfg["lanes"]
    rows_mode = distributed and args.shard == "rows"
    if not distributed or rows_mode:
        spp = args.spp or cfg["spp"]
        return dict(W=W, H=H, spp_rank=spp, spp_total=spp, lanes=lanes, rows_mode=rows_mode, scaling="strong" if rows_mode else "weak")
    if args.scaling == "weak" or world_size == 1:
        spp = args.spp or cfg["spp"]
        return dict(W=W, H=H, spp_rank=spp, spp_total=spp * world_size, lanes=lanes, rows_mode=False, scaling="weak")
    total = args.spp_total or args.spp or cfg.get("strong_spp", cfg["spp"])
    if total % world_size:
        raise SystemExit(f"bench.py: {total} samples per pixel do not split over {world_size} ranks")
    spp_rank = total // world_size
    # keep at least 8 samples per lane (the tail compaction's minimum) by narrowing the lanes per pixel
    while lanes > 1 and (spp_rank % lanes or spp_rank // lanes < 8):
        lanes //= 2
    return dict(W=W, H=H, spp_rank=spp_rank, spp_total=total, lanes=lanes, rows_mode=False, scaling="strong")


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
    job = plan_job(args, cfg, world_size, distributed)
    W, H, spp, spp_total, B_lanes, rows_mode = job["W"], job["H"], job["spp_rank"], job["spp_total"], job["lanes"], job["rows_mode"]
    ctx = B.Context(local_rank)
    info = ctx.device_info()
    ctx.upload_world(build_scene(*cfg["scene"]))

    render_flags = (0 if args.no_compact else B.FLAG_COMPACT_TAILS) | (B.FLAG_NO_PRIMARY_CULL if args.no_cull else 0) | \
                   (B.FLAG_DENSE_TRANSFORMS if args.dense_transforms else 0)
    base = B.default_params(W, H, spp, rng_mode=B.RNG_PIXEL, lanes_per_pixel=B_lanes, max_bounces=cfg["bounces"], flags=render_flags)
    total = args.steps + args.warmup
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
    combined = torch.zeros_like(radiance) if distributed and not rows_mode and rank == 0 else None   # rank 0: sum over the ranks

    def run_timed(overlap):
        """Warm-up and timed steps.  samples mode with `overlap`: the gather of step i travels (on RCCL's stream) while
        step i + 1 renders into the other buffer; rank 0 adds and resolves step i after step i + 1 has been enqueued,
        and the last step is drained inside the timed region."""
        rays = torch.zeros(total, dtype=torch.int64, device=device)
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(total)]
        stops = [torch.cuda.Event(enable_timing=True) for _ in range(total)]
        buffers = [radiance, torch.zeros_like(radiance)] if overlap else [radiance]
        in_flight = []      # at most one (handle) of a gather that has been started and not yet combined

        def combine_on_rank_0():
            if rank == 0:
                # fixed pairwise tree over the ranks, then color / sampleSize + pack (main.cpp:393)
                ctx.combine_device(gathered.parts.data_ptr(), world_size, H * W * 3, H * W * 3, combined.data_ptr(), stream.cuda_stream)
                ctx.resolve_device(combined.data_ptr(), W * H, spp_total, pixels.data_ptr(), None, stream.cuda_stream)

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
        kernel_ms = sum(starts[i].elapsed_time(stops[i]) for i in range(args.warmup, total)) / max(args.steps, 1)
        return elapsed, kernel_ms, rays[args.warmup:].sum()

    def check_composition():
        """Outside the timed region, N > 1: rank 0 renders the whole job alone (all the batches as streams of one
        render) and compares it with the image the ranks composed.  None where one render cannot hold the streams."""
        if rows_mode:
            whole = B.default_params(W, H, spp, rng_mode=B.RNG_PIXEL, lanes_per_pixel=B_lanes, max_bounces=cfg["bounces"])
        elif B_lanes * world_size <= 64:
            whole = B.default_params(W, H, spp_total, rng_mode=B.RNG_PIXEL, lanes_per_pixel=B_lanes * world_size, max_bounces=cfg["bounces"])
        else:
            return None
        check_pixels = torch.zeros((H, W), dtype=torch.int32, device=device)
        check_rays = torch.zeros(1, dtype=torch.int64, device=device)
        ctx.render_device(whole, check_pixels.data_ptr(), None, check_rays.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(device)
        return {"pixels_identical_to_one_gpu_render": bool(torch.equal(check_pixels, pixels)), "one_gpu_rays": int(check_rays.item())}

    overlap = distributed and not rows_mode and not args.no_overlap
    elapsed, kernel_ms, timed_rays = run_timed(overlap)
    composition = check_composition() if distributed and rank == 0 else None
    # A composed image that differs from the one-GPU render of the same batches is a wrong result.  With the
    # overlapped gather the first suspect is the overlap itself: every rank repeats the measurement with each
    # step's sums gathered and added before the next step starts, and the line says so.
    retried = False
    if distributed and overlap:
        bad = torch.tensor([1 if composition is not None and not composition["pixels_identical_to_one_gpu_render"] else 0],
                           dtype=torch.int64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.broadcast(bad, src=0)
        if int(bad.item()):
            retried = True
            elapsed, kernel_ms, timed_rays = run_timed(False)
            composition = check_composition() if rank == 0 else None

    # The same job on one GPU in its one-GPU form (the configuration's lanes per pixel, compacted), outside the timed
    # region: the denominator of the scaling efficiency.
    one_gpu_ms = None
    if distributed and rank == 0 and world_size > 1 and not args.rehearse_on_one_gpu:
        q = B.default_params(W, H, spp_total, rng_mode=B.RNG_PIXEL, lanes_per_pixel=args.lanes_per_pixel or cfg["lanes"],
                             max_bounces=cfg["bounces"], flags=render_flags)
        ctx.reserve(q)
        scratch_rays = torch.zeros(1, dtype=torch.int64, device=device)
        times = []
        for _ in range(3):
            t_start, t_stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_start.record(stream)
            ctx.render_device(q, pixels.data_ptr(), None, scratch_rays.data_ptr(), stream.cuda_stream)
            t_stop.record(stream)
            torch.cuda.synchronize(device)
            times.append(t_start.elapsed_time(t_stop))
        one_gpu_ms = min(times[1:])

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

    exit_code = 0
    if rank == 0:
        mrays = total_rays / elapsed / 1e6
        peak = info.fp32_vector_peak_tflops
        flop_terms = None
        flop_per_ray = float(FLOP_PER_RAY_CORNELL_UPPER) if args.config.startswith("cornell") else float("nan")
        profile = {}
        ffile = ROOT / "profiles" / "flop_per_ray.json"
        if ffile.exists():
            try:
                entry = json.loads(ffile.read_text()).get(args.config)
                if entry:
                    flop_terms, flop_per_ray = entry["terms"], float(entry["flop_per_ray"])
            except (ValueError, OSError, KeyError):
                pass
        # counters of this command collected in separate rocprofv3 / diagnostic-build runs (scripts/job_profiles_r03.sh)
        pfile = ROOT / "profiles" / "hbm_traffic.json"
        if pfile.exists():
            try:
                profile = json.loads(pfile.read_text())
            except (ValueError, OSError):
                profile = {}
        achieved = rank_rays_per_launch * flop_per_ray / (kernel_ms * 1e-3) / 1e12
        compacted = not args.no_compact and spp // B_lanes >= 8       # the library's rule (rt_api.hip)
        key = f"{args.config}_{W}x{H}_{spp}spp" + ("" if compacted else "_single_launch") + ("_no_cull" if args.no_cull else "") + \
              ("_dense_transforms" if args.dense_transforms else "")
        traffic = profile.get(key) if world_size == 1 and B_lanes == cfg["lanes"] else None
        pairs = None
        cfile = ROOT / "profiles" / "cull_stats.json"
        if cfile.exists() and not args.no_cull:
            try:
                pairs = json.loads(cfile.read_text())["rect_pairs_tested_per_ray"].get("cornell" if args.config.startswith("cornell") else args.config)
            except (ValueError, OSError, KeyError):
                pairs = None
        label = cfg["label"] if (W, H) == (cfg["width"], cfg["height"]) else f"{cfg['label'].rsplit(' ', 1)[0]} {W}x{H}"
        if rows_mode:
            workload = f"{label} {spp} spp, interleaved 8-row bands over {world_size} GPUs, RCCL all_gather"
        elif distributed:
            workload = (f"{label} {spp_total} spp in total: {spp} spp per GPU x {world_size} (sample batches), "
                        f"RCCL gather + fixed-order sum on rank 0")
        else:
            workload = f"{label} {spp} spp on one GPU"
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
            "scaling": job["scaling"],
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "rng_mode": "pixel", "lanes_per_pixel": B_lanes, "intersect_mode": "wide8",
                       "compact_tails": compacted, "primary_ray_cull": not args.no_cull,
                       "zero_terms_of_turned_rectangles_left_out": not args.dense_transforms, "max_bounces": cfg["bounces"],
                       "spp_total": spp_total, "spp_per_gpu": spp, "rays_per_step": total_rays // max(args.steps, 1),
                       "shard": args.shard if distributed else None, "rehearsal_on_one_gpu": bool(args.rehearse_on_one_gpu),
                       "gather_overlapped_with_next_step": bool(overlap and not retried) if distributed and not rows_mode else None,
                       "device": info.name.decode(), "compute_units": info.compute_units, "clock_mhz": info.clock_khz / 1e3},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": "profiles/hbm_traffic.json (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of one render with this launch sequence, "
                                           "collected in separate --pmc passes, not measured by this run)" if traffic else None,
                         "flop_per_ray": flop_per_ray, "flop_terms": flop_terms,
                         "rect_pairs_tested_per_ray": pairs, "kernel_ms": round(kernel_ms, 3),
                         "kernel": "rt::render_kernel<WIDE8, PIXEL> stage 1 (dominant) + stage 2 + rt::pixel_combine_kernel, timed together" if compacted else
                                   "rt::render_kernel<WIDE8, PIXEL>"},
        }
        if composition is not None:
            composition["rays_identical"] = composition.pop("one_gpu_rays") == total_rays // max(args.steps, 1)
            if retried:
                composition["first_attempt_with_overlapped_gather_differed"] = True
            line["composition_check"] = composition
            if not (composition["pixels_identical_to_one_gpu_render"] and composition["rays_identical"]):
                line["valid"] = False
                exit_code = 1
        if one_gpu_ms is not None:
            step_ms = elapsed / max(args.steps, 1) * 1e3
            line["one_gpu_same_job"] = {"ms": round(one_gpu_ms, 3), "mrays_per_s": round(total_rays / max(args.steps, 1) / one_gpu_ms / 1e3, 1),
                                        "form": f"{args.lanes_per_pixel or cfg['lanes']} lanes per pixel, " + ("compacted" if not args.no_compact else "single launch") + ", kernel time on rank 0's GPU, best of 2 after a warm-up"}
            line["efficiency_vs_one_gpu"] = round(one_gpu_ms / (world_size * step_ms), 4)
        if world_size == 1 and not distributed:
            line["cpu_baseline"] = cpu_baseline(args.config, cfg, W, H, args.cpu_spp, args.cpu_threads)
        print(json.dumps(line), flush=True)

    ctx.close()
    if distributed:
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
