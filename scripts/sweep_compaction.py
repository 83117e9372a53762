"""Kernel time of the compacted Cornell bench render for the lanes-per-pixel values given as first argument
(comma separated); RT_POOL_THRESHOLD / RT_POOL_MAX_SAMPLES in the environment select the hand-over bounds.
Run on the GPU box with the experiment build of the library (make -C raytracer_amd/csrc knobs), e.g.
RT_HIP_LIBRARY=$PWD/build/librt_hip_knobs.so RT_POOL_MAX_SAMPLES=8 python scripts/sweep_compaction.py 2,4"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repository root
from raytracer_amd import binding as B
from raytracer_amd.host_scene import build_scene
import torch
ctx = B.Context(0)
ctx.upload_world(build_scene("cornell"))
dev = torch.device("cuda", 0)
pix = torch.zeros((1080, 1920), dtype=torch.int32, device=dev)
rays = torch.zeros(1, dtype=torch.int64, device=dev)
for lpp in [int(v) for v in sys.argv[1].split(",")]:
    p = B.default_params(1920, 1080, 256, lanes_per_pixel=lpp, flags=B.FLAG_COMPACT_TAILS | B.FLAG_TIME_KERNEL)
    ts = []
    for _ in range(5):
        ctx.render_device(p, pix.data_ptr(), None, rays.data_ptr(), None)
        ts.append(ctx.last_kernel_ms())
    print("S", os.environ.get("RT_POOL_MAX_SAMPLES", "8"), "T", os.environ.get("RT_POOL_THRESHOLD", "32"), "lanes", lpp, "%.3f ms" % min(ts[1:]), flush=True)
