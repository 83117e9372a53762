"""Two and four processes, all rendering with the HIP library (here on the same GPU), composed by
raytracer_amd/distributed.py over gloo with host-side shards: everything of the multi-GPU path
except the RCCL transport itself, which needs more than one GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from raytracer_amd import binding as B
        from raytracer_amd import distributed as D
        from raytracer_amd.host_scene import build_scene
        ctx = B.Context(0)
        ctx.upload_world(build_scene("cornell"))
        base = B.default_params(120, 70, 16, lanes_per_pixel=4)

        def render_rows(p, shard):
            pix, _, rays = ctx.render(p, want_radiance=False)
            shard[:pix.shape[0]] = torch.from_numpy(pix.view(np.int32))
            return rays

        img, rays = D.render_rows(render_rows, base, rank, world_size, "cpu")
        total = D.sum_rays(torch.tensor([rays], dtype=torch.int64), world_size)
        full = img.assemble()

        def render_sum(p, radiance_sum):
            _, rad, rays = ctx.render(p, want_pixels=False)
            radiance_sum.copy_(torch.from_numpy(rad))
            return rays

        def resolve(radiance_sum, total_spp):
            dev = torch.device("cuda", 0)
            acc = radiance_sum.to(dev)
            pixels = torch.zeros(acc.shape[:2], dtype=torch.int32, device=dev)
            ctx.resolve_device(acc.data_ptr(), acc.shape[0] * acc.shape[1], total_spp, pixels.data_ptr(), None, None)
            torch.cuda.synchronize(dev)
            return pixels.cpu().numpy().view(np.uint32)

        pixels, srays = D.render_samples(render_sum, resolve, base, rank, world_size, "cpu", B.FLAG_OUTPUT_SUM)
        stotal = D.sum_rays(torch.tensor([srays], dtype=torch.int64), world_size)
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), rows_image=full.numpy().view(np.uint32), rows_rays=total.numpy(),
                     sample_pixels=pixels, sample_rays=stotal.numpy())
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 4])
def test_processes_compose_the_single_gpu_image(tmp_path, world_size):
    from raytracer_amd import binding as B
    from raytracer_amd.host_scene import build_scene
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    out = np.load(tmp_path / "out.npz")
    ctx = B.Context(0)
    ctx.upload_world(build_scene("cornell"))
    pix, _, rays = ctx.render(B.default_params(120, 70, 16, lanes_per_pixel=4))
    assert np.array_equal(out["rows_image"], pix) and int(out["rows_rays"][0]) == rays
    # N ranks x 16 spp with 4 batches each == one render of 16 N spp with 4 N batches (the ranks' sums are
    # added in the fixed pairwise tree, which continues the kernel's tree over a pixel's streams)
    pix_all, _, rays_all = ctx.render(B.default_params(120, 70, 16 * world_size, lanes_per_pixel=4 * world_size))
    assert int(out["sample_rays"][0]) == rays_all
    assert np.array_equal(out["sample_pixels"], pix_all)
    ctx.close()


@pytest.mark.parametrize("shard", ["samples", "rows"])
def test_bench_through_rccl_with_one_rank(shard):
    """bench.py's multi-rank path on the real transport: one rank under torch.distributed.run, process group on
    the GPU (backend "nccl" = RCCL), gather / all_gather of device tensors, fixed-order sum, resolve; the
    composed image and ray count equal a plain render of the job."""
    import json
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--through-process-group", "--shard", shard, "--cpu-spp", "0", "--width", "480", "--height", "270"]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, text=True, timeout=240).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["shard"] == shard
    assert line["composition_check"] == {"pixels_identical_to_one_gpu_render": True, "rays_identical": True}


@pytest.mark.parametrize("world_size,shard", [(2, "samples"), (4, "samples"), (8, "samples"), (4, "rows")])
def test_bench_rehearsal_composes_the_one_gpu_image(world_size, shard):
    """`python bench.py --gpus N` as the driver types it (self-launched ranks, fixed job split over them), rehearsed
    with all ranks on this one GPU and gloo in place of RCCL: the composed image and ray count equal one GPU's
    render of the whole job (N = 8: 128 samples per rank of 1024)."""
    import json
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", str(world_size), "--steps", "2", "--warmup", "1", "--rehearse-on-one-gpu",
           "--shard", shard, "--width", "320", "--height", "180"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, text=True, timeout=400, env=env).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world_size and line["scaling"] == "strong" and line["config"]["shard"] == shard
    if shard == "samples":
        assert line["config"]["spp_total"] == 1024 and line["config"]["spp_per_gpu"] * world_size == 1024
    assert line["composition_check"] == {"pixels_identical_to_one_gpu_render": True, "rays_identical": True}
