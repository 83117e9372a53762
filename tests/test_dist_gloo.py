"""The multi-rank composition (raytracer_amd/distributed.py) on two and on four CPU processes over
gloo, with the oracle standing in for the GPU renderer: sharded renders must reassemble into exactly
the single-process image.  Four ranks matter for the sample-sharded mode: with two, a + b is the same
in any order; with four, only the fixed pairwise tree ((0+1)+(2+3)) reproduces the single render of
all the batches, whereas a ring reduction (((0+1)+2)+3) would not."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from raytracer_amd import binding as B
from raytracer_amd import distributed as D
from raytracer_amd.scene_io import read_scene

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_params(p, **extra):
    return O.default_params(p.width, p.height, p.samples_per_pixel, row_begin=p.row_begin, row_end=p.row_end,
                            band_rows=p.band_rows, band_index=p.band_index, band_count=p.band_count,
                            rng_mode=p.rng_mode, rows_per_order=p.rows_per_order, lanes_per_pixel=p.lanes_per_pixel,
                            batch_offset=p.batch_offset, output_sum=1 if p.flags & B.FLAG_OUTPUT_SUM else 0, **extra)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        scene = O.OracleScene(read_scene(os.path.join(GOLDEN, "cornell.rtsc")))
        base = B.default_params(40, 27, 8, rng_mode=B.RNG_PIXEL, lanes_per_pixel=2)

        def render_rows(p, shard):
            pix, _, rays = scene.render(_oracle_params(p), threads=1)
            shard[:pix.shape[0]] = torch.from_numpy(pix.view(np.int32))
            return rays

        img, rays = D.render_rows(render_rows, base, rank, world_size, "cpu")
        total = D.sum_rays(torch.tensor([rays], dtype=torch.int64), world_size)
        full = img.assemble()

        def render_sum(p, radiance_sum):
            _, rad, rays = scene.render(_oracle_params(p), threads=1)
            radiance_sum.copy_(torch.from_numpy(rad))
            return rays

        combined = {}

        def resolve(radiance_sum, total_spp):
            combined["sum"] = radiance_sum.numpy().copy()
            mean = (radiance_sum / np.float32(total_spp)).numpy()
            return np.array([O.lib().rto_pack_srgb(np.ascontiguousarray(px).ctypes.data_as(C.c_void_p))
                             for px in mean.reshape(-1, 3)], dtype=np.uint32).reshape(mean.shape[:2])

        pixels, srays = D.render_samples(render_sum, resolve, base, rank, world_size, "cpu", B.FLAG_OUTPUT_SUM)
        stotal = D.sum_rays(torch.tensor([srays], dtype=torch.int64), world_size)
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), rows_image=full.numpy().view(np.uint32), rows_rays=total.numpy(),
                     sample_pixels=pixels, sample_rays=stotal.numpy(), sample_sum=combined["sum"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 4])
def test_ranks_reassemble_the_single_process_image(tmp_path, world_size):
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    out = np.load(tmp_path / "out.npz")
    scene = O.OracleScene(read_scene(os.path.join(GOLDEN, "cornell.rtsc")))

    # rows mode: identical to the unsharded render, pixel for pixel, ray for ray
    pix, _, rays = scene.render(O.default_params(40, 27, 8, rng_mode=O.PIXEL, lanes_per_pixel=2))
    assert np.array_equal(out["rows_image"], pix) and int(out["rows_rays"][0]) == rays

    # samples mode: N ranks x 8 spp (2 batches each) == one render of 8N spp with 2N batches, bit for bit
    pix_all, rad_all, rays_all = scene.render(O.default_params(40, 27, 8 * world_size, rng_mode=O.PIXEL,
                                                               lanes_per_pixel=2 * world_size, output_sum=1))
    assert int(out["sample_rays"][0]) == rays_all
    assert np.array_equal(out["sample_sum"].view(np.uint32), rad_all.view(np.uint32))
    assert np.array_equal(out["sample_pixels"], pix_all)


def test_tree_order_sum_is_the_kernel_tree_and_not_a_ring():
    """tree_order_sum adds ((0+1)+(2+3)); values chosen so that the ring order (((0+1)+2)+3) rounds differently."""
    parts = torch.tensor([[1.0], [2.0 ** -24], [2.0 ** -24], [2.0 ** -24]], dtype=torch.float32)
    tree = D.tree_order_sum(parts)
    ring = ((parts[0] + parts[1]) + parts[2]) + parts[3]
    expected = (parts[0] + parts[1]) + (parts[2] + parts[3])
    assert torch.equal(tree, expected) and not torch.equal(tree, ring)
    # counts that are not a power of two carry the odd block up: (0+1)+2, ((0+1)+(2+3))+4
    five = torch.arange(5, dtype=torch.float32).reshape(5, 1) * 0.1
    assert torch.equal(D.tree_order_sum(five[:3]), (five[0] + five[1]) + five[2])
    assert torch.equal(D.tree_order_sum(five), ((five[0] + five[1]) + (five[2] + five[3])) + five[4])


def test_band_helpers_match_the_c_abi():
    for rows, count in ((27, 3), (1080, 8), (9, 4)):
        for index in range(count):
            p = B.default_params(16, rows, 1, band_rows=8, band_index=index, band_count=count)
            assert D.local_rows(p) == B.local_row_count(p)
            assert D.global_rows(p) == list(B.global_rows(p))
