"""rt_render_device_noalloc (the entry point that allocates nothing: its workspace is a fixed per-GPU
array inside the library) against rt_render_device, through the C ABI: pixels, linear radiance and ray
count bit for bit, on two calls in a row of which the second is smaller, then beyond the limit."""
import numpy as np
import pytest

from raytracer_amd import binding as B
from raytracer_amd.host_scene import build_scene

pytestmark = pytest.mark.gpu


def _render(ctx, p, entry):
    import torch
    dev = torch.device("cuda", 0)
    rows = B.local_row_count(p)
    pixels = torch.full((rows, p.width), 0x55555555, dtype=torch.int32, device=dev)
    radiance = torch.full((rows, p.width, 3), -1.0, dtype=torch.float32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    entry(p, pixels.data_ptr(), radiance.data_ptr(), rays.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    return pixels.cpu().numpy().view(np.uint32), radiance.cpu().numpy().view(np.uint32), int(rays.item())


def test_noalloc_render_matches_render_device_twice_in_a_row():
    import torch
    ctx = B.Context(0)
    fresh = B.Context(0)        # a context that only ever sees rt_upload_world and the no-allocation entry point
    try:
        ctx.upload_world(build_scene("cornell"))
        fresh.upload_world(build_scene("cornell"))
        # first a larger image, then a smaller one: the second call finds the first call's data in the shared array
        cases = [B.default_params(640, 360, 256, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS),
                 B.default_params(200, 120, 128, lanes_per_pixel=2, flags=B.FLAG_COMPACT_TAILS)]
        expected = []
        for p in cases:
            ctx.reserve(p)
            expected.append(_render(ctx, p, ctx.render_device))
        # The fresh context has no workspace of its own (no rt_reserve, no rt_render_device): a call that used the
        # context's workspace instead of the fixed array would have to allocate 15 MB here.  Device memory in
        # use must be the same before and after the calls (the outputs of _render come from torch's cache, warmed
        # by the renders above), and the calls must be legal under stream capture, where hipMalloc is refused.
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        for p, want in zip(cases, expected):
            got = _render(fresh, p, fresh.render_device_noalloc)
            assert got[2] == want[2] > 0
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert torch.cuda.mem_get_info(0)[0] == free_before, "rt_render_device_noalloc changed the device memory in use"
        dev = torch.device("cuda", 0)
        p, want = cases[1], expected[1]
        pixels = torch.zeros((p.height, p.width), dtype=torch.int32, device=dev)
        rays = torch.zeros(1, dtype=torch.int64, device=dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="global"):
            fresh.render_device_noalloc(p, pixels.data_ptr(), None, rays.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        graph.replay()
        torch.cuda.synchronize(dev)
        assert int(rays.item()) == want[2] and np.array_equal(pixels.cpu().numpy().view(np.uint32), want[0])
        del graph
        # spheres + plane, glass: the other kernel variants through the same workspace, and the pool limit
        # (more waves than 196,000 / 32: the hand-over threshold is lowered, results unchanged)
        ctx.upload_world(build_scene("glass_spheres", 3))
        p = B.default_params(1024, 1024, 64, lanes_per_pixel=1, max_bounces=32, flags=B.FLAG_COMPACT_TAILS)
        ctx.reserve(p)
        want = _render(ctx, p, ctx.render_device)
        got = _render(ctx, p, ctx.render_device_noalloc)
        assert got[2] == want[2] and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # without the flag there is no workspace and no limit
        p = B.default_params(1500, 1100, 4, lanes_per_pixel=1)
        want = _render(ctx, p, ctx.render_device)
        got = _render(ctx, p, ctx.render_device_noalloc)
        assert got[2] == want[2] and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # beyond the fixed workspace: refused with a clear error, nothing launched
        with pytest.raises(B.RtError) as e:
            _render(ctx, B.default_params(1920, 1080, 256, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS), ctx.render_device_noalloc)
        assert e.value.status == 5 and "fixed workspace" in str(e.value)
    finally:
        ctx.close()
