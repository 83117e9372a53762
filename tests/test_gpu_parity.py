"""GPU parity: the HIP path, called through the C ABI, against (a) images and ray counts the
reference renderer itself produced (tests/golden) and (b) the CPU oracle on the same inputs.

Bars.  Linear radiance and ray counts: bit-exact.  Packed 8-bit pixels: at most 1 LSB on at most
1e-4 of the channels, because the device evaluates the sRGB pow in double precision while glibc's
powf (what the reference calls) is not correctly rounded.  Sphere normals in WIDE8 mode use the
correctly rounded 1/sqrt where the reference uses the x86 rsqrt approximation; there the oracle's
WIDE8_IEEE mode is the bit-exact twin and the reference image is matched statistically.
"""
import ctypes as C
import hashlib
import json
import pathlib

import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import binding as B
from raytracer_amd import scene_io
from raytracer_amd.host_scene import build_scene

pytestmark = pytest.mark.gpu

ORACLE_MODE = {B.INTERSECT_WIDE8: O.WIDE8_IEEE, B.INTERSECT_SCALAR: O.SCALAR, B.INTERSECT_WIDE4: O.WIDE4_IEEE}


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_params(p: B.RenderParams, mode=None):
    return O.default_params(
        p.width, p.height, p.samples_per_pixel, row_begin=p.row_begin, row_end=p.row_end,
        band_rows=p.band_rows, band_index=p.band_index, band_count=p.band_count,
        max_bounces=p.max_bounces, min_hit_distance=p.min_hit_distance, plane_tolerance=p.plane_tolerance,
        seed_constant=p.seed_constant, rng_mode=p.rng_mode, rows_per_order=p.rows_per_order,
        lanes_per_pixel=p.lanes_per_pixel, batch_offset=p.batch_offset,
        intersect_mode=ORACLE_MODE[p.intersect_mode] if mode is None else mode,
        output_sum=1 if p.flags & B.FLAG_OUTPUT_SUM else 0)


def channels(pix):
    return ((pix[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(int)


def assert_pixels_close(pix, ref):
    diff = np.abs(channels(pix) - channels(ref))
    assert diff.max() <= 1, f"8-bit difference {diff.max()}"
    assert (diff != 0).mean() <= 1e-4, f"{(diff != 0).sum()} channels differ by 1 LSB"
    assert np.array_equal(pix >> 24, ref >> 24)


def check_against_oracle(ctx, scene, p, threads=None, oracle_mode=None):
    ctx.upload_world(scene)
    pix, rad, rays = ctx.render(p)
    opix, orad, orays = O.OracleScene(scene).render(oracle_params(p, oracle_mode), threads=threads)
    assert rays == orays
    assert np.array_equal(bits(rad), bits(orad)), \
        f"radiance differs at {(bits(rad) != bits(orad)).sum()} values, max abs {np.nanmax(np.abs(rad - orad))}"
    if not p.flags & B.FLAG_OUTPUT_SUM:
        assert_pixels_close(pix, opix)
    return pix, rad, rays


def psnr8(a, b):
    mse = np.mean((channels(a) - channels(b)).astype(np.float64) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


# ---------------------------------------------------------------- against the reference's own output

def test_cornell_reproduces_the_reference_image(ctx, cornell, golden_dir):
    """ROW_SERIAL + WIDE8 on the Cornell box: same random streams, same arithmetic -> the
    reference's linear radiance and ray count, bit for bit (reference built with -ffp-contract=off)."""
    g = np.load(golden_dir / "image_cornell_64x36_16spp.npz")
    ctx.upload_world(cornell)
    pix, rad, rays = ctx.render(B.default_params(64, 36, 16, rng_mode=B.RNG_ROW_SERIAL))
    assert rays == int(g["rays"]) == 100810
    assert np.array_equal(bits(rad), bits(g["radiance"]))
    assert_pixels_close(pix, g["pixels"])


@pytest.mark.parametrize("name", ["cornell", "spheres", "random61", "glass3"])
def test_single_work_order_images(ctx, golden_dir, name):
    """One WorkOrder for the whole image (the reference's SINGLE_THREAD shape, main.cpp:418-430)."""
    g = np.load(golden_dir / f"image_{name}_48x20_8spp_oneorder.npz")
    scene = scene_io.read_scene(golden_dir / f"{name}.rtsc")
    ctx.upload_world(scene)
    p = B.default_params(48, 20, 8, rng_mode=B.RNG_ROW_SERIAL, rows_per_order=0)
    pix, rad, rays = ctx.render(p)
    if name == "cornell":
        assert rays == int(g["rays"]) and np.array_equal(bits(rad), bits(g["radiance"]))
    else:
        # spheres: the reference scales normals by the x86 rsqrt approximation (relative error
        # <= 3.7e-4); paths stay the same until that perturbation flips a decision, after which
        # the one serial stream of this order is consumed differently.  Statistical agreement:
        assert abs(rays - int(g["rays"])) <= 0.02 * int(g["rays"])
        assert abs(rad.mean() - g["radiance"].mean()) <= 0.15 * g["radiance"].mean() + 1e-3
    check_against_oracle(ctx, scene, p, threads=1)


def test_cornell_full_size_known_answer(ctx, cornell, golden_dir):
    """1280x720 at 512 spp, the size the reference ships: 1.29e9 rays through the ROW_SERIAL kernel
    must hash to the reference's radiance image (sha256 over the float32 bytes)."""
    k = json.loads((golden_dir / "known_answers.json").read_text())["cornell_1280x720_512spp_strict"]
    ctx.upload_world(cornell)
    pix, rad, rays = ctx.render(B.default_params(1280, 720, 512, rng_mode=B.RNG_ROW_SERIAL))
    assert rays == k["rays"] == 1293078337
    assert hashlib.sha256(rad.tobytes()).hexdigest() == k["radiance_sha256"]
    # 8-bit image: the reference's own tonemap (glibc powf) of that radiance is the reference's pixel image ...
    ref_pix = O.pack_srgb(rad)
    assert hashlib.sha256(ref_pix.tobytes()).hexdigest() == k["pixels_sha256"]
    # ... and the device's (pow evaluated in double, rounded once) differs from it by one LSB in a few channels
    diff = np.abs(channels(pix) - channels(ref_pix))
    print(f"\n1280x720x512 Cornell: {int((diff != 0).sum())} of {diff.size} 8-bit channels differ from the reference's, "
          f"max {int(diff.max())} LSB, PSNR {psnr8(pix, ref_pix):.2f} dB")
    assert diff.max() <= 1 and (diff != 0).mean() <= 1e-4


def test_cornell_full_size_known_answer_at_lane_width_four(ctx, cornell, golden_dir):
    """The same render as the reference BUILT AT LANE_WIDTH 4 produces it (simd.h:20-26; its closest-hit ties fall to
    other rectangles than at width 8: 24 rays fewer): RT_INTERSECT_WIDE4 must hash to that radiance image."""
    known = json.loads((golden_dir / "known_answers.json").read_text())
    k = known["cornell_1280x720_512spp_strict_lane4"]
    ctx.upload_world(cornell)
    pix, rad, rays = ctx.render(B.default_params(1280, 720, 512, rng_mode=B.RNG_ROW_SERIAL, intersect_mode=B.INTERSECT_WIDE4))
    assert rays == k["rays"] != known["cornell_1280x720_512spp_strict"]["rays"]
    assert hashlib.sha256(rad.tobytes()).hexdigest() == k["radiance_sha256"]
    assert hashlib.sha256(O.pack_srgb(rad).tobytes()).hexdigest() == k["pixels_sha256"]


def test_spheres_full_size_scalar_routine(ctx, spheres, golden_dir):
    """Spheres + plane at 1280x720x512 (the reference's spheres scene at the size it ships).  WIDE8 differs from the reference only in
    the sphere-normal rsqrt, so the image matches statistically; mean radiance within 0.5 %."""
    k = json.loads((golden_dir / "known_answers.json").read_text())["spheres_1280x720_512spp_strict"]
    ctx.upload_world(spheres)
    pix, rad, rays = ctx.render(B.default_params(1280, 720, 512, rng_mode=B.RNG_ROW_SERIAL))
    assert abs(rays - k["rays"]) <= 1e-3 * k["rays"]
    mean = rad.reshape(-1, 3).mean(axis=0, dtype=np.float64)
    assert np.allclose(mean, k["mean_radiance"], rtol=5e-3)


# ---------------------------------------------------------------- against the oracle, same inputs

@pytest.mark.parametrize("rng_mode", [B.RNG_ROW_SERIAL, B.RNG_PIXEL])
@pytest.mark.parametrize("imode", [B.INTERSECT_WIDE8, B.INTERSECT_SCALAR])
@pytest.mark.parametrize("name", ["cornell", "spheres", "random61", "glass3"])
def test_scenes_bit_exact(ctx, golden_dir, name, imode, rng_mode):
    scene = scene_io.read_scene(golden_dir / f"{name}.rtsc")
    check_against_oracle(ctx, scene, B.default_params(64, 36, 16, rng_mode=rng_mode, intersect_mode=imode))


@pytest.mark.parametrize("lanes", [1, 2, 8, 64])
def test_pixel_mode_lanes_per_pixel(ctx, cornell, lanes):
    check_against_oracle(ctx, cornell, B.default_params(50, 30, 64, rng_mode=B.RNG_PIXEL, lanes_per_pixel=lanes))


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (7, 3, 5), (65, 9, 2), (129, 17, 3)])
def test_ragged_sizes(ctx, cornell, w, h, spp):
    for rng_mode in (B.RNG_ROW_SERIAL, B.RNG_PIXEL):
        check_against_oracle(ctx, cornell, B.default_params(w, h, spp, rng_mode=rng_mode))


def test_row_ranges_and_bands(ctx, cornell):
    check_against_oracle(ctx, cornell, B.default_params(40, 30, 8, row_begin=7, row_end=19))
    check_against_oracle(ctx, cornell, B.default_params(40, 30, 8, rng_mode=B.RNG_ROW_SERIAL, row_begin=7, row_end=19,
                                                        rows_per_order=3))
    for index in range(3):
        check_against_oracle(ctx, cornell, B.default_params(40, 30, 8, band_rows=8, band_index=index, band_count=3))
        check_against_oracle(ctx, cornell, B.default_params(40, 30, 8, rng_mode=B.RNG_ROW_SERIAL, band_rows=4,
                                                            band_index=index, band_count=3, rows_per_order=2))


def test_bands_tile_the_full_render(ctx, cornell):
    ctx.upload_world(cornell)
    full_pix, full_rad, full_rays = ctx.render(B.default_params(64, 45, 8))
    rad = np.zeros_like(full_rad)
    rays = 0
    for index in range(4):
        p = B.default_params(64, 45, 8, band_rows=8, band_index=index, band_count=4)
        _, r, n = ctx.render(p)
        rad[B.global_rows(p)] = r
        rays += n
    assert rays == full_rays and np.array_equal(bits(rad), bits(full_rad))


def test_sample_batches_and_resolve(ctx, spheres):
    """Two passes with distinct batch offsets, summed, equal the one render with all the batches;
    rt_resolve_device tonemaps the sum like the fused epilogue."""
    import torch
    ctx.upload_world(spheres)
    whole_pix, whole, rays = ctx.render(B.default_params(48, 32, 32, lanes_per_pixel=8))
    dev = torch.device("cuda", 0)
    acc = torch.zeros((32, 48, 3), dtype=torch.float32, device=dev)
    part = torch.zeros_like(acc)
    nrays = torch.zeros(2, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    for i, off in enumerate((0, 4)):
        p = B.default_params(48, 32, 16, lanes_per_pixel=4, batch_offset=off, flags=B.FLAG_OUTPUT_SUM)
        ctx.render_device(p, None, part.data_ptr(), nrays.data_ptr() + 8 * i, stream.cuda_stream)
        acc += part
    pixels = torch.zeros((32, 48), dtype=torch.int32, device=dev)
    mean = torch.zeros_like(acc)
    ctx.resolve_device(acc.data_ptr(), 32 * 48, 32, pixels.data_ptr(), mean.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    assert int(nrays.sum()) == rays
    assert np.array_equal(bits(mean.cpu().numpy()), bits(whole))
    assert np.array_equal(pixels.cpu().numpy().view(np.uint32), whole_pix)


@pytest.mark.parametrize("bounces", [1, 3, 32])
def test_bounce_cap(ctx, golden_dir, bounces):
    scene = scene_io.read_scene(golden_dir / "glass3.rtsc")
    check_against_oracle(ctx, scene, B.default_params(48, 27, 16, max_bounces=bounces))
    check_against_oracle(ctx, scene, B.default_params(48, 27, 4, max_bounces=bounces, rng_mode=B.RNG_ROW_SERIAL))


def test_seed_constant_and_tolerances(ctx, spheres):
    a = check_against_oracle(ctx, spheres, B.default_params(32, 18, 8, seed_constant=777767))
    b = check_against_oracle(ctx, spheres, B.default_params(32, 18, 8))
    assert not np.array_equal(a[1], b[1])
    check_against_oracle(ctx, spheres, B.default_params(32, 18, 8, min_hit_distance=0.01, plane_tolerance=0.05))


def test_mixed_scene_spheres_and_rectangles(ctx, cornell, spheres):
    """Spheres inside the Cornell box: exercises the kernel variant that orders sphere/rectangle ties."""
    sp = spheres.spheres.copy()
    sp["position"] = sp["position"] * np.float32(0.8) + np.array([0, -7, -7], dtype=np.float32)
    sp["material_index"] = np.array([1, 2, 3, 1, 2, 3, 4, 1], dtype=np.uint32)
    scene = scene_io.Scene(cornell.materials, np.zeros(0, dtype=scene_io.PLANE), sp, cornell.rectangles, cornell.camera)
    for imode in (B.INTERSECT_WIDE8, B.INTERSECT_SCALAR):
        for rng_mode in (B.RNG_ROW_SERIAL, B.RNG_PIXEL):
            check_against_oracle(ctx, scene, B.default_params(64, 36, 16, intersect_mode=imode, rng_mode=rng_mode))


def test_odd_rectangle_count_is_padded(ctx, cornell):
    scene = scene_io.Scene(cornell.materials, cornell.planes, cornell.spheres, cornell.rectangles[:17], cornell.camera)
    check_against_oracle(ctx, scene, B.default_params(64, 36, 8))


def test_stress_scenes_reduced_size(ctx):
    """The LDS-stress (4096 spheres) and divergence-stress (glass, 32 bounces) scenes at a size the oracle renders in seconds."""
    check_against_oracle(ctx, build_scene("random_spheres", 4096, 1), B.default_params(64, 36, 4))
    check_against_oracle(ctx, build_scene("glass_spheres", 5), B.default_params(64, 36, 16, max_bounces=32))
    # spheres live in global memory, so their number is not bounded by the LDS: 20000 spheres (400 KB), and
    # counts that leave 1..3 spheres after the groups of four, and counts around the 64-record staging chunks
    check_against_oracle(ctx, build_scene("random_spheres", 20000, 1), B.default_params(32, 18, 2))
    for count in (1, 2, 3, 5, 6, 7, 63, 64, 65, 67, 129):
        check_against_oracle(ctx, build_scene("random_spheres", count, 3), B.default_params(48, 27, 4))


def test_bench_size_cornell_bit_exact(ctx, cornell):
    """The benchmarked workload itself - Cornell 1920x1080 at 256 spp, PIXEL mode, 1.45e9 rays - is
    bit-identical to the CPU oracle."""
    check_against_oracle(ctx, cornell, B.default_params(1920, 1080, 256, lanes_per_pixel=2), oracle_mode=O.WIDE8_AVX2)


def test_other_configurations_at_full_resolution(ctx, cornell):
    """The remaining benchmark configurations at their full image size and bounce cap, with the sample count
    cut to what the CPU oracle renders in seconds: 4K Cornell, the reference's spheres scene at its shipped
    720p, the glass lattice with 32 bounces at 1080p, and the 4096-sphere scene at half of 1080p."""
    check_against_oracle(ctx, cornell, B.default_params(3840, 2160, 32, lanes_per_pixel=2), oracle_mode=O.WIDE8_AVX2)
    check_against_oracle(ctx, build_scene("spheres"), B.default_params(1280, 720, 128, lanes_per_pixel=2))
    check_against_oracle(ctx, build_scene("glass_spheres", 5), B.default_params(1920, 1080, 32, lanes_per_pixel=2, max_bounces=32))
    check_against_oracle(ctx, build_scene("random_spheres", 4096, 1), B.default_params(960, 540, 4, lanes_per_pixel=2))


def _bench_config(name):
    import bench
    cfg = bench.CONFIGS[name]
    return cfg, build_scene(*cfg["scene"])


@pytest.mark.parametrize("name,size,spp", [("cornell1024", None, 64), ("cornell4k", None, 32), ("spheres", None, 512),
                                           ("glass", None, 128), ("spheres4096", (480, 270), 32)])
def test_benchmarked_launch_sequences_against_the_oracle(ctx, name, size, spp):
    """Every bench.py configuration in the form bench.py launches it - its scene, image size, bounce cap, lanes per
    pixel and RT_FLAG_COMPACT_TAILS (stage 1, stage 2, per-pixel combine) - against the CPU oracle, with the sample
    count cut to what the oracle renders in seconds (but at least 8 samples per lane, so that the compaction runs);
    the 4096-sphere scene also at a smaller image; the reference's spheres scene runs at its full 512 spp.  (The headline configuration at its full sample count:
    test_compacted_bench_workload_bit_exact.)"""
    cfg, scene = _bench_config(name)
    w, h = size or (cfg["width"], cfg["height"])
    assert spp // cfg["lanes"] >= 8
    check_against_oracle(ctx, scene, B.default_params(w, h, spp, lanes_per_pixel=cfg["lanes"], max_bounces=cfg["bounces"],
                                                      flags=B.FLAG_COMPACT_TAILS))


@pytest.mark.parametrize("name", ["cornell", "spheres", "spheres4096", "glass", "cornell1024", "cornell4k"])
def test_benchmarked_workloads_at_full_size(ctx, name):
    """The bench.py configurations at their FULL size and sample count: the compacted launch sequence that bench.py
    times gives the pixels and ray count of the single launch, bit for bit, and the ray count is the CPU oracle's for
    the whole job.  (The oracle needs up to ten minutes for one of these renders, so its full-size runs are a script,
    scripts/verify_full_configs.py, whose record - radiance bit for bit, ray counts - is profiles/r03_full_size_parity.jsonl.)"""
    import torch
    cfg, scene = _bench_config(name)
    recorded = {}
    for line in (pathlib.Path(__file__).resolve().parent.parent / "profiles" / "r03_full_size_parity.jsonl").read_text().splitlines():
        d = json.loads(line)
        assert d["rays_gpu"] == d["rays_oracle"] and d["radiance_values_differing"] == 0 and d["rows"] == [0, d["height"]]
        its = _bench_config(d["config"])[0]
        assert (d["width"], d["height"], d["spp"], d["lanes_per_pixel"], d["max_bounces"]) == \
            (its["width"], its["height"], its["spp"], its["lanes"], its["bounces"])
        recorded[d["config"]] = d["rays_oracle"]
    ctx.upload_world(scene)
    dev = torch.device("cuda", 0)
    w, h = cfg["width"], cfg["height"]
    out = []
    for flags in (B.FLAG_COMPACT_TAILS, 0):
        p = B.default_params(w, h, cfg["spp"], lanes_per_pixel=cfg["lanes"], max_bounces=cfg["bounces"], flags=flags)
        ctx.reserve(p)
        pixels = torch.zeros((h, w), dtype=torch.int32, device=dev)
        rays = torch.zeros(1, dtype=torch.int64, device=dev)
        ctx.render_device(p, pixels.data_ptr(), None, rays.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        out.append((pixels, int(rays.item())))
    assert out[0][1] == out[1][1], "ray count of the compacted and the single-launch render"
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == recorded[name]
    assert out[0][1] > w * h * cfg["spp"]


def test_image_statistics_match_the_reference(ctx, cornell, golden_dir):
    """PIXEL-mode streams differ from the reference's row streams, so agreement is statistical: image
    mean within 1 % of the reference's 320x180x64 render (two reference seeds differ by more)."""
    k = json.loads((golden_dir / "known_answers.json").read_text())["cornell_320x180_64spp_strict"]
    ctx.upload_world(cornell)
    _, rad, rays = ctx.render(B.default_params(320, 180, 64))
    mean = rad.reshape(-1, 3).mean(axis=0, dtype=np.float64)
    assert np.allclose(mean, k["mean_radiance"], rtol=1e-2)
    assert abs(rays - k["rays"]) <= 5e-3 * k["rays"]


def test_runs_are_deterministic(ctx, cornell):
    ctx.upload_world(cornell)
    a = ctx.render(B.default_params(200, 120, 32))
    b = ctx.render(B.default_params(200, 120, 32))
    assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))


# ---------------------------------------------------------------- error behaviour

def test_errors(ctx, cornell):
    import copy
    fresh = B.Context(0)
    with pytest.raises(B.RtError) as e:
        fresh.render(B.default_params(8, 8, 1))
    assert e.value.status == 4                                  # RT_ERR_NO_WORLD
    with pytest.raises(B.RtError) as e:
        fresh.reserve(B.default_params(8, 8, 256, flags=B.FLAG_COMPACT_TAILS))
    assert e.value.status == 4
    with pytest.raises(B.RtError) as e:
        fresh.resolve(np.zeros((4, 4, 3), dtype=np.float32), 0)
    assert e.value.status == 1 and "sample count" in str(e.value)
    fresh.close()
    ctx.upload_world(cornell)
    bad = [dict(samples_per_pixel=0), dict(width=0), dict(row_begin=5, row_end=5), dict(row_end=99),
           dict(lanes_per_pixel=3), dict(lanes_per_pixel=128), dict(samples_per_pixel=6, lanes_per_pixel=4),
           dict(max_bounces=0), dict(rng_mode=7), dict(intersect_mode=9), dict(struct_size=12),
           dict(band_rows=8, band_index=3, band_count=3)]
    for fields in bad:
        p = B.default_params(16, 16, 4)
        for k, v in fields.items():
            setattr(p, k, v)
        with pytest.raises(B.RtError) as e:
            ctx.render(p)
        assert e.value.status == 1, fields                      # RT_ERR_INVALID_ARGUMENT
    big = copy.deepcopy(cornell)                                # 3600 rectangles = 230 KB: more than one workgroup's LDS
    big.rectangles = np.tile(cornell.rectangles, 200)
    with pytest.raises(B.RtError) as e:
        ctx.upload_world(big)
    assert e.value.status == 5                                  # RT_ERR_TOO_LARGE
    # non-finite or huge scene values are refused at upload (the kernel's arithmetic forms assume bounded input)
    for field, index, value in (("rectangles", ("inverse_transform", 3, (1, 2)), np.nan), ("rectangles", ("normal", 0, 1), np.inf),
                                ("materials", ("emit_color", 2, 0), -np.inf), ("camera", ("position", 0, 2), 2e18)):
        broken = copy.deepcopy(cornell)
        name, row, col = index
        getattr(broken, field)[name][row][col] = value
        with pytest.raises(B.RtError) as e:
            ctx.upload_world(broken)
        assert e.value.status == 1 and "non-finite or huge" in str(e.value), field
    ctx.upload_world(cornell)                                   # the context is still usable
    ctx.render(B.default_params(16, 16, 4))


def test_kernel_timer(ctx, cornell):
    ctx.upload_world(cornell)
    import torch
    dev = torch.device("cuda", 0)
    pix = torch.zeros((90, 160), dtype=torch.int32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    ctx.render_device(B.default_params(160, 90, 64, flags=B.FLAG_TIME_KERNEL), pix.data_ptr(), None, rays.data_ptr(), None)
    ms = ctx.last_kernel_ms()
    assert 0 < ms < 1000 and int(rays.item()) > 160 * 90 * 64


@pytest.mark.parametrize("lanes", [1, 2, 4, 16])
def test_tail_compaction_changes_nothing_but_the_schedule(ctx, cornell, spheres, lanes):
    """RT_FLAG_COMPACT_TAILS (stage 1 + pool + stage 2 + per-pixel combine) gives the pixels, radiance and ray
    count of the single-launch render, bit for bit: rectangles, spheres + plane, glass with 32 bounces, ragged
    sizes, row bands, sum output."""
    cases = [(cornell, B.default_params(333, 77, 64 * lanes, lanes_per_pixel=lanes)),
             (spheres, B.default_params(160, 90, 48 * lanes, lanes_per_pixel=lanes, intersect_mode=B.INTERSECT_SCALAR)),
             (build_scene("glass_spheres", 3), B.default_params(96, 54, 64 * lanes, lanes_per_pixel=lanes, max_bounces=32)),
             (cornell, B.default_params(200, 120, 40 * lanes, lanes_per_pixel=lanes, band_rows=8, band_index=1, band_count=3,
                                        flags=B.FLAG_OUTPUT_SUM, batch_offset=5)),
             (cornell, B.default_params(64, 36, 4 * lanes, lanes_per_pixel=lanes))]     # too few samples: falls back to one launch
    for scene, p in cases:
        ctx.upload_world(scene)
        plain = ctx.render(p)
        p.flags |= B.FLAG_COMPACT_TAILS
        ctx.reserve(p)
        compact = ctx.render(p)
        assert compact[2] == plain[2]
        assert np.array_equal(compact[0], plain[0]) and np.array_equal(bits(compact[1]), bits(plain[1]))
    # ROW_SERIAL ignores the flag
    ctx.upload_world(cornell)
    p = B.default_params(64, 36, 16, rng_mode=B.RNG_ROW_SERIAL)
    plain = ctx.render(p)
    p.flags = B.FLAG_COMPACT_TAILS
    assert np.array_equal(ctx.render(p)[0], plain[0])


def test_tail_compaction_over_random_parameters(ctx, cornell, spheres):
    """120 random combinations of scene, image size (from 1x1), lanes per pixel (1..64), bounce cap, intersect
    routine, batch offset, row bands and sum output: compacted and single-launch renders are identical."""
    rng = np.random.default_rng(7)
    scenes = [cornell, spheres, build_scene("glass_spheres", 3), build_scene("random_spheres", 37, 2)]
    for it in range(120):
        ctx.upload_world(scenes[it % 4])
        lanes = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        kw = dict(lanes_per_pixel=lanes, max_bounces=int(rng.choice([1, 4, 8, 32])), intersect_mode=int(rng.integers(0, 3)),
                  batch_offset=int(rng.integers(0, 1000)))
        if rng.random() < 0.3 and h > 8:
            kw.update(band_rows=int(rng.choice([1, 4, 8])), band_count=int(rng.integers(2, 5)))
            kw["band_index"] = int(rng.integers(0, kw["band_count"]))
        if rng.random() < 0.3:
            kw["flags"] = B.FLAG_OUTPUT_SUM
        p = B.default_params(w, h, lanes * int(rng.integers(8, 90)), **kw)     # compaction is on from 8 samples per lane
        if B.local_row_count(p) == 0:
            continue
        plain = ctx.render(p)
        p.flags |= B.FLAG_COMPACT_TAILS
        compact = ctx.render(p)
        assert compact[2] == plain[2] and np.array_equal(compact[0], plain[0]) and \
            np.array_equal(bits(compact[1]), bits(plain[1])), (it, w, h, kw)


def test_compacted_bench_workload_bit_exact(ctx, cornell):
    """The benchmarked launch sequence (Cornell 1920x1080 at 256 spp, 4 lanes per pixel, compacted tails)
    against the CPU oracle."""
    check_against_oracle(ctx, cornell, B.default_params(1920, 1080, 256, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS),
                         oracle_mode=O.WIDE8_AVX2)


def test_render_device_can_be_recorded_into_a_graph(ctx, cornell):
    """rt_render_device allocates nothing (after rt_reserve) and sets no function attributes at launch, so
    it can be captured into a HIP graph - here the four kernels of a compacted render; replays give the
    pixels and ray count of a direct call."""
    import torch
    ctx.upload_world(cornell)
    dev = torch.device("cuda", 0)
    p = B.default_params(160, 90, 128, lanes_per_pixel=2, flags=B.FLAG_COMPACT_TAILS)
    ctx.reserve(p)                      # the workspace is allocated here, not inside the capture
    direct = torch.zeros((90, 160), dtype=torch.int32, device=dev)
    replayed = torch.zeros_like(direct)
    rays = torch.zeros(2, dtype=torch.int64, device=dev)
    ctx.render_device(p, direct.data_ptr(), None, rays.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ctx.render_device(p, replayed.data_ptr(), None, rays.data_ptr() + 8, torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        replayed.zero_()
        graph.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(replayed, direct) and int(rays[1]) == int(rays[0]) > 0


def test_packed_division_is_ieee_division():
    """rt::div2 (reciprocal + packed refinement, no operand scaling) against the compiler's IEEE
    division on 2.1e9 random quotients: mantissas uniform, exponents spanning 2^-40 .. 2^40 (the
    rectangle test's operands are distances and direction components of order 1e-7 .. 1e3)."""
    import pathlib
    lib = C.CDLL(str(pathlib.Path(B.LIB_PATH).with_name("librt_selftest.so")))
    lib.rt_selftest_div2.restype = C.c_int
    lib.rt_selftest_div2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
    bad, tested = C.c_uint64(0), C.c_uint64(0)
    first = np.zeros(4, dtype=np.float32)
    rc = lib.rt_selftest_div2(12345, 4096, 1024, 127 - 40, 127 + 40, C.byref(bad), C.byref(tested),
                              first.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert tested.value == 4096 * 256 * 1024 * 2
    assert bad.value == 0, f"{bad.value} of {tested.value} quotients differ, e.g. a, b, div2, ieee = {first}"


def test_inverse_length_equals_the_library_operations_on_every_float():
    """rt::inv_sqrt_exact (hardware sqrt moved to the correctly rounded neighbour + unscaled reciprocal
    refinement) against 1.0f / sqrtf(x) for all 2^32 bit patterns: the fast range, and the zero, denormal,
    huge, negative, inf and NaN arguments that take the library forms."""
    import pathlib
    lib = C.CDLL(str(pathlib.Path(B.LIB_PATH).with_name("librt_selftest.so")))
    lib.rt_selftest_inv_sqrt.restype = C.c_int
    lib.rt_selftest_inv_sqrt.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    bad = C.c_uint64(0)
    first = np.zeros(3, dtype=np.float32)
    rc = lib.rt_selftest_inv_sqrt(0, 1 << 32, C.byref(bad), first.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert bad.value == 0, f"{bad.value} arguments differ, e.g. x, inv_sqrt_exact, library = {first}"


def test_render_multi_assembles_the_full_image(ctx, cornell):
    """rt_render_multi with two contexts (here both on GPU 0, each with its own host thread) and with
    the same context listed three times: interleaved 8-row bands reassemble into the single render."""
    ctx.upload_world(cornell)
    p = B.default_params(96, 50, 8, row_begin=3, row_end=47)
    pix, rad, rays = ctx.render(p)
    other = B.Context(0)
    other.upload_world(cornell)
    for group in ([ctx, other], [ctx, ctx, ctx], [other]):
        mpix, mrad, mrays = B.render_multi(group, p)
        assert mrays == rays
        assert np.array_equal(mpix, pix) and np.array_equal(bits(mrad), bits(rad))
    # more contexts than bands: the surplus ones own no rows
    small = B.default_params(32, 10, 4)
    spix, _, srays = ctx.render(small)
    mpix, _, mrays = B.render_multi([ctx, other, ctx], small)
    assert mrays == srays and np.array_equal(mpix, spix)
    other.close()


def _random_scene(rng: np.random.Generator):
    """A random world: 0-6 planes, 0-40 spheres, 0-21 rectangles (random affine frames), 2-9 materials
    mixing diffuse, glossy, mirror, glass and emissive, and a camera looking at the origin."""
    nm = int(rng.integers(2, 10))
    mats = np.zeros(nm, dtype=scene_io.MATERIAL)
    mats["color"] = rng.uniform(0.05, 0.95, (nm, 3))
    mats["reflection"] = rng.choice([0.0, 0.0, 0.3, 0.9, 1.0], nm)
    mats["refractive_index"] = rng.choice([0.0, 0.0, 0.0, 1.5, 1.33, 2.4], nm)
    emissive = rng.random(nm) < 0.3
    mats["emit_color"][emissive] = rng.uniform(0.5, 12.0, (int(emissive.sum()), 3))
    mats["emit_color"][0] = rng.uniform(0.0, 0.4, 3)          # sky

    planes = np.zeros(int(rng.integers(0, 7)), dtype=scene_io.PLANE)
    if len(planes):
        n = rng.normal(size=(len(planes), 3)).astype(np.float32)
        planes["normal"] = n / np.linalg.norm(n, axis=1, keepdims=True)
        planes["d"] = rng.uniform(1.0, 9.0, len(planes))
        planes["material_index"] = rng.integers(1, nm, len(planes))

    spheres = np.zeros(int(rng.integers(0, 41)), dtype=scene_io.SPHERE)
    if len(spheres):
        spheres["position"] = rng.uniform(-6, 6, (len(spheres), 3))
        spheres["radius"] = rng.uniform(0.2, 1.8, len(spheres))
        spheres["material_index"] = rng.integers(1, nm, len(spheres))

    rects = np.zeros(int(rng.integers(0, 22)), dtype=scene_io.RECTANGLE)
    for r in rects:
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m = np.eye(4)
        m[:3, :3] = q * rng.uniform(0.5, 5.0, 3)              # rotation x scale
        m[:3, 3] = rng.uniform(-7, 7, 3)
        r["inverse_transform"] = np.linalg.inv(m).astype(np.float32)
        r["normal"] = q[:, 2].astype(np.float32)
        r["material_index"] = rng.integers(1, nm)

    pos = rng.uniform(-1, 1, 3) * 4 + np.array([0, 2, 14])
    z = pos / np.linalg.norm(pos)
    xv = np.cross([0, 1, 0], z); xv /= np.linalg.norm(xv)
    yv = np.cross(z, xv); yv /= np.linalg.norm(yv)
    cam = np.zeros(1, dtype=scene_io.CAMERA)
    cam["position"], cam["z_axis"], cam["y_axis"], cam["x_axis"] = pos, z, yv, xv
    return scene_io.Scene(mats, planes, spheres, rects, cam)


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_bit_exact(ctx, seed):
    """Randomly generated worlds (every primitive kind, tails of every length, glass inside glass,
    emitters, mirrors) in every intersect / stream mode: radiance and ray count bit-exact vs the oracle."""
    rng = np.random.default_rng(1000 + seed)
    scene = _random_scene(rng)
    w, h, spp = int(rng.integers(17, 70)), int(rng.integers(9, 40)), int(rng.choice([4, 8, 16]))
    bounces = int(rng.choice([2, 8, 8, 20]))
    for imode in (B.INTERSECT_WIDE8, B.INTERSECT_SCALAR, B.INTERSECT_WIDE4):
        check_against_oracle(ctx, scene, B.default_params(w, h, spp, intersect_mode=imode, max_bounces=bounces,
                                                          lanes_per_pixel=int(rng.choice([1, 2, 4]))))
        check_against_oracle(ctx, scene, B.default_params(w, h, spp, intersect_mode=imode, max_bounces=bounces,
                                                          rng_mode=B.RNG_ROW_SERIAL, rows_per_order=int(rng.choice([1, 3]))))
    # the compacted launch sequence, at a sample count that makes waves hand streams over
    lanes = int(rng.choice([1, 2, 4]))
    check_against_oracle(ctx, scene, B.default_params(w, h, 64 * lanes, max_bounces=bounces, lanes_per_pixel=lanes,
                                                      flags=B.FLAG_COMPACT_TAILS))


def tie_scene():
    """Five coplanar unit rectangles on top of each other (z = 0, facing the camera; a sixth, index 0, behind them),
    each with its own emissive material: every primary ray meets the five at exactly the same t, so the winner is
    decided by the order of the routine alone - index order in the scalar routine, (lane, group) order at lane
    width 8 or 4."""
    mats = np.zeros(8, dtype=scene_io.MATERIAL)
    mats["emit_color"][0] = (0.1, 0.1, 0.1)
    for i in range(1, 8):
        mats["emit_color"][i] = (i & 1, (i >> 1) & 1, (i >> 2) & 1)
    rects = np.zeros(6, dtype=scene_io.RECTANGLE)
    for i, r in enumerate(rects):
        r["inverse_transform"] = np.eye(4, dtype=np.float32)
        r["normal"] = (0, 0, 1)
        r["material_index"] = 1 + i
    rects["inverse_transform"][0][2, 3] = 0.5      # index 0 lies half a unit behind the others: it never wins
    # a seventh one in front of the right half only, so that lanes also have to yield to a strictly closer hit
    front = np.zeros(1, dtype=scene_io.RECTANGLE)
    m = np.eye(4); m[0, 0] = 0.5; m[0, 3] = 0.5; m[2, 3] = 0.25
    front["inverse_transform"][0] = np.linalg.inv(m).astype(np.float32)
    front["normal"][0] = (0, 0, 1)
    front["material_index"][0] = 7
    rects = np.concatenate([rects[:5], front, rects[5:]])
    cam = np.zeros(1, dtype=scene_io.CAMERA)
    cam["position"], cam["z_axis"], cam["y_axis"], cam["x_axis"] = (0, 0, 3), (0, 0, 1), (0, 1, 0), (1, 0, 0)
    return scene_io.Scene(mats, np.zeros(0, dtype=scene_io.PLANE), np.zeros(0, dtype=scene_io.SPHERE), rects, cam)


def test_ties_between_primitives_follow_the_lane_order(ctx):
    """Exact ties in t: each routine's winner is its oracle's (scalar: lowest index; wide: lowest lane, then lowest
    group, at width 8 and at width 4), and the three orders really give three different images here."""
    scene = tie_scene()
    images = {}
    for imode in (B.INTERSECT_SCALAR, B.INTERSECT_WIDE8, B.INTERSECT_WIDE4):
        _, rad, _ = check_against_oracle(ctx, scene, B.default_params(64, 36, 4, intersect_mode=imode, max_bounces=2))
        images[imode] = rad
    # 7 rectangles: width 8 keeps every primitive in its own lane (= index order, like the scalar routine: index 1
    # wins); at width 4 index 4 shares lane 0 with the farther index 0, replaces it there, and lane 0 goes first
    assert np.array_equal(images[B.INTERSECT_SCALAR], images[B.INTERSECT_WIDE8])
    assert not np.array_equal(images[B.INTERSECT_WIDE4], images[B.INTERSECT_WIDE8])


@pytest.mark.parametrize("width,imode", [(8, B.INTERSECT_WIDE8), (4, B.INTERSECT_WIDE4)])
def test_tie_scene_against_the_reference_at_both_lane_widths(ctx, golden_dir, width, imode):
    """The tie scene as the reference itself renders it, built at LANE_WIDTH 8 and at LANE_WIDTH 4 (two different
    images): the GPU's row-stream render in the matching mode is that image bit for bit; and the 2048 tie-scene
    rays' winners are those of the reference's 4-wide routine (through a one-bounce render of each... the oracle
    holds the per-ray vectors, tests/test_oracle_golden.py)."""
    g = np.load(golden_dir / f"image_tie_64x36_4spp_lane{width}.npz")
    scene = scene_io.read_scene(golden_dir / "tie.rtsc")
    assert np.array_equal(scene.rectangles.tobytes(), tie_scene().rectangles.tobytes())
    ctx.upload_world(scene)
    pix, rad, rays = ctx.render(B.default_params(64, 36, 4, rng_mode=B.RNG_ROW_SERIAL, intersect_mode=imode))
    assert rays == int(g["rays"]) and np.array_equal(bits(rad), bits(g["radiance"]))
    assert_pixels_close(pix, g["pixels"])


# ---------------------------------------------------------------- sums, passes and their composition, against the oracle

def test_sum_output_batches_and_bands_against_the_oracle(ctx, cornell, spheres):
    """The un-normalised sample sum (RT_FLAG_OUTPUT_SUM) with non-zero batch offsets, row bands and the
    compacted launch sequence: each is the oracle's sum (rto_render with output_sum) bit for bit."""
    check_against_oracle(ctx, cornell, B.default_params(64, 36, 16, lanes_per_pixel=4, flags=B.FLAG_OUTPUT_SUM))
    check_against_oracle(ctx, cornell, B.default_params(64, 36, 16, lanes_per_pixel=4, batch_offset=12, flags=B.FLAG_OUTPUT_SUM))
    check_against_oracle(ctx, spheres, B.default_params(80, 45, 8, lanes_per_pixel=2, batch_offset=1000, band_rows=8, band_index=2,
                                                        band_count=3, flags=B.FLAG_OUTPUT_SUM))
    check_against_oracle(ctx, build_scene("glass_spheres", 3), B.default_params(64, 36, 256, lanes_per_pixel=4, batch_offset=8,
                                                                                max_bounces=32, flags=B.FLAG_OUTPUT_SUM | B.FLAG_COMPACT_TAILS))


def test_accumulated_passes_against_the_oracle(ctx, cornell):
    """A three-pass progressive render (sum render, then two RT_FLAG_ACCUMULATE renders with advancing batch
    offsets) holds the float32 pass-order sum of three oracle sum renders; resolving it gives the pixels
    of packing that sum / 48 (the running average of win32gpu/compute.glsl:382-391, kept as a sum)."""
    ctx.upload_world(cornell)
    osc = O.OracleScene(cornell)
    lanes, spp = 4, 16
    expected = None
    rays_expected = 0
    got = None
    rays_got = 0
    for k in range(3):
        p = B.default_params(72, 40, spp, lanes_per_pixel=lanes, batch_offset=k * lanes,
                             flags=B.FLAG_OUTPUT_SUM if k == 0 else B.FLAG_ACCUMULATE)
        _, got, rays = ctx.render(p, want_pixels=False)
        rays_got += rays
        _, osum, orays = osc.render(oracle_params(B.default_params(72, 40, spp, lanes_per_pixel=lanes, batch_offset=k * lanes,
                                                                   flags=B.FLAG_OUTPUT_SUM)))
        expected = osum if expected is None else expected + osum      # float32, pass order
        rays_expected += orays
        assert np.array_equal(bits(got), bits(expected)), f"running sum differs after pass {k}"
    assert rays_got == rays_expected
    pixels, mean = ctx.resolve(got, 3 * spp)
    omean = expected / np.float32(3 * spp)
    assert np.array_equal(bits(mean), bits(omean))
    opix = np.array([O.lib().rto_pack_srgb(np.ascontiguousarray(px).ctypes.data_as(C.c_void_p)) for px in omean.reshape(-1, 3)],
                    dtype=np.uint32).reshape(pixels.shape)
    assert_pixels_close(pixels, opix)
    # the same chain through the device entry point with a caller-owned accumulator
    import torch
    dev = torch.device("cuda", 0)
    acc = torch.zeros((40, 72, 3), dtype=torch.float32, device=dev)
    rays = torch.zeros(3, dtype=torch.int64, device=dev)
    for k in range(3):
        p = B.default_params(72, 40, spp, lanes_per_pixel=lanes, batch_offset=k * lanes,
                             flags=B.FLAG_OUTPUT_SUM if k == 0 else B.FLAG_ACCUMULATE)
        ctx.render_device(p, None, acc.data_ptr(), rays.data_ptr() + 8 * k, None)
    torch.cuda.synchronize(dev)
    assert np.array_equal(bits(acc.cpu().numpy()), bits(expected)) and int(rays.sum()) == rays_expected


@pytest.mark.parametrize("parts", [1, 2, 3, 4, 5, 8])
def test_combine_adds_in_the_fixed_tree_order(ctx, parts):
    """rt_combine_device against the same tree written with element-wise float32 additions."""
    import torch
    from raytracer_amd import distributed as D
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device="cpu").manual_seed(parts)
    # magnitudes spread over 2^-20 .. 2^20, so that the order of the additions shows in the roundings
    data = (torch.randn((parts, 4099), generator=gen) * torch.exp2(torch.randint(-20, 21, (parts, 4099), generator=gen).float())).to(dev)
    out = torch.zeros(4099, dtype=torch.float32, device=dev)
    ctx.combine_device(data.data_ptr(), parts, 4099, 4099, out.data_ptr(), None)
    torch.cuda.synchronize(dev)
    assert torch.equal(out.cpu(), D.tree_order_sum(data.cpu()))
    if parts == 4:   # and it is not what a ring would give
        ring = ((data[0] + data[1]) + data[2]) + data[3]
        assert not torch.equal(out, ring)


def test_sample_sharded_composition_equals_one_render_of_all_batches(ctx, cornell):
    """Four sample shards (what four GPUs render in the `samples` mode: batch_offset = rank * lanes) added by
    rt_combine_device equal, bit for bit, the single render with 4 x lanes streams per pixel - and the oracle's."""
    import torch
    ctx.upload_world(cornell)
    dev = torch.device("cuda", 0)
    ranks, lanes, spp, w, h = 4, 4, 32, 96, 54
    shards = torch.zeros((ranks, h, w, 3), dtype=torch.float32, device=dev)
    rays = torch.zeros(ranks, dtype=torch.int64, device=dev)
    for r in range(ranks):
        p = B.default_params(w, h, spp, lanes_per_pixel=lanes, batch_offset=r * lanes, flags=B.FLAG_OUTPUT_SUM)
        ctx.render_device(p, None, shards[r].data_ptr(), rays.data_ptr() + 8 * r, None)
    total = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    ctx.combine_device(shards.data_ptr(), ranks, h * w * 3, h * w * 3, total.data_ptr(), None)
    torch.cuda.synchronize(dev)
    whole = B.default_params(w, h, spp * ranks, lanes_per_pixel=lanes * ranks, flags=B.FLAG_OUTPUT_SUM)
    _, rad, whole_rays = check_against_oracle(ctx, cornell, whole)
    assert int(rays.sum()) == whole_rays
    assert np.array_equal(bits(total.cpu().numpy()), bits(rad))


# ---------------------------------------------------------------- limits of the entry points

def test_two_contexts_render_concurrently_on_two_streams(cornell, spheres):
    """A compacted render's workspace belongs to its context, so concurrent renders need one context per
    stream: two contexts, two streams, renders enqueued alternately without waiting - each gives the image
    of a render on its own."""
    import torch
    dev = torch.device("cuda", 0)
    cases = [(cornell, B.default_params(320, 180, 256, lanes_per_pixel=4, flags=B.FLAG_COMPACT_TAILS)),
             (spheres, B.default_params(256, 144, 256, lanes_per_pixel=2, flags=B.FLAG_COMPACT_TAILS))]
    contexts, expected, streams, outputs = [], [], [], []
    for scene, p in cases:
        c = B.Context(0)
        c.upload_world(scene)
        c.reserve(p)
        contexts.append(c)
        expected.append(c.render(p, want_radiance=False))
        streams.append(torch.cuda.Stream(dev))
        outputs.append((torch.zeros((p.height, p.width), dtype=torch.int32, device=dev), torch.zeros(3, dtype=torch.int64, device=dev)))
    for repeat in range(3):
        for (scene, p), c, s, (pix, rays) in zip(cases, contexts, streams, outputs):
            c.render_device(p, pix.data_ptr(), None, rays.data_ptr() + 8 * repeat, s.cuda_stream)
    torch.cuda.synchronize(dev)
    for (want_pix, _, want_rays), (pix, rays) in zip(expected, outputs):
        assert np.array_equal(pix.cpu().numpy().view(np.uint32), want_pix)
        assert rays.cpu().tolist() == [want_rays] * 3
    for c in contexts:
        c.close()


def test_size_and_sharing_limits_are_refused(ctx, cornell):
    ctx.upload_world(cornell)
    # lanes and streams are indexed with 32 bits: 65536 x 65536 pixels at 4 lanes per pixel is refused, not wrapped
    with pytest.raises(B.RtError) as e:
        ctx.reserve(B.default_params(65536, 65536, 4, lanes_per_pixel=4))
    assert e.value.status == 1 and "2^32" in str(e.value)
    # a context keeps one running sum: listed twice, it cannot render sums through rt_render_multi
    with pytest.raises(B.RtError) as e:
        B.render_multi([ctx, ctx], B.default_params(64, 36, 8, flags=B.FLAG_OUTPUT_SUM))
    assert e.value.status == 1 and "more than once" in str(e.value)
    pix, _, _ = B.render_multi([ctx, ctx], B.default_params(64, 36, 8))      # plain renders may share a context
    assert np.array_equal(pix, ctx.render(B.default_params(64, 36, 8))[0])
