"""The stated tolerance against the reference renderer: per-pixel max-abs error on linear radiance and
PSNR of the 8-bit image, on the reference's own 320x180 renders at 64 spp (tests/golden/reference_*.npz,
made by oracle/make_golden.py from the reference's sources built with -ffp-contract=off).

Same random streams (RT_RNG_ROW_SERIAL, the reference's row seeds):
  Cornell box (rectangles)          max-abs 0,      PSNR infinite  (bit-exact)
  spheres scene (createScene)       max-abs 0.225,  PSNR 35.65 dB, 1.3 % of the pixels differ
  glass3 (dielectric spheres)       max-abs 0.770,  PSNR 36.52 dB, 2.5 % of the pixels differ
The sphere scenes differ from the reference in ONE operation: the reference scales sphere normals by the
x86 rsqrtps approximation (simd_avx2.h:139-144, relative error <= 1.5 * 2^-12, defined by the CPU model),
the GPU by the correctly rounded 1/sqrt.  A path keeps the reference's bits until that perturbation
flips one of its decisions; the oracle's RTO_INTERSECT_WIDE8 mode, which executes the x86 instruction,
reproduces the reference bit for bit on an x86 host, which pins everything else.

Other random streams (RT_RNG_PIXEL, the throughput mode): the image differs from the reference's like
the reference's own image with another seed constant does: PSNR at 64 spp 15.3 dB (Cornell), 19.7 dB
(spheres), 15.7 dB (glass3), equal to the reseeded reference's within 0.1 dB, and 16x16-pixel block means
agree within their Monte-Carlo error.
"""
import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import binding as B
from raytracer_amd.scene_io import read_scene

SCENES = ("cornell", "spheres", "glass3")
# (max-abs on linear radiance, PSNR of the 8-bit image in dB, fraction of differing pixels) of the
# correctly-rounded-rsqrt arithmetic against the reference, same streams, 320x180 at 64 spp
SAME_STREAM_BOUNDS = {"cornell": (0.0, 99.0, 0.0), "spheres": (0.2250, 35.65, 0.0132), "glass3": (0.7701, 36.52, 0.0253)}


def channels(pix):
    return ((pix[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(np.float64)


def psnr8(a, b):
    mse = np.mean((channels(a) - channels(b)) ** 2)
    return 99.0 if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


def load(golden_dir, name):
    g = np.load(golden_dir / f"reference_{name}_320x180_64spp.npz")
    return read_scene(golden_dir / f"{name}.rtsc"), g["radiance"], g["pixels"], int(g["rays"])


def check_same_streams(name, pix, rad, rays, ref_rad, ref_pix, ref_rays):
    max_abs, min_psnr, max_fraction = SAME_STREAM_BOUNDS[name]
    diff = np.abs(rad - ref_rad)
    assert diff.max() <= max_abs + 1e-4, f"{name}: max-abs {diff.max()}"
    assert psnr8(pix, ref_pix) >= min_psnr - 0.01, f"{name}: PSNR {psnr8(pix, ref_pix)}"
    assert (diff.max(axis=2) > 0).mean() <= max_fraction + 1e-4
    assert abs(rays - ref_rays) <= 4e-5 * ref_rays


def block_mean_z(rad, ref_rad, block=16):
    """z-scores of the difference of block means (radiance clipped to [0, 1], channels averaged) in units
    of the standard error estimated from the per-pixel differences inside each block."""
    d = (np.minimum(rad, 1.0) - np.minimum(ref_rad, 1.0)).mean(axis=2)
    h, w = (d.shape[0] // block) * block, (d.shape[1] // block) * block
    tiles = d[:h, :w].reshape(h // block, block, w // block, block).transpose(0, 2, 1, 3).reshape(-1, block * block)
    se = tiles.std(axis=1, ddof=1) / np.sqrt(block * block)
    keep = se > 0
    return tiles.mean(axis=1)[keep] / se[keep]


def check_other_streams(name, pix, rad, ref_rad, ref_pix, floor_psnr):
    assert psnr8(pix, ref_pix) >= floor_psnr - 0.3, f"{name}: PSNR {psnr8(pix, ref_pix)} below the reseeded reference's {floor_psnr}"
    assert abs(rad.mean() / ref_rad.mean() - 1.0) <= 0.01
    z = block_mean_z(rad, ref_rad)
    assert (np.abs(z) > 4.0).mean() <= 0.01 and 0.5 <= np.mean(z ** 2) <= 2.0, (np.abs(z).max(), np.mean(z ** 2))


def reseeded_reference_psnr(scene, ref_pix):
    """The reference's arithmetic with another seed constant (main.cpp:370) against its own image."""
    pix, _, _ = O.OracleScene(scene).render(O.default_params(320, 180, 64, intersect_mode=O.WIDE8, seed_constant=777767))
    return psnr8(pix, ref_pix)


# ------------------------------------------------------------------------------------- CPU: the oracle

@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_the_reference_images(golden_dir, name):
    """The restatement in the reference's own arithmetic (x86 rsqrt for sphere normals): bit for bit."""
    scene, ref_rad, ref_pix, ref_rays = load(golden_dir, name)
    pix, rad, rays = O.OracleScene(scene).render(O.default_params(320, 180, 64, intersect_mode=O.WIDE8))
    assert rays == ref_rays
    assert np.array_equal(rad.view(np.uint32), ref_rad.view(np.uint32)) and np.array_equal(pix, ref_pix)


@pytest.mark.parametrize("name", SCENES)
def test_lane_width_four_gives_the_reference_images_too(golden_dir, name):
    """The width-generic restatement at width 8 is the pinned 8-wide routine bit for bit, and at width 4 it still
    reproduces the reference's LANE_WIDTH 8 images of these scenes at this size (no tie between lanes decides a
    pixel here).  Where ties do decide - the tie scene, the Cornell box at 1280x720 - the width-4 order is pinned
    against the reference built at LANE_WIDTH 4 (tests/test_oracle_golden.py)."""
    scene, ref_rad, ref_pix, ref_rays = load(golden_dir, name)
    for mode in (O.WIDE8_LANES, O.WIDE4):
        pix, rad, rays = O.OracleScene(scene).render(O.default_params(320, 180, 64, intersect_mode=mode))
        assert rays == ref_rays and np.array_equal(rad, ref_rad) and np.array_equal(pix, ref_pix), (name, mode)


@pytest.mark.parametrize("name", SCENES)
def test_correctly_rounded_rsqrt_against_the_reference(golden_dir, name):
    """The arithmetic the GPU runs (correctly rounded 1/sqrt for sphere normals), same streams."""
    scene, ref_rad, ref_pix, ref_rays = load(golden_dir, name)
    pix, rad, rays = O.OracleScene(scene).render(O.default_params(320, 180, 64, intersect_mode=O.WIDE8_IEEE))
    check_same_streams(name, pix, rad, rays, ref_rad, ref_pix, ref_rays)


@pytest.mark.parametrize("name", SCENES)
def test_pixel_streams_against_the_reference_statistically(golden_dir, name):
    scene, ref_rad, ref_pix, _ = load(golden_dir, name)
    pix, rad, _ = O.OracleScene(scene).render(O.default_params(320, 180, 64, intersect_mode=O.WIDE8_IEEE, rng_mode=O.PIXEL,
                                                               lanes_per_pixel=4))
    check_other_streams(name, pix, rad, ref_rad, ref_pix, reseeded_reference_psnr(scene, ref_pix))


# ------------------------------------------------------------------------------------- GPU: the HIP path

@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_row_streams_against_the_reference(golden_dir, name):
    scene, ref_rad, ref_pix, ref_rays = load(golden_dir, name)
    with B.Context(0) as ctx:
        ctx.upload_world(scene)
        pix, rad, rays = ctx.render(B.default_params(320, 180, 64, rng_mode=B.RNG_ROW_SERIAL))
    check_same_streams(name, pix, rad, rays, ref_rad, ref_pix, ref_rays)
    if name == "cornell":
        assert rays == ref_rays and np.array_equal(rad.view(np.uint32), ref_rad.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_pixel_streams_against_the_reference(golden_dir, name):
    scene, ref_rad, ref_pix, _ = load(golden_dir, name)
    with B.Context(0) as ctx:
        ctx.upload_world(scene)
        pix, rad, _ = ctx.render(B.default_params(320, 180, 64, lanes_per_pixel=4))
    check_other_streams(name, pix, rad, ref_rad, ref_pix, reseeded_reference_psnr(scene, ref_pix))
