"""The native render driver (raytracer_amd/bin/raytracer), counterpart of the reference's main()."""
import pathlib
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
DRIVER = ROOT / "raytracer_amd" / "bin" / "raytracer"


def test_usage_without_arguments_needed():
    """--help needs no GPU: prints the option list and exits with status 2."""
    proc = subprocess.run([str(DRIVER), "--help"], capture_output=True, text=True)
    assert proc.returncode == 2
    for option in ("--scene", "--width", "--spp", "--bounces", "--rng", "--lanes-per-pixel", "--intersect", "--gpus", "--out"):
        assert option in proc.stderr
    assert subprocess.run([str(DRIVER), "--width", "0"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_driver_renders_the_same_image_as_the_library(tmp_path):
    """Report lines in the reference's format (main.cpp:468-471) and a BMP whose pixels equal rt_render's."""
    from raytracer_amd import binding as B
    from raytracer_amd.host_scene import build_scene
    out = tmp_path / "render.bmp"
    proc = subprocess.run([str(DRIVER), "--scene", "cornell", "--width", "160", "--height", "90", "--spp", "32",
                           "--lanes-per-pixel", "2", "--out", str(out)], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    m = re.search(r"Raytracing time: (\d+)ms\nTotal computed rays: (\d+)\nPerformance: ([\d.]+)Mray/s, ([\d.]+)ms/ray", proc.stdout)
    assert m, proc.stdout
    data = out.read_bytes()
    assert struct.unpack_from("<HI", data, 0) == (0x4D42, 54 + 160 * 90 * 4)
    pixels = np.frombuffer(data, dtype=np.uint32, offset=54).reshape(90, 160)
    with B.Context(0) as ctx:
        ctx.upload_world(build_scene("cornell"))
        ref, _, rays = ctx.render(B.default_params(160, 90, 32, lanes_per_pixel=2))
    assert int(m.group(2)) == rays
    assert np.array_equal(pixels, ref)


@pytest.mark.gpu
def test_driver_row_serial_mode_matches_the_reference_fixture(tmp_path, golden_dir):
    """--rng row: the reference's own streams; the Cornell 64x36x16 fixture's ray count and pixels (<= 1 LSB)."""
    g = np.load(golden_dir / "image_cornell_64x36_16spp.npz")
    out = tmp_path / "row.bmp"
    proc = subprocess.run([str(DRIVER), "--scene", "cornell", "--width", "64", "--height", "36", "--spp", "16",
                           "--rng", "row", "--out", str(out)], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    assert f"Total computed rays: {int(g['rays'])}" in proc.stdout
    pixels = np.frombuffer(out.read_bytes(), dtype=np.uint32, offset=54).reshape(36, 64)
    diff = np.abs(((pixels[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(int) -
                  ((g["pixels"][..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3


@pytest.mark.gpu
def test_driver_compact_option_writes_the_same_file(tmp_path):
    """--compact (RT_FLAG_COMPACT_TAILS) changes the launch schedule, not the image or the ray count."""
    common = [str(DRIVER), "--scene", "glass_spheres", "--width", "160", "--height", "90", "--spp", "256", "--bounces", "32",
              "--lanes-per-pixel", "4"]
    plain = subprocess.run(common + ["--out", str(tmp_path / "plain.bmp")], capture_output=True, text=True, timeout=120)
    compact = subprocess.run(common + ["--compact", "--out", str(tmp_path / "compact.bmp")], capture_output=True, text=True, timeout=120)
    assert plain.returncode == 0 and compact.returncode == 0, plain.stderr + compact.stderr
    rays = lambda out: re.search(r"Total computed rays: (\d+)", out).group(1)
    assert rays(plain.stdout) == rays(compact.stdout)
    assert (tmp_path / "plain.bmp").read_bytes() == (tmp_path / "compact.bmp").read_bytes()
