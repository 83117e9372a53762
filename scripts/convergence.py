#!/usr/bin/env python3
"""Convergence of the two stream layouts on the Cornell box (320x180): error of an N-spp render against a
65536-spp render, for the reference's row-serial streams and for the per-pixel streams.  Both are unbiased
estimators of the same image, so their error curves must coincide (and fall like 1/sqrt(N)); that is the
tolerance statement for comparing a per-pixel-stream image with the reference's image.
Writes a markdown table to the path given as first argument (default: stdout only)."""
import sys
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from raytracer_amd import binding as B                     # noqa: E402
from raytracer_amd import framebuffer as F                 # noqa: E402
from raytracer_amd.host_scene import build_scene           # noqa: E402

W, H = 320, 180


def channels(pix):
    return ((pix[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 255).astype(np.float64)


def psnr8(a, b):
    mse = np.mean((channels(a) - channels(b)) ** 2)
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def main():
    ctx = B.Context(0)
    ctx.upload_world(build_scene("cornell"))
    # truth: 16 passes of 4096 spp on batches far away from the ones used below
    p = B.default_params(W, H, 4096, lanes_per_pixel=4, batch_offset=1 << 20)
    acc = F.render_progressive(ctx, p, 16, accumulator=F.Accumulator(W, H, next_batch=1 << 20, lanes_per_pixel=4))
    truth_pix, truth = ctx.resolve(acc.sum, acc.samples)
    clip = lambda x: np.minimum(x, 1.0)     # errors are measured on the displayable range
    lines = ["| spp | row-serial streams: RMSE | max abs | PSNR 8-bit | per-pixel streams: RMSE | max abs | PSNR 8-bit | second row-serial seed: RMSE |",
             "|---|---|---|---|---|---|---|---|"]
    for spp in (16, 64, 256, 1024):
        row = []
        for params in (B.default_params(W, H, spp, rng_mode=B.RNG_ROW_SERIAL),
                       B.default_params(W, H, spp, lanes_per_pixel=2),
                       B.default_params(W, H, spp, rng_mode=B.RNG_ROW_SERIAL, seed_constant=971)):
            pix, rad, _ = ctx.render(params)
            err = clip(rad) - clip(truth)
            row.append((np.sqrt(np.mean(err ** 2)), np.abs(err).max(), psnr8(pix, truth_pix)))
        lines.append(f"| {spp} | {row[0][0]:.5f} | {row[0][1]:.3f} | {row[0][2]:.2f} dB | {row[1][0]:.5f} | {row[1][1]:.3f} | "
                     f"{row[1][2]:.2f} dB | {row[2][0]:.5f} |")
        print(lines[-1], flush=True)
    text = "\n".join(
        ["# Convergence of the stream layouts (Cornell box 320x180, errors against a 65536-spp render, radiance clipped to [0,1])", "",
         "`python scripts/convergence.py profiles/r01_convergence.md` on one MI355X.", ""] + lines + [""])
    if len(sys.argv) > 1:
        pathlib.Path(sys.argv[1]).write_text(text)
    print(text)


if __name__ == "__main__":
    main()
