"""Resumable float framebuffer ("RTFB" files): the running radiance sum of a progressive render.

Same file layout as the C++ host code (raytracer_amd/host/rt_framebuffer.h): 64-byte little-endian
header, then height*width*3 float32 sums, rows top-down.  The reference's CPU renderer keeps no such
state; its GPU front end averages frames in place (win32gpu/compute.glsl:382-391).  Here the state
is a sum, so adding passes is exact and order-defined, and the image is resolved from it at the end
(color / sampleSize and the sRGB pack of main.cpp:393).
"""
from __future__ import annotations

import dataclasses
import os
import struct

import numpy as np

MAGIC = b"RTFB"
VERSION = 1
HEADER = struct.Struct("<4sIIIIIIIQQQ8x")   # magic, version, w, h, channels, samples, next_batch, lanes, scene, settings, rays
assert HEADER.size == 64

_FNV_OFFSET, _FNV_PRIME, _MASK = 1469598103934665603, 1099511628211, (1 << 64) - 1


def _fnv(data: bytes, h: int = _FNV_OFFSET) -> int:
    for b in data:
        h = ((h ^ b) * _FNV_PRIME) & _MASK
    return h


def scene_hash(scene) -> int:
    """FNV-1a over the scene's records and camera: HashWorld of rt_framebuffer.cpp."""
    h = _FNV_OFFSET
    for arr in (scene.materials, scene.planes, scene.spheres, scene.rectangles):
        h = _fnv(struct.pack("<I", len(arr)), h)
        h = _fnv(np.ascontiguousarray(arr).tobytes(), h)
    return _fnv(np.ascontiguousarray(scene.camera).tobytes(), h)


def settings_hash(params) -> int:
    """FNV-1a over the parameters that define the estimator: HashSettings of rt_framebuffer.cpp."""
    return _fnv(struct.pack("<IIIIffIIII", params.width, params.height, params.samples_per_pixel, params.max_bounces,
                            params.min_hit_distance, params.plane_tolerance, params.seed_constant, params.rng_mode,
                            params.lanes_per_pixel, params.intersect_mode))


@dataclasses.dataclass
class Accumulator:
    width: int
    height: int
    samples: int = 0
    next_batch: int = 0
    lanes_per_pixel: int = 1
    scene_hash: int = 0
    settings_hash: int = 0
    rays: int = 0
    sum: np.ndarray | None = None      # [height, width, 3] float32

    def __post_init__(self):
        if self.sum is None:
            self.sum = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self.sum = np.ascontiguousarray(self.sum, dtype=np.float32)
        if self.sum.shape != (self.height, self.width, 3):
            raise ValueError(f"sum has shape {self.sum.shape}, expected {(self.height, self.width, 3)}")

    def add_pass(self, pass_sum: np.ndarray, pass_samples: int, pass_rays: int = 0) -> None:
        """sum += pass_sum in float32 (pass order defines the result), and the counters advance."""
        self.sum += np.asarray(pass_sum, dtype=np.float32).reshape(self.sum.shape)
        self.samples += int(pass_samples)
        self.next_batch += self.lanes_per_pixel
        self.rays += int(pass_rays)

    def write(self, path) -> None:
        tmp = f"{path}.tmp"
        with open(tmp, "wb") as f:
            f.write(HEADER.pack(MAGIC, VERSION, self.width, self.height, 3, self.samples, self.next_batch,
                                self.lanes_per_pixel, self.scene_hash, self.settings_hash, self.rays))
            f.write(self.sum.astype("<f4", copy=False).tobytes())
        os.replace(tmp, path)

    @classmethod
    def read(cls, path) -> "Accumulator":
        with open(path, "rb") as f:
            head = f.read(HEADER.size)
            if len(head) < HEADER.size:
                raise ValueError(f"{path}: shorter than the 64-byte header")
            magic, version, w, h, channels, samples, next_batch, lanes, scene, settings, rays = HEADER.unpack(head)
            if magic != MAGIC:
                raise ValueError(f"{path}: not an RTFB file")
            if version != VERSION:
                raise ValueError(f"{path}: unsupported version {version}")
            if channels != 3 or not (0 < w <= 65536 and 0 < h <= 65536):
                raise ValueError(f"{path}: bad image size or channel count")
            body = f.read()
        if len(body) < w * h * 12:
            raise ValueError(f"{path}: pixel data truncated")
        if len(body) > w * h * 12:
            raise ValueError(f"{path}: trailing bytes after the pixel data")
        data = np.frombuffer(body, dtype="<f4").reshape(h, w, 3).astype(np.float32)
        return cls(w, h, samples, next_batch, lanes, scene, settings, rays, data)


def render_progressive(ctx, params, passes: int, accumulator: Accumulator | None = None, checkpoint=None) -> Accumulator:
    """Runs `passes` renders of params.samples_per_pixel samples each on `ctx` (a binding.Context with its
    world uploaded), each on fresh sample batches, and adds their sums to `accumulator` (a new one if None);
    writes `checkpoint` after every pass.  The caller resolves the image with ctx.resolve(acc.sum, acc.samples)."""
    from . import binding as B
    if params.rng_mode != B.RNG_PIXEL:
        raise ValueError("progressive rendering needs RNG_PIXEL: a row-serial stream has no second batch")
    if (params.row_begin, params.row_end) != (0, params.height) or (params.band_rows and params.band_count > 1):
        raise ValueError("progressive rendering keeps a full-image sum: params must cover every row of the image")
    world = scene_hash(ctx._scene) if getattr(ctx, "_scene", None) is not None else 0
    settings = settings_hash(params)
    acc = accumulator or Accumulator(params.width, params.height, lanes_per_pixel=params.lanes_per_pixel,
                                     scene_hash=world, settings_hash=settings)
    if (acc.width, acc.height, acc.lanes_per_pixel) != (params.width, params.height, params.lanes_per_pixel):
        raise ValueError("accumulator belongs to another image size or lanes_per_pixel")
    # A sum of another scene or estimator must not be continued (the C++ driver refuses the same way, --resume).
    if acc.samples > 0 and (acc.scene_hash != world or acc.settings_hash != settings):
        raise ValueError("accumulator belongs to another scene or to other render settings (seed, bounces, spp, mode)")
    # The running sum stays in the context (FLAG_ACCUMULATE adds each pass to it on the GPU, in the order a host
    # loop `sum += pass` would); it is copied back for a checkpoint and after the last pass.
    flags, offset = params.flags, params.batch_offset
    started = acc.samples > 0
    try:
        if started:
            ctx.load_accumulator(params, acc.sum)
        for i in range(passes):
            params.batch_offset = acc.next_batch
            params.flags = (flags & ~(B.FLAG_OUTPUT_SUM | B.FLAG_ACCUMULATE)) | (B.FLAG_ACCUMULATE if started else B.FLAG_OUTPUT_SUM)
            need_sum = checkpoint is not None or i + 1 == passes
            _, total, rays = ctx.render(params, want_pixels=False, want_radiance=need_sum)
            started = True
            if need_sum:
                acc.sum = total
            acc.samples += params.samples_per_pixel
            acc.next_batch += acc.lanes_per_pixel
            acc.rays += rays
            if checkpoint is not None:
                acc.write(checkpoint)
    finally:
        params.flags, params.batch_offset = flags, offset
    return acc
