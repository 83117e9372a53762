"""Multi-GPU composition of renders: one process per GPU, torch.distributed (RCCL on GPUs).

The reference's unit of work is the image row (one WorkOrder per row, main.cpp:433-448) and
rows never exchange data, so a render shards with no collective on the data path; the only
exchange is assembling the result on rank 0:

  rows     each rank renders interleaved 8-row bands of the image (band b belongs to rank
           b mod N: the Cornell box is half black rows, so contiguous slabs would be uneven)
           and the packed rows are collected with ONE all_gather.
  samples  each rank renders every pixel with its own sample-batch indices
           (batch_offset = rank * lanes_per_pixel); the un-normalised radiance sums are
           collected on rank 0 with ONE gather and added there in a FIXED pairwise tree over
           the ranks (rank 0 + rank 1, rank 2 + rank 3, then the pairs, ...), then tonemapped.
           A reduce(SUM) would add in whatever order the transport's ring or tree visits the
           ranks; the fixed tree continues the kernel's own pairwise tree over the
           lanes_per_pixel streams of a pixel, so for a power-of-two number of ranks the image
           equals, bit for bit, the single-GPU render of all the batches
           (lanes_per_pixel * world_size streams per pixel).

The renderer itself is injected (`render` / `resolve` callables working on torch tensors),
so the same composition runs on GPUs with the HIP library and, in the CPU tests, over gloo.
"""
from __future__ import annotations

import copy
from typing import Callable

import torch
import torch.distributed as dist

BAND_ROWS = 8


def _host_staged() -> bool:
    """True when the process group cannot move device tensors itself (gloo): collectives then go through
    host copies.  RCCL (backend "nccl") moves device memory directly over xGMI."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def band_params(params, rank: int, world_size: int, band_rows: int = BAND_ROWS):
    """Copy of `params` restricted to the row bands of `rank`."""
    p = copy.copy(params)
    if world_size > 1:
        p.band_rows, p.band_index, p.band_count = band_rows, rank, world_size
    return p


def local_rows(params) -> int:
    total = params.row_end - params.row_begin
    if params.band_rows == 0 or params.band_count <= 1:
        return total
    bands = (total + params.band_rows - 1) // params.band_rows
    n = 0
    for b in range(params.band_index, bands, params.band_count):
        n += min((b + 1) * params.band_rows, total) - b * params.band_rows
    return n


def global_rows(params) -> list[int]:
    rows = []
    for local in range(local_rows(params)):
        if params.band_rows == 0 or params.band_count <= 1:
            rows.append(params.row_begin + local)
        else:
            band, within = divmod(local, params.band_rows)
            rows.append(params.row_begin + (band * params.band_count + params.band_index) * params.band_rows + within)
    return rows


def batch_params(params, rank: int, world_size: int, output_sum_flag: int):
    """Copy of `params` whose sample batches are those of `rank`; asks for the radiance sum."""
    p = copy.copy(params)
    p.batch_offset = params.batch_offset + rank * params.lanes_per_pixel
    p.flags = params.flags | output_sum_flag
    return p


class RowShardedImage:
    """Buffers and the gather for the `rows` mode.  All ranks hold equal-sized shards (the
    largest band set), as all_gather requires; rank 0 un-interleaves."""

    def __init__(self, params, rank: int, world_size: int, device, band_rows: int = BAND_ROWS):
        self.rank, self.world_size = rank, world_size
        self.params = band_params(params, rank, world_size, band_rows)
        self.all_params = [band_params(params, r, world_size, band_rows) for r in range(world_size)]
        self.max_rows = max(local_rows(p) for p in self.all_params)
        self.width = params.width
        self.shard = torch.zeros((self.max_rows, params.width), dtype=torch.int32, device=device)
        # concatenation of the shards along rows; viewed as [rank, row, x]
        self._gathered_flat = torch.zeros((world_size * self.max_rows, params.width), dtype=torch.int32, device=device)
        self.gathered = self._gathered_flat.view(world_size, self.max_rows, params.width)
        self._source_rows = None

    def gather(self):
        # With a process group the collective runs even for a single rank (it is then a copy by the transport:
        # a one-rank launch exercises the same calls as an N-rank one); without one the shard is the image.
        if dist.is_initialized() and self.shard.is_cuda and _host_staged():
            flat = torch.zeros(self._gathered_flat.shape, dtype=self.shard.dtype)
            dist.all_gather_into_tensor(flat, self.shard.cpu())
            self._gathered_flat.copy_(flat)
        elif dist.is_initialized():
            dist.all_gather_into_tensor(self._gathered_flat, self.shard)
        else:
            self.gathered[0].copy_(self.shard)

    def assemble(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """Full image [rows, W] in global row order (call after gather): one gather-by-row-index copy."""
        if self._source_rows is None:
            total = self.all_params[0].row_end - self.all_params[0].row_begin
            begin = self.all_params[0].row_begin
            source = torch.zeros(total, dtype=torch.long)
            for r, p in enumerate(self.all_params):
                for local, g in enumerate(global_rows(p)):
                    source[g - begin] = r * self.max_rows + local      # row of the concatenated shards
            self._source_rows = source.to(self._gathered_flat.device)
        if out is None:
            return self._gathered_flat.index_select(0, self._source_rows)
        return torch.index_select(self._gathered_flat, 0, self._source_rows, out=out)


class GatheredSums:
    """Receive buffer of the `samples` mode on rank 0: [world_size, rows, W, 3] float32, one slice per rank."""

    def __init__(self, like: torch.Tensor, rank: int, world_size: int):
        self.rank, self.world_size = rank, world_size
        self.parts = torch.zeros((world_size,) + tuple(like.shape), dtype=like.dtype, device=like.device) if rank == 0 else None

    def gather(self, radiance_sum: torch.Tensor):
        """Collects every rank's sum on rank 0 (one gather; nothing is added in transit)."""
        self.finish(self.start(radiance_sum))

    def start(self, radiance_sum: torch.Tensor):
        """Begins the gather of `radiance_sum` and returns a handle for finish().  On RCCL the transfer runs on
        the transport's own stream, ordered after everything enqueued so far on the current stream, and the
        current stream is free to go on (the next render) until finish(); `radiance_sum` must not be written
        and `parts` not read in between.  Host-staged groups (gloo) and single processes complete here."""
        if not dist.is_initialized():
            self.parts[0].copy_(radiance_sum)
            return None
        if radiance_sum.is_cuda and _host_staged():
            mine = radiance_sum.cpu()
            if self.rank == 0:
                received = [torch.zeros_like(mine) for _ in range(self.world_size)]
                dist.gather(mine, gather_list=received, dst=0)
                for r in range(self.world_size):
                    self.parts[r].copy_(received[r])
            else:
                dist.gather(mine, gather_list=None, dst=0)
            return None
        gather_list = [self.parts[r] for r in range(self.world_size)] if self.rank == 0 else None
        return dist.gather(radiance_sum, gather_list=gather_list, dst=0, async_op=True)

    def finish(self, handle):
        """Makes the current stream wait for the gather begun by start() (the host does not block on RCCL)."""
        if handle is not None:
            handle.wait()


def tree_order_sum(parts: torch.Tensor) -> torch.Tensor:
    """Sum over the first axis in the fixed pairwise order of the kernel's per-pixel tree: neighbours
    first (0+1, 2+3, ...), then neighbouring pairs, left operand first; a count that is not a power of
    two carries its odd block up unchanged.  Element-wise float32 additions only, so the result does not
    depend on who computes it."""
    level = [parts[i] for i in range(parts.shape[0])]
    while len(level) > 1:
        nxt = [level[i] + level[i + 1] for i in range(0, len(level) - 1, 2)]
        if len(level) % 2:
            nxt.append(level[-1])
        level = nxt
    return level[0]


def sum_rays(rays: torch.Tensor, world_size: int) -> torch.Tensor:
    if world_size > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    return rays


def render_rows(render: Callable, params, rank: int, world_size: int, device) -> tuple[RowShardedImage, int]:
    """rows mode in one call: render this rank's bands, gather.  `render(params, shard) -> rays`."""
    img = RowShardedImage(params, rank, world_size, device)
    rays = 0
    if local_rows(img.params):
        rays = render(img.params, img.shard)
    img.gather()
    return img, rays


def render_samples(render: Callable, resolve: Callable, params, rank: int, world_size: int, device,
                   output_sum_flag: int, combine: Callable = tree_order_sum):
    """samples mode in one call.  `render(params, radiance_sum) -> rays` fills [H, W, 3] float32;
    `combine(parts [N, H, W, 3]) -> sum` adds the ranks' sums in the fixed tree order and
    `resolve(radiance_sum, total_spp) -> pixels` tonemaps; both run on rank 0 only.
    Returns (pixels or None, rays)."""
    p = batch_params(params, rank, world_size, output_sum_flag)
    rows = p.row_end - p.row_begin
    radiance_sum = torch.zeros((rows, p.width, 3), dtype=torch.float32, device=device)
    rays = render(p, radiance_sum)
    gathered = GatheredSums(radiance_sum, rank, world_size)
    gathered.gather(radiance_sum)
    pixels = resolve(combine(gathered.parts), p.samples_per_pixel * world_size) if rank == 0 else None
    return pixels, rays
