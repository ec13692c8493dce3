"""Multi-GPU frames: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The frame is cut into 8x8-pixel tiles (one 64-lane wave each, row-major tile index t); rank r of N owns the
tiles with t % N == r -- interleaving balances sky-heavy and geometry-heavy regions.  Every
rank integrates its tiles into a zeroed full-frame sum buffer (three floats per pixel, packed), one `reduce(SUM)` puts the frame on
rank 0 (x + 0 = x, so the reduced frame is bit-identical to the single-GPU sums), and rank 0 resolves
(mean, sqrt gamma, RGBA8: render.rs:103-110).  The random streams are keyed by (seed, pixel, sample), never
by rank, so the image does not depend on N.

Reference for the decomposition it replaces: the per-pixel rayon loop, lib/core/render.rs:114-119.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

TILE = 8


def tile_grid(width: int, height: int):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def owned_pixel_mask(width: int, height: int, shard: int, n_shards: int) -> np.ndarray:
    """Boolean [height, width] mask of the pixels rank `shard` integrates (the kernel's mapping:
    tile = blockIdx.x * n_shards + shard, csrc/rtr_kernels.hip)."""
    if not (0 <= shard < n_shards):
        raise ValueError("shard must be in [0, n_shards)")
    tx, ty = tile_grid(width, height)
    tiles = np.arange(tx * ty, dtype=np.int64).reshape(ty, tx)
    owned = (tiles % n_shards) == shard
    return np.repeat(np.repeat(owned, TILE, axis=0), TILE, axis=1)[:height, :width]


def sample_range(shard: int, n_shards: int, spp: int):
    """(begin, count) of the samples rank `shard` integrates when a frame's spp are split over the ranks: contiguous
    ranges, the first spp % n_shards ranks take one sample more."""
    if not (0 <= shard < n_shards):
        raise ValueError("shard must be in [0, n_shards)")
    base, extra = divmod(spp, n_shards)
    begin = shard * base + min(shard, extra)
    return begin, base + (1 if shard < extra else 0)


def render_frame_distributed(renderer, width: int, height: int, seed: int, accum, rgba8=None, stream: int = 0,
                             group=None, integrate: Optional[Callable] = None, spp: Optional[int] = None,
                             resolve: Optional[Callable] = None):
    """One frame across the ranks of `group`.  `accum`: this rank's [H, W, 3] float32 tensor (device memory
    for the HIP path); `rgba8`: rank 0's [H, W, 4] uint8 output.  `integrate` / `resolve` default to the HIP
    entry points of `renderer`; the CPU (gloo) tests substitute stand-ins to exercise the sharding and the
    collective without a GPU."""
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    accum.zero_()
    if integrate is None:
        renderer.render_tiles_device(width, height, seed, rank, world, accum.data_ptr(), stream)
    else:
        integrate(width, height, seed, rank, world, accum)
    dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, group=group)
    if rank == 0 and rgba8 is not None:
        if spp is None:
            spp = renderer._scene.settings.samples_per_pixel
        if resolve is None:
            renderer.resolve_device(accum.data_ptr(), spp, width, height, rgba8.data_ptr(), 0, stream)
        else:
            resolve(accum, spp, rgba8)
    return accum


class _NullStream:
    """Stand-ins for torch.cuda streams / events when the pipeline runs on CPU tensors (gloo tests): every operation is
    synchronous there, so ordering is program order."""
    cuda_stream = 0

    def wait_event(self, event):
        pass

    def synchronize(self):
        pass


class _NullEvent:
    def record(self, stream=None):
        pass


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class FramePipeline:
    """Double-buffered multi-GPU frames: the RCCL reduce (+ rank 0's resolve) of frame i runs on a communication
    stream while the integrate kernel of frame i+1 runs on the compute stream, so a frame costs
    max(integrate, reduce) instead of their sum.  Two packed-RGB sum buffers (12 bytes per pixel: no padding on the wire) alternate; HIP events order
    zero -> integrate -> reduce -> resolve -> reuse.  Results are identical to `render_frame_distributed`.

    With `device` = "cpu" the same sequence runs on CPU tensors over whatever backend the process group has (gloo in
    tests/test_distributed_cpu.py, with a stand-in for the two renderer entry points): the sharding, buffer rotation and
    the collective are the code under test there, the kernels are not."""

    def __init__(self, renderer, width: int, height: int, spp: int, device, group=None, split: str = "samples"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.r, self.w, self.h, self.spp, self.group = renderer, width, height, spp, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.on_gpu = torch.device(device).type != "cpu"
        self.accum = [torch.zeros((height, width, 3), dtype=torch.float32, device=device) for _ in range(2)]  # packed RGB partial sums
        self.rgba8 = torch.zeros((height, width, 4), dtype=torch.uint8, device=device) if self.rank == 0 else None
        if self.on_gpu:
            self.compute = torch.cuda.Stream(device=device)
            self.comm = torch.cuda.Stream(device=device)
            self.rendered = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
        else:
            self.compute, self.comm = _NullStream(), _NullStream()
            self.rendered = [_NullEvent() for _ in range(2)]
            self.free = [_NullEvent() for _ in range(2)]
        # "samples": every rank integrates ALL pixels for its contiguous share of the samples -- each GPU keeps a
        # full grid of (shorter) waves, so the per-GPU kernel time scales 1/N; the reduce is a true sum of partial
        # sums (float addition order differs from one GPU: ~1e-7 relative).  "tiles": round-robin 8x8 tiles, all
        # samples: bit-identical to one GPU (x + 0 = x) but a grid of N-times fewer, equally long waves underfills
        # the GPU (measured on one MI355X: 43 % per-rank efficiency at N = 8, 60 % at N = 4).
        if split not in ("samples", "tiles"):
            raise ValueError("split must be 'samples' or 'tiles'")
        self.split = split if spp >= self.world else "tiles"
        self.n = 0
        if self.on_gpu:
            torch.cuda.synchronize(device)

    def _on(self, stream):
        return self.torch.cuda.stream(stream) if self.on_gpu else _NullContext()

    def submit(self, seed: int) -> None:
        dist = self.dist
        b = self.n % 2
        with self._on(self.compute):
            if self.n >= 2:
                self.compute.wait_event(self.free[b])  # frame n-2 has left this buffer
            if self.split == "tiles":
                self.accum[b].zero_()
                self.r.render_tiles_device(self.w, self.h, seed, self.rank, self.world, self.accum[b].data_ptr(),
                                           self.compute.cuda_stream)
            else:  # every pixel is stored: no zeroing needed
                begin, count = sample_range(self.rank, self.world, self.spp)
                self.r.render_samples_device(self.w, self.h, seed, begin, count, self.accum[b].data_ptr(),
                                             self.compute.cuda_stream)
            self.rendered[b].record(self.compute)
        with self._on(self.comm):
            self.comm.wait_event(self.rendered[b])
            dist.reduce(self.accum[b], dst=0, op=dist.ReduceOp.SUM, group=self.group)
            if self.rank == 0:
                self.r.resolve_device(self.accum[b].data_ptr(), self.spp, self.w, self.h, self.rgba8.data_ptr(), 0,
                                      self.comm.cuda_stream)
            self.free[b].record(self.comm)
        self.n += 1

    def wait(self) -> None:
        self.compute.synchronize()
        self.comm.synchronize()


def cpu_pipeline(stand_in, width: int, height: int, spp: int, tensors: dict, group=None, split: str = "samples") -> FramePipeline:
    """A FramePipeline over CPU tensors for the gloo tests.  `stand_in` provides render_samples_device /
    render_tiles_device / resolve_device with the renderer's signatures; they receive data_ptr() values, and
    `tensors` (filled here) maps those back to the pipeline's buffers."""
    pipe = FramePipeline(stand_in, width, height, spp, "cpu", group=group, split=split)
    for t in pipe.accum + ([pipe.rgba8] if pipe.rgba8 is not None else []):
        tensors[t.data_ptr()] = t
    return pipe
