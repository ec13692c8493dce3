"""world_size-2 (and 3) `gloo` runs of the multi-GPU frame path on CPU: tile ownership, the zero+integrate+
reduce+resolve sequence and rank-independence of the result.  The HIP integrator is replaced by a stand-in
that fills a rank's tiles from a frame the ORACLE rendered (the product cannot compute without a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyoracle
from helpers import benchmark_scene
from raytracingrust_amd import distributed as rd


def test_tile_ownership_partitions_the_frame():
    for (w, h) in [(1920, 1080), (97, 61), (8, 8), (7, 3), (64, 9)]:
        for n in [1, 2, 3, 4, 8]:
            masks = [rd.owned_pixel_mask(w, h, r, n) for r in range(n)]
            total = np.sum(np.stack(masks).astype(np.int32), axis=0)
            assert np.all(total == 1), (w, h, n)  # every pixel exactly once
            if w * h >= 64 * n * 4:
                sizes = [int(m.sum()) for m in masks]
                assert max(sizes) - min(sizes) <= 64 * (rd.tile_grid(w, h)[0] + 2)
    m = rd.owned_pixel_mask(24, 16, 1, 2)  # tiles row-major: 0 1 2 / 3 4 5 -> rank 1 owns 1, 3, 5
    assert m[0, 8] and not m[0, 0] and m[8, 0] and not m[8, 8] and m[8, 16]
    with pytest.raises(ValueError):
        rd.owned_pixel_mask(8, 8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, frame_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = np.load(frame_path)  # [h, w, 3] float32 oracle sums

    def integrate(width, height, seed, shard, n_shards, accum):
        mask = torch.from_numpy(rd.owned_pixel_mask(width, height, shard, n_shards))
        src = torch.from_numpy(frame).clone()  # [H, W, 3]: packed RGB partial sums
        accum[mask] = src[mask]  # the kernel STOREs its pixels and touches no others

    def resolve(accum, spp, rgba8):
        mean = accum[..., :3] / float(spp)
        rgba8[..., :3] = torch.clamp(torch.sqrt(mean) * 255.0, 0, 255).to(torch.uint8)
        rgba8[..., 3] = 255

    accum = torch.full((h, w, 3), 123.0)  # garbage: render_frame_distributed must zero it
    rgba8 = torch.zeros((h, w, 4), dtype=torch.uint8)
    for _ in range(2):  # twice: the buffer is re-zeroed every frame
        rd.render_frame_distributed(None, w, h, 7, accum, rgba8 if rank == 0 else None, integrate=integrate,
                                    spp=1, resolve=resolve)
    if rank == 0:
        np.savez(out_path, accum=accum.numpy(), rgba8=rgba8.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_frame_equals_single_frame(tmp_path, world):
    w, h = 97, 61
    scene = benchmark_scene(2)
    _, mean, _ = pyoracle.OracleScene(scene).render(w, h, 7, threads=0)
    frame_path = str(tmp_path / "frame.npy")
    out_path = str(tmp_path / "out.npz")
    np.save(frame_path, mean)
    mp.spawn(_worker, args=(world, _free_port(), w, h, frame_path, out_path), nprocs=world, join=True)
    out = np.load(out_path)
    assert np.array_equal(out["accum"][..., :3].view(np.uint32), mean.view(np.uint32))  # bit-identical: x + 0 = x
    assert out["accum"].shape[-1] == 3  # three floats per pixel on the wire: no padding word
    assert np.all(out["rgba8"][..., 3] == 255)


def test_sample_ranges_partition_the_samples():
    for spp in [1, 5, 13, 256, 1024]:
        for n in [1, 2, 3, 4, 8]:
            ranges = [rd.sample_range(r, n, spp) for r in range(n)]
            assert ranges[0][0] == 0 and sum(c for _, c in ranges) == spp
            for (b0, c0), (b1, _) in zip(ranges, ranges[1:]):
                assert b1 == b0 + c0
            counts = [c for _, c in ranges]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        rd.sample_range(3, 3, 8)


# ---- the DEFAULT decomposition: samples of every pixel split over the ranks (FramePipeline / bench.py) -------------
class _StandInRenderer:
    """The two entry points FramePipeline calls, computed on the CPU from per-sample radiance the ORACLE rendered:
    render_samples_device STOREs the in-order partial sum of the rank's sample range for EVERY pixel, resolve_device is
    render.rs:103-110.  (The product cannot compute without a GPU; only the sharding and the collective are under test.)"""

    def __init__(self, per_sample, tensors):
        self.per_sample = per_sample  # [spp, h, w, 3] float32: each sample's clamped radiance
        self.tensors = tensors        # data_ptr -> tensor
        self.calls = []

    def render_samples_device(self, w, h, seed, begin, count, ptr, stream=0):
        acc = np.zeros(self.per_sample.shape[1:], dtype=np.float32)
        for s in range(begin, begin + count):  # in sample order, as the kernel adds them
            acc = (acc + self.per_sample[s]).astype(np.float32)
        t = self.tensors[ptr]
        t[...] = torch.from_numpy(acc)  # [H, W, 3]: packed RGB partial sums
        self.calls.append((begin, count))

    def resolve_device(self, sum_ptr, spp, w, h, rgba_ptr, mean_ptr=0, stream=0):
        acc = self.tensors[sum_ptr]
        mean = acc[..., :3] / float(spp)
        out = self.tensors[rgba_ptr]
        out[..., :3] = torch.clamp(torch.sqrt(mean) * 255.0, 0, 255).to(torch.uint8)
        out[..., 3] = 255


def _sample_worker(rank, world, port, w, h, spp, samples_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per_sample = np.load(samples_path)
    tensors = {}
    stand_in = _StandInRenderer(per_sample, tensors)
    pipe = rd.cpu_pipeline(stand_in, w, h, spp, tensors, split="samples")
    for frame in range(3):  # three frames through the two alternating sum buffers
        pipe.submit(seed=frame)
    pipe.wait()
    assert stand_in.calls == [rd.sample_range(rank, world, spp)] * 3
    if rank == 0:
        np.savez(out_path, accum=pipe.accum[0].numpy(), rgba8=pipe.rgba8.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sample_split_pipeline_sums_partial_sums(tmp_path, world):
    """FramePipeline's default split over gloo: every rank integrates all pixels for its sample range, reduce(SUM) on rank
    0 equals the sum of the ranks' partial sums exactly (same additions in rank order for 2 ranks; any order within 1e-6
    relative in general) and the single-rank in-order sum within float-addition reordering; then resolve."""
    w, h, spp = 33, 17, 13
    scene = benchmark_scene(1)
    per_sample = np.zeros((spp, h, w, 3), dtype=np.float32)
    for s in range(spp):  # one-sample frames with different seeds stand in for the samples of one frame
        _, mean, _ = pyoracle.OracleScene(scene).render(w, h, 100 + s, threads=0)
        per_sample[s] = mean
    samples_path, out_path = str(tmp_path / "samples.npy"), str(tmp_path / "out.npz")
    np.save(samples_path, per_sample)
    mp.spawn(_sample_worker, args=(world, _free_port(), w, h, spp, samples_path, out_path), nprocs=world, join=True)
    out = np.load(out_path)
    partial = []
    for r in range(world):
        b, c = rd.sample_range(r, world, spp)
        acc = np.zeros((h, w, 3), dtype=np.float32)
        for s in range(b, b + c):
            acc = (acc + per_sample[s]).astype(np.float32)
        partial.append(acc)
    single = np.zeros((h, w, 3), dtype=np.float32)
    for s in range(spp):
        single = (single + per_sample[s]).astype(np.float32)
    got = out["accum"][..., :3]
    ref = np.sum(np.stack(partial).astype(np.float64), axis=0)
    assert np.allclose(got, ref, rtol=1e-6, atol=0)  # the reduce is a sum of the partial sums
    if world == 2:
        assert np.array_equal(got, (partial[0] + partial[1]).astype(np.float32))
    assert np.allclose(got, single, rtol=1e-5, atol=1e-7)  # only the order of the float additions differs from one rank
    assert out["accum"].shape[-1] == 3 and np.all(out["rgba8"][..., 3] == 255)
    want8 = np.clip(np.sqrt(got / np.float32(spp)) * 255.0, 0, 255).astype(np.uint8)
    assert np.max(np.abs(out["rgba8"][..., :3].astype(int) - want8.astype(int))) <= 1
