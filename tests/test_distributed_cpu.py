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
        src = torch.zeros((height, width, 4), dtype=torch.float32)
        src[..., :3] = torch.from_numpy(frame)
        accum[mask] = src[mask]  # the kernel STOREs its pixels and touches no others

    def resolve(accum, spp, rgba8):
        mean = accum[..., :3] / float(spp)
        rgba8[..., :3] = torch.clamp(torch.sqrt(mean) * 255.0, 0, 255).to(torch.uint8)
        rgba8[..., 3] = 255

    accum = torch.full((h, w, 4), 123.0)  # garbage: render_frame_distributed must zero it
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
    assert np.all(out["accum"][..., 3] == 0)
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
