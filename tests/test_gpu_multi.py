"""rtr_create_multi on the one-GPU box: n = 1 goes through RCCL (a one-rank ncclReduce), {0, 0} / {0, 0, 0} put every rank
on the same GPU (the library then adds the partial sums with a kernel instead of ncclReduce: RCCL cannot hold two ranks
on one device) -- same threads, streams, double buffering, sample / tile arithmetic and resolve as on a node of GPUs."""
import numpy as np
import pytest

import raytracingrust_amd as rt
from helpers import benchmark_scene, house_scene, simple_scene

pytestmark = pytest.mark.gpu


def _single(scene, w, h, seed):
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        rgba, mean = r.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
        return rgba, mean, int(r.stats().segments)


@pytest.mark.parametrize("builder,w,h", [(lambda: benchmark_scene(13), 97, 61), (lambda: simple_scene(6), 64, 48),
                                         (lambda: house_scene(5), 80, 45)])
def test_one_rank_through_rccl_is_the_single_context_frame(builder, w, h):
    scene = builder()
    rgba0, mean0, seg0 = _single(scene, w, h, 77)
    with rt.Renderer([0]) as r:
        r.upload_scene(scene)
        rgba, mean = r.render(w, h, seed=77, want_rgba8=True, want_mean=True)
        assert int(r.stats().segments) == seg0
    assert np.array_equal(rgba, rgba0) and np.array_equal(mean.view(np.uint32), mean0.view(np.uint32))


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("builder,w,h", [(lambda: benchmark_scene(13), 97, 61), (lambda: house_scene(5), 80, 45)])
def test_ranks_on_one_device_reproduce_the_frame(builder, w, h, n):
    scene = builder()
    rgba0, mean0, seg0 = _single(scene, w, h, 5)
    with rt.Renderer([0] * n) as r:
        r.upload_scene(scene)
        rgba, mean = r.render(w, h, seed=5, want_rgba8=True, want_mean=True)  # samples of every pixel split over the ranks
        assert int(r.stats().segments) == seg0  # the ranks draw exactly the single-GPU numbers
        rel = float(np.linalg.norm(mean.astype(np.float64) - mean0) / np.linalg.norm(mean0.astype(np.float64)))
        assert rel <= 1e-6 and int(np.max(np.abs(rgba.astype(int) - rgba0.astype(int)))) <= 1
        r.set_option("multi_split", "tiles")  # 8x8 tiles round-robin: x + 0 = x, bit-identical
        rgba, mean = r.render(w, h, seed=5, want_rgba8=True, want_mean=True)
        assert int(r.stats().segments) == seg0
        assert np.array_equal(rgba, rgba0) and np.array_equal(mean.view(np.uint32), mean0.view(np.uint32))


def test_pipelined_frames_keep_their_buffers_apart():
    """rtr_render_frame_device on a multi context is asynchronous and double buffered: six frames issued back to back,
    each into its own output, must each equal the single-context frame of its seed."""
    import torch

    scene = benchmark_scene(9)
    w, h = 96, 64
    want = [_single(scene, w, h, 100 + i)[0] for i in range(6)]
    outs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(6)]
    with rt.Renderer([0, 0]) as r:
        r.upload_scene(scene)
        for i in range(6):
            r.render_frame_device(w, h, 100 + i, outs[i].data_ptr(), 0)
        seg = r.segment_counts()[-6:]
        torch.cuda.synchronize()
        for i in range(6):
            assert int(np.max(np.abs(outs[i].cpu().numpy().astype(int) - want[i].astype(int)))) <= 1, i
        assert len(seg) == 6 and all(int(s) > 0 for s in seg)
        assert len(r.kernel_times_ms()) >= 6


def test_device_fault_of_a_rank_is_reported():
    scene = benchmark_scene(4)
    with rt.Renderer([0, 0]) as r:
        r.upload_scene(scene)
        r.set_option("debug_fault", "on")
        with pytest.raises(rt.RtrError) as e:
            r.render(32, 32, seed=1)
        assert e.value.code == -3 and "rank" in str(e.value)
        r.set_option("debug_fault", "off")
        r.render(32, 32, seed=1)
