"""rt_gpu_render_region_static (the entry point that allocates nothing) against
rt_gpu_render_region_device, through the C ABI, bit for bit."""
import numpy as np
import pytest

import _libs as L
from raytracing_amd import gpu, scenes

pytestmark = pytest.mark.gpu


def test_static_work_memory_render_equals_the_allocating_one():
    """Two calls in a row on one GPU, the second smaller than the first (so it runs on work memory
    that still holds the first call's records and samples): a 64x64 tile at 16 samples per pixel of
    the dog + cat scene inside a larger image, then a ragged 21x13 region at 3 samples accumulated
    on top of 2 earlier ones; each equals rt_gpu_render_region_device's pixels bit for bit.  Then the
    limit: one sample more than the work memory holds is refused with a clear error."""
    import torch
    assert gpu.lib().rt_gpu_device_count() >= 1, "no GPU visible: this test has no CPU fallback"
    sh = scenes.dogs_and_cat()
    gs = gpu.GpuScene(sh)
    W, H, seed = 200, 160, 91
    stream = torch.cuda.current_stream().cuda_stream

    def both(region, spp, options_factory):
        want = torch.full((H, W, 3), -3.0, dtype=torch.float32, device="cuda")
        got = torch.full((H, W, 3), -3.0, dtype=torch.float32, device="cuda")
        if options_factory is not None and (options_factory().flags & gpu.FLAG_ACCUMULATE):
            want.zero_(); got.zero_()
            first = gpu.Options(flags=gpu.FLAG_ACCUMULATE, sample_offset=0, spp_total=5)
            gs.render_region_device(want.data_ptr(), W, H, region, 2, seed, first, stream)
            gs.render_region_static(got.data_ptr(), W, H, region, 2, seed, first, stream)
        gs.render_region_device(want.data_ptr(), W, H, region, spp, seed, options_factory() if options_factory else None, stream)
        gs.render_region_static(got.data_ptr(), W, H, region, spp, seed, options_factory() if options_factory else None, stream)
        torch.cuda.synchronize()
        return got.cpu().numpy(), want.cpu().numpy()

    got, want = both((72, 56, 64, 64), 16, None)                      # 64 blocks x 64 x 16 = the whole work memory
    assert L.same_bits(got, want).all(), int((~L.same_bits(got, want)).sum())
    assert (got[:56] == -3.0).all() and np.isfinite(got[56:120, 72:136]).all() and got[56:120, 72:136].mean() > 0

    got, want = both((101, 77, 21, 13), 3, lambda: gpu.Options(flags=gpu.FLAG_ACCUMULATE, sample_offset=2, spp_total=5))
    assert L.same_bits(got, want).all(), int((~L.same_bits(got, want)).sum())
    assert got[77:90, 101:122].sum() > 0

    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(gpu.GpuError, match="exceed the static work memory"):
        gs.render_region_static(buf.data_ptr(), W, H, (72, 56, 64, 64), 17, seed, None, stream)
    with pytest.raises(gpu.GpuError, match="one frame per call"):
        gs.render_region_static(buf.data_ptr(), W, H, (0, 0, 8, 8), 1, seed, gpu.Options(frames=2), stream)
    gs.close()


def test_static_entry_allocates_nothing_and_can_be_captured_in_a_graph():
    """The claim itself, not only equal pixels: (a) the free device memory (hipMemGetInfo) is the same before
    and after a warm call of rt_gpu_render_region_static, while the allocating entry's first call on a fresh
    handle visibly takes memory; (b) the call is recorded into a HIP graph (stream capture forbids
    allocation, synchronisation and read-back: any of them would make the capture fail) and the graph,
    replayed twice into a cleared buffer, gives the pixels of rt_gpu_render_region_device bit for bit."""
    import torch
    assert gpu.lib().rt_gpu_device_count() >= 1, "no GPU visible: this test has no CPU fallback"
    sh = scenes.three_dogs()
    gs = gpu.GpuScene(sh)
    W, H, seed, spp = 96, 80, 7, 6
    region = (16, 8, 64, 64)
    got = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    want = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gs.render_region_static(got.data_ptr(), W, H, region, spp, seed, None, stream)       # warm: code objects loaded
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        gs.render_region_static(got.data_ptr(), W, H, region, spp, seed, None, stream)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, (free_before, free_after)
    gs.render_region_device(want.data_ptr(), W, H, region, spp, seed, None, stream)       # this one allocates its work memory
    torch.cuda.synchronize()
    free_device_entry, _ = torch.cuda.mem_get_info()
    assert free_device_entry < free_after

    def differing():
        return int((~L.same_bits(got.cpu().numpy(), want.cpu().numpy())).sum())
    assert differing() == 0

    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    got.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        gs.render_region_static(got.data_ptr(), W, H, region, spp, seed, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (got == 0).all()                       # captured, not run
    gs.reset_stats()
    for _ in range(3):
        got.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert differing() == 0
    st = gs.stats()                                # raises if a replay lost a path to a full list
    assert st["overflowed"] == 0 and st["rays"] > 0 and st["rays"] % 3 == 0      # three equal renders really ran
    del graph
    gs.close()
