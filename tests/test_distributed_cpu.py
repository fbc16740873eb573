"""The multi-device split on two CPU processes over gloo: each rank fills its packed band buffer
(rendered by the oracle here, by the HIP kernel in production), one all-gather combines them, and
the assembled frame must equal a single-process render bit for bit."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import _libs as L
from raytracing_amd import host, multi_gpu, scenes, scene_types as T


def oracle_rows(sh, width, height, rows, spp, seed):
    """Counter-mode render of the given image rows, packed in order."""
    img, arr = host.new_image(width, height)
    for y in rows:
        region = T.Rect2u()
        region.point.x, region.point.y, region.size.x, region.size.y = 0, y, width, 1
        L.oracle().orc_render_region_counter(sh.ptr, C.byref(img), C.byref(region), spp, 0, seed,
                                             L.MATH_PORTABLE, 64, None)
    return arr[rows].copy()


def _worker(rank, world, port, width, height, spp, seed, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = scenes.sphere_room()
    rows = multi_gpu.rows_of_rank(rank, world, height)
    per_rank = multi_gpu.packed_rows(world, height)
    assert len(rows) <= per_rank
    own = torch.from_numpy(oracle_rows(sh, width, height, rows, spp, seed))
    mine = torch.full((per_rank,) + tuple(own.shape[1:]), -7.0, dtype=torch.float32)     # padding rows must not reach the frame
    # a rank's bands sit at multiples of the band height; a short last band leaves the rest of its band empty
    for k in range(0, len(rows), multi_gpu.BAND_ROWS):
        band_rows = [y for y in rows if y // multi_gpu.BAND_ROWS == rows[k] // multi_gpu.BAND_ROWS]
        at = (rows[k] // multi_gpu.BAND_ROWS // world) * multi_gpu.BAND_ROWS
        mine[at:at + len(band_rows)] = own[k:k + len(band_rows)]
    gathered = torch.zeros((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, mine)
    frame = multi_gpu.assemble(gathered, world, height, width).numpy().copy()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, frame))


@pytest.mark.parametrize("world,height", [(2, 32), (3, 44)])
def test_band_split_all_gather(world, height):
    """two ranks with an even split (the assembly is a view permutation) and three ranks over a height that
    is no multiple of anything (ragged last band, a rank with one band fewer: assembly by row index)"""
    import torch.multiprocessing as mp
    width, spp, seed = 24, 2, 77
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, spp, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sh = scenes.sphere_room()
    whole = oracle_rows(sh, width, height, list(range(height)), spp, seed)
    for r in range(world):
        assert L.same_bits(frames[r], whole).all()


def test_rows_of_rank_partition():
    for world in (1, 2, 4, 8):
        for height in (64, 1024, 4096):
            seen = sorted(y for r in range(world) for y in multi_gpu.rows_of_rank(r, world, height))
            assert seen == list(range(height))
            assert all(len(multi_gpu.rows_of_rank(r, world, height)) == height // world for r in range(world))
            assert multi_gpu.packed_rows(world, height) == height // world
    for world, height, want in ((3, 64, 24), (3, 61, 24), (5, 8, 8), (8, 4096 + 3, 520)):
        seen = sorted(y for r in range(world) for y in multi_gpu.rows_of_rank(r, world, height))
        assert seen == list(range(height))
        assert multi_gpu.packed_rows(world, height) == want
        assert max(len(multi_gpu.rows_of_rank(r, world, height)) for r in range(world)) <= want


def _sum_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.rand((6, 5, 3), generator=g, dtype=torch.float32) * 37.0
    gathered = torch.zeros((world * 6, 5, 3), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, mine)
    frame = multi_gpu.combine_partial_sums(gathered, world, 64).numpy().copy()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, frame))


def test_sample_split_partial_sums_three_ranks():
    """Split by samples: partial sums gathered and added in rank order give every rank the same bits,
    equal to the sum taken in that order by hand, times the reference's final factor."""
    import torch
    import torch.multiprocessing as mp
    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = [(torch.rand((6, 5, 3), generator=torch.Generator().manual_seed(100 + r), dtype=torch.float32) * 37.0).numpy()
             for r in range(world)]
    want = ((parts[0] + parts[1]) + parts[2]) * multi_gpu.pixel_norm(64)
    for r in range(world):
        assert L.same_bits(frames[r], want.astype(np.float32)).all()
    assert [multi_gpu.sample_range(r, 3, 64) for r in range(3)] == [(0, 22), (22, 21), (43, 21)]
    assert sum(multi_gpu.sample_range(r, 8, 64)[1] for r in range(8)) == 64
    assert multi_gpu.pixel_norm(64) == np.float32(1.0 / (0.5 * np.pi * 64))


def test_c_side_band_layout_matches_the_python_split():
    """rt_gpu_multi_rows_per_rank (the packed buffer height the C multi-GPU entry gathers) against
    rows_of_rank for even and ragged heights; needs no GPU (pure arithmetic in the library)."""
    from raytracing_amd import gpu, multi_gpu
    lib = gpu.lib()
    for height in (8, 61, 64, 128, 1000, 1024, 4096):
        for world in (1, 2, 3, 4, 8):
            rows = lib.rt_gpu_multi_rows_per_rank(height, 8, world)
            owned = [len(multi_gpu.rows_of_rank(r, world, height)) for r in range(world)]
            assert rows >= max(owned) and rows % 8 == 0 and rows - max(owned) < 8 + 8 * (max(owned) % 8 != 0), (height, world, rows, owned)
            if height % (8 * world) == 0:
                assert rows == multi_gpu.packed_rows(world, height)
