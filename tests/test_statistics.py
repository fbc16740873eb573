"""Monte Carlo agreement between random streams and builds (CPU only).

The GPU path draws from the counter generator, the reference from its serial AES generator, and
the reference's own build flags let gcc fuse multiply-adds: single samples cannot agree, the
converged picture must.  Bounds: block means (16x16 pixels, >= 64 spp) within 8 % median /
50 % worst relative difference of the reference's 256-spp block means."""
import ctypes as C

import numpy as np
import pytest

import _libs as L
from raytracing_amd import host, scenes, scene_types as T


@pytest.mark.parametrize("name", ["deux_spheres", "sphere_room", "diffuse_room", "dog_room", "three_dogs"])
def test_counter_stream_converges_to_the_reference_picture(golden, name):
    g = golden("converged_stats")
    size, tile = int(g["size"]), int(g["tile"])
    spp = 64
    sh = scenes.CONFIGS[name]()
    img, arr = host.new_image(size, size)
    L.oracle().orc_render_tiles(sh.ptr, C.byref(img), spp, tile, 8, L.RNG_COUNTER, L.MATH_PORTABLE, 64,
                                None, 123, None)
    mine = np.nan_to_num(arr).reshape(size // tile, tile, size // tile, tile, 3).mean(axis=(1, 3))
    want = g[name + "_block_means"]
    rel = np.abs(mine - want) / (np.abs(want) + 1e-2)
    assert np.median(rel) < 0.08, float(np.median(rel))
    assert rel.max() < 0.5, float(rel.max())
    psnr = 10 * np.log10(want.max() ** 2 / np.mean((mine - want) ** 2))
    assert psnr > 25.0, psnr


def rmse(x, y):
    return float(np.sqrt(np.mean((np.nan_to_num(x).astype(np.float64) - np.nan_to_num(y).astype(np.float64)) ** 2)))


def bytes_of(arr):
    """save_image's 8-bit conversion (src/main.c:73-82) with this machine's powf."""
    img, buf = host.new_image(arr.shape[1], arr.shape[0])
    buf[:] = arr
    out = np.zeros(arr.size, np.uint8)
    L.oracle().orc_quantize(C.byref(img), out.ctypes.data)
    return out.reshape(arr.shape).astype(np.float64)


PAIR_RATIO = 1.25      # error against a reference run <= 1.25 x the distance between two reference runs


def check_against_reference_pair(mine, a, b, what):
    """Monte Carlo bound at equal sample count: `mine` is as far from either reference run
    (build.sh flags, independent seeds) as those two are from each other, within PAIR_RATIO, per pixel
    on the float buffer and on the 8-bit picture; the PSNR follows from the same numbers."""
    spread = rmse(a, b)
    spread8 = rmse(bytes_of(a), bytes_of(b))
    for ref in (a, b):
        assert rmse(mine, ref) <= PAIR_RATIO * spread, (what, rmse(mine, ref), spread)
        assert rmse(bytes_of(mine), bytes_of(ref)) <= PAIR_RATIO * spread8, (what, "8 bit", rmse(bytes_of(mine), bytes_of(ref)), spread8)
    peak = 255.0
    psnr_mine = 20 * np.log10(peak / max(1e-9, rmse(bytes_of(mine), bytes_of(a))))
    psnr_ref = 20 * np.log10(peak / max(1e-9, spread8))
    assert psnr_mine >= psnr_ref - 20 * np.log10(PAIR_RATIO)
    assert abs(np.nanmean(mine) - np.nanmean(a)) <= 3 * abs(np.nanmean(a) - np.nanmean(b)) + 0.01 * np.nanmean(a)
    return psnr_mine, psnr_ref


@pytest.mark.parametrize("name", sorted(scenes.CONFIGS))
def test_counter_stream_at_equal_sample_count_is_within_the_references_own_spread(golden, name):
    """The oracle in the product's mode (counter generator, portable sin/cos/log, single rounding)
    against two runs of the reference at its build.sh flags, same size (96 x 96), same sample count
    (256 per pixel).  The GPU renders the same comparison in tests/test_gpu_parity.py."""
    g = golden("converged_pairs")
    size, spp, tile = int(g["size"]), int(g["spp"]), int(g["tile"])
    sh = scenes.CONFIGS[name]()
    img, arr = host.new_image(size, size)
    L.oracle().orc_render_tiles(sh.ptr, C.byref(img), spp, tile, 8, L.RNG_COUNTER, L.MATH_PORTABLE, 64, None, 4242, None)
    check_against_reference_pair(arr, g[name + "_a"], g[name + "_b"], name)


def test_counter_mode_is_region_independent():
    """The same pixels come out whether a frame is rendered whole or in arbitrary pieces."""
    sh = scenes.sphere_room()
    W, H, spp, seed = 24, 20, 2, 5
    whole_img, whole = host.new_image(W, H)
    r = T.Rect2u()
    r.size.x, r.size.y = W, H
    L.oracle().orc_render_region_counter(sh.ptr, C.byref(whole_img), C.byref(r), spp, 0, seed, L.MATH_PORTABLE, 64, None)
    img, got = host.new_image(W, H)
    for (x, y, w, h) in [(0, 0, 7, 20), (7, 0, 17, 3), (7, 3, 17, 17)]:
        r.point.x, r.point.y, r.size.x, r.size.y = x, y, w, h
        L.oracle().orc_render_region_counter(sh.ptr, C.byref(img), C.byref(r), spp, 0, seed, L.MATH_PORTABLE, 64, None)
    assert L.same_bits(got, whole).all()


def test_rays_per_sample_matches_the_survey():
    """About 7.7 closest-hit queries per sample in the room scenes (camera + bounce + shadow rays)."""
    st = L.Stats()
    sh = scenes.dog_room()
    img, _ = host.new_image(64, 64)
    L.oracle().orc_render_tiles(sh.ptr, C.byref(img), 4, 16, 8, L.RNG_COUNTER, L.MATH_PORTABLE, 64, None, 1, C.byref(st))
    assert 7.0 < st.rays / st.samples < 8.5
    assert 1.0 < st.interior_pops / st.rays < 2.0
