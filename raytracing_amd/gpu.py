"""ctypes binding of the C ABI in include/rt_gpu.h (librt_gpu.so).

There is no fallback: if the HIP library is missing or no GPU is visible the
calls raise.  Everything here is plumbing around the C entry points.
"""
import ctypes as C
import importlib.util
import os

import numpy as np

from . import scene_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_GPU_LIB", os.path.join(_HERE, "librt_gpu.so"))   # override: kernel-variant experiments

FLAG_DETAILED_STATS = 1
FLAG_ACCUMULATE = 2
FLAG_COMPACT_BANDS = 4
FLAG_KERNEL_TIMES = 8
KERNEL_KINDS = ("new_paths", "resumed_paths", "tree", "finish")

EXPORTS = [
    "rt_gpu_device_count", "rt_gpu_last_error", "rt_gpu_scene_create", "rt_gpu_scene_destroy",
    "rt_gpu_scene_set_camera", "rt_gpu_scene_info", "rt_gpu_render_region", "rt_gpu_render_region_device", "rt_gpu_render_region_static",
    "rt_gpu_normalize_device", "rt_gpu_render_jobs", "rt_gpu_find_ray_hit", "rt_gpu_find_ray_hit_serial", "rt_gpu_quantize", "rt_gpu_quantize_device",
    "rt_gpu_multi_create", "rt_gpu_multi_destroy", "rt_gpu_multi_device_count", "rt_gpu_multi_render_frame",
    "rt_gpu_multi_rows_per_rank", "rt_gpu_assemble_bands_device",
    "rt_gpu_get_stats", "rt_gpu_get_stats_by_kernel", "rt_gpu_reset_stats", "rt_gpu_get_stack_histogram", "rt_gpu_eval_math", "rt_gpu_eval_unit",
]


class Options(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "max_segments", "sample_offset", "spp_total", "flags",
        "band_rows", "band_rank", "band_count", "frames", "lane")] + [("frame_stride", C.c_uint64)]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(Options)
        for k, v in kw.items():
            setattr(self, k, v)


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "rays", "shadow_rays", "root_tests", "interior_pops", "leaf_pops",
        "tri_tests", "tri_hits", "tex_lookups", "rng_draws", "light_tests", "sphere_tests", "plane_tests",
        "shade_steps", "pop_steps", "tri_steps", "shade_lanes", "pop_lanes", "tri_lanes", "tree_entries")] + [
        ("kernel_ms", C.c_double), ("shade_ms", C.c_double), ("tree_ms", C.c_double), ("fresh_ms", C.c_double), ("finish_ms", C.c_double),
        ("rounds", C.c_uint64), ("overflowed", C.c_uint64), ("launches", C.c_uint32), ("passes", C.c_uint32),
        ("passes_with_work", C.c_uint32), ("lists_full", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("flatten_ms", C.c_double), ("upload_ms", C.c_double), ("resident_bytes", C.c_uint64), ("work_bytes", C.c_uint64),
                ("mesh_count", C.c_uint32), ("instance_count", C.c_uint32), ("texture_count", C.c_uint32), ("stack_levels", C.c_uint32)]


class GpuError(RuntimeError):
    pass


_LIB = None


def lib():
    """Load librt_gpu.so (raises if it was not built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GpuError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
        # A process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64 /
        # libhsa-runtime64; if this library pulled in /opt/rocm's copy first, torch would later find
        # no GPU.  So when torch is installed, let it load its runtime first: the dynamic linker then
        # resolves this library's libamdhip64.so dependency to the copy already in the process.
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.rt_gpu_device_count.restype = C.c_int
        L.rt_gpu_last_error.restype = C.c_char_p
        L.rt_gpu_scene_create.argtypes = [C.POINTER(T.Scene), C.POINTER(C.c_void_p)]
        L.rt_gpu_scene_destroy.argtypes = [C.c_void_p]
        L.rt_gpu_scene_destroy.restype = None
        L.rt_gpu_scene_set_camera.argtypes = [C.c_void_p, C.POINTER(T.Camera)]
        L.rt_gpu_scene_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        L.rt_gpu_render_region.argtypes = [C.c_void_p, C.POINTER(T.Image3f), T.Rect2u, C.c_uint32,
                                           C.c_uint64, C.POINTER(Options)]
        L.rt_gpu_render_region_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, T.Rect2u,
                                                  C.c_uint32, C.c_uint64, C.POINTER(Options), C.c_void_p]
        L.rt_gpu_render_region_static.argtypes = L.rt_gpu_render_region_device.argtypes
        L.rt_gpu_normalize_device.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.rt_gpu_render_jobs.argtypes = [C.c_void_p, C.POINTER(T.Job), C.c_uint32, C.POINTER(Options)]
        L.rt_gpu_find_ray_hit.argtypes = [C.c_void_p, C.POINTER(T.Ray), C.c_uint32, C.c_float,
                                          C.POINTER(T.RayResult)]
        L.rt_gpu_find_ray_hit_serial.argtypes = L.rt_gpu_find_ray_hit.argtypes
        L.rt_gpu_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.rt_gpu_quantize_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rt_gpu_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rt_gpu_multi_create.argtypes = [C.POINTER(T.Scene), C.c_int, C.POINTER(C.c_void_p)]
        L.rt_gpu_multi_destroy.argtypes = [C.c_void_p]
        L.rt_gpu_multi_destroy.restype = None
        L.rt_gpu_multi_device_count.argtypes = [C.c_void_p]
        L.rt_gpu_multi_render_frame.argtypes = [C.c_void_p, C.POINTER(T.Image3f), C.c_uint32, C.c_uint64, C.POINTER(Options)]
        L.rt_gpu_multi_rows_per_rank.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.rt_gpu_multi_rows_per_rank.restype = C.c_uint32
        L.rt_gpu_assemble_bands_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rt_gpu_get_stats_by_kernel.argtypes = [C.c_void_p, C.POINTER(Stats * 4)]
        L.rt_gpu_reset_stats.argtypes = [C.c_void_p]
        L.rt_gpu_get_stack_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 32)]
        L.rt_gpu_eval_math.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
        L.rt_gpu_eval_unit.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise GpuError(f"rt_gpu error {rc}: {lib().rt_gpu_last_error().decode()}")


def rect(x, y, w, h):
    r = T.Rect2u()
    r.point.x, r.point.y, r.size.x, r.size.y = x, y, w, h
    return r


class GpuScene:
    """A scene resident in HBM (rt_gpu_scene_create / rt_gpu_scene_destroy)."""

    def __init__(self, scene_handle):
        self._keep = scene_handle
        self._h = C.c_void_p()
        _check(lib().rt_gpu_scene_create(scene_handle.ptr, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rt_gpu_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = SceneInfo()
        _check(lib().rt_gpu_scene_info(self._h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in out._fields_}

    def render_region(self, image, region, spp, seed, options=None):
        """rt_gpu_render_region: host image in, host image out."""
        _check(lib().rt_gpu_render_region(self._h, C.byref(image), rect(*region), spp, seed,
                                          C.byref(options) if options is not None else None))

    def render_region_device(self, device_ptr, width, height, region, spp, seed, options=None, stream=0):
        """rt_gpu_render_region_device: asynchronous render into device memory."""
        _check(lib().rt_gpu_render_region_device(self._h, device_ptr, width, height, rect(*region), spp, seed,
                                                 C.byref(options) if options is not None else None,
                                                 C.c_void_p(stream)))

    def render_region_static(self, device_ptr, width, height, region, spp, seed, options=None, stream=0):
        """rt_gpu_render_region_static: the same render out of the library's static work memory (no allocation)."""
        _check(lib().rt_gpu_render_region_static(self._h, device_ptr, width, height, rect(*region), spp, seed,
                                                 C.byref(options) if options is not None else None,
                                                 C.c_void_p(stream)))

    def render_jobs(self, jobs, options=None):
        arr = (T.Job * len(jobs))(*jobs)
        _check(lib().rt_gpu_render_jobs(self._h, arr, len(jobs),
                                        C.byref(options) if options is not None else None))

    def find_ray_hit(self, rays, max_distance, serial=False):
        """rays: (n, 6) float32 -> ctypes array of RayResult.  serial=True: the one-lane-per-ray
        walk instead of the render path's kernels (same results, a second implementation)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = len(rays)
        out = (T.RayResult * n)()
        call = lib().rt_gpu_find_ray_hit_serial if serial else lib().rt_gpu_find_ray_hit
        _check(call(self._h, C.cast(rays.ctypes.data, C.POINTER(T.Ray)), n, max_distance, out))
        return out

    def stats(self):
        st = Stats()
        _check(lib().rt_gpu_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def stats_by_kernel(self):
        """{kind: counters} for the four kinds of launch (KERNEL_KINDS)."""
        st = (Stats * 4)()
        _check(lib().rt_gpu_get_stats_by_kernel(self._h, C.byref(st)))
        return {name: st[i].as_dict() for i, name in enumerate(KERNEL_KINDS)}

    def reset_stats(self):
        _check(lib().rt_gpu_reset_stats(self._h))

    def stack_histogram(self):
        """Tree walks by deepest traversal stack (entry k: never more than k entries; 31 = 31 or more), from the calls
        with FLAG_DETAILED_STATS and the closest-hit probes since the last reset."""
        out = (C.c_uint64 * 32)()
        _check(lib().rt_gpu_get_stack_histogram(self._h, C.byref(out)))
        return [int(v) for v in out]


class MultiGpuFrame:
    """rt_gpu_multi_*: one frame over `device_count` GPUs of the node from this process (0 = all visible)."""

    def __init__(self, scene_handle, device_count=0):
        self._keep = scene_handle
        self._h = C.c_void_p()
        _check(lib().rt_gpu_multi_create(scene_handle.ptr, device_count, C.byref(self._h)))

    @property
    def device_count(self):
        return lib().rt_gpu_multi_device_count(self._h)

    def render_frame(self, image, spp, seed, options=None):
        _check(lib().rt_gpu_multi_render_frame(self._h, C.byref(image), spp, seed,
                                               C.byref(options) if options is not None else None))

    def close(self):
        if self._h:
            lib().rt_gpu_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def assemble_bands_device(all_ptr, frame_ptr, width, height, band_rows, world, stream=0):
    _check(lib().rt_gpu_assemble_bands_device(all_ptr, frame_ptr, width, height, band_rows, world, C.c_void_p(stream)))


def normalize_device(device_ptr, pixel_count, spp_total, stream=0):
    _check(lib().rt_gpu_normalize_device(device_ptr, pixel_count, spp_total, C.c_void_p(stream)))


def quantize(pixels):
    pixels = np.ascontiguousarray(pixels, dtype=np.float32)
    out = np.zeros(pixels.shape, dtype=np.uint8)
    _check(lib().rt_gpu_quantize(pixels.ctypes.data, pixels.size // 3, out.ctypes.data))
    return out


def quantize_device(device_pixels_ptr, pixel_count, device_rgb_ptr, stream=0):
    """Device float RGB -> device RGB8 (the display transfer of the reference's save_image)."""
    _check(lib().rt_gpu_quantize_device(device_pixels_ptr, pixel_count, device_rgb_ptr, C.c_void_p(stream)))


def eval_unit(v):
    """Device normalisation of the rows of an (n, 3) float array: (unit vectors, lengths) (self-test hook)."""
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
    unit, length = np.zeros_like(v), np.zeros(len(v), dtype=np.float32)
    _check(lib().rt_gpu_eval_unit(v.ctypes.data, len(v), unit.ctypes.data, length.ctypes.data))
    return unit, length


def eval_math(x, y=None):
    """Device sin, cos, log, x/y, sqrt of float arrays (self-test hook)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.size
    outs = [np.zeros(n, dtype=np.float32) for _ in range(6)]
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float32)
        yp = y.ctypes.data
    _check(lib().rt_gpu_eval_math(x.ctypes.data, yp, n, outs[0].ctypes.data, outs[1].ctypes.data,
                                  outs[2].ctypes.data, outs[3].ctypes.data if y is not None else None,
                                  outs[4].ctypes.data, outs[5].ctypes.data if y is not None else None))
    return dict(sin=outs[0], cos=outs[1], log=outs[2], div=outs[3] if y is not None else None, sqrt=outs[4],
                div_cached=outs[5] if y is not None else None)
