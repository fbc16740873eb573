"""Loaders for the checkers: the CPU oracle (oracle/librt_oracle*.so) and, where it
was built, the compiled reference (oracle/_ref/libref_*.so).  Test-only."""
import ctypes as C
import os

import numpy as np

from raytracing_amd import scene_types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

RNG_AES, RNG_COUNTER = 0, 1
MATH_LIBM, MATH_PORTABLE = 0, 1


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "rays", "shadow_rays", "root_tests", "interior_pops", "leaf_pops",
        "tri_tests", "tri_hits", "tex_lookups", "rng_draws")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_cache = {}


def oracle(contract=False):
    key = "oracle_contract" if contract else "oracle"
    if key not in _cache:
        path = os.path.join(ORACLE_DIR, "librt_oracle_contract.so" if contract else "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle oracle`")
        L = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        L.orc_render_region_aes.argtypes = [C.POINTER(T.Job), C.c_int, C.c_int, C.POINTER(Stats)]
        L.orc_render_region_aes.restype = None
        L.orc_render_region_counter.argtypes = [
            C.POINTER(T.Scene), C.POINTER(T.Image3f), C.POINTER(T.Rect2u), C.c_uint32, C.c_uint32,
            C.c_uint64, C.c_int, C.c_int, C.POINTER(Stats)]
        L.orc_render_region_counter.restype = None
        L.orc_find_ray_hit.argtypes = [C.POINTER(T.Scene), C.POINTER(T.Ray), C.c_float,
                                       C.POINTER(T.RayResult), C.POINTER(Stats)]
        L.orc_find_ray_hit.restype = None
        L.orc_aes_uniform4.argtypes = [C.POINTER(T.RandomSerie), C.c_float, C.c_float, C.POINTER(C.c_float * 4)]
        L.orc_aes_uniform4.restype = None
        L.orc_aes_gaussian2.argtypes = [C.POINTER(T.RandomSerie), C.c_float, C.c_float, C.c_int,
                                        C.POINTER(C.c_float * 2)]
        L.orc_aes_gaussian2.restype = None
        L.orc_threefry2x32_20.argtypes = [C.POINTER(C.c_uint32 * 2), C.POINTER(C.c_uint32 * 2),
                                          C.POINTER(C.c_uint32 * 2)]
        L.orc_threefry2x32_20.restype = None
        L.orc_counter_uniform2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_float * 2)]
        L.orc_counter_uniform2.restype = None
        L.orc_direct_light_aes.argtypes = [C.POINTER(T.Job), C.POINTER(T.RayResult), C.c_int,
                                           C.POINTER(C.c_float * 4), C.POINTER(T.RandomSerie)]
        L.orc_direct_light_aes.restype = None
        L.orc_trace_ray_aes.argtypes = [C.POINTER(T.Job), C.POINTER(T.Ray), C.c_int, C.c_int,
                                        C.POINTER(C.c_float * 3), C.POINTER(T.RandomSerie)]
        L.orc_trace_ray_aes.restype = None
        L.orc_sample_texture.argtypes = [C.POINTER(T.Image3f), C.POINTER(T.Vec2f), C.POINTER(C.c_float * 3)]
        L.orc_sample_texture.restype = None
        L.orc_sincosf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_sincosf.restype = None
        L.orc_logf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_logf.restype = None
        L.orc_quantize.argtypes = [C.POINTER(T.Image3f), C.c_void_p]
        L.orc_quantize.restype = None
        L.orc_quantize_portable.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_quantize_portable.restype = None
        L.orc_render_tiles.argtypes = [
            C.POINTER(T.Scene), C.POINTER(T.Image3f), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
            C.c_int, C.c_int, C.POINTER(T.RandomSerie), C.c_uint64, C.POINTER(Stats)]
        L.orc_render_tiles.restype = C.c_double
        _cache[key] = L
    return _cache[key]


class RefObjArrays(C.Structure):
    _fields_ = [
        ("position_count", C.c_uint32), ("positions", C.POINTER(C.c_float)),
        ("texcoord_count", C.c_uint32), ("texcoords", C.POINTER(C.c_float)),
        ("normal_count", C.c_uint32), ("normals", C.POINTER(C.c_float)),
        ("face_count", C.c_uint32), ("face_vertices", C.POINTER(C.c_uint32)),
        ("index_count", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
    ]


def have_ref(variant="strict"):
    return os.path.exists(os.path.join(REF_DIR, f"libref_{variant}.so"))


def ref(variant="strict"):
    """variant: 'build' (the reference's exact flags), 'strict' (+ -ffp-contract=off),
    'strict_pm' (strict, with cosf/sinf/logf taken from rt_math_portable.h)."""
    key = "ref_" + variant
    if key not in _cache:
        path = os.path.join(REF_DIR, f"libref_{variant}.so")
        L = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        L.ref_sizes.argtypes = [C.POINTER(C.c_uint32 * 16)]
        L.ref_raytrace_region.argtypes = [C.POINTER(T.Job)]
        L.ref_raytrace_region.restype = None
        L.ref_find_ray_hit.argtypes = [C.POINTER(T.Scene), C.POINTER(T.Ray), C.c_float, C.POINTER(T.RayResult)]
        L.ref_find_ray_hit.restype = None
        L.ref_compute_light_contribution.argtypes = [C.POINTER(T.Job), C.POINTER(T.Scene), C.POINTER(T.Ray),
                                                     C.POINTER(T.RayResult), C.POINTER(C.c_float * 4)]
        L.ref_compute_light_contribution.restype = None
        L.ref_raytrace_ray.argtypes = [C.POINTER(T.Job), C.POINTER(T.Scene), C.POINTER(T.Ray),
                                       C.POINTER(C.c_float * 3)]
        L.ref_raytrace_ray.restype = None
        L.ref_create_random_serie.argtypes = [C.POINTER(T.RandomSerie)]
        L.ref_random4f_uniform.argtypes = [C.POINTER(T.RandomSerie), C.c_float, C.c_float, C.POINTER(C.c_float * 4)]
        L.ref_random4f_uniform.restype = None
        L.ref_random2f_gaussian.argtypes = [C.POINTER(T.RandomSerie), C.c_float, C.c_float, C.POINTER(C.c_float * 2)]
        L.ref_random2f_gaussian.restype = None
        L.ref_sample_texture.argtypes = [C.POINTER(T.Image3f), C.POINTER(T.Vec2f), C.POINTER(C.c_float * 3)]
        L.ref_sample_texture.restype = None
        L.ref_load_mesh.argtypes = [C.c_char_p, C.POINTER(T.Mesh)]
        L.ref_load_mesh.restype = None
        L.ref_load_texture.argtypes = [C.c_char_p, C.POINTER(T.Image3f)]
        L.ref_load_texture.restype = None
        L.ref_quantize_image.argtypes = [C.POINTER(T.Image3f), C.c_char_p, C.c_void_p]
        L.ref_quantize_image.restype = C.c_int
        L.ref_fast_obj_read.argtypes = [C.c_char_p, C.POINTER(RefObjArrays)]
        L.ref_fast_obj_read.restype = C.c_int
        L.ref_stbi_load_rgb.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_stbi_load_rgb.restype = C.POINTER(C.c_uint8)
        L.ref_bvh_census.argtypes = [C.POINTER(T.Mesh), C.POINTER(C.c_uint32 * 4)]
        L.ref_bvh_census.restype = None
        L.ref_render_tiles.argtypes = [C.POINTER(T.Scene), C.POINTER(T.Image3f), C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.POINTER(T.RandomSerie)]
        L.ref_render_tiles.restype = C.c_double
        _cache[key] = L
    return _cache[key]


def serie_from_seed(seed):
    """A deterministic 32-byte generator state for tests."""
    s = T.RandomSerie()
    b = np.random.default_rng(seed).integers(0, 256, 32, dtype=np.uint8)
    for i in range(32):
        s.bytes[i] = int(b[i])
    return s


def make_job(scene_handle, image, region, spp, serie):
    job = T.Job()
    job.scene = scene_handle.ptr
    job.image = C.pointer(image)
    job.region.point.x, job.region.point.y, job.region.size.x, job.region.size.y = region
    job.ray_per_pixel = spp
    job.serie = serie
    return job


def same_bits(a, b):
    """Bitwise equality of float arrays, any NaN matching any NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return eq
