#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference compiled out of its own sources.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (this container):
`make -C oracle ref` builds oracle/_ref/libref_{build,strict,strict_pm}.so from
/root/reference/src/main.c, and this script drives those libraries through the wrappers of
oracle/ref_harness.c to record small input/output vectors.  The fixtures are data only.

Variants of the compiled reference:
  build      exactly the flags of the reference's build.sh (gcc may fuse multiply-adds)
  strict     the same plus -ffp-contract=off (every float operation rounds once)
  strict_pm  strict, and cosf/sinf/logf come from rt_math_portable.h instead of glibc

The dog model sits in the reference as a zip; it is unpacked to a scratch directory outside the
repository, parsed by the reference's own fast_obj and stored as arrays (dog_obj.npz); the
texture JPEG is decoded by the reference's own stb_image and stored as bytes
(dog_texture_rgb8.npz).  The cat model (the second mesh of the 4096 x 4096 configuration) goes the
same way.  These parsed assets are what the package's scenes load (raytracing_amd/assets/).
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _libs as L  # noqa: E402
from raytracing_amd import host, scenes, scene_types as T  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(ROOT, "raytracing_amd", "assets")
WORK = "/tmp/rt_ref_work"
REF_MODELS = "/root/reference/models"
DOG_DIR = "Australian_Cattle_Dog_v1_L3.123c9c6a5764-399b-4e86-9897-6bcb08b5e8ed"
CAT_DIR = "Cat_v1_L3.123cb1b1943a-2f48-4e44-8f71-6bbe19a3ab64"
# model name -> (zip, member pattern, diffuse map)
MODELS = {
    "dog": ("dog.zip", "*/13463_Australian_Cattle_Dog_v3.obj", os.path.join(DOG_DIR, "Australian_Cattle_Dog_dif.jpg")),
    "cat": (CAT_DIR + ".zip", "*/12221_Cat_v1_l3.obj", os.path.join(CAT_DIR, "Cat_diffuse.jpg")),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def unpack_model(name):
    os.makedirs(os.path.join(WORK, "models"), exist_ok=True)
    dst = os.path.join(WORK, "models", name + ".obj")
    if not os.path.exists(dst):
        archive, member, _ = MODELS[name]
        data = subprocess.run(["unzip", "-p", os.path.join(REF_MODELS, archive), member],
                              stdout=subprocess.PIPE, check=True).stdout
        with open(dst, "wb") as f:
            f.write(data)
    return dst


def unpack_dog():
    return unpack_model("dog")


def assets():
    R = L.ref("strict")
    os.makedirs(ASSETS, exist_ok=True)
    for name, (_, _, jpg_rel) in MODELS.items():
        a = L.RefObjArrays()
        assert R.ref_fast_obj_read(unpack_model(name).encode(), C.byref(a)) == 0
        pos = np.ctypeslib.as_array(a.positions, shape=(a.position_count * 3,)).copy().reshape(-1, 3)
        tc = np.ctypeslib.as_array(a.texcoords, shape=(a.texcoord_count * 2,)).copy().reshape(-1, 2)
        nr = np.ctypeslib.as_array(a.normals, shape=(a.normal_count * 3,)).copy().reshape(-1, 3)
        fv = np.ctypeslib.as_array(a.face_vertices, shape=(a.face_count,)).copy()
        idx = np.ctypeslib.as_array(a.indices, shape=(a.index_count * 3,)).copy().reshape(-1, 3)
        np.savez_compressed(os.path.join(ASSETS, name + "_obj.npz"), positions=pos, texcoords=tc, normals=nr,
                            face_vertices=fv.astype(np.uint8), indices=idx)
        w, h = C.c_int(), C.c_int()
        p = R.ref_stbi_load_rgb(os.path.join(REF_MODELS, jpg_rel).encode(), C.byref(w), C.byref(h))
        rgb = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
        np.savez_compressed(os.path.join(ASSETS, name + "_texture_rgb8.npz"), rgb=rgb)
        print(f"  {name}: {len(pos)} v, {len(tc)} vt, {len(nr)} vn, {len(fv)} faces (vertices per face: "
              f"{sorted(set(fv.tolist()))}), texture {w.value}x{h.value}")


def host_stage_fixture():
    """Tree, face permutation, texture and quantiser of the reference's own loaders."""
    R = L.ref("strict")
    cwd = os.getcwd()
    os.chdir(WORK)
    mesh = T.Mesh()
    R.ref_load_mesh(b"models/dog.obj", C.byref(mesh))
    os.chdir(cwd)
    census = (C.c_uint32 * 4)()
    R.ref_bvh_census(C.byref(mesh), C.byref(census))
    tree = host.mesh_dump_tree(mesh)
    faces = host.mesh_faces_array(mesh)
    tex = T.Image3f()
    jpg = os.path.join(REF_MODELS, DOG_DIR, "Australian_Cattle_Dog_dif.jpg")
    R.ref_load_texture(jpg.encode(), C.byref(tex))
    texels = host.image_as_array(tex)
    rng = np.random.default_rng(5)
    uv = rng.random((64, 2)).astype(np.float32)
    probes = np.zeros((64, 3), np.float32)
    for i in range(64):
        out = (C.c_float * 3)()
        R.ref_sample_texture(C.byref(tex), C.byref(T.Vec2f(float(uv[i, 0]), float(uv[i, 1]))), C.byref(out))
        probes[i] = list(out)
    # quantiser: edge values around every 8-bit step plus specials
    vals = np.concatenate([
        np.linspace(0, 1.2, 3000, dtype=np.float32),
        (np.arange(256, dtype=np.float32) / np.float32(255.0)) ** np.float32(2.2),
        np.array([np.nan, np.inf, 0.0, 1.0, 0.99999994, 1.0000001, 5.0, 1e-30], np.float32)])
    pad = (-len(vals)) % 3
    vals = np.concatenate([vals, np.zeros(pad, np.float32)])
    img, arr = host.new_image(len(vals) // 3, 1)
    arr.reshape(-1)[:] = vals
    q = np.zeros(len(vals), np.uint8)
    assert R.ref_quantize_image(C.byref(img), os.path.join(WORK, "q.png").encode(), q.ctypes.data) == 0
    # the cat through the same loaders
    os.chdir(WORK)
    unpack_model("cat")
    cat = T.Mesh()
    R.ref_load_mesh(b"models/cat.obj", C.byref(cat))
    os.chdir(cwd)
    cat_census = (C.c_uint32 * 4)()
    R.ref_bvh_census(C.byref(cat), C.byref(cat_census))
    cat_tex = T.Image3f()
    R.ref_load_texture(os.path.join(REF_MODELS, MODELS["cat"][2]).encode(), C.byref(cat_tex))
    np.savez_compressed(
        os.path.join(GOLD, "host_stages.npz"),
        cat_census=np.array(list(cat_census), np.uint32), cat_tree_depth=np.uint32(cat.tree_depth),
        cat_tree_sha=sha(host.mesh_dump_tree(cat)), cat_faces_sha=sha(host.mesh_faces_array(cat)),
        cat_texture_sha=sha(host.image_as_array(cat_tex)),
        census=np.array(list(census), np.uint32), tree_depth=np.uint32(mesh.tree_depth),
        tree_sha=sha(tree), faces_sha=sha(faces), root_bbox=tree[0, :6].view(np.float32),
        tree_head=tree[:64], texture_sha=sha(texels), texture_uv=uv, texture_probes=probes,
        quant_in=vals, quant_out=q)


def rng_fixture():
    R = L.ref("strict")
    # the reference's first Random_Serie under an unseeded libc rand() (math.c:148-156), and a seeded one
    series = {"libc_default": bytes.fromhex(
        "67c6697351ff4aec29cdbaabf2fbe3467cc254f81be8e78d765a2e63339fc99a")}
    series["seed7"] = bytes(L.serie_from_seed(7).bytes)
    out = {}
    for name, raw in series.items():
        s = T.RandomSerie()
        for i in range(32):
            s.bytes[i] = raw[i]
        u = np.zeros((32, 4), np.float32)
        for k in range(32):
            o = (C.c_float * 4)()
            R.ref_random4f_uniform(C.byref(s), 0.0, 1.0, C.byref(o))
            u[k] = list(o)
        out[name + "_state"] = np.frombuffer(raw, np.uint8)
        out[name + "_uniform01"] = u
        o = (C.c_float * 4)()
        R.ref_random4f_uniform(C.byref(s), -0.5, 2.0, C.byref(o))
        out[name + "_uniform_range"] = np.array(list(o), np.float32)
        for variant in ("strict", "strict_pm"):
            s2 = T.RandomSerie()
            for i in range(32):
                s2.bytes[i] = raw[i]
            gs = np.zeros((32, 2), np.float32)
            for k in range(32):
                o2 = (C.c_float * 2)()
                L.ref(variant).ref_random2f_gaussian(C.byref(s2), 0.0, 0.25, C.byref(o2))
                gs[k] = list(o2)
            out[f"{name}_gaussian_{variant}"] = gs
    np.savez_compressed(os.path.join(GOLD, "rng_kat.npz"), **out)


RESULT_FIELDS = ["hit", "object_type", "object_id", "mesh_id", "point", "normal", "t_min", "alpha", "beta",
                 "gamma", "uv", "enter_shape", "material_id"]


def results_to_arrays(res):
    n = len(res)
    d = dict(hit=np.zeros(n, np.uint8), object_type=np.zeros(n, np.int32), object_id=np.zeros(n, np.uint32),
             mesh_id=np.zeros(n, np.uint32), point=np.zeros((n, 3), np.float32), normal=np.zeros((n, 3), np.float32),
             t_min=np.zeros(n, np.float32), alpha=np.zeros(n, np.float32), beta=np.zeros(n, np.float32),
             gamma=np.zeros(n, np.float32), uv=np.zeros((n, 2), np.float32), enter_shape=np.zeros(n, np.uint8),
             material_id=np.zeros(n, np.int32))
    for i, r in enumerate(res):
        d["hit"][i] = r.hit
        d["object_type"][i] = r.object_type if r.hit else 0
        d["object_id"][i] = r.object_id
        d["mesh_id"][i] = r.mesh_id
        d["point"][i] = r.intersection_point.tuple()
        d["normal"][i] = r.surface_normal.tuple()
        d["t_min"][i] = r.t_min
        d["alpha"][i], d["beta"][i], d["gamma"][i] = r.triangle_alpha, r.triangle_beta, r.triangle_gamma
        d["uv"][i] = (r.triangle_uv.x, r.triangle_uv.y)
        d["enter_shape"][i] = r.enter_shape
        d["material_id"][i] = r.material_id
    return d


def probe_rays(scene_name, n, seed):
    """Rays that exercise the scene: from the camera through random pixels, from random points
    toward the meshes and lights, and a few axis-parallel / degenerate directions."""
    rng = np.random.default_rng(seed)
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    third = n // 3
    o[:third] = (0.0, 10.0, 85.0)
    tgt = rng.uniform(-40, 40, (third, 3)).astype(np.float32)
    tgt[:, 1] = rng.uniform(-10, 40, third)
    tgt[:, 2] = rng.uniform(-20, 20, third)
    d[:third] = tgt - o[:third]
    o[third:] = rng.uniform(-55, 55, (n - third, 3)).astype(np.float32)
    o[third:, 1] = rng.uniform(-9, 80, n - third)
    d[third:] = rng.standard_normal((n - third, 3)).astype(np.float32)
    norm = np.sqrt((d.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    d = (d / norm[:, None]).astype(np.float32)
    # specials: axis-parallel (zero components -> divisions by zero in the slab tests), unnormalised
    d[-1] = (0, 0, -1); d[-2] = (0, -1, 0); d[-3] = (1, 0, 0); d[-4] = (0, 0, -3.0)
    o[-1] = (0.0, 5.0, 85.0); o[-2] = (0.0, 50.0, 0.0); o[-3] = (-50.0, 0.0, 0.0); o[-4] = (25.0, 5.0, 80.0)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def hit_fixture():
    R = L.ref("strict")
    out = {}
    for name, n in (("sphere_room", 1024), ("dog_room", 2048), ("three_dogs", 2048), ("deux_spheres", 512),
                    ("dogs_and_cat", 2048)):
        sh = scenes.CONFIGS[name]()
        rays = probe_rays(name, n, 17)
        for tag, maxd in (("inf", 3.4028234663852886e38), ("d60", 60.0)):
            res = (T.RayResult * n)()
            for i in range(n):
                ray = T.Ray(T.Vec3f(*rays[i, :3].tolist()), T.Vec3f(*rays[i, 3:].tolist()))
                R.ref_find_ray_hit(sh.ptr, C.byref(ray), maxd, C.byref(res[i]))
            for k, v in results_to_arrays(res).items():
                out[f"{name}_{tag}_{k}"] = v
        out[name + "_rays"] = rays
    np.savez_compressed(os.path.join(GOLD, "hit_kat.npz"), **out)


def tile_jobs(sh, img, size, tile, spp, seed0):
    jobs = []
    k = 0
    for ty in range(0, size, tile):
        for tx in range(0, size, tile):
            jobs.append(L.make_job(sh, img, (tx, ty, min(tile, size - tx), min(tile, size - ty)), spp,
                                   L.serie_from_seed(seed0 + k)))
            k += 1
    return jobs


def region_fixture():
    """raytrace_region float buffers: 40x40 image cut into 8x8-pixel jobs, 2 samples per pixel,
    job k seeded with serie_from_seed(500 + k), for each scene and each reference variant."""
    out = {}
    size, tile, spp = 40, 8, 2
    for name in ("deux_spheres", "sphere_room", "diffuse_room", "dog_room", "three_dogs", "dogs_and_cat"):
        sh = scenes.CONFIGS[name]()
        for variant in ("build", "strict", "strict_pm"):
            img, arr = host.new_image(size, size)
            for job in tile_jobs(sh, img, size, tile, spp, 500):
                L.ref(variant).ref_raytrace_region(C.byref(job))
            out[f"{name}_{variant}"] = arr.copy()
    out["size"], out["tile"], out["spp"], out["seed0"] = size, tile, spp, 500
    np.savez_compressed(os.path.join(GOLD, "region_kat.npz"), **out)


def converged_fixture():
    """Per-16x16-block mean radiance of the reference at its build.sh flags: 128x128, 256 spp."""
    out = {}
    size, spp, tile = 128, 256, 16
    threads = min(8, os.cpu_count() or 1)
    for name in ("sphere_room", "dog_room", "three_dogs", "deux_spheres", "diffuse_room", "dogs_and_cat"):
        sh = scenes.CONFIGS[name]()
        img, arr = host.new_image(size, size)
        ntiles = (size // tile) ** 2
        seeds = (T.RandomSerie * ntiles)(*[L.serie_from_seed(9000 + i) for i in range(ntiles)])
        L.ref("build").ref_render_tiles(sh.ptr, C.byref(img), spp, tile, threads, seeds)
        blocks = arr.reshape(size // tile, tile, size // tile, tile, 3).mean(axis=(1, 3))
        out[name + "_block_means"] = blocks.astype(np.float32)
        out[name + "_nan"] = np.int64(np.isnan(arr).sum())
    out["size"], out["spp"], out["tile"] = size, spp, tile
    np.savez_compressed(os.path.join(GOLD, "converged_stats.npz"), **out)


def converged_pair_fixture():
    """Two renders of every scene by the reference at its build.sh flags, same size and sample
    count, independent generator seeds (one Random_Serie per 16x16 tile, as thread_entry_point
    seeds its tiles): the reference's own run-to-run spread is the yardstick for "the same
    picture at equal sample count"."""
    out = {}
    size, spp, tile = 96, 256, 16
    threads = min(8, os.cpu_count() or 1)
    ntiles = (size // tile) ** 2
    for name in ("deux_spheres", "sphere_room", "diffuse_room", "dog_room", "three_dogs", "dogs_and_cat"):
        sh = scenes.CONFIGS[name]()
        for tag, seed0 in (("a", 70000), ("b", 80000)):
            img, arr = host.new_image(size, size)
            seeds = (T.RandomSerie * ntiles)(*[L.serie_from_seed(seed0 + i) for i in range(ntiles)])
            t = L.ref("build").ref_render_tiles(sh.ptr, C.byref(img), spp, tile, threads, seeds)
            out[f"{name}_{tag}"] = arr.copy()
            print(f"  {name} {tag}: {t:.1f} s, mean {np.nanmean(arr):.5f}, NaN {int(np.isnan(arr).sum())}", flush=True)
    out["size"], out["spp"], out["tile"] = size, spp, tile
    np.savez_compressed(os.path.join(GOLD, "converged_pairs.npz"), **out)


def math_fixture():
    """Digest of the portable sin/cos/log over every argument the render loop can produce."""
    k = np.arange(1 << 23, dtype=np.uint32)
    u = k.astype(np.float32) * np.float32(2.0 ** -23)
    a = (2.0 * np.pi * u.astype(np.float64)).astype(np.float32)
    s = np.zeros_like(a)
    c = np.zeros_like(a)
    lg = np.zeros_like(a)
    L.oracle().orc_sincosf(a.ctypes.data, s.ctypes.data, c.ctypes.data, a.size)
    L.oracle().orc_logf(u.ctypes.data, lg.ctypes.data, u.size)
    np.savez_compressed(os.path.join(GOLD, "math_kat.npz"), sin_sha=sha(s), cos_sha=sha(c), log_sha=sha(lg),
                        sin_head=s[:4096:64], cos_head=c[:4096:64], log_head=lg[:4096:64])


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or ["assets", "host", "rng", "hit", "region", "converged", "pairs", "math"]
    steps = dict(assets=assets, host=host_stage_fixture, rng=rng_fixture, hit=hit_fixture,
                 region=region_fixture, converged=converged_fixture, pairs=converged_pair_fixture, math=math_fixture)
    for name in todo:
        print("golden:", name, flush=True)
        steps[name]()
    print("done")
