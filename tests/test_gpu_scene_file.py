"""Scenes that arrive as documents (raytracing_amd/scene_file.py: the data the reference hard-codes
in main(), src/main.c:885-1054, and the .mtl data its OBJ parser reads and never uses,
src/fast_obj.h:49-77) rendered by the HIP path through the C ABI: the loaded records give the
oracle's pixels bit for bit, so the front end feeds the device the same scene it feeds the CPU."""
import ctypes as C
import json
import os
import shutil

import numpy as np
import pytest

import _libs as L
from raytracing_amd import gpu, host, scene_file, scenes

pytestmark = pytest.mark.gpu


def oracle_frame(sh, W, H, spp, seed, stats=None):
    img, arr = host.new_image(W, H)
    r = gpu.rect(0, 0, W, H)
    L.oracle().orc_render_region_counter(sh.ptr, C.byref(img), C.byref(r), spp, 0, seed, L.MATH_PORTABLE, 64,
                                         C.byref(stats) if stats else None)
    return arr


def device_frame(sh, W, H, spp, seed):
    g = gpu.GpuScene(sh)
    img, got = host.new_image(W, H)
    g.reset_stats()
    g.render_region(img, (0, 0, W, H), spp, seed)
    rays = g.stats()["rays"]
    g.close()
    return got, rays


@pytest.mark.parametrize("name", ["sphere_room", "dogs_and_cat"])
def test_scene_written_as_a_document_renders_the_oracles_bits(name, tmp_path):
    """A hand-built scene -> JSON text on disk -> scene_file.load -> rt_gpu_scene_create + rt_gpu_render_region:
    the frame equals the oracle's frame of the LOADED records bit for bit (and the frame of the original
    scene: the document loses nothing), with the same number of rays."""
    assert gpu.lib().rt_gpu_device_count() >= 1, "no GPU visible: this test has no CPU fallback"
    sh = scenes.CONFIGS[name]()
    files = {}
    if name == "dogs_and_cat":
        files = dict(textures={"dog_diffuse": "dog_texture_rgb8.npz", "cat_diffuse": "cat_texture_rgb8.npz"},
                     meshes={"dog": "dog_obj.npz", "cat": "cat_obj.npz"},
                     objects={"dog": scenes.load_dog_mesh(), "cat": scenes.load_cat_mesh(),
                              "dog_diffuse": scenes.load_dog_texture(), "cat_diffuse": scenes.load_cat_texture()})
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene_file.describe(sh, **files)))
    loaded = scene_file.load(str(path), base_dir=scenes.ASSET_DIR)
    W, H, spp, seed = 88, 64, 3, 606
    st = L.Stats()
    want = oracle_frame(loaded, W, H, spp, seed, st)
    got, rays = device_frame(loaded, W, H, spp, seed)
    same = L.same_bits(got, want)
    assert same.all(), int((~same).sum())
    assert rays == st.rays
    assert L.same_bits(got, oracle_frame(sh, W, H, spp, seed)).all()
    assert np.isfinite(got).all() and got.mean() > 0


MTL = """newmtl Coat
Ni 1.0000
d 1.0000
Kd 0.9000 0.8500 0.8000
map_Kd textures\\dog_texture_rgb8.jpg
newmtl Crystal
Ni 1.4500
Tr 0.6000
newmtl Chalk
Kd 0.7000 0.7000 0.6500
"""


def test_materials_from_a_material_library_render_the_oracles_bits(tmp_path):
    """A document whose materials come from a Wavefront .mtl library (Kd, Ni, dissolve, map_Kd): a dog with the
    library's textured material, a dog of the library's glass, a sphere of its plain diffuse material, in the
    room's planes and lights.  The HIP path renders the oracle's bits for the loaded records; the materials are
    what the library says (texture bound, glass with the library's index)."""
    assert gpu.lib().rt_gpu_device_count() >= 1, "no GPU visible: this test has no CPU fallback"
    (tmp_path / "dog.mtl").write_text(MTL)
    for f in ("dog_texture_rgb8.npz", "dog_obj.npz"):
        shutil.copy(os.path.join(scenes.ASSET_DIR, f), tmp_path / f)
    room = scenes.sphere_room()
    doc = scene_file.describe(room)
    first = len(doc["materials"])
    doc["materials"] += [{"mtl": "dog.mtl", "name": "Coat"}, {"mtl": "dog.mtl", "name": "Crystal"}, {"mtl": "dog.mtl", "name": "Chalk"}]
    doc["textures"] = {"dog_texture_rgb8": "dog_texture_rgb8.npz"}
    doc["meshes"] = {"dog": "dog_obj.npz"}
    doc["instances"] = [{"mesh": "dog", "position": [12.0, -10.0, 0.0], "material": first},
                        {"mesh": "dog", "position": [-14.0, -10.0, 10.0], "material": first + 1, "mirrored_x": True}]
    doc["spheres"] = [{"center": [0.0, -4.0, 30.0], "radius": 6.0, "material": first + 2}]
    (tmp_path / "scene.json").write_text(json.dumps(doc))
    sh = scene_file.load(str(tmp_path / "scene.json"))
    coat, crystal, chalk = (sh.scene.materials[first + k] for k in range(3))
    assert coat.textured and coat.texture.contents.size.x == 1024 and not coat.transparent
    assert crystal.transparent and abs(crystal.n - 1.45) < 1e-6
    assert not chalk.textured and abs(chalk.diffuse_color.x - 0.7) < 1e-6
    W, H, spp, seed = 96, 72, 4, 17
    st = L.Stats()
    want = oracle_frame(sh, W, H, spp, seed, st)
    got, rays = device_frame(sh, W, H, spp, seed)
    same = L.same_bits(got, want)
    assert same.all(), int((~same).sum())
    assert rays == st.rays and st.tex_lookups > 0
    assert got.mean() > 0
