"""Scenes written as JSON documents give the very records the hand-built scenes have (no GPU needed)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _libs as L
from raytracing_amd import gpu, host, scene_file, scenes

DOG_FILES = dict(textures={"dog_diffuse": "dog_texture_rgb8.npz"}, meshes={"dog": "dog_obj.npz"})
BOTH_FILES = dict(textures={"dog_diffuse": "dog_texture_rgb8.npz", "cat_diffuse": "cat_texture_rgb8.npz"},
                  meshes={"dog": "dog_obj.npz", "cat": "cat_obj.npz"})


def _bytes(array, count):
    return bytes(C.string_at(C.addressof(array), C.sizeof(array._type_) * count))


def _render(sh, W=24, H=16, spp=2, seed=5):
    img, arr = host.new_image(W, H)
    r = gpu.rect(0, 0, W, H)
    L.oracle().orc_render_region_counter(sh.ptr, C.byref(img), C.byref(r), spp, 0, seed, L.MATH_PORTABLE, 64, None)
    return arr


@pytest.mark.parametrize("name", sorted(scenes.CONFIGS))
def test_document_round_trip_gives_the_same_records_and_the_same_picture(name, tmp_path):
    sh = scenes.CONFIGS[name]()
    files = DOG_FILES if sh.scene.mesh_count else {}
    if name == "dogs_and_cat":
        files = dict(BOTH_FILES, objects={"dog": scenes.load_dog_mesh(), "cat": scenes.load_cat_mesh(),
                                          "dog_diffuse": scenes.load_dog_texture(), "cat_diffuse": scenes.load_cat_texture()})
    text = json.dumps(scene_file.describe(sh, **files))
    path = tmp_path / "scene.json"
    path.write_text(text)
    back = scene_file.load(str(path), base_dir=scenes.ASSET_DIR)
    a, b = sh.scene, back.scene
    assert bytes(a.camera) == bytes(b.camera)
    for field, count in (("spheres", a.sphere_count), ("planes", a.plane_count), ("lights", a.light_count)):
        assert getattr(b, field.rstrip("s") + "_count") == count
        assert _bytes(getattr(sh, field), count) == _bytes(getattr(back, field), count)
    assert a.material_count == b.material_count and a.mesh_count == b.mesh_count
    for i in range(a.material_count):
        ma, mb = a.materials[i], b.materials[i]
        assert (bytes(ma.diffuse_color), ma.mirror, bytes(ma.mirror_color), ma.transparent, ma.n, ma.textured) == \
               (bytes(mb.diffuse_color), mb.mirror, bytes(mb.mirror_color), mb.transparent, mb.n, mb.textured)
    for i in range(a.mesh_count):
        assert bytes(a.meshes[i].transform) == bytes(b.meshes[i].transform)
        assert a.meshes[i].material_id == b.meshes[i].material_id
        assert a.meshes[i].mesh.contents.face_count == b.meshes[i].mesh.contents.face_count
    if name == "dogs_and_cat":      # two meshes, two textures: each instance and material got its own back
        assert [b.meshes[i].mesh.contents.face_count for i in range(4)] == [71968, 71968, 70576, 70576]
        assert C.addressof(b.materials[8].texture.contents) != C.addressof(b.materials[9].texture.contents)
    assert L.same_bits(_render(sh), _render(back)).all()


def test_fov_in_degrees_is_the_reference_expression():
    doc = {"camera": {"position": [0, 10, 85], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov_degrees": 60}}
    sh = scene_file.load(doc)
    assert bytes(sh.scene.camera) == bytes(scenes.room_camera())


@pytest.mark.parametrize("doc", [
    {},
    {"camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0]}},
    {"camera": {"position": [0, 0], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 1.0}},
    {"camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 1.0},
     "materials": [{"diffuse": [1, 1, 1]}], "spheres": [{"center": [0, 0, 0], "radius": 1, "material": 1}]},
    {"camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 1.0},
     "materials": [{"diffuse": [1, 1, 1], "mirror": [1, 1, 1]}]},
    {"camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 1.0},
     "materials": [{"diffuse": [1, 1, 1]}], "instances": [{"mesh": "nope", "position": [0, 0, 0], "material": 0}]},
])
def test_malformed_documents_are_refused(doc):
    with pytest.raises(scene_file.SceneFileError):
        scene_file.load(doc)


MTL = """# exporter header
newmtl Dog
\tNs 10.0000
\tNi 1.5000
\td 1.0000
\tTr 0.0000
\tKd 0.8000 0.7000 0.6000
\tmap_Kd textures\\dog_texture_rgb8.jpg
\tmap_bump dog_bump.jpg
newmtl Crystal
\tNi 1.3300
\tTr 0.7500
"""


def test_materials_from_a_model_material_library(tmp_path):
    """The .mtl data the reference's OBJ parser reads and the reference never uses (src/fast_obj.h:49-77):
    Kd becomes the diffuse colour, map_Kd the diffuse map (by file stem among the document's textures),
    a dissolve below 1 glass with refraction index Ni."""
    (tmp_path / "model.mtl").write_text(MTL)
    lib = scene_file.parse_mtl(MTL)
    assert lib["Dog"]["Kd"] == (0.8, 0.7, 0.6) and lib["Dog"]["Ni"] == 1.5 and lib["Dog"]["d"] == 1.0
    assert lib["Crystal"]["d"] == 0.25 and lib["Dog"]["map_bump"] == "dog_bump.jpg"
    import shutil
    shutil.copy(os.path.join(scenes.ASSET_DIR, "dog_texture_rgb8.npz"), tmp_path / "dog_texture_rgb8.npz")
    doc = {"camera": {"position": [0, 10, 85], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov_degrees": 60},
           "textures": {"dog_texture_rgb8": "dog_texture_rgb8.npz"},
           "materials": [{"mtl": "model.mtl", "name": "Dog"}, {"mtl": "model.mtl", "name": "Crystal"}],
           "spheres": [{"center": [0, 0, 0], "radius": 5, "material": 0}, {"center": [9, 0, 0], "radius": 3, "material": 1}]}
    sh = scene_file.load(doc, base_dir=str(tmp_path))
    m0, m1 = sh.scene.materials[0], sh.scene.materials[1]
    assert m0.textured and not m0.transparent and np.allclose(m0.diffuse_color.tuple(), (0.8, 0.7, 0.6))
    assert m0.texture.contents.size.x == 1024
    assert m1.transparent and abs(m1.n - 1.33) < 1e-6 and not m1.textured
    with pytest.raises(scene_file.SceneFileError, match="not in model.mtl"):
        scene_file.load(dict(doc, materials=[{"mtl": "model.mtl", "name": "Cat"}]), base_dir=str(tmp_path))
    with pytest.raises(scene_file.SceneFileError, match="needs a texture"):
        scene_file.load(dict(doc, textures={}), base_dir=str(tmp_path))
