"""The benchmark / parity scenes, assembled through the scene API.

The host program hard-codes its scene in main() (reference: src/main.c:885-1054);
the numbers below are that scene's data (materials :885-929, the optional
spheres :933-947, planes :951-977, lights :979-992, mesh instances :996-1030,
camera :1033-1038), regrouped into the configurations the benchmark
names (plus one with the second model, the cat).  Everything is plain data in ctypes records: no rendering logic here.
"""
import ctypes as C
import math
import os

import numpy as np

from . import scene_types as T
from . import host

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

WHITE, BLACK = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
RED, GREEN, BLUE = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
PINK, YELLOW = (1.0, 0.0, 1.0), (1.0, 1.0, 0.0)


def fov_radians(degrees):
    """The reference's `60.0f / 180.0f * M_PI` (src/main.c:1037): float quotient, double product, float store."""
    return float(np.float32(float(np.float32(degrees) / np.float32(180.0)) * math.pi))


class SceneHandle:
    """Owns the ctypes arrays a Scene points into (and keeps them alive)."""

    def __init__(self, name, camera, materials, spheres, planes, lights, meshes, keep=()):
        self.name = name
        self._keep = list(keep)
        self.materials = (T.Material * max(1, len(materials)))(*materials)
        self.spheres = (T.Sphere * max(1, len(spheres)))(*spheres)
        self.planes = (T.Plane * max(1, len(planes)))(*planes)
        self.lights = (T.Light * max(1, len(lights)))(*lights)
        self.meshes = (T.SceneMesh * max(1, len(meshes)))(*meshes)
        s = T.Scene()
        s.camera = camera
        s.material_count, s.materials = len(materials), self.materials
        s.sphere_count, s.spheres = len(spheres), self.spheres
        s.plane_count, s.planes = len(planes), self.planes
        s.light_count, s.lights = len(lights), self.lights
        s.mesh_count, s.meshes = len(meshes), self.meshes
        self.scene = s

    @property
    def ptr(self):
        return C.pointer(self.scene)


def _material(diffuse=BLACK, mirror=False, mirror_color=BLACK, transparent=False, n=0.0, texture=None):
    m = T.Material()
    m.diffuse_color = T.Vec3f(*diffuse)
    m.mirror = mirror
    m.mirror_color = T.Vec3f(*mirror_color)
    m.transparent = transparent
    m.n = n
    m.textured = texture is not None
    if texture is not None:
        m.texture = C.pointer(texture)
    return m


def room_materials(texture=None):
    """Material table of src/main.c:885-929 (index 8 is the textured one)."""
    return [
        _material(diffuse=(0.5, 0.5, 0.5)),          # 0 centre sphere
        _material(diffuse=GREEN),                     # 1 back wall
        _material(diffuse=BLUE),                      # 2 floor
        _material(diffuse=PINK),                      # 3 left wall
        _material(diffuse=YELLOW),                    # 4 right wall
        _material(mirror=True, mirror_color=WHITE),   # 5 mirror
        _material(diffuse=RED),                       # 6 ceiling
        _material(transparent=True, n=1.5),           # 7 glass
        _material(diffuse=WHITE, texture=texture),    # 8 textured (plain white without a texture)
    ]


def room_planes():
    """src/main.c:951-977"""
    rows = [((0, 0, 1), -60.0, 1), ((0, 1, 0), -10.0, 2), ((1, 0, 0), -60.0, 3),
            ((-1, 0, 0), -60.0, 4), ((0, -1, 0), -90.0, 6)]
    return [T.Plane(T.Vec3f(*n), d, m) for n, d, m in rows]


def room_lights():
    """src/main.c:979-992"""
    return [T.Light(T.Vec3f(-10.0, 20.0, 75.0), 3.0, T.Vec3f(*WHITE), 1e5),
            T.Light(T.Vec3f(10.0, 17.0, 75.0), 2.0, T.Vec3f(1.0, 0.2, 0.1), 1e4)]


def room_spheres():
    """The spheres of the disabled block src/main.c:933-947: grey, mirror, glass."""
    return [T.Sphere(T.Vec3f(0.0, 0.0, 0.0), 10.0, 0),
            T.Sphere(T.Vec3f(-8.0, 10.0, 15.0), 5.0, 5),
            T.Sphere(T.Vec3f(-4.0, 5.0, 25.0), 4.0, 7)]


def room_camera():
    """src/main.c:1033-1038"""
    return T.Camera(T.Vec3f(0.0, 10.0, 85.0), T.Vec3f(0.0, 0.0, 0.0), T.Vec3f(0.0, 1.0, 0.0), fov_radians(60.0))


def instance(mesh, position, material_id, mirrored_x=False):
    sm = T.SceneMesh()
    sm.mesh = C.pointer(mesh)
    sm.transform.position = T.Vec3f(*position)
    sm.transform.rotX = T.Vec3f(-1.0 if mirrored_x else 1.0, 0.0, 0.0)
    sm.transform.rotY = T.Vec3f(0.0, 1.0, 0.0)
    sm.transform.rotZ = T.Vec3f(0.0, 0.0, 1.0)
    sm.material_id = material_id
    return sm


# --------------------------------------------------------------------------- assets

_ASSET_CACHE = {}


def _load_mesh(name, tree="reference", leaf_triangles=0):
    """A model's mesh, prepared once per process.  With RT_ASSET_CACHE_DIR set, the prepared mesh (arrays,
    permuted faces, tree) is also kept there as one file keyed by the model file's digest and the builder's
    parameters, and later processes map it back instead of parsing and building."""
    key = name if tree == "reference" else f"{name}/{tree}/{leaf_triangles}"
    if key not in _ASSET_CACHE:
        source = os.path.join(ASSET_DIR, name + "_obj.npz")
        cache_dir = os.environ.get("RT_ASSET_CACHE_DIR")
        mesh = path = None
        if cache_dir:
            import hashlib
            with open(source, "rb") as f:
                digest = hashlib.sha256(f.read() + f"|{tree}|{leaf_triangles}|v1".encode()).digest()
            content_key = int.from_bytes(digest[:8], "little")
            path = os.path.join(cache_dir, f"{name}-{tree}-{leaf_triangles}-{digest[:8].hex()}.rtmesh")
            try:
                mesh = host.mesh_load(path, content_key)
            except RuntimeError:             # a damaged cache file is a miss: rebuild and write it anew
                mesh = None
        if mesh is None:
            with np.load(source, allow_pickle=False) as z:
                mesh = host.mesh_from_obj_arrays(
                    z["positions"], z["texcoords"], z["normals"], z["indices"], z["face_vertices"],
                    tree=tree, leaf_triangles=leaf_triangles)
            if path:
                os.makedirs(cache_dir, exist_ok=True)
                host.mesh_save(mesh, content_key, path)
        _ASSET_CACHE[key] = mesh
    return _ASSET_CACHE[key]


def _load_texture(name):
    key = name + "_texture"
    if key not in _ASSET_CACHE:
        with np.load(os.path.join(ASSET_DIR, name + "_texture_rgb8.npz"), allow_pickle=False) as z:
            _ASSET_CACHE[key] = host.texture_from_rgb8(z["rgb"])
    return _ASSET_CACHE[key]


def load_dog_mesh(tree="reference", leaf_triangles=0):
    """The dog model (71 968 triangles) from its parsed-OBJ arrays, triangulated and
    given its box tree by the host C code (include/rt_host.h): the reference's tree, or
    (tree="sah") the surface-area builder's."""
    return _load_mesh("dog", tree, leaf_triangles)


def load_dog_texture():
    """The dog's diffuse map (decoded 8-bit RGB), linearised by the host C code."""
    return _load_texture("dog")


def load_cat_mesh(tree="reference", leaf_triangles=0):
    """The cat model (70 576 triangles), the second mesh of the 4096 x 4096 configuration."""
    return _load_mesh("cat", tree, leaf_triangles)


def load_cat_texture():
    return _load_texture("cat")


# --------------------------------------------------------------------------- configurations

def deux_spheres():
    """Config 1: two diffuse spheres before a grey wall under one white light (direct light,
    shadows).  The screenshot of that name predates the reference's current source, so the
    parameters are this build's; the renderer exercised is the current one."""
    mats = [_material(diffuse=RED), _material(diffuse=GREEN), _material(diffuse=(0.5, 0.5, 0.5))]
    spheres = [T.Sphere(T.Vec3f(0.0, 0.0, 0.0), 10.0, 0), T.Sphere(T.Vec3f(16.0, -4.0, 6.0), 6.0, 1)]
    planes = [T.Plane(T.Vec3f(0.0, 0.0, 1.0), -40.0, 2), T.Plane(T.Vec3f(0.0, 1.0, 0.0), -10.0, 2)]
    lights = [T.Light(T.Vec3f(-15.0, 30.0, 40.0), 1.0, T.Vec3f(*WHITE), 2e5)]
    cam = T.Camera(T.Vec3f(0.0, 0.0, 55.0), T.Vec3f(0.0, 0.0, 0.0), T.Vec3f(0.0, 1.0, 0.0), fov_radians(60.0))
    return SceneHandle("deux_spheres", cam, mats, spheres, planes, lights, [])


def sphere_room():
    """Configs 2 and 3: the five-plane room, two area lights, grey + mirror + glass spheres."""
    return SceneHandle("sphere_room", room_camera(), room_materials(), room_spheres(),
                       room_planes(), room_lights(), [])


def diffuse_room():
    """Config 3 variant: the room with the three spheres all diffuse (indirect light only)."""
    spheres = room_spheres()
    spheres[1].material_id = 0
    spheres[2].material_id = 0
    return SceneHandle("diffuse_room", room_camera(), room_materials(), spheres,
                       room_planes(), room_lights(), [])


def dog_room(mesh=None, texture=None, position=(0.0, -10.0, 0.0), tree="reference", leaf_triangles=0):
    """Config 4 (the headline benchmark): the room with one textured dog."""
    mesh = mesh if mesh is not None else load_dog_mesh(tree, leaf_triangles)
    texture = texture if texture is not None else load_dog_texture()
    return SceneHandle("dog_room", room_camera(), room_materials(texture), [], room_planes(),
                       room_lights(), [instance(mesh, position, 8)], keep=[mesh, texture])


def three_dogs(mesh=None, texture=None, tree="reference", leaf_triangles=0):
    """Config 5 / the reference's own main() scene: mirror dog (mirrored in x), textured dog, glass dog."""
    mesh = mesh if mesh is not None else load_dog_mesh(tree, leaf_triangles)
    texture = texture if texture is not None else load_dog_texture()
    meshes = [instance(mesh, (-25.0, -10.0, 0.0), 5, mirrored_x=True),
              instance(mesh, (25.0, -10.0, 0.0), 8),
              instance(mesh, (0.0, -10.0, 10.0), 7)]
    return SceneHandle("three_dogs", room_camera(), room_materials(texture), [], room_planes(),
                       room_lights(), meshes, keep=[mesh, texture])


def dogs_and_cat(tree="reference", leaf_triangles=0):
    """Config 5 as the benchmark list words it (dog + cat meshes + mirrors): two distinct meshes, two
    textures, four instances (mirror dog mirrored in x, textured dog, textured cat, glass cat) and
    the mirror sphere of the room.  The reference's source has no such scene (its main() instances the
    dog three times and never loads the cat), so the placement is this build's; both models go
    through the same loader (src/main.c:177-262)."""
    dog, cat = load_dog_mesh(tree, leaf_triangles), load_cat_mesh(tree, leaf_triangles)
    dog_tex, cat_tex = load_dog_texture(), load_cat_texture()
    mats = room_materials(dog_tex) + [_material(diffuse=WHITE, texture=cat_tex)]     # 9: the cat's diffuse map
    meshes = [instance(dog, (-30.0, -10.0, 0.0), 5, mirrored_x=True),
              instance(dog, (25.0, -10.0, 10.0), 8),
              instance(cat, (-5.0, -10.0, -30.0), 9),
              instance(cat, (0.0, -10.0, 35.0), 7)]
    spheres = [T.Sphere(T.Vec3f(-8.0, 10.0, 15.0), 5.0, 5)]
    return SceneHandle("dogs_and_cat", room_camera(), mats, spheres, room_planes(), room_lights(), meshes,
                       keep=[dog, cat, dog_tex, cat_tex])


CONFIGS = {
    "deux_spheres": deux_spheres,
    "sphere_room": sphere_room,
    "diffuse_room": diffuse_room,
    "dog_room": dog_room,
    "three_dogs": three_dogs,
    "dogs_and_cat": dogs_and_cat,
}
