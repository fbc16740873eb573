"""ctypes mirror of include/rt_scene.h.

The records are layout twins of the host program's scene API (reference:
src/math.h:20-190, src/main.h:7-72), so one ctypes Scene can be handed to the
GPU library, to the CPU oracle and -- where it was built -- to the compiled
reference itself.
"""
import ctypes as C


class Vec2u(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32)]


class Vec2f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Vec3f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def tuple(self):
        return (self.x, self.y, self.z)


class Rect2u(C.Structure):
    _fields_ = [("point", Vec2u), ("size", Vec2u)]


class RandomSerie(C.Structure):
    _pack_ = 16
    _fields_ = [("bytes", C.c_uint8 * 32)]


class Transform(C.Structure):
    _fields_ = [("position", Vec3f), ("rotX", Vec3f), ("rotY", Vec3f), ("rotZ", Vec3f)]


class VertexRef(C.Structure):
    _fields_ = [("vertexId", C.c_uint32), ("texcoordId", C.c_uint32), ("normalId", C.c_uint32)]


class Triangle(C.Structure):
    _fields_ = [("ids", VertexRef * 3)]


class BBox(C.Structure):
    _fields_ = [("min", Vec3f), ("max", Vec3f)]


class MeshNode(C.Structure):
    pass


MeshNode._fields_ = [
    ("bbox", BBox),
    ("childs", C.POINTER(MeshNode)),
    ("first_triangle_id", C.c_uint32),
    ("last_triangle_id", C.c_uint32),
]


class Mesh(C.Structure):
    _fields_ = [
        ("point_count", C.c_uint32), ("points", C.POINTER(Vec3f)),
        ("texcoord_count", C.c_uint32), ("texcoords", C.POINTER(Vec2f)),
        ("normal_count", C.c_uint32), ("normals", C.POINTER(Vec3f)),
        ("face_count", C.c_uint32), ("faces", C.POINTER(Triangle)),
        ("tree_depth", C.c_uint32),
        ("root", MeshNode),
    ]


class SceneMesh(C.Structure):
    _fields_ = [("mesh", C.POINTER(Mesh)), ("transform", Transform), ("material_id", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3f), ("lookAt", Vec3f), ("up", Vec3f), ("field_of_view", C.c_float)]


class Image3f(C.Structure):
    _fields_ = [("size", Vec2u), ("pixels", C.POINTER(Vec3f))]


class Material(C.Structure):
    _fields_ = [
        ("diffuse_color", Vec3f),
        ("mirror", C.c_bool),
        ("mirror_color", Vec3f),
        ("transparent", C.c_bool),
        ("n", C.c_float),
        ("textured", C.c_bool),
        ("texture", C.POINTER(Image3f)),
    ]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3f), ("radius", C.c_float), ("material_id", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("center", Vec3f), ("radius", C.c_float), ("color", Vec3f), ("albedo", C.c_float)]


class Plane(C.Structure):
    _fields_ = [("normal", Vec3f), ("d", C.c_float), ("material_id", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("material_count", C.c_int32), ("materials", C.POINTER(Material)),
        ("sphere_count", C.c_int32), ("spheres", C.POINTER(Sphere)),
        ("plane_count", C.c_int32), ("planes", C.POINTER(Plane)),
        ("light_count", C.c_int32), ("lights", C.POINTER(Light)),
        ("mesh_count", C.c_int32), ("meshes", C.POINTER(SceneMesh)),
    ]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3f), ("direction", Vec3f)]


SPHERE, PLANE, LIGHT, TRIANGLE = 0, 1, 2, 3


class RayResult(C.Structure):
    _fields_ = [
        ("hit", C.c_bool),
        ("object_type", C.c_int32),
        ("object_id", C.c_uint32),
        ("mesh_id", C.c_uint32),
        ("intersection_point", Vec3f),
        ("surface_normal", Vec3f),
        ("t_min", C.c_float),
        ("triangle_alpha", C.c_float),
        ("triangle_beta", C.c_float),
        ("triangle_gamma", C.c_float),
        ("triangle_uv", Vec2f),
        ("enter_shape", C.c_bool),
        ("material_id", C.c_int32),
    ]


class Job(C.Structure):
    _fields_ = [
        ("scene", C.POINTER(Scene)),
        ("image", C.POINTER(Image3f)),
        ("region", Rect2u),
        ("ray_per_pixel", C.c_uint32),
        ("_pad", C.c_uint8 * 12),
        ("serie", RandomSerie),
    ]


# record sizes on x86-64 (Scene, Material, Sphere, ... in the order the layout tests walk them)
LAYOUT_SIZES = [
    (Scene, 120), (Material, 48), (Sphere, 20), (Plane, 20), (Light, 32), (SceneMesh, 64),
    (Mesh, 112), (MeshNode, 40), (Triangle, 36), (RayResult, 72), (Job, 80), (RandomSerie, 32),
    (Image3f, 16), (Camera, 40), (Transform, 48), (Ray, 24),
]
for _t, _n in LAYOUT_SIZES:
    assert C.sizeof(_t) == _n, (_t.__name__, C.sizeof(_t), _n)
assert Job.serie.offset == 48 and Mesh.root.offset == 72 and Scene.meshes.offset == 112
