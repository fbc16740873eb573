"""Python face of the host-side C helpers (include/rt_host.h, librt_host.so).

Plumbing only: every function forwards to the C implementation in
raytracing_amd/csrc/rt_host_assets.c.
"""
import ctypes as C
import os

import numpy as np

from . import scene_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.rt_host_mesh_build_tree.argtypes = [C.POINTER(T.Mesh)]
        L.rt_host_mesh_build_tree.restype = C.c_int
        L.rt_host_mesh_build_tree_sah.argtypes = [C.POINTER(T.Mesh), C.c_uint32]
        L.rt_host_mesh_build_tree_sah.restype = C.c_int
        L.rt_host_mesh_from_obj_arrays.argtypes = [
            C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
            C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(T.Mesh)]
        L.rt_host_mesh_from_obj_arrays.restype = C.c_int
        L.rt_host_mesh_save.argtypes = [C.POINTER(T.Mesh), C.c_uint64, C.c_char_p]
        L.rt_host_mesh_save.restype = C.c_int
        L.rt_host_mesh_load.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(T.Mesh)]
        L.rt_host_mesh_load.restype = C.c_int
        L.rt_host_mesh_free.argtypes = [C.POINTER(T.Mesh)]
        L.rt_host_mesh_free.restype = None
        L.rt_host_mesh_census.argtypes = [C.POINTER(T.Mesh), C.POINTER(C.c_uint32 * 4)]
        L.rt_host_mesh_census.restype = None
        L.rt_host_mesh_dump_tree.argtypes = [C.POINTER(T.Mesh), C.c_void_p]
        L.rt_host_mesh_dump_tree.restype = C.c_size_t
        L.rt_host_texture_from_rgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(T.Image3f)]
        L.rt_host_texture_from_rgb8.restype = C.c_int
        L.rt_host_image_free.argtypes = [C.POINTER(T.Image3f)]
        L.rt_host_image_free.restype = None
        L.rt_host_image_quantize.argtypes = [C.POINTER(T.Image3f), C.c_void_p]
        L.rt_host_image_quantize.restype = None
        _LIB = L
    return _LIB


def mesh_from_obj_arrays(positions, texcoords, normals, indices_ptn, face_vertices, tree="reference", leaf_triangles=0):
    """OBJ parser arrays (entry 0 of each attribute array = the parser's dummy) -> T.Mesh.
    tree: "reference" = the reference's splitting rule (its tree and face order, node for node);
    "sah" = the surface-area builder (rt_host_mesh_build_tree_sah; leaves of at most leaf_triangles, 0 = 4)."""
    if tree not in ("reference", "sah"):
        raise ValueError(f"unknown tree builder {tree!r}")
    positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    texcoords = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
    normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
    indices_ptn = np.ascontiguousarray(indices_ptn, dtype=np.uint32).reshape(-1, 3)
    face_vertices = np.ascontiguousarray(face_vertices, dtype=np.uint32)
    if int(face_vertices.sum(dtype=np.uint64)) > len(indices_ptn):
        raise ValueError(f"the faces name {int(face_vertices.sum(dtype=np.uint64))} corners, the index table holds {len(indices_ptn)}")
    mesh = T.Mesh()
    rc = lib().rt_host_mesh_from_obj_arrays(
        positions.ctypes.data, len(positions), texcoords.ctypes.data, len(texcoords),
        normals.ctypes.data, len(normals), indices_ptn.ctypes.data, len(indices_ptn), face_vertices.ctypes.data,
        len(face_vertices), C.byref(mesh))
    if rc != 0:
        raise RuntimeError(f"rt_host_mesh_from_obj_arrays failed: {rc}")
    if tree == "sah":
        rc = lib().rt_host_mesh_build_tree_sah(C.byref(mesh), leaf_triangles)
        if rc != 0:
            raise RuntimeError(f"rt_host_mesh_build_tree_sah failed: {rc}")
    return mesh


def mesh_save(mesh, key, path):
    """Write a prepared mesh (arrays, permuted faces, tree) to one file under a 64-bit content key."""
    rc = lib().rt_host_mesh_save(C.byref(mesh), key, os.fsencode(path))
    if rc != 0:
        raise RuntimeError(f"rt_host_mesh_save failed: {rc}")


def mesh_load(path, key):
    """A prepared mesh back from its file, or None when the file is missing or was written under another key;
    a malformed file is an error."""
    mesh = T.Mesh()
    rc = lib().rt_host_mesh_load(os.fsencode(path), key, C.byref(mesh))
    if rc in (-3, -7):
        return None
    if rc != 0:
        raise RuntimeError(f"rt_host_mesh_load({path}) failed: {rc}")
    return mesh


def mesh_census(mesh):
    out = (C.c_uint32 * 4)()
    lib().rt_host_mesh_census(C.byref(mesh), C.byref(out))
    return dict(nodes=out[0], leaves=out[1], depth=out[2], max_leaf=out[3])


def mesh_dump_tree(mesh):
    n = mesh_census(mesh)["nodes"]
    words = np.zeros((n, 8), dtype=np.uint32)
    got = lib().rt_host_mesh_dump_tree(C.byref(mesh), words.ctypes.data)
    assert got == n
    return words


def mesh_faces_array(mesh):
    """The (permuted) face index table as a (face_count, 9) uint32 array (a copy)."""
    n = mesh.face_count
    buf = (C.c_uint32 * (9 * n)).from_address(C.addressof(mesh.faces.contents))
    return np.frombuffer(buf, dtype=np.uint32).reshape(n, 9).copy()


def texture_from_rgb8(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    assert c == 3
    img = T.Image3f()
    rc = lib().rt_host_texture_from_rgb8(rgb.ctypes.data, w, h, C.byref(img))
    if rc != 0:
        raise RuntimeError(f"rt_host_texture_from_rgb8 failed: {rc}")
    return img


def image_as_array(image):
    """Zero-copy (H, W, 3) float32 view of an Image3f."""
    w, h = image.size.x, image.size.y
    buf = (C.c_float * (3 * w * h)).from_address(C.addressof(image.pixels.contents))
    return np.frombuffer(buf, dtype=np.float32).reshape(h, w, 3)


def new_image(width, height):
    """A zeroed Image3f backed by a numpy array; returns (image, array)."""
    arr = np.zeros((height, width, 3), dtype=np.float32)
    img = T.Image3f()
    img.size.x, img.size.y = width, height
    img.pixels = C.cast(arr.ctypes.data, C.POINTER(T.Vec3f))
    return img, arr


def quantize(image):
    w, h = image.size.x, image.size.y
    out = np.zeros((h, w, 3), dtype=np.uint8)
    lib().rt_host_image_quantize(C.byref(image), out.ctypes.data)
    return out
