"""Scenes as JSON text, read into the scene records the render loop takes.

The reference hard-codes its scene in main() (src/main.c:885-1054): materials, spheres, planes,
lights, mesh instances and the camera are C initialisers.  This is the same data as a document:

    {"name": "...",
     "camera": {"position": [x,y,z], "look_at": [x,y,z], "up": [x,y,z], "fov_degrees": 60},
     "materials": [{"diffuse": [r,g,b]} | {"mirror": [r,g,b]} | {"glass": 1.5} |
                   {"diffuse": [r,g,b], "texture": "name"} |
                   {"mtl": "file.mtl", "name": "newmtl name"}],     the model's own material library
     "spheres":   [{"center": [x,y,z], "radius": r, "material": i}],
     "planes":    [{"normal": [x,y,z], "d": d, "material": i}],
     "lights":    [{"center": [x,y,z], "radius": r, "color": [r,g,b], "albedo": a}],
     "textures":  {"name": "file.npz"},          decoded 8-bit RGB, array "rgb" (H, W, 3)
     "meshes":    {"name": "file.npz"},          parsed OBJ arrays (see host.mesh_from_obj_arrays)
     "instances": [{"mesh": "name", "position": [x,y,z], "material": i, "mirrored_x": false}]}

An "mtl" material takes its data from a Wavefront material library, the data the OBJ parser of the
reference already reads and the reference then ignores (src/fast_obj.h:49-77: Kd, Ni, d, map_Kd):
diffuse colour = Kd; dissolve d < 1 makes it glass with refraction index Ni; map_Kd names the
diffuse map, looked up among the document's textures by file name without extension (the decoded
pixels come from the document's "textures" files: decoding images stays with the host program).
File names are relative to the document.  Numbers become C floats exactly as the C initialisers
would (a decimal literal rounded to float); "fov_degrees" goes through the reference's own
expression (src/main.c:1037), "fov" gives radians directly.  Nothing here renders: the result is a
scenes.SceneHandle, the records of include/rt_scene.h.
"""
import json
import os

import numpy as np

from . import host, scenes
from . import scene_types as T


class SceneFileError(ValueError):
    pass


def _vec(v, what):
    if not (isinstance(v, (list, tuple)) and len(v) == 3 and all(isinstance(x, (int, float)) for x in v)):
        raise SceneFileError(f"{what}: expected three numbers, got {v!r}")
    return T.Vec3f(float(v[0]), float(v[1]), float(v[2]))


def _index(v, count, what):
    if not isinstance(v, int) or not 0 <= v < count:
        raise SceneFileError(f"{what}: material index {v!r} outside 0..{count - 1}")
    return v


def parse_mtl(text):
    """Wavefront .mtl text -> {material name: {"Kd": (r,g,b), "Ni": n, "d": alpha, "map_Kd": file, ...}}
    with the defaults of the reference's parser (src/fast_obj.h:mtl_default: Kd 1, Ni 1, d 1)."""
    out, cur = {}, None
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].split()
        if not line:
            continue
        key, args = line[0], line[1:]
        if key == "newmtl":
            cur = out.setdefault(" ".join(args), {"Kd": (1.0, 1.0, 1.0), "Ni": 1.0, "d": 1.0})
        elif cur is None:
            continue
        elif key in ("Ka", "Kd", "Ks", "Ke", "Tf") and len(args) >= 3:
            cur[key] = tuple(float(x) for x in args[:3])
        elif key in ("Ns", "Ni", "d") and args:
            cur[key] = float(args[0])
        elif key == "Tr" and args:                       # the parser reads Tr as 1 - d
            cur["d"] = 1.0 - float(args[0])
        elif key == "illum" and args:
            cur[key] = int(float(args[0]))
        elif key.startswith("map_") or key == "bump":
            cur["map_bump" if key == "bump" else key] = args[-1] if args else ""
    return out


def _material_from_mtl(m, textures, base, where):
    path = os.path.join(base, m["mtl"])
    try:
        with open(path) as f:
            library = parse_mtl(f.read())
    except OSError as e:
        raise SceneFileError(f"{where}: cannot read {path}: {e}")
    name = m.get("name") or (next(iter(library)) if len(library) == 1 else None)
    if name not in library:
        raise SceneFileError(f"{where}: material {name!r} is not in {m['mtl']} (it has {sorted(library)})")
    entry = library[name]
    if entry["d"] < 1.0:
        return scenes._material(transparent=True, n=float(entry["Ni"]))
    tex = None
    if entry.get("map_Kd"):
        stem = os.path.splitext(os.path.basename(entry["map_Kd"].replace("\\", "/")))[0]
        if stem not in textures:
            raise SceneFileError(f"{where}: map_Kd {entry['map_Kd']!r} needs a texture named {stem!r} in the document")
        tex = textures[stem]
    return scenes._material(diffuse=entry["Kd"], texture=tex)


def _material(m, textures, where, base="."):
    if "mtl" in m:
        return _material_from_mtl(m, textures, base, where)
    kinds = [k for k in ("diffuse", "mirror", "glass") if k in m]
    if len(kinds) != 1:
        raise SceneFileError(f"{where}: exactly one of diffuse / mirror / glass, got {sorted(m)}")
    if "mirror" in m:
        c = _vec(m["mirror"], where)
        return scenes._material(mirror=True, mirror_color=(c.x, c.y, c.z))
    if "glass" in m:
        return scenes._material(transparent=True, n=float(m["glass"]))
    c = _vec(m["diffuse"], where)
    tex = None
    if "texture" in m:
        if m["texture"] not in textures:
            raise SceneFileError(f"{where}: unknown texture {m['texture']!r}")
        tex = textures[m["texture"]]
    return scenes._material(diffuse=(c.x, c.y, c.z), texture=tex)


def load(source, base_dir=None):
    """source: a path to a JSON file, or the document as a dict.  Returns a scenes.SceneHandle."""
    if isinstance(source, dict):
        doc, base = source, base_dir or os.getcwd()
    else:
        with open(source) as f:
            doc = json.load(f)
        base = base_dir or os.path.dirname(os.path.abspath(source))
    if not isinstance(doc, dict) or "camera" not in doc:
        raise SceneFileError("a scene document is an object with at least a camera")

    textures, meshes, keep = {}, {}, []
    for name, rel in doc.get("textures", {}).items():
        with np.load(os.path.join(base, rel), allow_pickle=False) as z:
            textures[name] = host.texture_from_rgb8(z["rgb"])
        keep.append(textures[name])
    for name, rel in doc.get("meshes", {}).items():
        with np.load(os.path.join(base, rel), allow_pickle=False) as z:
            meshes[name] = host.mesh_from_obj_arrays(z["positions"], z["texcoords"], z["normals"],
                                                     z["indices"], z["face_vertices"])
        keep.append(meshes[name])

    cam = doc["camera"]
    if ("fov" in cam) == ("fov_degrees" in cam):
        raise SceneFileError("camera: give fov (radians) or fov_degrees, not both or neither")
    fov = scenes.fov_radians(cam["fov_degrees"]) if "fov_degrees" in cam else float(cam["fov"])
    camera = T.Camera(_vec(cam["position"], "camera.position"), _vec(cam["look_at"], "camera.look_at"),
                      _vec(cam["up"], "camera.up"), fov)

    mats = [_material(m, textures, f"materials[{i}]", base) for i, m in enumerate(doc.get("materials", []))]
    n_mat = len(mats)
    spheres = [T.Sphere(_vec(s["center"], f"spheres[{i}].center"), float(s["radius"]),
                        _index(s["material"], n_mat, f"spheres[{i}]"))
               for i, s in enumerate(doc.get("spheres", []))]
    planes = [T.Plane(_vec(p["normal"], f"planes[{i}].normal"), float(p["d"]),
                      _index(p["material"], n_mat, f"planes[{i}]"))
              for i, p in enumerate(doc.get("planes", []))]
    lights = [T.Light(_vec(l["center"], f"lights[{i}].center"), float(l["radius"]),
                      _vec(l["color"], f"lights[{i}].color"), float(l["albedo"]))
              for i, l in enumerate(doc.get("lights", []))]
    instances = []
    for i, inst in enumerate(doc.get("instances", [])):
        if inst.get("mesh") not in meshes:
            raise SceneFileError(f"instances[{i}]: unknown mesh {inst.get('mesh')!r}")
        p = _vec(inst["position"], f"instances[{i}].position")
        instances.append(scenes.instance(meshes[inst["mesh"]], (p.x, p.y, p.z),
                                         _index(inst["material"], n_mat, f"instances[{i}]"),
                                         mirrored_x=bool(inst.get("mirrored_x", False))))
    return scenes.SceneHandle(doc.get("name", "scene"), camera, mats, spheres, planes, lights, instances, keep=keep)


def describe(handle, textures=None, meshes=None, objects=None):
    """The document of a SceneHandle, given the files its textures and meshes came from
    ({"name": "file.npz"}).  A scene with several meshes or textures also needs `objects`
    ({"name": the Mesh or Image3f record the scene points at}) to tell which is which; with one
    of each the only name is used.  Floats are written with 9 significant digits, which a C float
    survives unchanged."""
    import ctypes as C

    def v(x):
        return [float(np.float32(x.x)), float(np.float32(x.y)), float(np.float32(x.z))]
    s = handle.scene
    by_address = {C.addressof(obj): name for name, obj in (objects or {}).items()}

    def name_of(pointer, names, what):
        if not pointer:
            raise SceneFileError(f"{what}: null pointer")
        address = C.addressof(pointer.contents)
        if address in by_address:
            return by_address[address]
        if names and len(names) == 1:
            return next(iter(names))
        raise SceneFileError(f"{what}: cannot tell which file it came from (pass objects=)")
    mats = []
    for i in range(s.material_count):
        m = s.materials[i]
        if m.mirror:
            mats.append({"mirror": v(m.mirror_color)})
        elif m.transparent:
            mats.append({"glass": float(np.float32(m.n))})
        else:
            d = {"diffuse": v(m.diffuse_color)}
            if m.textured:
                d["texture"] = name_of(m.texture, textures, f"materials[{i}].texture")
            mats.append(d)
    doc = {
        "name": handle.name,
        "camera": {"position": v(s.camera.position), "look_at": v(s.camera.lookAt), "up": v(s.camera.up),
                   "fov": float(np.float32(s.camera.field_of_view))},
        "materials": mats,
        "spheres": [{"center": v(s.spheres[i].center), "radius": float(np.float32(s.spheres[i].radius)),
                     "material": int(s.spheres[i].material_id)} for i in range(s.sphere_count)],
        "planes": [{"normal": v(s.planes[i].normal), "d": float(np.float32(s.planes[i].d)),
                    "material": int(s.planes[i].material_id)} for i in range(s.plane_count)],
        "lights": [{"center": v(s.lights[i].center), "radius": float(np.float32(s.lights[i].radius)),
                    "color": v(s.lights[i].color), "albedo": float(np.float32(s.lights[i].albedo))}
                   for i in range(s.light_count)],
        "textures": dict(textures or {}),
        "meshes": dict(meshes or {}),
        "instances": [{"mesh": name_of(s.meshes[i].mesh, meshes, f"instances[{i}].mesh"), "position": v(s.meshes[i].transform.position),
                       "material": int(s.meshes[i].material_id),
                       "mirrored_x": bool(s.meshes[i].transform.rotX.x < 0)} for i in range(s.mesh_count)],
    }
    return doc
