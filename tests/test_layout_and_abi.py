"""The drop-in boundary: record layouts and exported symbols (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import _libs as L
from raytracing_amd import gpu, scene_types as T

ROOT = L.ROOT


def test_ctypes_records_match_the_c_header(tmp_path):
    """sizeof() of every record, as C compiles include/rt_scene.h, equals the ctypes mirror
    (and the sizes measured on the reference: src/main.h, src/math.h)."""
    names = ["rt_scene", "rt_material", "rt_sphere", "rt_plane", "rt_light", "rt_scene_mesh", "rt_mesh",
             "rt_mesh_node", "rt_triangle", "rt_ray_result", "rt_job", "rt_random_serie", "rt_image3f",
             "rt_camera", "rt_transform", "rt_ray"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rt_scene.h"\nint main(void){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.split()]
    assert got == [n for _, n in T.LAYOUT_SIZES]
    assert [C.sizeof(t) for t, _ in T.LAYOUT_SIZES] == got


@pytest.mark.skipif(not L.have_ref("strict"), reason="compiled reference not present")
def test_records_match_the_compiled_reference():
    out = (C.c_uint32 * 16)()
    L.ref("strict").ref_sizes(C.byref(out))
    assert list(out) == [n for _, n in T.LAYOUT_SIZES]


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_gpu_library_exports_every_declared_entry_point():
    """librt_gpu.so loads on a machine without a GPU and exports exactly what include/rt_gpu.h declares."""
    if not os.path.exists(gpu.LIB_PATH):
        pytest.fail("raytracing_amd/librt_gpu.so is not built: run __graft_entry__.build()")
    lib = C.CDLL(gpu.LIB_PATH)
    declared = _declared("rt_gpu.h", "rt_gpu_")
    assert declared == sorted(gpu.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_host_library_exports_every_declared_entry_point():
    lib = C.CDLL(os.path.join(ROOT, "raytracing_amd", "librt_host.so"))
    declared = _declared("rt_host.h", "rt_host_")
    assert len(declared) == 11
    for name in declared:
        assert getattr(lib, name) is not None


def test_product_does_not_reach_into_the_oracle():
    """Nothing under raytracing_amd/ or include/ may import, include or link the checkers."""
    bad = []
    for base in ("raytracing_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".c", ".cpp", ".hip", "Makefile")):
                    text = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"oracle/|rt_oracle|librt_oracle|_ref/|libref_", text):
                        bad.append(os.path.join(d, f))
    assert bad == []


def test_work_model_constants_are_the_record_sizes():
    """include/rt_work_model.h prices the counted work with the sizes of the reference's records
    (src/math.h, src/main.h): a constant that drifts from the layout would skew every roofline."""
    text = open(os.path.join(ROOT, "include", "rt_work_model.h")).read()
    m = {k: int(v) for k, v in re.findall(r"#define\s+(RT_(?:BYTES|FLOPS)_\w+)\s+(\d+)", text)}
    size = {t.__name__: n for t, n in T.LAYOUT_SIZES}
    assert m["RT_BYTES_LIGHT_TEST"] == size["Light"] and m["RT_BYTES_SPHERE_TEST"] == size["Sphere"]
    assert m["RT_BYTES_PLANE_TEST"] == size["Plane"]
    assert m["RT_BYTES_ROOT_TEST"] == size["Transform"] + 24
    assert m["RT_BYTES_INTERIOR_POP"] == 2 * size["MeshNode"]
    assert m["RT_BYTES_TRI_TEST"] == size["Triangle"] + 3 * 12
    assert m["RT_BYTES_TRI_HIT"] == 3 * 12 + 3 * 8 and m["RT_BYTES_TEX_LOOKUP"] == 12 and m["RT_BYTES_PIXEL"] == 12
    assert len([k for k in m if k.startswith("RT_FLOPS_")]) == 8
    import bench
    assert bench.work_model() == m


@pytest.mark.skipif(not os.path.exists("/root/reference/src/main.c"), reason="reference sources not present")
def test_documented_binding_compiles_inside_the_reference_program(tmp_path):
    """The block a maintainer would put into src/main.c: rt_gpu.h with the host's own records as its
    types, the render block of src/main.c:1056-1088 replaced by library calls.  Compiled (not run)
    against the reference's own source with its build.sh flags."""
    src = tmp_path / "with_gpu.c"
    src.write_text('''
#define main ref_program_main
#include "/root/reference/src/main.c"
#undef main
#define RT_SCENE_TYPES_FROM_HOST
#include "rt_gpu.h"
int render_with_gpu(Scene *scene, Image3f *images, int count)
{
    rt_gpu_scene *gpu_scene = NULL;
    if (rt_gpu_scene_create(scene, &gpu_scene) != 0) { fprintf(stderr, "%s\\n", rt_gpu_last_error()); return -1; }
    for (int i = 0; i < count; i++) {
        Rect2i whole = { .point = {0, 0}, .size = {IMAGE_SIZE, IMAGE_SIZE} };
        if (rt_gpu_render_region(gpu_scene, images + i, whole, RAY_PER_PIXEL, i, NULL) != 0) return -2;
    }
    Job job; memset(&job, 0, sizeof(job));
    Ray ray; Ray_Result hit; memset(&ray, 0, sizeof(ray));
    if (rt_gpu_render_jobs(gpu_scene, &job, 1, NULL) != 0) return -3;
    if (rt_gpu_find_ray_hit(gpu_scene, &ray, 1, 1e30f, &hit) != 0) return -4;
    rt_gpu_scene_destroy(gpu_scene);
    return 0;
}
''')
    obj = tmp_path / "with_gpu.o"
    r = subprocess.run(["gcc", "-c", str(src), "-o", str(obj), "-O1", "-maes", "-mfma", "-D_REENTRANT", "-DUSE_THREADS=1",
                        "-I", os.path.join(ROOT, "include"), "-w"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert obj.stat().st_size > 0


def test_options_and_stats_records_match_the_c_header(tmp_path):
    """rt_gpu_options / rt_gpu_stats as C lays them out against the ctypes mirrors in raytracing_amd/gpu.py."""
    src = tmp_path / "opts.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rt_gpu.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rt_gpu_options), offsetof(rt_gpu_options, frames), '
                   'offsetof(rt_gpu_options, frame_stride), sizeof(rt_gpu_stats), offsetof(rt_gpu_stats, kernel_ms), '
                   'offsetof(rt_gpu_stats, launches)); return 0;}\n')
    exe = tmp_path / "opts"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.split()]
    assert got == [C.sizeof(gpu.Options), gpu.Options.frames.offset, gpu.Options.frame_stride.offset,
                   C.sizeof(gpu.Stats), gpu.Stats.kernel_ms.offset, gpu.Stats.launches.offset]
