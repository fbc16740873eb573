/*
 * render_main.c -- a host program in plain C that renders with librt_gpu.so, the way the
 * reference's main() would once it binds the library: the scene is filled in by hand
 * as C records (reference: src/main.c:885-1054), the block that starts the tile threads and calls
 * raytrace_region (src/main.c:1056-1088) is replaced by calls into the C ABI of include/rt_gpu.h.
 *
 *   render_main --probe                      load the library, print the device count, exit 0
 *   render_main OUT.bin W H SPP SEED         render and write the results (needs a GPU)
 *
 * OUT.bin: 8 uint32 (magic, W, H, spp, seed, number of frames that follow, positions, faces of the
 * mesh), the mesh arrays this program made (positions: 3 floats each, entry 0 the parser-style
 * dummy; faces: 4 position indices each), then frames of W*H*3 floats:
 *   0  rt_gpu_render_region             whole image, host pixels
 *   1  rt_gpu_render_region             the same picture rendered as four quarter regions
 *   2  rt_gpu_multi_render_frame        one device
 *   3  rt_gpu_multi_render_frame        one device, through the RCCL path (RT_GPU_MULTI_FORCE_RCCL=1)
 *   4  rt_gpu_render_region_static      64x64 tile at (8, 8), min(SPP, 16) samples, device pixels
 *                                       from hipMalloc, the rest of the frame zero
 * Test infrastructure: tests/test_c_host.py compares every frame with the ctypes render of the same
 * scene, bit for bit.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "rt_gpu.h"
#include "rt_host.h"

#define GRID 24                            /* the mesh: a GRID x GRID sheet of quads with integer-made heights */

static rt_material diffuse(float r, float g, float b)
{
    rt_material m;
    memset(&m, 0, sizeof(m));
    m.diffuse_color.x = r; m.diffuse_color.y = g; m.diffuse_color.z = b;
    return m;
}

static int die(const char *what)
{
    fprintf(stderr, "render_main: %s: %s\n", what, rt_gpu_last_error());
    return 2;
}

int main(int argc, char **argv)
{
    if (argc == 2 && strcmp(argv[1], "--probe") == 0) {
        printf("devices: %d\n", rt_gpu_device_count());
        return 0;
    }
    if (argc != 6) {
        fprintf(stderr, "usage: %s --probe | OUT.bin W H SPP SEED\n", argv[0]);
        return 1;
    }
    const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), spp = (uint32_t)atoi(argv[4]);
    const uint64_t seed = strtoull(argv[5], NULL, 10);
    if (W < 72 || H < 72 || !spp) { fprintf(stderr, "W, H >= 72 and SPP >= 1\n"); return 1; }
    if (rt_gpu_device_count() < 1) { fprintf(stderr, "render_main: no GPU\n"); return 3; }

    /* ---- the scene, as main() writes it: materials :885-929, spheres :933-947, planes :951-977,
     * lights :979-992, camera :1033-1038 */
    rt_material materials[9];
    materials[0] = diffuse(0.5f, 0.5f, 0.5f);
    materials[1] = diffuse(0.0f, 1.0f, 0.0f);
    materials[2] = diffuse(0.0f, 0.0f, 1.0f);
    materials[3] = diffuse(1.0f, 0.0f, 1.0f);
    materials[4] = diffuse(1.0f, 1.0f, 0.0f);
    materials[5] = diffuse(0.0f, 0.0f, 0.0f); materials[5].mirror = true;
    materials[5].mirror_color.x = materials[5].mirror_color.y = materials[5].mirror_color.z = 1.0f;
    materials[6] = diffuse(1.0f, 0.0f, 0.0f);
    materials[7] = diffuse(0.0f, 0.0f, 0.0f); materials[7].transparent = true; materials[7].n = 1.5f;
    materials[8] = diffuse(1.0f, 1.0f, 1.0f);
    rt_sphere spheres[3] = {
        {{0.0f, 0.0f, 0.0f}, 10.0f, 0}, {{-8.0f, 10.0f, 15.0f}, 5.0f, 5}, {{-4.0f, 5.0f, 25.0f}, 4.0f, 7}};
    rt_plane planes[5] = {
        {{0, 0, 1}, -60.0f, 1}, {{0, 1, 0}, -10.0f, 2}, {{1, 0, 0}, -60.0f, 3}, {{-1, 0, 0}, -60.0f, 4}, {{0, -1, 0}, -90.0f, 6}};
    rt_light lights[2] = {
        {{-10.0f, 20.0f, 75.0f}, 3.0f, {1.0f, 1.0f, 1.0f}, 1e5f}, {{10.0f, 17.0f, 75.0f}, 2.0f, {1.0f, 0.2f, 0.1f}, 1e4f}};

    /* a mesh made here (no model file): the arrays an OBJ parser would hand over, quads only */
    const uint32_t n_pos = (GRID + 1) * (GRID + 1) + 1, n_faces = GRID * GRID;
    float *positions = calloc((size_t)n_pos * 3, sizeof(float));
    uint32_t *faces = calloc((size_t)n_faces * 4, sizeof(uint32_t));
    uint32_t *triples = calloc((size_t)n_faces * 4 * 3, sizeof(uint32_t));
    uint32_t *corners = calloc(n_faces, sizeof(uint32_t));
    float texcoords[2] = {0.0f, 0.0f}, normals[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    if (!positions || !faces || !triples || !corners) { fprintf(stderr, "out of memory\n"); return 2; }
    for (uint32_t j = 0; j <= GRID; j++)
        for (uint32_t i = 0; i <= GRID; i++) {
            float *p = positions + 3 * (1 + j * (GRID + 1) + i);
            p[0] = (float)i - 12.0f;
            p[1] = (float)j - 12.0f;
            p[2] = (float)((i * 7u + j * 13u + i * j) % 11u) * 0.25f;       /* exact in binary32 */
        }
    for (uint32_t j = 0; j < GRID; j++)
        for (uint32_t i = 0; i < GRID; i++) {
            uint32_t *f = faces + 4 * (j * GRID + i);
            f[0] = 1 + j * (GRID + 1) + i; f[1] = f[0] + 1; f[2] = f[1] + GRID + 1; f[3] = f[0] + GRID + 1;
            for (int k = 0; k < 4; k++) {
                uint32_t *t = triples + 3 * (4 * (j * GRID + i) + k);
                t[0] = f[k]; t[1] = 0; t[2] = 1;
            }
            corners[j * GRID + i] = 4;
        }
    rt_mesh mesh;
    memset(&mesh, 0, sizeof(mesh));
    int rc = rt_host_mesh_from_obj_arrays(positions, n_pos, texcoords, 1, normals, 2, triples, (uint64_t)n_faces * 4,
                                          corners, n_faces, &mesh);
    if (rc) { fprintf(stderr, "render_main: rt_host_mesh_from_obj_arrays: %d\n", rc); return 2; }
    rt_scene_mesh instances[2];
    memset(instances, 0, sizeof(instances));
    for (int k = 0; k < 2; k++) {
        instances[k].mesh = &mesh;
        instances[k].transform.rotX.x = k ? -1.0f : 1.0f;
        instances[k].transform.rotY.y = 1.0f;
        instances[k].transform.rotZ.z = 1.0f;
    }
    instances[0].transform.position.x = 28.0f; instances[0].transform.position.y = -9.0f; instances[0].transform.position.z = 10.0f;
    instances[0].material_id = 8;
    instances[1].transform.position.x = -30.0f; instances[1].transform.position.y = -8.0f; instances[1].transform.position.z = 20.0f;
    instances[1].material_id = 5;

    rt_scene scene;
    memset(&scene, 0, sizeof(scene));
    scene.camera.position.y = 10.0f; scene.camera.position.z = 85.0f;
    scene.camera.up.y = 1.0f;
    scene.camera.field_of_view = 60.0f / 180.0f * 3.14159265358979323846;      /* src/main.c:1037 */
    scene.material_count = 9; scene.materials = materials;
    scene.sphere_count = 3; scene.spheres = spheres;
    scene.plane_count = 5; scene.planes = planes;
    scene.light_count = 2; scene.lights = lights;
    scene.mesh_count = 2; scene.meshes = instances;

    const size_t frame_floats = (size_t)W * H * 3;
    enum { FRAMES = 5 };
    float *out = calloc(frame_floats * FRAMES, sizeof(float));
    if (!out) { fprintf(stderr, "out of memory\n"); return 2; }
    rt_image3f image;
    image.size.x = W; image.size.y = H;

    /* ---- frames 0 and 1: the drop-in for raytrace_region */
    rt_gpu_scene *gpu_scene = NULL;
    if (rt_gpu_scene_create(&scene, &gpu_scene) != 0) return die("rt_gpu_scene_create");
    rt_rect2u whole = {{0, 0}, {W, H}};
    image.pixels = (rt_vec3f *)out;
    if (rt_gpu_render_region(gpu_scene, &image, whole, spp, seed, NULL) != 0) return die("rt_gpu_render_region");
    image.pixels = (rt_vec3f *)(out + frame_floats);
    for (uint32_t q = 0; q < 4; q++) {
        rt_rect2u part = {{(q & 1) ? W / 2 : 0, (q & 2) ? H / 2 : 0}, {(q & 1) ? W - W / 2 : W / 2, (q & 2) ? H - H / 2 : H / 2}};
        if (rt_gpu_render_region(gpu_scene, &image, part, spp, seed, NULL) != 0) return die("rt_gpu_render_region (quarter)");
    }

    /* ---- frames 2 and 3: the drop-in for the whole render block, one device of the node */
    for (int pass = 0; pass < 2; pass++) {
        if (pass) setenv("RT_GPU_MULTI_FORCE_RCCL", "1", 1);
        rt_gpu_multi *node = NULL;
        if (rt_gpu_multi_create(&scene, 1, &node) != 0) return die("rt_gpu_multi_create");
        image.pixels = (rt_vec3f *)(out + (2 + pass) * frame_floats);
        if (rt_gpu_multi_render_frame(node, &image, spp, seed, NULL) != 0) return die("rt_gpu_multi_render_frame");
        rt_gpu_multi_destroy(node);
    }
    unsetenv("RT_GPU_MULTI_FORCE_RCCL");

    /* ---- frame 4: the entry that allocates nothing, into device memory of the host's own */
    float *device_pixels = NULL;
    if (hipMalloc((void **)&device_pixels, frame_floats * sizeof(float)) != hipSuccess ||
        hipMemset(device_pixels, 0, frame_floats * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    rt_rect2u tile = {{8, 8}, {64, 64}};
    if (rt_gpu_render_region_static(gpu_scene, device_pixels, W, H, tile, spp < 16 ? spp : 16, seed, NULL, NULL) != 0)
        return die("rt_gpu_render_region_static");
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(out + 4 * frame_floats, device_pixels, frame_floats * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        fprintf(stderr, "copy back failed\n"); return 2;
    }
    (void)hipFree(device_pixels);

    rt_gpu_stats stats;
    if (rt_gpu_get_stats(gpu_scene, &stats) != 0) return die("rt_gpu_get_stats");
    printf("rays %llu\n", (unsigned long long)stats.rays);
    rt_gpu_scene_destroy(gpu_scene);

    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 2; }
    const uint32_t header[8] = {0x52544331u, W, H, spp, (uint32_t)seed, FRAMES, n_pos, n_faces};
    int ok = fwrite(header, sizeof(header), 1, f) == 1 &&
             fwrite(positions, sizeof(float), (size_t)n_pos * 3, f) == (size_t)n_pos * 3 &&
             fwrite(faces, sizeof(uint32_t), (size_t)n_faces * 4, f) == (size_t)n_faces * 4 &&
             fwrite(out, sizeof(float), frame_floats * FRAMES, f) == frame_floats * FRAMES;
    ok = fclose(f) == 0 && ok;
    rt_host_mesh_free(&mesh);
    free(out); free(positions); free(faces); free(triples); free(corners);
    if (!ok) { fprintf(stderr, "render_main: short write to %s\n", argv[1]); return 2; }
    return 0;
}
