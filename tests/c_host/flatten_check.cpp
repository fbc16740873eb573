/*
 * flatten_check.cpp -- the scene flattener of the HIP library (rt_flatten.cpp: host pointer graph -> flat arrays) driven on
 * the CPU and meant to be compiled with -fsanitize=address,undefined together with rt_flatten.cpp and rt_host_assets.c
 * (tests/test_host_sanitizers.py): a scene of two meshes in five instances with a texture, then the inputs the flattener has
 * to refuse (material, texture coordinate and vertex indices outside their tables, instances without geometry, negative
 * counts, counts without tables).  Prints "ok" and exits 0.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_flatten.h"
#include "rt_host.h"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "check failed, line %d: %s\n", __LINE__, #cond); std::exit(2); } } while (0)

static rt_mesh grid_mesh(uint32_t side, float lift)
{
    const uint32_t position_count = (side + 1) * (side + 1) + 1;
    std::vector<float> positions(position_count * 3, 0.0f), texcoords = {0, 0, 0, 0, 1, 0, 1, 1, 0, 1}, normals = {0, 0, 0, 0, 0, 1};
    for (uint32_t y = 0; y <= side; y++)
        for (uint32_t x = 0; x <= side; x++) {
            float *p = positions.data() + 3 * (1 + y * (side + 1) + x);
            p[0] = (float)x; p[1] = (float)y; p[2] = lift * (float)((x * 7 + y * 13) % 5);
        }
    std::vector<uint32_t> face_vertices(side * side, 4), indices;
    for (uint32_t y = 0; y < side; y++)
        for (uint32_t x = 0; x < side; x++) {
            const uint32_t a = 1 + y * (side + 1) + x, quad[4] = {a, a + 1, a + side + 2, a + side + 1};
            for (uint32_t k = 0; k < 4; k++) { indices.push_back(quad[k]); indices.push_back(1 + k); indices.push_back(1); }
        }
    rt_mesh mesh;
    std::memset(&mesh, 0, sizeof(mesh));
    CHECK(rt_host_mesh_from_obj_arrays(positions.data(), position_count, texcoords.data(), 5, normals.data(), 2, indices.data(),
                                       indices.size() / 3, face_vertices.data(), side * side, &mesh) == 0);
    return mesh;
}

int main()
{
    rt_mesh big = grid_mesh(30, 0.25f), small = grid_mesh(3, 0.5f);
    std::vector<uint8_t> rgb(16 * 8 * 3);
    for (size_t i = 0; i < rgb.size(); i++) rgb[i] = (uint8_t)(i * 37u);
    rt_image3f texture;
    std::memset(&texture, 0, sizeof(texture));
    CHECK(rt_host_texture_from_rgb8(rgb.data(), 16, 8, &texture) == 0);

    std::vector<rt_material> materials(3);
    std::memset(materials.data(), 0, sizeof(rt_material) * materials.size());
    materials[0].diffuse_color = {0.5f, 0.5f, 0.5f};
    materials[1].mirror = true; materials[1].mirror_color = {1, 1, 1};
    materials[2].diffuse_color = {1, 1, 1}; materials[2].textured = true; materials[2].texture = &texture;
    std::vector<rt_sphere> spheres = {{{0, 0, 0}, 2.0f, 0}};
    std::vector<rt_plane> planes = {{{0, 1, 0}, -10.0f, 0}, {{0.6f, 0.8f, 0}, -50.0f, 1}};
    std::vector<rt_light> lights = {{{0, 30, 0}, 2.0f, {1, 1, 1}, 1e4f}};
    std::vector<rt_scene_mesh> instances(5);
    for (size_t i = 0; i < instances.size(); i++) {
        std::memset(&instances[i], 0, sizeof(rt_scene_mesh));
        instances[i].mesh = i % 2 ? &small : &big;
        instances[i].transform.position = {(float)i * 40.0f, 0, 0};
        instances[i].transform.rotX = {1, 0, 0}; instances[i].transform.rotY = {0, 1, 0}; instances[i].transform.rotZ = {0, 0, 1};
        instances[i].material_id = (uint32_t)(i % 3);
    }
    rt_scene scene;
    std::memset(&scene, 0, sizeof(scene));
    scene.camera.position = {0, 10, 85}; scene.camera.lookAt = {0, 0, 0}; scene.camera.up = {0, 1, 0}; scene.camera.field_of_view = 1.0f;
    scene.material_count = (int32_t)materials.size(); scene.materials = materials.data();
    scene.sphere_count = 1; scene.spheres = spheres.data();
    scene.plane_count = 2; scene.planes = planes.data();
    scene.light_count = 1; scene.lights = lights.data();
    scene.mesh_count = (int32_t)instances.size(); scene.meshes = instances.data();

    {
        rt_flat_scene flat;
        CHECK(rt_flatten_scene(&scene, flat) == 0);
        CHECK(flat.meshes.size() == 2 && flat.instances.size() == 5 && flat.textures.size() == 1);
        CHECK(flat.tris.size() == (size_t)big.face_count + small.face_count && flat.attrs.size() == flat.tris.size());
        CHECK(flat.texels.size() == 16 * 8 * 3 && flat.stack_cap >= 2 && !flat.nodes.empty());
        rt_dev_camera cam;
        rt_flatten_camera(&scene.camera, cam);
        CHECK(rt_camera_focal(scene.camera.field_of_view, 480) > 0.0f && rt_pixel_norm(64) > 0.0f);
    }
    auto refused = [&](const char *what) {
        rt_flat_scene flat;
        const int rc = rt_flatten_scene(&scene, flat);
        if (rc == 0 || flat.error.empty()) { std::fprintf(stderr, "not refused: %s\n", what); std::exit(3); }
    };
    spheres[0].material_id = 3; refused("sphere material"); spheres[0].material_id = -1; refused("negative sphere material"); spheres[0].material_id = 0;
    planes[1].material_id = 99; refused("plane material"); planes[1].material_id = 1;
    instances[4].material_id = 3; refused("instance material"); instances[4].material_id = 1;
    instances[2].mesh = nullptr; refused("instance without a mesh"); instances[2].mesh = &big;
    { const uint32_t keep = big.faces[10].ids[1].vertexId; big.faces[10].ids[1].vertexId = big.point_count + 1; refused("vertex index"); big.faces[10].ids[1].vertexId = keep; }
    { const uint32_t keep = small.faces[2].ids[0].texcoordId; small.faces[2].ids[0].texcoordId = 77; refused("texture coordinate index"); small.faces[2].ids[0].texcoordId = keep; }
    { const uint32_t keep = small.faces[1].ids[2].normalId; small.faces[1].ids[2].normalId = 9; refused("normal index"); small.faces[1].ids[2].normalId = keep; }
    scene.plane_count = -1; refused("negative count"); scene.plane_count = 2;
    scene.lights = nullptr; refused("count without a table"); scene.lights = lights.data();
    { rt_image3f none; std::memset(&none, 0, sizeof(none)); materials[2].texture = &none; refused("empty texture"); materials[2].texture = &texture; }
    {
        rt_flat_scene flat;                              /* and it still flattens after all that */
        CHECK(rt_flatten_scene(&scene, flat) == 0);
    }
    rt_host_image_free(&texture);
    rt_host_mesh_free(&big);
    rt_host_mesh_free(&small);
    std::puts("ok");
    return 0;
}
