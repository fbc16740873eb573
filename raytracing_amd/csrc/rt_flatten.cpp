/*
 * rt_flatten.cpp -- host scene graph -> flat arrays for HBM (see
 * rt_device_scene.h for the layout).  Pure host code, no HIP calls.
 *
 * Reads what the reference's render loop reads through Scene*
 * (src/main.h:7-25): materials, spheres, planes, lights, mesh instances, and
 * behind them Mesh (src/math.h:113-129), its pointer-linked box tree
 * (src/math.h:103-111) and texture images (src/math.h:147-151).
 *
 * Must be compiled with -ffp-contract=off: the per-triangle edge vectors and
 * normal and the camera basis are computed here, once, with the very float
 * operations the reference performs per ray (src/main.c:454-461, 753-755), and
 * have to round the same way.
 */
#include "rt_flatten.h"

#include <cmath>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <thread>
#include <queue>

namespace {

struct V3 { float x, y, z; };
inline V3 ld(const rt_vec3f &v) { return V3{v.x, v.y, v.z}; }
inline void st(float *dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b)
{
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float dot(V3 a, V3 b)
{
    float r = 0.0f;
    r += a.x * b.x; r += a.y * b.y; r += a.z * b.z;
    return r;
}
inline V3 unit(V3 a)
{
    float n = std::sqrt(dot(a, a));
    return V3{a.x / n, a.y / n, a.z / n};
}

uint32_t leaf_entry(uint32_t first, uint32_t count)
{
    return RT_ENTRY_LEAF | ((count - 1) << RT_ENTRY_COUNT_SHIFT) | first;
}

struct MeshWalk {
    const rt_mesh *mesh;
    rt_flat_scene &out;
    uint32_t node_base;
    uint32_t deepest = 1;
    bool chained = false;
    std::string error;

    /* a leaf of any size as one entry; oversized leaves become a chain of
     * pseudo records (chunk of 16, then the rest) that the kernel walks in
     * order without any box test */
    uint32_t entry_for_leaf(uint32_t first, uint32_t count)
    {
        if (count <= RT_LEAF_MAX) return leaf_entry(first, count);
        chained = true;
        uint32_t chunks = (count + RT_LEAF_MAX - 1) / RT_LEAF_MAX;
        uint32_t last_first = first + (chunks - 1) * RT_LEAF_MAX;
        uint32_t tail = leaf_entry(last_first, count - (chunks - 1) * RT_LEAF_MAX);
        for (uint32_t k = chunks - 1; k-- > 0;) {
            rt_dev_node rec;
            std::memset(&rec, 0, sizeof(rec));
            rec.flags = RT_NODE_PSEUDO;
            rec.entry0 = leaf_entry(first + k * RT_LEAF_MAX, RT_LEAF_MAX);
            rec.entry1 = tail;
            tail = (uint32_t)out.nodes.size() - node_base;
            out.nodes.push_back(rec);
        }
        return tail;
    }

    bool check_range(const rt_mesh_node *n)
    {
        if (n->first_triangle_id > n->last_triangle_id || n->last_triangle_id >= mesh->face_count) {
            error = "tree node with a triangle range outside the face table";
            return false;
        }
        return true;
    }

    /* breadth-first, so the top levels of every tree sit at the front of its node range */
    bool run(uint32_t *root_entry)
    {
        const rt_mesh_node *root = &mesh->root;
        if (!check_range(root)) return false;
        if (!root->childs) {
            *root_entry = entry_for_leaf(root->first_triangle_id,
                                         root->last_triangle_id - root->first_triangle_id + 1);
            return true;
        }
        struct Item { const rt_mesh_node *node; uint32_t record; uint32_t depth; };
        std::queue<Item> todo;
        size_t budget = 4 * (size_t)mesh->face_count + 16;     /* a proper tree has < 2*faces nodes */
        {
            rt_dev_node rec; std::memset(&rec, 0, sizeof(rec));
            out.nodes.push_back(rec);
        }
        *root_entry = 0;
        todo.push(Item{root, node_base, 1});
        while (!todo.empty()) {
            Item it = todo.front(); todo.pop();
            if (budget-- == 0) { error = "box tree is cyclic or absurdly large"; return false; }
            const rt_mesh_node *kids = it.node->childs;
            uint32_t entry[2];
            for (int k = 0; k < 2; k++) {
                const rt_mesh_node *kid = kids + k;
                if (!check_range(kid)) return false;
                if (it.depth + 1 > deepest) deepest = it.depth + 1;
                if (kid->childs) {
                    rt_dev_node rec; std::memset(&rec, 0, sizeof(rec));
                    entry[k] = (uint32_t)out.nodes.size() - node_base;
                    out.nodes.push_back(rec);
                    todo.push(Item{kid, (uint32_t)out.nodes.size() - 1, it.depth + 1});
                } else {
                    entry[k] = entry_for_leaf(kid->first_triangle_id,
                                              kid->last_triangle_id - kid->first_triangle_id + 1);
                }
            }
            rt_dev_node &rec = out.nodes[it.record];
            st(rec.lo0, ld(kids[0].bbox.min)); st(rec.hi0, ld(kids[0].bbox.max));
            st(rec.lo1, ld(kids[1].bbox.min)); st(rec.hi1, ld(kids[1].bbox.max));
            rec.entry0 = entry[0]; rec.entry1 = entry[1];
            if (out.nodes.size() - node_base >= RT_ENTRY_LEAF) { error = "too many tree nodes"; return false; }
        }
        return true;
    }
};

} // namespace

int rt_flatten_scene(const rt_scene *scene, rt_flat_scene &out)
{
    out = rt_flat_scene();
    if (!scene) { out.error = "scene is NULL"; return -1; }
    if (scene->material_count < 0 || scene->sphere_count < 0 || scene->plane_count < 0 ||
        scene->light_count < 0 || scene->mesh_count < 0) { out.error = "negative element count"; return -2; }
    if ((scene->material_count && !scene->materials) || (scene->sphere_count && !scene->spheres) ||
        (scene->plane_count && !scene->planes) || (scene->light_count && !scene->lights) ||
        (scene->mesh_count && !scene->meshes)) { out.error = "element count without an array"; return -2; }

    rt_flatten_camera(&scene->camera, out.camera);
    out.field_of_view = scene->camera.field_of_view;

    auto material_ok = [&](int32_t id) { return id >= 0 && id < scene->material_count; };

    /* textures, de-duplicated by pointer */
    std::map<const rt_image3f *, int32_t> texture_index;
    for (int32_t i = 0; i < scene->material_count; i++) {
        const rt_material &m = scene->materials[i];
        rt_dev_material d;
        std::memset(&d, 0, sizeof(d));
        st(d.diffuse, ld(m.diffuse_color));
        st(d.mirror_color, ld(m.mirror_color));
        d.ior = m.n;
        d.kind = m.mirror ? 1u : (m.transparent ? 2u : 0u);
        d.texture = -1;
        if (m.textured) {
            const rt_image3f *t = m.texture;
            if (!t || !t->pixels || !t->size.x || !t->size.y) { out.error = "textured material without an image"; return -2; }
            auto found = texture_index.find(t);
            if (found == texture_index.end()) {
                rt_dev_texture dt;
                dt.width = t->size.x; dt.height = t->size.y; dt.pad = 0;
                size_t texels = (size_t)t->size.x * t->size.y;
                if (out.texels.size() / 3 + texels > 0x7FFFFFFFu) { out.error = "textures too large"; return -2; }
                dt.base = (uint32_t)(out.texels.size() / 3);
                const float *src = &t->pixels[0].x;
                out.texels.insert(out.texels.end(), src, src + 3 * texels);
                found = texture_index.emplace(t, (int32_t)out.textures.size()).first;
                out.textures.push_back(dt);
            }
            d.texture = found->second;
        }
        out.materials.push_back(d);
    }

    for (int32_t i = 0; i < scene->light_count; i++) {
        const rt_light &l = scene->lights[i];
        rt_dev_light d;
        st(d.c, ld(l.center)); d.radius = l.radius; st(d.color, ld(l.color)); d.albedo = l.albedo;
        out.lights.push_back(d);
    }
    for (int32_t i = 0; i < scene->sphere_count; i++) {
        const rt_sphere &s = scene->spheres[i];
        if (!material_ok(s.material_id)) { out.error = "sphere with a material index outside the table"; return -2; }
        rt_dev_sphere d;
        st(d.c, ld(s.center)); d.radius = s.radius; d.material = s.material_id;
        out.spheres.push_back(d);
    }
    for (int32_t i = 0; i < scene->plane_count; i++) {
        const rt_plane &p = scene->planes[i];
        if (!material_ok(p.material_id)) { out.error = "plane with a material index outside the table"; return -2; }
        rt_dev_plane d;
        st(d.n, ld(p.normal)); d.d = p.d; d.material = p.material_id;
        out.planes.push_back(d);
    }

    /* meshes, de-duplicated by pointer (the reference scene instances one dog three times) */
    std::map<const rt_mesh *, uint32_t> mesh_index;
    out.stack_cap = 2;
    for (int32_t i = 0; i < scene->mesh_count; i++) {
        const rt_scene_mesh &sm = scene->meshes[i];
        const rt_mesh *mesh = sm.mesh;
        if (!mesh || !mesh->faces || !mesh->points || mesh->face_count == 0) { out.error = "mesh instance without geometry"; return -2; }
        if (!material_ok((int32_t)sm.material_id)) { out.error = "mesh instance with a material index outside the table"; return -2; }
        auto found = mesh_index.find(mesh);
        if (found == mesh_index.end()) {
            if (mesh->face_count > RT_ENTRY_FIRST_MASK) { out.error = "mesh has too many triangles"; return -2; }
            rt_dev_mesh dm;
            std::memset(&dm, 0, sizeof(dm));
            dm.tri_base = (uint32_t)out.tris.size();
            dm.node_base = (uint32_t)out.nodes.size();
            st(dm.root_lo, ld(mesh->root.bbox.min)); st(dm.root_hi, ld(mesh->root.bbox.max));

            /* one record pair per triangle, gathered through the face's indices; the faces are independent, so
             * large meshes are cut into ranges that a few threads fill side by side */
            const size_t tri0 = out.tris.size();
            out.tris.resize(tri0 + mesh->face_count);
            out.attrs.resize(tri0 + mesh->face_count);
            std::atomic<bool> bad_index(false);
            auto gather = [&](uint32_t from, uint32_t to) {
                for (uint32_t f = from; f < to; f++) {
                    const rt_triangle &tri = mesh->faces[f];
                    for (int k = 0; k < 3; k++) {
                        if (tri.ids[k].vertexId >= mesh->point_count ||
                            (mesh->normals && tri.ids[k].normalId >= mesh->normal_count) ||
                            (mesh->texcoords && tri.ids[k].texcoordId >= mesh->texcoord_count)) {
                            bad_index = true; return;
                        }
                    }
                    V3 a = ld(mesh->points[tri.ids[0].vertexId]);
                    V3 b = ld(mesh->points[tri.ids[1].vertexId]);
                    V3 c = ld(mesh->points[tri.ids[2].vertexId]);
                    V3 e1 = sub(b, a), e2 = sub(c, a);
                    rt_dev_tri &dt = out.tris[tri0 + f];
                    st(dt.a, a); st(dt.e1, e1); st(dt.e2, e2); st(dt.n, cross(e1, e2));
                    rt_dev_attr &da = out.attrs[tri0 + f];
                    std::memset(&da, 0, sizeof(da));
                    if (mesh->normals) {
                        st(da.n0, ld(mesh->normals[tri.ids[0].normalId]));
                        st(da.n1, ld(mesh->normals[tri.ids[1].normalId]));
                        st(da.n2, ld(mesh->normals[tri.ids[2].normalId]));
                    }
                    if (mesh->texcoords) {
                        const rt_vec2f &t0 = mesh->texcoords[tri.ids[0].texcoordId];
                        const rt_vec2f &t1 = mesh->texcoords[tri.ids[1].texcoordId];
                        const rt_vec2f &t2 = mesh->texcoords[tri.ids[2].texcoordId];
                        da.uv0[0] = t0.x; da.uv0[1] = t0.y; da.uv1[0] = t1.x; da.uv1[1] = t1.y;
                        da.uv2[0] = t2.x; da.uv2[1] = t2.y;
                    }
                }
            };
            {
                unsigned workers = mesh->face_count >= 16384u ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
                std::vector<std::thread> pool;
                const uint32_t share = (mesh->face_count + workers - 1) / workers;
                for (unsigned w = 1; w < workers; w++) {
                    const uint32_t from = std::min(mesh->face_count, w * share), to = std::min(mesh->face_count, (w + 1) * share);
                    try { pool.emplace_back(gather, from, to); } catch (...) { gather(from, to); }
                }
                gather(0, std::min(mesh->face_count, share));
                for (auto &t : pool) t.join();
            }
            if (bad_index) { out.error = "triangle with an index outside its attribute table"; return -2; }

            MeshWalk walk{mesh, out, dm.node_base};
            if (!walk.run(&dm.root_entry)) { out.error = walk.error; return -2; }
            {   /* are all box coordinates in the range where the kernels' shared-reciprocal slab test is exact? */
                auto tame = [](float v) { float a = std::fabs(v); return v == 0.0f || (a >= 0x1p-40f && a <= 0x1p40f); };
                bool ok = true;
                for (int k = 0; k < 3; k++) ok = ok && tame(dm.root_lo[k]) && tame(dm.root_hi[k]);
                for (size_t n = dm.node_base; ok && n < out.nodes.size(); n++) {
                    const rt_dev_node &nd = out.nodes[n];
                    if (nd.flags & RT_NODE_PSEUDO) continue;
                    for (int k = 0; k < 3; k++)
                        ok = ok && tame(nd.lo0[k]) && tame(nd.hi0[k]) && tame(nd.lo1[k]) && tame(nd.hi1[k]);
                }
                dm.tame_boxes = ok ? 1u : 0u;
            }
            dm.node_count = (uint32_t)(out.nodes.size() - dm.node_base);
            dm.stack_need = walk.deepest + 2 + (walk.chained ? 2 : 0);
            if (dm.stack_need > out.stack_cap) out.stack_cap = dm.stack_need;
            found = mesh_index.emplace(mesh, (uint32_t)out.meshes.size()).first;
            out.meshes.push_back(dm);
        }
        rt_dev_instance di;
        st(di.pos, ld(sm.transform.position));
        st(di.rx, ld(sm.transform.rotX)); st(di.ry, ld(sm.transform.rotY)); st(di.rz, ld(sm.transform.rotZ));
        di.mesh = found->second;
        di.material = (int32_t)sm.material_id;
        out.instances.push_back(di);
    }
    return 0;
}

/* camera basis, src/main.c:753-755 (the focal length depends on the image
 * height and is set per render call) */
void rt_flatten_camera(const rt_camera *camera, rt_dev_camera &out)
{
    V3 pos = ld(camera->position);
    V3 forward = unit(sub(ld(camera->lookAt), pos));
    V3 right = unit(cross(forward, ld(camera->up)));
    V3 up = cross(right, forward);
    st(out.pos, pos); st(out.forward, forward); st(out.right, right); st(out.up, up);
    out.focal = 0.0f;
}

/* src/main.c:767: (float)height / (2.0f * tanf(fov / 2.0f)) */
float rt_camera_focal(float field_of_view, uint32_t image_height)
{
    return (float)image_height / (2.0f * tanf(field_of_view / 2.0f));
}

/* src/main.c:760,788: 1.0f / (sigma*sigma*2.0f * M_PI * spp), sigma = 0.5f; the
 * float product 0.5f is widened, the rest is double, the result narrowed */
float rt_pixel_norm(uint32_t spp_total)
{
    float sigma = 0.5f;
    return (float)(1.0f / (sigma * sigma * 2.0f * M_PI * spp_total));
}
