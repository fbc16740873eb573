/*
 * rt_scene.h -- the scene description the GPU renderer reads.
 *
 * These types are byte-for-byte layout twins of the host program's own scene
 * API (reference: src/math.h:20-190 for the math/geometry records and
 * src/main.h:7-72 for Scene / Ray / Ray_Result / Job).  The renderer never
 * owns a scene: the host fills these records exactly as it does today and
 * hands over a pointer.
 *
 * Two ways to use this header:
 *   - stand-alone (tests, tools, bindings): include it as is, the rt_* names
 *     below are defined here;
 *   - inside the host program, which already defines Scene, Image3f, ... :
 *     define RT_SCENE_TYPES_FROM_HOST before including rt_gpu.h and the rt_*
 *     names become aliases of the host's own types (see rt_gpu.h).
 *
 * Every size and field offset is pinned by a static assertion at the bottom
 * so a drifting layout fails at compile time, not at render time.
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef RT_SCENE_TYPES_FROM_HOST

/* ---- small vectors (reference: src/math.h:20-76) ------------------------ */
typedef struct rt_vec2u { uint32_t x, y; } rt_vec2u;       /* Vec2i: x|width, y|height */
typedef struct rt_vec2f { float x, y; } rt_vec2f;          /* Vec2f */
typedef struct rt_vec3f { float x, y, z; } rt_vec3f;       /* Vec3f: xyz | rgb */
typedef struct rt_vec4f { float x, y, z, w; } rt_vec4f;    /* Vec4f */
typedef struct rt_rect2u { rt_vec2u point, size; } rt_rect2u; /* Rect2i (src/math.h:46-50) */

/* ---- RNG state of one job (reference: src/math.h:11-18) ------------------
 * 32 bytes, 16-byte aligned (the host keeps it in two SSE registers). */
typedef struct rt_random_serie {
    uint8_t bytes[32];
} __attribute__((aligned(16))) rt_random_serie;

/* ---- mesh records (reference: src/math.h:78-137) ------------------------ */
typedef struct rt_transform {           /* rigid placement of a mesh instance */
    rt_vec3f position;
    rt_vec3f rotX, rotY, rotZ;          /* images of the mesh-space axes in world space */
} rt_transform;

typedef struct rt_vertex_ref {          /* Vertex: three 1-based OBJ indices */
    uint32_t vertexId, texcoordId, normalId;
} rt_vertex_ref;

typedef struct rt_triangle { rt_vertex_ref ids[3]; } rt_triangle;

typedef struct rt_bbox { rt_vec3f min, max; } rt_bbox;

typedef struct rt_mesh_node {
    rt_bbox bbox;
    struct rt_mesh_node *childs;        /* NULL for a leaf, else two adjacent nodes */
    uint32_t first_triangle_id;
    uint32_t last_triangle_id;          /* inclusive */
} rt_mesh_node;

typedef struct rt_mesh {
    uint32_t      point_count;    rt_vec3f    *points;
    uint32_t      texcoord_count; rt_vec2f    *texcoords;
    uint32_t      normal_count;   rt_vec3f    *normals;
    uint32_t      face_count;     rt_triangle *faces;
    uint32_t      tree_depth;     /* depth of the deepest node, root = 1 */
    rt_mesh_node  root;
} rt_mesh;

typedef struct rt_scene_mesh {
    rt_mesh     *mesh;
    rt_transform transform;
    uint32_t     material_id;
} rt_scene_mesh;

/* ---- camera, image, material, primitives (reference: src/math.h:139-190) - */
typedef struct rt_camera {
    rt_vec3f position, lookAt, up;
    float    field_of_view;             /* radians */
} rt_camera;

typedef struct rt_image3f {
    rt_vec2u  size;                     /* x = width, y = height */
    rt_vec3f *pixels;                   /* row-major, row 0 on top; no alignment promised */
} rt_image3f;

typedef struct rt_material {
    rt_vec3f    diffuse_color;
    bool        mirror;
    rt_vec3f    mirror_color;
    bool        transparent;
    float       n;                      /* refraction index */
    bool        textured;
    rt_image3f *texture;
} rt_material;

typedef struct rt_sphere { rt_vec3f center; float radius; int32_t material_id; } rt_sphere;
typedef struct rt_light  { rt_vec3f center; float radius; rt_vec3f color; float albedo; } rt_light;
typedef struct rt_plane  { rt_vec3f normal; float d; int32_t material_id; } rt_plane;

/* ---- scene, ray, hit record, job (reference: src/main.h:7-72) ------------ */
typedef struct rt_scene {
    rt_camera camera;
    int32_t material_count; rt_material   *materials;
    int32_t sphere_count;   rt_sphere     *spheres;
    int32_t plane_count;    rt_plane      *planes;
    int32_t light_count;    rt_light      *lights;
    int32_t mesh_count;     rt_scene_mesh *meshes;
} rt_scene;

typedef struct rt_ray { rt_vec3f origin, direction; } rt_ray;

typedef enum rt_object_type { RT_SPHERE = 0, RT_PLANE = 1, RT_LIGHT = 2, RT_TRIANGLE = 3 } rt_object_type;

typedef struct rt_ray_result {
    bool           hit;
    rt_object_type object_type;
    uint32_t       object_id;
    uint32_t       mesh_id;
    rt_vec3f       intersection_point;
    rt_vec3f       surface_normal;
    float          t_min;
    float          triangle_alpha, triangle_beta, triangle_gamma;
    rt_vec2f       triangle_uv;
    bool           enter_shape;
    int32_t        material_id;
} rt_ray_result;

typedef struct rt_job {
    rt_scene       *scene;
    rt_image3f     *image;
    rt_rect2u       region;
    uint32_t        ray_per_pixel;
    rt_random_serie serie;
} rt_job;

#define RT_LAYOUT_CHECK(type, size) _Static_assert(sizeof(type) == (size), #type " size drifted")
#define RT_OFFSET_CHECK(type, field, off) _Static_assert(offsetof(type, field) == (off), #type "." #field " moved")
#ifdef __cplusplus
#undef RT_LAYOUT_CHECK
#undef RT_OFFSET_CHECK
#define RT_LAYOUT_CHECK(type, size) static_assert(sizeof(type) == (size), #type " size drifted")
#define RT_OFFSET_CHECK(type, field, off) static_assert(offsetof(type, field) == (off), #type "." #field " moved")
#endif

RT_LAYOUT_CHECK(rt_vec3f, 12);       RT_LAYOUT_CHECK(rt_rect2u, 16);
RT_LAYOUT_CHECK(rt_random_serie, 32);
RT_LAYOUT_CHECK(rt_transform, 48);   RT_LAYOUT_CHECK(rt_triangle, 36);
RT_LAYOUT_CHECK(rt_mesh_node, 40);   RT_OFFSET_CHECK(rt_mesh_node, childs, 24);
RT_LAYOUT_CHECK(rt_mesh, 112);       RT_OFFSET_CHECK(rt_mesh, faces, 56);
RT_OFFSET_CHECK(rt_mesh, tree_depth, 64); RT_OFFSET_CHECK(rt_mesh, root, 72);
RT_LAYOUT_CHECK(rt_scene_mesh, 64);  RT_OFFSET_CHECK(rt_scene_mesh, material_id, 56);
RT_LAYOUT_CHECK(rt_camera, 40);      RT_LAYOUT_CHECK(rt_image3f, 16);
RT_LAYOUT_CHECK(rt_material, 48);    RT_OFFSET_CHECK(rt_material, mirror_color, 16);
RT_OFFSET_CHECK(rt_material, n, 32); RT_OFFSET_CHECK(rt_material, texture, 40);
RT_LAYOUT_CHECK(rt_sphere, 20);      RT_LAYOUT_CHECK(rt_light, 32);
RT_LAYOUT_CHECK(rt_plane, 20);
RT_LAYOUT_CHECK(rt_scene, 120);      RT_OFFSET_CHECK(rt_scene, materials, 48);
RT_OFFSET_CHECK(rt_scene, meshes, 112);
RT_LAYOUT_CHECK(rt_ray, 24);         RT_LAYOUT_CHECK(rt_ray_result, 72);
RT_OFFSET_CHECK(rt_ray_result, t_min, 40); RT_OFFSET_CHECK(rt_ray_result, material_id, 68);
RT_LAYOUT_CHECK(rt_job, 80);         RT_OFFSET_CHECK(rt_job, serie, 48);

#endif /* !RT_SCENE_TYPES_FROM_HOST */

#ifdef __cplusplus
}
#endif
#endif /* RT_SCENE_H */
