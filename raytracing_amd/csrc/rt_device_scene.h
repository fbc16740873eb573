/*
 * rt_device_scene.h -- the scene as the kernels see it in HBM.
 *
 * The host scene graph is a web of pointers (reference: src/main.h:7-25,
 * src/math.h:103-137).  The flattener (rt_flatten.cpp) turns it into a handful
 * of flat arrays; this header is the contract between flattener and kernels.
 *
 *   nodes      one 64-byte record per INTERIOR tree node: the boxes of its two
 *              children and a 32-bit "entry" for each child.  The reference
 *              tests both children's boxes when it pops a node
 *              (src/main.c:402-447), so one record = one pop = one 64-byte line.
 *   entries    bit 31 clear: index of an interior node record.
 *              bit 31 set:   a leaf; bits 30..27 = triangle count - 1 (1..16),
 *              bits 26..0 = index of its first triangle.  Leaves need no record.
 *   tris       one 48-byte record per triangle, in the host's (permuted) face
 *              order so that triangle index == the reference's object_id:
 *              vertex a, edges e1 = b-a, e2 = c-a, and n = e1 x e2, computed on
 *              the host with the same single-rounding float operations the
 *              reference performs per test (src/main.c:450-461).
 *   attrs      one 64-byte record per triangle: the three vertex normals and
 *              texture coordinates, gathered through normalId / texcoordId.
 *   textures   float RGB texels (12 bytes), all textures back to back.
 */
#ifndef RT_DEVICE_SCENE_H
#define RT_DEVICE_SCENE_H

#include <stdint.h>

#define RT_ENTRY_LEAF        0x80000000u
#define RT_ENTRY_COUNT_SHIFT 27
#define RT_ENTRY_FIRST_MASK  0x07FFFFFFu
#define RT_LEAF_MAX          16u
#define RT_NODE_PSEUDO       1u          /* record only chains the chunks of an oversized leaf */

struct rt_dev_node {                     /* 64 bytes */
    float    lo0[3], hi0[3];             /* child 0 box */
    float    lo1[3], hi1[3];             /* child 1 box */
    uint32_t entry0, entry1;
    uint32_t flags, pad;
};

struct rt_dev_tri {                      /* 48 bytes */
    float a[3], e1[3], e2[3], n[3];
};

struct rt_dev_attr {                     /* 64 bytes */
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad;
};

struct rt_dev_mesh {                     /* one per distinct host Mesh */
    float    root_lo[3], root_hi[3];
    uint32_t root_entry;
    uint32_t tri_base;                   /* entries and object ids are mesh-local; add this for tris/attrs */
    uint32_t node_base;
    uint32_t stack_need;                 /* deepest level + 2 */
    uint32_t tame_boxes;                 /* 1: every box coordinate is 0 or has magnitude in [2^-40, 2^40] */
    uint32_t node_count;                 /* interior node records of this mesh */
    uint32_t pad[2];
};

struct rt_dev_instance {                 /* one per Scene_Mesh */
    float    pos[3];
    float    rx[3], ry[3], rz[3];        /* Transform.rotX/Y/Z */
    uint32_t mesh;
    int32_t  material;
};

struct rt_dev_light  { float c[3]; float radius; float color[3]; float albedo; };
struct rt_dev_sphere { float c[3]; float radius; int32_t material; };
struct rt_dev_plane  { float n[3]; float d; int32_t material; };

struct rt_dev_material {
    float    diffuse[3];
    float    mirror_color[3];
    float    ior;
    uint32_t kind;                       /* 0 diffuse, 1 mirror, 2 transparent (the reference's test order) */
    int32_t  texture;                    /* index into textures, -1 = none */
};

struct rt_dev_texture { uint32_t width, height, base, pad; };   /* base: texel index of row 0 */

struct rt_dev_camera {
    float pos[3], forward[3], right[3], up[3];
    float focal;                         /* height / (2 tan(fov/2)) */
};

struct rt_dev_scene {
    rt_dev_camera cam;
    int32_t light_count, sphere_count, plane_count, instance_count, material_count, texture_count, mesh_count;
    const rt_dev_light    *lights;
    const rt_dev_sphere   *spheres;
    const rt_dev_plane    *planes;
    const rt_dev_instance *instances;
    const rt_dev_mesh     *meshes;
    const rt_dev_material *materials;
    const rt_dev_texture  *textures;
    const rt_dev_node     *nodes;
    const rt_dev_tri      *tris;
    const rt_dev_attr     *attrs;
    const float           *texels;
    uint32_t stack_cap;                  /* entries of traversal stack each lane needs */
};

#endif /* RT_DEVICE_SCENE_H */
