/*
 * rt_host_assets.c -- host-side (CPU, plain C) preparation of what the GPU
 * renderer consumes: triangulated meshes with their bounding-box tree,
 * linearised textures, and the 8-bit output quantiser.
 *
 * These are the host stages on either side of the render loop.  The tree
 * shape is observable in rendered images (flat boxes are never entered, equal
 * hit distances are resolved by visiting order), so the builder reproduces
 * the reference's splitting rule and its in-place face permutation exactly
 * (reference: src/main.c:92-175 split_mesh_node, src/main.c:177-262 load_mesh,
 * src/main.c:45-63 load_texture, src/main.c:65-90 save_image).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host.h"

#define LEAF_MAX_TRIANGLES 16u      /* src/main.c:120 */
#define SPLIT_RETRY_LIMIT 200       /* the reference retries forever on coincident centroids */
#define TREE_DEPTH_LIMIT 4096u      /* the reference recurses without bound; no real model comes near (the dog: 24 levels) */
#define FORK_LEVELS 3u              /* the two subtrees of a node at depth <= 3 are built by two threads (8 at most) */
#define FORK_MIN_TRIANGLES 4096u    /* ... when the node is large enough to be worth a thread */

typedef struct node_pool { rt_mesh_node *nodes; size_t used, cap; } node_pool;

static inline float fsel_min(float a, float b) { return a < b ? a : b; }
static inline float fsel_max(float a, float b) { return a > b ? a : b; }
static inline float axis_of(const rt_vec3f *v, int axis) { return (&v->x)[axis]; }

/* bounds of the node's triangle range, then split on the midpoint of the
 * longest axis by centroid; src/main.c:92-175 */
static void build_node(rt_mesh_node *node, rt_mesh *mesh, uint32_t depth, node_pool *pool);

typedef struct subtree_job { rt_mesh_node *node; rt_mesh *mesh; uint32_t depth; node_pool *pool; } subtree_job;
static void *subtree_thread(void *arg)
{
    subtree_job *job = arg;
    build_node(job->node, job->mesh, job->depth, job->pool);
    return NULL;
}

/* The two subtrees of a node work on disjoint ranges of the face table, so they can be built side by side:
 * the same boxes, the same swaps in the same order within each range, hence the same tree and the same face
 * permutation as the sequential recursion (only the children's addresses in the pool differ). */
static void build_node(rt_mesh_node *node, rt_mesh *mesh, uint32_t depth, node_pool *pool)
{
    uint32_t seen = __atomic_load_n(&mesh->tree_depth, __ATOMIC_RELAXED);
    while (depth > seen && !__atomic_compare_exchange_n(&mesh->tree_depth, &seen, depth, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}

    rt_bbox box = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
    for (uint32_t i = node->first_triangle_id; i <= node->last_triangle_id; i++) {
        const rt_triangle *f = mesh->faces + i;
        for (int k = 0; k < 3; k++) {
            const rt_vec3f *p = mesh->points + f->ids[k].vertexId;
            box.min.x = fsel_min(box.min.x, p->x); box.min.y = fsel_min(box.min.y, p->y); box.min.z = fsel_min(box.min.z, p->z);
            box.max.x = fsel_max(box.max.x, p->x); box.max.y = fsel_max(box.max.y, p->y); box.max.z = fsel_max(box.max.z, p->z);
        }
    }
    node->bbox = box;
    node->childs = NULL;

    uint32_t first = node->first_triangle_id, last = node->last_triangle_id;
    if (last - first + 1 <= LEAF_MAX_TRIANGLES) return;
    if (depth >= TREE_DEPTH_LIMIT) return;          /* degenerate geometry (one triangle split off per level): a large leaf ends the recursion */

    int axis = 0;
    if (axis_of(&box.max, 1) - axis_of(&box.min, 1) > axis_of(&box.max, axis) - axis_of(&box.min, axis)) axis = 1;
    if (axis_of(&box.max, 2) - axis_of(&box.min, 2) > axis_of(&box.max, axis) - axis_of(&box.min, axis)) axis = 2;

    float middle = (axis_of(&box.max, axis) + axis_of(&box.min, axis)) / 2;
    uint32_t pivot = first;
    int tries = 0;
    for (;;) {
        for (uint32_t i = first; i <= last; i++) {
            const rt_triangle *f = mesh->faces + i;
            float centre = 0;
            for (int k = 0; k < 3; k++) centre += axis_of(mesh->points + f->ids[k].vertexId, axis) / 3;
            if (centre < middle) {
                rt_triangle tmp = mesh->faces[pivot];
                mesh->faces[pivot] = mesh->faces[i];
                mesh->faces[i] = tmp;
                pivot++;
            }
        }
        if (pivot == first) {
            middle = (middle + axis_of(&box.max, axis)) / 2;
        } else if (pivot == last + 1) {
            middle = (axis_of(&box.min, axis) + middle) / 2;
            pivot = first;
        } else {
            break;
        }
        if (++tries > SPLIT_RETRY_LIMIT) return;   /* unsplittable: keep it a (large) leaf */
    }

    rt_mesh_node *pair = pool->nodes + __atomic_fetch_add(&pool->used, 2, __ATOMIC_RELAXED);
    memset(pair, 0, 2 * sizeof(*pair));
    node->childs = pair;
    pair[0].first_triangle_id = first;  pair[0].last_triangle_id = pivot - 1;
    pair[1].first_triangle_id = pivot;  pair[1].last_triangle_id = last;
    pthread_t helper;
    subtree_job job = { pair + 0, mesh, depth + 1, pool };
    int forked = depth <= FORK_LEVELS && last - first + 1 >= FORK_MIN_TRIANGLES &&
                 pthread_create(&helper, NULL, subtree_thread, &job) == 0;
    if (!forked) build_node(pair + 0, mesh, depth + 1, pool);
    build_node(pair + 1, mesh, depth + 1, pool);
    if (forked) pthread_join(helper, NULL);
}

int rt_host_mesh_build_tree(rt_mesh *mesh)
{
    if (!mesh || !mesh->faces || mesh->face_count == 0) return -1;
    node_pool pool;
    pool.cap = 2 * (size_t)mesh->face_count + 2;
    pool.used = 0;
    pool.nodes = calloc(pool.cap, sizeof(rt_mesh_node));
    if (!pool.nodes) return -2;
    mesh->tree_depth = 0;
    memset(&mesh->root, 0, sizeof(mesh->root));
    mesh->root.first_triangle_id = 0;
    mesh->root.last_triangle_id = mesh->face_count - 1;
    build_node(&mesh->root, mesh, 1, &pool);
    if (!mesh->root.childs) free(pool.nodes);       /* single leaf: nothing was taken from the pool */
    return 0;
}

/* OBJ arrays -> rt_mesh; src/main.c:183-258.  Index 0 of each attribute array
 * is the parser's dummy entry and indices are used 1-based as they come.
 * Coordinates are permuted (x,y,z) -> (y,z,x) by the two swaps y<->z, x<->z. */
int rt_host_mesh_from_obj_arrays(const float *positions, uint32_t position_count,
                                 const float *texcoords, uint32_t texcoord_count,
                                 const float *normals, uint32_t normal_count,
                                 const uint32_t *indices_ptn, uint64_t index_count,
                                 const uint32_t *face_vertices, uint32_t face_count, rt_mesh *out)
{
    if (!positions || !indices_ptn || !face_vertices || !out) return -1;
    memset(out, 0, sizeof(*out));

    uint64_t tri_total = 0, corners = 0;
    for (uint32_t i = 0; i < face_count; i++) {
        if (face_vertices[i] < 3) return -3;
        tri_total += face_vertices[i] - 2;
        corners += face_vertices[i];
    }
    if (tri_total == 0 || tri_total > 0x07FFFFFFu) return -4;
    if (corners > index_count) return -6;          /* the faces name more corners than the index table holds */

    out->point_count = position_count;
    out->texcoord_count = texcoord_count;
    out->normal_count = normal_count;
    out->points = malloc(sizeof(rt_vec3f) * (position_count ? position_count : 1));
    out->texcoords = malloc(sizeof(rt_vec2f) * (texcoord_count ? texcoord_count : 1));
    out->normals = malloc(sizeof(rt_vec3f) * (normal_count ? normal_count : 1));
    out->faces = malloc(sizeof(rt_triangle) * tri_total);
    if (!out->points || !out->texcoords || !out->normals || !out->faces) { rt_host_mesh_free(out); return -2; }
    memcpy(out->points, positions, sizeof(rt_vec3f) * position_count);
    if (texcoord_count) memcpy(out->texcoords, texcoords, sizeof(rt_vec2f) * texcoord_count);
    if (normal_count) memcpy(out->normals, normals, sizeof(rt_vec3f) * normal_count);

    for (uint32_t i = 0; i < position_count; i++) {
        rt_vec3f *p = out->points + i;
        float t = p->y; p->y = p->z; p->z = t;
        t = p->x; p->x = p->z; p->z = t;
    }
    for (uint32_t i = 0; i < normal_count; i++) {
        rt_vec3f *p = out->normals + i;
        float t = p->y; p->y = p->z; p->z = t;
        t = p->x; p->x = p->z; p->z = t;
    }

    /* a quad (0,1,2,3) becomes (0,1,2) and (0,2,3); larger polygons fan the same way */
    uint32_t tri = 0;
    size_t cursor = 0;
    for (uint32_t i = 0; i < face_count; i++) {
        const uint32_t *corner = indices_ptn + 3 * cursor;
        for (uint32_t k = 1; k + 1 < face_vertices[i]; k++) {
            const uint32_t pick[3] = { 0, k, k + 1 };
            for (int c = 0; c < 3; c++) {
                const uint32_t *src = corner + 3 * pick[c];
                if (src[0] >= position_count || (texcoord_count && src[1] >= texcoord_count) ||
                    (normal_count && src[2] >= normal_count)) { rt_host_mesh_free(out); return -5; }
                out->faces[tri].ids[c].vertexId = src[0];
                out->faces[tri].ids[c].texcoordId = src[1];
                out->faces[tri].ids[c].normalId = src[2];
            }
            tri++;
        }
        cursor += face_vertices[i];
    }
    out->face_count = tri;
    int rc = rt_host_mesh_build_tree(out);
    if (rc) { rt_host_mesh_free(out); return rc; }
    return 0;
}

void rt_host_mesh_free(rt_mesh *mesh)
{
    if (!mesh) return;
    free(mesh->points); free(mesh->texcoords); free(mesh->normals); free(mesh->faces);
    free(mesh->root.childs);      /* the node pool starts at the root's first child */
    memset(mesh, 0, sizeof(*mesh));
}

static void census(const rt_mesh_node *n, uint32_t depth, uint32_t acc[4])
{
    acc[0]++;
    if (depth > acc[2]) acc[2] = depth;
    if (!n->childs) {
        uint32_t count = n->last_triangle_id - n->first_triangle_id + 1;
        acc[1]++;
        if (count > acc[3]) acc[3] = count;
        return;
    }
    census(n->childs + 0, depth + 1, acc);
    census(n->childs + 1, depth + 1, acc);
}

void rt_host_mesh_census(const rt_mesh *mesh, uint32_t out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    census(&mesh->root, 1, out);
}

/* depth-first dump of the tree, 8 floats per node: bbox (6), first, last
 * triangle id (as float bit patterns of the integers); children follow their
 * parent, child 0 first.  Lets two trees be compared node for node. */
static void dump(const rt_mesh_node *n, uint32_t *words, size_t *at)
{
    uint32_t *w = words + 8 * (*at)++;
    memcpy(w, &n->bbox, 24);
    w[6] = n->first_triangle_id;
    w[7] = n->childs ? (n->last_triangle_id | 0x80000000u) : n->last_triangle_id;
    if (n->childs) { dump(n->childs + 0, words, at); dump(n->childs + 1, words, at); }
}
size_t rt_host_mesh_dump_tree(const rt_mesh *mesh, uint32_t *words8_per_node)
{
    size_t at = 0;
    dump(&mesh->root, words8_per_node, &at);
    return at;
}

/* 8-bit RGB rows (top row first) -> linear float image stored bottom row
 * first; src/main.c:52-59: powf(byte / 255.0f, 2.2f) per channel. */
int rt_host_texture_from_rgb8(const uint8_t *rgb, uint32_t width, uint32_t height, rt_image3f *out)
{
    if (!rgb || !out || !width || !height) return -1;
    rt_vec3f *px = malloc(sizeof(rt_vec3f) * (size_t)width * height);
    if (!px) return -2;
    float lut[256];
    for (int v = 0; v < 256; v++) lut[v] = powf((float)v / 255.0f, 2.2f);
    for (uint32_t j = 0; j < height; j++)
        for (uint32_t i = 0; i < width; i++) {
            const uint8_t *s = rgb + 3 * ((size_t)width * j + i);
            rt_vec3f *d = px + (size_t)width * (height - 1 - j) + i;
            d->x = lut[s[0]]; d->y = lut[s[1]]; d->z = lut[s[2]];
        }
    out->size.x = width; out->size.y = height; out->pixels = px;
    return 0;
}

void rt_host_image_free(rt_image3f *image)
{
    if (!image) return;
    free(image->pixels);
    memset(image, 0, sizeof(*image));
}

/* linear float image -> 8-bit RGB; src/main.c:73-82: min(c,1) (a NaN becomes
 * 1), powf(c, 1/2.2f), times 255, truncation. */
void rt_host_image_quantize(const rt_image3f *image, uint8_t *rgb)
{
    size_t n = (size_t)image->size.x * image->size.y;
    const float *p = &image->pixels[0].x;
    for (size_t i = 0; i < 3 * n; i++) {
        float c = p[i] < 1.0f ? p[i] : 1.0f;
        c = powf(c, 1.0f / 2.2f);
        c = c * 255.0f;
        rgb[i] = (uint8_t)c;
    }
}
