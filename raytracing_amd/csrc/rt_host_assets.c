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
#include <unistd.h>

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
    free(mesh->root.childs);                         /* a tree built before (the pool starts at the root's first child) */
    mesh->tree_depth = 0;
    memset(&mesh->root, 0, sizeof(mesh->root));
    mesh->root.first_triangle_id = 0;
    mesh->root.last_triangle_id = mesh->face_count - 1;
    build_node(&mesh->root, mesh, 1, &pool);
    if (!mesh->root.childs) free(pool.nodes);       /* single leaf: nothing was taken from the pool */
    return 0;
}

/* ---- a second builder: surface-area heuristic over binned centroids ---------------------------
 *
 * Not the reference's tree (src/main.c:92-175 splits the longest axis at its midpoint and stops at 16
 * triangles): this one picks, per node, the axis and plane that minimise
 *     area(left box) * count(left) + area(right box) * count(right)
 * over 16 bins per axis, and stops at small leaves.  Same records (rt_mesh_node: box, adjacent child pair,
 * first/last triangle of the in-place permuted face table), so everything downstream is unchanged; rays
 * visit fewer boxes and far fewer triangles.  The picture differs from the reference tree's only where the
 * tree shape is observable (flat boxes that are never entered, which of two equal distances wins): an
 * opt-in for hosts that want speed over bit-for-bit agreement with the reference's builder. */
#define SAH_BINS 16
#define SAH_BALANCED_BELOW 40u      /* from this depth on nodes are cut at the median (bounds the depth) */

typedef struct sah_tri { float lo[3], hi[3], c[3]; } sah_tri;
typedef struct sah_ctx { rt_mesh *mesh; sah_tri *tri; node_pool *pool; uint32_t leaf_max; } sah_ctx;

static inline float half_area(const float lo[3], const float hi[3])
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

static void sah_swap(sah_ctx *cx, uint32_t a, uint32_t b)
{
    if (a == b) return;
    rt_triangle f = cx->mesh->faces[a]; cx->mesh->faces[a] = cx->mesh->faces[b]; cx->mesh->faces[b] = f;
    sah_tri t = cx->tri[a]; cx->tri[a] = cx->tri[b]; cx->tri[b] = t;
}

static void sah_node(sah_ctx *cx, rt_mesh_node *node, uint32_t depth);
typedef struct sah_job { sah_ctx *cx; rt_mesh_node *node; uint32_t depth; } sah_job;
static void *sah_thread(void *arg)
{
    sah_job *job = arg;
    sah_node(job->cx, job->node, job->depth);
    return NULL;
}

static void sah_node(sah_ctx *cx, rt_mesh_node *node, uint32_t depth)
{
    rt_mesh *mesh = cx->mesh;
    uint32_t seen = __atomic_load_n(&mesh->tree_depth, __ATOMIC_RELAXED);
    while (depth > seen && !__atomic_compare_exchange_n(&mesh->tree_depth, &seen, depth, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    const uint32_t first = node->first_triangle_id, last = node->last_triangle_id, count = last - first + 1;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    float clo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, chi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t i = first; i <= last; i++) {
        const sah_tri *t = cx->tri + i;
        for (int k = 0; k < 3; k++) {
            lo[k] = fsel_min(lo[k], t->lo[k]); hi[k] = fsel_max(hi[k], t->hi[k]);
            clo[k] = fsel_min(clo[k], t->c[k]); chi[k] = fsel_max(chi[k], t->c[k]);
        }
    }
    node->bbox.min.x = lo[0]; node->bbox.min.y = lo[1]; node->bbox.min.z = lo[2];
    node->bbox.max.x = hi[0]; node->bbox.max.y = hi[1]; node->bbox.max.z = hi[2];
    node->childs = NULL;
    if (count <= cx->leaf_max || depth >= TREE_DEPTH_LIMIT) return;

    uint32_t pivot = first;                          /* first triangle of the right child */
    int found = 0;
    if (depth < SAH_BALANCED_BELOW) {
        float best = FLT_MAX; int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; axis++) {
            const float extent = chi[axis] - clo[axis];
            if (!(extent > 0)) continue;
            const float scale = (float)SAH_BINS / extent;
            uint32_t n[SAH_BINS]; float blo[SAH_BINS][3], bhi[SAH_BINS][3];
            for (int b = 0; b < SAH_BINS; b++) {
                n[b] = 0;
                for (int k = 0; k < 3; k++) { blo[b][k] = FLT_MAX; bhi[b][k] = -FLT_MAX; }
            }
            for (uint32_t i = first; i <= last; i++) {
                const sah_tri *t = cx->tri + i;
                int b = (int)((t->c[axis] - clo[axis]) * scale);
                if (b < 0) b = 0;
                if (b >= SAH_BINS) b = SAH_BINS - 1;
                n[b]++;
                for (int k = 0; k < 3; k++) { blo[b][k] = fsel_min(blo[b][k], t->lo[k]); bhi[b][k] = fsel_max(bhi[b][k], t->hi[k]); }
            }
            float right_area[SAH_BINS]; uint32_t right_n[SAH_BINS];
            float rlo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, rhi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            uint32_t rn = 0;
            for (int b = SAH_BINS - 1; b > 0; b--) {
                rn += n[b];
                for (int k = 0; k < 3; k++) { rlo[k] = fsel_min(rlo[k], blo[b][k]); rhi[k] = fsel_max(rhi[k], bhi[b][k]); }
                right_n[b] = rn; right_area[b] = rn ? half_area(rlo, rhi) : 0.0f;
            }
            float llo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, lhi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            uint32_t ln = 0;
            for (int b = 0; b + 1 < SAH_BINS; b++) {         /* plane after bin b */
                ln += n[b];
                for (int k = 0; k < 3; k++) { llo[k] = fsel_min(llo[k], blo[b][k]); lhi[k] = fsel_max(lhi[k], bhi[b][k]); }
                if (!ln || !right_n[b + 1]) continue;
                const float cost = half_area(llo, lhi) * (float)ln + right_area[b + 1] * (float)right_n[b + 1];
                if (cost < best) { best = cost; best_axis = axis; best_bin = b; }
            }
        }
        /* a leaf of up to 16 triangles stays a leaf when no plane beats testing them all here */
        if (best_axis >= 0 && !(count <= LEAF_MAX_TRIANGLES && best >= half_area(lo, hi) * (float)count)) {
            const float scale = (float)SAH_BINS / (chi[best_axis] - clo[best_axis]);
            for (uint32_t i = first; i <= last; i++) {
                int b = (int)((cx->tri[i].c[best_axis] - clo[best_axis]) * scale);
                if (b < 0) b = 0;
                if (b >= SAH_BINS) b = SAH_BINS - 1;
                if (b <= best_bin) sah_swap(cx, pivot++, i);
            }
            found = pivot > first && pivot <= last;
        } else if (best_axis >= 0 || count <= LEAF_MAX_TRIANGLES) {
            if (count <= LEAF_MAX_TRIANGLES) return;
        }
    }
    if (!found) {
        /* coincident centroids, or deep in a lopsided branch: cut at the median of the widest centroid axis
         * (selection by repeated partition; equal keys are split by position) */
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        const uint32_t mid = first + count / 2;
        uint32_t a = first, b = last;
        while (a < b) {
            const float key = cx->tri[a + (b - a) / 2].c[axis];
            uint32_t i = a, j = b;
            while (i <= j) {
                while (cx->tri[i].c[axis] < key) i++;
                while (cx->tri[j].c[axis] > key) j--;
                if (i <= j) { sah_swap(cx, i, j); i++; if (j == 0) break; j--; }
            }
            if (mid <= j) b = j; else if (mid >= i) a = i; else break;
        }
        pivot = mid;
    }

    rt_mesh_node *pair = cx->pool->nodes + __atomic_fetch_add(&cx->pool->used, 2, __ATOMIC_RELAXED);
    memset(pair, 0, 2 * sizeof(*pair));
    node->childs = pair;
    pair[0].first_triangle_id = first;  pair[0].last_triangle_id = pivot - 1;
    pair[1].first_triangle_id = pivot;  pair[1].last_triangle_id = last;
    pthread_t helper;
    sah_job job = { cx, pair + 0, depth + 1 };
    int forked = depth <= FORK_LEVELS && count >= FORK_MIN_TRIANGLES && pthread_create(&helper, NULL, sah_thread, &job) == 0;
    if (!forked) sah_node(cx, pair + 0, depth + 1);
    sah_node(cx, pair + 1, depth + 1);
    if (forked) pthread_join(helper, NULL);
}

int rt_host_mesh_build_tree_sah(rt_mesh *mesh, uint32_t leaf_triangles)
{
    if (!mesh || !mesh->faces || mesh->face_count == 0) return -1;
    if (leaf_triangles == 0) leaf_triangles = 4;
    if (leaf_triangles > LEAF_MAX_TRIANGLES) leaf_triangles = LEAF_MAX_TRIANGLES;
    node_pool pool;
    pool.cap = 2 * (size_t)mesh->face_count + 2;
    pool.used = 0;
    pool.nodes = calloc(pool.cap, sizeof(rt_mesh_node));
    sah_tri *tri = malloc(sizeof(sah_tri) * (size_t)mesh->face_count);
    if (!pool.nodes || !tri) { free(pool.nodes); free(tri); return -2; }
    for (uint32_t i = 0; i < mesh->face_count; i++) {
        sah_tri *t = tri + i;
        for (int k = 0; k < 3; k++) { t->lo[k] = FLT_MAX; t->hi[k] = -FLT_MAX; }
        for (int c = 0; c < 3; c++) {
            const rt_vec3f *p = mesh->points + mesh->faces[i].ids[c].vertexId;
            for (int k = 0; k < 3; k++) { t->lo[k] = fsel_min(t->lo[k], axis_of(p, k)); t->hi[k] = fsel_max(t->hi[k], axis_of(p, k)); }
        }
        for (int k = 0; k < 3; k++) t->c[k] = 0.5f * t->lo[k] + 0.5f * t->hi[k];
    }
    free(mesh->root.childs);                         /* a tree built before (the pool starts at the root's first child) */
    mesh->tree_depth = 0;
    memset(&mesh->root, 0, sizeof(mesh->root));
    mesh->root.first_triangle_id = 0;
    mesh->root.last_triangle_id = mesh->face_count - 1;
    sah_ctx cx = { mesh, tri, &pool, leaf_triangles };
    sah_node(&cx, &mesh->root, 1);
    free(tri);
    if (!mesh->root.childs) free(pool.nodes);
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

/* ---- a prepared mesh as one file ---------------------------------------------------------------
 * What load_mesh (src/main.c:177-262) leaves in memory -- attribute arrays, the permuted face table,
 * the tree -- written out so that a later run maps it back without parsing or building anything.
 * Layout: a 64-byte header {magic, version, point, texcoord, normal, face, node counts, tree depth,
 * 64-bit key}, then points, texcoords, normals, faces, and the tree as rt_host_mesh_dump_tree writes
 * it (8 words per node, depth first).  The key is the caller's (a hash of the source model and of the
 * builder's parameters): a file with another key is refused. */
#include <stdio.h>
#define MESH_FILE_MAGIC 0x4D545231u   /* "1RTM" */
#define MESH_FILE_VERSION 1u
typedef struct mesh_file_header {
    uint32_t magic, version, points, texcoords, normals, faces, nodes, tree_depth;
    uint64_t key;
    uint32_t reserved[6];
} mesh_file_header;

int rt_host_mesh_save(const rt_mesh *mesh, uint64_t key, const char *path)
{
    if (!mesh || !path || !mesh->faces) return -1;
    uint32_t cen[4];
    rt_host_mesh_census(mesh, cen);
    uint32_t *tree = malloc((size_t)cen[0] * 32);
    if (!tree) return -2;
    rt_host_mesh_dump_tree(mesh, tree);
    mesh_file_header h;
    memset(&h, 0, sizeof(h));
    h.magic = MESH_FILE_MAGIC; h.version = MESH_FILE_VERSION;
    h.points = mesh->point_count; h.texcoords = mesh->texcoord_count; h.normals = mesh->normal_count;
    h.faces = mesh->face_count; h.nodes = cen[0]; h.tree_depth = mesh->tree_depth; h.key = key;
    /* written under a name of this process's own, then renamed over the target: processes started together that all
     * miss and all build never read one another's half-written file */
    size_t len = strlen(path);
    char *tmp = malloc(len + 32);
    if (!tmp) { free(tree); return -2; }
    snprintf(tmp, len + 32, "%s.%ld.tmp", path, (long)getpid());
    FILE *f = fopen(tmp, "wb");
    int ok = f != NULL;
    ok = ok && fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && (!h.points || fwrite(mesh->points, sizeof(rt_vec3f), h.points, f) == h.points);
    ok = ok && (!h.texcoords || fwrite(mesh->texcoords, sizeof(rt_vec2f), h.texcoords, f) == h.texcoords);
    ok = ok && (!h.normals || fwrite(mesh->normals, sizeof(rt_vec3f), h.normals, f) == h.normals);
    ok = ok && fwrite(mesh->faces, sizeof(rt_triangle), h.faces, f) == h.faces;
    ok = ok && fwrite(tree, 32, h.nodes, f) == h.nodes;
    if (f) ok = (fclose(f) == 0) && ok;
    free(tree);
    ok = ok && rename(tmp, path) == 0;
    if (!ok) remove(tmp);
    free(tmp);
    return ok ? 0 : -3;
}

/* node `*at` of the dump and its subtree back into linked records; 0 on a malformed dump */
static int restore_node(rt_mesh_node *node, const uint32_t *words, uint32_t count, uint32_t *at, node_pool *pool,
                        uint32_t faces, uint32_t depth, uint32_t *deepest)
{
    if (*at >= count || depth > TREE_DEPTH_LIMIT) return 0;
    const uint32_t *w = words + 8 * (size_t)(*at)++;
    memcpy(&node->bbox, w, 24);
    node->first_triangle_id = w[6];
    node->last_triangle_id = w[7] & 0x7FFFFFFFu;
    node->childs = NULL;
    if (node->first_triangle_id > node->last_triangle_id || node->last_triangle_id >= faces) return 0;
    if (depth > *deepest) *deepest = depth;
    if (!(w[7] & 0x80000000u)) return 1;
    if (pool->used + 2 > pool->cap) return 0;
    rt_mesh_node *pair = pool->nodes + pool->used;
    pool->used += 2;
    node->childs = pair;
    if (!restore_node(pair + 0, words, count, at, pool, faces, depth + 1, deepest)) return 0;
    if (!restore_node(pair + 1, words, count, at, pool, faces, depth + 1, deepest)) return 0;
    /* children split their parent's range */
    return pair[0].first_triangle_id == node->first_triangle_id && pair[1].last_triangle_id == node->last_triangle_id &&
           pair[0].last_triangle_id + 1 == pair[1].first_triangle_id;
}

int rt_host_mesh_load(const char *path, uint64_t key, rt_mesh *out)
{
    if (!path || !out) return -1;
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "rb");
    if (!f) return -3;
    mesh_file_header h;
    int rc = 0;
    uint32_t *tree = NULL;
    node_pool pool = { NULL, 0, 0 };
    if (fread(&h, sizeof(h), 1, f) != 1 || h.magic != MESH_FILE_MAGIC || h.version != MESH_FILE_VERSION) rc = -4;
    else if (h.key != key) rc = -7;
    else if (!h.faces || h.faces > 0x07FFFFFFu || !h.nodes || h.nodes > 2u * h.faces + 1u || !(h.nodes & 1u)) rc = -4;
    if (!rc) {
        out->point_count = h.points; out->texcoord_count = h.texcoords; out->normal_count = h.normals; out->face_count = h.faces;
        /* an attribute the file does not hold is one zeroed entry, and every id of it must then be 0 (checked below) */
        out->points = calloc(h.points ? h.points : 1, sizeof(rt_vec3f));
        out->texcoords = calloc(h.texcoords ? h.texcoords : 1, sizeof(rt_vec2f));
        out->normals = calloc(h.normals ? h.normals : 1, sizeof(rt_vec3f));
        out->faces = malloc(sizeof(rt_triangle) * (size_t)h.faces);
        tree = malloc((size_t)h.nodes * 32);
        pool.cap = h.nodes; pool.nodes = calloc(pool.cap, sizeof(rt_mesh_node));
        if (!out->points || !out->texcoords || !out->normals || !out->faces || !tree || !pool.nodes) rc = -2;
    }
    if (!rc) {
        int ok = (!h.points || fread(out->points, sizeof(rt_vec3f), h.points, f) == h.points) &&
                 (!h.texcoords || fread(out->texcoords, sizeof(rt_vec2f), h.texcoords, f) == h.texcoords) &&
                 (!h.normals || fread(out->normals, sizeof(rt_vec3f), h.normals, f) == h.normals) &&
                 fread(out->faces, sizeof(rt_triangle), h.faces, f) == h.faces &&
                 fread(tree, 32, h.nodes, f) == h.nodes;
        if (!ok) rc = -4;
    }
    if (!rc) {                                       /* indices in range: the file is input, not trusted */
        for (uint32_t i = 0; i < h.faces && !rc; i++)
            for (int c = 0; c < 3; c++) {
                const rt_vertex_ref *v = &out->faces[i].ids[c];
                if (v->vertexId >= h.points || v->texcoordId >= (h.texcoords ? h.texcoords : 1u) || v->normalId >= (h.normals ? h.normals : 1u)) rc = -5;
            }
    }
    if (!rc) {
        uint32_t at = 0, deepest = 0;
        if (!restore_node(&out->root, tree, h.nodes, &at, &pool, h.faces, 1, &deepest) || at != h.nodes ||
            out->root.first_triangle_id != 0 || out->root.last_triangle_id != h.faces - 1 || deepest != h.tree_depth) rc = -4;
        out->tree_depth = deepest;
    }
    fclose(f);
    free(tree);
    if (rc) {
        free(pool.nodes);
        out->root.childs = NULL;
        rt_host_mesh_free(out);
        return rc;
    }
    if (!out->root.childs) free(pool.nodes);
    return 0;
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
