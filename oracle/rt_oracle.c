/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY.  A scalar CPU restatement of the
 * reference's per-pixel path-tracing loop, used by tests/, the smoke check and
 * the cpu_baseline leg of bench.py to CHECK the GPU renderer.  The product
 * never links, loads or calls anything in this directory.
 *
 * What it restates (reference file:line given at each function):
 *   raytrace_region  src/main.c:742-792     -> orc_render_region_*
 *   raytrace_ray     src/main.c:611-740     -> orc_trace_path
 *   find_ray_hit     src/main.c:264-538     -> orc_closest_hit
 *   compute_light_contribution  src/main.c:546-609 -> orc_direct_light
 *   sample_texture   src/math.c:269-276     -> orc_texel
 *   vector helpers   src/math.c:3-146,278-311
 *   Random_Serie     src/math.c:158-235     -> orc_aes_* (AES-DEC in plain C)
 *
 * Every float operation below is meant to be ONE IEEE-754 binary32 rounding in
 * the order the reference's source performs it (doubles appear exactly where
 * the reference's C promotes to double).  Build with -ffp-contract=off
 * (oracle/Makefile does) and the result is bit-identical to the reference
 * compiled from its own sources with the same flag: tests/test_oracle_vs_reference_live.py
 * checks that on whole float buffers, and tests/golden/ holds the vectors for
 * machines that do not have the reference.
 *
 * Two switches exist because a GPU can reproduce neither AES-NI nor glibc:
 *   rng:   ORC_RNG_AES      the reference's own serial 256-bit AES-DEC stream
 *          ORC_RNG_COUNTER  Threefry-2x32-20 keyed by (seed, pixel, sample), the
 *                           generator the GPU kernels use for parallel sampling
 *                           (restated here independently of the device code)
 *   math:  ORC_MATH_LIBM     cosf/sinf/logf from the C library (as the reference)
 *          ORC_MATH_PORTABLE rt_math_portable.h (shared with the GPU kernels)
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rt_scene.h"
#include "rt_math_portable.h"

#define ORC_EXPORT __attribute__((visibility("default")))

enum { ORC_RNG_AES = 0, ORC_RNG_COUNTER = 1 };
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };

/* Work counters (same meaning as the GPU library's rt_gpu_stats). */
typedef struct orc_stats {
    uint64_t samples;        /* camera paths started */
    uint64_t rays;           /* closest-hit queries: camera + bounce + shadow */
    uint64_t shadow_rays;    /* of which shadow rays */
    uint64_t root_tests;     /* mesh instances whose root box was tested */
    uint64_t interior_pops;  /* interior BVH nodes popped (two child boxes each) */
    uint64_t leaf_pops;      /* leaves popped */
    uint64_t tri_tests;      /* triangle intersection tests */
    uint64_t tri_hits;       /* queries whose final hit is a triangle */
    uint64_t tex_lookups;    /* texture fetches */
    uint64_t rng_draws;      /* generator steps */
} orc_stats;

/* ------------------------------------------------------------------------ */
/* vector helpers: src/math.c:3-146, each a literal transcription of the    */
/* operation order (dot starts from +0 and adds x, y, z products in turn).  */
/* ------------------------------------------------------------------------ */
typedef rt_vec3f v3;

static inline v3 v3_make(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline float v3_dot(v3 a, v3 b)
{
    float acc = 0.0f;
    acc += a.x * b.x;
    acc += a.y * b.y;
    acc += a.z * b.z;
    return acc;
}
static inline float v3_len2(v3 a) { return v3_dot(a, a); }
static inline float v3_len(v3 a) { return sqrtf(v3_len2(a)); }
static inline v3 v3_unit(v3 a)            /* three divisions by the norm, src/math.c:29-40 */
{
    float n = v3_len(a);
    return v3_make(a.x / n, a.y / n, a.z / n);
}
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_div(v3 a, v3 b) { return v3_make(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* the reference's MIN / MAX are (a)<(b)?(a):(b) and (a)>(b)?(a):(b), src/math.h:8-9:
 * with a NaN operand they return the SECOND operand */
static inline float sel_min(float a, float b) { return a < b ? a : b; }
static inline float sel_max(float a, float b) { return a > b ? a : b; }
/* a*w.x + b*w.y + c*w.z per component, src/math.c:278-287 */
static inline v3 v3_blend(v3 a, v3 b, v3 c, v3 w)
{
    return v3_make(a.x * w.x + b.x * w.y + c.x * w.z,
                   a.y * w.x + b.y * w.y + c.y * w.z,
                   a.z * w.x + b.z * w.y + c.z * w.z);
}

/* ------------------------------------------------------------------------ */
/* RNG                                                                      */
/* ------------------------------------------------------------------------ */

/* AES inverse S-box and GF(2^8) helpers, built once (FIPS-197 section 5.3). */
static uint8_t g_inv_sbox[256];
static int g_aes_ready = 0;

static uint8_t gf_mul(uint8_t a, uint8_t b)
{
    uint8_t p = 0;
    for (int i = 0; i < 8; i++) {
        if (b & 1) p ^= a;
        uint8_t hi = a & 0x80;
        a <<= 1;
        if (hi) a ^= 0x1b;
        b >>= 1;
    }
    return p;
}

static void aes_init(void)
{
    if (g_aes_ready) return;
    for (int x = 0; x < 256; x++) {
        /* multiplicative inverse by brute force, then the affine map */
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv;
        uint8_t r = inv;
        for (int k = 0; k < 4; k++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        s ^= 0x63;
        g_inv_sbox[s] = (uint8_t)x;
    }
    __sync_synchronize();
    g_aes_ready = 1;
}

/* One AESDEC step as the instruction defines it: InvShiftRows, InvSubBytes,
 * InvMixColumns, then xor with the round key.  Bytes are column-major:
 * byte i sits in row i%4 of column i/4. */
static void aes_dec_round(uint8_t state[16], const uint8_t key[16])
{
    uint8_t t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            t[r + 4 * c] = g_inv_sbox[state[r + 4 * ((c - r) & 3)]];
    for (int c = 0; c < 4; c++) {
        const uint8_t *a = t + 4 * c;
        state[4 * c + 0] = gf_mul(a[0], 14) ^ gf_mul(a[1], 11) ^ gf_mul(a[2], 13) ^ gf_mul(a[3], 9);
        state[4 * c + 1] = gf_mul(a[0], 9) ^ gf_mul(a[1], 14) ^ gf_mul(a[2], 11) ^ gf_mul(a[3], 13);
        state[4 * c + 2] = gf_mul(a[0], 13) ^ gf_mul(a[1], 9) ^ gf_mul(a[2], 14) ^ gf_mul(a[3], 11);
        state[4 * c + 3] = gf_mul(a[0], 11) ^ gf_mul(a[1], 13) ^ gf_mul(a[2], 9) ^ gf_mul(a[3], 14);
    }
    for (int i = 0; i < 16; i++) state[i] ^= key[i];
}

/* Threefry-2x32, 20 rounds (Salmon et al., "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11; rotation constants of the Random123 library). */
static inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
static void threefry2x32_20(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2])
{
    static const int rot[8] = { 13, 15, 26, 6, 17, 29, 16, 24 };
    uint32_t ks[3] = { key[0], key[1], 0x1BD11BDAu ^ key[0] ^ key[1] };
    uint32_t x0 = ctr[0] + ks[0], x1 = ctr[1] + ks[1];
    for (int r = 0; r < 20; r++) {
        x0 += x1;
        x1 = rotl32(x1, rot[r & 7]);
        x1 ^= x0;
        if ((r & 3) == 3) {
            int s = (r >> 2) + 1;
            x0 += ks[s % 3];
            x1 += ks[(s + 1) % 3] + (uint32_t)s;
        }
    }
    out[0] = x0; out[1] = x1;
}

typedef struct orc_ctx {
    int rng_mode, math_mode;
    /* AES mode: the job's 256-bit state (src/math.h:11-18) */
    uint8_t aes[32];
    /* counter mode: per-sample key and the draw counter */
    uint32_t seed[2], sample_key[2], draw;
    orc_stats *stats;
} orc_ctx;

/* counter mode: a sample's key is K = Threefry(seed; pixel, sample); draw 0 of
 * the sample is K itself, draw d >= 1 is Threefry(K; d, 0). */
static void ctx_begin_sample(orc_ctx *cx, uint32_t pixel, uint32_t sample)
{
    uint32_t ctr[2] = { pixel, sample };
    threefry2x32_20(cx->seed, ctr, cx->sample_key);
    cx->draw = 0;
}


/* One generator step -> two uniforms in [lo, hi).  src/math.c:158-192 and
 * 206-215: 23 random mantissa bits under exponent 0 give v in [1,2); the
 * result is the FUSED v*(hi-lo) + (2*lo-hi).  The render loop only ever
 * consumes lanes 0 and 1 of the four 32-bit lanes a step produces. */
static void ctx_uniform4(orc_ctx *cx, float lo, float hi, float out[4])
{
    uint32_t w[4];
    if (cx->stats) cx->stats->rng_draws++;
    if (cx->rng_mode == ORC_RNG_AES) {
        /* src/math.c:158-162: x0 = aesdec(x0, x1); x1 = aesdec(x0_new, x1) */
        uint8_t next1[16];
        aes_dec_round(cx->aes, cx->aes + 16);
        memcpy(next1, cx->aes, 16);
        aes_dec_round(next1, cx->aes + 16);
        memcpy(cx->aes + 16, next1, 16);
        for (int i = 0; i < 4; i++) {
            uint32_t a, b;
            memcpy(&a, cx->aes + 4 * i, 4);
            memcpy(&b, cx->aes + 16 + 4 * i, 4);
            w[i] = a ^ b;
        }
    } else {
        if (cx->draw == 0) { w[0] = cx->sample_key[0]; w[1] = cx->sample_key[1]; }
        else { uint32_t ctr[2] = { cx->draw, 0 }; threefry2x32_20(cx->sample_key, ctr, w); }
        cx->draw++;
        w[2] = w[3] = 0;     /* lanes 2 and 3 are never consumed */
    }
    const float scale = hi - lo;
    const float step = -hi + 2 * lo;
    for (int i = 0; i < 4; i++) {
        uint32_t bits = (w[i] & 0x007FFFFFu) | 0x3F800000u;
        float v;
        memcpy(&v, &bits, 4);
        out[i] = fmaf(v, scale, step);
    }
}

static inline void ctx_sincos(const orc_ctx *cx, float a, float *s, float *c)
{
    if (cx->math_mode == ORC_MATH_LIBM) { *s = sinf(a); *c = cosf(a); }
    else rt_sincosf(a, s, c);
}
static inline float ctx_log(const orc_ctx *cx, float a)
{
    return cx->math_mode == ORC_MATH_LIBM ? logf(a) : rt_logf(a);
}

/* Box-Muller pair, src/math.c:225-235: the angle 2*pi*u0 is formed in double
 * and narrowed to float at the call. */
static void ctx_gaussian2(orc_ctx *cx, float mean, float sigma, float out[2])
{
    float u[4];
    ctx_uniform4(cx, 0, 1, u);
    float angle = (float)(2.0f * M_PI * u[0]);
    float s, c;
    ctx_sincos(cx, angle, &s, &c);
    float radius = sqrtf(-2 * ctx_log(cx, u[1]));
    out[0] = mean + sigma * c * radius;
    out[1] = mean + sigma * s * radius;
}

/* ------------------------------------------------------------------------ */
/* closest hit: src/main.c:264-538                                          */
/* ------------------------------------------------------------------------ */

/* slab test of one box (division, then the reference's MIN/MAX selects),
 * src/main.c:379-386 and 404-411 */
static inline void slab(const rt_bbox *b, v3 o, v3 d, float *enter, float *leave)
{
    v3 t0 = v3_div(v3_sub(b->min, o), d);
    v3 t1 = v3_div(v3_sub(b->max, o), d);
    v3 lo = v3_make(sel_min(t0.x, t1.x), sel_min(t0.y, t1.y), sel_min(t0.z, t1.z));
    v3 hi = v3_make(sel_max(t0.x, t1.x), sel_max(t0.y, t1.y), sel_max(t0.z, t1.z));
    *enter = sel_max(sel_max(lo.x, lo.y), lo.z);
    *leave = sel_min(sel_min(hi.x, hi.y), hi.z);
}

/* sphere-shaped primitive (lights and spheres share it), src/main.c:283-313:
 * returns 1 and the accepted t when the primitive becomes the closest hit */
static inline int ball_hit(v3 center, float radius, const rt_ray *ray, float best, float *t_out, int *entered)
{
    v3 oc = v3_sub(ray->origin, center);
    float b = v3_dot(ray->direction, oc);
    float c = v3_len2(oc) - radius * radius;
    float det = b * b - c;
    if (det < 0.0f) return 0;
    float root = sqrtf(det);
    float t = -b - root;
    if ((double)t < 1e-3) t = -b + root;
    else *entered = 1;
    if (t > best) return 0;
    if ((double)t < 1e-3) return 0;
    *t_out = t;
    return 1;
}

static void orc_closest_hit(const rt_scene *scene, const rt_ray *ray, float max_distance,
                            rt_ray_result *res, orc_stats *st, int is_shadow)
{
    memset(res, 0, sizeof(*res));
    res->t_min = max_distance;
    res->material_id = -1;
    if (st) { st->rays++; if (is_shadow) st->shadow_rays++; }

    for (int i = 0; i < scene->light_count; i++) {            /* :280-314 */
        const rt_light *l = scene->lights + i;
        float t; int entered = 0;
        int ok = ball_hit(l->center, l->radius, ray, res->t_min, &t, &entered);
        if (entered) res->enter_shape = true;
        if (!ok) continue;
        res->hit = true; res->object_type = RT_LIGHT; res->object_id = (uint32_t)i; res->t_min = t;
    }
    for (int i = 0; i < scene->sphere_count; i++) {           /* :317-352 */
        const rt_sphere *s = scene->spheres + i;
        float t; int entered = 0;
        int ok = ball_hit(s->center, s->radius, ray, res->t_min, &t, &entered);
        if (entered) res->enter_shape = true;
        if (!ok) continue;
        res->hit = true; res->object_type = RT_SPHERE; res->object_id = (uint32_t)i; res->t_min = t;
        res->material_id = s->material_id;
    }
    for (int i = 0; i < scene->plane_count; i++) {            /* :355-369 */
        const rt_plane *p = scene->planes + i;
        float t = (p->d - v3_dot(ray->origin, p->normal)) / v3_dot(ray->direction, p->normal);
        if ((double)t < 1e-3) continue;
        if (t > res->t_min) continue;
        res->hit = true; res->object_type = RT_PLANE; res->object_id = (uint32_t)i; res->t_min = t;
        res->material_id = p->material_id;
    }

    for (int i = 0; i < scene->mesh_count; i++) {             /* :372-485 */
        const rt_scene_mesh *inst = scene->meshes + i;
        const rt_mesh *mesh = inst->mesh;
        const rt_transform *tr = &inst->transform;
        /* ray into mesh space, src/math.c:299-311: component k of the result is
         * rotK.x*v.x + rotK.y*v.y + rotK.z*v.z (left-to-right sums, no leading zero) */
        v3 rel = v3_sub(ray->origin, tr->position);
        v3 o = v3_make(tr->rotX.x * rel.x + tr->rotX.y * rel.y + tr->rotX.z * rel.z,
                       tr->rotY.x * rel.x + tr->rotY.y * rel.y + tr->rotY.z * rel.z,
                       tr->rotZ.x * rel.x + tr->rotZ.y * rel.y + tr->rotZ.z * rel.z);
        v3 wd = ray->direction;
        v3 u = v3_make(tr->rotX.x * wd.x + tr->rotX.y * wd.y + tr->rotX.z * wd.z,
                       tr->rotY.x * wd.x + tr->rotY.y * wd.y + tr->rotY.z * wd.z,
                       tr->rotZ.x * wd.x + tr->rotZ.y * wd.y + tr->rotZ.z * wd.z);
        if (st) st->root_tests++;
        {
            float enter, leave;
            slab(&mesh->root.bbox, o, u, &enter, &leave);
            if (leave < enter) continue;                      /* equal passes */
            if (enter > res->t_min) continue;
        }
        const rt_mesh_node **stack = malloc(sizeof(*stack) * (mesh->tree_depth + 2));
        int top = 0;
        stack[0] = &mesh->root;
        while (top >= 0) {
            const rt_mesh_node *node = stack[top--];
            if (node->childs) {                               /* :402-447 */
                if (st) st->interior_pops++;
                float e0, l0, e1, l1;
                slab(&node->childs[0].bbox, o, u, &e0, &l0);
                slab(&node->childs[1].bbox, o, u, &e1, &l1);
                int h0 = (e0 < l0) && (e0 < res->t_min);      /* strict: flat boxes never enter */
                int h1 = (e1 < l1) && (e1 < res->t_min);
                if (h0 && h1) {
                    if (e0 > e1) { stack[++top] = node->childs + 0; stack[++top] = node->childs + 1; }
                    else         { stack[++top] = node->childs + 1; stack[++top] = node->childs + 0; }
                } else if (h0) stack[++top] = node->childs + 0;
                else if (h1) stack[++top] = node->childs + 1;
            } else {                                          /* :449-482 */
                if (st) st->leaf_pops++;
                for (uint32_t j = node->first_triangle_id; j <= node->last_triangle_id; j++) {
                    if ((int)j < 0) break;                    /* the reference's index is an int */
                    if (st) st->tri_tests++;
                    const rt_triangle *f = mesh->faces + j;
                    v3 a = mesh->points[f->ids[0].vertexId];
                    v3 b = mesh->points[f->ids[1].vertexId];
                    v3 c = mesh->points[f->ids[2].vertexId];
                    v3 e1v = v3_sub(b, a), e2v = v3_sub(c, a);
                    v3 oa = v3_sub(o, a);
                    v3 oau = v3_cross(oa, u);
                    v3 n = v3_cross(e1v, e2v);
                    float t = -v3_dot(oa, n) / v3_dot(u, n);
                    if ((double)t < 1e-3) continue;
                    if (t > res->t_min) continue;
                    float beta = -v3_dot(e2v, oau) / v3_dot(u, n);
                    float gamma = v3_dot(e1v, oau) / v3_dot(u, n);
                    if ((beta < 0) || (beta > 1)) continue;
                    if ((gamma < 0) || (gamma > 1)) continue;
                    if (beta + gamma > 1) continue;
                    res->hit = true; res->t_min = t; res->object_type = RT_TRIANGLE;
                    res->object_id = j; res->mesh_id = (uint32_t)i;
                    res->triangle_beta = beta; res->triangle_gamma = gamma;
                    res->material_id = (int32_t)inst->material_id;
                }
            }
        }
        free((void *)stack);
    }

    if (!res->hit) return;
    switch (res->object_type) {                               /* :487-535 */
    case RT_SPHERE: {
        const rt_sphere *s = scene->spheres + res->object_id;
        res->intersection_point = v3_add(ray->origin, v3_scale(ray->direction, res->t_min));
        res->surface_normal = v3_unit(v3_sub(res->intersection_point, s->center));
    } break;
    case RT_PLANE:
        res->intersection_point = v3_add(ray->origin, v3_scale(ray->direction, res->t_min));
        res->surface_normal = scene->planes[res->object_id].normal;
        break;
    case RT_TRIANGLE: {
        const rt_scene_mesh *inst = scene->meshes + res->mesh_id;
        const rt_mesh *mesh = inst->mesh;
        const rt_triangle *f = mesh->faces + res->object_id;
        if (st) st->tri_hits++;
        res->triangle_alpha = 1 - res->triangle_beta - res->triangle_gamma;
        v3 w = v3_make(res->triangle_alpha, res->triangle_beta, res->triangle_gamma);
        v3 n = v3_blend(mesh->normals[f->ids[0].normalId], mesh->normals[f->ids[1].normalId],
                        mesh->normals[f->ids[2].normalId], w);      /* not renormalised */
        rt_vec2f t0 = mesh->texcoords[f->ids[0].texcoordId];
        rt_vec2f t1 = mesh->texcoords[f->ids[1].texcoordId];
        rt_vec2f t2 = mesh->texcoords[f->ids[2].texcoordId];
        res->triangle_uv.x = t0.x * w.x + t1.x * w.y + t2.x * w.z;
        res->triangle_uv.y = t0.y * w.x + t1.y * w.y + t2.y * w.z;
        res->intersection_point = v3_add(ray->origin, v3_scale(ray->direction, res->t_min));
        res->surface_normal = v3_blend(inst->transform.rotX, inst->transform.rotY, inst->transform.rotZ, n);
    } break;
    default: break;                                           /* light: nothing */
    }
}

/* tangent of a frame around n, src/main.c:561-576 and 702-717 (the first two
 * branches are the same expression) */
static v3 frame_tangent(v3 n)
{
    v3 t = v3_make(0, 0, 0);
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax >= ay && ax >= az) { t.x = n.y; t.y = -n.x; t.z = 0; }
    else if (ay >= ax && ay >= az) { t.x = n.y; t.y = -n.x; t.z = 0; }
    else if (az >= ax && az >= ay) { t.x = 0; t.y = -n.z; t.z = n.y; }
    return v3_unit(t);
}

/* next-event estimation over every light: src/main.c:546-609.
 * out = {r, g, b, number of unoccluded samples} */
static void orc_direct_light(orc_ctx *cx, const rt_scene *scene, const rt_ray_result *at, float out[4])
{
    v3 sum = v3_make(0, 0, 0);
    float reached = 0;
    for (int32_t i = 0; i < scene->light_count; i++) {
        const rt_light *l = scene->lights + i;
        float r[4];
        ctx_uniform4(cx, 0, 1, r);                            /* drawn before any early-out */
        float s, c;
        ctx_sincos(cx, (float)(2 * M_PI * r[0]), &s, &c);
        float ring = sqrtf(2 * r[1] - r[1] * r[1]);
        v3 h = v3_make(c * ring, s * ring, 1 - r[1]);         /* uniform on a hemisphere */

        v3 axis = v3_unit(v3_sub(l->center, at->intersection_point));
        v3 t0 = frame_tangent(axis);
        v3 t1 = v3_cross(axis, t0);
        v3 q = v3_add(v3_scale(axis, -h.z), v3_add(v3_scale(t0, -h.y), v3_scale(t1, -h.x)));
        q = v3_scale(v3_unit(q), l->radius);
        v3 target = v3_add(q, l->center);
        v3 path = v3_sub(target, at->intersection_point);
        v3 dir = v3_unit(path);
        float cosine = v3_dot(dir, at->surface_normal);
        if (cosine < 0) continue;

        rt_ray shadow; shadow.origin = at->intersection_point; shadow.direction = dir;
        rt_ray_result blocker;
        orc_closest_hit(scene, &shadow, v3_len(path), &blocker, cx->stats, 1);
        if (blocker.hit) continue;

        float power = (float)(l->albedo / (4.0f * M_PI * v3_len2(path)));
        sum = v3_add(sum, v3_scale(l->color, power * cosine));
        reached += 1;
    }
    out[0] = sum.x; out[1] = sum.y; out[2] = sum.z; out[3] = reached;
}

/* nearest texel, src/math.c:269-276.  The index arithmetic is in double and
 * truncates toward zero.  The reference reads out of bounds for uv outside
 * [0,1]; that is undefined there and clamped here. */
static v3 orc_texel(const rt_image3f *tex, rt_vec2f uv, orc_stats *st)
{
    if (st) st->tex_lookups++;
    double fu = uv.x * (tex->size.x - 0.5);
    double fv = uv.y * (tex->size.y - 0.5);
    int ui = !(fu >= 0.0) ? 0 : (fu >= (double)tex->size.x ? (int)tex->size.x - 1 : (int)fu);
    int vi = !(fv >= 0.0) ? 0 : (fv >= (double)tex->size.y ? (int)tex->size.y - 1 : (int)fv);
    return tex->pixels[(size_t)vi * tex->size.x + ui];
}

/* one camera path: src/main.c:611-740 */
static v3 orc_trace_path(orc_ctx *cx, const rt_scene *scene, rt_ray ray, int max_segments)
{
    float lights_reached = 0;
    v3 throughput = v3_make(1.0f, 1.0f, 1.0f);
    v3 radiance = v3_make(0.0f, 0.0f, 0.0f);
    int inside_glass = 0;

    for (int seg = 0; seg < max_segments; seg++) {
        rt_ray_result hit;
        orc_closest_hit(scene, &ray, FLT_MAX, &hit, cx->stats, 0);
        if (!hit.hit) break;

        if (hit.object_type == RT_LIGHT) {                    /* :631-643 */
            lights_reached += 1;
            const rt_light *l = scene->lights + hit.object_id;
            float scale = (float)(l->albedo / (4.0f * M_PI * hit.t_min * hit.t_min));
            radiance = v3_add(radiance, v3_mul(throughput, v3_scale(l->color, scale)));
            break;
        }
        const rt_material *m = scene->materials + hit.material_id;
        if (m->mirror) {                                      /* :644-653 */
            throughput = v3_mul(throughput, m->mirror_color);
            v3 along = v3_scale(hit.surface_normal, v3_dot(hit.surface_normal, ray.direction));
            v3 across = v3_sub(ray.direction, along);
            ray.origin = hit.intersection_point;
            ray.direction = v3_sub(across, v3_scale(along, 2.0f));   /* d - 3 n (n.d), kept unnormalised */
        } else if (m->transparent) {                          /* :654-674 */
            v3 along = v3_scale(hit.surface_normal, v3_dot(hit.surface_normal, ray.direction));
            v3 across = v3_sub(ray.direction, along);
            float ratio = inside_glass ? m->n : (1.0f / m->n);
            v3 bent = v3_scale(across, ratio);
            if (v3_len2(bent) > 1) {                          /* total internal reflection */
                ray.origin = hit.intersection_point;
                ray.direction = v3_sub(across, along);
            } else {
                inside_glass = !inside_glass;
                v3 forward = v3_scale(v3_unit(along), sqrtf(1 - v3_len2(bent)));
                ray.origin = hit.intersection_point;
                ray.direction = v3_add(forward, bent);
            }
        } else {                                              /* :675-732 */
            float direct[4];
            orc_direct_light(cx, scene, &hit, direct);
            lights_reached += direct[3];
            v3 albedo = m->diffuse_color;
            if (m->textured) albedo = orc_texel(m->texture, hit.triangle_uv, cx->stats);
            radiance = v3_add(radiance, v3_mul(throughput, v3_mul(v3_make(direct[0], direct[1], direct[2]), albedo)));
            throughput = v3_mul(throughput, albedo);
            if ((double)v3_len2(throughput) < 1e-6) break;
            ray.origin = hit.intersection_point;

            float r[4];
            ctx_uniform4(cx, 0, 1, r);
            float s, c;
            ctx_sincos(cx, (float)(2.0f * M_PI * r[0]), &s, &c);
            v3 local = v3_make(c * sqrtf(1 - r[1]), s * sqrtf(1 - r[1]), sqrtf(r[1]));
            v3 n = hit.surface_normal;
            v3 t0 = frame_tangent(n);
            v3 t1 = v3_cross(n, t0);
            v3 dir = v3_add(v3_scale(n, local.z), v3_add(v3_scale(t0, -local.y), v3_scale(t1, -local.x)));
            dir = v3_unit(dir);
            throughput = v3_scale(throughput, v3_dot(dir, n));
            if ((double)v3_len2(throughput) < 1e-6) break;
            ray.direction = dir;
        }
    }
    if (lights_reached == 0) return v3_make(0.0f, 0.0f, 0.0f);
    return v3_scale(radiance, 1 / lights_reached);
}

/* camera basis and the per-frame constants of src/main.c:753-755,760,767,788 */
typedef struct orc_camera { v3 pos, forward, right, up; float focal; float norm; } orc_camera;

static orc_camera camera_setup(const rt_scene *scene, const rt_image3f *image, uint32_t spp_total)
{
    orc_camera cam;
    const rt_camera *c = &scene->camera;
    cam.pos = c->position;
    cam.forward = v3_unit(v3_sub(c->lookAt, c->position));
    cam.right = v3_unit(v3_cross(cam.forward, c->up));
    cam.up = v3_cross(cam.right, cam.forward);
    cam.focal = (float)image->size.y / (2.0f * tanf(c->field_of_view / 2.0f));
    float sigma = 0.5f;
    cam.norm = (float)(1.0f / (sigma * sigma * 2.0f * M_PI * spp_total));
    return cam;
}

/* one sample of pixel (x, y): src/main.c:764-779 */
static v3 orc_sample(orc_ctx *cx, const rt_scene *scene, const rt_image3f *image, const orc_camera *cam,
                     uint32_t x, uint32_t y, int max_segments)
{
    float j[2];
    ctx_gaussian2(cx, 0, 0.25f, j);
    if (cx->stats) cx->stats->samples++;
    v3 dr = v3_scale(cam->right, (float)x - (float)image->size.x / 2.0f + 0.5f + j[0]);
    v3 du = v3_scale(cam->up, -((float)y - (float)image->size.y / 2.0f + 0.5f + j[1]));
    v3 dm = v3_scale(cam->forward, cam->focal);
    rt_ray ray;
    ray.origin = cam->pos;
    ray.direction = v3_unit(v3_add(v3_add(dr, du), dm));
    v3 c = orc_trace_path(cx, scene, ray, max_segments);
    if ((double)v3_len2(c) > 1e4) c = v3_scale(c, (float)(1e2 / v3_len(c)));   /* firefly clamp */
    return c;
}

/* ------------------------------------------------------------------------ */
/* exported entry points                                                    */
/* ------------------------------------------------------------------------ */

/* raytrace_region with the reference's own generator (serial through the region). */
ORC_EXPORT void orc_render_region_aes(const rt_job *job, int math_mode, int max_segments, orc_stats *stats)
{
    aes_init();
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_AES; cx.math_mode = math_mode; cx.stats = stats;
    memcpy(cx.aes, job->serie.bytes, 32);
    const rt_image3f *image = job->image;
    orc_camera cam = camera_setup(job->scene, image, job->ray_per_pixel);
    uint32_t x1 = job->region.point.x + job->region.size.x;
    uint32_t y1 = job->region.point.y + job->region.size.y;
    for (uint32_t y = job->region.point.y; y < y1; y++)
        for (uint32_t x = job->region.point.x; x < x1; x++) {
            v3 total = v3_make(0, 0, 0);
            for (uint32_t i = 0; i < job->ray_per_pixel; i++)
                total = v3_add(total, orc_sample(&cx, job->scene, image, &cam, x, y, max_segments));
            image->pixels[(size_t)image->size.x * y + x] = v3_scale(total, cam.norm);
        }
}

/* The same loop with the counter generator: sample i of pixel p draws from
 * the stream keyed (seed, p, sample_offset + i), whatever region, tile or
 * device renders it.  spp_total is what the final normalisation divides by
 * (the caller may render a sample batch of a longer run); accumulate != 0
 * adds the un-normalised sum to the image instead of storing. */
ORC_EXPORT void orc_render_region_counter(const rt_scene *scene, rt_image3f *image, const rt_rect2u *region,
                                          uint32_t spp, uint32_t sample_offset, uint64_t seed,
                                          int math_mode, int max_segments, orc_stats *stats)
{
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_COUNTER; cx.math_mode = math_mode; cx.stats = stats;
    cx.seed[0] = (uint32_t)seed; cx.seed[1] = (uint32_t)(seed >> 32);
    orc_camera cam = camera_setup(scene, image, spp);
    uint32_t x1 = region->point.x + region->size.x, y1 = region->point.y + region->size.y;
    for (uint32_t y = region->point.y; y < y1; y++)
        for (uint32_t x = region->point.x; x < x1; x++) {
            uint32_t pixel = image->size.x * y + x;
            v3 total = v3_make(0, 0, 0);
            for (uint32_t i = 0; i < spp; i++) {
                ctx_begin_sample(&cx, pixel, sample_offset + i);
                total = v3_add(total, orc_sample(&cx, scene, image, &cam, x, y, max_segments));
            }
            image->pixels[(size_t)image->size.x * y + x] = v3_scale(total, cam.norm);
        }
}

/* single-function probes for known-answer tests */
ORC_EXPORT void orc_find_ray_hit(const rt_scene *scene, const rt_ray *ray, float max_distance,
                                 rt_ray_result *out, orc_stats *stats)
{
    orc_closest_hit(scene, ray, max_distance, out, stats, 0);
}

ORC_EXPORT void orc_aes_uniform4(rt_random_serie *serie, float lo, float hi, float out[4])
{
    aes_init();
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_AES;
    memcpy(cx.aes, serie->bytes, 32);
    ctx_uniform4(&cx, lo, hi, out);
    memcpy(serie->bytes, cx.aes, 32);
}

ORC_EXPORT void orc_aes_gaussian2(rt_random_serie *serie, float mean, float sigma, int math_mode, float out[2])
{
    aes_init();
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_AES; cx.math_mode = math_mode;
    memcpy(cx.aes, serie->bytes, 32);
    ctx_gaussian2(&cx, mean, sigma, out);
    memcpy(serie->bytes, cx.aes, 32);
}

ORC_EXPORT void orc_threefry2x32_20(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2])
{
    threefry2x32_20(key, ctr, out);
}

/* uniforms of draw `draw` of (seed, pixel, sample) in counter mode */
ORC_EXPORT void orc_counter_uniform2(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw, float out[2])
{
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_COUNTER;
    cx.seed[0] = (uint32_t)seed; cx.seed[1] = (uint32_t)(seed >> 32);
    ctx_begin_sample(&cx, pixel, sample);
    cx.draw = draw;
    float u[4];
    ctx_uniform4(&cx, 0, 1, u);
    out[0] = u[0]; out[1] = u[1];
}

ORC_EXPORT void orc_direct_light_aes(const rt_job *job_with_serie, const rt_ray_result *at, int math_mode,
                                     float out[4], rt_random_serie *serie_after)
{
    aes_init();
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_AES; cx.math_mode = math_mode;
    memcpy(cx.aes, job_with_serie->serie.bytes, 32);
    orc_direct_light(&cx, job_with_serie->scene, at, out);
    if (serie_after) memcpy(serie_after->bytes, cx.aes, 32);
}

ORC_EXPORT void orc_trace_ray_aes(const rt_job *job_with_serie, const rt_ray *ray, int math_mode,
                                  int max_segments, float out[3], rt_random_serie *serie_after)
{
    aes_init();
    orc_ctx cx; memset(&cx, 0, sizeof(cx));
    cx.rng_mode = ORC_RNG_AES; cx.math_mode = math_mode;
    memcpy(cx.aes, job_with_serie->serie.bytes, 32);
    v3 c = orc_trace_path(&cx, job_with_serie->scene, *ray, max_segments);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    if (serie_after) memcpy(serie_after->bytes, cx.aes, 32);
}

ORC_EXPORT void orc_sample_texture(const rt_image3f *tex, const rt_vec2f *uv, float out[3])
{
    v3 c = orc_texel(tex, *uv, NULL);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

ORC_EXPORT void orc_sincosf(const float *x, float *s, float *c, size_t n)
{
    for (size_t i = 0; i < n; i++) rt_sincosf(x[i], s + i, c + i);
}
ORC_EXPORT void orc_logf(const float *x, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = rt_logf(x[i]);
}

/* gamma + truncating quantiser of src/main.c:65-90 (without the PNG write):
 * min(c, 1) with the reference's operand order (NaN -> 1), powf(c, 1/2.2f),
 * times 255, C cast to uint8_t. */
ORC_EXPORT void orc_quantize(const rt_image3f *image, uint8_t *rgb)
{
    size_t n = (size_t)image->size.x * image->size.y;
    for (size_t i = 0; i < n; i++) {
        const float *p = &image->pixels[i].x;
        for (int k = 0; k < 3; k++) {
            float c = sel_min(p[k], 1.0f);
            c = powf(c, 1.0f / 2.2f);
            c = c * 255.0f;
            rgb[3 * i + k] = (uint8_t)c;
        }
    }
}

/* the same conversion with the power evaluated by rt_gamma_powf instead of libm: what the
 * device epilogue computes (oracle/check_gamma.c compares the two over every input) */
ORC_EXPORT void orc_quantize_portable(const float *values, size_t n, uint8_t *bytes)
{
    for (size_t i = 0; i < n; i++) bytes[i] = rt_gamma_byte(values[i]);
}

/* ------------------------------------------------------------------------ */
/* threaded tile driver for CPU timing: the shape of src/main.c:804-833      */
/* (atomic tile counter, one generator stream per tile)                      */
/* ------------------------------------------------------------------------ */
typedef struct orc_tile_run {
    const rt_scene *scene; rt_image3f *image;
    uint32_t tile, tiles_x, tiles_y, spp;
    int rng_mode, math_mode, max_segments;
    const rt_random_serie *seeds; uint64_t seed;
    volatile uint32_t next;
    orc_stats *per_thread_stats; uint32_t nthreads;
    volatile uint32_t thread_ids;
} orc_tile_run;

static void *tile_worker(void *arg)
{
    orc_tile_run *run = arg;
    uint32_t me = __sync_fetch_and_add(&run->thread_ids, 1);
    orc_stats *st = run->per_thread_stats ? run->per_thread_stats + me : NULL;
    for (;;) {
        uint32_t id = __sync_fetch_and_add(&run->next, 1);
        if (id >= run->tiles_x * run->tiles_y) return NULL;
        uint32_t tx = id % run->tiles_x, ty = id / run->tiles_x;
        rt_job job; memset(&job, 0, sizeof(job));
        job.scene = (rt_scene *)run->scene; job.image = run->image;
        job.region.point.x = tx * run->tile; job.region.point.y = ty * run->tile;
        uint32_t w = run->image->size.x - job.region.point.x, h = run->image->size.y - job.region.point.y;
        job.region.size.x = w < run->tile ? w : run->tile;
        job.region.size.y = h < run->tile ? h : run->tile;
        job.ray_per_pixel = run->spp;
        if (run->rng_mode == ORC_RNG_AES) {
            job.serie = run->seeds[id];
            orc_render_region_aes(&job, run->math_mode, run->max_segments, st);
        } else {
            orc_render_region_counter(run->scene, run->image, &job.region, run->spp, 0, run->seed,
                                      run->math_mode, run->max_segments, st);
        }
    }
}

/* returns wall seconds of the render alone; stats (may be NULL) receives the sum over threads */
ORC_EXPORT double orc_render_tiles(const rt_scene *scene, rt_image3f *image, uint32_t spp, uint32_t tile,
                                   uint32_t nthreads, int rng_mode, int math_mode, int max_segments,
                                   const rt_random_serie *seeds, uint64_t seed, orc_stats *stats)
{
    aes_init();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    orc_tile_run run; memset(&run, 0, sizeof(run));
    run.scene = scene; run.image = image; run.tile = tile; run.spp = spp;
    run.tiles_x = (image->size.x + tile - 1) / tile;
    run.tiles_y = (image->size.y + tile - 1) / tile;
    run.rng_mode = rng_mode; run.math_mode = math_mode; run.max_segments = max_segments;
    run.seeds = seeds; run.seed = seed; run.nthreads = nthreads;
    run.per_thread_stats = stats ? calloc(nthreads, sizeof(orc_stats)) : NULL;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_t th[256];
    for (uint32_t i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, tile_worker, &run);
    for (uint32_t i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (uint32_t i = 0; i < nthreads; i++) {
            const uint64_t *src = (const uint64_t *)(run.per_thread_stats + i);
            uint64_t *dst = (uint64_t *)stats;
            for (size_t k = 0; k < sizeof(orc_stats) / 8; k++) dst[k] += src[k];
        }
        free(run.per_thread_stats);
    }
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
