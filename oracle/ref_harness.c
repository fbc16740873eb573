/*
 * ref_harness.c -- TEST INFRASTRUCTURE, never part of the product.
 *
 * Builds the UNMODIFIED reference sources, where they lie under
 * /root/reference, into a shared object whose exported ref_* wrappers let the
 * tests call the reference's own functions on arbitrary scenes.  Nothing from
 * the reference is copied: its translation unit is pulled in by absolute path,
 * so this file only compiles in a container that has /root/reference.
 *
 * The reference is a unity build (src/main.c includes math.c, the stb
 * implementations and the fast_obj implementation), so including main.c gives
 * this object every function of the program; its main() is renamed out of the
 * way.  The wrappers pass structs by pointer because ctypes is happier so.
 */
#ifdef REF_PORTABLE_MATH
/* Diagnostic variant: the reference's source with its three libm calls on the
 * render path (cosf, sinf, logf) answered by rt_math_portable.h, the same
 * correctly-rounded routines the GPU kernels use.  glibc's own versions are
 * faithful but not correctly rounded (about 1.3 % of the reachable arguments
 * differ in the last bit), which no GPU can imitate. */
#include <math.h>
#include "rt_math_portable.h"
static float ref_pm_sinf(float x) { float s, c; rt_sincosf(x, &s, &c); return s; }
static float ref_pm_cosf(float x) { float s, c; rt_sincosf(x, &s, &c); return c; }
static float ref_pm_logf(float x) { return rt_logf(x); }
#define sinf(x) ref_pm_sinf(x)
#define cosf(x) ref_pm_cosf(x)
#define logf(x) ref_pm_logf(x)
#endif

#define main ref_program_main
#include "/root/reference/src/main.c"
#undef main

#include <string.h>
#include <time.h>

/* sizeof() of every record that crosses the drop-in boundary, in a fixed
 * order (tests compare them with include/rt_scene.h). */
void ref_sizes(uint32_t out[16])
{
    uint32_t v[16] = {
        sizeof(Scene), sizeof(Material), sizeof(Sphere), sizeof(Plane),
        sizeof(Light), sizeof(Scene_Mesh), sizeof(Mesh), sizeof(Mesh_Node),
        sizeof(Triangle), sizeof(Ray_Result), sizeof(Job), sizeof(Random_Serie),
        sizeof(Image3f), sizeof(Camera), sizeof(Transform), sizeof(Ray),
    };
    memcpy(out, v, sizeof(v));
}

/* src/main.c:742 */
void ref_raytrace_region(Job *job) { raytrace_region(job); }

/* src/main.c:264 */
void ref_find_ray_hit(Scene *scene, const Ray *ray, float max_distance, Ray_Result *out)
{
    *out = find_ray_hit(scene, *ray, max_distance);
}

/* src/main.c:546; out = {color.r, color.g, color.b, light_hit} */
void ref_compute_light_contribution(Job *job, Scene *scene, const Ray *previous_ray,
                                    const Ray_Result *previous_result, float out[4])
{
    Light_Contribution c = compute_light_contribution(job, scene, *previous_ray, *previous_result);
    out[0] = c.color.r; out[1] = c.color.g; out[2] = c.color.b; out[3] = c.light_hit;
}

/* src/main.c:611 */
void ref_raytrace_ray(Job *job, Scene *scene, const Ray *ray, float out[3])
{
    Vec3f c = raytrace_ray(job, scene, *ray);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}

/* src/math.c:148 (seeds from libc rand()) */
void ref_create_random_serie(Random_Serie *out) { *out = create_random_serie(); }

/* src/math.c:164 */
void ref_random4f_uniform(Random_Serie *serie, float min, float max, float out[4])
{
    Vec4f v = random4f_uniform(serie, min, max);
    memcpy(out, v.coord, sizeof(v.coord));
}

/* src/math.c:225 */
void ref_random2f_gaussian(Random_Serie *serie, float mean, float sigma, float out[2])
{
    Vec2f v = random2f_gaussian(serie, mean, sigma);
    out[0] = v.x; out[1] = v.y;
}

/* src/math.c:269 */
void ref_sample_texture(Image3f *texture, const Vec2f *uv, float out[3])
{
    Vec3f c = sample_texture(texture, *uv);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}

/* src/main.c:177; the Mesh is returned in caller storage */
void ref_load_mesh(const char *filename, Mesh *out) { *out = load_mesh((char *)filename); }

/* src/main.c:45 */
void ref_load_texture(const char *filename, Image3f *out) { *out = load_texture((char *)filename); }

/* src/main.c:26 */
void ref_create_blank_image(uint32_t w, uint32_t h, Image3f *out) { *out = create_blank_image(w, h); }

/* src/main.c:65: gamma + truncating quantiser.  save_image only writes a PNG,
 * so the bytes are recovered by decoding that (lossless) file again. */
int ref_quantize_image(Image3f *image, const char *tmp_png, uint8_t *rgb_out)
{
    if (!save_image((char *)tmp_png, image)) return -1;
    int w = 0, h = 0, c = 0;
    uint8_t *px = stbi_load(tmp_png, &w, &h, &c, 3);
    if (!px) return -2;
    memcpy(rgb_out, px, (size_t)w * h * 3);
    free(px);
    return 0;
}

/* Raw parse of an OBJ file by the vendored parser (src/fast_obj.h:1258), before
 * load_mesh touches it: used once, to turn the dog model into a data fixture. */
typedef struct Ref_Obj_Arrays {
    uint32_t position_count; float *positions;
    uint32_t texcoord_count; float *texcoords;
    uint32_t normal_count;   float *normals;
    uint32_t face_count;     uint32_t *face_vertices;
    uint32_t index_count;    uint32_t *indices;       /* p,t,n triples */
} Ref_Obj_Arrays;

int ref_fast_obj_read(const char *filename, Ref_Obj_Arrays *out)
{
    fastObjMesh *m = fast_obj_read(filename);
    if (!m) return -1;
    out->position_count = m->position_count; out->positions = m->positions;
    out->texcoord_count = m->texcoord_count; out->texcoords = m->texcoords;
    out->normal_count = m->normal_count;     out->normals = m->normals;
    out->face_count = m->face_count;         out->face_vertices = m->face_vertices;
    out->index_count = array_size(m->indices);
    out->indices = (uint32_t *)m->indices;
    return 0;
}

/* JPEG/PNG decode by the vendored decoder (src/stb_image.h:1306), 3 channels. */
uint8_t *ref_stbi_load_rgb(const char *filename, int *w, int *h)
{
    int c = 0;
    return stbi_load(filename, w, h, &c, 3);
}

/* BVH census: node count, leaf count, deepest level, largest leaf. */
static void census(const Mesh_Node *n, int depth, uint32_t acc[4])
{
    acc[0]++;
    if ((uint32_t)depth > acc[2]) acc[2] = depth;
    if (!n->childs) {
        acc[1]++;
        uint32_t cnt = n->last_triangle_id - n->first_triangle_id + 1;
        if (cnt > acc[3]) acc[3] = cnt;
        return;
    }
    census(n->childs + 0, depth + 1, acc);
    census(n->childs + 1, depth + 1, acc);
}
void ref_bvh_census(const Mesh *mesh, uint32_t out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    census(&mesh->root, 1, out);
}

/*
 * CPU timing baseline: the reference's own tile scheduler shape
 * (src/main.c:804-833: atomic tile counter, one fresh Random_Serie per tile)
 * driving the reference's own raytrace_region, but for any image size / spp
 * and with caller-provided seeds (thread_entry_point itself hard-wires the
 * IMAGE_SIZE / RAY_PER_PIXEL macros and seeds racily from rand()).
 */
typedef struct Ref_Tile_Run {
    Scene *scene; Image3f *image;
    uint32_t tile, tiles_x, tiles_y, spp;
    const Random_Serie *seeds;          /* one per tile */
    volatile uint32_t next;
} Ref_Tile_Run;

static void *ref_tile_worker(void *arg)
{
    Ref_Tile_Run *run = arg;
    for (;;) {
        uint32_t id = __sync_fetch_and_add(&run->next, 1);
        if (id >= run->tiles_x * run->tiles_y) return NULL;
        uint32_t tx = id % run->tiles_x, ty = id / run->tiles_x;
        Job job = {
            .scene = run->scene, .image = run->image,
            .region.point.x = tx * run->tile, .region.point.y = ty * run->tile,
            .region.size.x = MIN(run->tile, run->image->size.width - tx * run->tile),
            .region.size.y = MIN(run->tile, run->image->size.height - ty * run->tile),
            .ray_per_pixel = run->spp,
            .serie = run->seeds[id],
        };
        raytrace_region(&job);
    }
}

/* Returns wall seconds of the render alone. */
double ref_render_tiles(Scene *scene, Image3f *image, uint32_t spp, uint32_t tile,
                        uint32_t nthreads, const Random_Serie *seeds)
{
    Ref_Tile_Run run = {
        .scene = scene, .image = image, .tile = tile, .spp = spp, .seeds = seeds,
        .tiles_x = (image->size.width + tile - 1) / tile,
        .tiles_y = (image->size.height + tile - 1) / tile,
        .next = 0,
    };
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_t th[256];
    if (nthreads > 256) nthreads = 256;
    for (uint32_t i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, ref_tile_worker, &run);
    for (uint32_t i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
