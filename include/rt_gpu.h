/*
 * rt_gpu.h -- C ABI of the MI355X render loop (librt_gpu.so).
 *
 * The library replaces ONE function of the host program, the per-pixel
 * path-tracing loop
 *
 *     void raytrace_region(Job *job);            (reference: src/main.c:742)
 *
 * together with everything it calls (raytrace_ray src/main.c:611, find_ray_hit
 * src/main.c:264, compute_light_contribution src/main.c:546, sample_texture
 * src/math.c:269, the Random_Serie generator src/math.c:158-235) and the tile
 * loop that feeds it (thread_entry_point src/main.c:804-833).  Scene loading,
 * the scene records and image output stay with the host program.
 *
 * Plain C: pointers and sizes only.  All entry points return 0 on success and
 * a negative rt_gpu_status otherwise; rt_gpu_last_error() gives the text.
 * A scene handle may be used by one host thread at a time; different handles
 * (also of the same scene) may be used concurrently, each on its own stream.
 */
#ifndef RT_GPU_H
#define RT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef RT_SCENE_TYPES_FROM_HOST
/* Compiled inside the host program: reuse its own records (same layout). */
typedef Scene      rt_scene;
typedef Image3f    rt_image3f;
typedef Rect2i     rt_rect2u;
typedef Job        rt_job;
typedef Ray        rt_ray;
typedef Ray_Result rt_ray_result;
#else
#include "rt_scene.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_gpu_status {
    RT_GPU_OK = 0,
    RT_GPU_ERR_ARGUMENT = -1,      /* NULL / out-of-range argument */
    RT_GPU_ERR_SCENE = -2,         /* scene graph inconsistent (bad index, bad pointer, tree too deep) */
    RT_GPU_ERR_DEVICE = -3,        /* a HIP call failed */
    RT_GPU_ERR_NO_DEVICE = -4,     /* no usable GPU */
    RT_GPU_ERR_MEMORY = -5         /* host or device allocation failed */
} rt_gpu_status;

typedef struct rt_gpu_scene rt_gpu_scene;      /* opaque: the scene flattened and resident in HBM */

enum {
    RT_GPU_FLAG_DETAILED_STATS = 1u << 0,  /* count node pops, triangle tests, ... (slower) */
    RT_GPU_FLAG_ACCUMULATE     = 1u << 1,  /* pixels hold running UN-normalised sums: read, add, store */
    RT_GPU_FLAG_COMPACT_BANDS  = 1u << 2,  /* device output holds only this rank's bands, packed */
    RT_GPU_FLAG_KERNEL_TIMES   = 1u << 3   /* time every launch with HIP events on its stream (rt_gpu_stats.*_ms; implied by DETAILED_STATS) */
};

/* Optional knobs; pass NULL for the host program's behaviour.  Zero-fill, then
 * set struct_size = sizeof(rt_gpu_options). */
typedef struct rt_gpu_options {
    uint32_t struct_size;
    uint32_t max_segments;   /* path length cap; 0 = 64 (reference: src/main.c:624) */
    uint32_t sample_offset;  /* index of the first sample of this call (sample batches) */
    uint32_t spp_total;      /* sample count the final normalisation divides by; 0 = spp */
    uint32_t flags;          /* RT_GPU_FLAG_* */
    uint32_t band_rows;      /* split over devices: rows per band (multiple of 8); 0 = no split */
    uint32_t band_rank;      /* this device renders bands rank, rank + count, ... of the region */
    uint32_t band_count;
    uint32_t frames;         /* device render only: frames rendered by this call; 0 or 1 = one.  Frame k is made of
                              * samples sample_offset + k*spp ... + spp - 1 of every pixel and is stored frame_stride
                              * floats after frame k-1: the same pictures as `frames` calls with those sample offsets,
                              * from launches `frames` times as large (small regions fill the GPU better that way) */
    uint32_t lane;           /* device render only: which of the handle's work-memory lanes (own stream, own lists) the
                              * call starts on, < RT_GPU_STREAMS.  Calls of one handle on different lanes, each ordered on
                              * its own hip_stream, overlap on the GPU: two or three frames in flight from one host
                              * thread.  0 = the first lane */
    uint64_t frame_stride;   /* floats between the first pixels of consecutive frames; 0 = the size of one frame's
                              * output: width*height*3, or with packed bands rt_gpu_multi_rows_per_rank(region height,
                              * band_rows, band_count) rows of the region's width (the buffer of the rank with the most
                              * rows: the same on every rank, as an all-gather of the ranks' buffers needs) */
} rt_gpu_options;

/* Work counters of the calls since the last rt_gpu_reset_stats.  A "ray" is
 * one closest-hit query (camera, bounce and shadow rays alike). */
typedef struct rt_gpu_stats {
    uint64_t samples, rays, shadow_rays;
    uint64_t root_tests, interior_pops, leaf_pops, tri_tests, tri_hits, tex_lookups, rng_draws;
    uint64_t light_tests, sphere_tests, plane_tests;
    uint64_t shade_steps, pop_steps, tri_steps;  /* wave-level scheduler steps of each kind ...          */
    uint64_t shade_lanes, pop_lanes, tri_lanes;  /* ... and the lanes that took part in them (of 64 each) */
    uint64_t tree_entries;                       /* (ray, instance) pairs whose tree was entered          */
                             /* all but the first line only with RT_GPU_FLAG_DETAILED_STATS */
    double   kernel_ms;      /* device time of the shade, tree and finish kernels (HIP events on their stream; calls made
                              * with RT_GPU_FLAG_KERNEL_TIMES or DETAILED_STATS only) */
    double   shade_ms, tree_ms;  /* ... and each kernel's share */
    double   fresh_ms;           /* part of shade_ms spent in the launches that start new paths (first of each batch) */
    double   finish_ms;          /* rt_finish_kernel (last paths of a batch), not part of shade_ms or tree_ms */
    uint64_t rounds;         /* shade launches that had work, summed over batches (the first of a batch included) */
    uint64_t overflowed;     /* paths that found the parked list full (the lists hold every path of a batch, or the
                              * batch's paths start in several passes, so 0; if not, rt_gpu_get_stats fails) */
    uint32_t launches;       /* render calls x sample batches */
    uint32_t passes;         /* launch sequences enqueued for those batches by rt_gpu_render_region_device: one per batch
                              * when the lists of paths in transit hold every path of the batch, up to 8 when they are
                              * smaller (the new paths of a batch then start in turns); a pass that finds no path left to
                              * start returns at once */
    uint32_t passes_with_work;   /* ... and how many of them started paths: equal to `launches` as long as one pass per batch does */
    uint32_t lists_full;     /* 1: the handle's lists are sized for every path of a batch.  A handle starts with lists of
                              * RT_GPU_PARKED_PERCENT (default 40) of that size; the first batch that needs a second pass
                              * makes the handle's later calls allocate full lists (up to RT_GPU_PARKED_MB and the free memory) */
} rt_gpu_stats;

/* Library / device ------------------------------------------------------- */

/* Number of usable GPUs (0 if none); never fails. */
int rt_gpu_device_count(void);

/* Text of the calling thread's last error ("" if none). */
const char *rt_gpu_last_error(void);

/* Scene ------------------------------------------------------------------
 * Walks the host scene graph once (Scene, src/main.h:7-25: materials, spheres,
 * planes, lights, mesh instances with their Mesh and box tree, textures),
 * de-duplicates meshes and textures by pointer, flattens the box trees and
 * uploads everything to the CURRENT HIP device.  No host pointer is kept
 * after it returns.  What raytrace_region reads through job->scene on every
 * call is read here once. */
int rt_gpu_scene_create(const rt_scene *scene, rt_gpu_scene **out);
void rt_gpu_scene_destroy(rt_gpu_scene *scene);

/* What making the scene resident cost and what it holds. */
typedef struct rt_gpu_scene_info_t {
    double   flatten_ms;       /* host: walking the scene graph into flat arrays */
    double   upload_ms;        /* device allocations, copies, streams and events */
    uint64_t resident_bytes;   /* tree nodes, triangles, attributes and texels in HBM */
    uint64_t work_bytes;       /* work memory the render calls so far have made the handle allocate (all lanes) */
    uint32_t mesh_count, instance_count, texture_count;   /* distinct meshes and textures after de-duplication */
    uint32_t stack_levels;     /* traversal stack entries per ray */
} rt_gpu_scene_info_t;
int rt_gpu_scene_info(const rt_gpu_scene *scene, rt_gpu_scene_info_t *out);

/* Replace the camera of a resident scene (Scene.camera, src/main.h:9). */
int rt_gpu_scene_set_camera(rt_gpu_scene *scene, const void *camera /* Camera, 40 bytes */);

/* Render ------------------------------------------------------------------
 * Drop-in for raytrace_region (src/main.c:742) plus the tile loop around it
 * (src/main.c:804-833): renders `spp` samples for every pixel of `region`
 * and stores image->pixels[width*y + x] (host memory) exactly as the
 * reference does.  Sample i of pixel p draws its random numbers from a
 * counter-based stream keyed (seed, p, sample_offset + i), so the result does
 * not depend on how the frame is cut into regions, bands or devices. */
int rt_gpu_render_region(rt_gpu_scene *scene, rt_image3f *image, rt_rect2u region,
                         uint32_t spp, uint64_t seed, const rt_gpu_options *options);

/* The same render into DEVICE memory (width*height float RGB triples, or the
 * packed bands of this rank with RT_GPU_FLAG_COMPACT_BANDS), ordered on `hip_stream`
 * (a hipStream_t; NULL = the default stream): the render starts after the work
 * already on that stream and the stream goes on after the render's last kernel.
 * The call is asynchronous: it enqueues a fixed sequence of launches (the number of
 * rounds a frame needs is decided on the device; launches with nothing to do
 * return at once) and reads nothing back.  The samples of a call are rendered in
 * batches that go round-robin over a few streams of the library's own
 * (RT_GPU_STREAMS, default 3; RT_GPU_BATCH_PATHS paths per batch, default 80 Mi, then
 * cut down to what RT_GPU_SCRATCH_MB, RT_GPU_PARKED_MB and the free device memory
 * allow: a 1024 x 1024 x 64 spp frame is one batch; the lists of paths in transit start at
 * RT_GPU_PARKED_PERCENT, default 40, of the size that every path of a batch would need and grow to
 * it only after a batch needed more than one pass: rt_gpu_stats.lists_full), so the latency-bound end of one
 * batch overlaps the start of the next; the picture does not depend on the batching.  Work that consumes the pixels goes
 * on `hip_stream` (or synchronises with it).  The buffer belongs to the caller
 * (e.g. a tensor that a collective will gather). */
int rt_gpu_render_region_device(rt_gpu_scene *scene, float *device_pixels, uint32_t width, uint32_t height,
                                rt_rect2u region, uint32_t spp, uint64_t seed,
                                const rt_gpu_options *options, void *hip_stream);

/* The same render as rt_gpu_render_region_device for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The call makes no hipMalloc / hipFree / event creation and no
 * read-back: the kernels' work memory (queue counters, per-sample values, the list of paths in
 * transit) is a 13 MB array that is part of the library's code object, one copy per GPU, held until
 * the process ends.  Limit: (8x8 pixel blocks of the region) x 64 x spp <= 65536 paths, e.g. a 64x64
 * tile (the reference's tile size, src/main.c:794-824) at 16 samples per pixel; larger calls fail
 * with RT_GPU_ERR_ARGUMENT.  options.frames must be 0 or 1; the other options (sample_offset,
 * spp_total, ACCUMULATE, bands) work as above, so a frame of any sample count can be accumulated
 * from several calls.  Same pixels as rt_gpu_render_region_device, bit for bit.  All calls on one
 * GPU share that GPU's array: calls that overlap in time on one GPU (different streams or threads)
 * race; calls on one stream are ordered and safe. */
int rt_gpu_render_region_static(rt_gpu_scene *scene, float *device_pixels, uint32_t width, uint32_t height,
                                rt_rect2u region, uint32_t spp, uint64_t seed,
                                const rt_gpu_options *options, void *hip_stream);

/* Multiply un-normalised sums (RT_GPU_FLAG_ACCUMULATE) by the reference's
 * final factor 1/(0.5*0.5*2*pi*spp_total) (src/main.c:788). */
int rt_gpu_normalize_device(float *device_pixels, size_t pixel_count, uint32_t spp_total, void *hip_stream);

/* save_image's conversion (src/main.c:73-82: min(c,1), powf(c, 1/2.2f), x255, truncation) on
 * device pixels, 3 bytes per pixel into device memory, on the given stream: the frame leaves
 * the GPU as 3 MB of RGB8 instead of 12.6 MB of floats.  The power is evaluated in double
 * (rt_gamma_powf, raytracing_amd/csrc/rt_math_portable.h); checked over every float from 0 to
 * 1.125 against the host libm's powf: the bytes agree for all 1 066 401 793 inputs but one
 * (0x3c364a2a, where glibc 2.35's powf is one ulp above the correctly rounded value: 33
 * instead of 32).  Negative inputs, which the render loop cannot produce, are undefined in
 * the reference (NaN converted to uint8_t) and give 0 here. */
int rt_gpu_quantize_device(const float *device_pixels, size_t pixel_count, uint8_t *device_rgb, void *hip_stream);

/* Bit-compatible replay of raytrace_region(Job*) for a batch of jobs, each
 * with its own region, sample count and Random_Serie (Job, src/main.h:65-72):
 * one GPU lane walks one job in the reference's order and consumes the job's
 * AES generator stream exactly as the reference does.  jobs[i].scene is
 * ignored (the resident scene is used); all jobs must share one image.
 * Meant for validation and for small tiles; the counter-based entry points
 * above are the fast path. */
int rt_gpu_render_jobs(rt_gpu_scene *scene, const rt_job *jobs, uint32_t job_count,
                       const rt_gpu_options *options);

/* Batched find_ray_hit (src/main.c:264): out[i] = closest hit of rays[i] within
 * max_distance, every field as the reference fills it.  Runs the render path's own kernels
 * (primitive and root-box tests as the shade kernel makes them, rt_tree_kernel for the mesh
 * trees), so it probes exactly the code that renders. */
int rt_gpu_find_ray_hit(rt_gpu_scene *scene, const rt_ray *rays, uint32_t ray_count,
                        float max_distance, rt_ray_result *out);

/* The same query (src/main.c:264) by a plain one-lane-per-ray walk in the reference's loop
 * order: an independent second implementation to check the one above against. */
int rt_gpu_find_ray_hit_serial(rt_gpu_scene *scene, const rt_ray *rays, uint32_t ray_count,
                               float max_distance, rt_ray_result *out);

/* Gamma + truncating 8-bit quantiser of save_image (src/main.c:73-82) on host
 * float pixels; rgb receives 3 bytes per pixel. */
int rt_gpu_quantize(const float *pixels, size_t pixel_count, uint8_t *rgb);

/* One frame over several GPUs of a node, from ONE host process -------------------------
 * Replaces the thread pool of the host program (src/main.c:1056-1074: 15 threads, 64x64 tiles from an
 * atomic counter) by the node's GPUs: the frame is cut into 8-row bands dealt round-robin to the
 * devices, every device renders its bands (the scene is resident on each), one RCCL all-gather
 * over xGMI leaves every device with all bands, device 0 puts them in image order and the frame
 * lands in image->pixels.  The frame is the single-device frame bit for bit.  RCCL is loaded at
 * run time (librccl.so) and only when more than one device is used. */
typedef struct rt_gpu_multi rt_gpu_multi;
int rt_gpu_multi_create(const rt_scene *scene, int device_count /* 0 = every visible device */, rt_gpu_multi **out);
void rt_gpu_multi_destroy(rt_gpu_multi *multi);
int rt_gpu_multi_device_count(const rt_gpu_multi *multi);
/* Drop-in for the whole render block of main() (src/main.c:1056-1088) for one image. */
int rt_gpu_multi_render_frame(rt_gpu_multi *multi, rt_image3f *image, uint32_t spp, uint64_t seed,
                              const rt_gpu_options *options /* max_segments, sample_offset, spp_total; NULL = defaults */);
/* Rows of a device's packed band buffer (the last band of the image may be short: padding rows). */
uint32_t rt_gpu_multi_rows_per_rank(uint32_t height, uint32_t band_rows, uint32_t world);
/* The assembly step alone: `world` packed band buffers one after the other (as the all-gather leaves
 * them, rt_gpu_multi_rows_per_rank rows each) -> width*height pixels in image order, on device memory. */
int rt_gpu_assemble_bands_device(const float *device_all, float *device_frame, uint32_t width, uint32_t height,
                                 uint32_t band_rows, uint32_t world, void *hip_stream);

/* Statistics ---------------------------------------------------------------- */
int rt_gpu_get_stats(rt_gpu_scene *scene, rt_gpu_stats *out);
/* The same counters split by the kind of launch that did the work: out[0] the launches that start
 * new paths, out[1] the launches that resume parked paths, out[2] the tree launches, out[3] the
 * finishing launches; kernel_ms of each is that kind's device time. */
int rt_gpu_get_stats_by_kernel(rt_gpu_scene *scene, rt_gpu_stats out[4]);
int rt_gpu_reset_stats(rt_gpu_scene *scene);
/* Tree walks by the deepest traversal stack they needed, from the calls made with RT_GPU_FLAG_DETAILED_STATS (and the
 * closest-hit probes) since the last reset: out[k] = rays parked for a tree whose stack never held more than k entries
 * (k = 31: 31 or more).  The reference sizes its stack for the tree's depth + 1 (src/main.c:393-395); the walk only
 * stacks the far one of two entered children, so the occupied depth is far smaller, and the tree kernel keeps
 * RT_GPU_TREE_STACK entries per ray (default 6) in LDS and deeper ones in device memory. */
int rt_gpu_get_stack_histogram(rt_gpu_scene *scene, uint64_t out[32]);

/* Self-test hook: evaluates the device's sin/cos/log/divide/sqrt building
 * blocks on n host floats (x, and y for the quotients x/y) so that a test can
 * compare them bit for bit with the host's.  div_cached_out receives x/y as the
 * tree kernel computes it for box tests (reciprocal refined once per ray).  Any
 * output pointer may be NULL. */
int rt_gpu_eval_math(const float *x, const float *y, size_t n,
                     float *sin_out, float *cos_out, float *log_out, float *div_out, float *sqrt_out,
                     float *div_cached_out);
/* Self-test hook: n vectors (xyz, 3n floats) -> each vector divided by its length (3n floats) and the length
 * (n floats) as the path code computes them; the reference's normalize and norm (src/math.c:21-40: one square
 * root, three divisions) give the same bits. */
int rt_gpu_eval_unit(const float *xyz, size_t n, float *unit_out, float *length_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_GPU_H */
