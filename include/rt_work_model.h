/*
 * rt_work_model.h -- the constants that turn the library's work counters
 * (rt_gpu_stats, include/rt_gpu.h) into ALGORITHMIC bytes and flops of the
 * reference's path: what the reference's own records and expressions would
 * move and compute for the primitives, boxes, triangles, attributes and texels
 * that were actually tested.  One place, so that every tool prices a frame the
 * same way (bench.py reads this file).
 *
 * Bytes: sizes of the reference's records.
 * Flops: additions, multiplications, divisions, comparisons and square roots
 * counted from the reference's expressions, one each.
 */
#ifndef RT_WORK_MODEL_H
#define RT_WORK_MODEL_H

/* bytes per counted event */
#define RT_BYTES_LIGHT_TEST     32   /* Light (src/math.h:177-183) */
#define RT_BYTES_SPHERE_TEST    20   /* Sphere (src/math.h:170-175) */
#define RT_BYTES_PLANE_TEST     20   /* Plane (src/math.h:185-190) */
#define RT_BYTES_ROOT_TEST      72   /* Transform 48 (src/math.h:78-83) + root box 24 */
#define RT_BYTES_INTERIOR_POP   80   /* the two child Mesh_Nodes (src/math.h:103-111) */
#define RT_BYTES_TRI_TEST       72   /* Triangle 36 (src/math.h:85-95) + 3 points */
#define RT_BYTES_TRI_HIT        60   /* 3 normals + 3 texture coordinates (src/main.c:507-526) */
#define RT_BYTES_TEX_LOOKUP     12   /* one texel (src/math.c:269-276) */
#define RT_BYTES_PIXEL          12   /* the pixel store (src/main.c:789) */

/* flops per counted event */
#define RT_FLOPS_LIGHT_TEST     17   /* reduced quadratic up to the determinant test (src/main.c:283-313) */
#define RT_FLOPS_SPHERE_TEST    17   /* the same for spheres (src/main.c:318-350) */
#define RT_FLOPS_PLANE_TEST     12   /* two dot products, one division, two compares (src/main.c:358) */
#define RT_FLOPS_ROOT_TEST      55   /* ray into mesh space 33 (src/math.c:299-311) + slab test 22 (src/main.c:379-386) */
#define RT_FLOPS_INTERIOR_POP   44   /* two slab tests (src/main.c:400-440) */
#define RT_FLOPS_TRI_TEST       38   /* up to the distance reject (src/main.c:450-465) */
#define RT_FLOPS_TRI_HIT        58   /* the rest of the test 13 (src/main.c:466-472) + attributes 45 (src/main.c:507-526) */
#define RT_FLOPS_PATH_SEGMENT  150   /* shading of one camera or bounce ray's hit (src/main.c:631-732), transcendentals excluded */

#endif /* RT_WORK_MODEL_H */
