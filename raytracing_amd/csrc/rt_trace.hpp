/*
 * rt_trace.hpp -- device code of the path-tracing loop for gfx950 (wave64).
 *
 * Arithmetic contract: every float operation is a single IEEE binary32
 * rounding in the order the host program performs it (reference:
 * src/main.c:264-792, src/math.c:3-146,278-311); doubles appear exactly where
 * its C expressions promote to double.  The file must be compiled with
 * -ffp-contract=off; division and square root are the correctly rounded ones
 * (hipcc's default).  With that, a pixel computed here equals the pixel a
 * strict-IEEE CPU build computes, bit for bit.
 *
 * Structure (not the reference's): one lane owns one pixel (or one job);
 * scene constants come through scalar loads; the box tree is walked with a
 * per-lane stack of 32-bit entries in LDS, laid out [level][lane] so a wave's
 * push or pop touches 64 consecutive banks.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device_scene.h"
#include "rt_math_portable.h"

#define RT_WAVE 64
#define RT_PI_D 3.14159265358979323846

namespace rt {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
/* dot: ((0 + x) + y) + z like src/math.c:3-12 (the leading +0 turns a -0 product into +0) */
__device__ __forceinline__ float dot(f3 a, f3 b)
{
    float r = 0.0f;
    r += a.x * b.x; r += a.y * b.y; r += a.z * b.z;
    return r;
}
__device__ __forceinline__ float len2(f3 a) { return dot(a, a); }
__device__ __forceinline__ float len(f3 a) { return __builtin_sqrtf(len2(a)); }
__device__ __forceinline__ f3 unit(f3 a) { float n = len(a); return mk(a.x / n, a.y / n, a.z / n); }
__device__ __forceinline__ f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 scale(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 cross(f3 a, f3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* the host's MIN/MAX macros (src/math.h:8-9): a NaN operand selects the second */
__device__ __forceinline__ float sel_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sel_max(float a, float b) { return a > b ? a : b; }
/* a*w.x + b*w.y + c*w.z, src/math.c:278-287 */
__device__ __forceinline__ f3 blend(f3 a, f3 b, f3 c, f3 w)
{
    return mk(a.x * w.x + b.x * w.y + c.x * w.z,
              a.y * w.x + b.y * w.y + c.y * w.z,
              a.z * w.x + b.z * w.y + c.z * w.z);
}

/* ------------------------------------------------------------------ counters */
struct Counters {
    uint32_t samples, rays, shadow_rays, root_tests, interior_pops, leaf_pops, tri_tests, tri_hits,
             tex_lookups, rng_draws, light_tests, sphere_tests, plane_tests;
};

/* ------------------------------------------------------------------ generators */

/* Threefry-2x32 with 20 rounds (Salmon, Moraes, Dror, Shaw: "Parallel random
 * numbers: as easy as 1, 2, 3", SC'11).  Adds, rotates and xors only: the
 * integer multiplier of a CDNA SIMD runs at quarter rate. */
__device__ __forceinline__ void threefry2x32_20(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                                uint32_t &o0, uint32_t &o1)
{
    const uint32_t k2 = 0x1BD11BDAu ^ k0 ^ k1;
    uint32_t x0 = c0 + k0, x1 = c1 + k1;
#define RT_TF_ROUND(r) x0 += x1; x1 = __builtin_rotateleft32(x1, r); x1 ^= x0;
    RT_TF_ROUND(13) RT_TF_ROUND(15) RT_TF_ROUND(26) RT_TF_ROUND(6)
    x0 += k1; x1 += k2 + 1u;
    RT_TF_ROUND(17) RT_TF_ROUND(29) RT_TF_ROUND(16) RT_TF_ROUND(24)
    x0 += k2; x1 += k0 + 2u;
    RT_TF_ROUND(13) RT_TF_ROUND(15) RT_TF_ROUND(26) RT_TF_ROUND(6)
    x0 += k0; x1 += k1 + 3u;
    RT_TF_ROUND(17) RT_TF_ROUND(29) RT_TF_ROUND(16) RT_TF_ROUND(24)
    x0 += k1; x1 += k2 + 4u;
    RT_TF_ROUND(13) RT_TF_ROUND(15) RT_TF_ROUND(26) RT_TF_ROUND(6)
    x0 += k2; x1 += k0 + 5u;
#undef RT_TF_ROUND
    o0 = x0; o1 = x1;
}

/* 23 random mantissa bits under exponent 0 give v in [1,2); the host then
 * evaluates fma(v, 1, -1) for the range [0,1) (src/math.c:166-186), which is
 * exactly v - 1. */
__device__ __forceinline__ float bits_to_unit(uint32_t w)
{
    return __uint_as_float((w & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
}

/* Counter generator: the key of a sample is Threefry(seed; pixel, sample);
 * draw d of that sample is Threefry(sample key; d, 0).  Nothing is carried
 * from one sample to the next, so any lane can compute any sample. */
struct CounterRng {
    uint32_t k0, k1, draw;
    __device__ __forceinline__ void begin(uint32_t seed0, uint32_t seed1, uint32_t pixel, uint32_t sample)
    {
        threefry2x32_20(seed0, seed1, pixel, sample, k0, k1);
        draw = 0;
    }
    __device__ __forceinline__ void uniform2(float &u0, float &u1)
    {
        uint32_t a, b;
        threefry2x32_20(k0, k1, draw, 0u, a, b);
        draw++;
        u0 = bits_to_unit(a); u1 = bits_to_unit(b);
    }
};

/* The host program's own generator (src/math.c:158-192): 256 bits of state,
 * x0 = AESDEC(x0, x1); x1 = AESDEC(x0, x1); output lanes = x0 ^ x1.  AESDEC =
 * InvShiftRows, InvSubBytes, InvMixColumns, xor key, done with one 256-entry
 * table of combined InvSubBytes+InvMixColumns columns in LDS. */
struct AesRng {
    uint32_t a[4], b[4];
    const uint32_t *td;       /* LDS */
    __device__ __forceinline__ uint32_t col(const uint32_t s[4], int c) const
    {
        uint32_t w = td[s[c] & 0xFF];
        w ^= __builtin_rotateleft32(td[(s[(c + 3) & 3] >> 8) & 0xFF], 8);
        w ^= __builtin_rotateleft32(td[(s[(c + 2) & 3] >> 16) & 0xFF], 16);
        w ^= __builtin_rotateleft32(td[(s[(c + 1) & 3] >> 24) & 0xFF], 24);
        return w;
    }
    __device__ __forceinline__ void uniform2(float &u0, float &u1)
    {
        uint32_t n[4];
        n[0] = col(a, 0) ^ b[0]; n[1] = col(a, 1) ^ b[1]; n[2] = col(a, 2) ^ b[2]; n[3] = col(a, 3) ^ b[3];
        a[0] = n[0]; a[1] = n[1]; a[2] = n[2]; a[3] = n[3];
        n[0] = col(a, 0) ^ b[0]; n[1] = col(a, 1) ^ b[1]; n[2] = col(a, 2) ^ b[2]; n[3] = col(a, 3) ^ b[3];
        b[0] = n[0]; b[1] = n[1]; b[2] = n[2]; b[3] = n[3];
        u0 = bits_to_unit(a[0] ^ b[0]); u1 = bits_to_unit(a[1] ^ b[1]);
    }
};

/* ------------------------------------------------------------------ closest hit */

enum : int { HIT_NONE = -1, HIT_SPHERE = 0, HIT_PLANE = 1, HIT_LIGHT = 2, HIT_TRIANGLE = 3 };

struct Hit {
    int      type;
    uint32_t id, inst;
    float    t, beta, gamma;
    int      material;
    bool     entered;      /* the host's never-read enter_shape flag */
};

/* per-lane traversal stack in LDS: entry i of this lane at base[i * blockDim.x] */
struct Stack {
    uint32_t *base;
    uint32_t stride;
    __device__ __forceinline__ uint32_t &at(int i) const { return base[(uint32_t)i * stride]; }
};

/* slab test by division, src/main.c:379-386 / 404-411 */
__device__ __forceinline__ void slab(f3 lo, f3 hi, f3 o, f3 d, float &enter, float &leave)
{
    float ax = (lo.x - o.x) / d.x, ay = (lo.y - o.y) / d.y, az = (lo.z - o.z) / d.z;
    float bx = (hi.x - o.x) / d.x, by = (hi.y - o.y) / d.y, bz = (hi.z - o.z) / d.z;
    float nx = sel_min(ax, bx), ny = sel_min(ay, by), nz = sel_min(az, bz);
    float fx = sel_max(ax, bx), fy = sel_max(ay, by), fz = sel_max(az, bz);
    enter = sel_max(sel_max(nx, ny), nz);
    leave = sel_min(sel_min(fx, fy), fz);
}

/* ball test shared by lights and spheres, src/main.c:283-308 */
__device__ __forceinline__ bool ball(f3 center, float radius, f3 o, f3 d, float best, float &t_out, bool &entered)
{
    f3 oc = sub(o, center);
    float b = dot(d, oc);
    float c = len2(oc) - radius * radius;
    float det = b * b - c;
    if (det < 0.0f) return false;
    float root = __builtin_sqrtf(det);
    float t = -b - root;
    if (t < 1e-3f) t = -b + root;      /* the host compares with the double 1e-3: same set of floats */
    else entered = true;
    if (t > best) return false;
    if (t < 1e-3f) return false;
    t_out = t;
    return true;
}

/* SHADOW: only "is anything within tmax" is wanted, so the first accepted
 * primitive ends the query (the host runs the full query and reads .hit). */
template <bool SHADOW, bool STATS>
__device__ __forceinline__ void closest_hit(const rt_dev_scene &S, f3 o, f3 d, float tmax, Stack stk,
                                            Hit &h, Counters &cn)
{
    h.type = HIT_NONE; h.id = 0; h.inst = 0; h.t = tmax; h.beta = 0.0f; h.gamma = 0.0f;
    h.material = -1; h.entered = false;
    cn.rays++;
    if (STATS && SHADOW) cn.shadow_rays++;

    for (int i = 0; i < S.light_count; i++) {
        const rt_dev_light &l = S.lights[i];
        float t;
        if (STATS) cn.light_tests++;
        if (ball(ld3(l.c), l.radius, o, d, h.t, t, h.entered)) {
            h.type = HIT_LIGHT; h.id = (uint32_t)i; h.t = t;
            if (SHADOW) return;
        }
    }
    for (int i = 0; i < S.sphere_count; i++) {
        const rt_dev_sphere &s = S.spheres[i];
        float t;
        if (STATS) cn.sphere_tests++;
        if (ball(ld3(s.c), s.radius, o, d, h.t, t, h.entered)) {
            h.type = HIT_SPHERE; h.id = (uint32_t)i; h.t = t; h.material = s.material;
            if (SHADOW) return;
        }
    }
    for (int i = 0; i < S.plane_count; i++) {                        /* src/main.c:355-369 */
        const rt_dev_plane &p = S.planes[i];
        f3 n = ld3(p.n);
        if (STATS) cn.plane_tests++;
        float t = (p.d - dot(o, n)) / dot(d, n);
        if (t < 1e-3f) continue;
        if (t > h.t) continue;
        h.type = HIT_PLANE; h.id = (uint32_t)i; h.t = t; h.material = p.material;
        if (SHADOW) return;
    }

    for (int i = 0; i < S.instance_count; i++) {                     /* src/main.c:372-485 */
        const rt_dev_instance &in = S.instances[i];
        const rt_dev_mesh &m = S.meshes[in.mesh];
        /* ray into mesh space (src/math.c:299-311): row k = rotK . v */
        f3 rel = sub(o, ld3(in.pos));
        f3 rx = ld3(in.rx), ry = ld3(in.ry), rz = ld3(in.rz);
        f3 mo = mk(rx.x * rel.x + rx.y * rel.y + rx.z * rel.z,
                   ry.x * rel.x + ry.y * rel.y + ry.z * rel.z,
                   rz.x * rel.x + rz.y * rel.y + rz.z * rel.z);
        f3 mu = mk(rx.x * d.x + rx.y * d.y + rx.z * d.z,
                   ry.x * d.x + ry.y * d.y + ry.z * d.z,
                   rz.x * d.x + rz.y * d.y + rz.z * d.z);
        if (STATS) cn.root_tests++;
        {
            float enter, leave;
            slab(ld3(m.root_lo), ld3(m.root_hi), mo, mu, enter, leave);
            if (leave < enter) continue;            /* equal passes: non-strict at the root */
            if (enter > h.t) continue;
        }
        const rt_dev_node *nodes = S.nodes + m.node_base;
        const rt_dev_tri *tris = S.tris + m.tri_base;
        int top = 0;
        stk.at(0) = m.root_entry;
        while (top >= 0) {
            uint32_t e = stk.at(top--);
            if (!(e & RT_ENTRY_LEAF)) {
                const rt_dev_node &nd = nodes[e];
                const float4 *q = reinterpret_cast<const float4 *>(&nd);
                float4 q0 = q[0], q1 = q[1], q2 = q[2];
                uint4 q3 = *reinterpret_cast<const uint4 *>(q + 3);
                if (q3.z & RT_NODE_PSEUDO) {         /* chunks of an oversized leaf, in order, no box test */
                    stk.at(++top) = q3.y;
                    stk.at(++top) = q3.x;
                    continue;
                }
                if (STATS) cn.interior_pops++;
                float e0, l0, e1, l1;
                slab(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mo, mu, e0, l0);
                slab(mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), mo, mu, e1, l1);
                bool h0 = (e0 < l0) && (e0 < h.t);   /* strict: a flat box is never entered */
                bool h1 = (e1 < l1) && (e1 < h.t);
                if (h0 && h1) {                      /* farther child first, nearer on top; ties: child 0 on top */
                    bool swap = e0 > e1;
                    stk.at(++top) = swap ? q3.x : q3.y;
                    stk.at(++top) = swap ? q3.y : q3.x;
                } else if (h0) stk.at(++top) = q3.x;
                else if (h1) stk.at(++top) = q3.y;
            } else {                                 /* src/main.c:449-482 */
                if (STATS) cn.leaf_pops++;
                uint32_t first = e & RT_ENTRY_FIRST_MASK;
                uint32_t count = ((e >> RT_ENTRY_COUNT_SHIFT) & 15u) + 1u;
                for (uint32_t j = first; j < first + count; j++) {
                    if (STATS) cn.tri_tests++;
                    const float4 *q = reinterpret_cast<const float4 *>(tris + j);
                    float4 q0 = q[0], q1 = q[1], q2 = q[2];
                    f3 a = mk(q0.x, q0.y, q0.z), e1v = mk(q0.w, q1.x, q1.y);
                    f3 e2v = mk(q1.z, q1.w, q2.x), n = mk(q2.y, q2.z, q2.w);
                    f3 oa = sub(mo, a);
                    float den = dot(mu, n);
                    float t = -dot(oa, n) / den;
                    if (t < 1e-3f) continue;
                    if (t > h.t) continue;
                    f3 oau = cross(oa, mu);
                    float beta = -dot(e2v, oau) / den;
                    float gamma = dot(e1v, oau) / den;
                    if ((beta < 0) || (beta > 1)) continue;
                    if ((gamma < 0) || (gamma > 1)) continue;
                    if (beta + gamma > 1) continue;
                    h.type = HIT_TRIANGLE; h.t = t; h.id = j; h.inst = (uint32_t)i;
                    h.beta = beta; h.gamma = gamma; h.material = in.material;
                    if (SHADOW) return;
                }
            }
        }
    }
}

/* hit point, normal and uv of the winning primitive, src/main.c:487-535 */
struct Surface { f3 p, n; float u, v, alpha; };

template <bool STATS>
__device__ __forceinline__ void surface_of(const rt_dev_scene &S, f3 o, f3 d, const Hit &h, Surface &sf,
                                           Counters &cn)
{
    sf.p = mk(0, 0, 0); sf.n = mk(0, 0, 0); sf.u = 0; sf.v = 0; sf.alpha = 0;
    if (h.type == HIT_SPHERE) {
        sf.p = add(o, scale(d, h.t));
        sf.n = unit(sub(sf.p, ld3(S.spheres[h.id].c)));
    } else if (h.type == HIT_PLANE) {
        sf.p = add(o, scale(d, h.t));
        sf.n = ld3(S.planes[h.id].n);
    } else if (h.type == HIT_TRIANGLE) {
        if (STATS) cn.tri_hits++;
        const rt_dev_instance &in = S.instances[h.inst];
        const rt_dev_mesh &m = S.meshes[in.mesh];
        const float4 *q = reinterpret_cast<const float4 *>(S.attrs + m.tri_base + h.id);
        float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        sf.alpha = 1 - h.beta - h.gamma;
        f3 w = mk(sf.alpha, h.beta, h.gamma);
        f3 nm = blend(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), w);  /* not renormalised */
        sf.u = q2.y * w.x + q2.w * w.y + q3.y * w.z;
        sf.v = q2.z * w.x + q3.x * w.y + q3.z * w.z;
        sf.p = add(o, scale(d, h.t));
        sf.n = blend(ld3(in.rx), ld3(in.ry), ld3(in.rz), nm);
    }
}

/* tangent of a frame around n, src/main.c:561-576 = 702-717 */
__device__ __forceinline__ f3 frame_tangent(f3 n)
{
    f3 t = mk(0, 0, 0);
    float ax = __builtin_fabsf(n.x), ay = __builtin_fabsf(n.y), az = __builtin_fabsf(n.z);
    if ((ax >= ay && ax >= az) || (ay >= ax && ay >= az)) { t.x = n.y; t.y = -n.x; t.z = 0; }
    else if (az >= ax && az >= ay) { t.x = 0; t.y = -n.z; t.z = n.y; }
    return unit(t);
}

/* next-event estimation, src/main.c:546-609 */
template <class Rng, bool STATS>
__device__ __forceinline__ void direct_light(const rt_dev_scene &S, Rng &rng, Stack stk, const Surface &sf,
                                             f3 &sum, float &reached, Counters &cn)
{
    sum = mk(0, 0, 0);
    reached = 0;
    for (int i = 0; i < S.light_count; i++) {
        const rt_dev_light &l = S.lights[i];
        float r0, r1;
        rng.uniform2(r0, r1);                          /* drawn before any early-out */
        if (STATS) cn.rng_draws++;
        float sn, cs;
        rt_sincosf((float)(2 * RT_PI_D * (double)r0), &sn, &cs);
        float ring = __builtin_sqrtf(2 * r1 - r1 * r1);
        f3 hs = mk(cs * ring, sn * ring, 1 - r1);
        f3 lc = ld3(l.c);
        f3 axis = unit(sub(lc, sf.p));
        f3 t0 = frame_tangent(axis);
        f3 t1 = cross(axis, t0);
        f3 q = add(scale(axis, -hs.z), add(scale(t0, -hs.y), scale(t1, -hs.x)));
        q = scale(unit(q), l.radius);
        f3 target = add(q, lc);
        f3 path = sub(target, sf.p);
        f3 dir = unit(path);
        float cosine = dot(dir, sf.n);
        if (cosine < 0) continue;
        Hit blocker;
        closest_hit<true, STATS>(S, sf.p, dir, len(path), stk, blocker, cn);
        if (blocker.type != HIT_NONE) continue;
        float power = (float)((double)l.albedo / (4.0 * RT_PI_D * (double)len2(path)));
        sum = add(sum, scale(ld3(l.color), power * cosine));
        reached += 1;
    }
}

/* nearest texel, src/math.c:269-276 (double index arithmetic, truncation;
 * out-of-range coordinates, undefined for the host, are clamped) */
__device__ __forceinline__ f3 texel(const rt_dev_scene &S, int texture, float u, float v)
{
    const rt_dev_texture &t = S.textures[texture];
    double fu = (double)u * ((double)t.width - 0.5);
    double fv = (double)v * ((double)t.height - 0.5);
    int ui = !(fu >= 0.0) ? 0 : (fu >= (double)t.width ? (int)t.width - 1 : (int)fu);
    int vi = !(fv >= 0.0) ? 0 : (fv >= (double)t.height ? (int)t.height - 1 : (int)fv);
    const float *px = S.texels + 3 * ((size_t)t.base + (size_t)vi * t.width + (size_t)ui);
    return mk(px[0], px[1], px[2]);
}

/* one camera path, src/main.c:611-740 */
template <class Rng, bool STATS>
__device__ __forceinline__ f3 trace_path(const rt_dev_scene &S, Rng &rng, Stack stk, f3 o, f3 d,
                                         int max_segments, Counters &cn)
{
    float lights_reached = 0;
    f3 throughput = mk(1.0f, 1.0f, 1.0f);
    f3 radiance = mk(0.0f, 0.0f, 0.0f);
    bool inside_glass = false;

    for (int seg = 0; seg < max_segments; seg++) {
        Hit h;
        closest_hit<false, STATS>(S, o, d, 3.402823466e+38f, stk, h, cn);
        if (h.type == HIT_NONE) break;
        if (h.type == HIT_LIGHT) {
            lights_reached += 1;
            const rt_dev_light &l = S.lights[h.id];
            float s = (float)((double)l.albedo / (4.0 * RT_PI_D * (double)h.t * (double)h.t));
            radiance = add(radiance, mul(throughput, scale(ld3(l.color), s)));
            break;
        }
        Surface sf;
        surface_of<STATS>(S, o, d, h, sf, cn);
        const rt_dev_material &m = S.materials[h.material];
        if (m.kind == 1u) {                                   /* mirror, src/main.c:644-653 */
            throughput = mul(throughput, ld3(m.mirror_color));
            f3 along = scale(sf.n, dot(sf.n, d));
            f3 across = sub(d, along);
            o = sf.p;
            d = sub(across, scale(along, 2.0f));              /* unnormalised on purpose */
        } else if (m.kind == 2u) {                            /* glass, src/main.c:654-674 */
            f3 along = scale(sf.n, dot(sf.n, d));
            f3 across = sub(d, along);
            float ratio = inside_glass ? m.ior : (1.0f / m.ior);
            f3 bent = scale(across, ratio);
            float b2 = len2(bent);
            o = sf.p;
            if (b2 > 1) {
                d = sub(across, along);
            } else {
                inside_glass = !inside_glass;
                f3 forward = scale(unit(along), __builtin_sqrtf(1 - b2));
                d = add(forward, bent);
            }
        } else {                                              /* diffuse, src/main.c:675-732 */
            f3 direct; float reached;
            direct_light<Rng, STATS>(S, rng, stk, sf, direct, reached, cn);
            lights_reached += reached;
            f3 albedo = ld3(m.diffuse);
            if (m.texture >= 0) { albedo = texel(S, m.texture, sf.u, sf.v); if (STATS) cn.tex_lookups++; }
            radiance = add(radiance, mul(throughput, mul(direct, albedo)));
            throughput = mul(throughput, albedo);
            if (len2(throughput) <= 9.99999997475242707878e-7f) break;   /* < 1e-6 (double) on floats */
            o = sf.p;
            float r0, r1;
            rng.uniform2(r0, r1);
            if (STATS) cn.rng_draws++;
            float sn, cs;
            rt_sincosf((float)(2.0 * RT_PI_D * (double)r0), &sn, &cs);
            float ring = __builtin_sqrtf(1 - r1);
            f3 local = mk(cs * ring, sn * ring, __builtin_sqrtf(r1));
            f3 t0 = frame_tangent(sf.n);
            f3 t1 = cross(sf.n, t0);
            f3 nd = add(scale(sf.n, local.z), add(scale(t0, -local.y), scale(t1, -local.x)));
            nd = unit(nd);
            throughput = scale(throughput, dot(nd, sf.n));
            if (len2(throughput) <= 9.99999997475242707878e-7f) break;
            d = nd;
        }
    }
    if (lights_reached == 0) return mk(0.0f, 0.0f, 0.0f);
    return scale(radiance, 1 / lights_reached);
}

/* one sample of pixel (x, y), src/main.c:764-779; half_w/half_h = size/2.0f */
template <class Rng, bool STATS>
__device__ __forceinline__ f3 sample_pixel(const rt_dev_scene &S, Rng &rng, Stack stk, uint32_t x, uint32_t y,
                                           float half_w, float half_h, int max_segments, Counters &cn)
{
    float u0, u1;
    rng.uniform2(u0, u1);
    if (STATS) { cn.rng_draws++; cn.samples++; }
    float sn, cs;
    rt_sincosf((float)(2.0 * RT_PI_D * (double)u0), &sn, &cs);
    float radius = __builtin_sqrtf(-2 * rt_logf(u1));
    float jx = 0.0f + 0.25f * cs * radius;                    /* mean + sigma*cos*r, mean = 0 */
    float jy = 0.0f + 0.25f * sn * radius;
    f3 dr = scale(ld3(S.cam.right), (float)x - half_w + 0.5f + jx);
    f3 du = scale(ld3(S.cam.up), -((float)y - half_h + 0.5f + jy));
    f3 dm = scale(ld3(S.cam.forward), S.cam.focal);
    f3 dir = unit(add(add(dr, du), dm));
    f3 c = trace_path<Rng, STATS>(S, rng, stk, ld3(S.cam.pos), dir, max_segments, cn);
    float c2 = len2(c);
    if (c2 > 1e4f) c = scale(c, (float)(1e2 / (double)__builtin_sqrtf(c2)));    /* firefly clamp */
    return c;
}

} // namespace rt
