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

/* RT_DOUBLE=n (timing experiment, same pictures): one part of a shade step is computed twice, the second time on
 * operands the compiler cannot tell are the same; the extra time of a launch is what that part costs where it runs.
 * 3 sin/cos, 4 lights + spheres + planes, 6 instance root boxes (of the lanes' own steps; a new path's first event is not doubled) */
#ifndef RT_DOUBLE
#define RT_DOUBLE 0
#endif
#define RT_OPAQUE(x) __asm__ volatile("" : "+v"(x))
/* RT_SHADE_CLOCKS (timing experiment): one wave of a shade launch prints the clock counts of the sections of its loop */
#if defined(RT_SHADE_CLOCKS)
#define RT_STICK(t, i) { const unsigned long long now_ = __builtin_readcyclecounter(); (t)[i] += now_ - (t)[15]; (t)[15] = now_; }
#else
#define RT_STICK(t, i)
#endif
namespace rt {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

/* Small scene tables (lights, spheres, planes, instances, meshes) never change while a kernel
 * runs.  Reading them through a constant-address-space pointer with a wave-uniform index lets the
 * compiler fetch them with scalar loads into SGPRs; through a plain global pointer it must assume
 * the kernel's own stores could alias them and issues a vector load per lane. */
#define RT_CMEM __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const RT_CMEM T *cmem(const T *p) { return (const RT_CMEM T *)(p); }
__device__ __forceinline__ f3 ld3(const RT_CMEM float *p) { return mk(p[0], p[1], p[2]); }
/* dot: ((0 + x) + y) + z like src/math.c:3-12 (the leading +0 turns a -0 product into +0) */
__device__ __forceinline__ float dot(f3 a, f3 b)
{
    float r = 0.0f;
    r += a.x * b.x; r += a.y * b.y; r += a.z * b.z;
    return r;
}
__device__ __forceinline__ float len2(f3 a) { return dot(a, a); }

/* ---- correctly rounded square root and the three quotients of a normalisation, without the parts of
 * hipcc's expansions that only serve operands of extreme magnitude --------------------------------
 *
 * hipcc expands sqrtf(x) into: scale x by 2^32 when x < 2^-96; s = v_sqrt_f32(x) (within one unit in the
 * last place); the two neighbours of s tested with one fma each (s - 1ulp if fma(-(s - 1ulp), s, x) <= 0,
 * s + 1ulp if fma(-(s + 1ulp), s, x) > 0); unscale; x itself for zero and infinity.  The middle part alone
 * returns the same value whenever no scaling takes place, zero, infinity and NaN included (the neighbours'
 * residuals are then NaN or zero and select nothing): sixteen instructions become nine.  RT_LEAN_MATH=0
 * keeps the compiler's expansions everywhere. */
#ifndef RT_LEAN_MATH
#define RT_LEAN_MATH 1
#endif
/* precondition: x is +0, +infinity, NaN, or a positive number of at least 2^-96 */
__device__ __forceinline__ float sqrt_unscaled(float x)
{
#if RT_LEAN_MATH
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = __builtin_fmaf(-below, s, x), r_above = __builtin_fmaf(-above, s, x);
    float r = (0.0f >= r_below) ? below : s;
    r = (0.0f < r_above) ? above : r;
    return r;
#else
    return __builtin_sqrtf(x);
#endif
}
/* any operand: the short form for positive numbers from 2^-96 up to infinity (one integer comparison of the
 * bits); zeros, smaller and negative numbers, NaNs take the compiler's expansion (v_sqrt_f32 flushes
 * subnormal operands, negative ones to -0) */
__device__ __forceinline__ float sqrt_rn(float x)
{
#if RT_LEAN_MATH
    if (__float_as_uint(x) - 0x0F800000u <= 0x7F800000u - 0x0F800000u) return sqrt_unscaled(x);
#endif
    return __builtin_sqrtf(x);
}
__device__ __forceinline__ float len(f3 a) { return sqrt_rn(len2(a)); }

__device__ __forceinline__ float refined_rcp(float b);
__device__ __forceinline__ float div_by_cached(float a, float b, float y1);
/* a / |a| as the host computes it (src/math.c:29-40: one square root, three divisions by it).  The three
 * divisions share their divisor, so the part of the division's expansion that depends on the divisor only is
 * done once (div_by_cached below: the same operations as hipcc's expansion, hence the same bits) whenever the
 * expansion's scaling and special cases cannot apply: squared length in [2^-60, 2^60] (then the length is a
 * normal number in [2^-30, 2^30]) and every component zero or at least 2^-63 in magnitude (v_frexp_exp gives 0
 * for a zero, exponent + 1 otherwise), so every quotient is zero or at least 2^-93.  A zero component keeps its
 * sign as in the host's quotient by a positive number (the shared form alone could turn -0 into +0).
 * 42 instructions with one reciprocal instead of 54 with three. */
__device__ __forceinline__ f3 unit(f3 a, float &length)
{
    const float n2 = len2(a);
#if RT_LEAN_MATH
    const int ex = __builtin_amdgcn_frexp_expf(a.x), ey = __builtin_amdgcn_frexp_expf(a.y), ez = __builtin_amdgcn_frexp_expf(a.z);
    const int emin = ex < ey ? (ex < ez ? ex : ez) : (ey < ez ? ey : ez);
    if (__float_as_uint(n2) - 0x21800000u <= 0x5D800000u - 0x21800000u && emin >= -62) {
        const float n = sqrt_unscaled(n2);
        const float y = refined_rcp(n);
        length = n;
        return mk(__builtin_copysignf(div_by_cached(a.x, n, y), a.x), __builtin_copysignf(div_by_cached(a.y, n, y), a.y),
                  __builtin_copysignf(div_by_cached(a.z, n, y), a.z));
    }
#endif
    const float n = __builtin_sqrtf(n2);
    length = n;
    return mk(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ f3 unit(f3 a) { float n; return unit(a, n); }
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
             tex_lookups, rng_draws, light_tests, sphere_tests, plane_tests,
             shade_steps, pop_steps, tri_steps, shade_lanes, pop_lanes, tri_lanes,   /* scheduler: wave steps (lane 0) and lanes taking part */
             tree_entries;                                                          /* instance trees entered */
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
#if defined(RT_ABLATE_RNG)       /* timing experiment only: wrong bits */
    x0 += x1; x1 = __builtin_rotateleft32(x1, 13); x1 ^= x0; x0 += k2; x0 ^= x0 >> 15; x0 *= 0x2c1b3c6du; x1 += x0; x1 ^= x1 >> 13;
    o0 = x0; o1 = x1; return;
#endif
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

/* Counter generator.  The key of a sample is K = Threefry(seed; pixel, sample).
 * Draw 0 of the sample IS K (its two words), draw d >= 1 is Threefry(K; d, 0).
 * Nothing is carried from one sample to the next, so any lane can compute any
 * sample; and every draw, the first included, costs exactly one Threefry call
 * (no divergent "new sample" branch inside a wave). */
struct CounterRng {
    uint32_t seed0, seed1;        /* wave-uniform */
    uint32_t k0, k1, c0, c1;
    bool fresh;
    __device__ __forceinline__ void begin(uint32_t pixel, uint32_t sample)
    {
        c0 = pixel; c1 = sample; fresh = true;
    }
    __device__ __forceinline__ void uniform2(float &u0, float &u1)
    {
        uint32_t a, b;
        threefry2x32_20(fresh ? seed0 : k0, fresh ? seed1 : k1, c0, c1, a, b);
        if (fresh) { k0 = a; k1 = b; c0 = 1u; c1 = 0u; fresh = false; }
        else c0++;
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
#if defined(RT_STACK_CLAMP)      /* timing experiment only: a shorter stack, deeper entries overwrite the last one (wrong pictures) */
    __device__ __forceinline__ uint32_t &at(int i) const { return base[(uint32_t)(i < RT_STACK_CLAMP ? i : RT_STACK_CLAMP - 1) * stride]; }
#else
    __device__ __forceinline__ uint32_t &at(int i) const { return base[(uint32_t)i * stride]; }
#endif
    __device__ __forceinline__ void put(int i, uint32_t v) const { at(i) = v; }
    __device__ __forceinline__ uint32_t get(int i) const { return at(i); }
};

/* The tree kernel's stack: the first `levels` entries of a lane in LDS as above, deeper ones in a slice of device
 * memory of the lane's own (entry i at spill[(i - levels) * spill_stride], lanes side by side).  The stack of the
 * walk holds only the far children of nodes where both children were entered (the near child is visited at once),
 * so it is nearly always a few entries deep although the tree has two dozen levels: LDS is paid for the common depth,
 * the rare deep walk costs an L2 access per entry, and the order of the entries is the stack's own either way. */
struct TreeStack {
    uint32_t *base;              /* LDS, this lane's entry 0 */
    uint32_t stride;
    uint32_t *spill;             /* wave-uniform: entry 0 of the launch's first thread; the lane's own offset is added at each use,
                                  * so the pointer lives in scalar registers and the lane keeps one 32-bit index */
    uint32_t spill_stride;       /* threads of the launch */
    uint32_t thread;             /* this lane's index among them */
    int levels;
    __device__ __forceinline__ void put(int i, uint32_t v) const
    {
        if (i < levels) base[(uint32_t)i * stride] = v;
        else spill[(uint32_t)(i - levels) * spill_stride + thread] = v;
    }
    __device__ __forceinline__ uint32_t get(int i) const
    {
        if (i < levels) return base[(uint32_t)i * stride];
        return spill[(uint32_t)(i - levels) * spill_stride + thread];
    }
};

/* number of set bits of a wave mask below this lane (what a running count over the lanes needs): two instructions,
 * no per-lane mask to keep in registers */
__device__ __forceinline__ uint32_t bits_below_lane(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

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

/* ---- the same slab test with the divisor's reciprocal computed once per ray ----
 *
 * hipcc expands the IEEE quotient a/b (no scaling needed, see below) into
 *   y0 = rcp(b); e = fma(-b,y0,1); y1 = fma(e,y0,y0);                 <- depends on b only
 *   q0 = a*y1; r0 = fma(-b,q0,a); q1 = fma(r0,y1,q0); r1 = fma(-b,q1,a); q = fma(r1,y1,q1)
 * A tree walk divides by the same three direction components at every node,
 * so y1 is kept per ray and each quotient costs the second line only: the very
 * same operations, hence the very same bits.  The expansion's pre-scaling
 * (v_div_scale) and special-case fix-up (v_div_fixup) do nothing when b is
 * normal with |b| in [2^-40, 2^40] and a is 0 or has |a| in [2^-63, 2^41]
 * (then |a/b| lies in [2^-103, 2^81]); rays or meshes outside that range take
 * the plain division.  The one difference, the sign of a zero quotient, cannot
 * change a comparison, and comparisons are all a slab test is used for. */
__device__ __forceinline__ float refined_rcp(float b)
{
    float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_by_cached(float a, float b, float y1)
{
    float q0 = a * y1;
    float r0 = __builtin_fmaf(-b, q0, a);
    float q1 = __builtin_fmaf(r0, y1, q0);
    float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}
__device__ __forceinline__ bool tame_or_zero(float v)
{
    float a = __builtin_fabsf(v);
    return v == 0.0f || (a >= 0x1p-40f && a <= 0x1p40f);
}
__device__ __forceinline__ bool tame(float v)
{
    float a = __builtin_fabsf(v);
    return a >= 0x1p-40f && a <= 0x1p40f;
}
__device__ __forceinline__ void slab_cached(f3 lo, f3 hi, f3 o, f3 d, f3 y, float &enter, float &leave)
{
    float ax = div_by_cached(lo.x - o.x, d.x, y.x), ay = div_by_cached(lo.y - o.y, d.y, y.y),
          az = div_by_cached(lo.z - o.z, d.z, y.z);
    float bx = div_by_cached(hi.x - o.x, d.x, y.x), by = div_by_cached(hi.y - o.y, d.y, y.y),
          bz = div_by_cached(hi.z - o.z, d.z, y.z);
    /* every quotient is finite here (the guard excludes zero divisors and non-finite operands), so the
     * hardware minimum / maximum give what the host's select macros give, up to the sign of a zero,
     * which no later comparison can see: one instruction instead of compare + select */
    float nx = __builtin_fminf(ax, bx), ny = __builtin_fminf(ay, by), nz = __builtin_fminf(az, bz);
    float fx = __builtin_fmaxf(ax, bx), fy = __builtin_fmaxf(ay, by), fz = __builtin_fmaxf(az, bz);
    enter = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    leave = __builtin_fminf(__builtin_fminf(fx, fy), fz);
}

/* ---- a box test decided from approximate quotients -------------------------------------
 *
 * A slab test is only ever used for comparisons (does the ray enter the box before it leaves it, and before
 * the closest hit so far), and a correctly rounded division costs eleven instructions where a reciprocal and a
 * multiply cost two.  The approximate quotient a * rcp(b) lies within 2.5 units in the last place of the host's
 * a / b, so the approximate entry and exit distances lie within that of the host's; whenever the two sides of a
 * comparison are further apart than RT_SLAB_MARGIN times their magnitudes (a margin several times the error),
 * the host's comparison has the same outcome.  Otherwise, and for rays or boxes outside the magnitudes the
 * bound is proven for (direction components in [2^-40, 2^40], numerators up to 2^40, distances that are zero or at
 * least 2^-100: below that a quotient may be subnormal and its error is absolute, not relative), the verdict is
 * "undecided" and the caller makes the host's test by division (a fraction of a percent of the waves). */
#ifndef RT_SLAB_APPROX
#define RT_SLAB_APPROX 1              /* 0: root boxes are always tested by division in the shade kernels */
#endif
#define RT_SLAB_MARGIN 4e-6f           /* relative; the error of an approximate distance is below 3e-7 */
enum : int { BOX_MISSED = 0, BOX_ENTERED = 1, BOX_UNDECIDED = 2 };
__device__ __forceinline__ bool clearly_apart(float x, float y)
{
    return __builtin_fabsf(x - y) > __builtin_fmaf(RT_SLAB_MARGIN, __builtin_fabsf(x), RT_SLAB_MARGIN * __builtin_fabsf(y));
}
/* the root test of src/main.c:379-391: entered unless leave < enter or enter > closest */
__device__ __forceinline__ int root_box_verdict(f3 lo, f3 hi, f3 o, f3 d, float closest)
{
    const float ax0 = lo.x - o.x, ay0 = lo.y - o.y, az0 = lo.z - o.z, bx0 = hi.x - o.x, by0 = hi.y - o.y, bz0 = hi.z - o.z;
    const float dmin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float dmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ax0), __builtin_fabsf(ay0)), __builtin_fabsf(az0)),
                                       __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(bx0), __builtin_fabsf(by0)), __builtin_fabsf(bz0)));
    /* direction components and numerators of tame magnitude: reciprocals and products are finite, normal or zero */
    const bool tame_ray = dmin >= 0x1p-40f && dmax <= 0x1p40f && amax <= 0x1p40f;
    const float rx = __builtin_amdgcn_rcpf(d.x), ry = __builtin_amdgcn_rcpf(d.y), rz = __builtin_amdgcn_rcpf(d.z);
    const float ax = ax0 * rx, ay = ay0 * ry, az = az0 * rz, bx = bx0 * rx, by = by0 * ry, bz = bz0 * rz;
    const float enter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float leave = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    /* the relative error bound of an approximate distance holds for normal numbers: a distance that is neither zero nor
     * at least 2^-100 in magnitude (a product that lands among the subnormals has an absolute, not a relative, error)
     * leaves the verdict to the division */
    const float tiny = 0x1p-100f;
    const bool sound = (enter == 0.0f || __builtin_fabsf(enter) >= tiny) && (leave == 0.0f || __builtin_fabsf(leave) >= tiny) &&
                       (closest == 0.0f || __builtin_fabsf(closest) >= tiny);
    if (!tame_ray || !sound || !clearly_apart(enter, leave) || !clearly_apart(enter, closest)) return BOX_UNDECIDED;
    return (leave < enter || enter > closest) ? BOX_MISSED : BOX_ENTERED;
}

/* ball test shared by lights and spheres, src/main.c:283-308 */
__device__ __forceinline__ bool ball(f3 center, float radius, f3 o, f3 d, float best, float &t_out, bool &entered)
{
    /* the early returns stay branches: whole waves miss a ball and skip its square root (a straight-line
     * form with masks was measured 2 % slower on the launch that starts the paths) */
    f3 oc = sub(o, center);
    float b = dot(d, oc);
    float c = len2(oc) - radius * radius;
    float det = b * b - c;
    if (det < 0.0f) return false;
    float root = sqrt_rn(det);
    float t = -b - root;
    if (t < 1e-3f) t = -b + root;      /* the host compares with the double 1e-3: same set of floats */
    else entered = true;
    if (t > best) return false;
    if (t < 1e-3f) return false;
    t_out = t;
    return true;
}

/* ---- a closest-hit query in three resumable pieces ------------------------
 *
 * find_ray_hit (src/main.c:264-485) tests lights, spheres, planes, then walks
 * each mesh instance's tree.  The tree walk is the only part whose length
 * differs wildly between the lanes of a wave, so it is cut into micro-steps
 * (one node pop or one triangle test) that a lane can be parked in between:
 *   query_begin   lights, spheres, planes                    (:280-369)
 *   query_scan    next instance whose root box the ray enters (:372-395)
 *   query_step    one pop / one triangle of that instance     (:397-483)
 * Per ray the sequence of operations is the host's; any_hit (shadow rays: the
 * host only reads .hit of those) ends the query at the first accepted
 * primitive.
 */
struct Trav {
    f3 mo, mu;                   /* ray in mesh space */
    uint32_t node_base, tri_base;
    uint32_t tri_cur, tri_end;   /* triangles of the current leaf still to test */
    int top;                     /* stack pointer, -1 = empty */
    int inst;                    /* instance being walked / next to scan */
    int material;
    uint32_t mesh;               /* index of the instance's mesh (tree kernel: which staged tree top is its own) */
    f3 inv;                      /* refined reciprocals of mu (only if cached) */
    bool cached;                 /* the shared-reciprocal slab test is exact for this ray and mesh */
};

#ifndef RT_SHADOW_EARLY_OUT
#define RT_SHADOW_EARLY_OUT 0         /* 1: a shadow ray that has found its occluder leaves the primitive loops at once, as the counting
                                       * builds do (they must count the host's tests).  A lane that leaves early only idles while the
                                       * rest of its wave goes on, and the branches cost more than they save: without them the launch
                                       * that starts the paths is 7 % faster; the answer (any hit at all) is the same */
#endif
#define RT_EARLY_OUT(stats) (RT_SHADOW_EARLY_OUT || (stats))
template <bool STATS>
__device__ __forceinline__ bool query_begin(const rt_dev_scene &S, f3 o, f3 d, float tmax, bool any_hit,
                                            Hit &h, Counters &cn)
{
    h.type = HIT_NONE; h.id = 0; h.inst = 0; h.t = tmax; h.beta = 0.0f; h.gamma = 0.0f;
    h.material = -1; h.entered = false;
    cn.rays++;
    if (STATS && any_hit) cn.shadow_rays++;

    const RT_CMEM rt_dev_light *lights = cmem(S.lights);
    const RT_CMEM rt_dev_sphere *spheres = cmem(S.spheres);
    const RT_CMEM rt_dev_plane *planes = cmem(S.planes);
    /* each table entry is fetched (scalar loads) one iteration ahead of its use, so the fetch of
     * entry i+1 overlaps the arithmetic on entry i */
    if (S.light_count > 0) {
        f3 c = ld3(lights[0].c); float radius = lights[0].radius;
        for (int i = 0; i < S.light_count; i++) {
            const int nx = i + 1 < S.light_count ? i + 1 : i;
            const f3 c_next = ld3(lights[nx].c); const float radius_next = lights[nx].radius;
            float t;
            if (STATS) cn.light_tests++;
            if (ball(c, radius, o, d, h.t, t, h.entered)) {
                h.type = HIT_LIGHT; h.id = (uint32_t)i; h.t = t;
                if (RT_EARLY_OUT(STATS) && any_hit) return true;
            }
            c = c_next; radius = radius_next;
        }
    }
    if (S.sphere_count > 0) {
        f3 c = ld3(spheres[0].c); float radius = spheres[0].radius; int material = spheres[0].material;
        for (int i = 0; i < S.sphere_count; i++) {
            const int nx = i + 1 < S.sphere_count ? i + 1 : i;
            const f3 c_next = ld3(spheres[nx].c); const float radius_next = spheres[nx].radius;
            const int material_next = spheres[nx].material;
            float t;
            if (STATS) cn.sphere_tests++;
            if (ball(c, radius, o, d, h.t, t, h.entered)) {
                h.type = HIT_SPHERE; h.id = (uint32_t)i; h.t = t; h.material = material;
                if (RT_EARLY_OUT(STATS) && any_hit) return true;
            }
            c = c_next; radius = radius_next; material = material_next;
        }
    }
    if (S.plane_count > 0) {                                         /* src/main.c:355-369 */
        f3 n = ld3(planes[0].n); float pd = planes[0].d; int material = planes[0].material;
        for (int i = 0; i < S.plane_count; i++) {
            const int nx = i + 1 < S.plane_count ? i + 1 : i;
            const f3 n_next = ld3(planes[nx].n); const float pd_next = planes[nx].d;
            const int material_next = planes[nx].material;
            if (STATS) cn.plane_tests++;
            const float t = (pd - dot(o, n)) / dot(d, n);
            if (!(t < 1e-3f) && !(t > h.t)) {
                h.type = HIT_PLANE; h.id = (uint32_t)i; h.t = t; h.material = material;
                if (RT_EARLY_OUT(STATS) && any_hit) return true;
            }
            n = n_next; pd = pd_next; material = material_next;
        }
    }
    if (!RT_EARLY_OUT(STATS) && any_hit && h.type != HIT_NONE) return true;
    return S.instance_count == 0;
}

/* from instance tv.inst on: true = tv is set up to walk an instance, false = no instance left.
 * CACHE: prepare the per-ray reciprocals for slab_cached when ray and mesh allow it. */
template <bool STATS, bool CACHE = false, class StackT = Stack>
__device__ __forceinline__ bool query_scan(const rt_dev_scene &S, f3 o, f3 d, const Hit &h, Trav &tv,
                                           StackT stk, Counters &cn)
{
    for (; tv.inst < S.instance_count; tv.inst++) {
        const rt_dev_instance &in = S.instances[tv.inst];
        const rt_dev_mesh &m = S.meshes[in.mesh];
        /* ray into mesh space (src/math.c:299-311): row k = rotK . v */
        f3 rel = sub(o, ld3(in.pos));
        f3 rx = ld3(in.rx), ry = ld3(in.ry), rz = ld3(in.rz);
        tv.mo = mk(rx.x * rel.x + rx.y * rel.y + rx.z * rel.z,
                   ry.x * rel.x + ry.y * rel.y + ry.z * rel.z,
                   rz.x * rel.x + rz.y * rel.y + rz.z * rel.z);
        tv.mu = mk(rx.x * d.x + rx.y * d.y + rx.z * d.z,
                   ry.x * d.x + ry.y * d.y + ry.z * d.z,
                   rz.x * d.x + rz.y * d.y + rz.z * d.z);
        if (STATS) cn.root_tests++;
        float enter, leave;
        slab(ld3(m.root_lo), ld3(m.root_hi), tv.mo, tv.mu, enter, leave);
        if (leave < enter) continue;                /* equal passes: non-strict at the root */
        if (enter > h.t) continue;
        tv.node_base = m.node_base; tv.tri_base = m.tri_base; tv.material = in.material;
        tv.top = 0; tv.tri_cur = 0; tv.tri_end = 0;
        tv.mesh = in.mesh;
        tv.cached = false;
        if (CACHE) {
            tv.cached = m.tame_boxes != 0u && tame(tv.mu.x) && tame(tv.mu.y) && tame(tv.mu.z) &&
                        tame_or_zero(tv.mo.x) && tame_or_zero(tv.mo.y) && tame_or_zero(tv.mo.z);
            tv.inv = mk(refined_rcp(tv.mu.x), refined_rcp(tv.mu.y), refined_rcp(tv.mu.z));
        }
        stk.put(0, m.root_entry);
        if (STATS) cn.tree_entries++;
        return true;
    }
    return false;
}

#ifndef RT_PROBE_ROOT_CHILDREN
#define RT_PROBE_ROOT_CHILDREN 0      /* 1: the shade kernels test the two boxes under an instance's root before parking a ray for it
                                       * (experiment: 11 % fewer parked rays in the dog scene, but the test costs the new-path launch more
                                       * than the tree kernel saves) */
#endif
/* first instance from `first` on whose root box the ray enters (the test of
 * query_scan without touching a stack), or -1 */
template <bool STATS>
__device__ __forceinline__ int query_probe(const rt_dev_scene &S, f3 o, f3 d, const Hit &h, int first, Counters &cn)
{
    const RT_CMEM rt_dev_instance *instances = cmem(S.instances);
    const RT_CMEM rt_dev_mesh *meshes = cmem(S.meshes);
    for (int i = first; i < S.instance_count; i++) {
        const RT_CMEM rt_dev_instance &in = instances[i];
        const RT_CMEM rt_dev_mesh &m = meshes[in.mesh];
        f3 rel = sub(o, ld3(in.pos));
        f3 rx = ld3(in.rx), ry = ld3(in.ry), rz = ld3(in.rz);
        f3 mo = mk(rx.x * rel.x + rx.y * rel.y + rx.z * rel.z,
                   ry.x * rel.x + ry.y * rel.y + ry.z * rel.z,
                   rz.x * rel.x + rz.y * rel.y + rz.z * rel.z);
        f3 mu = mk(rx.x * d.x + rx.y * d.y + rx.z * d.z,
                   ry.x * d.x + ry.y * d.y + ry.z * d.z,
                   rz.x * d.x + rz.y * d.y + rz.z * d.z);
        if (STATS) cn.root_tests++;
#if RT_SLAB_APPROX
        const int verdict = root_box_verdict(ld3(m.root_lo), ld3(m.root_hi), mo, mu, h.t);
        bool entered = verdict == BOX_ENTERED;
        if (__ballot(verdict == BOX_UNDECIDED)) {            /* some lane is too close to call: the host's test by division */
            if (verdict == BOX_UNDECIDED) {
                float enter, leave;
                slab(ld3(m.root_lo), ld3(m.root_hi), mo, mu, enter, leave);
                entered = !(leave < enter) && !(enter > h.t);
            }
        }
        if (!entered) continue;
#else
        float enter, leave;
        slab(ld3(m.root_lo), ld3(m.root_hi), mo, mu, enter, leave);
        if (leave < enter) continue;
        if (enter > h.t) continue;
#endif
#if RT_PROBE_ROOT_CHILDREN
        /* The walk of this instance would begin by testing the two boxes under the root (src/main.c:402-447).  A ray
         * that enters neither has nothing to find in the instance: that first step is made here (the root's record
         * comes through scalar loads: one address for the whole wave), and the ray is spared the trip through the
         * parked list and the tree kernel.  The boxes are tested by division as the host does; the tree kernel's
         * shared-reciprocal test decides the same. */
        if (!(m.root_entry & RT_ENTRY_LEAF)) {
            const RT_CMEM rt_dev_node &nd = cmem(S.nodes)[m.node_base + m.root_entry];
            if (!(nd.flags & RT_NODE_PSEUDO)) {
                float n0, l0, n1, l1;
                slab(ld3(nd.lo0), ld3(nd.hi0), mo, mu, n0, l0);
                slab(ld3(nd.lo1), ld3(nd.hi1), mo, mu, n1, l1);
                const bool h0 = (n0 < l0) && (n0 < h.t), h1 = (n1 < l1) && (n1 < h.t);
                if (!h0 && !h1) {
                    if (STATS) { cn.tree_entries++; cn.interior_pops++; }
                    continue;
                }
            }
        }
#endif
        return i;
    }
    return -1;
}

enum : int { STEP_MORE = 0, STEP_INSTANCE_DONE = 1, STEP_QUERY_DONE = 2 };

/* one triangle of the current leaf: src/main.c:449-482 (precondition tv.tri_cur < tv.tri_end) */
template <bool STATS>
__device__ __forceinline__ int query_tri_step(const rt_dev_scene &S, bool any_hit, Hit &h, Trav &tv, Counters &cn)
{
    const uint32_t j = tv.tri_cur++;
    if (STATS) cn.tri_tests++;
    const float4 *q = reinterpret_cast<const float4 *>(S.tris + tv.tri_base + j);
    float4 q0 = q[0], q1 = q[1], q2 = q[2];
    f3 a = mk(q0.x, q0.y, q0.z), e1v = mk(q0.w, q1.x, q1.y);
    f3 e2v = mk(q1.z, q1.w, q2.x), n = mk(q2.y, q2.z, q2.w);
    f3 oa = sub(tv.mo, a);
    float den = dot(tv.mu, n);
    float t = -dot(oa, n) / den;
    if (t < 1e-3f) return STEP_MORE;
    if (t > h.t) return STEP_MORE;
    f3 oau = cross(oa, tv.mu);
    float beta = -dot(e2v, oau) / den;
    float gamma = dot(e1v, oau) / den;
    if ((beta < 0) || (beta > 1)) return STEP_MORE;
    if ((gamma < 0) || (gamma > 1)) return STEP_MORE;
    if (beta + gamma > 1) return STEP_MORE;
    h.type = HIT_TRIANGLE; h.t = t; h.id = j; h.inst = (uint32_t)tv.inst;
    h.beta = beta; h.gamma = gamma; h.material = tv.material;
    return any_hit ? STEP_QUERY_DONE : STEP_MORE;
}

/* one stack pop: src/main.c:397-447 (precondition: no triangle pending) */
template <bool STATS, bool CACHE = false>
__device__ __forceinline__ int query_pop_step(const rt_dev_scene &S, Hit &h, Trav &tv, Stack stk, Counters &cn)
{
    if (tv.top < 0) { tv.inst++; return STEP_INSTANCE_DONE; }
    const uint32_t e = stk.get(tv.top--);
    if (e & RT_ENTRY_LEAF) {
        if (STATS) cn.leaf_pops++;
        tv.tri_cur = e & RT_ENTRY_FIRST_MASK;
        tv.tri_end = tv.tri_cur + ((e >> RT_ENTRY_COUNT_SHIFT) & 15u) + 1u;
        return STEP_MORE;
    }
    const float4 *q = reinterpret_cast<const float4 *>(S.nodes + tv.node_base + e);
    float4 q0 = q[0], q1 = q[1], q2 = q[2];
    uint4 q3 = *reinterpret_cast<const uint4 *>(q + 3);
    if (q3.z & RT_NODE_PSEUDO) {                 /* chunks of an oversized leaf, in order, no box test */
        stk.put(++tv.top, q3.y);
        stk.put(++tv.top, q3.x);
        return STEP_MORE;
    }
    if (STATS) cn.interior_pops++;
    float e0, l0, e1, l1;
    if (CACHE && tv.cached) {
        slab_cached(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), tv.mo, tv.mu, tv.inv, e0, l0);
        slab_cached(mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), tv.mo, tv.mu, tv.inv, e1, l1);
    } else {
        slab(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), tv.mo, tv.mu, e0, l0);
        slab(mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), tv.mo, tv.mu, e1, l1);
    }
    bool h0 = (e0 < l0) && (e0 < h.t);           /* strict: a flat box is never entered */
    bool h1 = (e1 < l1) && (e1 < h.t);
    if (h0 && h1) {                              /* farther child first, nearer on top; ties: child 0 on top */
        bool swap = e0 > e1;
        stk.put(++tv.top, swap ? q3.x : q3.y);
        stk.put(++tv.top, swap ? q3.y : q3.x);
    } else if (h0) stk.put(++tv.top, q3.x);
    else if (h1) stk.put(++tv.top, q3.y);
    return STEP_MORE;
}

/* ---- the same walk with the next node fetched one step ahead (tree kernel) -------------
 *
 * A step of the walk above starts with a dependent chain: stack slot (LDS) -> record address ->
 * 64 bytes from L2 -> arithmetic.  Here the entry that will be processed next lives in a
 * register (`cur`, the would-be top of the stack); after every step walk_advance() decides what
 * the lane does next: for an interior node it returns the record's address and the caller
 * requests the 64 bytes at once, at ONE place in its loop, so the latency is covered by whatever
 * the wave does until the lane's next step; a leaf entry is opened on the spot and waits for the
 * wave's next leaf pass.  Per ray the sequence of box tests and the hit record they see are
 * unchanged. */
#define RT_ENTRY_NONE 0xFFFFFFFFu     /* never a valid entry: a leaf's first + count stays below 2^27 */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

/* the fetched interior node: src/main.c:397-447; leaves the entry to visit next in cur (or none) */
template <bool STATS, class StackT = Stack>
__device__ __forceinline__ void walk_node(u32x4 fa, u32x4 fb, u32x4 fc, u32x4 fd, const Hit &h, Trav &tv, StackT stk,
                                          uint32_t &cur, Counters &cn)
{
    const uint32_t e0 = fd.x, e1 = fd.y;
    if (fd.z & RT_NODE_PSEUDO) {                         /* chunks of an oversized leaf, in order, no box test */
        stk.put(++tv.top, e1);
        cur = e0;
        return;
    }
    if (STATS) cn.interior_pops++;
    const f3 lo0 = mk(__uint_as_float(fa.x), __uint_as_float(fa.y), __uint_as_float(fa.z));
    const f3 hi0 = mk(__uint_as_float(fa.w), __uint_as_float(fb.x), __uint_as_float(fb.y));
    const f3 lo1 = mk(__uint_as_float(fb.z), __uint_as_float(fb.w), __uint_as_float(fc.x));
    const f3 hi1 = mk(__uint_as_float(fc.y), __uint_as_float(fc.z), __uint_as_float(fc.w));
    float n0, l0, n1, l1;
    if (tv.cached) {
        slab_cached(lo0, hi0, tv.mo, tv.mu, tv.inv, n0, l0);
        slab_cached(lo1, hi1, tv.mo, tv.mu, tv.inv, n1, l1);
    } else {
        slab(lo0, hi0, tv.mo, tv.mu, n0, l0);
        slab(lo1, hi1, tv.mo, tv.mu, n1, l1);
    }
    const bool h0 = (n0 < l0) && (n0 < h.t);             /* strict: a flat box is never entered */
    const bool h1 = (n1 < l1) && (n1 < h.t);
    cur = RT_ENTRY_NONE;
    if (h0 && h1) {                                      /* nearer child next, farther one stacked; ties: child 0 first */
        const bool swap = n0 > n1;
        stk.put(++tv.top, swap ? e0 : e1);
        cur = swap ? e1 : e0;
    } else if (h0) cur = e0;
    else if (h1) cur = e1;
}

/* ---- leaves tested by the whole wave --------------------------------------------------
 * A leaf holds up to 16 triangles that the host tests one after the other for one ray
 * (src/main.c:449-482).  A lane doing that keeps 16 steps busy while lanes in the node part of
 * their walk wait.  Here the rays that have reached a leaf pool their (ray, triangle) pairs and
 * the wave tests 64 pairs per step.  The sequential rule "accept when 1e-3 <= t <= the closest
 * hit so far, barycentrics in range; equal t replaces" reduces to: among the valid pairs of a
 * ray with t <= its entry distance, the smallest t wins and ties go to the later triangle.
 * That is an integer minimum over keys (bits of t) << 32 | (0xFFFFFFFE - position in the
 * leaf), taken with 64-bit LDS atomics; t >= 1e-3 > 0, so float order is unsigned order.
 * The one case the rule does not cover, a NaN distance (the host then accepts whatever comes
 * next), sends the ray back to the one-by-one test: key 0 marks it. */
#ifndef RT_PAIR_STRAIGHT
#define RT_PAIR_STRAIGHT 1             /* the pair test without early returns (0: with them; measured 4 % slower tree launches) */
#endif
enum : int { PAIR_REJECTED = 0, PAIR_VALID = 1, PAIR_UNORDERED = 2 };

/* one (ray, triangle) pair against the distance the ray entered the leaf with */
__device__ __forceinline__ int leaf_pair(u32x4 fa, u32x4 fb, u32x4 fc, f3 mo, f3 mu, float entry_t,
                                         float &t, float &beta, float &gamma)
{
    const f3 a = mk(__uint_as_float(fa.x), __uint_as_float(fa.y), __uint_as_float(fa.z));
    const f3 e1v = mk(__uint_as_float(fa.w), __uint_as_float(fb.x), __uint_as_float(fb.y));
    const f3 e2v = mk(__uint_as_float(fb.z), __uint_as_float(fb.w), __uint_as_float(fc.x));
    const f3 n = mk(__uint_as_float(fc.y), __uint_as_float(fc.z), __uint_as_float(fc.w));
    const f3 oa = sub(mo, a);
    const float den = dot(mu, n);
    t = -dot(oa, n) / den;
#if RT_PAIR_STRAIGHT
    /* the host's early returns as one mask: among 64 pairs some need the barycentrics anyway */
    const f3 oau = cross(oa, mu);
    beta = -dot(e2v, oau) / den;
    gamma = dot(e1v, oau) / den;
    const bool rejected = (t < 1e-3f) | (t > entry_t) | (beta < 0) | (beta > 1) | (gamma < 0) | (gamma > 1) | (beta + gamma > 1);
    return rejected ? PAIR_REJECTED : (t != t ? PAIR_UNORDERED : PAIR_VALID);
#else
    if (t < 1e-3f) return PAIR_REJECTED;
    if (t > entry_t) return PAIR_REJECTED;
    const f3 oau = cross(oa, mu);
    beta = -dot(e2v, oau) / den;
    gamma = dot(e1v, oau) / den;
    if ((beta < 0) || (beta > 1)) return PAIR_REJECTED;
    if ((gamma < 0) || (gamma > 1)) return PAIR_REJECTED;
    if (beta + gamma > 1) return PAIR_REJECTED;
    return t != t ? PAIR_UNORDERED : PAIR_VALID;
#endif
}

/* what the lane does after a step: NEXT_NODE = visit the interior node whose record is at
 * `addr`; NEXT_LEAF = tv.tri_cur..tv.tri_end wait for the wave's next leaf pass; NEXT_DONE = no
 * instance left, the query is answered */
enum : int { NEXT_DONE = 0, NEXT_NODE = 1, NEXT_LEAF = 2, NEXT_SCAN = 3 };
/* may_scan = false: a lane whose instance is done and that has further instances to test does not test them
 * now; it returns NEXT_SCAN unchanged and the caller lets such lanes gather (the test of an instance's root
 * box -- two table records, a transform, six divisions -- is long, and run for the one or two lanes that
 * finish an instance in a given step it costs the wave as much as a step of all its lanes) */
template <bool STATS, class StackT = Stack>
__device__ __forceinline__ int walk_advance(const rt_dev_scene &S, f3 o, f3 d, const Hit &h, Trav &tv, StackT stk,
                                            uint32_t &cur, const u32x4 *&addr, Counters &cn, bool may_scan = true)
{
    if (cur == RT_ENTRY_NONE) {
        if (tv.top < 0) {                                /* this instance is done: on to the next one */
            if (tv.inst + 1 >= S.instance_count) { tv.inst++; return NEXT_DONE; }
            if (!may_scan) return NEXT_SCAN;
            tv.inst++;
            if (!query_scan<STATS, true>(S, o, d, h, tv, stk, cn)) return NEXT_DONE;
        }
        cur = stk.get(tv.top--);
    }
    if (!(cur & RT_ENTRY_LEAF)) {
        addr = reinterpret_cast<const u32x4 *>(S.nodes + tv.node_base + cur);
        return NEXT_NODE;
    }
    if (STATS) cn.leaf_pops++;
    tv.tri_cur = cur & RT_ENTRY_FIRST_MASK;
    tv.tri_end = tv.tri_cur + ((cur >> RT_ENTRY_COUNT_SHIFT) & 15u) + 1u;
    cur = RT_ENTRY_NONE;
    return NEXT_LEAF;
}

template <bool STATS>
__device__ __forceinline__ int query_step(const rt_dev_scene &S, bool any_hit, Hit &h, Trav &tv, Stack stk,
                                          Counters &cn)
{
    if (tv.tri_cur < tv.tri_end) return query_tri_step<STATS>(S, any_hit, h, tv, cn);
    return query_pop_step<STATS>(S, h, tv, stk, cn);
}

/* the whole query in one go (closest-hit probe kernel) */
template <bool STATS>
__device__ __forceinline__ void closest_hit(const rt_dev_scene &S, f3 o, f3 d, float tmax, bool any_hit,
                                            Stack stk, Hit &h, Counters &cn)
{
    if (query_begin<STATS>(S, o, d, tmax, any_hit, h, cn)) return;
    Trav tv;
    tv.inst = 0;
    while (query_scan<STATS>(S, o, d, h, tv, stk, cn)) {
        int r;
        do { r = query_step<STATS>(S, any_hit, h, tv, stk, cn); } while (r == STEP_MORE);
        if (r == STEP_QUERY_DONE) return;
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

/* ------------------------------------------------------------------ the path loop
 *
 * The host program nests its loops: samples > path segments > lights > tree
 * nodes (src/main.c:762, 624, 550, 397).  A wave that executed them nested
 * keeps most lanes idle: paths end at different segments, and a few lanes walk
 * deep into a mesh tree while the rest wait (measured on the first kernel: 8
 * of 64 lanes busy, 2-3 during tree walks).  Here every lane runs a flat loop
 * with two kinds of step, and the WAVE decides by ballot which kind to run:
 *
 *   shade step  (lanes that are not inside a tree walk)
 *       book the finished query: radiance, throughput, what comes next;
 *       one "event" = ONE generator draw turned into a ray:
 *          CAMERA  jittered primary ray                    (src/main.c:764-768)
 *          NEE     shadow ray to a point on light `li`     (src/main.c:553-597)
 *          BOUNCE  cosine-distributed continuation ray     (src/main.c:694-730)
 *       (all three share draw, sincos(2 pi u0) and one square root; NEE and
 *       BOUNCE share the tangent frame and the combination of the three axes);
 *       lights / spheres / planes for the new ray, then the instance root boxes;
 *   tree step   (lanes inside a tree walk)
 *       one node pop or one triangle test.
 *
 * Lanes that need a tree walk are parked until enough of them have gathered
 * (or nothing else is runnable); then the wave runs tree steps until enough
 * lanes are ready to shade again.  Both kinds of step thus run with many
 * lanes.  A lane executes, for its own path, exactly the operations the
 * nested host loops execute, in the same order; only the interleaving between
 * lanes differs, so results are unchanged.
 */
enum : int { EV_CAMERA = 0, EV_NEE = 1, EV_BOUNCE = 2, EV_TRACE_ONLY = 3 };
enum : int { RAY_NONE = 0, RAY_SEGMENT = 1, RAY_SHADOW = 2 };
enum : int { PH_SHADE = 0, PH_SCAN = 1, PH_TREE = 2, PH_DEAD = 3, PH_PARK = 4 };

/* A path in transit between the shade kernel and the tree kernel: 40 words.
 * Words 0..15 (the ray part) are all the tree kernel reads and writes:
 *   0 flags: valid | ev<<1 | ray<<3 | inside<<5 | fresh<<6 | ready<<7 | seg<<8      1 x | y<<16
 *     (ready: the query is complete, the record only waits to be carried on: no tree to walk)
 *   2..4 origin   5..7 direction   8 t  9 type  10 id  11 inst  12 beta  13 gamma  14 material
 *   15 instance to (re)start the walk at | light index<<16
 * then 16..18 throughput, 19..21 radiance, 22..24 normal, 25..27 albedo, 28..30 direct,
 * 31 lights reached, 32 reached here, 33 shadow length^2, 34 shadow cosine, 35..38 generator,
 * 39 index of the sample's slot in the sample scratch. */
#define RT_RAY_WORDS 16        /* the part of a record the tree kernel reads and answers (words 0..15): one aligned 64-byte piece */
#define RT_PATH_WORDS 24       /* the rest of the path (words 16..39) */
#define RT_RECORD_WORDS (RT_RAY_WORDS + RT_PATH_WORDS)
/* A list of `capacity` records keeps the ray parts one after the other, then the path parts: the tree
 * kernel's 64 bytes per record are whole memory sectors of their own, and the shade kernels move 160 bytes
 * per record in two dense streams. */
__device__ __forceinline__ uint32_t *ray_part(uint32_t *list, size_t i) { return list + i * RT_RAY_WORDS; }
__device__ __forceinline__ const uint32_t *ray_part(const uint32_t *list, size_t i) { return list + i * RT_RAY_WORDS; }
__device__ __forceinline__ uint32_t *path_part(uint32_t *list, size_t capacity, size_t i) { return list + capacity * RT_RAY_WORDS + i * RT_PATH_WORDS; }
__device__ __forceinline__ const uint32_t *path_part(const uint32_t *list, size_t capacity, size_t i) { return list + capacity * RT_RAY_WORDS + i * RT_PATH_WORDS; }

/* SPECULAR = false compiles the mirror and glass branches out: for scenes without such materials
 * (the host checks), whose kernels then carry neither their code nor their registers. */
template <class Rng, bool STATS, bool SPECULAR = true, bool CACHE = false>
struct PathMachine {
    const rt_dev_scene &S;
    Stack stk;
    Counters &cn;
    Rng rng;
    float half_w, half_h;
    int max_segments;

    /* per-lane state */
    int phase;                    /* PH_* */
    int ev;                       /* next event */
    int ray;                      /* kind of the query in flight / just finished */
    uint32_t x, y;                /* pixel */
    f3 o, d;                      /* current ray */
    Hit h;                        /* its (partial) result */
    Trav tv;
    f3 throughput, radiance;
    float lights_reached;
    bool inside_glass;
    int seg;
    /* the diffuse vertex being lit */
    f3 n, albedo, direct;
    float reached_here, shadow_len2, shadow_cos;
    int li;

    __device__ __forceinline__ PathMachine(const rt_dev_scene &S_, Stack stk_, Counters &cn_)
        : S(S_), stk(stk_), cn(cn_) {}

    __device__ __forceinline__ void start_path()
    {
        throughput = mk(1.0f, 1.0f, 1.0f);
        radiance = mk(0.0f, 0.0f, 0.0f);
        lights_reached = 0;
        inside_glass = false;
        seg = 0;
        ev = EV_CAMERA;
        ray = RAY_NONE;
        phase = PH_SHADE;
    }

    /* value of the finished path, src/main.c:735-739 and the clamp of 777-779 */
    __device__ __forceinline__ f3 path_value() const
    {
        f3 c = mk(0.0f, 0.0f, 0.0f);
        if (!(lights_reached == 0)) c = scale(radiance, 1 / lights_reached);
        float c2 = len2(c);
        if (c2 > 1e4f) c = scale(c, (float)(1e2 / (double)sqrt_unscaled(c2)));
        return c;
    }

    /* a segment was consumed: the host's loop counter (src/main.c:624) */
    __device__ __forceinline__ bool segment_budget_left()
    {
        seg++;
        return seg < max_segments;
    }

    /* after the last light of a diffuse vertex: src/main.c:677-692 */
    __device__ __forceinline__ bool finish_vertex()
    {
        lights_reached += reached_here;
        radiance = add(radiance, mul(throughput, mul(direct, albedo)));
        throughput = mul(throughput, albedo);
        if (len2(throughput) <= 9.99999997475242707878e-7f) return false;   /* < 1e-6 (double) on floats */
        ev = EV_BOUNCE;
        return true;
    }

    /* book the query that just finished; true = the path ended */
    __device__ __forceinline__ bool book()
    {
        bool ended = false;
        if (ray == RAY_SEGMENT) {
            if (h.type == HIT_NONE) {
                ended = true;
            } else if (h.type == HIT_LIGHT) {                             /* src/main.c:631-643 */
                lights_reached += 1;
                const rt_dev_light &l = S.lights[h.id];
                float s = (float)((double)l.albedo / (4.0 * RT_PI_D * (double)h.t * (double)h.t));
                radiance = add(radiance, mul(throughput, scale(ld3(l.color), s)));
                ended = true;
            } else {
                Surface sf;
                surface_of<STATS>(S, o, d, h, sf, cn);
                const rt_dev_material &m = S.materials[h.material];
                if (SPECULAR && m.kind == 1u) {                           /* mirror, src/main.c:644-653 */
                    throughput = mul(throughput, ld3(m.mirror_color));
                    f3 along = scale(sf.n, dot(sf.n, d));
                    f3 across = sub(d, along);
                    o = sf.p;
                    d = sub(across, scale(along, 2.0f));                  /* unnormalised on purpose */
                    if (segment_budget_left()) ev = EV_TRACE_ONLY; else ended = true;
                } else if (SPECULAR && m.kind == 2u) {                    /* glass, src/main.c:654-674 */
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
                        f3 forward = scale(unit(along), sqrt_rn(1 - b2));
                        d = add(forward, bent);
                    }
                    if (segment_budget_left()) ev = EV_TRACE_ONLY; else ended = true;
                } else {                                                  /* diffuse, src/main.c:675-683 */
                    o = sf.p;
                    n = sf.n;
                    albedo = ld3(m.diffuse);
                    if (m.texture >= 0) { albedo = texel(S, m.texture, sf.u, sf.v); if (STATS) cn.tex_lookups++; }
                    direct = mk(0, 0, 0);
                    reached_here = 0;
                    li = 0;
                    if (S.light_count > 0) ev = EV_NEE;
                    else if (!finish_vertex()) ended = true;
                }
            }
        } else if (ev == EV_NEE) {                                        /* a light sample came back (or was skipped) */
            if (ray == RAY_SHADOW && h.type == HIT_NONE) {                /* src/main.c:599-605 */
                const rt_dev_light &l = S.lights[li];
                float power = (float)((double)l.albedo / (4.0 * RT_PI_D * (double)shadow_len2));
                direct = add(direct, scale(ld3(l.color), power * shadow_cos));
                reached_here += 1;
            }
            li++;
            if (li >= S.light_count && !finish_vertex()) ended = true;
        }
        ray = RAY_NONE;
        return ended;
    }

    /* one generator draw -> one ray (ray, o, d, tmax); true = the path ended instead */
    __device__ __forceinline__ bool event(float &tmax)
    {
        tmax = 3.402823466e+38f;
        if (ev == EV_TRACE_ONLY) { ray = RAY_SEGMENT; return false; }
        bool ended = false;
        float u0, u1;
        rng.uniform2(u0, u1);
        if (STATS) cn.rng_draws++;
        float sn, cs;
#if defined(RT_ABLATE_TRIG)      /* timing experiment only: wrong bits */
        sn = __sinf((float)(2.0 * RT_PI_D * (double)u0)); cs = __cosf((float)(2.0 * RT_PI_D * (double)u0));
#else
        rt_sincosf((float)(2.0 * RT_PI_D * (double)u0), &sn, &cs);
#if RT_DOUBLE == 3
        {
            float x2 = (float)(2.0 * RT_PI_D * (double)u0), s2, c2;
            RT_OPAQUE(x2);
            rt_sincosf(x2, &s2, &c2);
            if (s2 != sn) { sn = s2; cs = c2; }
        }
#endif
#endif
        float arg;
#if defined(RT_ABLATE_TRIG)
        if (ev == EV_CAMERA) arg = -2 * __logf(u1);
#else
        if (ev == EV_CAMERA) arg = -2 * rt_logf(u1);
#endif                      /* Box-Muller radius^2, src/math.c:230 */
        else if (ev == EV_NEE) arg = 2 * u1 - u1 * u1;                    /* src/main.c:555 */
        else arg = 1 - u1;                                                /* src/main.c:696 */
        float ring = sqrt_unscaled(arg);                /* arg is zero, infinite or at least 2^-24: u1 is a multiple of 2^-23 */
        if (ev == EV_CAMERA) {
            if (STATS) cn.samples++;
            float jx = 0.0f + 0.25f * cs * ring;
            float jy = 0.0f + 0.25f * sn * ring;
            f3 dr = scale(ld3(S.cam.right), (float)x - half_w + 0.5f + jx);
            f3 du = scale(ld3(S.cam.up), -((float)y - half_h + 0.5f + jy));
            f3 dm = scale(ld3(S.cam.forward), S.cam.focal);
            o = ld3(S.cam.pos);
            d = unit(add(add(dr, du), dm));
            ray = RAY_SEGMENT;
        } else {
            f3 axis, lc = mk(0, 0, 0);
            float c0, c1 = -(sn * ring), c2 = -(cs * ring);
            if (ev == EV_NEE) {
                lc = ld3(S.lights[li].c);
                axis = unit(sub(lc, o));                                  /* o is the vertex */
                c0 = -(1 - u1);
            } else {
                axis = n;
                c0 = sqrt_unscaled(u1);
            }
            f3 t0 = frame_tangent(axis);
            f3 t1 = cross(axis, t0);
            f3 v = unit(add(scale(axis, c0), add(scale(t0, c1), scale(t1, c2))));
            if (ev == EV_NEE) {
                f3 q = scale(v, S.lights[li].radius);
                f3 path = sub(add(q, lc), o);
                float path_len;
                f3 dir = unit(path, path_len);
                shadow_cos = dot(dir, n);
                if (!(shadow_cos < 0)) {
                    d = dir;
                    shadow_len2 = len2(path);
                    tmax = path_len;                                      /* norm(path), src/main.c:597 */
                    ray = RAY_SHADOW;
                }
            } else {
                throughput = scale(throughput, dot(v, n));
                if (len2(throughput) <= 9.99999997475242707878e-7f) ended = true;
                else {
                    d = v;
                    if (segment_budget_left()) ray = RAY_SEGMENT; else ended = true;
                }
            }
        }
        return ended;
    }

    /* shade step for the lanes in PH_SHADE / PH_SCAN.  on_end() is the caller's
     * bookkeeping when a path ends (add path_value() to the pixel, start the
     * next path or set phase = PH_DEAD). */
    template <class OnEnd>
    __device__ __forceinline__ void shade_step(OnEnd &&on_end)
    {
        if (phase == PH_SHADE) {
            if (book()) on_end();
            if (phase == PH_SHADE) {
                float tmax;
                if (event(tmax)) on_end();
                else if (ray != RAY_NONE) {
                    if (!query_begin<STATS>(S, o, d, tmax, ray == RAY_SHADOW, h, cn)) { tv.inst = 0; phase = PH_SCAN; }
                }
            }
        }
        if (phase == PH_SCAN)
            phase = query_scan<STATS, CACHE>(S, o, d, h, tv, stk, cn) ? PH_TREE : PH_SHADE;
    }

    /* ---- hand-over between kernels (counter generator only) ------------------- */
    int park_inst;                /* PH_PARK: instance whose tree the ray has to walk first */

    __device__ __forceinline__ void store_record(uint32_t *ray_words, uint32_t *path_words, uint32_t sample_index, bool ready = false) const
    {
        uint4 *q = reinterpret_cast<uint4 *>(ray_words);
        uint4 *r = reinterpret_cast<uint4 *>(path_words);
        const uint32_t flags = 1u | ((uint32_t)ev << 1) | ((uint32_t)ray << 3) | ((uint32_t)inside_glass << 5) |
                               ((uint32_t)rng.fresh << 6) | ((uint32_t)ready << 7) | ((uint32_t)seg << 8);
        q[0] = make_uint4(flags, x | (y << 16), __float_as_uint(o.x), __float_as_uint(o.y));
        q[1] = make_uint4(__float_as_uint(o.z), __float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z));
        q[2] = make_uint4(__float_as_uint(h.t), (uint32_t)h.type, h.id, h.inst);
        q[3] = make_uint4(__float_as_uint(h.beta), __float_as_uint(h.gamma), (uint32_t)h.material,
                          ((uint32_t)park_inst & 0xFFFFu) | ((uint32_t)li << 16));   /* park_inst is -1 or unset when the query is complete */
        r[0] = make_uint4(__float_as_uint(throughput.x), __float_as_uint(throughput.y), __float_as_uint(throughput.z),
                          __float_as_uint(radiance.x));
        r[1] = make_uint4(__float_as_uint(radiance.y), __float_as_uint(radiance.z), __float_as_uint(n.x),
                          __float_as_uint(n.y));
        r[2] = make_uint4(__float_as_uint(n.z), __float_as_uint(albedo.x), __float_as_uint(albedo.y),
                          __float_as_uint(albedo.z));
        r[3] = make_uint4(__float_as_uint(direct.x), __float_as_uint(direct.y), __float_as_uint(direct.z),
                          __float_as_uint(lights_reached));
        r[4] = make_uint4(__float_as_uint(reached_here), __float_as_uint(shadow_len2), __float_as_uint(shadow_cos),
                          rng.k0);
        r[5] = make_uint4(rng.k1, rng.c0, rng.c1, sample_index);
    }

    /* false = the slot holds no path */
    __device__ __forceinline__ bool load_record(const uint32_t *ray_words, const uint32_t *path_words, uint32_t &sample_index)
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(ray_words);
        const uint4 *r = reinterpret_cast<const uint4 *>(path_words);
        const uint4 a = q[0];
        if (!(a.x & 1u)) return false;
        unpack(a, q[1], q[2], q[3], r[0], r[1], r[2], r[3], r[4], r[5], sample_index);
        return true;
    }

    /* the record's ten 16-byte quads `stride` quads apart: a batch of records staged quad by quad in LDS */
    __device__ __forceinline__ bool load_record(const uint4 *q, uint32_t stride, uint32_t &sample_index)
    {
        const uint4 a = q[0];
        if (!(a.x & 1u)) return false;
        unpack(a, q[stride], q[2 * stride], q[3 * stride], q[4 * stride], q[5 * stride],
               q[6 * stride], q[7 * stride], q[8 * stride], q[9 * stride], sample_index);
        return true;
    }

    __device__ __forceinline__ void unpack(const uint4 a, const uint4 b, const uint4 c, const uint4 e, const uint4 f, const uint4 g,
                                           const uint4 i, const uint4 j, const uint4 k, const uint4 l, uint32_t &sample_index)
    {
        ev = (int)((a.x >> 1) & 3u); ray = (int)((a.x >> 3) & 3u); inside_glass = ((a.x >> 5) & 1u) != 0;
        rng.fresh = ((a.x >> 6) & 1u) != 0; seg = (int)(a.x >> 8);
        x = a.y & 0xFFFFu; y = a.y >> 16;
        o = mk(__uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x));
        d = mk(__uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
        h.t = __uint_as_float(c.x); h.type = (int)c.y; h.id = c.z; h.inst = c.w;
        h.beta = __uint_as_float(e.x); h.gamma = __uint_as_float(e.y); h.material = (int)e.z; h.entered = false;
        li = (int)(e.w >> 16);
        throughput = mk(__uint_as_float(f.x), __uint_as_float(f.y), __uint_as_float(f.z));
        radiance = mk(__uint_as_float(f.w), __uint_as_float(g.x), __uint_as_float(g.y));
        n = mk(__uint_as_float(g.z), __uint_as_float(g.w), __uint_as_float(i.x));
        albedo = mk(__uint_as_float(i.y), __uint_as_float(i.z), __uint_as_float(i.w));
        direct = mk(__uint_as_float(j.x), __uint_as_float(j.y), __uint_as_float(j.z));
        lights_reached = __uint_as_float(j.w);
        reached_here = __uint_as_float(k.x); shadow_len2 = __uint_as_float(k.y); shadow_cos = __uint_as_float(k.z);
        rng.k0 = k.w; rng.k1 = l.x; rng.c0 = l.y; rng.c1 = l.z;
        sample_index = l.w;
        phase = PH_SHADE;
    }

    /* shade step of the shade kernel: as shade_step, but a ray that enters an
     * instance's root box is not walked here: the lane asks to be parked
     * (PH_PARK, park_inst) and the tree kernel finishes the query. */
    unsigned long long *ticks = nullptr;   /* RT_SHADE_CLOCKS only */
    template <class OnEnd>
    __device__ __forceinline__ void shade_step_park(OnEnd &&on_end)
    {
        if (book()) on_end();
        RT_STICK(ticks, 2)
        if (phase == PH_SHADE) {
            float tmax;
            const bool ended_ = event(tmax);
            RT_STICK(ticks, 3)
            if (ended_) on_end();
            else if (ray != RAY_NONE) {
                if (RT_EARLY_OUT(STATS)) {
                    if (!query_begin<STATS>(S, o, d, tmax, ray == RAY_SHADOW, h, cn)) {
                        park_inst = query_probe<STATS>(S, o, d, h, 0, cn);
                        if (park_inst >= 0) phase = PH_PARK;
                    }
                } else {
                    /* the root boxes are tested whether or not the primitives already answered the query: some lane
                     * of the wave needs the test anyway, and one straight run of code beats a branch around it */
                    const bool answered = query_begin<STATS>(S, o, d, tmax, ray == RAY_SHADOW, h, cn);
                    RT_STICK(ticks, 4)
#if RT_DOUBLE == 4
                    {
                        Hit h2; f3 o2 = o; RT_OPAQUE(o2.x);
                        (void)query_begin<STATS>(S, o2, d, tmax, ray == RAY_SHADOW, h2, cn);
                        cn.rays--;
                        if (h2.t != h.t && h.t == h.t) h = h2;
                    }
#endif
                    park_inst = query_probe<STATS>(S, o, d, h, 0, cn);
#if RT_DOUBLE == 6
                    {
                        f3 o2 = o; RT_OPAQUE(o2.x);
                        const int p2 = query_probe<STATS>(S, o2, d, h, 0, cn);
                        if (p2 != park_inst) park_inst = p2;
                    }
#endif
                    if (!answered && park_inst >= 0) phase = PH_PARK;
                    RT_STICK(ticks, 5)
                }
            }
        }
    }

    __device__ __forceinline__ bool wants_shade() const { return phase == PH_SHADE || phase == PH_SCAN; }
    __device__ __forceinline__ bool wants_tri() const { return phase == PH_TREE && tv.tri_cur < tv.tri_end; }
    __device__ __forceinline__ bool wants_pop() const { return phase == PH_TREE && !(tv.tri_cur < tv.tri_end); }

    __device__ __forceinline__ void after_tree(int r)
    {
        if (r == STEP_INSTANCE_DONE) phase = tv.inst < S.instance_count ? PH_SCAN : PH_SHADE;
        else if (r == STEP_QUERY_DONE) phase = PH_SHADE;
    }

    /* the wave's scheduler: every lane wants exactly one kind of step (shade,
     * pop a tree node, test a triangle); the wave runs the kind most lanes
     * want, weighted (a shade step is long, so it should run fuller).  Runs
     * until every lane is PH_DEAD. */
    template <class OnEnd>
    __device__ __forceinline__ void run(int shade_weight_q4, OnEnd &&on_end)
    {
        run(shade_weight_q4, on_end, []() {});
    }

    /* refill() is called at the top of every iteration, by all lanes: it may give PH_DEAD lanes
     * a new path (the loop ends when every lane is dead after it) */
    template <class OnEnd, class Refill>
    __device__ __forceinline__ void run(int shade_weight_q4, OnEnd &&on_end, Refill &&refill)
    {
        for (;;) {
            refill();
            const int n_shade = __popcll(__ballot(wants_shade()));
            const int n_tri = __popcll(__ballot(wants_tri()));
            const int n_pop = __popcll(__ballot(wants_pop()));
            if (!(n_shade | n_tri | n_pop)) break;
            const int s_shade = n_shade * shade_weight_q4;          /* weights in 1/16ths */
            const int s_tri = n_tri * 16, s_pop = n_pop * 16;
            if (s_shade >= s_tri && s_shade >= s_pop) {
                if (STATS) { if ((threadIdx.x & 63) == 0) cn.shade_steps++; if (wants_shade()) cn.shade_lanes++; }
                shade_step(on_end);
            } else if (s_pop >= s_tri) {
                if (STATS) { if ((threadIdx.x & 63) == 0) cn.pop_steps++; if (wants_pop()) cn.pop_lanes++; }
                if (wants_pop()) after_tree(query_pop_step<STATS, CACHE>(S, h, tv, stk, cn));
            } else {
                if (STATS) { if ((threadIdx.x & 63) == 0) cn.tri_steps++; if (wants_tri()) cn.tri_lanes++; }
                if (wants_tri()) after_tree(query_tri_step<STATS>(S, ray == RAY_SHADOW, h, tv, cn));
            }
        }
    }
};

} // namespace rt
