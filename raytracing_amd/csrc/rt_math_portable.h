/*
 * rt_math_portable.h -- sinf/cosf/logf replacements whose result bits do not
 * depend on the machine, the compiler or a math library.
 *
 * The render loop calls cosf/sinf on 2*pi*u and logf on u, u a multiple of
 * 2^-23 in [0,1) (reference: src/math.c:230-231, src/main.c:555-556,696-697).
 * A GPU has no libm; its native sin/cos/log instructions are approximate and
 * differ from any CPU's.  These versions use only IEEE double add, multiply,
 * fused multiply-add, one division, rint and conversions, every one written
 * out explicitly (no a*b+c left for a compiler to fuse or not), so a host C
 * build and a gfx950 device build produce identical floats.  The error before
 * the final rounding to float is below 2^-45 relative, so the result is the
 * correctly rounded float except when the exact value lies within 2^-21 ulp
 * of a rounding boundary.
 *
 * Usable from C, C++ and HIP device code.
 */
#ifndef RT_MATH_PORTABLE_H
#define RT_MATH_PORTABLE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_MATH_FN __host__ __device__ static inline
#else
#define RT_MATH_FN static inline
#endif

/* RT_K(c): a double constant of a polynomial.  On the host it is the literal.  In gfx950 device
 * code a 64-bit constant cannot be an instruction operand; the compiler builds each one in a
 * register pair and, inside a loop, keeps all of them alive from before the loop on: the thirty
 * constants of the functions below then hold sixty vector registers (or get spilled to scratch
 * memory).  Here every use makes its constant afresh in a scalar register pair with two scalar
 * moves, right where a multiply-add reads it: same value, no long-lived registers. */
#if defined(__HIP_DEVICE_COMPILE__)
template <unsigned long long BITS>
__device__ static inline double rt_k_scalar()
{
    unsigned lo, hi;
    __asm__ volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi)
                     : "i"((unsigned)(BITS & 0xFFFFFFFFull)), "i"((unsigned)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
#define RT_K(c) rt_k_scalar<__builtin_bit_cast(unsigned long long, (double)(c))>()
/* RT_FMA_K(a, b, c) = fma(a, b, c) with c such a constant.  Left to itself the compiler copies the scalar
 * pair into two vector registers and uses the two-operand multiply-add that overwrites its addend: three
 * vector instructions per step of a polynomial.  The three-operand form takes the scalar pair as its
 * addend directly: one instruction, the same fused operation. */
template <unsigned long long BITS>
__device__ static inline double rt_fma_k(double a, double b)
{
    unsigned lo, hi;
    __asm__ volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi)
                     : "i"((unsigned)(BITS & 0xFFFFFFFFull)), "i"((unsigned)(BITS >> 32)));
    const unsigned long long k = ((unsigned long long)hi << 32) | (unsigned long long)lo;
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#define RT_FMA_K(a, b, c) rt_fma_k<__builtin_bit_cast(unsigned long long, (double)(c))>((a), (b))
#else
#define RT_K(c) (c)
#define RT_FMA_K(a, b, c) __builtin_fma((a), (b), (c))
#endif

/* sine and cosine of a float argument (radians), |x| <= 2^20; NaN beyond.
 * Cody-Waite reduction to |y| <= pi/4 with a 33-bit + 53-bit split of pi/2
 * (n * hi is exact), then Taylor polynomials through y^15 / y^16. */
RT_MATH_FN void rt_sincosf(float x, float *sin_out, float *cos_out)
{
    const double xd = (double)x;
    if (!(__builtin_fabs(xd) <= 1048576.0)) {
        const float bad = x - x;   /* NaN for inf/NaN; large finite values are outside the contract */
        *sin_out = bad + __builtin_nanf("");
        *cos_out = bad + __builtin_nanf("");
        return;
    }
    const double fn = __builtin_rint(xd * RT_K(6.36619772367581382433e-01));       /* x * 2/pi */
    double y = __builtin_fma(-fn, RT_K(1.57079632673412561417e+00), xd);          /* pi/2, high 33 bits */
    y = __builtin_fma(-fn, RT_K(6.07710050650619224932e-11), y);                   /* pi/2, the rest */
    const int q = (int)fn & 3;
    const double y2 = y * y;

    double ps = RT_K(1.0 / 1307674368000.0);                                       /* 1/15! */
    ps = RT_FMA_K(ps, y2, -1.0 / 6227020800.0);                         /* 1/13! */
    ps = RT_FMA_K(ps, y2, 1.0 / 39916800.0);
    ps = RT_FMA_K(ps, y2, -1.0 / 362880.0);
    ps = RT_FMA_K(ps, y2, 1.0 / 5040.0);
    ps = RT_FMA_K(ps, y2, -1.0 / 120.0);
    ps = RT_FMA_K(ps, y2, 1.0 / 6.0);
    const double y3 = y * y2;
    const double sv = __builtin_fma(-y3, ps, y);                             /* y - y^3 (1/6 - ...) */

    double pc = RT_K(1.0 / 20922789888000.0);                                      /* 1/16! */
    pc = RT_FMA_K(pc, y2, -1.0 / 87178291200.0);                        /* 1/14! */
    pc = RT_FMA_K(pc, y2, 1.0 / 479001600.0);
    pc = RT_FMA_K(pc, y2, -1.0 / 3628800.0);
    pc = RT_FMA_K(pc, y2, 1.0 / 40320.0);
    pc = RT_FMA_K(pc, y2, -1.0 / 720.0);
    pc = RT_FMA_K(pc, y2, 1.0 / 24.0);
    pc = __builtin_fma(pc, y2, -0.5);
    const double cv = __builtin_fma(y2, pc, 1.0);

    const double s = (q & 1) ? cv : sv;
    const double c = (q & 1) ? sv : cv;
    *sin_out = (float)((q & 2) ? -s : s);
    *cos_out = (float)(((q + 1) & 2) ? -c : c);
}

/* natural logarithm of a float; log(0) = -inf, log(x<0) = NaN, as libm.
 * x = m * 2^e with m in (sqrt(1/2), sqrt(2)], log m = 2 atanh((m-1)/(m+1)). */
RT_MATH_FN float rt_logf(float x)
{
    if (!(x > 0.0f)) {
        if (x == 0.0f) return -__builtin_inff();
        return (x - x) + __builtin_nanf("");
    }
    if (x == __builtin_inff()) return x;
    const double xd = (double)x;             /* every finite float is a normal double */
    uint64_t bits;
    __builtin_memcpy(&bits, &xd, 8);
    int e = (int)(bits >> 52) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    __builtin_memcpy(&m, &bits, 8);
    if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = RT_K(1.0 / 23.0);
    p = RT_FMA_K(p, s2, 1.0 / 21.0);
    p = RT_FMA_K(p, s2, 1.0 / 19.0);
    p = RT_FMA_K(p, s2, 1.0 / 17.0);
    p = RT_FMA_K(p, s2, 1.0 / 15.0);
    p = RT_FMA_K(p, s2, 1.0 / 13.0);
    p = RT_FMA_K(p, s2, 1.0 / 11.0);
    p = RT_FMA_K(p, s2, 1.0 / 9.0);
    p = RT_FMA_K(p, s2, 1.0 / 7.0);
    p = RT_FMA_K(p, s2, 1.0 / 5.0);
    p = RT_FMA_K(p, s2, 1.0 / 3.0);
    const double s3 = s * s2;
    const double atanh_s = __builtin_fma(s3, p, s);
    const double two_atanh = atanh_s + atanh_s;
    /* ln 2 split so that e * hi is exact (hi has 32 significant bits) */
    double r = __builtin_fma((double)e, RT_K(1.90821492927058770002e-10), two_atanh);
    r = __builtin_fma((double)e, RT_K(6.93147180369123816490e-01), r);
    return (float)r;
}

/* c^(1/2.2f) for the display transfer of src/main.c:73-82, c in [0, 1] (c < 0 or NaN: NaN):
 * exp(g * log c) in double, log as in rt_logf before its final rounding, exp by Cody-Waite
 * reduction with a split ln 2 and the Taylor series through y^13 (|y| <= 0.35), 2^k put together
 * from its exponent bits.  Relative error before the rounding to float below 2^-48. */
RT_MATH_FN float rt_gamma_powf(float c)
{
    if (!(c > 0.0f)) {
        if (c == 0.0f) return 0.0f;
        return (c - c) + __builtin_nanf("");
    }
    if (c == 1.0f) return 1.0f;
    const double xd = (double)c;
    uint64_t bits;
    __builtin_memcpy(&bits, &xd, 8);
    int e = (int)(bits >> 52) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    __builtin_memcpy(&m, &bits, 8);
    if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = RT_K(1.0 / 23.0);
    p = RT_FMA_K(p, s2, 1.0 / 21.0);
    p = RT_FMA_K(p, s2, 1.0 / 19.0);
    p = RT_FMA_K(p, s2, 1.0 / 17.0);
    p = RT_FMA_K(p, s2, 1.0 / 15.0);
    p = RT_FMA_K(p, s2, 1.0 / 13.0);
    p = RT_FMA_K(p, s2, 1.0 / 11.0);
    p = RT_FMA_K(p, s2, 1.0 / 9.0);
    p = RT_FMA_K(p, s2, 1.0 / 7.0);
    p = RT_FMA_K(p, s2, 1.0 / 5.0);
    p = RT_FMA_K(p, s2, 1.0 / 3.0);
    const double s3 = s * s2;
    const double atanh_s = __builtin_fma(s3, p, s);
    const double two_atanh = atanh_s + atanh_s;
    double l = __builtin_fma((double)e, RT_K(1.90821492927058770002e-10), two_atanh);
    l = __builtin_fma((double)e, RT_K(6.93147180369123816490e-01), l);            /* ln c */
    const double z = l * (double)(1.0f / 2.2f);                               /* the host's float exponent, widened */
    const double fk = __builtin_rint(z * RT_K(1.44269504088896338700e+00));         /* z / ln 2 */
    double y = __builtin_fma(-fk, RT_K(6.93147180369123816490e-01), z);
    y = __builtin_fma(-fk, RT_K(1.90821492927058770002e-10), y);
    double q = RT_K(1.0 / 6227020800.0);                                            /* 1/13! */
    q = RT_FMA_K(q, y, 1.0 / 479001600.0);
    q = RT_FMA_K(q, y, 1.0 / 39916800.0);
    q = RT_FMA_K(q, y, 1.0 / 3628800.0);
    q = RT_FMA_K(q, y, 1.0 / 362880.0);
    q = RT_FMA_K(q, y, 1.0 / 40320.0);
    q = RT_FMA_K(q, y, 1.0 / 5040.0);
    q = RT_FMA_K(q, y, 1.0 / 720.0);
    q = RT_FMA_K(q, y, 1.0 / 120.0);
    q = RT_FMA_K(q, y, 1.0 / 24.0);
    q = RT_FMA_K(q, y, 1.0 / 6.0);
    q = __builtin_fma(q, y, 0.5);
    q = __builtin_fma(q, y, 1.0);
    q = __builtin_fma(q, y, 1.0);                                             /* exp(y) */
    const int k = (int)fk;                                                    /* -70 <= k <= 0 for c in [2^-149, 1) */
    const uint64_t scale_bits = (uint64_t)(1023 + k) << 52;
    double scale;
    __builtin_memcpy(&scale, &scale_bits, 8);
    return (float)(q * scale);
}

/* one channel of save_image (src/main.c:73-82): min(c, 1) with a NaN giving 1, the transfer
 * function, times 255, truncation */
RT_MATH_FN uint8_t rt_gamma_byte(float c)
{
    float v = c < 1.0f ? c : 1.0f;
    v = rt_gamma_powf(v);
    v = v * 255.0f;
    return (uint8_t)v;
}

#endif /* RT_MATH_PORTABLE_H */
