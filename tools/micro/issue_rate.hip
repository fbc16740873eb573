// Issue rates of the instruction classes the render kernels are made of, on one GPU: 64-lane instructions per second for
// independent chains of v_fma_f32, v_fma_f64, v_rcp_f32, v_sqrt_f32, v_cndmask_b32, v_add_u32 + v_alignbit (the generator's
// round), at 1, 2, 4 and 8 waves per SIMD.  The ratios are the cost of a class in units of one fp32 instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 16384
template <int KIND>
__global__ void chain(float *out, float a, float b, unsigned k)
{
    float x[8]; double y[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + i + 1.0f; y[i] = x[i]; u[i] = threadIdx.x * 2654435761u + i; }
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (KIND == 1) y[i] = __builtin_fma(y[i], (double)a, (double)b);
            if (KIND == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (KIND == 3) x[i] = __builtin_amdgcn_sqrtf(x[i]);
            if (KIND == 4) { u[i] = (u[i] > k) ? u[i] - k : u[i] + 7u; }                 /* compare + select + add */
            if (KIND == 5) { u[i] += u[(i + 1) & 7]; u[(i + 1) & 7] = __builtin_rotateleft32(u[(i + 1) & 7], 13) ^ u[i]; }   /* add, rotate, xor */
            if (KIND == 6) x[i] = x[i] * a;
            if (KIND == 7) x[i] = x[i] + b;
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i] + (float)y[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> static void run(const char *name, double per_iter, float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 20; rep++) chain<0><<<2048, 256>>>(d, 0.999f, 0.001f, 12345u);   /* some tens of milliseconds of load first: clocks settle */
    hipDeviceSynchronize();
    for (int waves = 1; waves <= 8; waves *= 2) {          /* waves per SIMD: 256 CUs x 4 SIMDs x waves, 4 waves per workgroup */
        const int blocks = 256 * waves;
        float ms = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            float t = 0;
            hipEventRecord(e0);
            chain<KIND><<<blocks, 256>>>(d, 0.999f, 0.001f, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1);
            if (t < ms) ms = t;
        }
        const double insts = (double)blocks * 4 * N_ITER * 8 * per_iter;
        printf("%-28s %d waves/SIMD: %7.3f ms, %7.1f G wave-instructions/s\n", name, waves, ms, insts / ms / 1e6);
    }
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 1, d);
    run<6>("v_mul_f32", 1, d);
    run<7>("v_add_f32", 1, d);
    run<1>("v_fma_f64", 1, d);
    run<2>("v_rcp_f32", 1, d);
    run<3>("v_sqrt_f32", 1, d);
    run<4>("v_cmp + v_cndmask + v_add (3)", 3, d);
    run<5>("v_alignbit + v_xad_u32 (2)", 2, d);
    return 0;
}
