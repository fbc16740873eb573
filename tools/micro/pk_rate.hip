// Issue rate of v_fma_f32 against v_pk_fma_f32 (two fp32 FMAs per lane per instruction) on one GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ITER 4096
__global__ void scalar_fma(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITER; i++) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void packed_fma(float *out, float a, float b)
{
    f2 aa = {a, a}, bb = {b, b};
    f2 x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITER; i++) {
        x0 = __builtin_elementwise_fma(x0, aa, bb); x1 = __builtin_elementwise_fma(x1, aa, bb);
        x2 = __builtin_elementwise_fma(x2, aa, bb); x3 = __builtin_elementwise_fma(x3, aa, bb);
        x4 = __builtin_elementwise_fma(x4, aa, bb); x5 = __builtin_elementwise_fma(x5, aa, bb);
        x6 = __builtin_elementwise_fma(x6, aa, bb); x7 = __builtin_elementwise_fma(x7, aa, bb);
    }
    f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (which == 0) scalar_fma<<<4096, 256>>>(d, 0.999f, 0.001f); else packed_fma<<<4096, 256>>>(d, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = 4096.0 * 4 * N_ITER * 8;   // wave instructions
            if (rep == 2) printf("%s: %.3f ms, %.1f G wave-instructions/s, %.1f TFLOP/s\n", which ? "v_pk_fma_f32" : "v_fma_f32", ms,
                                 insts / ms / 1e6, insts * 64 * 2 * (which ? 2 : 1) / ms / 1e9);
        }
    }
    return 0;
}
