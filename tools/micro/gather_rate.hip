/*
 * gather_rate.hip -- what a 64-byte record per lane costs in the vector memory path of a gfx950 CU, by the way it is asked for.
 *
 * Every lane of a wave wants the 64-byte record of its own (pseudo-random) index, as a tree walk wants its next node.
 *   own    : four 16-byte loads per lane from the lane's own record: a wave instruction touches up to 64 cache lines
 *   shared : instruction k has the four lanes of a group of four load the four quads of the record that lane 4g+k wants:
 *            a wave instruction touches 16 records (the data then sits in the wrong lane: the walk would take it through LDS
 *            with global_load_lds, whose destination is laid out by lane; here only the requests are measured)
 *   shared_lds : the same requests as global_load_lds, each lane reading its record back from LDS
 * Table sizes: 16 KB (L1), 2 MB (L2), 64 MB (beyond L2).  Output: wave instructions per microsecond per CU and shader
 * clocks per wave instruction per CU (at 2.4 GHz), grid = 7 workgroups of 256 lanes per CU.
 *
 * build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/gather_rate tools/micro/gather_rate.hip
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) gather(const u32x4 *table, uint32_t record_mask, int iterations, uint32_t *out)
{
    __shared__ u32x4 stage[4][4 * 64];                 /* per wave: quad slot k * 64 + lane */
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    u32x4 acc = 0;
    for (int i = 0; i < iterations; i++) {
        state = state * 1664525u + 1013904223u;
        const uint32_t record = (state >> 8) & record_mask;
        if (MODE == 0) {
            const u32x4 *q = table + (size_t)record * 4u;
            acc += q[0]; acc += q[1]; acc += q[2]; acc += q[3];
        } else {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)record, 0x00, 0xF, 0xF, false);   /* quad_perm:[0,0,0,0] */
            const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)record, 0x55, 0xF, 0xF, false);
            const uint32_t r2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)record, 0xAA, 0xF, 0xF, false);
            const uint32_t r3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)record, 0xFF, 0xF, 0xF, false);
            const uint32_t part = lane & 3u;
            if (MODE == 1) {
                acc += table[(size_t)r0 * 4u + part]; acc += table[(size_t)r1 * 4u + part];
                acc += table[(size_t)r2 * 4u + part]; acc += table[(size_t)r3 * 4u + part];
            } else {
                uint32_t *base = reinterpret_cast<uint32_t *>(&stage[wave][0]);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(table + (size_t)r0 * 4u + part), base, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(table + (size_t)r1 * 4u + part), base + 256, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(table + (size_t)r2 * 4u + part), base + 512, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t *>(table + (size_t)r3 * 4u + part), base + 768, 16, 0, 0);
                __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0) */
                const u32x4 *mine = &stage[wave][(lane & 3u) * 64u + (lane & ~3u)];
                acc += mine[0]; acc += mine[1]; acc += mine[2]; acc += mine[3];
            }
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 7, iterations = 4000;
    const size_t sizes[3] = {16u << 10, 2u << 20, 64u << 20};
    u32x4 *table; uint32_t *out;
    CHECK(hipMalloc(&table, sizes[2])); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<uint32_t> host(sizes[2] / 4);
    for (size_t i = 0; i < host.size(); i++) host[i] = (uint32_t)(i * 2654435761u);
    CHECK(hipMemcpy(table, host.data(), sizes[2], hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[3] = {"own", "shared", "shared_lds"};
    printf("%d CUs, %d workgroups of 256, %d records per lane; a record is 64 bytes = 4 wave instructions\n", cus, blocks, iterations);
    for (int s = 0; s < 3; s++) for (int mode = 0; mode < 3; mode++) {
        const uint32_t mask = (uint32_t)(sizes[s] / 64u) - 1u;
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0, nullptr));
            if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, nullptr, table, mask, iterations, out);
            else if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, nullptr, table, mask, iterations, out);
            else hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, nullptr, table, mask, iterations, out);
            CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4.0 * iterations * 4.0;     /* waves x records x 4 loads */
        const double per_cu_per_us = wave_instr / cus / (best * 1e3);
        printf("table %8zu KB  %-10s %8.3f ms  %7.2f wave loads per us per CU  %6.1f clocks per wave load per CU (2.4 GHz)  %6.1f per record of a wave\n",
               sizes[s] >> 10, names[mode], best, per_cu_per_us, 2400.0 / per_cu_per_us, 4.0 * 2400.0 / per_cu_per_us);
    }
    return 0;
}
