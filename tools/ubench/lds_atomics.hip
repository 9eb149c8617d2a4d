// micro-benchmark: what does an LDS atomic cost on gfx950, per wave-wide instruction, as a function of the address pattern?
// Question behind it (grid_binned.hip, round 4): k_bin_accum adds 16.8 M x C contributions into a 64 KiB LDS bin with ds_add_f32 and
// measured ~230 cycles per wave instruction.  Is that the float atomic, the bank conflicts of random rows, or same-address lanes?
// Kinds: ds_add_f32, ds_add_u32, ds_add_rtn_u32, ds_add_rtn_f32, plain read+add+write (not atomic: the floor of a read-modify-write).
// Patterns: lane-consecutive dwords, random dwords in 16 K, all lanes one dword, 8 distinct dwords per wave, random rows x stride 8
// (row-major [row][8 channels], one channel per instruction: what a row-major accumulator would do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int N_DW = 16384;

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, int pattern, uint32_t seed, float *out, unsigned long long *cyc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < N_DW; i += 256) lds[i] = 0.0f;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = seed + threadIdx.x * 2654435761u + blockIdx.x * 805459861u;
    float acc = 0.0f;
    uint32_t *ldsu = reinterpret_cast<uint32_t *>(lds);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        uint32_t a;
        switch (pattern) {
            case 0: a = (lane + (uint32_t)it * 64u) & (N_DW - 1); break;
            case 1: a = (s >> 8) & (N_DW - 1); break;
            case 2: a = ((uint32_t)it * 97u) & (N_DW - 1); break;
            case 3: a = (((s >> 8) & 7u) * 1031u + (uint32_t)it * 8u) & (N_DW - 1); break;
            default: a = (((s >> 8) & 2047u) * 8u + ((uint32_t)it & 7u)) & (N_DW - 1); break;
        }
        if constexpr (KIND == 0) unsafeAtomicAdd(&lds[a], 1.0f);
        else if constexpr (KIND == 1) atomicAdd(&ldsu[a], 1u);
        else if constexpr (KIND == 2) acc += (float)atomicAdd(&ldsu[a], 1u);
        else if constexpr (KIND == 3) acc += atomicAdd(&lds[a], 1.0f);
        else { const float v = lds[a]; lds[a] = v + 1.0f; }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 12345.678f || lds[threadIdx.x] == -1.0f) out[threadIdx.x] = acc;
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8 * 4096);
    const char *kinds[5] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "ds_add_rtn_f32", "read+add+write (non-atomic)"};
    const char *pats[5] = {"lane-consecutive", "random in 16K dwords", "all lanes one dword", "8 distinct dwords", "random row x 8, one channel"};
    const int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs : {1, 2}) {                      // workgroups per CU (4 / 8 waves per CU)
        const uint32_t nblk = 256u * wgs;
        for (int kind = 0; kind < 5; ++kind)
            for (int p = 0; p < 5; ++p) {
                auto launch = [&] {
                    const size_t l = N_DW * 4;
                    switch (kind) {
                        case 0: hipLaunchKernelGGL((k<0>), dim3(nblk), dim3(256), l, 0, iters, p, 7u, out, cyc); break;
                        case 1: hipLaunchKernelGGL((k<1>), dim3(nblk), dim3(256), l, 0, iters, p, 7u, out, cyc); break;
                        case 2: hipLaunchKernelGGL((k<2>), dim3(nblk), dim3(256), l, 0, iters, p, 7u, out, cyc); break;
                        case 3: hipLaunchKernelGGL((k<3>), dim3(nblk), dim3(256), l, 0, iters, p, 7u, out, cyc); break;
                        default: hipLaunchKernelGGL((k<4>), dim3(nblk), dim3(256), l, 0, iters, p, 7u, out, cyc); break;
                    }
                };
                launch(); hipDeviceSynchronize();
                hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                // per CU: wgs workgroups x 4 waves x iters wave-instructions share one LDS
                const double instr_per_cu = (double)wgs * 4 * iters;
                printf("%d wg/CU  %-28s %-30s %8.3f ms  %7.1f ns per wave-instruction per CU (= %6.1f cycles at 2.4 GHz)\n", wgs, kinds[kind], pats[p], ms,
                       ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
            }
    }
    return 0;
}
