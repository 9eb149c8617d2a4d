// micro-benchmark: does a cache-policy bit (nt / sc0 / sc1) change what a scattered 8-byte gather costs per distinct 128-byte
// line?  (the final stage's fine hash levels touch ~1 line per lane and reuse nothing)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int POL> __device__ __forceinline__ float2 ld(const char *base, uint32_t off) {
    float2 v;
    if constexpr (POL == 0) v = *reinterpret_cast<const float2 *>(base + off);
    else if constexpr (POL == 1) asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(v) : "v"(off), "s"(base));
    else if constexpr (POL == 2) asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(v) : "v"(off), "s"(base));
    else if constexpr (POL == 3) asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(off), "s"(base));
    else if constexpr (POL == 4) asm volatile("global_load_dwordx2 %0, %1, %2 sc0" : "=v"(v) : "v"(off), "s"(base));
    else asm volatile("global_load_dwordx2 %0, %1, %2 sc1 nt" : "=v"(v) : "v"(off), "s"(base));
    return v;
}
// LINES distinct 128-byte lines per wave instruction (lanes of a group share a line)
template <int POL, int LINES>
__global__ __launch_bounds__(256) void k(const char *__restrict__ tab, uint32_t mask, int iters, float *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = t >> 6, lane = t & 63u;
    float ax = 0, ay = 0;
    for (int it = 0; it < iters; ++it) {
        float2 v[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const uint32_t row = ((hash32(wave * 977u + it * 131u + g + 7919u * (lane / (64u / LINES))) & mask) & ~15u) | (lane & 15u);
            v[g] = ld<POL>(tab, row * 8u);
        }
        if constexpr (POL != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 16; ++g) { ax += v[g].x; ay += v[g].y; }
    }
    if (ax == 12345.678f) out[t] = ax + ay;
}
int main() {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float *out; hipMalloc(&out, 1 << 24);
    char *tab; hipMalloc(&tab, (size_t)1 << 30); hipMemset(tab, 0, (size_t)1 << 30);
    const int iters = 64; const uint32_t nblk = 2048; const double n = (double)nblk * 256 * iters * 16;
    const char *pn[6] = {"plain", "nt", "sc1", "sc0 sc1", "sc0", "sc1 nt"};
    auto run = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("%-60s %8.3f ms  %6.2f G wave-instr/s\n", name, ms, n / 64 / ms / 1e6);
    };
    for (uint32_t log2rows : {19u, 22u}) {       // 4 MiB (L2-resident), 32 MiB (Infinity Cache)
        const uint32_t mask = (1u << log2rows) - 1u; char nm[128];
#define RUN(P, L) snprintf(nm, sizeof nm, "%-8s %2d lines/instr, table 2^%u rows", pn[P], L, log2rows); \
        run(nm, [&] { hipLaunchKernelGGL((k<P, L>), dim3(nblk), dim3(256), 0, 0, tab, mask, iters, out); });
        RUN(0, 32) RUN(1, 32) RUN(2, 32) RUN(3, 32) RUN(4, 32) RUN(5, 32)
        RUN(0, 64) RUN(1, 64) RUN(2, 64) RUN(3, 64)
        RUN(0, 4) RUN(1, 4) RUN(2, 4)
    }
    return 0;
}
