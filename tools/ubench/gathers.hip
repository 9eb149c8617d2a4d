// micro-benchmark: 8-byte gather throughput on gfx950 as a function of table size (L2 / Infinity Cache / HBM),
// lane coherence and loads in flight (informs the hash-grid gather design)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// MODE 0: every lane random row; MODE 1: lanes of a wave within one 4x4x4-vertex neighbourhood (rows r0 + small offsets);
// MODE 2: all lanes of a wave the same row (broadcast)
template <int G, int MODE>
__global__ __launch_bounds__(256) void k_gather(const float2 *__restrict__ tab, uint32_t mask, int iters, float *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = t >> 6, lane = t & 63u;
    float ax = 0, ay = 0;
    for (int it = 0; it < iters; ++it) {
        float2 v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t row;
            if (MODE == 0) row = hash32(t * 977u + it * 131u + g) & mask;
            else if (MODE == 1) row = (hash32(wave * 977u + it * 131u + g) + (lane & 3u) + 64u * ((lane >> 2) & 3u) + 4096u * (lane >> 4)) & mask;
            else if (MODE == 2) row = hash32(wave * 977u + it * 131u + g) & mask;
            else row = ((hash32(wave * 977u + it * 131u + g + 7919u * (lane / (64u / MODE))) & mask) & ~15u) | (lane & 15u);  // MODE = distinct 128-B lines per instruction
            v[g] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(tab) + (size_t)row * 8u);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) { ax += v[g].x; ay += v[g].y; }
    }
    if (ax == 12345.678f) out[t] = ax + ay;
}
// cost of partially active gather instructions: ACT = 0 all lanes, 1 = even lanes only, 2 = lanes 0-31 only, 3 = random half;
// every active lane reads its own line of an L2-resident table (the worst case for the line cost) or, COH, one shared line
template <int ACT, bool COH>
__global__ __launch_bounds__(256) void k_gather_masked(const float2 *__restrict__ tab, uint32_t mask, int iters, float *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = t >> 6, lane = t & 63u;
    const bool act = ACT == 0 ? true : ACT == 1 ? (lane & 1u) == 0u : ACT == 2 ? lane < 32u : (hash32(lane * 7919u + 13u) & 1u) != 0u;
    float ax = 0, ay = 0;
    for (int it = 0; it < iters; ++it) {
        float2 v[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const uint32_t row = COH ? ((hash32(wave * 977u + it * 131u + g) & mask & ~15u) | (lane & 15u)) : (hash32(t * 977u + it * 131u + g) & mask);
            v[g] = make_float2(0.f, 0.f);
            if (act) v[g] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(tab) + (size_t)row * 8u);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) { ax += v[g].x; ay += v[g].y; }
    }
    if (ax == 12345.678f) out[t] = ax + ay;
}

// address-phase cost by element width: coherent gathers (one line per wave) of 4 / 8 / 16 bytes per lane
template <typename V>
__global__ __launch_bounds__(256) void k_gather_width(const char *__restrict__ tab, uint32_t mask, int iters, float *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = t >> 6, lane = t & 63u;
    float ax = 0;
    for (int it = 0; it < iters; ++it) {
        V v[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const uint32_t line = hash32(wave * 977u + it * 131u + g) & (mask >> 4);
            v[g] = *reinterpret_cast<const V *>(tab + (size_t)line * 128u + (lane * sizeof(V)) % 128u);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) ax += *reinterpret_cast<const float *>(&v[g]);
    }
    if (ax == 12345.678f) out[t] = ax;
}

int main() {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float *out; hipMalloc(&out, 1 << 24);
    const size_t max_rows = 1u << 27;   // 1 GiB
    float2 *tab; hipMalloc(&tab, max_rows * 8); hipMemset(tab, 0, max_rows * 8);
    auto run = [&](const char *name, auto launch, double n) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("%-64s %8.3f ms  %8.1f G lane-gathers/s  %6.2f G wave-instr/s\n", name, ms, n / ms / 1e6, n / 64 / ms / 1e6);
    };
    const int iters = 64;
    for (int blocks_per_cu : {2, 4, 8}) {
        const uint32_t nblk = 256 * blocks_per_cu * 2;
        const double n = (double)nblk * 256 * iters;
        char nm[128];
        for (uint32_t log2rows : {19u, 22u, 27u}) {   // 4 MiB, 32 MiB, 1 GiB of 8-byte rows
            const uint32_t mask = (1u << log2rows) - 1u;
            snprintf(nm, sizeof nm, "random, table 2^%u rows, 8 loads in flight, %d blk/CU-ish", log2rows, blocks_per_cu);
            run(nm, [&] { hipLaunchKernelGGL((k_gather<8, 0>), dim3(nblk), dim3(256), 0, 0, tab, mask, iters, out); }, n * 8);
            snprintf(nm, sizeof nm, "random, table 2^%u rows, 32 loads in flight, %d blk/CU-ish", log2rows, blocks_per_cu);
            run(nm, [&] { hipLaunchKernelGGL((k_gather<32, 0>), dim3(nblk), dim3(256), 0, 0, tab, mask, iters, out); }, n * 32);
        }
        snprintf(nm, sizeof nm, "4x4x4 neighbourhood per wave, 2^22 rows, 32 in flight, %d blk/CU-ish", blocks_per_cu);
        run(nm, [&] { hipLaunchKernelGGL((k_gather<32, 1>), dim3(nblk), dim3(256), 0, 0, tab, (1u << 22) - 1u, iters, out); }, n * 32);
        snprintf(nm, sizeof nm, "same row for the whole wave, 2^22 rows, 32 in flight, %d blk/CU-ish", blocks_per_cu);
        run(nm, [&] { hipLaunchKernelGGL((k_gather<32, 2>), dim3(nblk), dim3(256), 0, 0, tab, (1u << 22) - 1u, iters, out); }, n * 32);
    }
    // cost of one wave-wide 8-byte gather as a function of the number of distinct 128-byte lines its lanes touch
    {
        const uint32_t nblk = 2048; const double n = (double)nblk * 256 * iters * 16;
        for (uint32_t log2rows : {19u, 23u}) {
            const uint32_t mask = (1u << log2rows) - 1u; char nm[128];
#define LINES(L) snprintf(nm, sizeof nm, "%2d distinct lines per instruction, table 2^%u rows", L, log2rows); \
            run(nm, [&] { hipLaunchKernelGGL((k_gather<16, L>), dim3(nblk), dim3(256), 0, 0, tab, mask, iters, out); }, n);
            LINES(4) LINES(8) LINES(16) LINES(32) LINES(64)
        }
    }
    {
        const uint32_t nblk = 2048; const double n = (double)nblk * 256 * iters * 16; const uint32_t mask = (1u << 19) - 1u;
        run("coherent, 4 bytes per lane (dword)", [&] { hipLaunchKernelGGL((k_gather_width<uint32_t>), dim3(nblk), dim3(256), 0, 0, (const char *)tab, mask, iters, out); }, n);
        run("coherent, 8 bytes per lane (dwordx2)", [&] { hipLaunchKernelGGL((k_gather_width<uint2>), dim3(nblk), dim3(256), 0, 0, (const char *)tab, mask, iters, out); }, n);
        run("coherent, 16 bytes per lane (dwordx4)", [&] { hipLaunchKernelGGL((k_gather_width<uint4>), dim3(nblk), dim3(256), 0, 0, (const char *)tab, mask, iters, out); }, n);
    }
    // does the address path skip inactive lanes?
    {
        const uint32_t nblk = 2048; const double n = (double)nblk * 256 * iters * 16; const uint32_t mask = (1u << 19) - 1u;
#define MASKED(A, C, label) run(label, [&] { hipLaunchKernelGGL((k_gather_masked<A, C>), dim3(nblk), dim3(256), 0, 0, tab, mask, iters, out); }, n);
        MASKED(0, true, "coherent (1 line), all 64 lanes active")
        MASKED(1, true, "coherent (1 line), even lanes active")
        MASKED(2, true, "coherent (1 line), lanes 0-31 active")
        MASKED(3, true, "coherent (1 line), random half active")
        MASKED(0, false, "64 lines, all 64 lanes active")
        MASKED(1, false, "64 lines, even lanes active (32 lines)")
        MASKED(2, false, "64 lines, lanes 0-31 active (32 lines)")
        MASKED(3, false, "64 lines, random half active (32 lines)")
    }
    return 0;
}
