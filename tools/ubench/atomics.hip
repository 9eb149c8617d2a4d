// micro-benchmark: fp32 global atomic-add throughput patterns on gfx950 (informs grid_encode_backward design)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// each thread: R rows (pseudo-random or coherent), C consecutive channels per row
template <int C, int MODE>
__global__ void k_atomics(float *buf, uint32_t rows, uint32_t n_threads, int R) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    for (int r = 0; r < R; ++r) {
        uint32_t row;
        if (MODE == 0) row = hash32(t * 131u + r) % rows;                 // random rows
        else if (MODE == 1) row = (hash32((t >> 5) * 131u + r)) % rows;   // 32 neighbouring lanes share a row (contention)
        else row = (t * R + r) % rows;                                    // unique consecutive rows (streaming)
        float *p = buf + (size_t)row * C;
#pragma unroll
        for (int c = 0; c < C; ++c) unsafeAtomicAdd(p + c, 1.0f);
    }
}
template <int C>
__global__ void k_rmw(float *buf, uint32_t rows, uint32_t n_threads, int R) {   // non-atomic RMW for comparison
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    for (int r = 0; r < R; ++r) {
        uint32_t row = hash32(t * 131u + r) % rows;
        float *p = buf + (size_t)row * C;
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] += 1.0f;
    }
}
int main() {
    const uint32_t rows = 5258512; const int C = 8;
    float *buf; hipMalloc(&buf, (size_t)rows * C * 4); hipMemset(buf, 0, (size_t)rows * C * 4);
    const uint32_t n = 131072 * 16; const int R = 8;   // = 16.8M row updates x 8 channels = 134M atomics
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto launch, double natom) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("%-44s %8.3f ms  %7.2f G atomics/s\n", name, ms, natom / ms / 1e6);
    };
    dim3 g((n + 255) / 256), blk(256);
    run("random rows, 8 ch/row (134M)", [&] { hipLaunchKernelGGL((k_atomics<8, 0>), g, blk, 0, 0, buf, rows, n, R); }, (double)n * R * 8);
    run("random rows, 2 ch/row (34M)", [&] { hipLaunchKernelGGL((k_atomics<2, 0>), g, blk, 0, 0, buf, rows, n, R); }, (double)n * R * 2);
    run("random rows, 1 ch/row (17M)", [&] { hipLaunchKernelGGL((k_atomics<1, 0>), g, blk, 0, 0, buf, rows, n, R); }, (double)n * R);
    run("32 lanes share a row, 8 ch (134M)", [&] { hipLaunchKernelGGL((k_atomics<8, 1>), g, blk, 0, 0, buf, rows, n, R); }, (double)n * R * 8);
    run("unique consecutive rows, 8 ch (134M)", [&] { hipLaunchKernelGGL((k_atomics<8, 2>), g, blk, 0, 0, buf, rows, n, R); }, (double)n * R * 8);
    run("random rows in 4096-row table, 8 ch (134M)", [&] { hipLaunchKernelGGL((k_atomics<8, 0>), g, blk, 0, 0, buf, 4096u, n, R); }, (double)n * R * 8);
    run("random rows in 65536-row table, 8 ch", [&] { hipLaunchKernelGGL((k_atomics<8, 0>), g, blk, 0, 0, buf, 65536u, n, R); }, (double)n * R * 8);
    run("non-atomic RMW random rows, 8 ch", [&] { hipLaunchKernelGGL((k_rmw<8>), g, blk, 0, 0, buf, rows, n, R); }, (double)n * R * 8);
    return 0;
}
