// mfma_rate.hip -- issue interval of the fp16 matrix-core instructions on gfx950, one wave per SIMD and two: how many shader cycles does
// a v_mfma_f32_32x32x16_f16 (16 384 MACs, 8 passes = 32 cycles at the 2.5 PFLOP/s peak) really take when the accumulator changes
// every instruction, every third instruction (the hi/lo-split triple on one accumulator) or never?  And the 16x16x32 form?
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate_ub
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    half8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    floatx16 c[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) c[t][i] = 0.0f;
    floatx4 d[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) d[t][i] = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {           // 8 accumulators round robin: every MFMA switches accumulator
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
        } else if constexpr (MODE == 1) {    // triples on one accumulator, 8 accumulators
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 3; ++r) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
        } else if constexpr (MODE == 2) {    // one accumulator, 24 dependent MFMAs
#pragma unroll
            for (int r = 0; r < 24; ++r) c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[0], 0, 0, 0);
        } else if constexpr (MODE == 3) {    // 16x16x32, 8 accumulators round robin (48 = the MACs of 24 32x32x16)
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[t], 0, 0, 0);
        } else if constexpr (MODE == 5) {    // as mode 0, accumulators in AGPRs (what k_mlp_wide's 128 accumulator registers are)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[t]) : "v"(a), "v"(b));
        } else if constexpr (MODE == 6) {    // AGPR accumulators, A operand alternating between two register quads re-loaded from LDS each time
            __shared__ half8_t sm[256];
            sm[threadIdx.x] = a;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const half8_t aa = *reinterpret_cast<volatile half8_t *>(&sm[(threadIdx.x + t) & 255]);
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[t]) : "v"(aa), "v"(b));
                }
        } else {                             // 16x16x32, triples on one accumulator
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 6; ++r) d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int t = 0; t < 8; ++t) { for (int i = 0; i < 16; ++i) s += c[t][i]; for (int i = 0; i < 4; ++i) s += d[t][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int mfma_per_iter, double macs_each) {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks = 256 * ((256 * waves_per_simd) / threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, cyc, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / iters / mfma_per_iter;
    const double tf = 2.0 * macs_each * mfma_per_iter * iters * (blocks * (threads / 64.0)) / (ms * 1e-3) / 1e12;
    printf("%-58s %d wave(s)/SIMD  %6.1f shader cycles per MFMA per wave  %7.1f TFLOP/s chip\n", name, waves_per_simd, per, tf);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("32x32x16 f16, accumulator changes every instruction", w, 24, 16384.0);
        run<1>("32x32x16 f16, three in a row per accumulator", w, 24, 16384.0);
        run<2>("32x32x16 f16, one accumulator (24 dependent)", w, 24, 16384.0);
        run<3>("16x16x32 f16, accumulator changes every instruction", w, 48, 8192.0);
        run<4>("16x16x32 f16, six in a row per accumulator", w, 48, 8192.0);
        run<5>("32x32x16 f16, AGPR accumulators, changing every instruction", w, 24, 16384.0);
        run<6>("32x32x16 f16, AGPR accumulators, A operand from LDS each time", w, 24, 16384.0);
    }
    return 0;
}
