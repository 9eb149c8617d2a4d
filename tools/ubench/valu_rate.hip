// micro-benchmark: issue cost of scalar vs packed fp32 vector instructions on gfx950 as a function of waves per SIMD.
// Question behind it: the fused stages retire ~one vector instruction per 4 cycles per SIMD; does a
// v_pk_fma_f32 (two FMAs per lane) cost one such slot or two?  Chains are independent (8 accumulators per lane), so the
// result is the issue rate, not the dependent latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float seed, float *out) {
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                else if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if constexpr (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
                else if constexpr (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(m), "v"(c));
                else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}
int main() {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float *out; hipMalloc(&out, 4096);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const char *names[6] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_mul_f32", "v_pk_fma_f32 op_sel bcast", "v_xor_b32"};
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {                  // waves per SIMD: blocks of 256 threads, wps blocks per CU
        const uint32_t nblk = 256u * wps;
        auto run = [&](int kind, auto launch) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr_per_simd = (double)wps * iters * 64;        // per SIMD: wps waves x iters x 64 instructions
            printf("%d waves/SIMD  %-28s %8.3f ms  %6.2f ns/instr/SIMD = %5.2f cycles at %.2f GHz (nominal)\n", wps, names[kind], ms,
                   ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * clk_khz * 1e-6, clk_khz * 1e-6);
        };
#define RUN(KD) run(KD, [&] { hipLaunchKernelGGL((k<KD>), dim3(nblk), dim3(256), 0, 0, iters, 1.0f, out); });
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    }
    return 0;
}
