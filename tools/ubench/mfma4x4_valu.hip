// mfma4x4_valu.hip -- can the 10 -> 16 layer of the proposal MLP run on v_mfma_f32_4x4x1_16b_f32 without any data movement?
//   The 16-block 4x4x1 form multiplies, in each block of 4 lanes, a 4x1 column (A, lane = row) by a 1x4 row (B, lane = column) and adds a
//   4x4 tile (register = row, lane = column).  With B = feature k of the lane's own ray and A = W[4g + lane % 4][k] the tile is
//   hidden[4g + r] of the lane's own ray in register r: input and output are both in the one-lane-per-ray layout of k_prop_stage.  The
//   A operand is the same for every block: cbsz:4 abid:s broadcasts the 4 lanes of block s, so ONE register holds 16 (k, g) columns
//   and the whole 16 x 10 matrix sits in 3 loop-invariant registers.
// Part A: is the k-ascending chain of 10 such instructions bit-identical to the oracle's fmaf chain (also with denormal products, zeros, infinities)?
// Part B: issue rate of 40 MFMAs, of 160 v_fma_f32, and of both in one instruction stream, at 1..8 waves per SIMD: do the fp32 matrix
//         instructions share the vector ALUs (then nothing is gained) or run beside them?
// build: hipcc --offload-arch=gfx950 -O3 mfma4x4_valu.hip -o mfma4x4_valu_ub
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int S>
__device__ __forceinline__ void step(f4 (&acc)[4], const float (&wa)[3], const float (&x)[10]) {
    acc[S % 4] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[S / 16], x[S / 4], acc[S % 4], 4, S % 16, 0);
}
template <int... S>
__device__ __forceinline__ void layer(f4 (&acc)[4], const float (&wa)[3], const float (&x)[10], std::integer_sequence<int, S...>) {
    (step<S>(acc, wa, x), ...);
}

__global__ void k_exact(const float *W, const float *X, float *H) {        // W [16][10], X [10][64], H [16][64]
    const int l = threadIdx.x;
    float wa[3], x[10];
    for (int v = 0; v < 3; ++v) {
        const int s = 16 * v + l / 4, kk = s / 4, g = s % 4;
        wa[v] = s < 40 ? W[(4 * g + l % 4) * 10 + kk] : 0.0f;
    }
    for (int kk = 0; kk < 10; ++kk) x[kk] = X[kk * 64 + l];
    f4 acc[4];
    for (int g = 0; g < 4; ++g) acc[g] = f4{0, 0, 0, 0};
    layer(acc, wa, x, std::make_integer_sequence<int, 40>{});
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 4; ++r) H[(4 * g + r) * 64 + l] = acc[g][r];
}

// MODE 0: 160 v_fma_f32 (8 independent chains)   1: 40 MFMAs (4 chains)   2: both, 4 fma after every MFMA   3: 523 fma (the stage's other vector work) + 40 MFMAs
// 4: 683 fma (the stage today)   5: 523 fma
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(int iters, float seed, float *out) {
    float a[8], wa[3], x[10];
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    for (int i = 0; i < 3; ++i) wa[i] = seed * 0.001f + i;
    for (int i = 0; i < 10; ++i) x[i] = seed * 0.002f - i;
    const float m = 1.0001f, c = 1e-3f;
    f4 acc[4];
    for (int g = 0; g < 4; ++g) acc[g] = f4{0, 0, 0, 0};
    auto fma8 = [&](int n) {
#pragma unroll
        for (int r = 0; r < n; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    };
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) fma8(20);
        else if constexpr (MODE == 1) layer(acc, wa, x, std::make_integer_sequence<int, 40>{});
        else if constexpr (MODE == 2) {
            // the compiler keeps asm volatile in order and places the builtins between them
#define Q(S) step<S>(acc, wa, x); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(S) % 8]) : "v"(m), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(S + 1) % 8]) : "v"(m), "v"(c)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(S + 2) % 8]) : "v"(m), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(S + 3) % 8]) : "v"(m), "v"(c));
#define Q4(S) Q(S) Q(S + 1) Q(S + 2) Q(S + 3)
            Q4(0) Q4(4) Q4(8) Q4(12) Q4(16) Q4(20) Q4(24) Q4(28) Q4(32) Q4(36)
        } else if constexpr (MODE == 3) {
            fma8(23);
            Q4(0) Q4(4) Q4(8) Q4(12) Q4(16) Q4(20) Q4(24) Q4(28) Q4(32) Q4(36)
            fma8(23);
        } else if constexpr (MODE == 4) fma8(85);
        else fma8(65);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int g = 0; g < 4; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    // ---- part A
    std::vector<float> W(160), X(640), H(1024);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    float *dW, *dX, *dH;
    hipMalloc(&dW, 640); hipMalloc(&dX, 2560); hipMalloc(&dH, 4096);
    const char *kinds[5] = {"random in (-8, 8)", "features x 1e-3", "denormal products (both x 1e-20)", "half of the features zero, some weights -0", "infinities and NaN among the features"};
    for (int kind = 0; kind < 5; ++kind) {
        int bad = 0, total = 0, denorm = 0;
        for (int trial = 0; trial < 100; ++trial) {
            for (auto &v : W) v = rnd();
            for (auto &v : X) v = rnd();
            if (kind == 1) for (auto &v : X) v *= 1e-3f;
            if (kind == 2) { for (auto &v : X) v *= 1e-20f; for (auto &v : W) v *= 1e-20f; }
            if (kind == 3) { for (size_t i = 0; i < X.size(); i += 2) X[i] = 0.0f; for (size_t i = 0; i < W.size(); i += 7) W[i] = -0.0f; }
            if (kind == 4) { X[trial % 640] = INFINITY; X[(trial * 7 + 3) % 640] = -INFINITY; X[(trial * 13 + 5) % 640] = NAN; }
            hipMemcpy(dW, W.data(), 640, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), 2560, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dW, dX, dH);
            hipMemcpy(H.data(), dH, 4096, hipMemcpyDeviceToHost);
            for (int h = 0; h < 16; ++h) for (int n = 0; n < 64; ++n) {
                float acc = 0;
                for (int kk = 0; kk < 10; ++kk) acc = fmaf(W[h * 10 + kk], X[kk * 64 + n], acc);
                const float got = H[h * 64 + n];
                ++total;
                if (acc != 0.0f && std::fabs(acc) < 1.17549435e-38f) ++denorm;
                if (std::isnan(acc) && std::isnan(got)) continue;           // NaN payloads are not part of the contract (the kernels never produce one from finite tables)
                if (memcmp(&acc, &got, 4) != 0) ++bad;
            }
        }
        printf("exact  %-46s %d of %d outputs differ from the k-ascending fmaf chain (%d denormal results among them)\n", kinds[kind], bad, total, denorm);
    }
    // ---- part B
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    float *out; hipMalloc(&out, 4096);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const char *names[6] = {"160 v_fma_f32", "40 v_mfma_f32_4x4x1", "160 v_fma_f32 + 40 mfma interleaved", "523 v_fma_f32 + 40 mfma", "683 v_fma_f32", "523 v_fma_f32"};
    const int iters = 2048;
    for (int wps : {1, 2, 4, 5, 8}) {
        const uint32_t nblk = 256u * wps;
        auto run = [&](int mode, auto launch) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(ea); launch(); hipEventRecord(eb); hipEventSynchronize(eb);
            float ms; hipEventElapsedTime(&ms, ea, eb);
            const double per_simd = (double)wps * iters;                     // sample-iterations per SIMD
            printf("rate   %d waves/SIMD  %-38s %8.3f ms  %8.1f cycles per wave-iteration per SIMD at %.2f GHz (nominal)\n", wps, names[mode], ms,
                   ms * 1e-3 / per_simd * clk_khz * 1e3, clk_khz * 1e-6);
        };
#define RUN(MD) run(MD, [&] { hipLaunchKernelGGL((k_rate<MD>), dim3(nblk), dim3(256), 0, 0, iters, 1.0f, out); });
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    }
    return 0;
}
