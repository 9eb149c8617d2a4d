// micro-test: is v_mfma_f32_16x16x4_f32 bit-identical to a k-ascending fmaf chain?  (would allow the proposal MLP, whose outputs decide the
// sample indices and must match the oracle's fmaf chains bit for bit, to run on the matrix cores)
// D[m][n] = sum_k A[m][k] * B[k][n], accumulated over 3 chained instructions (K = 12, the last two columns zero = the 10-input layer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D) {          // A [16][12], B [12][16], D [16][16]
    const int l = threadIdx.x, r16 = l & 15, q = l >> 4;
    f4 acc = {0, 0, 0, 0};
    for (int s = 0; s < 3; ++s) {
        const float a = A[r16 * 12 + 4 * s + q];                        // lane l: A[m = l%16][k = 4s + l/16]
        const float b = B[(4 * s + q) * 16 + r16];                      // lane l: B[k = 4s + l/16][n = l%16]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + r16] = acc[r];     // lane l, reg r: D[m = 4*(l/16) + r][n = l%16]
}
int main() {
    std::vector<float> A(16 * 12), B(12 * 16), D(256), ref(256);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    int bad_total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd() * (trial % 3 == 0 ? 1e-3f : 1.0f);
        for (int m = 0; m < 16; ++m) { A[m * 12 + 10] = 0; A[m * 12 + 11] = 0; }
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            float acc = 0;
            for (int kk = 0; kk < 12; ++kk) acc = fmaf(A[m * 12 + kk], B[kk * 16 + n], acc);
            if (memcmp(&acc, &D[m * 16 + n], 4) != 0) ++bad;
        }
        bad_total += bad;
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    printf("v_mfma_f32_16x16x4_f32 x3 vs k-ascending fmaf chain: %d of %d outputs differ\n", bad_total, 200 * 256);
    return 0;
}
