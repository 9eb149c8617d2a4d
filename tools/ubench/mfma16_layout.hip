// Layout probe of v_mfma_f32_16x16x32_f16 (mlp16.inc relies on it): lane l holds A[m = l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][n = l & 15],
// and register r of lane l of the result is D[4 (l >> 4) + r][l & 15].  Prints "layout OK" or the first mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D) {   // A [16][32], B [32][16], D [16][16]
    const int l = threadIdx.x, q = l >> 4, i = l & 15;
    half8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[i * 32 + 8 * q + j]; b[j] = (_Float16)B[(8 * q + j) * 16 + i]; }
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}
int main() {
    float hA[512], hB[512], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6) / 4.0f; hB[i] = (float)((i * 5 + 1) % 11 - 5) / 8.0f; }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (fabsf(hD[i] - ref[i]) > 1e-3f) { if (bad < 5) printf("mismatch D[%d][%d] = %f want %f\n", i / 16, i % 16, hD[i], ref[i]); ++bad; }
    printf(bad ? "layout WRONG (%d mismatches)\n" : "layout OK\n", bad);
    return bad != 0;
}
