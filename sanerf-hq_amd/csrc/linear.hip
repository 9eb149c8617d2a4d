// linear.hip — a GENERAL fp32 matrix product on the matrix cores in true fp32 (gfx950): the route of every layer shape the specialised kernels
// (mlp_small.hip: the reference network's three small perceptrons; mlp.hip: 256-wide heads) do not instantiate.
//
//   sn_gemm_f32      C[M,N] = act(A B + bias),  A[i][k] = a[i * a_row + k * a_col],  B[k][j] = b[k * b_row + j * b_col]   (one stride of each = 1)
//
// What it replaces: torch.nn.functional.linear / `@` (rocBLAS) on the cold routes of ops.py -- nn.Linear forward (A = x, B = W^T), input gradient
// (A = dy, B = W) and, for layers wider than sn_linear_wgrad takes, the weight gradient (A = dy^T, B = x); nerf/network.py:9-66 with layer widths
// other than the reference's.  With it no BLAS library call is left anywhere in the product.
//
// Shape: 64 x 64 output tile per 256-thread workgroup, K in chunks of 32 through LDS (rows padded to 33 floats: the operand reads of 32
// consecutive rows hit 32 different banks), each wave one 32 x 32 quadrant on v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulation, ONE
// k-ascending chain per output element (deterministic, equal to the fmaf chain of the oracle's dense layers).  Tiles are loaded element-wise
// with the unit stride along the fast thread index whichever operand is transposed, one chunk ahead (the next chunk's elements are in flight
// under this chunk's matrix instructions).  This is the cold path: it is not tuned beyond that
// (measured: tests/test_gpu_ops.py prints nothing; tools/gemm_f32_bench.py does).
#include "sn_common.h"

namespace sn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *a, *b, *bias;
    float *c;
    uint32_t M, N, K;
    int64_t a_row, a_col, b_row, b_col, c_row;
    int32_t act;                    // 0 none, 1 ReLU, 2 leaky ReLU (slope 0.01)
};

constexpr uint32_t GM_T = 64, GM_KB = 32, GM_LD = GM_KB + 1;

__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
    __shared__ float as[GM_T * GM_LD], bs[GM_T * GM_LD];        // as[i][k], bs[j][k]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t m0 = blockIdx.x * GM_T, n0 = blockIdx.y * GM_T;
    const uint32_t wm = (wave & 1u) * 32u, wn = (wave >> 1) * 32u;
    const bool a_kfast = g.a_col == 1, b_kfast = g.b_row == 1;
    constexpr uint32_t PER = (GM_T * GM_KB) / 256u;              // elements of each tile per thread
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // the elements this thread stages: fixed (row, k) slots, the unit stride along the fast thread index whichever operand is transposed
    float ra[PER], rb[PER];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t e = tid + 256u * r;
            {
                const uint32_t i = a_kfast ? e / GM_KB : e % GM_T, k = a_kfast ? e % GM_KB : e / GM_T;
                const bool in = m0 + i < g.M && k0 + k < g.K;
                ra[r] = in ? g.a[(int64_t)(m0 + i) * g.a_row + (int64_t)(k0 + k) * g.a_col] : 0.0f;
            }
            {
                const uint32_t j = b_kfast ? e / GM_KB : e % GM_T, k = b_kfast ? e % GM_KB : e / GM_T;
                const bool in = n0 + j < g.N && k0 + k < g.K;
                rb[r] = in ? g.b[(int64_t)(k0 + k) * g.b_row + (int64_t)(n0 + j) * g.b_col] : 0.0f;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t e = tid + 256u * r;
            const uint32_t i = a_kfast ? e / GM_KB : e % GM_T, ka = a_kfast ? e % GM_KB : e / GM_T;
            const uint32_t j = b_kfast ? e / GM_KB : e % GM_T, kb = b_kfast ? e % GM_KB : e / GM_T;
            as[i * GM_LD + ka] = ra[r];
            bs[j * GM_LD + kb] = rb[r];
        }
    };
    if (g.K) fetch(0u);
    for (uint32_t k0 = 0; k0 < g.K; k0 += GM_KB) {
        stage();
        __syncthreads();
        if (k0 + GM_KB < g.K) fetch(k0 + GM_KB);       // the next chunk's elements fly under this chunk's matrix instructions
        // (rows of zeros beyond K: fma(0, 0, acc) = acc, the chain of an output is unchanged)
        const float *ap = as + (wm + (lane & 31u)) * GM_LD + (lane >> 5), *bp = bs + (wn + (lane & 31u)) * GM_LD + (lane >> 5);
#pragma unroll
        for (uint32_t k = 0; k < GM_KB; k += 2u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
        __syncthreads();
    }
    // register r of lane l = C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] of the quadrant
    const uint32_t col = n0 + wn + (lane & 31u);
    if (col < g.N) {
        const float bj = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
        for (uint32_t r = 0; r < 16u; ++r) {
            const uint32_t row = m0 + wm + (r & 3u) + 8u * (r >> 2) + 4u * (lane >> 5);
            if (row < g.M) {
                float v = acc[r] + bj;
                if (g.act == 1) v = __builtin_fmaxf(v, 0.0f);
                else if (g.act == 2) v = __builtin_fmaxf(v, v * 0.01f);
                g.c[(int64_t)row * g.c_row + col] = v;
            }
        }
    }
}

}  // namespace sn

using namespace sn;

extern "C" int sn_gemm_f32(const float *a, int64_t a_row, int64_t a_col, const float *b, int64_t b_row, int64_t b_col, const float *bias,
                           int32_t act, uint32_t M, uint32_t N, uint32_t K, float *c, int64_t c_row, sn_stream_t stream) {
    SN_REQUIRE(c || M == 0 || N == 0, "gemm_f32: NULL output");
    if (M == 0 || N == 0) return SN_OK;
    SN_REQUIRE((a && b) || K == 0, "gemm_f32: NULL operand");
    SN_REQUIRE(act >= 0 && act <= 2, "gemm_f32: activation %d (0 none, 1 ReLU, 2 leaky ReLU)", act);
    SN_REQUIRE((a_row == 1 || a_col == 1) && (b_row == 1 || b_col == 1), "gemm_f32: one stride of each operand must be 1 (got %lld/%lld, %lld/%lld)",
               (long long)a_row, (long long)a_col, (long long)b_row, (long long)b_col);
    SN_REQUIRE(c_row >= (int64_t)N, "gemm_f32: output row stride %lld < N = %u", (long long)c_row, N);
    const uint32_t gx = div_up(M, GM_T), gy = div_up(N, GM_T);
    SN_REQUIRE(gy <= 65535u, "gemm_f32: N = %u is beyond 65535 column tiles", N);
    GemmArgs g;
    g.a = a; g.b = b; g.bias = bias; g.c = c; g.M = M; g.N = N; g.K = K;
    g.a_row = a_row; g.a_col = a_col; g.b_row = b_row; g.b_col = b_col; g.c_row = c_row; g.act = act;
    hipLaunchKernelGGL(k_gemm_f32, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK("k_gemm_f32");
    return SN_OK;
}
