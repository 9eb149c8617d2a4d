// mlp_small.hip — the reference's SMALL bias-free ReLU perceptrons under training, on the matrix cores in TRUE fp32 (gfx950).
//
//   sn_mlp_small_forward_train   y = W_L relu(... relu(W_1 x)), every hidden output saved, output activation folded in
//   sn_mlp_small_backward        the whole backward DATA path: gradient of the input and of every pre-activation
//
// What it replaces: nerf/network.py:9-29 (`MLP`: F.linear + F.relu per layer) as instantiated by NeRFNetwork (network.py:93-98,
// 131-143): grid_mlp 32-64-64-16 (per sample), prop_mlp 10-16-1 (per sample, two stages), view_mlp 31-32-32-3 (per ray), together with
// what follows them on the training path: trunc_exp on the density channel (activation.py:5-17, network.py:155,179) and
// sigmoid + background blend on the colour (renderer.py:349-353).  Through torch that is one BLAS GEMM + one elementwise pass per layer and
// direction with [rows, width] round trips (round 5: 23 rocBLAS launches and ~90 elementwise launches per RGB step).
//
// Shape of the work (forward and backward are the SAME kernel template over a chain of matrices):
//   * transposed formulation H^T = M X^T on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation, fixed order: deterministic):
//     A = 32 output features x 2 inputs of the layer matrix, B = 2 inputs x 32 rows (samples), a wave owns 64 rows (two B tiles).
//   * the accumulator layout of a 32x32 tile (register r of lane l = D[(r&3) + 8 (r>>2) + 4 (l>>5)][l & 31]) IS a B-operand layout:
//     register r of the lanes' two halves holds features f and f+4 of the lane's row.  A hidden layer therefore feeds the next one from
//     registers; only the k order of the next matrix is permuted, and that happens once per workgroup when the matrices are laid into LDS
//     (one conflict-free ds_read_b32 per A operand, shared by the wave's two row tiles).
//   * forward chain: M_l = W_l, gate = ReLU, the gated output is stored (hidden[l]).  backward chain: M_l = W_{L+1-l}^T, gate = the sign
//     of the saved hidden output (torch's in-place ReLU backward), the gated result is stored (grad_hidden = gradient of the
//     pre-activation: what sn_linear_wgrad needs).  Nothing else round-trips memory.
//   * persistent workgroups: the matrices (<= 32 KiB) are permuted into LDS once, then the waves walk 64-row tiles.
// Work: grid_mlp forward = 256 MFMAs of 64 cycles per 64 rows -> 131 072 rows in ~15 us of matrix time; the kernels are bound by
// writing / reading the [rows, width] tensors the weight gradients need.
#include "sn_common.h"
#include <type_traits>

namespace sn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int SM_MAXL = 4;              // layers of a chain


struct SmallArgs {
    const float *w[SM_MAXL];            // nn.Linear.weight [out, in] of FORWARD layer l
    const float *in;                    // forward: x [N, d0];  backward: grad of the raw output [N, dL] or NULL
    const float *aux_grad;              // backward: grad of the activated output (act 1: [N]; act 2: [N, dL]) or NULL
    const float *raw;                   // backward, act != 0: the forward's raw output [N, dL]
    const float *aux_in;                // forward, act 2: weights_sum [N]
    const float *mask[SM_MAXL];         // backward: saved hidden output of forward layer l+1 (post-ReLU)
    float *store[SM_MAXL];              // forward: hidden[l]; backward: grad_hidden[l] (both indexed by FORWARD layer)
    float *out;                         // forward: raw output [N, dL];  backward: grad_in [N, d0] or NULL
    float *aux_out;                     // forward: activated output;  backward: combined gradient of the last pre-activation [N, dL]
    float *aux2_out;                    // backward, act 2: grad of aux_in [N] or NULL
    uint32_t N;
    int32_t act;
    float bg;
};

// A chain of NL matrices over widths C0 -> C1 -> ... (forward: the layer widths; backward: reversed).
template <bool BWD_, int NL_, int C0, int C1, int C2, int C3, int C4>
struct Chain {
    static constexpr bool BWD = BWD_;
    static constexpr int NL = NL_;
    static constexpr int c(int i) { return i == 0 ? C0 : i == 1 ? C1 : i == 2 ? C2 : i == 3 ? C3 : C4; }
    static constexpr int tiles(int i) { return (c(i) + 31) / 32; }
    static constexpr int S1 = (C0 + 1) / 2;                                  // k-steps of chain layer 1: lane half h supplies k = s + h * S1
    // k-steps of chain layer l >= 2 are (q, r) pairs of the previous layer's accumulator registers whose half-0 feature exists
    static constexpr bool step_ok(int l, int q, int r) { return 32 * q + 8 * (r >> 2) + (r & 3) < c(l - 1); }
    static constexpr int step_index(int l, int q, int r) {                   // position of (q, r) among the valid steps of layer l
        int n = 0;
        for (int qq = 0; qq < 2; ++qq)
            for (int rr = 0; rr < 16; ++rr)
                if ((qq < q || (qq == q && rr < r)) && qq < tiles(l - 1) && step_ok(l, qq, rr)) ++n;
        return n;
    }
    static constexpr int steps(int l) { return l == 1 ? S1 : step_index(l, 2, 0); }
    static constexpr int lds_off(int l) {                                    // first float of chain layer l's A operands
        int o = 0;
        for (int i = 1; i < l; ++i) o += steps(i) * tiles(i) * 64;
        return o;
    }
    static constexpr int lds_floats = lds_off(NL + 1);
};

// element [i][k] of chain layer l's matrix (i: output feature, k: input feature)
template <class P>
__device__ __forceinline__ float chain_weight(const SmallArgs &a, int l, uint32_t i, uint32_t k) {
    if constexpr (!P::BWD) return a.w[l - 1][(size_t)i * (uint32_t)P::c(l - 1) + k];            // W_l [c(l), c(l-1)]
    else return a.w[P::NL - l][(size_t)k * (uint32_t)P::c(l) + i];                               // W_f^T with f = NL+1-l: W_f is [c(l-1), c(l)]
}

// store / load a C-layout tile pair of width E: register group g of tile q of lane (j, h) = features 32 q + 8 g + 4 h + (0..3) of row j
template <int E>
__device__ __forceinline__ void tile_store(float *__restrict__ p, uint32_t row, bool row_ok, uint32_t h, int q, const floatx16 &v) {
    if (!row_ok) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t col = 32u * (uint32_t)q + 8u * (uint32_t)g + 4u * h;
        if constexpr (E % 4 == 0) {
            if (col < (uint32_t)E) *reinterpret_cast<float4 *>(p + (size_t)row * E + col) = float4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (col + (uint32_t)m < (uint32_t)E) p[(size_t)row * E + col + m] = v[4 * g + m];
        }
    }
}

template <int E>
__device__ __forceinline__ void tile_gate(const float *__restrict__ p, uint32_t row, bool row_ok, uint32_t h, int q, floatx16 &v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t col = 32u * (uint32_t)q + 8u * (uint32_t)g + 4u * h;
        if constexpr (E % 4 == 0) {
            float4 m = float4{0, 0, 0, 0};
            if (row_ok && col < (uint32_t)E) m = *reinterpret_cast<const float4 *>(p + (size_t)row * E + col);
            v[4 * g] = m.x > 0.0f ? v[4 * g] : 0.0f; v[4 * g + 1] = m.y > 0.0f ? v[4 * g + 1] : 0.0f;
            v[4 * g + 2] = m.z > 0.0f ? v[4 * g + 2] : 0.0f; v[4 * g + 3] = m.w > 0.0f ? v[4 * g + 3] : 0.0f;
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float mv = (row_ok && col + (uint32_t)m < (uint32_t)E) ? p[(size_t)row * E + col + m] : 0.0f;
                v[4 * g + m] = mv > 0.0f ? v[4 * g + m] : 0.0f;
            }
        }
    }
}

__device__ __forceinline__ float sigmoid_det(float x) { return 1.0f / (1.0f + expf_det(-x)); }

// chain layer L >= 2: acc = M_L * gate(prev), operands from the previous layer's accumulators
template <class P, int L>
__device__ __forceinline__ void layer_from_regs(const float *__restrict__ lds, uint32_t lane, const floatx16 (&prev)[2][2], floatx16 (&acc)[2][2]) {
    constexpr int NT = P::tiles(L);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[t][jt] = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < P::tiles(L - 1); ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!P::step_ok(L, q, r)) continue;
            const int s = P::step_index(L, q, r);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float av = lds[P::lds_off(L) + (s * NT + t) * 64 + lane];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) acc[t][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, prev[q][jt][r], acc[t][jt], 0, 0, 0);
            }
        }
}

// after chain layer L: gate + store (hidden layers) or the final store with the output activation
template <class P, int L>
__device__ __forceinline__ void layer_finish(const SmallArgs &a, const uint32_t (&row)[2], uint32_t h, floatx16 (&acc)[2][2]) {
    constexpr int E = P::c(L);
    if constexpr (L < P::NL) {
        // forward layer this gate belongs to (0-based index into store / mask): forward: L-1; backward: the hidden output of forward layer NL-L
        constexpr int F = P::BWD ? P::NL - L - 1 : L - 1;
#pragma unroll
        for (int q = 0; q < P::tiles(L); ++q)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bool ok = row[jt] < a.N;
                if constexpr (!P::BWD) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][jt][r] = acc[q][jt][r] > 0.0f ? acc[q][jt][r] : 0.0f;
                } else {
                    tile_gate<E>(a.mask[F], row[jt], ok, h, q, acc[q][jt]);
                }
                tile_store<E>(a.store[F], row[jt], ok, h, q, acc[q][jt]);
            }
    } else {
        if (a.out) {
#pragma unroll
            for (int q = 0; q < P::tiles(L); ++q)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) tile_store<E>(a.out, row[jt], row[jt] < a.N, h, q, acc[q][jt]);
        }
        if constexpr (!P::BWD) {
            // output feature i < 4 is register i of the half-0 lanes
            if (a.act == SN_SMALL_ACT_TRUNC_EXP0) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    if (h == 0u && row[jt] < a.N) a.aux_out[row[jt]] = expf_det(acc[0][jt][0]);
            } else if (a.act == SN_SMALL_ACT_SIGMOID_BG) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    if (h == 0u && row[jt] < a.N) {
                        const float t = (1.0f - a.aux_in[row[jt]]) * a.bg;                  // renderer.py:353
#pragma unroll
                        for (int i = 0; i < (E < 4 ? E : 4); ++i) a.aux_out[(size_t)row[jt] * E + i] = sigmoid_det(acc[0][jt][i]) + t;
                    }
            }
        }
    }
}

template <class P>
__global__ __launch_bounds__(256, 2) void k_mlp_small(SmallArgs a) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(16))) float sm_lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, j = lane & 31u, h = lane >> 5;
    constexpr int NLc = (P::BWD ? P::NL : P::NL);          // (the last backward chain layer is skipped at run time when grad_in is not wanted)

    // ---- the chain's matrices, permuted into A-operand order ----
    {
        constexpr int NT1 = P::tiles(1);
#pragma unroll 4
        for (int s = 0; s < P::S1; ++s)
            for (uint32_t e = tid; e < (uint32_t)NT1 * 64u; e += 256u) {
                const uint32_t t = e >> 6, l = e & 63u, i = 32u * t + (l & 31u), k = (uint32_t)s + (l >> 5) * (uint32_t)P::S1;
                sm_lds[P::lds_off(1) + (s * NT1) * 64 + e] = (i < (uint32_t)P::c(1) && k < (uint32_t)P::c(0)) ? chain_weight<P>(a, 1, i, k) : 0.0f;
            }
    }
    auto fill_layer = [&](auto Lc) {
        constexpr int L = decltype(Lc)::value;
        constexpr int NT = P::tiles(L);
#pragma unroll
        for (int q = 0; q < P::tiles(L - 1); ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!P::step_ok(L, q, r)) continue;
                const int s = P::step_index(L, q, r);
                for (uint32_t e = tid; e < (uint32_t)NT * 64u; e += 256u) {
                    const uint32_t t = e >> 6, l = e & 63u, i = 32u * t + (l & 31u);
                    const uint32_t k = 32u * (uint32_t)q + 8u * (uint32_t)(r >> 2) + (uint32_t)(r & 3) + 4u * (l >> 5);
                    sm_lds[P::lds_off(L) + (s * NT) * 64 + e] = (i < (uint32_t)P::c(L) && k < (uint32_t)P::c(L - 1)) ? chain_weight<P>(a, L, i, k) : 0.0f;
                }
            }
    };
    if constexpr (NLc >= 2) fill_layer(std::integral_constant<int, 2>{});
    if constexpr (NLc >= 3) fill_layer(std::integral_constant<int, 3>{});
    if constexpr (NLc >= 4) fill_layer(std::integral_constant<int, 4>{});
    __syncthreads();

    const uint32_t ntiles = (a.N + 63u) / 64u;
    for (uint32_t tile = blockIdx.x * 4u + wave; tile < ntiles; tile += gridDim.x * 4u) {
        uint32_t row[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) row[jt] = tile * 64u + 32u * (uint32_t)jt + j;

        // ---- chain layer 1: B operands from memory (lane (j, h) supplies inputs h*S1 + s of its row) ----
        constexpr int C0 = P::c(0), S1 = P::S1;
        float xin[2][S1];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const bool ok = row[jt] < a.N;
            const uint32_t k0 = h * (uint32_t)S1;
            if (a.in) {
                const float *src = a.in + (size_t)(ok ? row[jt] : 0u) * C0 + k0;
                if constexpr (C0 % 4 == 0 && S1 % 4 == 0) {
#pragma unroll
                    for (int s4 = 0; s4 < S1 / 4; ++s4) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * s4);
                        xin[jt][4 * s4] = ok ? v.x : 0.0f; xin[jt][4 * s4 + 1] = ok ? v.y : 0.0f;
                        xin[jt][4 * s4 + 2] = ok ? v.z : 0.0f; xin[jt][4 * s4 + 3] = ok ? v.w : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < S1; ++s) {
                        const bool kv = ok && k0 + (uint32_t)s < (uint32_t)C0;
                        const float v = src[kv ? s : -(int)k0];            // (an address inside the row either way)
                        xin[jt][s] = kv ? v : 0.0f;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < S1; ++s) xin[jt][s] = 0.0f;
            }
            if constexpr (P::BWD) {
                // gradient that arrives through the activated output, folded into the raw output's gradient
                if (a.act == SN_SMALL_ACT_TRUNC_EXP0 && a.aux_grad) {
                    if (h == 0u && ok) {      // feature 0: half 0, step 0.  activation.py:13-17: g * exp(clamp(x, -15, 15))
                        const float x = a.raw[(size_t)row[jt] * C0];
                        xin[jt][0] = __builtin_fmaf(a.aux_grad[row[jt]], expf_det(__builtin_amdgcn_fmed3f(x, -15.0f, 15.0f)), xin[jt][0]);
                    }
                } else if (a.act == SN_SMALL_ACT_SIGMOID_BG && a.aux_grad) {
                    float gsum = 0.0f;
#pragma unroll
                    for (int s = 0; s < S1; ++s) {
                        const uint32_t k = k0 + (uint32_t)s;
                        if (ok && k < (uint32_t)C0) {
                            const float g = a.aux_grad[(size_t)row[jt] * C0 + k], sg = sigmoid_det(a.raw[(size_t)row[jt] * C0 + k]);
                            xin[jt][s] = __builtin_fmaf(g, sg * (1.0f - sg), xin[jt][s]);
                            gsum += g;
                        }
                    }
                    if (a.aux2_out) {         // d/d weights_sum of (1 - weights_sum) * bg summed over the channels
                        const float other = __shfl_xor(gsum, 32);
                        if (h == 0u && ok) a.aux2_out[row[jt]] = -a.bg * (gsum + other);
                    }
                }
                if (a.aux_out && ok) {        // the combined gradient of the last pre-activation (the last layer's weight gradient reads it)
#pragma unroll
                    for (int s = 0; s < S1; ++s)
                        if (k0 + (uint32_t)s < (uint32_t)C0) a.aux_out[(size_t)row[jt] * C0 + k0 + s] = xin[jt][s];
                }
            }
        }
        floatx16 accA[2][2], accB[2][2];
        {
            constexpr int NT = P::tiles(1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) accA[t][jt] = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < S1; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float av = sm_lds[P::lds_off(1) + (s * NT + t) * 64 + lane];
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) accA[t][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xin[jt][s], accA[t][jt], 0, 0, 0);
                }
        }
        if constexpr (P::NL == 1) {
            if (!P::BWD || a.out) layer_finish<P, 1>(a, row, h, accA);
        } else {
            layer_finish<P, 1>(a, row, h, accA);
            if constexpr (P::NL == 2) {
                if (!P::BWD || a.out) { layer_from_regs<P, 2>(sm_lds, lane, accA, accB); layer_finish<P, 2>(a, row, h, accB); }
            } else {
                layer_from_regs<P, 2>(sm_lds, lane, accA, accB);
                layer_finish<P, 2>(a, row, h, accB);
                if constexpr (P::NL == 3) {
                    if (!P::BWD || a.out) { layer_from_regs<P, 3>(sm_lds, lane, accB, accA); layer_finish<P, 3>(a, row, h, accA); }
                } else {
                    layer_from_regs<P, 3>(sm_lds, lane, accB, accA);
                    layer_finish<P, 3>(a, row, h, accA);
                    if (!P::BWD || a.out) { layer_from_regs<P, 4>(sm_lds, lane, accA, accB); layer_finish<P, 4>(a, row, h, accB); }
                }
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------
// The shapes this build instantiates: the reference network's three perceptrons (network.py:94, 98, 143) and the two of BASELINE configs[0].
#define SN_SMALL_SHAPES(X) \
    X(2, 10, 16, 1, 0, 0)  \
    X(3, 32, 64, 64, 16, 0) \
    X(3, 31, 32, 32, 3, 0) \
    X(2, 16, 32, 16, 0, 0) \
    X(2, 31, 32, 3, 0, 0)

static bool small_shape(const sn_mlp_desc *m, int nl, int d0, int d1, int d2, int d3, int d4) {
    const int d[5] = {d0, d1, d2, d3, d4};
    if ((int)m->num_layers != nl) return false;
    for (int i = 0; i <= nl; ++i)
        if ((int)m->dims[i] != d[i]) return false;
    return true;
}

static int small_check(const sn_mlp_desc *m, const char *who) {
    SN_REQUIRE(m, "%s: NULL descriptor", who);
    SN_REQUIRE(m->num_layers >= 1 && m->num_layers <= (uint32_t)SM_MAXL, "%s: 1..%d layers (got %u)", who, SM_MAXL, m->num_layers);
    SN_REQUIRE(m->activation == 0 && m->skip_mask == 0, "%s: ReLU perceptrons without skip layers only", who);
    for (uint32_t l = 0; l < m->num_layers; ++l) {
        SN_REQUIRE(m->weight[l] != nullptr, "%s: weight[%u] is NULL", who, l);
        SN_REQUIRE(m->bias[l] == nullptr, "%s: bias-free layers only (network.py:94,98,143 pass bias=False)", who);
    }
    return SN_OK;
}

template <class P>
static int small_launch(const SmallArgs &a, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t lds = (size_t)P::lds_floats * sizeof(float);
    if (!attr_set && lds > 48 * 1024) {
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_small<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const uint32_t ntiles = (a.N + 63u) / 64u;
    uint32_t blocks = div_up(ntiles, 4);
    if (blocks > 512u) blocks = 512u;                      // persistent: two workgroups per CU walk the tiles
    hipLaunchKernelGGL((k_mlp_small<P>), dim3(blocks), dim3(256), lds, st, a);
    SN_LAUNCH_CHECK("k_mlp_small");
    return SN_OK;
}

}  // namespace sn

extern "C" {

int sn_mlp_small_supported(const sn_mlp_desc *mlp) {
    if (!mlp || mlp->activation != 0 || mlp->skip_mask != 0) return 0;
    for (uint32_t l = 0; l < mlp->num_layers && l < SN_MAX_LAYERS; ++l)
        if (mlp->bias[l]) return 0;
#define X(nl, d0, d1, d2, d3, d4) if (sn::small_shape(mlp, nl, d0, d1, d2, d3, d4)) return 1;
    SN_SMALL_SHAPES(X)
#undef X
    return 0;
}

int sn_mlp_small_forward_train(const sn_mlp_desc *mlp, const float *x, uint32_t N, float *const *hidden, float *out, int32_t act,
                               const float *aux_in, float bg, float *aux_out, sn_stream_t stream) {
    using namespace sn;
    if (int rc = small_check(mlp, "mlp_small_forward_train")) return rc;
    if (N == 0) return SN_OK;                      // empty batch: nothing to launch, pointers may be NULL
    SN_REQUIRE(x && out, "mlp_small_forward_train: NULL pointer");
    SN_REQUIRE(act >= SN_SMALL_ACT_NONE && act <= SN_SMALL_ACT_SIGMOID_BG, "mlp_small_forward_train: unknown output activation %d", act);
    SN_REQUIRE(act == SN_SMALL_ACT_NONE || aux_out, "mlp_small_forward_train: the activated output needs a destination");
    SN_REQUIRE(act != SN_SMALL_ACT_SIGMOID_BG || (aux_in && mlp->dims[mlp->num_layers] <= 4), "mlp_small_forward_train: sigmoid + background needs weights_sum and <= 4 outputs");
    const uint32_t nl = mlp->num_layers;
    SN_REQUIRE(nl == 1 || hidden, "mlp_small_forward_train: hidden is NULL");
    if (N == 0) return SN_OK;
    SmallArgs a{};
    for (uint32_t l = 0; l < nl; ++l) a.w[l] = mlp->weight[l];
    for (uint32_t l = 0; l + 1 < nl; ++l) {
        SN_REQUIRE(hidden[l] && table_aligned(hidden[l]), "mlp_small_forward_train: hidden[%u] NULL or not 16-byte aligned", l);
        a.store[l] = hidden[l];
    }
    SN_REQUIRE(table_aligned(x) && table_aligned(out), "mlp_small_forward_train: x / out must be 16-byte aligned");
    a.in = x; a.out = out; a.aux_in = aux_in; a.aux_out = aux_out; a.N = N; a.act = act; a.bg = bg;
#define X(nl_, d0, d1, d2, d3, d4) if (small_shape(mlp, nl_, d0, d1, d2, d3, d4)) return small_launch<Chain<false, nl_, d0, d1, d2, d3, d4>>(a, (hipStream_t)stream);
    SN_SMALL_SHAPES(X)
#undef X
    set_error("mlp_small_forward_train: layer widths not instantiated in this build (see SN_SMALL_SHAPES)");
    return SN_ERR_UNSUPPORTED;
}

int sn_mlp_small_backward(const sn_mlp_desc *mlp, const float *grad_out, const float *grad_aux, int32_t act, const float *out_raw, float bg,
                          const float *const *hidden, uint32_t N, float *grad_in, float *const *grad_hidden, float *grad_last, float *grad_aux_in,
                          sn_stream_t stream) {
    using namespace sn;
    if (int rc = small_check(mlp, "mlp_small_backward")) return rc;
    if (N == 0) return SN_OK;
    SN_REQUIRE(act >= SN_SMALL_ACT_NONE && act <= SN_SMALL_ACT_SIGMOID_BG, "mlp_small_backward: unknown output activation %d", act);
    SN_REQUIRE(grad_out || grad_aux, "mlp_small_backward: no incoming gradient");
    SN_REQUIRE(!grad_aux || (act != SN_SMALL_ACT_NONE && out_raw), "mlp_small_backward: grad_aux needs the activation and the forward's raw output");
    SN_REQUIRE(grad_last, "mlp_small_backward: grad_last is NULL");
    const uint32_t nl = mlp->num_layers;
    SN_REQUIRE(nl == 1 || (hidden && grad_hidden), "mlp_small_backward: hidden / grad_hidden is NULL");
    if (N == 0) return SN_OK;
    SmallArgs a{};
    for (uint32_t l = 0; l < nl; ++l) a.w[l] = mlp->weight[l];
    for (uint32_t l = 0; l + 1 < nl; ++l) {
        SN_REQUIRE(hidden[l] && grad_hidden[l] && table_aligned(hidden[l]) && table_aligned(grad_hidden[l]),
                   "mlp_small_backward: hidden[%u] / grad_hidden[%u] NULL or not 16-byte aligned", l, l);
        a.mask[l] = hidden[l];
        a.store[l] = grad_hidden[l];
    }
    SN_REQUIRE((!grad_out || table_aligned(grad_out)) && (!grad_in || table_aligned(grad_in)), "mlp_small_backward: grad_out / grad_in must be 16-byte aligned");
    a.in = grad_out; a.aux_grad = grad_aux; a.raw = out_raw; a.out = grad_in; a.aux_out = grad_last; a.aux2_out = grad_aux_in;
    a.N = N; a.act = act; a.bg = bg;
    // backward chain = the widths reversed
#define X(nl_, d0, d1, d2, d3, d4)                                                                                              \
    if (small_shape(mlp, nl_, d0, d1, d2, d3, d4)) {                                                                            \
        constexpr int d[5] = {d0, d1, d2, d3, d4};                                                                              \
        return small_launch<Chain<true, nl_, d[nl_], d[nl_ - 1], nl_ >= 2 ? d[nl_ - 2] : 0, nl_ >= 3 ? d[nl_ - 3] : 0, nl_ >= 4 ? d[nl_ - 4] : 0>>(a, (hipStream_t)stream); \
    }
    SN_SMALL_SHAPES(X)
#undef X
    set_error("mlp_small_backward: layer widths not instantiated in this build (see SN_SMALL_SHAPES)");
    return SN_ERR_UNSUPPORTED;
}

}  // extern "C"
