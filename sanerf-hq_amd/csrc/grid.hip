// grid.hip — multiresolution hash / tiled grid encoder for gfx950.
//
// Replaces gridencoder/src/gridencoder.cu (kernel_grid, kernel_grid_backward,
// kernel_input_backward, kernel_grad_tv, kernel_grad_wd) behind the C ABI of
// include/sanerf_hip.h.  Stand-alone operator: one lane per (sample, level), levels on
// blockIdx.y so that one level's table stays hot in the XCD's L2 while its blocks run.
// The fused renderer (render.hip) does not call these kernels; it inlines the same
// arithmetic (sn_common.h) next to the MLPs.
#include "sn_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace sn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return SN_ERR_HIP;
}

uint32_t level_resolution(uint32_t level, float S, uint32_t H) {
    // gridencoder.cu:133, every operation in fp32 on the host, once per level
    const float e = exp2f((float)level * S);
    return (uint32_t)ceilf(e * (float)H);
}

int build_grid_levels(GridLevels *g, const int32_t *offsets, uint32_t D, uint32_t C, uint32_t L,
                      float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    SN_REQUIRE(offsets != nullptr, "grid: offsets (host) is NULL");
    SN_REQUIRE(L >= 1 && L <= SN_MAX_LEVELS, "grid: L=%u outside 1..%d", L, SN_MAX_LEVELS);
    SN_REQUIRE(D >= 2 && D <= 5, "GridEncoding: D must be 2, 3, 4 or 5.");
    SN_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8 || C == 16 || C == 32, "GridEncoding: C must be 1, 2, 4, 8, 16 or 32.");
    memset(g, 0, sizeof(*g));
    g->L = L; g->D = D; g->C = C;
    g->gridtype = gridtype; g->align_corners = align_corners ? 1u : 0u; g->interp = interp;
    for (uint32_t l = 0; l < L; ++l) {
        SN_REQUIRE(offsets[l + 1] > offsets[l], "grid: offsets not increasing at level %u", l);
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
        const uint32_t res = level_resolution(l, S, H);
        uint32_t stride = 1, nd = 0;
        for (uint32_t d = 0; d < D && stride <= size; ++d) { stride *= res; ++nd; }   // gridencoder.cu:66-70
        const bool hashed = (gridtype == 0 && stride > size);                           // gridencoder.cu:74
        uint32_t modk;
        if (!hashed && stride <= size) modk = 0;            // dense walk covers all dims it took and fits
        else if ((size & (size - 1)) == 0) modk = 1;
        else modk = 2;
        g->res[l] = res; g->size[l] = size; g->off[l] = (uint32_t)offsets[l];
        g->mode[l] = (hashed ? 1u : 0u) | (modk << 1) | (nd << 4);
    }
    return SN_OK;
}

// ------------------------------------------------------------------------------------------
// forward (+ optional dy_dx): gridencoder.cu:82-249
// ------------------------------------------------------------------------------------------
template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_forward(const float *__restrict__ inputs, const T *__restrict__ table,
                                                      float *__restrict__ outputs, float *__restrict__ dy_dx,
                                                      uint32_t B, GridLevels g, int layout) {
    SN_POISON_ALL();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const uint32_t L = g.L;
    float *out = layout == SN_LAYOUT_LBC ? outputs + ((size_t)level * B + b) * C : outputs + ((size_t)b * L + level) * C;
    float *dd = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;

    float x01[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        x01[d] = inputs[(size_t)b * D + d];
        if (x01[d] < 0 || x01[d] > 1) oob = true;
    }
    if (oob) {  // gridencoder.cu:113-130
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) out[c] = 0;
        if (dd) for (uint32_t i = 0; i < D * C; ++i) dd[i] = 0;
        return;
    }
    const uint32_t res = g.res[level], size = g.size[level], mode = g.mode[level];
    const T *tab = table + (size_t)g.off[level] * C;
    float pos[D], deriv[D];
    uint32_t cell[D];
    grid_locate<D>(x01, res, g.align_corners != 0, g.interp, pos, deriv, cell);

    float acc[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) acc[c] = 0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1;
        uint32_t p[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; p[d] = cell[d]; }
            else { w *= pos[d]; p[d] = umin(cell[d] + 1, res - 1); }
        }
        const T *row = tab + (size_t)grid_row<D>(p, res, size, mode) * C;
        float v[C];
        load_row<T, (int)C>(row, v);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] = __builtin_fmaf(w, v[c], acc[c]);
    }
    if constexpr (C % 4 == 0) {   // 16-byte stores (outputs base is 16-byte aligned, checked on the host)
#pragma unroll
        for (uint32_t q = 0; q < C / 4; ++q)
            reinterpret_cast<float4 *>(out)[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2 *>(out) = make_float2(acc[0], acc[1]);
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) out[c] = acc[c];
    }

    if (dd) {  // gridencoder.cu:205-248
#pragma unroll
        for (uint32_t gd = 0; gd < D; ++gd) {
            float rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) rg[c] = 0;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
                float w = (float)(g.align_corners ? res - 1 : res);
                uint32_t p[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; ++nd) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; p[d] = cell[d]; }
                    else { w *= pos[d]; p[d] = umin(cell[d] + 1, res - 1); }
                }
                p[gd] = cell[gd];
                const T *rl = tab + (size_t)grid_row<D>(p, res, size, mode) * C;
                p[gd] = umin(cell[gd] + 1, res - 1);
                const T *rr = tab + (size_t)grid_row<D>(p, res, size, mode) * C;
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) {
                    const float diff = table_ld<T>(rr + c) - table_ld<T>(rl + c);
                    rg[c] = __builtin_fmaf(w * diff, deriv[gd], rg[c]);
                }
            }
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) dd[gd * C + c] = rg[c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: scatter w*grad to the 2^D corners (gridencoder.cu:252-349) with hardware fp32 atomics
// ------------------------------------------------------------------------------------------
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_backward(const float *__restrict__ grad, const float *__restrict__ inputs,
                                                       float *__restrict__ grad_table, uint32_t B, GridLevels g, int layout) {
    SN_POISON_ALL();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float x01[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        x01[d] = inputs[(size_t)b * D + d];
        if (x01[d] < 0 || x01[d] > 1) return;  // grad_table is zero-initialised
    }
    const float *gsrc = layout == SN_LAYOUT_LBC ? grad + ((size_t)level * B + b) * C : grad + ((size_t)b * g.L + level) * C;
    float gc[C];
    bool any = false;
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) { gc[c] = gsrc[c]; any |= (gc[c] != 0.0f); }
    if (!any) return;  // adding +0 to a table is a no-op: skip the atomics
    const uint32_t res = g.res[level], size = g.size[level], mode = g.mode[level];
    float *gt = grad_table + (size_t)g.off[level] * C;
    float pos[D], deriv[D];
    uint32_t cell[D];
    grid_locate<D>(x01, res, g.align_corners != 0, g.interp, pos, deriv, cell);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1;
        uint32_t p[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; p[d] = cell[d]; }
            else { w *= pos[d]; p[d] = umin(cell[d] + 1, res - 1); }
        }
        float *row = gt + (size_t)grid_row<D>(p, res, size, mode) * C;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(row + c, w * gc[c]);
    }
}

// gridencoder.cu:352-378
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_input_backward(const float *__restrict__ grad, const float *__restrict__ dy_dx,
                                                             float *__restrict__ grad_inputs, uint32_t B, uint32_t L, int layout) {
    SN_POISON_ALL();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *dd = dy_dx + (size_t)b * L * D * C;
    float r = 0;
    for (uint32_t l = 0; l < L; ++l) {
        const float *gs = layout == SN_LAYOUT_LBC ? grad + ((size_t)l * B + b) * C : grad + ((size_t)b * L + l) * C;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) r = __builtin_fmaf(gs[c], dd[(size_t)l * D * C + d * C + c], r);
    }
    grad_inputs[t] = r;
}

// ------------------------------------------------------------------------------------------
// TV regulariser gradient (gridencoder.cu:525-631) and level-wise weight decay (:670-703)
// ------------------------------------------------------------------------------------------
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_tv(const float *__restrict__ inputs, const float *__restrict__ table,
                                                 float *__restrict__ grad, float weight, uint32_t B, GridLevels g) {
    SN_POISON_ALL();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float x01[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        x01[d] = inputs[(size_t)b * D + d];
        if (x01[d] < 0 || x01[d] > 1) return;
    }
    // Neighbour coordinates can be one past the last vertex (the reference's guard at gridencoder.cu:593 is
    // always true); its unconditional `% hashmap_size` (gridencoder.cu:78) wraps them, so force the generic modulo.
    const uint32_t res = g.res[level], size = g.size[level], mode = (g.mode[level] & ~6u) | (2u << 1);
    const float *tab = table + (size_t)g.off[level] * C;
    float *gt = grad + (size_t)g.off[level] * C;
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        if (g.align_corners) {
            const float p = x01[d] * (float)(res - 1);
            pg[d] = umin((uint32_t)floorf(p), res - 2);
        } else {
            float p = __builtin_fmaf(x01[d], (float)res, -0.5f);
            p = fminf(fmaxf(p, 0.0f), (float)(res - 1));
            pg[d] = (uint32_t)floorf(p);
        }
    }
    float results[C], idelta[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) { results[c] = 0; idelta[c] = 0; }
    const size_t index = (size_t)grid_row<D>(pg, res, size, mode) * C;
    const float w = weight / (2 * D);
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const uint32_t cur = pg[d];
        if (cur < res) {
            pg[d] = cur + 1;
            const size_t ir = (size_t)grid_row<D>(pg, res, size, mode) * C;
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) {
                const float gv = tab[index + c] - tab[ir + c];
                results[c] += gv; idelta[c] = __builtin_fmaf(gv, gv, idelta[c]);
            }
        }
        if (cur > 0) {
            pg[d] = cur - 1;
            const size_t il = (size_t)grid_row<D>(pg, res, size, mode) * C;
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) {
                const float gv = tab[index + c] - tab[il + c];
                results[c] += gv; idelta[c] = __builtin_fmaf(gv, gv, idelta[c]);
            }
        }
        pg[d] = cur;
    }
#pragma unroll
    for (uint32_t c = 0; c < C; ++c)
        unsafeAtomicAdd(gt + index + c, w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f)));
}

__global__ __launch_bounds__(256) void k_grid_wd(const float *__restrict__ table, float *__restrict__ grad, float weight,
                                                 uint32_t rows, uint32_t C, GridLevels g) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)rows * C) return;
    const uint32_t n = (uint32_t)(t / C);
    uint32_t level = 0, l = 0, r = g.L;
    while (l < r) {
        const uint32_t m = (l + r) / 2;
        if (g.off[m] <= n) { level = m; l = m + 1; } else { r = m; }
    }
    grad[t] += 2 * weight * table[t] / (float)g.size[level];
}

// ------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------
#define SN_GRID_DISPATCH_DC(D, C, CALL)                                                         \
    do {                                                                                        \
        bool done__ = true;                                                                     \
        if (D == 3 && C == 1) { CALL(3, 1); }                                                   \
        else if (D == 3 && C == 2) { CALL(3, 2); }                                              \
        else if (D == 3 && C == 4) { CALL(3, 4); }                                              \
        else if (D == 3 && C == 8) { CALL(3, 8); }                                              \
        else if (D == 3 && C == 16) { CALL(3, 16); }                                            \
        else if (D == 3 && C == 32) { CALL(3, 32); }                                            \
        else if (D == 2 && C == 1) { CALL(2, 1); }                                              \
        else if (D == 2 && C == 2) { CALL(2, 2); }                                              \
        else if (D == 2 && C == 4) { CALL(2, 4); }                                              \
        else if (D == 2 && C == 8) { CALL(2, 8); }                                              \
        else if (D == 2 && C == 16) { CALL(2, 16); }                                            \
        else if (D == 2 && C == 32) { CALL(2, 32); }                                            \
        else if (D == 4 && C == 1) { CALL(4, 1); }                                              \
        else if (D == 4 && C == 2) { CALL(4, 2); }                                              \
        else if (D == 4 && C == 4) { CALL(4, 4); }                                              \
        else if (D == 4 && C == 8) { CALL(4, 8); }                                              \
        else if (D == 4 && C == 16) { CALL(4, 16); }                                            \
        else if (D == 4 && C == 32) { CALL(4, 32); }                                            \
        else if (D == 5 && C == 1) { CALL(5, 1); }                                              \
        else if (D == 5 && C == 2) { CALL(5, 2); }                                              \
        else if (D == 5 && C == 4) { CALL(5, 4); }                                              \
        else if (D == 5 && C == 8) { CALL(5, 8); }                                              \
        else if (D == 5 && C == 16) { CALL(5, 16); }                                            \
        else if (D == 5 && C == 32) { CALL(5, 32); }                                            \
        else done__ = false;                                                                    \
        if (!done__) {                                                                          \
            ::sn::set_error("grid: D=%u C=%u is outside the reference's instantiations (gridencoder.cu:385-411: D in 2..5, C in 1,2,4,8,16,32)", D, C); \
            return SN_ERR_UNSUPPORTED;                                                          \
        }                                                                                       \
    } while (0)


// Forward for the [B, L*C] layout (what the MLPs consume), D = 3, no dy_dx: in k_grid_forward a lane writes its C floats
// at a stride of L*C floats -- 8 or 32 bytes of every 128/512 -- so each store instruction of a wave touches 64 partial
// lines (3.6 ms for the 5.12 M samples of a 400x400 mask render, 2.6 GB of output).  Here a workgroup owns 64
// consecutive samples and ALL levels: wave w evaluates levels w, w+4, ... (still one level per wave instruction, so the
// lanes of a gather stay in one level's table), rows are parked in LDS (row stride padded by 4 floats: conflict-free
// 16-byte stores) and the 64 x L*C block -- contiguous in memory -- leaves with fully coalesced 16-byte stores.
// `extra` [B, E] (may be NULL, E = 0): appended to every row, i.e. outputs = cat([grid(inputs), extra], -1) -- the mask
// head's MLP input (renderer.py:380: cat([m_grid(xyz), geo_feat])) without the 2 ms concatenation pass at 400x400.
template <typename T, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_forward_rows(const float *__restrict__ inputs, const T *__restrict__ table,
                                                           float *__restrict__ outputs, uint32_t B, GridLevels g, uint32_t max_level,
                                                           const float *__restrict__ extra, uint32_t E) {
    SN_POISON_ALL();
    constexpr uint32_t D = 3;
    extern __shared__ __attribute__((aligned(16))) float rows[];           // [64][L*C + 4]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t L = g.L, row_f = L * C, stride = row_f + 4u;
    const uint32_t b0 = blockIdx.x * 64u, b_raw = b0 + lane;
    const uint32_t b = b_raw < B ? b_raw : B - 1u;
    float x01[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        x01[d] = inputs[(size_t)b * D + d];
        if (x01[d] < 0 || x01[d] > 1) oob = true;
    }
    for (uint32_t level = wave; level < L; level += 4u) {
        float acc[C];
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] = 0;
        if (!oob && level < max_level) {                                    // gridencoder.cu:113-130: zeros outside [0,1]
            const uint32_t res = g.res[level], size = g.size[level], mode = g.mode[level];
            const T *tab = table + (size_t)g.off[level] * C;
            float pos[D], deriv[D];
            uint32_t cell[D];
            grid_locate<D>(x01, res, g.align_corners != 0, g.interp, pos, deriv, cell);
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << D); ++idx) {
                float w = 1;
                uint32_t p[D];
#pragma unroll
                for (uint32_t d = 0; d < D; ++d) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; p[d] = cell[d]; }
                    else { w *= pos[d]; p[d] = umin(cell[d] + 1, res - 1); }
                }
                float v[C];
                load_row<T, (int)C>(tab + (size_t)grid_row<D>(p, res, size, mode) * C, v);
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) acc[c] = __builtin_fmaf(w, v[c], acc[c]);
            }
        }
        float *dst = rows + lane * stride + level * C;
        if constexpr (C % 4 == 0) {
#pragma unroll
            for (uint32_t q = 0; q < C / 4; ++q)
                reinterpret_cast<float4 *>(dst)[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) dst[c] = acc[c];
        }
    }
    __syncthreads();
    // the block's 64 rows are one contiguous span of the output
    const uint32_t nrows = B - b0 < 64u ? B - b0 : 64u;
    if (E == 0u) {
        const uint32_t q_per_row = row_f / 4u;                              // row_f % 4 == 0 (checked on the host)
        float4 *out4 = reinterpret_cast<float4 *>(outputs + (size_t)b0 * row_f);
        for (uint32_t q = threadIdx.x; q < nrows * q_per_row; q += 256u) {
            const uint32_t r = q / q_per_row, cq = q - r * q_per_row;
            out4[q] = *reinterpret_cast<const float4 *>(rows + r * stride + 4u * cq);
        }
    } else {                                                                // rows of row_f + E floats: 4-byte stores, still one contiguous span
        const uint32_t wide = row_f + E;
        float *out = outputs + (size_t)b0 * wide;
        const float *ex = extra + (size_t)b0 * E;
        for (uint32_t q = threadIdx.x; q < nrows * wide; q += 256u) {
            const uint32_t r = q / wide, c = q - r * wide;
            out[q] = c < row_f ? rows[r * stride + c] : ex[r * E + (c - row_f)];
        }
    }
}

}  // namespace sn

using namespace sn;

extern "C" {

int sn_abi_version(void) { return SN_ABI_VERSION; }
const char *sn_last_error(void) { return sn::g_err; }

int sn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return sn::hip_fail(e, "hipGetDeviceCount");
    return n;
}

int sn_grid_encode_forward(const float *inputs, const void *embeddings, int table_dtype,
                           const int32_t *offsets_host, float *outputs,
                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                           float S, uint32_t H, float *dy_dx,
                           uint32_t gridtype, int align_corners, uint32_t interp,
                           int layout, sn_stream_t stream) {
    if (B == 0) return SN_OK;   // empty batch: nothing to launch, pointers may be NULL
    SN_REQUIRE(inputs && embeddings && outputs, "grid_encode_forward: inputs/embeddings/outputs must be device pointers");
    SN_REQUIRE(table_dtype == SN_F32 || table_dtype == SN_F16, "grid_encode_forward: embeddings must be float32 or float16");
    SN_REQUIRE(layout == SN_LAYOUT_LBC || layout == SN_LAYOUT_BLC, "grid_encode_forward: bad layout %d", layout);
    SN_REQUIRE(table_aligned(embeddings) && table_aligned(outputs),
               "grid_encode_forward: embeddings and outputs must be 16-byte aligned (rows are read/written with vector accesses)");
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, D, C, L, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    if (max_level > L) max_level = L;
    if (B == 0 || max_level == 0) return SN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (layout == SN_LAYOUT_BLC && D == 3 && dy_dx == nullptr && (L * C) % 4 == 0 && (C == 2 || C == 4 || C == 8) &&
        (size_t)64 * (L * C + 4) * sizeof(float) <= 64 * 1024) {
        // rows assembled in LDS, coalesced stores (levels >= max_level are written as zeros)
        const size_t lds = (size_t)64 * (L * C + 4) * sizeof(float);
        const dim3 gr(div_up(B, 64));
#define CALL_ROWS(CC)                                                                                                               \
        do {                                                                                                                        \
            if (table_dtype == SN_F32) hipLaunchKernelGGL((k_grid_forward_rows<float, CC>), gr, dim3(256), lds, st, inputs,         \
                                                          (const float *)embeddings, outputs, B, g, max_level, nullptr, 0u);        \
            else hipLaunchKernelGGL((k_grid_forward_rows<__half, CC>), gr, dim3(256), lds, st, inputs, (const __half *)embeddings,   \
                                    outputs, B, g, max_level, nullptr, 0u);                                                         \
        } while (0)
        if (C == 2) CALL_ROWS(2); else if (C == 4) CALL_ROWS(4); else CALL_ROWS(8);
#undef CALL_ROWS
        SN_LAUNCH_CHECK("k_grid_forward_rows");
        return SN_OK;
    }
    const dim3 grid(div_up(B, 256), max_level), block(256);
#define CALL_FWD(DD, CC)                                                                                          \
    if (table_dtype == SN_F32)                                                                                    \
        hipLaunchKernelGGL((k_grid_forward<float, DD, CC>), grid, block, 0, st, inputs, (const float *)embeddings, \
                           outputs, dy_dx, B, g, layout);                                                         \
    else                                                                                                          \
        hipLaunchKernelGGL((k_grid_forward<__half, DD, CC>), grid, block, 0, st, inputs, (const __half *)embeddings, \
                           outputs, dy_dx, B, g, layout)
    SN_GRID_DISPATCH_DC(D, C, CALL_FWD);
#undef CALL_FWD
    SN_LAUNCH_CHECK("k_grid_forward");
    return SN_OK;
}

int sn_grid_encode_forward_cat(const float *inputs, const void *embeddings, int table_dtype, const int32_t *offsets_host,
                               const float *extra, uint32_t E, float *outputs, uint32_t B, uint32_t C, uint32_t L,
                               float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, sn_stream_t stream) {
    if (B == 0) return SN_OK;
    SN_REQUIRE(inputs && embeddings && outputs && extra && E >= 1, "grid_encode_forward_cat: NULL pointer / no extra columns");
    SN_REQUIRE(table_dtype == SN_F32 || table_dtype == SN_F16, "grid_encode_forward_cat: embeddings must be float32 or float16");
    SN_REQUIRE(C == 2 || C == 4 || C == 8, "grid_encode_forward_cat: level_dim must be 2, 4 or 8 (got %u)", C);
    SN_REQUIRE(table_aligned(embeddings), "grid_encode_forward_cat: embeddings must be 16-byte aligned");
    const size_t lds = (size_t)64 * (L * C + 4) * sizeof(float);
    SN_REQUIRE(lds <= 64 * 1024, "grid_encode_forward_cat: %u levels x %u features do not fit the row tile", L, C);
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, 3, C, L, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gr(div_up(B, 64));
#define CALL_ROWS(CC)                                                                                                               \
    do {                                                                                                                            \
        if (table_dtype == SN_F32) hipLaunchKernelGGL((k_grid_forward_rows<float, CC>), gr, dim3(256), lds, st, inputs,             \
                                                      (const float *)embeddings, outputs, B, g, L, extra, E);                      \
        else hipLaunchKernelGGL((k_grid_forward_rows<__half, CC>), gr, dim3(256), lds, st, inputs, (const __half *)embeddings,       \
                                outputs, B, g, L, extra, E);                                                                        \
    } while (0)
    if (C == 2) CALL_ROWS(2); else if (C == 4) CALL_ROWS(4); else CALL_ROWS(8);
#undef CALL_ROWS
    SN_LAUNCH_CHECK("k_grid_forward_rows");
    return SN_OK;
}

int sn_grid_encode_backward(const float *grad, const float *inputs, const void *embeddings, int table_dtype,
                            const int32_t *offsets_host, float *grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                            float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                            uint32_t gridtype, int align_corners, uint32_t interp,
                            int layout, sn_stream_t stream) {
    (void)embeddings; (void)table_dtype;
    if (B == 0) return SN_OK;
    SN_REQUIRE(grad && inputs && grad_embeddings, "grid_encode_backward: grad/inputs/grad_embeddings must be device pointers");
    SN_REQUIRE(layout == SN_LAYOUT_LBC || layout == SN_LAYOUT_BLC, "grid_encode_backward: bad layout %d", layout);
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, D, C, L, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    if (max_level > L) max_level = L;
    if (B == 0 || max_level == 0) return SN_OK;
    const dim3 grid(div_up(B, 256), max_level), block(256);
    hipStream_t st = (hipStream_t)stream;
#define CALL_BWD(DD, CC) hipLaunchKernelGGL((k_grid_backward<DD, CC>), grid, block, 0, st, grad, inputs, grad_embeddings, B, g, layout)
    SN_GRID_DISPATCH_DC(D, C, CALL_BWD);
#undef CALL_BWD
    SN_LAUNCH_CHECK("k_grid_backward");
    if (dy_dx && grad_inputs) {
        const dim3 g2(div_up((uint64_t)B * D, 256));
#define CALL_IB(DD, CC) hipLaunchKernelGGL((k_grid_input_backward<DD, CC>), g2, block, 0, st, grad, dy_dx, grad_inputs, B, L, layout)
        SN_GRID_DISPATCH_DC(D, C, CALL_IB);
#undef CALL_IB
        SN_LAUNCH_CHECK("k_grid_input_backward");
    }
    return SN_OK;
}

int sn_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                            const int32_t *offsets_host, float weight,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, uint32_t gridtype, int align_corners,
                            sn_stream_t stream) {
    SN_REQUIRE(inputs && embeddings && grad, "grad_total_variation: NULL device pointer");
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, D, C, L, S, H, gridtype, align_corners, 0);
    if (rc) return rc;
    if (B == 0) return SN_OK;
    const dim3 grid(div_up(B, 256), L), block(256);
    hipStream_t st = (hipStream_t)stream;
#define CALL_TV(DD, CC) hipLaunchKernelGGL((k_grid_tv<DD, CC>), grid, block, 0, st, inputs, embeddings, grad, weight, B, g)
    SN_GRID_DISPATCH_DC(D, C, CALL_TV);
#undef CALL_TV
    SN_LAUNCH_CHECK("k_grid_tv");
    return SN_OK;
}

int sn_grad_weight_decay(const float *embeddings, float *grad, const int32_t *offsets_host,
                         float weight, uint32_t B, uint32_t C, uint32_t L, sn_stream_t stream) {
    SN_REQUIRE(embeddings && grad && offsets_host, "grad_weight_decay: NULL pointer");
    SN_REQUIRE(L >= 1 && L <= SN_MAX_LEVELS, "grad_weight_decay: L=%u outside 1..%d", L, SN_MAX_LEVELS);
    GridLevels g;
    memset(&g, 0, sizeof(g));
    g.L = L;
    for (uint32_t l = 0; l < L; ++l) { g.off[l] = (uint32_t)offsets_host[l]; g.size[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]); }
    if (B == 0) return SN_OK;
    hipLaunchKernelGGL(k_grid_wd, dim3(div_up((uint64_t)B * C, 256)), dim3(256), 0, (hipStream_t)stream, embeddings, grad, weight, B, C, g);
    SN_LAUNCH_CHECK("k_grid_wd");
    return SN_OK;
}

}  // extern "C"
