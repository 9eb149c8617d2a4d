// heads.hip — feature-head operators of the render path (gfx950).
//
//   sn_rm_grid_composite   f[n, :] = sum_j w[n,j] * grid(xyz[n,j])      (nerf/renderer.py:301-302 + 361)
//
// The reference encodes every sample of the last stage through s_grid ([N*T, L*C] floats written to
// memory, 2.6 GB at 400x400x32 with L*C = 128) and then reduces it with the sample weights.  Here the
// weighted sum runs inside the gather kernel: lane = ray, a workgroup pass covers LG levels, the
// per-level accumulators stay in registers and only [N, L*C] is written.  Table rows are fetched with
// 16-byte loads (C = 8 fp32 -> 2 per corner, fp16 -> 1).
#include "sn_common.h"

namespace sn {

// FAST: every level is "hashed with a power-of-two size" or "dense over all three dimensions", align_corners =
// False, linear interpolation (levels_fast(); every grid the reference builds) -> branch-free corner offsets, so
// the LG*8 row fetches of a sample are issued back to back.  VEC: T % 4 == 0 and 16-byte aligned inputs -> the
// per-ray weight / position rows are read four samples at a time with dwordx4 (a lane's row is contiguous but
// lanes are T*4 bytes apart, so every load instruction touches 64 lines: fewer, wider instructions).
template <typename T, int C, int LG, bool FAST, bool VEC>
__global__ __launch_bounds__(256) void k_grid_composite(const float *__restrict__ xyzs, const float *__restrict__ weights,
                                                        const T *__restrict__ table, float *__restrict__ out,
                                                        uint32_t N, uint32_t Tn, GridLevels g, float bound, float inv_den,
                                                        uint32_t tile_w, uint32_t rows) {
    SN_POISON_ALL();
    // lane -> ray: 8x8 pixel tiles per wave when the rays are an image (neighbouring lanes then share table lines)
    uint32_t n;
    bool ok;
    if (tile_w) {
        const uint32_t tiles_x = (tile_w + 15u) >> 4;
        const uint32_t by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const uint32_t py = by * 16u + (wave >> 1) * 8u + (lane >> 3), px = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
        ok = px < tile_w && py < rows;
        n = ok ? py * tile_w + px : 0u;
    } else {
        n = blockIdx.x * 256u + threadIdx.x;
        ok = n < N;
        if (!ok) n = 0u;
    }
    const uint32_t l0 = blockIdx.y * LG;
    float acc[LG][C];
#pragma unroll
    for (int i = 0; i < LG; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[i][c] = 0.0f;

    const float *wrow = weights + (size_t)n * Tn;
    const float *xrow = xyzs + (size_t)n * Tn * 3u;
    const float den = 2.0f * bound;

    auto sample = [&](float w, float px, float py, float pz) {
        float x01[3];
        const float p3[3] = {px, py, pz};
        bool oob = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float s = p3[d] + bound;                                 // grid.py:156
            x01[d] = inv_den != 0.0f ? s * inv_den : s / den;
            if (x01[d] < 0.0f || x01[d] > 1.0f) oob = true;               // gridencoder.cu:105-130 -> zeros
        }
        const float wz = oob ? 0.0f : w;     // fmaf(w, 0, acc) == fmaf(0, feat, acc) for finite operands
        if constexpr (FAST) {
            float pos[LG][3];
            float cv[LG][8][C];
#pragma unroll
            for (int i = 0; i < LG; ++i) {   // phase 1: addresses + all row fetches of the LG levels
                const uint32_t l = l0 + i;
                uint32_t cell[3], offs[8];
                locate_linear(x01, g.res[l], pos[i], cell);
                corner_offsets<-1, (uint32_t)(C * sizeof(T))>(cell, g.res[l], g.size[l], g.mode[l], offs);
                const char *tab = reinterpret_cast<const char *>(table + (size_t)g.off[l] * C);
#pragma unroll
                for (int k = 0; k < 8; ++k) load_row<T, C>(reinterpret_cast<const T *>(tab + offs[k]), cv[i][k]);
            }
#pragma unroll
            for (int i = 0; i < LG; ++i) {   // phase 2: blend (gridencoder.cu:171-192 order) + weighted accumulate
                float feat[C];
#pragma unroll
                for (int c = 0; c < C; ++c) feat[c] = 0.0f;
#pragma unroll
                for (uint32_t idx = 0; idx < 8u; ++idx) {
                    float cw = 1.0f;
#pragma unroll
                    for (uint32_t d = 0; d < 3u; ++d) cw *= (idx & (1u << d)) ? pos[i][d] : 1.0f - pos[i][d];
#pragma unroll
                    for (int c = 0; c < C; ++c) feat[c] = __builtin_fmaf(cw, cv[i][idx][c], feat[c]);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) acc[i][c] = __builtin_fmaf(wz, feat[c], acc[i][c]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < LG; ++i) {
                const uint32_t l = l0 + i;
                if (l >= g.L) break;
                const uint32_t res = g.res[l], size = g.size[l], mode = g.mode[l];
                const T *tab = table + (size_t)g.off[l] * C;
                float pos[3], deriv[3];
                uint32_t cell[3];
                grid_locate<3>(x01, res, g.align_corners != 0, g.interp, pos, deriv, cell);
                float feat[C];
#pragma unroll
                for (int c = 0; c < C; ++c) feat[c] = 0.0f;
#pragma unroll
                for (uint32_t idx = 0; idx < 8u; ++idx) {                  // gridencoder.cu:171-192
                    float cw = 1.0f;
                    uint32_t p[3];
#pragma unroll
                    for (uint32_t d = 0; d < 3u; ++d) {
                        if ((idx & (1u << d)) == 0u) { cw *= 1.0f - pos[d]; p[d] = cell[d]; }
                        else { cw *= pos[d]; p[d] = umin(cell[d] + 1u, res - 1u); }
                    }
                    float v[C];
                    load_row<T, C>(tab + (size_t)grid_row<3>(p, res, size, mode) * C, v);
#pragma unroll
                    for (int c = 0; c < C; ++c) feat[c] = __builtin_fmaf(cw, v[c], feat[c]);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) acc[i][c] = __builtin_fmaf(wz, feat[c], acc[i][c]);
            }
        }
    };

    if constexpr (VEC) {
        for (uint32_t j = 0; j < Tn; j += 4u) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wrow + j);
            const float4 a = *reinterpret_cast<const float4 *>(xrow + j * 3u);
            const float4 b = *reinterpret_cast<const float4 *>(xrow + j * 3u + 4u);
            const float4 c = *reinterpret_cast<const float4 *>(xrow + j * 3u + 8u);
            // one sample at a time (a rolled loop with wave-uniform selects): unrolled, the compiler overlaps the
            // four samples' gathers and runs out of registers
#pragma unroll 1
            for (int jj = 0; jj < 4; ++jj) {
                const float w = jj == 0 ? w4.x : jj == 1 ? w4.y : jj == 2 ? w4.z : w4.w;
                const float px = jj == 0 ? a.x : jj == 1 ? a.w : jj == 2 ? b.z : c.y;
                const float py = jj == 0 ? a.y : jj == 1 ? b.x : jj == 2 ? b.w : c.z;
                const float pz = jj == 0 ? a.z : jj == 1 ? b.y : jj == 2 ? c.x : c.w;
                sample(w, px, py, pz);
            }
        }
    } else {
        for (uint32_t j = 0; j < Tn; ++j) sample(wrow[j], xrow[j * 3u], xrow[j * 3u + 1u], xrow[j * 3u + 2u]);
    }
    if (!ok) return;
    float *o = out + (size_t)n * g.L * C + (size_t)l0 * C;
#pragma unroll
    for (int i = 0; i < LG; ++i) {
        if (l0 + i >= g.L) break;
        if constexpr (C % 4 == 0) {
#pragma unroll
            for (int q = 0; q < C / 4; ++q)
                reinterpret_cast<float4 *>(o + i * C)[q] = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) o[i * C + c] = acc[i][c];
        }
    }
}

}  // namespace sn

using namespace sn;

extern "C" int sn_rm_grid_composite(const float *xyzs, const float *weights, uint32_t N, uint32_t T, float bound,
                                    const sn_grid_desc *grid, uint32_t tile_w, float *out, sn_stream_t stream) {
    if (N == 0) return SN_OK;
    SN_REQUIRE(xyzs && weights && grid && out, "grid_composite: NULL pointer");
    SN_REQUIRE(grid->embeddings, "grid_composite: grid has no table");
    SN_REQUIRE(grid->D == 3, "grid_composite: grids must be 3-D (got D=%u)", grid->D);
    SN_REQUIRE(grid->table_dtype == SN_F32 || grid->table_dtype == SN_F16, "grid_composite: table must be float32 or float16");
    SN_REQUIRE(table_aligned(grid->embeddings) && table_aligned(out), "grid_composite: table and out must be 16-byte aligned");
    SN_REQUIRE(bound > 0.0f, "grid_composite: bound must be positive");
    GridLevels g;
    int rc = build_grid_levels(&g, grid->offsets, grid->D, grid->C, grid->L, grid->S, grid->H, grid->gridtype,
                               (int)grid->align_corners, grid->interp);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (T == 0) { SN_HIP_OK(hipMemsetAsync(out, 0, (size_t)N * g.L * g.C * sizeof(float), st)); return SN_OK; }
    int e; const float m = frexpf(2.0f * bound, &e);
    const float inv_den = m == 0.5f ? 1.0f / (2.0f * bound) : 0.0f;     // exact reciprocal only for powers of two
    uint32_t rows = 0, nblk;
    if (tile_w) {
        SN_REQUIRE(N % tile_w == 0, "grid_composite: N=%u is not a whole number of rows of width %u", N, tile_w);
        rows = N / tile_w;
        nblk = ((tile_w + 15u) >> 4) * ((rows + 15u) >> 4);
    } else {
        nblk = div_up(N, 256);
    }
    const bool h16 = grid->table_dtype == SN_F16;
    const bool vec = (T % 4u) == 0u && table_aligned(xyzs) && table_aligned(weights);
    // levels per workgroup pass: more passes = more workgroups (the launch is a few waves per CU at image sizes of
    // interest) and fewer registers, at the price of re-reading weights/positions once per pass
    const int lg = 2;
    const bool fast = levels_fast(g) && (g.L % (uint32_t)lg) == 0u;
    const dim3 blk(256);
#define SN_GC4(TT, CC, LGG, FF, VV)                                                                                    \
    hipLaunchKernelGGL((k_grid_composite<TT, CC, LGG, FF, VV>), dim3(nblk, div_up(g.L, LGG)), blk, 0, st, xyzs, weights, \
                       (const TT *)grid->embeddings, out, N, T, g, bound, inv_den, tile_w, rows)
#define SN_GC3(TT, CC, LGG)                                                                                            \
    do {                                                                                                               \
        if (fast && vec) SN_GC4(TT, CC, LGG, true, true);                                                              \
        else if (fast) SN_GC4(TT, CC, LGG, true, false);                                                               \
        else SN_GC4(TT, CC, LGG, false, false);                                                                        \
    } while (0)
#define SN_GC(CC)                                                                                                      \
    do {                                                                                                               \
        if (h16) { if (lg == 4) SN_GC3(__half, CC, 4); else if (lg == 2) SN_GC3(__half, CC, 2); else SN_GC3(__half, CC, 1); } \
        else { if (lg == 4) SN_GC3(float, CC, 4); else if (lg == 2) SN_GC3(float, CC, 2); else SN_GC3(float, CC, 1); }  \
    } while (0)
    switch (g.C) {
        case 1: SN_GC(1); break;
        case 2: SN_GC(2); break;
        case 4: SN_GC(4); break;
        case 8: SN_GC(8); break;
        default:
            set_error("grid_composite: level_dim %u is not instantiated (1, 2, 4, 8)", g.C);
            return SN_ERR_UNSUPPORTED;
    }
#undef SN_GC
#undef SN_GC3
#undef SN_GC4
    SN_LAUNCH_CHECK("k_grid_composite");
    return SN_OK;
}
