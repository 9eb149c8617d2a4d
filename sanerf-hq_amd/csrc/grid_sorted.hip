// grid_sorted.hip — atomics-free gradient scatter of the grid encoder (sn_grid_encode_backward_sorted).
//
// Why: device-scope fp32 atomics on MI355X retire at ~2.1e10 lane-ops/s whatever the address pattern
// (tools/ubench/atomics.hip: random rows, streaming rows and a 4096-row table all measure 20-21 G/s; lanes
// sharing a row 13.7 G/s), so the reference's scatter (gridencoder.cu:252-349: 2^D * C atomics per
// sample-level) costs >= 6.7 ms for the mask-field step of BASELINE configs[4] (134 M atomics) and 12.6 ms
// as a straight port.  A 23-bit radix sort of the 16.8 M (row, contribution) pairs takes 0.44 ms on the
// same chip, after which each table row is owned by one thread and is written with plain stores.
//
//   1. k_bwd_keys    one lane per (sample, level): key = global table row of each of the 2^D corners,
//                    value = (sample << 8) | (level << 3) | corner        (out-of-range samples: sentinel key)
//   2. hipcub::DeviceRadixSort::SortPairs on the low bits that can be set
//   3. k_bwd_reduce  one lane per CHUNK consecutive sorted pairs: recomputes the blend weight of each
//                    contribution from the sample position, accumulates runs of equal rows in registers,
//                    stores runs that lie inside the chunk, and uses an atomic add only for the (at most
//                    two) runs that continue into a neighbouring chunk.
// grad_embeddings must be zero-initialised by the caller (as for the atomic path, grid.py:83).
#include "sn_common.h"

#include <hipcub/hipcub.hpp>

namespace sn {

constexpr uint32_t CHUNK = 8;

template <uint32_t D>
__global__ __launch_bounds__(256) void k_bwd_keys(const float *__restrict__ inputs, uint32_t B, GridLevels g,
                                                  uint32_t sentinel, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    constexpr uint32_t NC = 1u << D;
    const size_t base = ((size_t)level * B + b) * NC;
    float x01[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        x01[d] = inputs[(size_t)b * D + d];
        oob |= (x01[d] < 0 || x01[d] > 1);
    }
    const uint32_t res = g.res[level], size = g.size[level], mode = g.mode[level];
    float pos[D], deriv[D];
    uint32_t cell[D];
    grid_locate<D>(x01, res, g.align_corners != 0, g.interp, pos, deriv, cell);
#pragma unroll
    for (uint32_t idx = 0; idx < NC; ++idx) {
        uint32_t p[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) p[d] = (idx & (1u << d)) ? umin(cell[d] + 1, res - 1) : cell[d];
        keys[base + idx] = oob ? sentinel : g.off[level] + grid_row<D>(p, res, size, mode);
        vals[base + idx] = (b << 8) | (level << 3) | idx;
    }
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_bwd_reduce(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n,
                                                    uint32_t sentinel, const float *__restrict__ grad, const float *__restrict__ inputs,
                                                    float *__restrict__ grad_table, uint32_t B, GridLevels g, int layout) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t first = (uint64_t)t * CHUNK;
    if (first >= n) return;
    const uint32_t count = (uint32_t)((n - first) < CHUNK ? (n - first) : CHUNK);
    const uint32_t *k = keys + first, *v = vals + first;
    uint32_t cur = k[0];
    bool open_left = first > 0 && keys[first - 1] == cur;   // the run continues from the previous chunk
    float acc[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;

    auto flush = [&](uint32_t row, bool shared) {
        if (row >= sentinel) return;
        float *dst = grad_table + (size_t)row * C;
        if (shared) {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(dst + c, acc[c]);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) dst[c] = acc[c];
        }
    };

    for (uint32_t e = 0; e < count; ++e) {
        const uint32_t key = k[e];
        if (key != cur) {
            flush(cur, open_left);
            cur = key; open_left = false;
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
        }
        if (key >= sentinel) continue;
        const uint32_t val = v[e];
        const uint32_t b = val >> 8, level = (val >> 3) & 31u, idx = val & 7u;
        float x01[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) x01[d] = inputs[(size_t)b * D + d];
        float pos[D], deriv[D];
        uint32_t cell[D];
        grid_locate<D>(x01, g.res[level], g.align_corners != 0, g.interp, pos, deriv, cell);
        float w = 1.0f;   // gridencoder.cu:315-327
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) w *= (idx & (1u << d)) ? pos[d] : 1 - pos[d];
        const float *gs = layout == SN_LAYOUT_LBC ? grad + ((size_t)level * B + b) * C : grad + ((size_t)b * g.L + level) * C;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] = __builtin_fmaf(w, gs[c], acc[c]);
    }
    const bool open_right = first + count < n && keys[first + count] == cur;
    flush(cur, open_left || open_right);
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int key_bits(uint32_t sentinel) {
    int bits = 1;
    while (bits < 32 && (sentinel >> bits) != 0u) ++bits;
    return bits;
}

}  // namespace sn

using namespace sn;

extern "C" {

size_t sn_grid_backward_sorted_workspace_bytes(uint32_t B, uint32_t D, uint32_t max_level) {
    if (D < 2 || D > 3) return 0;
    const uint64_t n = (uint64_t)B * max_level * (1u << D);
    if (n == 0 || n >= (1ull << 31)) return 0;
    size_t temp = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                       (uint32_t *)nullptr, (int)n, 0, 32);
    return 4 * align256((size_t)n * sizeof(uint32_t)) + align256(temp) + 256;
}

int sn_grid_encode_backward_sorted(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                   float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    if (B == 0 || max_level == 0) return SN_OK;
    SN_REQUIRE(grad && inputs && grad_embeddings && workspace, "grid_encode_backward_sorted: NULL device pointer");
    SN_REQUIRE(layout == SN_LAYOUT_LBC || layout == SN_LAYOUT_BLC, "grid_encode_backward_sorted: bad layout %d", layout);
    if (D != 3 && D != 2) { set_error("grid_encode_backward_sorted: D=%u not instantiated (use sn_grid_encode_backward)", D); return SN_ERR_UNSUPPORTED; }
    SN_REQUIRE(B < (1u << 24), "grid_encode_backward_sorted: B=%u must be < 2^24 (value packing)", B);
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, D, C, L, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    if (max_level > L) max_level = L;
    const uint64_t n64 = (uint64_t)B * max_level * (1u << D);
    SN_REQUIRE(n64 < (1ull << 31), "grid_encode_backward_sorted: %llu contributions exceed 2^31", (unsigned long long)n64);
    const uint32_t n = (uint32_t)n64;
    const size_t need = sn_grid_backward_sorted_workspace_bytes(B, D, max_level);
    if (workspace_bytes < need) { set_error("grid_encode_backward_sorted: workspace too small (%zu bytes, need %zu)", workspace_bytes, need); return SN_ERR_WORKSPACE; }
    const size_t slab = align256((size_t)n * sizeof(uint32_t));
    char *w = reinterpret_cast<char *>(workspace);
    uint32_t *k0 = reinterpret_cast<uint32_t *>(w), *v0 = reinterpret_cast<uint32_t *>(w + slab);
    uint32_t *k1 = reinterpret_cast<uint32_t *>(w + 2 * slab), *v1 = reinterpret_cast<uint32_t *>(w + 3 * slab);
    void *temp = w + 4 * slab;
    size_t temp_bytes = workspace_bytes - 4 * slab;
    const uint32_t sentinel = (uint32_t)offsets_host[L];     // one past the last row
    hipStream_t st = (hipStream_t)stream;
    const dim3 gk(div_up(B, 256), max_level), blk(256);
    if (D == 3) hipLaunchKernelGGL((k_bwd_keys<3>), gk, blk, 0, st, inputs, B, g, sentinel, k0, v0);
    else hipLaunchKernelGGL((k_bwd_keys<2>), gk, blk, 0, st, inputs, B, g, sentinel, k0, v0);
    SN_LAUNCH_CHECK("k_bwd_keys");
    SN_HIP_OK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k0, k1, v0, v1, (int)n, 0, key_bits(sentinel), st));
    const dim3 gr(div_up(div_up(n, CHUNK), 256));
#define SN_REDUCE(DD, CC) hipLaunchKernelGGL((k_bwd_reduce<DD, CC>), gr, blk, 0, st, k1, v1, n, sentinel, grad, inputs, grad_embeddings, B, g, layout)
    bool ok = true;
    if (D == 3) {
        switch (C) { case 1: SN_REDUCE(3, 1); break; case 2: SN_REDUCE(3, 2); break; case 4: SN_REDUCE(3, 4); break;
                     case 8: SN_REDUCE(3, 8); break; case 16: SN_REDUCE(3, 16); break; case 32: SN_REDUCE(3, 32); break; default: ok = false; }
    } else {
        switch (C) { case 1: SN_REDUCE(2, 1); break; case 2: SN_REDUCE(2, 2); break; case 4: SN_REDUCE(2, 4); break;
                     case 8: SN_REDUCE(2, 8); break; default: ok = false; }
    }
#undef SN_REDUCE
    if (!ok) { set_error("grid_encode_backward_sorted: C=%u not instantiated for D=%u", C, D); return SN_ERR_UNSUPPORTED; }
    SN_LAUNCH_CHECK("k_bwd_reduce");
    return SN_OK;
}

}  // extern "C"
