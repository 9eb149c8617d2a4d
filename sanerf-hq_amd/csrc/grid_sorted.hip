// grid_sorted.hip — atomics-free gradient scatter of the grid encoder (sn_grid_encode_backward_sorted).
//
// Why: device-scope fp32 atomics on MI355X retire at ~2.1e10 lane-ops/s whatever the address pattern
// (tools/ubench/atomics.hip: random rows, streaming rows and a 4096-row table all measure 20-21 G/s; lanes
// sharing a row 13.7 G/s), so the reference's scatter (gridencoder.cu:252-349: 2^D * C atomics per
// sample-level) costs >= 6.7 ms for the mask-field step of BASELINE configs[4] (134 M atomics) and 12.6 ms
// as a straight port.  A 23-bit radix sort of the 16.8 M (row, contribution) pairs takes 0.44 ms on the
// same chip, after which each table row is owned by one thread and is written with plain stores.
//
//   1. k_bwd_keys    one lane per (sample, level): for each of the 2^D corners key = global table row, value = the
//                    pair's own index, and contrib[index][0..C) = w_corner * grad[sample, level, 0..C) (the
//                    product the reference hands to atomicAdd, gridencoder.cu:340); out-of-range samples get a
//                    sentinel key
//   2. hipcub::DeviceRadixSort::SortPairs on the low bits that can be set
//   3. k_bwd_reduce  one lane per CHUNK consecutive sorted pairs: fetches their contributions (all CHUNK row
//                    fetches in flight at once), accumulates runs of equal rows in registers, stores runs that lie
//                    inside the chunk, and uses an atomic add only for the (at most two) runs that continue into
//                    a neighbouring chunk.
// grad_embeddings must be zero-initialised by the caller (as for the atomic path, grid.py:83).
#include "sn_common.h"

#include <hipcub/hipcub.hpp>

namespace sn {

constexpr uint32_t CHUNK = 8;

// One lane per (sample, corner) pair of a level (blockIdx.y): consecutive lanes write consecutive keys, pair indices
// and contribution rows, so every store instruction of a wave covers one contiguous block (with one lane per sample
// each lane owned 8 pairs and a wave's store touched 16-64 partial lines: 0.39 ms for the mask grid's 16.8 M pairs,
// 1.7 TB/s).  The 8 lanes of a sample read the same position and gradient row (broadcast) and redo the cell
// arithmetic, which is cheap next to the 12 + 4C bytes each pair writes.
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_bwd_keys(const float *__restrict__ inputs, const float *__restrict__ grad, uint32_t B,
                                                  GridLevels g, uint32_t sentinel, int layout, uint32_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals, float *__restrict__ contrib) {
    constexpr uint32_t NC = 1u << D;
    const uint64_t pl = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // pair within the level
    if (pl >= (uint64_t)B * NC) return;
    const uint32_t b = (uint32_t)(pl / NC), idx = (uint32_t)(pl % NC);
    const uint32_t level = blockIdx.y;
    const size_t pair = (size_t)level * B * NC + pl;
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
    float gs[C];
    load_row<float, (int)C>(layout == SN_LAYOUT_LBC ? grad + ((size_t)level * B + b) * C : grad + ((size_t)b * g.L + level) * C, gs);
    uint32_t p[D];
    float w = 1.0f;   // gridencoder.cu:315-327
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const bool up = (idx & (1u << d)) != 0u;
        p[d] = up ? umin(cell[d] + 1, res - 1) : cell[d];
        w *= up ? pos[d] : 1 - pos[d];
    }
    keys[pair] = oob ? sentinel : g.off[level] + grid_row<D>(p, res, size, mode);
    vals[pair] = (uint32_t)pair;
    float *dst = contrib + pair * C;
    if constexpr (C % 4 == 0) {
#pragma unroll
        for (uint32_t q = 0; q < C / 4; ++q)
            reinterpret_cast<float4 *>(dst)[q] = make_float4(w * gs[4 * q], w * gs[4 * q + 1], w * gs[4 * q + 2], w * gs[4 * q + 3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(w * gs[0], w * gs[1]);
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) dst[c] = w * gs[c];
    }
}

// Runs of equal rows are summed hierarchically: inside a lane's chunk in registers, across the lanes of a wave by a
// segmented scan (a run of a coarse level spans tens of lanes: level 0 of the mask grid receives 256 contributions per
// row), and only the two runs that leave the wave's 512 pairs fall back to atomics.  One atomic per chunk per channel
// on the same few rows serialised in L2 and cost 1.9 of the kernel's 2.1 ms.
template <uint32_t C>
__global__ __launch_bounds__(256) void k_bwd_reduce(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n,
                                                    uint32_t sentinel, const float *__restrict__ contrib, float *__restrict__ grad_table) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t first = (uint64_t)t * CHUNK;
    const bool live = first < n;                                          // whole waves past the end still take part in the shuffles
    const uint32_t count = live ? (uint32_t)((n - first) < CHUNK ? (n - first) : CHUNK) : 0u;
    uint32_t key[CHUNK], pid[CHUNK];
    if (count == CHUNK) {     // n is a multiple of 8 for D = 3; the arrays are 256-byte aligned
        static_assert(CHUNK == 8, "two 16-byte loads per array");
        const uint4 ka = reinterpret_cast<const uint4 *>(keys + first)[0], kb = reinterpret_cast<const uint4 *>(keys + first)[1];
        const uint4 va = reinterpret_cast<const uint4 *>(vals + first)[0], vb = reinterpret_cast<const uint4 *>(vals + first)[1];
        key[0] = ka.x; key[1] = ka.y; key[2] = ka.z; key[3] = ka.w; key[4] = kb.x; key[5] = kb.y; key[6] = kb.z; key[7] = kb.w;
        pid[0] = va.x; pid[1] = va.y; pid[2] = va.z; pid[3] = va.w; pid[4] = vb.x; pid[5] = vb.y; pid[6] = vb.z; pid[7] = vb.w;
    } else {
#pragma unroll
        for (uint32_t e = 0; e < CHUNK; ++e) {
            key[e] = e < count ? keys[first + e] : sentinel;
            pid[e] = e < count ? vals[first + e] : 0u;
        }
    }
    float val[CHUNK][C];
#pragma unroll
    for (uint32_t e = 0; e < CHUNK; ++e) load_row<float, (int)C>(contrib + (size_t)pid[e] * C, val[e]);   // 8 independent fetches

    auto store_row = [&](uint32_t row, const float (&v)[C], bool shared) {
        if (row >= sentinel) return;
        float *dst = grad_table + (size_t)row * C;
        if (shared) {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(dst + c, v[c]);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) dst[c] = v[c];
        }
    };

    // ---- inside the chunk: head = the run of key[0], tail = the run of key[7]; runs strictly inside are complete ----
    const uint32_t kfirst = key[0], klast = key[CHUNK - 1];
    const bool whole = kfirst == klast;
    float head[C], acc[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) { head[c] = 0.0f; acc[c] = 0.0f; }
    uint32_t cur = kfirst;
    bool in_head = true;
#pragma unroll
    for (uint32_t e = 0; e < CHUNK; ++e) {
        if (key[e] != cur) {
            if (in_head) {
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) head[c] = acc[c];
                in_head = false;
            } else {
                store_row(cur, acc, false);                               // starts and ends inside this chunk
            }
            cur = key[e];
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
        }
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] += val[e][c];
    }
    // acc = the tail run's partial sum (== the whole chunk's sum if `whole`)

    // ---- across the wave: inclusive segmented scan of (reset, value) ----
    const uint32_t prev_last = __shfl_up(klast, 1);
    bool start;                                                            // a new run begins with this lane's first pair
    if (lane == 0u) start = !(first > 0 && live && keys[first - 1] == kfirst);
    else start = prev_last != kfirst;
    bool reset = !whole || start;                                          // the tail run does not extend to the left of this lane ...
    if (lane == 0u && whole && !start) reset = false;                      // ... unless it comes from the previous wave
    float sv[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) sv[c] = acc[c];
    bool sr = reset;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const int pr = __shfl_up((int)sr, d);
        float pv[C];
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) pv[c] = __shfl_up(sv[c], d);
        if (lane >= d) {
            if (!sr) {
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) sv[c] += pv[c];
            }
            sr = sr || (pr != 0);
        }
    }
    // exclusive values: what arrives from the left at this lane's first pair
    float cin[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) { const float v = __shfl_up(sv[c], 1); cin[c] = (lane == 0u || start) ? 0.0f : v; }
    const int cin_reset_i = __shfl_up((int)sr, 1);
    // does the run that reaches this lane from the left begin inside this wave?
    const bool left_in_wave = lane == 0u ? false : (cin_reset_i != 0);
    const uint32_t next_first = __shfl_down(kfirst, 1);
    bool cont_right;                                                       // the tail run goes on in the next lane / wave
    if (lane == 63u || !live) cont_right = live && first + count < n && keys[first + count] == klast;
    else cont_right = next_first == klast && (first + CHUNK) < n;
    if (!live) return;

    if (!whole) {
        // head run ends inside this chunk: complete it with what came from the left
        float tot[C];
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) tot[c] = head[c] + cin[c];
        store_row(kfirst, tot, !start && !left_in_wave);
        // tail run started inside this chunk
        if (!cont_right || lane == 63u) store_row(klast, acc, cont_right);
    } else {
        if (!cont_right || lane == 63u) {
            // the run ends with this chunk (or leaves the wave): sv holds its sum since its start / since lane 0
            const bool from_before_wave = !sr;                             // no reset anywhere on the way: it began before lane 0
            store_row(klast, sv, from_before_wave || cont_right);
        }
    }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int key_bits(uint32_t sentinel) {
    int bits = 1;
    while (bits < 32 && (sentinel >> bits) != 0u) ++bits;
    return bits;
}

// temporary storage of the (key, pair index) sort in its DoubleBuffer form: the four slabs of the workspace are the
// ping-pong buffers, so this is only the sort's control state (digit histograms, look-back flags, block counter)
static size_t sort_temp_bytes(uint32_t n) {
    size_t temp = 0;
    hipcub::DoubleBuffer<uint32_t> dk(nullptr, nullptr), dv(nullptr, nullptr);
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp, dk, dv, (int)n, 0, 32) != hipSuccess) return 0;
    return temp;
}

}  // namespace sn

using namespace sn;

extern "C" {

size_t sn_grid_backward_sorted_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t max_level) {
    if (D < 2 || D > 3 || C == 0) return 0;
    const uint64_t n = (uint64_t)B * max_level * (1u << D);
    if (n == 0 || n >= (1ull << 31)) return 0;
    const size_t temp = sort_temp_bytes((uint32_t)n);
    if (temp == 0) return 0;
    return 4 * align256((size_t)n * sizeof(uint32_t)) + align256((size_t)n * C * sizeof(float)) + align256(temp) + 256;
}

int sn_grid_encode_backward_sorted(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                   float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    if (B == 0 || max_level == 0) return SN_OK;
    SN_REQUIRE(grad && inputs && grad_embeddings && workspace, "grid_encode_backward_sorted: NULL device pointer");
    SN_REQUIRE(layout == SN_LAYOUT_LBC || layout == SN_LAYOUT_BLC, "grid_encode_backward_sorted: bad layout %d", layout);
    if (D != 3 && D != 2) { set_error("grid_encode_backward_sorted: D=%u not instantiated (use sn_grid_encode_backward)", D); return SN_ERR_UNSUPPORTED; }
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, D, C, L, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    if (max_level > L) max_level = L;
    const uint64_t n64 = (uint64_t)B * max_level * (1u << D);
    SN_REQUIRE(n64 < (1ull << 31), "grid_encode_backward_sorted: %llu contributions exceed 2^31", (unsigned long long)n64);
    const uint32_t n = (uint32_t)n64;
    const size_t need = sn_grid_backward_sorted_workspace_bytes(B, D, C, max_level);
    SN_REQUIRE(table_aligned(grad) && table_aligned(workspace), "grid_encode_backward_sorted: grad / workspace must be 16-byte aligned");
    if (workspace_bytes < need) { set_error("grid_encode_backward_sorted: workspace too small (%zu bytes, need %zu)", workspace_bytes, need); return SN_ERR_WORKSPACE; }
    const size_t slab = align256((size_t)n * sizeof(uint32_t));
    char *w = reinterpret_cast<char *>(workspace);
    uint32_t *k0 = reinterpret_cast<uint32_t *>(w), *v0 = reinterpret_cast<uint32_t *>(w + slab);
    uint32_t *k1 = reinterpret_cast<uint32_t *>(w + 2 * slab), *v1 = reinterpret_cast<uint32_t *>(w + 3 * slab);
    float *contrib = reinterpret_cast<float *>(w + 4 * slab);
    const size_t cslab = align256((size_t)n * C * sizeof(float));
    void *temp = w + 4 * slab + cslab;
    size_t temp_bytes = sort_temp_bytes(n);                   // exactly what the sort asks for (the cached workspace may be far larger)
    const uint32_t sentinel = (uint32_t)offsets_host[L];     // one past the last row
    hipStream_t st = (hipStream_t)stream;
    const dim3 gk(div_up((uint64_t)B << D, 256), max_level), blk(256);
    const dim3 gr(div_up(div_up(n, CHUNK), 256));
    bool ok = true;
#define SN_KEYS(DD, CC) hipLaunchKernelGGL((k_bwd_keys<DD, CC>), gk, blk, 0, st, inputs, grad, B, g, sentinel, layout, k0, v0, contrib)
    if (D == 3) {
        switch (C) { case 1: SN_KEYS(3, 1); break; case 2: SN_KEYS(3, 2); break; case 4: SN_KEYS(3, 4); break;
                     case 8: SN_KEYS(3, 8); break; case 16: SN_KEYS(3, 16); break; case 32: SN_KEYS(3, 32); break; default: ok = false; }
    } else {
        switch (C) { case 1: SN_KEYS(2, 1); break; case 2: SN_KEYS(2, 2); break; case 4: SN_KEYS(2, 4); break;
                     case 8: SN_KEYS(2, 8); break; default: ok = false; }
    }
#undef SN_KEYS
    if (!ok) { set_error("grid_encode_backward_sorted: C=%u not instantiated for D=%u", C, D); return SN_ERR_UNSUPPORTED; }
    SN_LAUNCH_CHECK("k_bwd_keys");
    // The sort's control storage is cleared by us: the same workspace serves grids with different pair counts (different
    // layouts of that storage), and a replayed HIP graph of the training step faulted inside the onesweep kernel
    // (scatter through stale state) until this memset was added.  A few hundred KiB.
    SN_HIP_OK(hipMemsetAsync(temp, 0, temp_bytes, st));
    hipcub::DoubleBuffer<uint32_t> dkeys(k0, k1), dvals(v0, v1);
    SN_HIP_OK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, dkeys, dvals, (int)n, 0, key_bits(sentinel), st));
    const uint32_t *ks = dkeys.Current(), *vs = dvals.Current();        // whichever slab the last pass wrote
#define SN_REDUCE(CC) hipLaunchKernelGGL((k_bwd_reduce<CC>), gr, blk, 0, st, ks, vs, n, sentinel, contrib, grad_embeddings)
    switch (C) { case 1: SN_REDUCE(1); break; case 2: SN_REDUCE(2); break; case 4: SN_REDUCE(4); break;
                 case 8: SN_REDUCE(8); break; case 16: SN_REDUCE(16); break; case 32: SN_REDUCE(32); break; default: ok = false; }
#undef SN_REDUCE
    if (!ok) { set_error("grid_encode_backward_sorted: C=%u not instantiated for D=%u", C, D); return SN_ERR_UNSUPPORTED; }
    SN_LAUNCH_CHECK("k_bwd_reduce");
    return SN_OK;
}

}  // extern "C"
