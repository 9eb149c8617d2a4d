// render.hip — fused whole-path renderer for gfx950 (sn_rm_render_rays).
//
// Replaces the torch op chain of NeRFRenderer.run (nerf/renderer.py:221-357) together with
// NeRFNetwork.density / forward (nerf/network.py:146-186) for perturb=False rendering.
//
// Mapping: ONE LANE = ONE RAY for the whole march.  A wave owns an 8x8 pixel
// tile, a 256-thread workgroup a 16x16 tile, and all 64 lanes step through the samples in
// lock-step, so at step j the wave's 64 gathers per corner land in one small 3-D patch:
// coarse levels collapse to a handful of cache lines and fine levels share vertices between
// neighbouring pixels.  Per-ray state (transmittance, colour accumulators, cdf merge
// cursors) lives in registers, in order, so the fp64 prefix sums that decide the sample
// indices are sequential exactly like the CPU oracle's.
//
//   k_prop_stage   stage k < last: bins -> xyz -> contract -> hash grid (L<=8,C=2) ->
//                  tiny MLP (VALU fmaf chains, weights in SGPRs) -> sigma -> weights
//                  (scratch [T][Npad]) -> in-kernel sample_pdf merge -> next bins (scratch).
//   k_final_stage  last stage: hash grid (L=16,C=2) -> 32->64->64->16 MLP on the matrix
//                  cores (v_mfma_f32_32x32x2_f32, exact fp32; operands exchanged between the
//                  two half-waves with v_permlane32_swap, weights pre-packed in LDS in
//                  A-operand order) -> sigma/geo features -> ordered compositing in
//                  registers -> per-ray view MLP -> sigmoid -> image/depth/weights_sum.
#include "sn_common.h"
#include "sh_basis.inc"

#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <cstdio>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace sn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct RayCommon {
    const float *rays_o, *rays_d, *cnf;
    uint32_t N;        // rays in this launch
    uint32_t W;        // image width (tile mode) or 0
    uint32_t rows;     // image rows in this launch (tile mode)
    uint32_t Npad;     // scratch row stride = gridDim.x * 256
    float aabb[6];
    float min_near, bound;
    int contract, last_opaque;
    float bg;
    int xcd_swizzle;
    uint32_t tlw;      // log2 of the wave tile's width in pixels (3: 8x8; sn_render_tuning.wave_tile)
    float inv_den;     // 1 / (2*bound) when that is a power of two (exact scaling), else 0
};

// Workgroup id -> tile id.  The dispatcher places workgroup b on XCD b % 8 (observed, not contractual: used
// for speed only).  Remapping so that each XCD owns a contiguous range of tile ids keeps neighbouring image
// tiles -- which share fine-level table lines -- behind the same private L2.  Bijective for any grid size.
__device__ __forceinline__ uint32_t tile_id_x(const RayCommon &rc, uint32_t b, uint32_t nwg) {
    if (!rc.xcd_swizzle) return b;
    const uint32_t xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (b >> 3);
}
__device__ __forceinline__ uint32_t tile_id(const RayCommon &rc) { return tile_id_x(rc, blockIdx.x, gridDim.x); }

// lane -> ray.  Tile mode: wave = 8x8 pixels; a workgroup takes four CONSECUTIVE wave tiles of this order: wave-tile rows go in
// pairs, column-major inside a pair ((0,0) (0,1) (1,0) (1,1) (2,0) ...), so that a workgroup is a 16x16-pixel block wherever the
// image has one, and an odd last wave-tile row is walked left to right (32x8 pixels per workgroup).  The launch has
// ceil(wave tiles / 4) workgroups whatever the shape -- a 200-row band of a 1600-wide image (BASELINE configs[3] on 8 GPUs) is 1250
// workgroups, not the 1300 of whole 16x16 tiles: tools/staircase.py shows the stages' time stepping at multiples of 256 workgroups
// (one per CU), 1300 is just past the step at 1280 (profiles/r04/staircase_1600.txt).  Host twin: blocks_for().
__device__ __forceinline__ bool ray_of_lane(const RayCommon &rc, uint32_t wg, uint32_t &n) {
    const uint32_t tid = threadIdx.x;
    if (rc.W) {
        // wave tile = 2^tlw x 2^(6 - tlw) pixels (8x8 unless sn_render_tuning.wave_tile says otherwise; tx8 / ty8 keep their round-1 names)
        const uint32_t tlw = rc.tlw, tlh = 6u - tlw, tw = 1u << tlw, th = 1u << tlh;
        const uint32_t tx8 = (rc.W + tw - 1u) >> tlw, ty8 = (rc.rows + th - 1u) >> tlh;
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t g = wg * 4u + wave;                       // wave tile
        const uint32_t pair = 2u * tx8, paired = (ty8 >> 1) * pair;
        uint32_t wx, wy;
        if (g < paired) { const uint32_t p = g / pair, i = g - p * pair; wx = i >> 1; wy = 2u * p + (i & 1u); }
        else { wx = g - paired; wy = ty8 & ~1u; }                // the odd last row (wx >= tx8: padding of the last workgroup)
        const uint32_t py = (wy << tlh) + (lane >> tlw);
        const uint32_t px = (wx << tlw) + (lane & (tw - 1u));
        const bool ok = wx < tx8 && px < rc.W && py < rc.rows;
        n = ok ? py * rc.W + px : 0u;
        return ok;
    }
    n = wg * 256u + tid;
    const bool ok = n < rc.N;
    if (!ok) n = 0;
    return ok;
}

struct RaySetup {
    float o[3], d[3];
    float s_near, s_far;
};

__device__ __forceinline__ void setup_ray(const RayCommon &rc, uint32_t n, RaySetup &rs) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { rs.o[k] = rc.rays_o[(size_t)n * 3 + k]; rs.d[k] = rc.rays_d[(size_t)n * 3 + k]; }
    float near, far;
    near_far_one(rs.o, rs.d, rc.aabb, rc.min_near, near, far);
    if (rc.cnf) {  // renderer.py:233-235
        const float cn = rc.cnf[(size_t)n * 2], cf = rc.cnf[(size_t)n * 2 + 1];
        near = cn > near ? cn : near;
        far = cf < far ? cf : far;
    }
    rs.s_near = spacing_fn(near);
    rs.s_far = spacing_fn(far);
}

// renderer.py:277 — normalised bin -> distance along the ray
__device__ __forceinline__ float real_bin(const RaySetup &rs, float b) {
    const float a = rs.s_near * (1.0f - b);
    const float c = rs.s_far * b;
    return spacing_inv(a + c);
}

// renderer.py:279-285 + grid.py:156: sample position -> table coordinate in [0,1]
__device__ __forceinline__ void sample_x01(const RayCommon &rc, const RaySetup &rs, float tmid, float (&p)[3], float (&x01)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float m = rs.d[k] * tmid; p[k] = rs.o[k] + m; }
    if (rc.contract) contract3(p[0], p[1], p[2]);
    const float den = 2.0f * rc.bound;
    if (rc.inv_den != 0.0f) {   // 2*bound is a power of two (bound = 2 when contracted): x / den == x * (1/den) exactly
#pragma unroll
        for (int k = 0; k < 3; ++k) x01[k] = (p[k] + rc.bound) * rc.inv_den;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) x01[k] = (p[k] + rc.bound) / den;
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- two-phase level groups -----------------------------------------------------------------
// hipcc schedules "8 loads, wait, 16 fmas" level by level if the source is written level by level,
// which leaves 8 gathers in flight per lane and exposes one full memory latency per level.  The
// march therefore handles GROUP levels at a time in two explicit phases separated by scheduling
// barriers: (1) locate + row indices + ISSUE all GROUP*8 gathers (32-bit byte offsets from a
// wave-uniform base -> saddr-form global loads, one address VGPR each), (2) blend.
template <typename T, int C>
struct Corner { float v[C]; };

// fp16 tables, C = 2: a fetched row stays PACKED (its half2 bit pattern in v[0]) and the blend multiplies it with
// v_fma_mix_f32, which widens an f16 operand on the fly -- the same fp32 fma on the same exactly-converted value as
// v_cvt_f32_f16 + v_fma_f32, so results are bit-identical, but the 16 conversions per level (256 of ~1500 vector
// instructions per sample of the final stage, 80 of ~700 in a proposal stage) disappear.
template <typename T, int C>
__device__ __forceinline__ void corner_set_half2(Corner<T, C> &c, uint32_t bits) {
    static_assert(C == 2 && sizeof(T) == 2, "packed rows: fp16 tables with two features per level");
    c.v[0] = __uint_as_float(bits); c.v[1] = 0.0f;
}
__device__ __forceinline__ float fma_mix_lo(float w, uint32_t packed, float acc) {      // fmaf(w, (float)packed.lo, acc)
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(packed), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float w, uint32_t packed, float acc) {      // fmaf(w, (float)packed.hi, acc)
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(packed), "v"(acc));
    return d;
}

// XSWAP (hashed fine levels of the final stage): the two x-corners of a cell sit in the same 128-byte line 15 times
// out of 16 (index = x ^ y*P1 ^ z*P2: x and x+1 differ in the low bits only), but as two instructions each line is
// looked up twice and a fine-level instruction already touches ~40 distinct lines (its cost).
// The half-waves therefore trade addresses with one v_permlane32_swap per corner pair: the first load serves BOTH
// x-corners of lanes 0-31, the second those of lanes 32-63 -- about half the distinct lines per instruction, same
// instruction count -- and blend_level_x swaps the fetched values back.
#ifndef SN_PROP_GROUP
#define SN_PROP_GROUP 2      // proposal stage: levels gathered in groups of 2 + 2 + 1.  History: all 5 at once spilled at 128 VGPRs; 3 + 2 fitted
                             // (4.70 -> 4.55 ms); 2 + 2 + 1 fits 96 VGPRs = 5 waves per SIMD instead of 4 (SN_PROP_WAVES): [128,64,32] 4.53 -> 4.44 ms fp32
#endif
#ifndef SN_PROP_GROUP_H
#define SN_PROP_GROUP_H 2    // the same for fp16 tables (78 VGPRs): 4.10 -> 3.98 ms
#endif
template <typename T, int KIND, int l>
constexpr bool xswap_level() {
    return KIND == 1 && l >= 0;
}
#ifndef SN_XSWAP_MODE
#define SN_XSWAP_MODE 0      // which lanes trade the two x-corners of a cell (A/B, tools/xswap_ab.sh, profiles/r06/xswap_ab.txt; all bit-identical): 0 = the wave's
                             // halves (v_permlane32_swap; 6.21-6.27 ms); 1 = neighbours (lanes 2i, 2i+1: both corners of a ray in ONE quad of the address unit, 13 %
                             // fewer quad-line pairs but 40 % more distinct lines per instruction and a select per swap: 6.43-6.45 ms); 2 = 16-lane rows
                             // (v_permlane16_swap: 6.19-6.20 ms, inside the noise of the default)
#endif
__device__ __forceinline__ void half_wave_swap(uint32_t &a, uint32_t &b) {
#if SN_XSWAP_MODE == 0
    // a' = [a.lanes0-31 | b.lanes0-31], b' = [a.lanes32-63 | b.lanes32-63]
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
#elif SN_XSWAP_MODE == 1
    // lane 2i: a' = a, b' = a of lane 2i+1;  lane 2i+1: a' = b of lane 2i, b' = b  (its own inverse, like the half-wave swap)
    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    const uint32_t na = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true);
    const bool odd = (threadIdx.x & 1u) != 0u;
    a = odd ? nb : a; b = odd ? b : na;
#else
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
#endif
}

// Dense levels re-laid out for the final stage (k_pack_pairs, once per render call, a few MB): pair row i of a level
// = (row i, row of the +1 neighbour in x, clamped at the border) -> the two x-corners of a cell come from ONE
// naturally aligned 16-byte load and the border case needs no select (fp16: quad rows, the y-neighbours too).  The gather address
// rate (one wave instruction per 17.5 cycles per CU) is what bounds the final stage: 4 loads
// instead of 8 on the 5 dense levels = 108 instead of 128 gather instructions per sample.
struct PairTab {
    const void *base;            // device; NULL = not available
    uint32_t off[8];             // first pair row of each dense level
};

template <typename T, int C, int KIND, bool PAIRA, bool XSWAP = false>
__device__ __forceinline__ void issue_level(const T *__restrict__ table, const GridLevels &g, int l, const float (&x01)[3],
                                            float (&pos)[3], Corner<T, C> (&cv)[8], const PairTab *pt = nullptr) {
    const uint32_t res = g.res[l], size = g.size[l], mode = g.mode[l];
    const T *tab = table + (size_t)g.off[l] * C;
    uint32_t cell[3], offs[8];
    locate_linear(x01, res, pos, cell);
    if constexpr (PAIRA && KIND == 0 && C == 2 && sizeof(T) == 2) {
        // fp16 rows are 4 bytes: a 16-byte load (which costs what an 8-byte one costs) carries the x- AND the
        // y-neighbour: quad row i = rows (x,y), (x+1,y), (x,y+1), (x+1,y+1), clamped -> 2 loads per level
        constexpr uint32_t QB = 16u;
        const uint32_t sy = res * QB, sz = res * res * QB, top = res - 1u;
        const uint32_t base_off = cell[0] * QB + __umul24(cell[1], sy);
        const uint32_t Z0 = __umul24(cell[2], sz), Z1 = umin(Z0 + sz, top * sz);
        const char *pbase = reinterpret_cast<const char *>(pt->base) + (size_t)pt->off[l] * QB;
#pragma unroll
        for (int zi = 0; zi < 2; ++zi) {
            const uint4 t = *reinterpret_cast<const uint4 *>(pbase + base_off + (zi ? Z1 : Z0));
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) corner_set_half2(cv[4 * zi + q], w[q]);   // q = x + 2 y -> corner index x + 2 y + 4 z
        }
        return;
    }
    if constexpr (PAIRA && KIND == 0 && C == 2) {
        constexpr uint32_t PB = (uint32_t)(2 * C * sizeof(T));
        const uint32_t sy = res * PB, sz = res * res * PB, top = res - 1u;
        const uint32_t X0 = cell[0] * PB, Y0 = __umul24(cell[1], sy), Z0 = __umul24(cell[2], sz);
        const uint32_t Y1 = umin(Y0 + sy, top * sy), Z1 = umin(Z0 + sz, top * sz);
        const char *pbase = reinterpret_cast<const char *>(pt->base) + (size_t)pt->off[l] * PB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t off = X0 + ((i & 1) ? Y1 : Y0) + ((i & 2) ? Z1 : Z0);
            if constexpr (sizeof(T) == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(pbase + off);
                cv[2 * i].v[0] = t.x; cv[2 * i].v[1] = t.y; cv[2 * i + 1].v[0] = t.z; cv[2 * i + 1].v[1] = t.w;
            } else {
                const uint2 t = *reinterpret_cast<const uint2 *>(pbase + off);
                corner_set_half2(cv[2 * i], t.x); corner_set_half2(cv[2 * i + 1], t.y);
            }
        }
        return;
    }
    corner_offsets<KIND, (uint32_t)(C * sizeof(T))>(cell, res, size, mode, offs);
    if constexpr (XSWAP) {
#pragma unroll
        for (int p = 0; p < 4; ++p) half_wave_swap(offs[2 * p], offs[2 * p + 1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const T *row = reinterpret_cast<const T *>(reinterpret_cast<const char *>(tab) + offs[i]);
        if constexpr (C == 2 && sizeof(T) == 4) {
            const float2 t = *reinterpret_cast<const float2 *>(row);
            cv[i].v[0] = t.x; cv[i].v[1] = t.y;
        } else if constexpr (C == 2 && sizeof(T) == 2) {
            if constexpr (XSWAP) {   // keep the packed row: blend_level_x swaps one register per corner, then unpacks
                cv[i].v[0] = __uint_as_float(*reinterpret_cast<const uint32_t *>(row));
                cv[i].v[1] = 0.0f;
            } else {
                corner_set_half2(cv[i], *reinterpret_cast<const uint32_t *>(row));
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) cv[i].v[c] = table_ld<T>(row + c);
        }
    }
}

// PKW: the 8 corner weights as 6 packed multiplies (x-pairs: (wx0, wx1) * wy, then * wz) instead of 12 scalar ones;
// every weight is still (wx * wy) * wz, every channel still one corner-ascending fma chain.  Final stage only (the
// proposal kernels have no registers to spare for the even-aligned pairs).
template <typename T, int C, bool PKW = false>
__device__ __forceinline__ void blend_level(const float (&pos)[3], const Corner<T, C> (&cv)[8], float (&acc)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    if constexpr (C == 2 && sizeof(T) == 2) {   // packed fp16 rows (corner_set_half2): widening fma, same order
        float w[8];
        if constexpr (PKW) {                // the 8 corner weights as 6 packed multiplies: every weight still (wx * wy) * wz
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 wx = {1.0f - pos[0], pos[0]};
            const float wy0 = 1.0f - pos[1], wz0 = 1.0f - pos[2];
            const f2 xy0 = wx * f2{wy0, wy0}, xy1 = wx * f2{pos[1], pos[1]};
            const f2 w01 = xy0 * f2{wz0, wz0}, w23 = xy1 * f2{wz0, wz0}, w45 = xy0 * f2{pos[2], pos[2]}, w67 = xy1 * f2{pos[2], pos[2]};
            w[0] = w01.x; w[1] = w01.y; w[2] = w23.x; w[3] = w23.y; w[4] = w45.x; w[5] = w45.y; w[6] = w67.x; w[7] = w67.y;
        } else {
#pragma unroll
            for (uint32_t idx = 0; idx < 8; ++idx) {
                float t = 1.0f;
#pragma unroll
                for (uint32_t d = 0; d < 3; ++d) t *= (idx & (1u << d)) ? pos[d] : 1.0f - pos[d];
                w[idx] = t;
            }
        }
#pragma unroll
        for (uint32_t idx = 0; idx < 8; ++idx) {
            const uint32_t bits = __float_as_uint(cv[idx].v[0]);
            acc[0] = fma_mix_lo(w[idx], bits, acc[0]);
            acc[1] = fma_mix_hi(w[idx], bits, acc[1]);
        }
        return;
    }
    if constexpr (PKW && C == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 wx = {1.0f - pos[0], pos[0]};
        const float wy0 = 1.0f - pos[1], wz0 = 1.0f - pos[2];
        const f2 xy0 = wx * f2{wy0, wy0}, xy1 = wx * f2{pos[1], pos[1]};
        const f2 w01 = xy0 * f2{wz0, wz0}, w23 = xy1 * f2{wz0, wz0}, w45 = xy0 * f2{pos[2], pos[2]}, w67 = xy1 * f2{pos[2], pos[2]};
        const float w[8] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y, w67.x, w67.y};
        f2 a = {0.0f, 0.0f};
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) a = __builtin_elementwise_fma(f2{w[idx], w[idx]}, f2{cv[idx].v[0], cv[idx].v[1]}, a);
        acc[0] = a.x; acc[1] = a.y;
        return;
    }
#pragma unroll
    for (uint32_t idx = 0; idx < 8; ++idx) {   // corner order and weight products exactly as gridencoder.cu:171-192
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) w *= (idx & (1u << d)) ? pos[d] : 1.0f - pos[d];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = __builtin_fmaf(w, cv[idx].v[c], acc[c]);
    }
}

// blend of a level whose gathers were issued with XSWAP: first give every lane its own corner values back
template <typename T, int C, bool PKW = false>
__device__ __forceinline__ void blend_level_x(const float (&pos)[3], const Corner<T, C> (&cv)[8], float (&acc)[C]) {
    Corner<T, C> own[8];
    if constexpr (C == 2 && sizeof(T) == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t a = __float_as_uint(cv[2 * p].v[0]), b = __float_as_uint(cv[2 * p + 1].v[0]);
            half_wave_swap(a, b);
            corner_set_half2(own[2 * p], a); corner_set_half2(own[2 * p + 1], b);
        }
        blend_level<T, C, PKW>(pos, own, acc);
        return;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t a = __float_as_uint(cv[2 * p].v[c]), b = __float_as_uint(cv[2 * p + 1].v[c]);
            half_wave_swap(a, b);
            own[2 * p].v[c] = __uint_as_float(a); own[2 * p + 1].v[c] = __uint_as_float(b);
        }
    }
    blend_level<T, C, PKW>(pos, own, acc);
}

// Encodes all L levels; `emit(l, acc)` receives each level's C features (zeros when out of range).
// K = number of leading dense levels (levels >= K are hashed); K < 0: level kind decided at run time.
// Returns true for lanes whose coordinate is outside [0,1] (gridencoder.cu:105-130 writes zeros for them): the
// caller zeroes those lanes' features on a wave-uniform, practically never taken branch instead of paying a
// select per level (positions are contracted into [0,1] on this path).
template <typename T, int L, int C, int GROUP, int K, bool PAIR, typename Emit>
__device__ __forceinline__ bool encode_grouped(const T *__restrict__ table, const GridLevels &g, const float (&x01)[3], Emit emit,
                                               const PairTab *pt = nullptr) {
    // (a last, shorter group is allowed: L = 5 levels in groups of 3 + 2)
    bool oob = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) oob |= (x01[d] < 0.0f || x01[d] > 1.0f);
    constexpr int G = GROUP >= L ? L : GROUP;
    static_for<0, (L + G - 1) / G>([&](auto grp) {
        constexpr int l0 = decltype(grp)::value * G;
        constexpr int GN = (L - l0) < G ? (L - l0) : G;        // levels in this group
        float pos[GN][3];
        Corner<T, C> cv[GN][8];
        static_for<0, GN>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            constexpr int KIND = K < 0 ? -1 : ((l0 + k) < K ? 0 : 1);
            issue_level<T, C, KIND, (PAIR && K > 0 && K <= 8)>(table, g, l0 + k, x01, pos[k], cv[k], pt);
        });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < GN; ++k) {
            float acc[C];
            blend_level<T, C>(pos[k], cv[k], acc);
            emit(l0 + k, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    return oob;
}

// ---- software-pipelined form of encode_grouped ------------------------------------------------
// issue_group<GRP> starts the GROUP*8 gathers of level group GRP; blend_group<GRP> consumes them.
// The final stage uses the pair to keep the FIRST group of sample j+1 in flight across the
// matrix-core phase of sample j (the compiler never hoists loads over the loop back-edge).
template <typename T, int C, int G>
struct GroupRegs {
    float pos[G][3];
    Corner<T, C> cv[G][8];
    bool oob;
};

template <typename T, int C, int G, int K, int GRP>
__device__ __forceinline__ void issue_group(const T *__restrict__ table, const GridLevels &g, const float (&x01)[3],
                                            GroupRegs<T, C, G> &r, const PairTab &pt) {
    r.oob = false;   // out-of-range lanes are fixed up by the caller on a wave-uniform rare path
    static_for<0, G>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int l = GRP * G + k;
        constexpr int KIND = K < 0 ? -1 : (l < K ? 0 : 1);
        issue_level<T, C, KIND, (K > 0 && K <= 8), xswap_level<T, KIND, l>()>(table, g, l, x01, r.pos[k], r.cv[k], &pt);
    });
}

template <typename T, int C, int G, int K, int GRP, typename Emit>
__device__ __forceinline__ void blend_group(const GroupRegs<T, C, G> &r, Emit emit) {
    static_for<0, G>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int l = GRP * G + k;
        constexpr int KIND = K < 0 ? -1 : (l < K ? 0 : 1);
        float acc[C];
        if constexpr (xswap_level<T, KIND, l>()) blend_level_x<T, C, true>(r.pos[k], r.cv[k], acc);
        else blend_level<T, C, true>(r.pos[k], r.cv[k], acc);
        emit(l, acc);
    });
}

// ---- final stage, specialised instantiations (K > 0: dense prefix + hashed tail): per-level constants from the host ----
// The generic code above derives every per-level quantity from GridLevels inside the march: u32 -> f32 conversions of the
// resolutions, 64-bit base pointers per level (more than the scalar file holds: ~150 v_readlane / v_writelane spill
// instructions per sample), border selects and clamps on every level.  The final stage issues one instruction per
// 4 cycles per SIMD whatever its type and is bound by exactly that, so this variant removes
// instructions, bit-identically:
//   * resolutions / limits arrive as floats and byte quantities in the kernel argument (FinalLv);
//   * hashed levels of a GridEncoder table are equal-sized and contiguous (grid.py:121-136: a level is hashed iff it hit
//     the 2^log2_hashmap_size cap), so ONE base pointer + a per-level byte offset OR-ed into the masked x term serves all
//     of them; the two prime multiplies become full-rate 24-bit multiplies (only the bits under the mask matter);
//   * FAST (wave-uniform, decided per sample): every lane's coordinate is at least one coarsest-hashed-level cell away
//     from the border of [0,1]^3 -> on hashed levels neither the position clamp nor the clamp of the +1 neighbour
//     (gridencoder.cu:148,182) can trigger, and out-of-range zeroing cannot apply.  Otherwise the general form runs.
struct FinalLv {
    float res_f[16];                 // level resolution (gridencoder.cu:133) as float
    float top_f[16];                 // res - 1
    uint32_t d_sy[8], d_sz[8];       // dense levels inside the pair / quad-row table (PairTab): y and z strides in bytes,
    uint32_t d_ylim[8], d_zlim[8];   //   (res - 1) * stride, the clamp of the +1 neighbour,
    uint32_t d_off[8];               //   byte offset of the level
    uint32_t h_off[16];              // hashed levels: byte offset from the first hashed level (a multiple of size * row bytes)
    uint32_t h_mask;                 // (size - 1) * row bytes, the same for every hashed level
    float in_lo, in_hi;              // FAST <=> in_lo <= x01[d] <= in_hi for all lanes and dimensions
    const char *pair_base, *hash_base;
};

template <typename T, bool DENSE, bool FAST, bool XSWAP, int l>
__device__ __forceinline__ void issue_level_lv(const FinalLv &lv, const float (&x01)[3], float (&pos)[3], Corner<T, 2> (&cv)[8]) {
    constexpr uint32_t RB = 2u * (uint32_t)sizeof(T);              // bytes per table row (C = 2)
    const float rf = lv.res_f[l];
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = __builtin_fmaf(x01[d], rf, -0.5f);
        if constexpr (DENSE || !FAST) p = __builtin_amdgcn_fmed3f(p, 0.0f, lv.top_f[l]);   // = min(max(p, 0), res-1), gridencoder.cu:148
        cell[d] = (uint32_t)p;
        pos[d] = __builtin_amdgcn_fractf(p);
    }
    if constexpr (DENSE) {
        const uint32_t sy = lv.d_sy[l], sz = lv.d_sz[l];
        const uint32_t X0 = __umul24(cell[0], 16u) + lv.d_off[l];
        const uint32_t Y0 = __umul24(cell[1], sy), Z0 = __umul24(cell[2], sz);
        const uint32_t Y1 = umin(Y0 + sy, lv.d_ylim[l]), Z1 = umin(Z0 + sz, lv.d_zlim[l]);
        if constexpr (sizeof(T) == 2) {          // quad rows: (x,y), (x+1,y), (x,y+1), (x+1,y+1)
#pragma unroll
            for (int zi = 0; zi < 2; ++zi) {
                const uint4 t = *reinterpret_cast<const uint4 *>(lv.pair_base + (X0 + Y0 + (zi ? Z1 : Z0)));
                const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) corner_set_half2(cv[4 * zi + q], w[q]);
            }
        } else {                                 // pair rows: (x, x+1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t off = X0 + ((i & 1) ? Y1 : Y0) + ((i & 2) ? Z1 : Z0);
                const float4 t = *reinterpret_cast<const float4 *>(lv.pair_base + off);
                cv[2 * i].v[0] = t.x; cv[2 * i].v[1] = t.y; cv[2 * i + 1].v[0] = t.z; cv[2 * i + 1].v[1] = t.w;
            }
        }
    } else {
        constexpr uint32_t MY = (2654435761u * RB) & 0xFFFFFFu, MZ = (805459861u * RB) & 0xFFFFFFu;   // gridencoder.cu:49, low 24 bits
        const uint32_t mask = lv.h_mask, ho = lv.h_off[l];
        const uint32_t X0 = cell[0] * RB, Y0 = __umul24(cell[1], MY), Z0 = __umul24(cell[2], MZ);
        uint32_t X1 = X0 + RB, Y1 = Y0 + MY, Z1 = Z0 + MZ;
        if constexpr (!FAST) {                   // the +1 neighbour is clamped to res-1 (gridencoder.cu:182)
            const uint32_t top = (uint32_t)lv.top_f[l];
            X1 = cell[0] < top ? X1 : X0; Y1 = cell[1] < top ? Y1 : Y0; Z1 = cell[2] < top ? Z1 : Z0;
        }
        const uint32_t X0m = (X0 & mask) | ho, X1m = (X1 & mask) | ho, Y0m = Y0 & mask, Y1m = Y1 & mask, Z0m = Z0 & mask, Z1m = Z1 & mask;
        uint32_t offs[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) offs[i] = ((i & 1u) ? X1m : X0m) ^ ((i & 2u) ? Y1m : Y0m) ^ ((i & 4u) ? Z1m : Z0m);
        if constexpr (XSWAP) {
#pragma unroll
            for (int q = 0; q < 4; ++q) half_wave_swap(offs[2 * q], offs[2 * q + 1]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char *row = lv.hash_base + offs[i];
            if constexpr (sizeof(T) == 4) {
                const float2 t = *reinterpret_cast<const float2 *>(row);
                cv[i].v[0] = t.x; cv[i].v[1] = t.y;
            } else if constexpr (XSWAP) {
                cv[i].v[0] = __uint_as_float(*reinterpret_cast<const uint32_t *>(row)); cv[i].v[1] = 0.0f;
            } else {
                corner_set_half2(cv[i], *reinterpret_cast<const uint32_t *>(row));
            }
        }
    }
}

// levels L0 .. L0+G-1 (a "span": the gather groups of the final stage need not be equally long)
template <typename T, int L0, int G, int K, bool FAST>
__device__ __forceinline__ void issue_span_lv(const FinalLv &lv, const float (&x01)[3], GroupRegs<T, 2, G> &r) {
    r.oob = false;
    static_for<0, G>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int l = L0 + k;
        constexpr bool DENSE = l < K;
        issue_level_lv<T, DENSE, FAST, xswap_level<T, DENSE ? 0 : 1, l>(), l>(lv, x01, r.pos[k], r.cv[k]);
    });
}
template <typename T, int G, int K, int GRP, bool FAST>
__device__ __forceinline__ void issue_group_lv(const FinalLv &lv, const float (&x01)[3], GroupRegs<T, 2, G> &r) {
    issue_span_lv<T, GRP * G, G, K, FAST>(lv, x01, r);
}
template <typename T, int L0, int G, int K, typename Emit>
__device__ __forceinline__ void blend_span(const GroupRegs<T, 2, G> &r, Emit emit) {
    static_for<0, G>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        constexpr int l = L0 + k;
        constexpr int KIND = K < 0 ? -1 : (l < K ? 0 : 1);
        float acc[2];
        if constexpr (xswap_level<T, KIND, l>()) blend_level_x<T, 2, true>(r.pos[k], r.cv[k], acc);
        else blend_level<T, 2, true>(r.pos[k], r.cv[k], acc);
        emit(l, acc);
    });
}
// gather spans of the final stage (FinalLv path).  Span 0 is issued for sample j+1 before the matrix-core phase of sample j
// (its registers are live across that phase), so it has to be short when the MLP needs many registers; it must be all dense.
#ifndef SN_FINAL_SPANS_LT_H
#define SN_FINAL_SPANS_LT_H 4  // the same for fp16 tables
#endif
#ifndef SN_FINAL_SPANS_LT
#define SN_FINAL_SPANS_LT 4    // linear-tail instantiation: 64 weight-accumulated hidden values per lane live through the march -> short span 0
#endif
template <int CFG> struct FinalSpans;
template <> struct FinalSpans<0> { static constexpr int N = 4; static constexpr int B[5] = {0, 4, 8, 12, 16}; };
template <> struct FinalSpans<1> { static constexpr int N = 4; static constexpr int B[5] = {0, 2, 6, 11, 16}; };
template <> struct FinalSpans<2> { static constexpr int N = 4; static constexpr int B[5] = {0, 3, 7, 12, 16}; };
template <> struct FinalSpans<3> { static constexpr int N = 4; static constexpr int B[5] = {0, 1, 6, 11, 16}; };
template <> struct FinalSpans<4> { static constexpr int N = 5; static constexpr int B[6] = {0, 2, 5, 9, 13, 16}; };
template <> struct FinalSpans<5> { static constexpr int N = 5; static constexpr int B[6] = {0, 1, 4, 8, 12, 16}; };
template <> struct FinalSpans<6> { static constexpr int N = 5; static constexpr int B[6] = {0, 2, 4, 8, 12, 16}; };
template <> struct FinalSpans<7> { static constexpr int N = 5; static constexpr int B[6] = {0, 3, 6, 9, 13, 16}; };
template <> struct FinalSpans<8> { static constexpr int N = 5; static constexpr int B[6] = {0, 4, 7, 10, 13, 16}; };
template <> struct FinalSpans<9> { static constexpr int N = 6; static constexpr int B[7] = {0, 2, 5, 8, 11, 14, 16}; };
template <> struct FinalSpans<10> { static constexpr int N = 3; static constexpr int B[4] = {0, 2, 9, 16}; };
template <> struct FinalSpans<11> { static constexpr int N = 4; static constexpr int B[5] = {0, 2, 7, 12, 16}; };
template <> struct FinalSpans<12> { static constexpr int N = 3; static constexpr int B[4] = {0, 4, 10, 16}; };
template <> struct FinalSpans<13> { static constexpr int N = 3; static constexpr int B[4] = {0, 1, 8, 16}; };
template <> struct FinalSpans<14> { static constexpr int N = 2; static constexpr int B[3] = {0, 5, 16}; };
template <> struct FinalSpans<15> { static constexpr int N = 3; static constexpr int B[4] = {0, 3, 9, 16}; };

// FAST test of one sample (see FinalLv): wave-uniform
__device__ __forceinline__ bool all_interior(const FinalLv &lv, const float (&x01)[3]) {
    const bool in = (x01[0] >= lv.in_lo && x01[0] <= lv.in_hi) && (x01[1] >= lv.in_lo && x01[1] <= lv.in_hi) &&
                    (x01[2] >= lv.in_lo && x01[2] <= lv.in_hi);
    return __all(in) != 0;
}

// all levels of one grid at one position into registers; D = 3.  gridencoder.cu:94-201 per level.
template <typename T, int L, int C, int K, bool PAIRX, int GROUP = L>
__device__ __forceinline__ void encode_levels(const T *__restrict__ table, const GridLevels &g, const float (&x01)[3],
                                              float (&feat)[L * C], const PairTab *pt = nullptr) {
    const bool oob = encode_grouped<T, L, C, GROUP, K, PAIRX>(table, g, x01, [&](int l, const float (&acc)[C]) {
#pragma unroll
        for (int c = 0; c < C; ++c) feat[l * C + c] = acc[c];
    }, pt);
    // register variant (proposal stages): L*C unconditional selects; a branch here costs more in scheduling than it saves
#pragma unroll
    for (int i = 0; i < L * C; ++i) feat[i] = oob ? 0.0f : feat[i];
}

// U positions of one lane through the same grid at once: every level group issues the gathers of all U positions before the
// first blend, so a wave carries U independent gather -> blend chains (the proposal stage is bound by the length of that chain,
// not by an execution unit).  Values identical to U calls of encode_levels.
template <typename T, int L, int C, int K, bool PAIRX, int GROUP, int U>
__device__ __forceinline__ void encode_levels_multi(const T *__restrict__ table, const GridLevels &g, const float (&x01)[U][3],
                                                    float (&feat)[U][L * C], const PairTab *pt = nullptr) {
    constexpr int G = GROUP >= L ? L : GROUP;
    static_for<0, (L + G - 1) / G>([&](auto grp) {
        constexpr int l0 = decltype(grp)::value * G;
        constexpr int GN = (L - l0) < G ? (L - l0) : G;
        float pos[U][GN][3];
        Corner<T, C> cv[U][GN][8];
        static_for<0, U>([&](auto uu) {
            constexpr int u = decltype(uu)::value;
            static_for<0, GN>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                constexpr int KIND = K < 0 ? -1 : ((l0 + k) < K ? 0 : 1);
                issue_level<T, C, KIND, (PAIRX && K > 0 && K <= 8)>(table, g, l0 + k, x01[u], pos[u][k], cv[u][k], pt);
            });
        });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < GN; ++k) {
                float acc[C];
                blend_level<T, C>(pos[u][k], cv[u][k], acc);
#pragma unroll
                for (int c = 0; c < C; ++c) feat[u][(l0 + k) * C + c] = acc[c];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int u = 0; u < U; ++u) {
        bool oob = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) oob |= (x01[u][d] < 0.0f || x01[u][d] > 1.0f);
#pragma unroll
        for (int i = 0; i < L * C; ++i) feat[u][i] = oob ? 0.0f : feat[u][i];
    }
}

// y = act(W x), W [OUT][IN] row-major at a wave-uniform address (SGPR / scalar-cache loads);
// each output is one k-ascending fmaf chain (the oracle's order).
template <int IN, int OUT, int ACT>
__device__ __forceinline__ void dense_uniform(const float *__restrict__ W, const float (&x)[IN], float (&y)[OUT]) {
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < IN; ++k) acc = __builtin_fmaf(W[o * IN + k], x[k], acc);
        if (ACT == 1) acc = __builtin_fmaxf(acc, 0.0f);       // ReLU as one v_max_f32 (= acc > 0 ? acc : 0 up to the sign of a zero; NaN -> 0 either way)
        y[o] = acc;
    }
}

// An always-zero offset the optimiser cannot see through.  Adding it to an LDS index inside the
// sample loop stops loop-invariant code motion from hoisting every (loop-invariant) weight
// read out of the loop and pinning hundreds of VGPRs for the whole march.
__device__ __forceinline__ uint32_t opaque_zero() {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

// Tiny-MLP weights are staged once per workgroup in LDS with rows padded to a multiple of 4
// floats, and read back at wave-uniform addresses (LDS broadcast, one ds_read_b128 per 4
// weights).  Reading them through global pointers makes hipcc hoist hundreds of vector loads
// (it cannot prove the buffers read-only next to the kernel's stores) and wrecks occupancy.
template <int IN> struct PadIn { static constexpr int value = (IN + 3) & ~3; };

// dst[k * OUTP + o] = W[o][k]  (k-major; rows padded to a multiple of 4 outputs)
template <int IN, int OUT>
__device__ __forceinline__ void stage_weights_t(float *__restrict__ dst, const float *__restrict__ src) {
    constexpr int OUTP = PadIn<OUT>::value;
    for (uint32_t i = threadIdx.x; i < (uint32_t)(IN * OUTP); i += blockDim.x) {
        const uint32_t k = i / OUTP, o = i - k * OUTP;
        dst[i] = o < (uint32_t)OUT ? src[o * IN + k] : 0.0f;
    }
}

// y = act(W x), W k-major in LDS at a wave-uniform address (broadcast ds_read_b128 of 4 adjacent outputs);
// every output is still ONE k-ascending fmaf chain starting from 0 (the oracle's order), the chains of
// adjacent outputs just advance together, which lets the compiler use packed fp32 FMAs without shuffles.
// PIN: the weights of input k+1 are fetched under the FMAs of input k and never earlier -- their address carries a fake dependence on the
// accumulators of step k-1 (an empty asm).  In a register-hungry caller the scheduler otherwise fetches all IN * OUT weights ahead (160 registers).
template <int IN, int OUT, int ACT, bool PIN = false>
__device__ __forceinline__ void dense_ldsw_t(const float *__restrict__ Wl, const float (&x)[IN], float (&y)[OUT]) {
    constexpr int OUTP = PadIn<OUT>::value;
    float acc[OUTP];
#pragma unroll
    for (int o = 0; o < OUTP; ++o) acc[o] = 0.0f;
    uint32_t z = 0;
    float4 w[2][OUTP / 4];
    if (PIN) {
#pragma unroll
        for (int o4 = 0; o4 < OUTP / 4; ++o4) w[0][o4] = *reinterpret_cast<const float4 *>(Wl + 4 * o4);
    }
#pragma unroll
    for (int k = 0; k < IN; ++k) {
#pragma unroll
        for (int o4 = 0; o4 < OUTP / 4; ++o4) {
            const float4 wk = PIN ? w[k & 1][o4] : *reinterpret_cast<const float4 *>(Wl + k * OUTP + 4 * o4);
            if (PIN && o4 == 0 && k + 1 < IN) {
                static_assert(!PIN || OUTP == 16, "the fake dependence lists 16 accumulators");
                // (all accumulators as they stand after step k-1: the fetch of step k+1 cannot start before step k-1 is complete)
                asm volatile("" : "+v"(z) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]),
                                            "v"(acc[8 % OUTP]), "v"(acc[9 % OUTP]), "v"(acc[10 % OUTP]), "v"(acc[11 % OUTP]), "v"(acc[12 % OUTP]), "v"(acc[13 % OUTP]),
                                            "v"(acc[14 % OUTP]), "v"(acc[15 % OUTP]));
#pragma unroll
                for (int q = 0; q < OUTP / 4; ++q) w[(k + 1) & 1][q] = *reinterpret_cast<const float4 *>(Wl + z + (k + 1) * OUTP + 4 * q);
            }
            acc[4 * o4 + 0] = __builtin_fmaf(wk.x, x[k], acc[4 * o4 + 0]);
            acc[4 * o4 + 1] = __builtin_fmaf(wk.y, x[k], acc[4 * o4 + 1]);
            acc[4 * o4 + 2] = __builtin_fmaf(wk.z, x[k], acc[4 * o4 + 2]);
            acc[4 * o4 + 3] = __builtin_fmaf(wk.w, x[k], acc[4 * o4 + 3]);
        }
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) y[o] = (ACT == 1) ? __builtin_fmaxf(acc[o], 0.0f) : acc[o];
}

// Tiny-MLP weights in the [out][in_padded] layout (view MLP, evaluated once per ray)
template <int IN, int OUT>
__device__ __forceinline__ void stage_weights(float *__restrict__ dst, const float *__restrict__ src) {
    constexpr int INP = PadIn<IN>::value;
    for (uint32_t i = threadIdx.x; i < (uint32_t)(OUT * INP); i += blockDim.x) {
        const uint32_t o = i / INP, k = i - o * INP;
        dst[i] = k < (uint32_t)IN ? src[o * IN + k] : 0.0f;
    }
}

// y = act(W x) with W in LDS (padded rows), fully unrolled; k-ascending fmaf chain per output.
template <int IN, int OUT, int ACT>
__device__ __forceinline__ void dense_ldsw(const float *__restrict__ Wl, const float (&x)[IN], float (&y)[OUT]) {
    constexpr int INP = PadIn<IN>::value;
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < INP / 4; ++k4) {
            const float4 w = *reinterpret_cast<const float4 *>(Wl + o * INP + 4 * k4);
            if (4 * k4 + 0 < IN) acc = __builtin_fmaf(w.x, x[4 * k4 + 0], acc);
            if (4 * k4 + 1 < IN) acc = __builtin_fmaf(w.y, x[4 * k4 + 1], acc);
            if (4 * k4 + 2 < IN) acc = __builtin_fmaf(w.z, x[4 * k4 + 2], acc);
            if (4 * k4 + 3 < IN) acc = __builtin_fmaf(w.w, x[4 * k4 + 3], acc);
        }
        if (ACT == 1) acc = __builtin_fmaxf(acc, 0.0f);       // ReLU as one v_max_f32 (= acc > 0 ? acc : 0 up to the sign of a zero; NaN -> 0 either way)
        y[o] = acc;
    }
}

// same, but the output loop stays rolled and activations travel through an LDS column
// (xin[k*stride] -> yout[o*stride]); x is read completely before the first write, so
// xin == yout is allowed.
template <int IN, int OUT, int ACT>
__device__ __forceinline__ void dense_ldsw_col(const float *__restrict__ Wl, const float *xin, float *yout, uint32_t stride) {
    constexpr int INP = PadIn<IN>::value;
    float x[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = xin[k * stride];
#pragma unroll 1
    for (int o = 0; o < OUT; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < INP / 4; ++k4) {
            const float4 w = *reinterpret_cast<const float4 *>(Wl + o * INP + 4 * k4);
            if (4 * k4 + 0 < IN) acc = __builtin_fmaf(w.x, x[4 * k4 + 0], acc);
            if (4 * k4 + 1 < IN) acc = __builtin_fmaf(w.y, x[4 * k4 + 1], acc);
            if (4 * k4 + 2 < IN) acc = __builtin_fmaf(w.z, x[4 * k4 + 2], acc);
            if (4 * k4 + 3 < IN) acc = __builtin_fmaf(w.w, x[4 * k4 + 3], acc);
        }
        if (ACT == 1) acc = __builtin_fmaxf(acc, 0.0f);       // ReLU as one v_max_f32 (= acc > 0 ? acc : 0 up to the sign of a zero; NaN -> 0 either way)
        yout[o * stride] = acc;
    }
}

// degree-4 real SH (16 values) of a unit vector; basis generated by tools/gen_sh.py
__device__ __forceinline__ void sh_degree4(float x, float y, float z, float (&o)[16]) {
    const uint32_t C = 4u;
    SN_SH_POWERS
    (void)x4; (void)x5; (void)x6; (void)x7; (void)y4; (void)y5; (void)y6; (void)y7; (void)z4; (void)z5; (void)z6; (void)z7;
    SN_SH_VALUES(o);
}

__device__ __forceinline__ float nan_to_num(float v) {
    if (v != v) return 0.0f;
    if (v == __builtin_inff()) return FLT_MAX;
    if (v == -__builtin_inff()) return -FLT_MAX;
    return v;
}

// ------------------------------------------------------------------------------------------
// proposal stage
// ------------------------------------------------------------------------------------------
struct PropArgs {
    RayCommon rc;
    GridLevels g;
    const void *table;
    const float *w0, *w1;        // MLP weights (device), [HID][IN], [1][HID]
    uint32_t T, Tn;              // steps of this stage; bins of the next stage = Tn + 1
    const float *bins_in;        // scratch [T+1][Npad] or NULL (stage 0)
    const float *bins0_tab;      // device [T+1] (bins0_stride 0) or per ray [N][bins0_stride], or NULL (stage 0 only)
    const float *u_tab;          // device [Tn+1] (u_stride 0) or per ray [N][u_stride], or NULL
    uint32_t bins0_stride, u_stride;
    float *dbg_bins_next;        // [N,Tn+1] or NULL: the resampled bins in the reference's layout (io->skip_final)
    float *w_scr;                // scratch [T][Npad]
    float *bins_out;             // scratch [Tn+1][Npad]
    float *dbg_bins, *dbg_w, *dbg_sigma;  // [N,T+1], [N,T], [N,T] or NULL
    int32_t *dbg_inds;                    // [N,Tn+1] or NULL
    PairTab pairs;                        // dense levels as aligned x-pairs
    int skip_miss;                        // the last stage runs k_final_stage_cmp: waves whose rays all miss the aabb leave at once
};

// renderer.py:133-135: near = far = 1e9 marks a ray that misses the aabb
__device__ __forceinline__ bool ray_misses(const RayCommon &rc, const RaySetup &rs) {
    float nr, fr;
    near_far_one(rs.o, rs.d, rc.aabb, rc.min_near, nr, fr);
    return nr == 1e9f && fr == 1e9f;
}

#ifndef SN_PROP_PB
#define SN_PROP_PB 4         // cdf entries per block of the sample_pdf merge (pass 2 of the proposal stage); 8: same, 16: slower (select chains)
#endif
#ifndef SN_PROP_WAVES
#define SN_PROP_WAVES 5      // waves per SIMD the proposal stage is compiled for (register budget 512 / N in steps of 8: 96 VGPRs); the
                             // stage is bound by each wave's dependent chain, so a fifth wave buys 3-4 % (profiles/r02/ab_round2_experiments.txt)
#endif
#ifndef SN_PROP_GROUP_U2
#define SN_PROP_GROUP_U2 1   // levels per gather group of the two-samples-at-once instantiation (x 2 positions)
#endif
#ifndef SN_PROP_WAVES_U2
#define SN_PROP_WAVES_U2 3   // the two-samples-at-once instantiation (UN = 2): 168 VGPRs
#endif
// UN: samples of a ray evaluated together in pass 1 (their gathers and their MLP chains interleave: two dependent chains per wave instead of
// one).  Every sample goes through the same arithmetic and the weights / the normaliser are formed in the same order: bit-identical to UN = 1.
template <typename TT, int L, int C, int HID, int K, int UN = 1>
__global__ __launch_bounds__(256, (UN == 1 ? SN_PROP_WAVES : SN_PROP_WAVES_U2)) void k_prop_stage(PropArgs a) {
    SN_POISON_ALL();
    constexpr int IN = L * C;
    __shared__ __attribute__((aligned(16))) float lds_w0[IN * PadIn<HID>::value];   // k-major [IN][HID]
    __shared__ __attribute__((aligned(16))) float lds_w1[PadIn<HID>::value];        // [1][HID]
    stage_weights_t<IN, HID>(lds_w0, a.w0);
    stage_weights<HID, 1>(lds_w1, a.w1);
    __syncthreads();
    uint32_t n;
    const uint32_t wg = tile_id(a.rc);
    const bool ok = ray_of_lane(a.rc, wg, n);
    const uint32_t r = wg * 256u + threadIdx.x;           // scratch column
    const uint32_t Npad = a.rc.Npad;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    // opt-in compaction (cfg->compact_live): k_final_stage_cmp gives rays that miss the aabb (and lanes beyond the image
    // edge) no samples and never reads their bins, so a wave made of such rays only has nothing to produce (no barrier follows)
    if (a.skip_miss && __all(!ok || ray_misses(a.rc, rs))) return;
    const uint32_t T = a.T;
    const float b0step = 1.0f / (float)T;                 // (1-0)/(steps-1), steps = T+1

    auto bin_at = [&](uint32_t j) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + r];
        if (a.bins0_tab) return a.bins0_tab[(size_t)n * a.bins0_stride + j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };

    // ---- pass 1: sigma -> weights (renderer.py:277-325) ----
    float bprev = bin_at(0);
    float rb_prev = real_bin(rs, bprev);
    if (ok && a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1)] = bprev;
    double cum = 0.0, wacc = 0.0;
    const TT *table = reinterpret_cast<const TT *>(a.table);
    const bool early_out = !a.dbg_bins && !a.dbg_sigma && !a.dbg_w;
    // samples j .. j+U-1; returns true once the transmittance of all 64 rays of the wave has underflowed to exactly 0
    auto march = [&](auto un_tag, uint32_t j) -> bool {
        constexpr int U = decltype(un_tag)::value;
        float bnext[U], rb_next[U], x01[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bnext[u] = bin_at(j + (uint32_t)u + 1u);
            rb_next[u] = real_bin(rs, bnext[u]);
            const float tmid = (rb_next[u] + (u ? rb_next[u ? u - 1 : 0] : rb_prev)) / 2.0f;
            float p[3];
            sample_x01(a.rc, rs, tmid, p, x01[u]);
        }
        float feat[U][L * C], raw[U];
        if constexpr (U == 1) encode_levels<TT, L, C, K, true, (sizeof(TT) == 4 ? SN_PROP_GROUP : SN_PROP_GROUP_H)>(table, a.g, x01[0], feat[0], &a.pairs);
        else encode_levels_multi<TT, L, C, K, true, SN_PROP_GROUP_U2, U>(table, a.g, x01, feat, &a.pairs);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float h[HID], r1[1];
            const uint32_t oz = opaque_zero();
            dense_ldsw_t<IN, HID, 1, (U > 1)>(lds_w0 + oz, feat[u], h);
            dense_ldsw<HID, 1, 0>(lds_w1 + oz, h, r1);
            raw[u] = r1[0];
        }
        bool dark = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ju = j + (uint32_t)u;
            const float rbp = u ? rb_next[u ? u - 1 : 0] : rb_prev;
            const float sigma = expf_det(raw[u]);                // trunc_exp forward (activation.py:9)
            const float delta = rb_next[u] - rbp;
            float ds = delta * sigma;
            if (a.rc.last_opaque && ju == T - 1u) ds = __builtin_inff();
            const float alpha = 1.0f - expf_det(-ds);
            const float tr = expf_det(-(float)cum);
            float w = alpha * tr;
            if (w != w) w = 0.0f;
            a.w_scr[(size_t)ju * Npad + r] = w;
            cum += (double)ds;
            wacc += (double)(w + 0.01f);
            if (ok) {
                if (a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1) + ju + 1] = bnext[u];
                if (a.dbg_sigma) a.dbg_sigma[(size_t)n * T + ju] = sigma;
                if (a.dbg_w) a.dbg_w[(size_t)n * T + ju] = w;
            }
            // EXACT early-out (round 4): once the transmittance of all 64 rays of the wave has underflowed to exactly 0 it stays 0 (the optical
            // depth never decreases), so every later weight is alpha * 0 = 0 whatever the density: the remaining samples are not evaluated, their
            // weights are written as 0 and the pdf normaliser keeps taking its (0 + 0.01) terms in the same order.  Opaque scenes only; bit-identical.
            // (with U > 1 the samples of the group that follow the underflow are still evaluated: their weights come out as the same zeros)
            if (u == U - 1) dark = early_out && __all(tr == 0.0f);
        }
        rb_prev = rb_next[U - 1];
        return dark;
    };
    {
        uint32_t j = 0;
        bool dark = false;
        if constexpr (UN > 1) {
            for (; j + (uint32_t)UN <= T && !dark; j += (uint32_t)UN) dark = march(std::integral_constant<int, UN>{}, j);
        }
        for (; j < T && !dark; ++j) dark = march(std::integral_constant<int, 1>{}, j);
        for (; j < T; ++j) {                                   // (only after an early-out)
            a.w_scr[(size_t)j * Npad + r] = 0.0f;
            wacc += (double)(0.0f + 0.01f);
        }
    }

    // ---- pass 2: sample_pdf (renderer.py:84-119) as one merge of cdf against u ----
    // The cdf is walked in blocks of PB entries: the block's weights and bins are fetched with independent, coalesced loads
    // (the former one-entry-at-a-time merge chained T dependent global loads per ray and cost 19 % of the stage), its PB
    // prefix values are formed in the oracle's order (fp64 running sum of the pdf, rounded per prefix, clamped at 1), and
    // every lane then emits the outputs whose searchsorted(right=True) count falls inside the block.
    const float wsum = (float)wacc;
    const uint32_t Tq = a.Tn + 1u;
    const float ustart = (float)(0.5 / Tq), uend = (float)(1 - 0.5 / Tq);
    const float ustep = (uend - ustart) / (float)(Tq - 1u);
    auto u_at = [&](uint32_t j) -> float { return a.u_tab ? a.u_tab[(size_t)n * a.u_stride + (j < Tq ? j : Tq - 1u)] : linspace_at(ustart, uend, ustep, Tq, j); };
    constexpr uint32_t PB = SN_PROP_PB;
    uint32_t jq = 0;
    float uj = u_at(0);
    double acc = 0.0;
    float c_start = 0.0f, b_start = bin_at(0);              // cdf[ib], bins[ib]
    for (uint32_t ib = 0; ib < T; ib += PB) {
        const bool last_block = ib + PB >= T;
        float e_c[PB + 1], e_b[PB + 1];                     // entries ib .. ib+PB of the cdf and the bins (past T: repeats of entry T)
        e_c[0] = c_start; e_b[0] = b_start;
        float wv[PB];
#pragma unroll
        for (uint32_t k = 0; k < PB; ++k) {
            const uint32_t idx = ib + k < T ? ib + k : T - 1u;
            wv[k] = a.w_scr[(size_t)idx * Npad + r];
            e_b[k + 1] = bin_at(idx + 1u);
        }
#pragma unroll
        for (uint32_t k = 0; k < PB; ++k) {
            if (ib + k < T) {                               // uniform
                const float pdf = (wv[k] + 0.01f) / wsum;
                acc += (double)pdf;
                const float c = (float)acc;
                e_c[k + 1] = c > 1.0f ? 1.0f : c;
            } else {
                e_c[k + 1] = e_c[k];
            }
        }
        const uint32_t nb = last_block ? T - ib : PB;       // real entries after e[0] in this block
        // outputs of this block: all whose u is below the block's last cdf value; in the last block all that remain
        while (jq < Tq && (last_block || e_c[PB] > uj)) {
            uint32_t cnt = 0;                                // entries e[0..nb] that are <= uj (e is non-decreasing)
#pragma unroll
            for (uint32_t m = 0; m <= PB; ++m) cnt += (m <= nb && e_c[m] <= uj) ? 1u : 0u;
            const uint32_t i = ib + cnt;                     // = searchsorted(cdf, u, right=True): every entry before ib is <= e[0]
            // below = entry i-1, above = entry i; i == 0 -> both entry 0; i > T -> both entry T (renderer.py:104-107 clamps)
            const uint32_t lo_m = cnt == 0u ? 0u : cnt - 1u, hi_m = cnt > nb ? nb : cnt;
            float c0 = e_c[0], c1 = e_c[0], bb0 = e_b[0], bb1 = e_b[0];
#pragma unroll
            for (uint32_t m = 1; m <= PB; ++m) {
                c0 = lo_m == m ? e_c[m] : c0; bb0 = lo_m == m ? e_b[m] : bb0;
                c1 = hi_m == m ? e_c[m] : c1; bb1 = hi_m == m ? e_b[m] : bb1;
            }
            if (cnt > nb) { c0 = c1; bb0 = bb1; }            // i > T: below = above = the last entry
            float t = nan_to_num((uj - c0) / (c1 - c0));
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            const float m = t * (bb1 - bb0);
            a.bins_out[(size_t)jq * Npad + r] = bb0 + m;
            if (ok && a.dbg_bins_next) a.dbg_bins_next[(size_t)n * Tq + jq] = bb0 + m;
            if (ok && a.dbg_inds) a.dbg_inds[(size_t)n * Tq + jq] = (int32_t)i;
            ++jq;
            uj = u_at(jq);
        }
        c_start = e_c[PB]; b_start = e_b[PB];
    }
}

// ------------------------------------------------------------------------------------------
// proposal stage, sample-parallel variant for small ray batches (training steps: a few thousand rays)
// ------------------------------------------------------------------------------------------
// k_prop_stage walks one ray per lane: 4096 rays are 64 waves on a 1024-SIMD chip, each alone with its memory
// latency, and a 128-sample stage takes 128 serial steps (0.55 ms for 4096 rays).  Here 8 lanes share a ray: each
// evaluates a contiguous chunk of T/8 samples (the expensive part: position, grid, MLP), everything that must keep
// the oracle's sequential order -- the fp64 optical-depth prefix, the cdf -- is done by the ray's first lane over
// values parked in LDS (a few adds per sample), and the T_next + 1 resampled bins are found by a binary search
// per output, 8 outputs at a time.  Same arithmetic, same order where order matters: bit-identical results.
// Linear ray order only (scratch column = ray index), T <= SP_MAX_T.
// Round 6: the lanes per ray are a template parameter (8, 16 or 32: 32, 16 or 8 rays per workgroup).  With 8 lanes a 4096-ray training batch is
// 128 workgroups on 256 CUs and every lane walks T/8 = 16 samples one after the other; the launcher now picks the smallest count that gives
// >= 512 workgroups (4096 rays: 32 lanes, 4 samples per lane).  The serial parts (prefix, cdf) and every sample's arithmetic are unchanged and
// the fp64 sum of the weights is exact in any order: bit-identical for every choice (sn_render_tuning.prop_sp_lanes forces one).
constexpr uint32_t SP_MAX_T = 128;        // 3 arrays x (256 / LPR) rays x (T+4) floats: 50 KiB of static LDS at 8 lanes per ray
constexpr uint32_t SP_STRIDE = SP_MAX_T + 4;   // floats per ray per LDS array

template <typename TT, int L, int C, int HID, int K, uint32_t SP_LPR = 8>
__global__ __launch_bounds__(256, 3) void k_prop_stage_sp(PropArgs a) {
    SN_POISON_ALL();
    static_assert(SP_LPR == 8 || SP_LPR == 16 || SP_LPR == 32, "a ray's lanes live in one wave");
    constexpr uint32_t RPB = 256u / SP_LPR;      // rays per workgroup
    constexpr int IN = L * C;
    __shared__ __attribute__((aligned(16))) float lds_w0[IN * PadIn<HID>::value];
    __shared__ __attribute__((aligned(16))) float lds_w1[PadIn<HID>::value];
    __shared__ float l_bins[RPB][SP_STRIDE];    // stage bins b_0..b_T of the rays of this workgroup
    __shared__ float l_ds[RPB][SP_STRIDE];      // delta*sigma, then reused for the weights
    __shared__ float l_cum[RPB][SP_STRIDE];     // (float) of the fp64 exclusive prefix of delta*sigma, then the cdf
    stage_weights_t<IN, HID>(lds_w0, a.w0);
    stage_weights<HID, 1>(lds_w1, a.w1);
    const uint32_t tid = threadIdx.x, c = tid & (SP_LPR - 1u), rl = tid / SP_LPR;      // chunk index, local ray
    const uint32_t n_raw = blockIdx.x * RPB + rl;
    const bool ok = n_raw < a.rc.N;
    const uint32_t n = ok ? n_raw : 0u, r = n_raw;                                     // scratch column = ray index (linear order)
    const uint32_t Npad = a.rc.Npad, T = a.T;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    const float b0step = 1.0f / (float)T;
    auto bin_src = [&](uint32_t j) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + r];
        if (a.bins0_tab) return a.bins0_tab[(size_t)n * a.bins0_stride + j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };
    for (uint32_t j = c; j <= T; j += SP_LPR) l_bins[rl][j] = bin_src(j);
    __syncthreads();                                                                   // weights + bins in LDS
    if (ok && a.dbg_bins) for (uint32_t j = c; j <= T; j += SP_LPR) a.dbg_bins[(size_t)n * (T + 1) + j] = l_bins[rl][j];

    // ---- per-sample work, T / SP_LPR consecutive samples per lane ----
    const uint32_t spl = (T + SP_LPR - 1u) / SP_LPR;
    const TT *table = reinterpret_cast<const TT *>(a.table);
    for (uint32_t i = 0; i < spl; ++i) {
        const uint32_t jr = c * spl + i;
        const uint32_t j = jr < T ? jr : T - 1u;                                       // lanes past the end redo the last sample
        const float rb_prev = real_bin(rs, l_bins[rl][j]), rb_next = real_bin(rs, l_bins[rl][j + 1u]);
        const float tmid = (rb_next + rb_prev) / 2.0f;
        float p[3], x01[3];
        sample_x01(a.rc, rs, tmid, p, x01);
        float feat[L * C];
        encode_levels<TT, L, C, K, true>(table, a.g, x01, feat, &a.pairs);
        float h[HID], raw[1];
        const uint32_t oz = opaque_zero();
        dense_ldsw_t<IN, HID, 1>(lds_w0 + oz, feat, h);
        dense_ldsw<HID, 1, 0>(lds_w1 + oz, h, raw);
        const float sigma = expf_det(raw[0]);
        float ds = (rb_next - rb_prev) * sigma;
        if (a.rc.last_opaque && j == T - 1u) ds = __builtin_inff();
        if (jr < T) {
            l_ds[rl][j] = ds;
            if (ok && a.dbg_sigma) a.dbg_sigma[(size_t)n * T + j] = sigma;
        }
    }
    __syncthreads();
    // ---- exclusive prefix of delta*sigma in the oracle's order: fp64 running sum, rounded per prefix ----
    if (c == 0u) {
        double cum = 0.0;
        for (uint32_t j = 0; j < T; ++j) { l_cum[rl][j] = (float)cum; cum += (double)l_ds[rl][j]; }
    }
    __syncthreads();
    // ---- weights (renderer.py:308-325); their fp64 sum is exact in any order (addends within 2^7 of each other) ----
    double wacc = 0.0;
    for (uint32_t i = 0; i < spl; ++i) {
        const uint32_t j = c * spl + i;
        if (j < T) {
            const float alpha = 1.0f - expf_det(-l_ds[rl][j]);
            const float tr = expf_det(-l_cum[rl][j]);
            float w = alpha * tr;
            if (w != w) w = 0.0f;
            wacc += (double)(w + 0.01f);
            l_ds[rl][j] = w;                                                             // own slot: no hazard
            a.w_scr[(size_t)j * Npad + r] = w;                                           // padding columns too, like k_prop_stage
            if (ok && a.dbg_w) a.dbg_w[(size_t)n * T + j] = w;
        }
    }
#pragma unroll
    for (int d = 1; d < (int)SP_LPR; d <<= 1) wacc += __shfl_xor(wacc, d, SP_LPR);
    const float wsum = (float)wacc;
    __syncthreads();
    // ---- cdf (renderer.py:92-96): fp64 running sum of the pdf, rounded per prefix, clamped at 1 ----
    if (c == 0u) {
        double acc = 0.0;
        l_cum[rl][0] = 0.0f;
        for (uint32_t j = 0; j < T; ++j) {
            const float pdf = (l_ds[rl][j] + 0.01f) / wsum;
            acc += (double)pdf;
            const float cv = (float)acc;
            l_cum[rl][j + 1u] = cv > 1.0f ? 1.0f : cv;
        }
    }
    __syncthreads();
    // ---- resample: searchsorted(cdf, u, right=True) per output, then the interpolation of renderer.py:104-119 ----
    const uint32_t Tq = a.Tn + 1u;
    const float ustart = (float)(0.5 / Tq), uend = (float)(1 - 0.5 / Tq);
    const float ustep = (uend - ustart) / (float)(Tq - 1u);
    for (uint32_t jq = c; jq < Tq; jq += SP_LPR) {
        const float uj = a.u_tab ? a.u_tab[(size_t)n * a.u_stride + jq] : linspace_at(ustart, uend, ustep, Tq, jq);
        uint32_t lo = 0u, hi = T + 1u;                                                   // number of cdf[0..T] entries <= uj
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (l_cum[rl][mid] <= uj) lo = mid + 1u; else hi = mid;
        }
        const uint32_t i = lo;
        float c0, c1, bb0, bb1;
        if (i == 0u) { c0 = c1 = l_cum[rl][0]; bb0 = bb1 = l_bins[rl][0]; }
        else if (i > T) { c0 = c1 = l_cum[rl][T]; bb0 = bb1 = l_bins[rl][T]; }
        else { c0 = l_cum[rl][i - 1u]; c1 = l_cum[rl][i]; bb0 = l_bins[rl][i - 1u]; bb1 = l_bins[rl][i]; }
        float t = nan_to_num((uj - c0) / (c1 - c0));
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        const float m = t * (bb1 - bb0);
        a.bins_out[(size_t)jq * Npad + r] = bb0 + m;
        if (ok && a.dbg_bins_next) a.dbg_bins_next[(size_t)n * Tq + jq] = bb0 + m;
        if (ok && a.dbg_inds) a.dbg_inds[(size_t)n * Tq + jq] = (int32_t)i;
    }
}

// ------------------------------------------------------------------------------------------
// final stage
// ------------------------------------------------------------------------------------------
struct FinalArgs {
    RayCommon rc;
    GridLevels g;
    const void *table;
    const float *mlp_pack;       // MFMA path: A-operand-ordered weights (floats, PACK_FLOATS)
    const float *w[3];           // VALU path: raw [out][in] weights of grid_mlp
    const float *vw[3];          // view_mlp weights
    uint32_t T;
    const float *bins_in;        // scratch [T+1][Npad] or NULL (single-stage)
    const float *bins0_tab;
    uint32_t bins0_stride;       // 0: bins0_tab is one shared table; else per ray [N][bins0_stride]
    uint32_t sh_degree;
    float *image, *depth, *wsum;
    uint32_t istride, sstride;   // floats between consecutive rays of image / of depth and wsum (3 and 1, or sn_render_io.out_stride for both)
    float *dbg_bins, *dbg_w, *dbg_sigma, *dbg_xyz, *dbg_geo, *dbg_fimg;
    uint32_t fimg_stride;        // floats between consecutive rays of dbg_fimg (geo + 16, or sn_render_io.head_stride)
    float *head_rgbd;            // sn_render_io.head_stride: rgb | depth of ray n also go to head_rgbd[n * fimg_stride + 0..3] (the SAM head's MLP input), or NULL
    float *w_out;                // scratch [T][Npad] for the feature stage, or NULL
    float stop_cum;              // > 0: a wave leaves the march once every lane's optical depth exceeds this (-ln eps)
    PairTab pairs;               // dense levels of the main grid as aligned x-pairs (K > 0 instantiations)
    FinalLv lv;                  // per-level constants of the K > 0 instantiations
};

// shader-clock probe of the measurement hook: s_memtime ticks at the shader clock, s_memrealtime at the constant
// wall-clock rate (hipDeviceAttributeWallClockRate); their ratio over one workgroup's lifetime is the clock the
// kernel actually ran at (DVFS: well below the 2.4 GHz peak under this kernel), which prices the cycle-count ceilings
// (device globals rather than a kernel argument: the final stage has no scalar registers to spare)
__device__ int g_clk_on = 0;
__device__ unsigned long long g_clk_buf[4];
__device__ __forceinline__ void clock_probe(int slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && g_clk_on) {
        g_clk_buf[2 * slot] = __builtin_readcyclecounter();
        g_clk_buf[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// A-operand packing for the 32->64->64->16 MLP on v_mfma_f32_32x32x2_f32.
//   D[m][j] += sum_k A[m][k] * B[k][j],  A = weights (m = output neuron), B = activations
//   (j = sample).  Lane l supplies A[m = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the
//   accumulator register r of lane l holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//   Because that accumulator layout IS a valid B-operand layout (low half-wave: row h0(r),
//   high half-wave: row h0(r)+4), layer n+1 consumes layer n's accumulators in place; only
//   the weights are permuted, once, here.
constexpr int PACK_L1 = 0;                       // [mt 2][step 16][64]
constexpr int PACK_L2 = PACK_L1 + 2 * 16 * 64;   // [mt 2][step 32][64]
constexpr int PACK_L3 = PACK_L2 + 2 * 32 * 64;   // [step 32][64]
constexpr int PACK_FLOATS = PACK_L3 + 32 * 64;   // 8192 floats = 32 KiB

__global__ void k_pack_grid_mlp(const float *__restrict__ w1, const float *__restrict__ w2, const float *__restrict__ w3,
                                float *__restrict__ pack) {
    SN_POISON_ALL();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)PACK_FLOATS) return;
    const uint32_t lane = t & 63u, vec = t >> 6;
    const uint32_t m = lane & 31u, hi = lane >> 5;
    float v;
    if (vec < 32u) {                       // layer 1: in 32 -> out 64
        const uint32_t mt = vec >> 4, s = vec & 15u;
        v = w1[(mt * 32u + m) * 32u + 2u * s + hi];
    } else if (vec < 96u) {                // layer 2: in 64 -> out 64
        const uint32_t q = vec - 32u, mt = q >> 5, s = q & 31u, src_mt = s >> 4, r = s & 15u;
        const uint32_t h = src_mt * 32u + (r & 3u) + 8u * (r >> 2) + 4u * hi;
        v = w2[(mt * 32u + m) * 64u + h];
    } else {                               // layer 3: in 64 -> out 16 (rows 16..31 are zero padding)
        const uint32_t s = vec - 96u, src_mt = s >> 4, r = s & 15u;
        const uint32_t h = src_mt * 32u + (r & 3u) + 8u * (r >> 2) + 4u * hi;
        v = m < 16u ? w3[m * 64u + h] : 0.0f;
    }
    pack[t] = v;
}

// max(t, 0) of a matrix-core result as one v_max_i32 on the bit pattern (negative floats are negative integers,
// -0 -> +0 like `t > 0 ? t : 0`).  The float form costs two instructions there: the compiler cannot prove an MFMA
// output canonical and puts a v_max_f32 t, t, t in front of every v_max_f32 0, t (145 of the final stage's 2236
// vector instructions per sample).  Differs from the select only for a NaN input with the sign bit clear.
__device__ __forceinline__ float relu_bits(float t) {
    const int b = __builtin_bit_cast(int, t);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ floatx16 relu16(floatx16 v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = relu_bits(v[i]);
    return v;
}

// 32 -> 64 -> 64 -> 16 for the wave's 64 samples (lane = sample).
// Features arrive through the wave's LDS slab fe[k][lane] (written by encode_levels_lds), which
// is already a B-operand image: step s of sample tile t reads fe[2s + (lane>>5)][32t + (lane&31)].
__device__ __forceinline__ void grid_mlp_mfma(const float *__restrict__ lds_pack, const float *__restrict__ fe, float (&out)[16]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lo = lane & 31u, hi = lane >> 5;
    float res[2][8];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        floatx16 h1[2], h2[2], o3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 16; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_pack[PACK_L1 + (mt * 16 + s) * 64 + lane],
                                                           fe[(2 * s + hi) * 64 + tile * 32 + lo], acc, 0, 0, 0);
            h1[mt] = relu16(acc);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 32; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_pack[PACK_L2 + (mt * 32 + s) * 64 + lane], h1[s >> 4][s & 15], acc, 0, 0, 0);
            h2[mt] = relu16(acc);
        }
        {
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 32; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_pack[PACK_L3 + s * 64 + lane], h2[s >> 4][s & 15], acc, 0, 0, 0);
            o3 = acc;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) res[tile][r] = o3[r];
    }
    // bring every sample's 16 outputs to its own lane: accumulator reg r of tile t holds rows
    // base(r) [low half-wave] / base(r)+4 [high half-wave] of sample 32t + (lane&31)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(res[0][r]), __float_as_uint(res[1][r]), false, false);
        const int base = (r & 3) + 8 * (r >> 2);
        out[base] = __uint_as_float(rr[0]);
        out[base + 4] = __uint_as_float(rr[1]);
    }
}

// ------------------------------------------------------------------------------------------
// fp16 hi/lo split path of the same MLP (MLP_F16X3, the default).
// fp32 MFMA issues at the vector-ALU rate (64 cycles per 32x32x2), which made the matrix pipe the
// co-bound of the final stage.  Writing every operand as x = hi + lo with hi = f16(x),
// lo = f16(x - hi) (22 significant bits) and accumulating the three products hi*hi + hi*lo + lo*hi
// in fp32 on v_mfma_f32_32x32x16_f16 costs 3 x 32 cycles per 32x32x16 block instead of 8 x 64:
// 5.3x fewer matrix-pipe cycles at ~2^-22 relative error per product (f16 x f16 is exact in fp32;
// the dropped lo*lo term is 2^-22).  Range: |activation| must stay below 65504.
//   A operand (weights):     lane l holds W[m = l&31][k = 8*(l>>5) + 0..7] of the k-step
//   B operand (activations): lane l holds X[k = 8*(l>>5) + 0..7][j = l&31]
//   C/D layout as for every 32x32 MFMA: reg r of lane l = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// so 8 consecutive accumulator registers of a lane are a valid B operand of the next layer
// (low half-wave rows {0-3, 8-11} / high half-wave rows {4-7, 12-15} of that 16-row block); only
// the weights' k order is permuted, once, by k_pack_grid_mlp_f16.
// ------------------------------------------------------------------------------------------
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
constexpr int PACK16_VECS = 4 + 8 + 4;                 // layer1 [mt 2][s 2], layer2 [mt 2][q 4], layer3 [q 4]
constexpr int PACK16_U4 = PACK16_VECS * 2 * 64;        // x (hi, lo) x 64 lanes, 16 B each = 32 KiB
constexpr int SLAB_STRIDE = 20;                        // dwords per sample row (16 levels + 4 pad: conflict-free b128 reads)

__device__ __forceinline__ void split2(float a, float b, uint32_t &hi, uint32_t &lo) {
    const half2_t h = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    // x - hi (exact in fp32: hi is x rounded to 11 bits) straight from the packed halves with the mixed-precision fma:
    // v_fma_mix_f32 widens its f16 operand for free, where the compiler's form was two v_cvt_f32_f16 + one v_pk_add_f32
    // (5 -> 4 instructions per pair, 80 pairs per sample in the final stage)
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(b));
    const half2_t l = {(_Float16)l0, (_Float16)l1};
    lo = __builtin_bit_cast(uint32_t, l);
}

__global__ void k_pack_grid_mlp_f16(const float *__restrict__ w1, const float *__restrict__ w2, const float *__restrict__ w3,
                                    uint4 *__restrict__ pack) {
    SN_POISON_ALL();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;        // one thread per (vec, lane)
    if (t >= (uint32_t)PACK16_VECS * 64u) return;
    const uint32_t lane = t & 63u, vec = t >> 6;
    const uint32_t m = lane & 31u, hi = lane >> 5;
    float v[8];
    if (vec < 4u) {                       // layer 1: k = 16 s + 8 hi + i  (feature 2*level + c)
        const uint32_t mt = vec >> 1, sst = vec & 1u;
        for (uint32_t i = 0; i < 8; ++i) v[i] = w1[(mt * 32u + m) * 32u + 16u * sst + 8u * hi + i];
    } else {
        const bool l3 = vec >= 12u;
        const uint32_t q = l3 ? vec - 12u : (vec - 4u) & 3u, mt = l3 ? 0u : (vec - 4u) >> 2;
        const uint32_t src_mt = q >> 1, half = q & 1u;
        for (uint32_t i = 0; i < 8; ++i) {
            const uint32_t r = 8u * half + i;
            const uint32_t h = src_mt * 32u + (r & 3u) + 8u * (r >> 2) + 4u * hi;
            v[i] = l3 ? (m < 16u ? w3[m * 64u + h] : 0.0f) : w2[(mt * 32u + m) * 64u + h];
        }
    }
    uint4 ph, pl;
    split2(v[0], v[1], ph.x, pl.x); split2(v[2], v[3], ph.y, pl.y);
    split2(v[4], v[5], ph.z, pl.z); split2(v[6], v[7], ph.w, pl.w);
    pack[(vec * 2u + 0u) * 64u + lane] = ph;
    pack[(vec * 2u + 1u) * 64u + lane] = pl;
}

__device__ __forceinline__ floatx16 mfma3(const uint4 &ah, const uint4 &al, const uint4 &bh, const uint4 &bl, floatx16 acc) {
    const half8_t Ah = __builtin_bit_cast(half8_t, ah), Al = __builtin_bit_cast(half8_t, al);
    const half8_t Bh = __builtin_bit_cast(half8_t, bh), Bl = __builtin_bit_cast(half8_t, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc, 0, 0, 0);   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc, 0, 0, 0);
    return acc;
}

// Reduced-product form of the split arithmetic (sn_render_tuning.mlp_mode = SN_MLP_F16X1; opt-in, NOT fp32-class, NOT within the 1e-4 bar):
//   NP = 3  Al*Bh + Ah*Bl + Ah*Bh   (the default: every product to 2^-22)
//   NP = 1  Ah*Bh                   plain fp16 operands with fp32 accumulation -- what an autocast fp16 run of the reference multiplies
// Measured in round 6 (profiles/r06/mlp_modes_ab.json): 800x800 [128] 6.19 -> 5.44 ms, and max |dRGB| against the reference's own output on
// the stress-init fixtures 3.2e-4 / 3.5e-4 (x3: 6e-7 / 7e-6) -- over the north star's 1e-4, so it can never be the default.  A two-product
// form (weights exact, activations rounded to fp16) was measured too: 1.9e-4 / 2.6e-4, i.e. over the bar as well, and dropped.
template <int NP>
__device__ __forceinline__ floatx16 mfma_np(const uint4 &ah, const uint4 &al, const uint4 &bh, const uint4 &bl, floatx16 acc) {
    if constexpr (NP == 3) return mfma3(ah, al, bh, bl, acc);
    const half8_t Ah = __builtin_bit_cast(half8_t, ah), Bh = __builtin_bit_cast(half8_t, bh);
    if constexpr (NP == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, al), Bh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    const half2_t h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, h);
}
// relu + fp16 rounding only (NP < 3: the lo image of an activation is never multiplied)
__device__ __forceinline__ void acc_to_b_hi(const floatx16 &v, int half, uint4 &bh) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = relu_bits(v[8 * half + i]);
    bh.x = pack_h2(x[0], x[1]); bh.y = pack_h2(x[2], x[3]); bh.z = pack_h2(x[4], x[5]); bh.w = pack_h2(x[6], x[7]);
}

// relu + split of 8 consecutive accumulator registers -> (hi, lo) B operand of the next layer
__device__ __forceinline__ void acc_to_b(const floatx16 &v, int half, uint4 &bh, uint4 &bl) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = relu_bits(v[8 * half + i]);
    split2(x[0], x[1], bh.x, bl.x); split2(x[2], x[3], bh.y, bl.y);
    split2(x[4], x[5], bh.z, bl.z); split2(x[6], x[7], bh.w, bl.w);
}

// slab_hi / slab_lo: this wave's [64 samples][SLAB_STRIDE dwords] images written by encode_levels_split
__device__ __forceinline__ void grid_mlp_mfma16(const uint4 *__restrict__ pk, const uint32_t *__restrict__ slab_hi,
                                                const uint32_t *__restrict__ slab_lo, float (&out)[16]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lo = lane & 31u, hi = lane >> 5;
    float res[2][8];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        const uint32_t row = (tile * 32u + lo) * SLAB_STRIDE + 4u * hi;
        floatx16 h1[2], h2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 bh = *reinterpret_cast<const uint4 *>(slab_hi + row + 8 * st);
                const uint4 bl = *reinterpret_cast<const uint4 *>(slab_lo + row + 8 * st);
                const int vec = mt * 2 + st;
                acc = mfma3(pk[(vec * 2 + 0) * 64 + lane], pk[(vec * 2 + 1) * 64 + lane], bh, bl, acc);
            }
            h1[mt] = acc;
            __builtin_amdgcn_sched_barrier(0);   // keep the next block's operand reads from being hoisted (VGPR pressure)
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 bh, bl;
                acc_to_b(h1[q >> 1], q & 1, bh, bl);
                const int vec = 4 + mt * 4 + q;
                acc = mfma3(pk[(vec * 2 + 0) * 64 + lane], pk[(vec * 2 + 1) * 64 + lane], bh, bl, acc);
            }
            h2[mt] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
        floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 bh, bl;
            acc_to_b(h2[q >> 1], q & 1, bh, bl);
            const int vec = 12 + q;
            acc = mfma3(pk[(vec * 2 + 0) * 64 + lane], pk[(vec * 2 + 1) * 64 + lane], bh, bl, acc);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) res[tile][r] = acc[r];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(res[0][r]), __float_as_uint(res[1][r]), false, false);
        const int base = (r & 3) + 8 * (r >> 2);
        out[base] = __uint_as_float(rr[0]);
        out[base + 4] = __uint_as_float(rr[1]);
    }
}

// The same MLP with BOTH 32-sample tiles of the wave advancing together: every weight operand is read from LDS once per
// wave-sample instead of once per tile (64 -> 32 ds_read_b128 of 1 KiB; the weight stream was 12 % of the final stage:
// timing with the reads removed, profiles/r03/ab_round3_experiments.txt) and the matrix pipe always has two to four
// independent accumulators in flight.  Every accumulator still receives exactly the same sequence of products (k-steps
// ascending, small terms first), so the results are bit-identical to grid_mlp_mfma16.  Costs 64 more live registers at
// the layer-2 peak (both tiles' h1 and all four layer-2 accumulators), paid for with a shorter prefetched gather span.
__device__ __forceinline__ void grid_mlp_mfma16_2t(const uint4 *__restrict__ pk, const uint32_t *__restrict__ slab_hi,
                                                   const uint32_t *__restrict__ slab_lo, float (&out)[16]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lo = lane & 31u, hi = lane >> 5;
    const floatx16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    floatx16 h1[2][2];                                    // [tile][mt]
#pragma unroll
    for (int t = 0; t < 2; ++t) { h1[t][0] = zero16; h1[t][1] = zero16; }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        uint4 bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t row = (t * 32u + lo) * SLAB_STRIDE + 4u * hi + 8u * st;
            bh[t] = *reinterpret_cast<const uint4 *>(slab_hi + row);
            bl[t] = *reinterpret_cast<const uint4 *>(slab_lo + row);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int vec = mt * 2 + st;
            const uint4 ah = pk[(vec * 2 + 0) * 64 + lane], al = pk[(vec * 2 + 1) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) h1[t][mt] = mfma3(ah, al, bh[t], bl[t], h1[t][mt]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    floatx16 h2[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { h2[t][0] = zero16; h2[t][1] = zero16; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint4 bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc_to_b(h1[t][q >> 1], q & 1, bh[t], bl[t]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int vec = 4 + mt * 4 + q;
            const uint4 ah = pk[(vec * 2 + 0) * 64 + lane], al = pk[(vec * 2 + 1) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) h2[t][mt] = mfma3(ah, al, bh[t], bl[t], h2[t][mt]);
        }
        __builtin_amdgcn_sched_barrier(0);                // one k-step at a time: h1 dies in halves behind this loop
    }
    floatx16 o3[2] = {zero16, zero16};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint4 bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc_to_b(h2[t][q >> 1], q & 1, bh[t], bl[t]);
        const int vec = 12 + q;
        const uint4 ah = pk[(vec * 2 + 0) * 64 + lane], al = pk[(vec * 2 + 1) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 2; ++t) o3[t] = mfma3(ah, al, bh[t], bl[t], o3[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(o3[0][r]), __float_as_uint(o3[1][r]), false, false);
        const int base = (r & 3) + 8 * (r >> 2);
        out[base] = __uint_as_float(rr[0]);
        out[base + 4] = __uint_as_float(rr[1]);
    }
}

// Slab row addressing.  Padded form: 20 dwords per sample row.  Swizzled form (SW): 16 dwords per row, the four 16-byte blocks
// of a row XOR-ed with (row >> 2) & 3 -- conflict-free for the ds_read_b128 of the B operands without the 4 padding dwords,
// which frees 8 KiB per workgroup for the LDS-resident coarse level below.
template <bool SW>
__device__ __forceinline__ uint32_t slab_dword(uint32_t row, uint32_t d) {
    if constexpr (SW) return row * 16u + ((((d >> 2) ^ (row >> 2)) & 3u) << 2) + (d & 3u);
    else return row * (uint32_t)SLAB_STRIDE + d;
}

// LDS-resident level 0 (north_star: "LDS staging of per-tile grid voxels"; the reference's only gesture at table locality is its
// level-major launch, gridencoder.cu:383-399).  The coarsest level of the main grid is 16^3 vertices = 16 KiB of fp16 rows: every
// workgroup stages it once and its 2 x 128 samples per lane read their 8 corners with ds_read_b32 instead of two 16-byte gathers
// through the texture path (98 -> 96 gather instructions per wave-sample).  Arithmetic as issue_level_lv's dense branch.
[[maybe_unused]] constexpr int L0_MAX_ROWS = 4096;   // (used by experiments builds only)
template <int l>
__device__ __forceinline__ void issue_level0_lds(const FinalLv &lv, const uint32_t *__restrict__ l0tab, const float (&x01)[3], float (&pos)[3],
                                                 Corner<__half, 2> (&cv)[8]) {
    static_assert(l == 0, "level 0 only");
    const float rf = lv.res_f[0];
    uint32_t cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = __builtin_fmaf(x01[d], rf, -0.5f);
        p = __builtin_amdgcn_fmed3f(p, 0.0f, lv.top_f[0]);
        cell[d] = (uint32_t)p;
        pos[d] = __builtin_amdgcn_fractf(p);
    }
    // vertex index in dwords: x + res * y + res^2 * z (the quad-row strides of FinalLv are 16 bytes per vertex: >> 4)
    const uint32_t sy = lv.d_sy[0] >> 4, sz = lv.d_sz[0] >> 4, top = (uint32_t)lv.top_f[0];
    const uint32_t X0 = cell[0], X1 = umin(X0 + 1u, top);
    const uint32_t Y0 = __umul24(cell[1], sy), Y1 = umin(Y0 + sy, lv.d_ylim[0] >> 4);
    const uint32_t Z0 = __umul24(cell[2], sz), Z1 = umin(Z0 + sz, lv.d_zlim[0] >> 4);
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        corner_set_half2(cv[i], l0tab[((i & 1u) ? X1 : X0) + ((i & 2u) ? Y1 : Y0) + ((i & 4u) ? Z1 : Z0)]);
    }
}

// "Linear tail" form of the final stage (k_final_stage<..., LT = true>).  The third layer has no activation behind it and the
// compositing is linear in what it produces, so for the 15 geometry channels
//     sum_j w_j * (W3[1:16] relu(h2_j))  =  W3[1:16] * (sum_j w_j relu(h2_j)),
// (renderer.py:332-336 composites `color`, network.py:175-186 forms it from grid_mlp's outputs 1..15): only the density row of
// the third layer is needed per sample -- one 64-term dot product, in true fp32 on the vector ALU -- while the geometry rows are
// applied ONCE per ray to the weight-accumulated hidden vector.  That takes the third layer off the matrix cores: 72 instead
// of 96 MFMAs per wave-sample and 48 instead of 64 KiB of LDS weight reads, which matters because the kernel runs at the
// clock the power management allows (the matrix-core MLP costs ~4 % in cycles and ~14 % in clock).
// A documented re-association like SH(d) * sum_j w_j; fp32 round-off class, not bit-identical to the
// per-sample form the other final-stage kernels (and per-sample geometry outputs) keep.
//
// layers 1 and 2 of ONE tile (32 samples): x[mt * 16 + r] = relu(h2) of hidden row mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)
template <bool SW = false, int NP = 3>
__device__ __forceinline__ void grid_mlp_mfma16_l12(const uint4 *__restrict__ pk, const uint32_t *__restrict__ slab_hi,
                                                    const uint32_t *__restrict__ slab_lo, int tile, float (&x)[32]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lo = lane & 31u, hi = lane >> 5;
    const uint32_t srow = (uint32_t)tile * 32u + lo;
    floatx16 h1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const uint32_t off = slab_dword<SW>(srow, 4u * hi + 8u * (uint32_t)st);
            const uint4 bh = *reinterpret_cast<const uint4 *>(slab_hi + off);
            uint4 bl = bh;
            if constexpr (NP == 3) bl = *reinterpret_cast<const uint4 *>(slab_lo + off);
            const int vec = mt * 2 + st;
            const uint4 ah = pk[(vec * 2 + 0) * 64 + lane];
            uint4 al = ah;
            if constexpr (NP >= 2) al = pk[(vec * 2 + 1) * 64 + lane];
            acc = mfma_np<NP>(ah, al, bh, bl, acc);
        }
        h1[mt] = acc;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 bh, bl;
            if constexpr (NP == 3) acc_to_b(h1[q >> 1], q & 1, bh, bl);
            else { acc_to_b_hi(h1[q >> 1], q & 1, bh); bl = bh; }
            const int vec = 4 + mt * 4 + q;
            const uint4 ah = pk[(vec * 2 + 0) * 64 + lane];
            uint4 al = ah;
            if constexpr (NP >= 2) al = pk[(vec * 2 + 1) * 64 + lane];
            acc = mfma_np<NP>(ah, al, bh, bl, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) x[mt * 16 + r] = relu_bits(acc[r]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// W3 re-ordered for the linear tail: w3p[half][o][i] = W3[o][mt*32 + (r&3) + 8*(r>>2) + 4*half], i = mt*16 + r -- the hidden
// rows a lane of that half-wave holds, in its register order (o = 0: density row, 1..15: geometry rows)
constexpr int W3P_FLOATS = 2 * 16 * 32;
constexpr int W3P_OFFSET = (12 * 2) * 64 * 4;            // floats: the packed layer-3 operands' place in the LDS weight image (unused here)
static_assert(W3P_OFFSET + W3P_FLOATS <= PACK_FLOATS, "the re-ordered W3 fits where the packed layer-3 operands were");
__device__ __forceinline__ void stage_w3p(float *__restrict__ dst, const float *__restrict__ w3) {
    for (uint32_t t = threadIdx.x; t < (uint32_t)W3P_FLOATS; t += blockDim.x) {
        const uint32_t i = t & 31u, o = (t >> 5) & 15u, half = t >> 9;
        const uint32_t mt = i >> 4, r = i & 15u;
        dst[t] = w3[o * 64u + mt * 32u + (r & 3u) + 8u * (r >> 2) + 4u * half];
    }
}
// sum_i w[i] * x[i] over the lane's 32 hidden rows: four interleaved ascending fmaf chains, combined (s0 + s1) + (s2 + s3)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot32_lds(const float *__restrict__ w, const float (&x)[32]) {
    f2v s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int i4 = 0; i4 < 8; ++i4) {
        const float4 ww = *reinterpret_cast<const float4 *>(w + 4 * i4);
        s01 = __builtin_elementwise_fma(f2v{ww.x, ww.y}, f2v{x[4 * i4 + 0], x[4 * i4 + 1]}, s01);
        s23 = __builtin_elementwise_fma(f2v{ww.z, ww.w}, f2v{x[4 * i4 + 2], x[4 * i4 + 3]}, s23);
    }
    return (s01.x + s01.y) + (s23.x + s23.y);
}

// hash-grid features of one position, split into f16 hi / lo and written to this lane's slab rows
// (dword l = the level's two features as a half2).
template <typename T, int L, int GROUP, int K>
__device__ __forceinline__ void encode_levels_split(const T *__restrict__ table, const GridLevels &g, const float (&x01)[3],
                                                    uint32_t *__restrict__ row_hi, uint32_t *__restrict__ row_lo) {
    const bool oob = encode_grouped<T, L, 2, GROUP, K, false>(table, g, x01, [&](int l, const float (&acc)[2]) {
        uint32_t ph, pl;
        split2(acc[0], acc[1], ph, pl);
        row_hi[l] = ph;
        row_lo[l] = pl;
    });
    if (__builtin_expect(__any(oob), 0)) {
        if (oob) for (int l = 0; l < L; ++l) { row_hi[l] = 0u; row_lo[l] = 0u; }
    }
}

// Scalar-pipe fallback of the same MLP (SN_RENDER_MLP=valu): activations live in per-thread LDS
// columns act[k][tid]; every neuron is one k-ascending fmaf chain (the oracle's order).
template <int IN, int OUT, int ACT>
__device__ __forceinline__ void dense_lds(const float *__restrict__ W, const float *__restrict__ xin, float *__restrict__ yout, uint32_t stride) {
    float x[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = xin[k * stride];
#pragma unroll 1
    for (int o = 0; o < OUT; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < IN; ++k) acc = __builtin_fmaf(W[o * IN + k], x[k], acc);
        if (ACT == 1) acc = __builtin_fmaxf(acc, 0.0f);       // ReLU as one v_max_f32 (= acc > 0 ? acc : 0 up to the sign of a zero; NaN -> 0 either way)
        yout[o * stride] = acc;
    }
}

// hash-grid features of one position written to an LDS column fe[k * stride]
template <typename T, int L, int C, int GROUP, int K, bool PAIRX = false>
__device__ __forceinline__ void encode_levels_lds(const T *__restrict__ table, const GridLevels &g, const float (&x01)[3],
                                                  float *__restrict__ fe, uint32_t stride) {
    const bool oob = encode_grouped<T, L, C, GROUP, K, PAIRX>(table, g, x01, [&](int l, const float (&acc)[C]) {
#pragma unroll
        for (int c = 0; c < C; ++c) fe[(l * C + c) * stride] = acc[c];
    });
    if (__builtin_expect(__any(oob), 0)) {
        if (oob) for (int i = 0; i < L * C; ++i) fe[i * stride] = 0.0f;
    }
}

enum { MLP_VALU = 0, MLP_F32 = 1, MLP_F16X3 = 2 };
#ifndef SN_RS_SPANS_H
#define SN_RS_SPANS_H 6      // role-split producers, fp16 tables: 6 spans like fp32 tables, or 4 spans of 4 levels
#endif
#ifndef SN_FINAL_WAVES
#ifndef SN_FINAL_SP_MIN_BLOCKS
#define SN_FINAL_SP_MIN_BLOCKS 256u    // the several-lanes-per-ray last stage takes fewer samples per lane until the launch has this many workgroups
                                       // (round 6, same box: 512 / 1024 are 3-8 % slower from 4096 to 16 384 rays -- one workgroup per CU is this kernel's optimum)
#endif
#define SN_FINAL_WAVES 2     // waves per SIMD k_final_stage is compiled for (register budget); experiments only
#endif

// AUX: the instantiation that also serves the feature stage (weights -> scratch) and the opt-in early termination;
// the plain one carries neither (one spilled register less in the march of the headline configuration)
template <typename TT, int L, int C, int H1, int H2, int NOUT, int VH, int MODE, int K, bool AUX = false, bool LT = false, bool L0L = false, bool EO = false, int NP = 3>
__global__ __launch_bounds__(256, MODE == MLP_VALU ? 1 : SN_FINAL_WAVES) void k_final_stage(FinalArgs a) {
    SN_POISON_ALL();
    static_assert(NP == 3 || (LT && !L0L && !AUX), "reduced-product MLP forms: plain linear-tail instantiations only");
    static_assert(!LT || (MODE == MLP_F16X3 && K >= 4 && K <= 8), "linear tail: split-fp16 MLP on the FinalLv path");
    static_assert(!L0L || (LT && sizeof(TT) == 2), "LDS-resident level 0: fp16 tables, linear-tail instantiation (80 KiB: 32 packed weights + 4 x 8 swizzled slabs + 16 level 0)");
    constexpr int SLAB_DW = L0L ? 16 : SLAB_STRIDE;       // dwords per slab row
    constexpr bool MFMA = MODE != MLP_VALU;
    constexpr int IN = L * C;
    constexpr int GEO = NOUT - 1;
    constexpr int NSH = 16;                       // sh degree 4 (network.py:97)
    constexpr int NCOL = GEO + NSH;
    static_assert(!MFMA || (IN == 32 && H1 == 64 && H2 == 64 && NOUT == 16), "MFMA path is the 32-64-64-16 MLP");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- LDS carve-up ----
    //  MFMA : [ packed weights 32 KiB | per-wave feature slabs fe[IN][64] x 4 ]
    //  VALU : [ activation columns actA[64][256] | actB[64][256] ]  (weights read through the scalar cache)
    float *fe;            // this lane's feature column base
    uint32_t fstride;     // distance between consecutive features of one lane
    float *actB = nullptr;
    constexpr int VW0 = VH * PadIn<NCOL>::value, VW1 = VH * PadIn<VH>::value;   // third layer: 3 * PadIn<VH> floats follow
    float *lds_vw;        // view_mlp weights (padded rows), all three layers back to back
    uint32_t *slab_hi = nullptr, *slab_lo = nullptr;      // MLP_F16X3: this wave's split feature images
    if constexpr (MODE == MLP_F16X3) {
        static_assert(PACK16_U4 * 4 == PACK_FLOATS, "both packings fill the same 32 KiB");
        for (uint32_t i = threadIdx.x; i < (uint32_t)PACK16_U4; i += 256u)
            reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.mlp_pack)[i];
        if constexpr (LT) { __syncthreads(); stage_w3p(lds + W3P_OFFSET, a.w[2]); }     // over the (unused) packed layer-3 operands
        constexpr int WAVE_SLAB = 2 * 64 * SLAB_DW;       // dwords: hi image + lo image
        float *wave_base = lds + PACK_FLOATS + (threadIdx.x >> 6) * WAVE_SLAB;
        slab_hi = reinterpret_cast<uint32_t *>(wave_base);
        slab_lo = slab_hi + 64 * SLAB_DW;
        if constexpr (L0L) {                              // level 0 of the main grid, rows as stored (half2 = one dword per vertex)
            uint32_t *l0w = reinterpret_cast<uint32_t *>(lds + PACK_FLOATS + 4 * WAVE_SLAB);
            const uint32_t rows0 = a.g.res[0] * a.g.res[0] * a.g.res[0];
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.table) + a.g.off[0];
            for (uint32_t i = threadIdx.x; i < rows0; i += 256u) l0w[i] = src[i];
        }
        fe = wave_base + (threadIdx.x & 63u);             // after the march the slab is reused as fp32 columns
        fstride = 64u;
        lds_vw = lds;                                     // overlays the packed weights once the march is over
    } else if constexpr (MODE == MLP_F32) {
        for (uint32_t i = threadIdx.x; i < (uint32_t)PACK_FLOATS / 4u; i += 256u)
            reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(a.mlp_pack)[i];
        fe = lds + PACK_FLOATS + (threadIdx.x >> 6) * (IN * 64) + (threadIdx.x & 63u);
        fstride = 64u;
        lds_vw = lds;
    } else {
        fe = lds + threadIdx.x;
        actB = lds + 64 * 256 + threadIdx.x;
        fstride = 256u;
        lds_vw = lds + 2 * 64 * 256;
    }
    if constexpr (!MFMA) {
        stage_weights<NCOL, VH>(lds_vw, a.vw[0]);
        stage_weights<VH, VH>(lds_vw + VW0, a.vw[1]);
        stage_weights<VH, 3>(lds_vw + VW0 + VW1, a.vw[2]);
    }
    __syncthreads();
    clock_probe(0);
    const uint32_t *l0tab = reinterpret_cast<const uint32_t *>(lds + PACK_FLOATS + 4 * 2 * 64 * SLAB_DW);   // L0L: level 0 of the main grid

    uint32_t n;
    const uint32_t wg = tile_id(a.rc);
    const bool ok = ray_of_lane(a.rc, wg, n);
    const uint32_t r = wg * 256u + threadIdx.x;
    const uint32_t Npad = a.rc.Npad;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    const uint32_t T = a.T;
    const float b0step = 1.0f / (float)T;
    auto bin_at = [&](uint32_t j) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + r];
        if (a.bins0_tab) return a.bins0_tab[(size_t)n * a.bins0_stride + j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };

    // view direction, normalised twice like the reference (renderer.py:294, sphere_harmonics.py:82)
    float dirn[3] = {rs.d[0], rs.d[1], rs.d[2]};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float aa = dirn[0] * dirn[0], bb = dirn[1] * dirn[1], cc = dirn[2] * dirn[2];
        const float nrm = sqrtf((aa + bb) + cc);
        dirn[0] = dirn[0] / nrm; dirn[1] = dirn[1] / nrm; dirn[2] = dirn[2] / nrm;
    }

    float fimg[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) fimg[c] = 0.0f;
    float dep = 0.0f;
    double cum = 0.0, wsum = 0.0;
    float tr_seen = 1.0f;                                  // this lane's ray: transmittance in front of the sample just evaluated
    uint32_t j_stop = a.T;                                 // first sample index NOT evaluated (exact early-out below)
    const bool per_sample_export = a.dbg_bins || a.dbg_w || a.dbg_sigma || a.dbg_xyz || a.dbg_geo;
    const TT *table = reinterpret_cast<const TT *>(a.table);
    // linear tail: sum_j w_j relu(h2_j) of the lane's 32 hidden rows, for the tile-0 and the tile-1 sample it shares
    float hacc[LT ? 2 : 1][LT ? 32 : 1];
    if constexpr (LT) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 32; ++i) hacc[t][i] = 0.0f;
    }
    const float *w3p = lds + W3P_OFFSET + (threadIdx.x & 32u) * 16u;          // this half-wave's rows: [o][32]

    float bprev = bin_at(0);
    float rb_prev = real_bin(rs, bprev);
    if (ok && a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1)] = bprev;
    // software pipeline (MLP_F16X3): group 0 of sample j+1 is issued before the matrix-core phase of sample j
    constexpr int PG = 4;                                   // levels per gather group (generic instantiations)
    constexpr bool LV = MODE == MLP_F16X3 && K >= PG && K <= 8;   // FinalLv path; the prefetched span is all dense
    using SP = FinalSpans<LV ? (LT ? (sizeof(TT) == 2 ? SN_FINAL_SPANS_LT_H : SN_FINAL_SPANS_LT) : 0) : 0>;   // FinalLv path: gather spans (span 0 crosses the matrix-core phase)
    constexpr int G0 = LV ? SP::B[1] : PG;
    static_assert(!LV || (L == 16 && SP::B[SP::N] == L && G0 <= K), "spans cover the 16 levels; span 0 is dense");
    GroupRegs<TT, 2, G0> g0;
    float bnext_n = bin_at(1);
    float rb_next_n = real_bin(rs, bnext_n);
    float tmid_n = (rb_next_n + rb_prev) / 2.0f;
    float p_n[3], x01_n[3];
    sample_x01(a.rc, rs, tmid_n, p_n, x01_n);
    bool fast_n = false;
    if constexpr (LV) {
        fast_n = all_interior(a.lv, x01_n);
        if constexpr (L0L) {
            issue_level0_lds<0>(a.lv, l0tab, x01_n, g0.pos[0], g0.cv[0]);             // level 0 from LDS, the rest of span 0 through the texture path
            static_for<1, G0>([&](auto kk) { constexpr int k = decltype(kk)::value; issue_level_lv<TT, true, false, false, k>(a.lv, x01_n, g0.pos[k], g0.cv[k]); });
        } else issue_span_lv<TT, 0, G0, K, false>(a.lv, x01_n, g0);
        __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (MODE == MLP_F16X3) {
        issue_group<TT, 2, PG, K, 0>(table, a.g, x01_n, g0, a.pairs);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t j = 0; j < T; ++j) {
        const float bnext = bnext_n;
        const float rb_next = rb_next_n;
        const float tmid = tmid_n;
        float p[3] = {p_n[0], p_n[1], p_n[2]}, x01[3] = {x01_n[0], x01_n[1], x01_n[2]};
        float h[NOUT];
        if constexpr (MODE == MLP_F16X3) {
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t *row_hi = slab_hi + lane * SLAB_STRIDE, *row_lo = slab_lo + lane * SLAB_STRIDE;
            auto emit = [&](int l, const float (&acc)[2]) {
                if constexpr (NP < 3) { row_hi[l] = pack_h2(acc[0], acc[1]); return; }      // fp16-rounded features: no lo image
                uint32_t ph, pl;
                split2(acc[0], acc[1], ph, pl);
                if constexpr (L0L) { const uint32_t o = slab_dword<true>(lane, (uint32_t)l); slab_hi[o] = ph; slab_lo[o] = pl; }
                else { row_hi[l] = ph; row_lo[l] = pl; }
            };
            if constexpr (LV) blend_span<TT, 0, G0, K>(g0, emit);
            else blend_group<TT, 2, PG, K, 0>(g0, emit);
            __builtin_amdgcn_sched_barrier(0);
            auto zero_oob = [&]() {
                const bool oob = (x01[0] < 0.0f || x01[0] > 1.0f) || (x01[1] < 0.0f || x01[1] > 1.0f) || (x01[2] < 0.0f || x01[2] > 1.0f);
                if (__builtin_expect(__any(oob), 0)) {      // gridencoder.cu:105-130: zeros outside [0,1]
                    if (oob) for (int l = 0; l < L; ++l) {
                        if constexpr (L0L) { const uint32_t o = slab_dword<true>(lane, (uint32_t)l); slab_hi[o] = 0u; slab_lo[o] = 0u; }
                        else { row_hi[l] = 0u; row_lo[l] = 0u; }
                    }
                }
            };
            if constexpr (LV) {
                auto rest = [&](auto fast_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    static_for<1, SP::N>([&](auto gg) {
                        constexpr int SPAN = decltype(gg)::value, L0 = SP::B[SPAN], G = SP::B[SPAN + 1] - L0;
                        GroupRegs<TT, 2, G> gr;
                        issue_span_lv<TT, L0, G, K, FAST>(a.lv, x01, gr);
                        __builtin_amdgcn_sched_barrier(0);
                        blend_span<TT, L0, G, K>(gr, emit);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
                if (__builtin_expect(fast_n, 1)) rest(std::true_type{});          // fast_n: decided for this sample one iteration ago
                else { rest(std::false_type{}); zero_oob(); }
            } else {
                static_for<1, L / PG>([&](auto gg) {
                    constexpr int GRP = decltype(gg)::value;
                    GroupRegs<TT, 2, PG> gr;
                    issue_group<TT, 2, PG, K, GRP>(table, a.g, x01, gr, a.pairs);
                    __builtin_amdgcn_sched_barrier(0);
                    blend_group<TT, 2, PG, K, GRP>(gr, emit);
                    __builtin_amdgcn_sched_barrier(0);
                });
                zero_oob();
            }
            {   // geometry of the next sample (the last iteration re-issues its own sample: in bounds, unused)
                const uint32_t jn = j + 2u <= T ? j + 2u : T;
                bnext_n = bin_at(jn);
                rb_next_n = real_bin(rs, bnext_n);
                const float rbp = j + 2u <= T ? rb_next : rb_prev;
                tmid_n = (rb_next_n + rbp) / 2.0f;
                sample_x01(a.rc, rs, tmid_n, p_n, x01_n);
                if constexpr (LV) {
                    fast_n = all_interior(a.lv, x01_n);
                    if constexpr (L0L) {
                        issue_level0_lds<0>(a.lv, l0tab, x01_n, g0.pos[0], g0.cv[0]);             // level 0 from LDS, the rest of span 0 through the texture path
                        static_for<1, G0>([&](auto kk) { constexpr int k = decltype(kk)::value; issue_level_lv<TT, true, false, false, k>(a.lv, x01_n, g0.pos[k], g0.cv[k]); });
                    } else issue_span_lv<TT, 0, G0, K, false>(a.lv, x01_n, g0);
                } else {
                    issue_group<TT, 2, PG, K, 0>(table, a.g, x01_n, g0, a.pairs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (!LT) grid_mlp_mfma16(reinterpret_cast<const uint4 *>(lds) + opaque_zero(), slab_hi, slab_lo, h);
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (MODE == MLP_F32) {
            encode_levels_lds<TT, L, C, 2, K>(table, a.g, x01, fe, fstride);
            // the slab is private to this wave and LDS serves a wave's requests in order
            __builtin_amdgcn_wave_barrier();
            grid_mlp_mfma(lds + opaque_zero(), fe - (threadIdx.x & 63u), h);
            __builtin_amdgcn_wave_barrier();
        } else {
            encode_levels_lds<TT, L, C, 2, K>(table, a.g, x01, fe, fstride);
            dense_lds<IN, H1, 1>(a.w[0], fe, actB, fstride);
            dense_lds<H1, H2, 1>(a.w[1], actB, fe, fstride);
            dense_lds<H2, NOUT, 0>(a.w[2], fe, actB, fstride);
#pragma unroll
            for (int k = 0; k < NOUT; ++k) h[k] = actB[k * fstride];
        }
        if constexpr (LT && (sizeof(TT) == 2 ? 0 : 0)) {
            // tile by tile: layers 1-2, density dot, the compositing step of that tile's home lanes, the accumulator update -- the
            // 32 activations of tile 0 are dead before tile 1's matrix phase begins (one compositing pass over both tiles would hold
            // them across it: 32 registers the kernel does not have)
            const float delta = rb_next - rb_prev;
            const uint32_t oz = opaque_zero();
            const uint32_t my_half = (threadIdx.x >> 5) & 1u;
            static_for<0, 2>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                float x[32];
                grid_mlp_mfma16_l12<L0L>(reinterpret_cast<const uint4 *>(lds) + oz, slab_hi, slab_lo, t, x);
                const float part = dot32_lds(w3p + oz, x);                            // this half's share of the density row
                auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                const float h0 = __uint_as_float(pr[0]) + __uint_as_float(pr[1]);     // half 0's share + half 1's share, in every lane
                float wv = 0.0f;
                if (my_half == (uint32_t)t) {                                         // home lanes of this tile's samples
                    const float sigma = expf_det(h0);                                 // network.py:151
                    float ds = delta * sigma;
                    if (a.rc.last_opaque && j == T - 1u) ds = __builtin_inff();
                    const float alpha = 1.0f - expf_det(-ds);
                    const float tr = expf_det(-(float)cum);
                    tr_seen = tr;
                    float w = alpha * tr;
                    if (w != w) w = 0.0f;
                    cum += (double)ds;
                    wsum += (double)w;
                    dep = __builtin_fmaf(w, tmid, dep);
                    if constexpr (AUX) { if (a.w_out) a.w_out[(size_t)j * Npad + r] = w; }
                    wv = w;
                }
                // every lane holds 32 hidden rows of this tile's sample (lane & 31): its weight comes from the home lane
                auto ww = __builtin_amdgcn_permlane32_swap(__float_as_uint(wv), __float_as_uint(wv), false, false);
                const float wt = __uint_as_float(ww[t]);
#pragma unroll
                for (int i = 0; i < 32; i += 2) {
                    const f2v r = __builtin_elementwise_fma(f2v{wt, wt}, f2v{x[i], x[i + 1]}, f2v{hacc[t][i], hacc[t][i + 1]});
                    hacc[t][i] = r.x; hacc[t][i + 1] = r.y;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (AUX) { if (a.stop_cum > 0.0f && __all((float)cum > a.stop_cum)) break; }
            if constexpr (EO) { if (!per_sample_export && __all(tr_seen == 0.0f)) { j_stop = j + 1u; break; } }      // exact early-out (see below)
            rb_prev = rb_next;
            continue;
        }
        float xh[LT ? 2 : 1][LT ? 32 : 1];
        if constexpr (LT) {
            const uint32_t oz = opaque_zero();
            float part[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                grid_mlp_mfma16_l12<L0L, NP>(reinterpret_cast<const uint4 *>(lds) + oz, slab_hi, slab_lo, t, xh[t]);
                part[t] = dot32_lds(w3p + oz, xh[t]);                                  // this half's share of the density row
            }
            __builtin_amdgcn_wave_barrier();
            // lanes 0-31 are the home of the tile-0 samples, lanes 32-63 of the tile-1 samples: half 0's share + half 1's share
            auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[0]), __float_as_uint(part[1]), false, false);
            h[0] = __uint_as_float(pr[0]) + __uint_as_float(pr[1]);
        }
        const float sigma = expf_det(h[0]);                  // network.py:151
        const float delta = rb_next - rb_prev;
        float ds = delta * sigma;
        if (a.rc.last_opaque && j == T - 1u) ds = __builtin_inff();
        const float alpha = 1.0f - expf_det(-ds);
        const float tr = expf_det(-(float)cum);
        tr_seen = tr;
        float w = alpha * tr;
        if (w != w) w = 0.0f;
        cum += (double)ds;
        wsum += (double)w;
        dep = __builtin_fmaf(w, tmid, dep);
        if constexpr (LT) {
            // every lane needs the weights of BOTH samples whose hidden rows it holds: w of ray (lane & 31) and of ray 32 + (lane & 31)
            auto ww = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
            const float w0 = __uint_as_float(ww[0]), w1 = __uint_as_float(ww[1]);
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const f2v r0 = __builtin_elementwise_fma(f2v{w0, w0}, f2v{xh[0][i], xh[0][i + 1]}, f2v{hacc[0][i], hacc[0][i + 1]});
                const f2v r1 = __builtin_elementwise_fma(f2v{w1, w1}, f2v{xh[1][i], xh[1][i + 1]}, f2v{hacc[1][i], hacc[1][i + 1]});
                hacc[0][i] = r0.x; hacc[0][i + 1] = r0.y; hacc[1][i] = r1.x; hacc[1][i + 1] = r1.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < GEO; ++c) fimg[c] = __builtin_fmaf(w, h[1 + c], fimg[c]);
        }
        if constexpr (AUX) { if (a.w_out) a.w_out[(size_t)j * Npad + r] = w; }
        if constexpr (!LT) if (ok) {                         // (the linear-tail instantiation is not chosen when per-sample tensors are wanted)
            if (a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1) + j + 1] = bnext;
            if (a.dbg_sigma) a.dbg_sigma[(size_t)n * T + j] = sigma;
            if (a.dbg_w) a.dbg_w[(size_t)n * T + j] = w;
            if (a.dbg_xyz) { float *q = a.dbg_xyz + ((size_t)n * T + j) * 3; q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
            if (a.dbg_geo) { float *q = a.dbg_geo + ((size_t)n * T + j) * GEO;
#pragma unroll
                for (int c = 0; c < GEO; ++c) q[c] = h[1 + c]; }
        }
        // opt-in early termination (not reference behaviour): all 64 rays of the wave are opaque to
        // within eps -> the remaining samples could add at most eps to any weight sum
        if constexpr (AUX) { if (a.stop_cum > 0.0f && __all((float)cum > a.stop_cum)) break; }
        // EXACT early-out (round 4, always on when no per-sample tensor is exported): once the transmittance exp(-optical depth) of all 64
        // rays of the wave has underflowed to exactly 0 it stays 0 (the optical depth never decreases), every later weight is alpha * 0 = 0
        // and adds nothing to image, depth or weights_sum: the march ends.  Opaque scenes only (optical depth > 103); bit-identical.  A template
        // switch (EO instantiations, chosen by sn_render_tuning.exact_early_out): in the kernel of the single-stage bench line the test --
        // a compare and a branch per sample -- perturbs the schedule by 0.5-1 % (same-box A/B), so that launch keeps the plain instantiation.
        if constexpr (EO) { if (!per_sample_export && __all(tr_seen == 0.0f)) { j_stop = j + 1u; break; } }
        rb_prev = rb_next;
        if constexpr (MODE != MLP_F16X3) {   // un-pipelined modes: geometry of the next sample
            const uint32_t jn = j + 2u <= T ? j + 2u : T;
            bnext_n = bin_at(jn);
            rb_next_n = real_bin(rs, bnext_n);
            tmid_n = (rb_next_n + (j + 2u <= T ? rb_next : rb_prev)) / 2.0f;
            sample_x01(a.rc, rs, tmid_n, p_n, x01_n);
        }
    }

    clock_probe(1);
    if constexpr (AUX) {            // the feature stage reads every weight: the samples behind the exact early-out carry weight 0 (and are skipped there too)
        if (a.w_out) for (uint32_t jj = j_stop; jj < T; ++jj) a.w_out[(size_t)jj * Npad + r] = 0.0f;
    }
    if constexpr (LT) {
        // geometry channels, once per ray: W3[1 + c] . (sum_j w_j relu(h2_j)); both tiles' shares, then the half-wave exchange
#pragma unroll 1
        for (int c = 0; c < GEO; ++c) {
            const float *wr = w3p + (1 + c) * 32;
            const float q0 = dot32_lds(wr, hacc[0]), q1 = dot32_lds(wr, hacc[1]);
            auto qq = __builtin_amdgcn_permlane32_swap(__float_as_uint(q0), __float_as_uint(q1), false, false);
            const float v = __uint_as_float(qq[0]) + __uint_as_float(qq[1]);
            // (runtime c: keeps the loop rolled; fimg is indexed statically below)
            switch (c) {
#define SN_FIMG_CASE(I) case I: fimg[I] = v; break;
                SN_FIMG_CASE(0) SN_FIMG_CASE(1) SN_FIMG_CASE(2) SN_FIMG_CASE(3) SN_FIMG_CASE(4) SN_FIMG_CASE(5) SN_FIMG_CASE(6) SN_FIMG_CASE(7)
                SN_FIMG_CASE(8) SN_FIMG_CASE(9) SN_FIMG_CASE(10) SN_FIMG_CASE(11) SN_FIMG_CASE(12) SN_FIMG_CASE(13) SN_FIMG_CASE(14)
#undef SN_FIMG_CASE
                default: break;
            }
        }
    }
    // ---- per-ray colour head: view_mlp(f_image) -> sigmoid -> + (1 - wsum) * bg (renderer.py:340-357) ----
    static_assert(VH <= IN && NCOL <= IN && IN * 64 <= 2 * 64 * SLAB_DW, "view MLP activations reuse the feature column / slab");
    if constexpr (MFMA) {
        // every wave marches the same T steps: once all are done the 32 KiB of packed MLP weights are dead and
        // the view-MLP weights take their place (keeps the workgroup under 80 KiB of LDS = 2 workgroups per CU)
        __syncthreads();
        stage_weights<NCOL, VH>(lds_vw, a.vw[0]);
        stage_weights<VH, VH>(lds_vw + VW0, a.vw[1]);
        stage_weights<VH, 3>(lds_vw + VW0 + VW1, a.vw[2]);
        __syncthreads();
    }
    __builtin_amdgcn_wave_barrier();
    float rgb[3];
    const float ws = (float)wsum;
    {   // sum_t w_t * SH_c(d) = SH_c(d) * sum_t w_t: the direction is constant along the ray
        // (renderer.py:293-295 evaluates it per sample), so the basis is evaluated once per ray.
        float sh[NSH];
        sh_degree4(dirn[0], dirn[1], dirn[2], sh);
#pragma unroll
        for (int c = 0; c < NSH; ++c) fimg[GEO + c] = sh[c] * ws;
    }
#pragma unroll
    for (int c = 0; c < NCOL; ++c) fe[c * fstride] = fimg[c];
    dense_ldsw_col<NCOL, VH, 1>(lds_vw, fe, fe, fstride);
    dense_ldsw_col<VH, VH, 1>(lds_vw + VW0, fe, fe, fstride);
    {
        float v2[VH];
#pragma unroll
        for (int k = 0; k < VH; ++k) v2[k] = fe[k * fstride];
        dense_ldsw<VH, 3, 0>(lds_vw + VW0 + VW1, v2, rgb);
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sg = 1.0f / (1.0f + expf_det(-rgb[c]));
            const float bgm = (1.0f - ws) * a.rc.bg;
            a.image[(size_t)n * a.istride + c] = sg + bgm;
        }
        a.depth[(size_t)n * a.sstride] = dep;
        a.wsum[(size_t)n * a.sstride] = ws;
        if (a.dbg_fimg) {
#pragma unroll
            for (int c = 0; c < NCOL; ++c) a.dbg_fimg[(size_t)n * a.fimg_stride + c] = fimg[c];
        }
        if (a.head_rgbd) {
            float *q = a.head_rgbd + (size_t)n * a.fimg_stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = a.image[(size_t)n * a.istride + c];
            q[3] = dep;
        }
    }
}

// ------------------------------------------------------------------------------------------
// final stage for fields of OTHER sizes (k_final_stage_any)
// ------------------------------------------------------------------------------------------
// renderer.py:221-357 does not care how large the field is: run() calls self.density / self(...) of whatever subclass it is
// given (BASELINE configs[0]: L = 8 levels, T = 2^14, a 16-32-16 grid_mlp and a 31-32-3 view_mlp).  The kernels above are
// instantiated for NeRFNetwork's own sizes (network.py:93-98); this one takes every size at run time, for any field of the
// same STRUCTURE -- hash / tiled grid with level_dim 2 (up to 16 levels), bias-free ReLU grid_mlp -> [sigma_raw | geo_feat],
// trunc_exp, compositing, degree-4 SH of the view direction, bias-free ReLU view_mlp, sigmoid -- with up to 4 layers per
// MLP, layers up to 64 wide and up to 31 geometry channels.  One lane = one ray as everywhere; a lane's activations live in
// its own LDS columns ([row][256 lanes]: two ping-pong buffers of 16 / 32 / 64 rows + the compositing rows; 160 KiB at the widest),
// the weights are wave-uniform and come through the scalar cache, every neuron is one k-ascending fmaf chain and the grid
// blend is k_grid_forward's (grid.hip) -- i.e. the oracle's arithmetic in the oracle's order.  No matrix cores: a field
// this small is bound by its launch, and the proposal stages in front of it (if the field has the reference's) stay fused.
struct AnyShape {
    uint32_t ng, nv;               // linear layers of grid_mlp / view_mlp
    uint32_t dg[5], dv[5];         // widths: dg[0] = L * 2 ... dg[ng] = 1 + geo;  dv[0] = geo + 16 ... dv[nv] = 3
    const float *wg[4], *wv[4];    // nn.Linear.weight [out][in]
    uint32_t rows;                 // LDS rows per activation buffer (16 / 32 / 64); 2 * rows + geo rows of 256 floats in all
};
constexpr uint32_t ANY_W = 64, ANY_GEO = 31, ANY_LAYERS = 4;

// one layer: the lane's inputs come out of its LDS column ONCE into registers (a first version read them inside the k loop: one LDS round
// trip per multiply-add, 2.2 ms for a 64x64 image of the configs[0] field); the weights are wave-uniform (scalar loads, 8 at a time); the k
// loop is unrolled for the bucket INB >= in, whole groups of 8 beyond `in` are skipped, a partial group multiplies clamped weights by 0 --
// fmaf(0, x, acc) = acc -- so every neuron is still the oracle's k-ascending chain
template <int INB>
__device__ __forceinline__ void dense_any_b(const float *__restrict__ W, uint32_t in, uint32_t out, bool relu, const float *xin, float *yout) {
    float x[INB];
#pragma unroll
    for (int k = 0; k < INB; ++k) {      // rows beyond `in` exist (buffers hold the bucket) but may hold anything, NaN included: 0 * NaN is NaN, so they read as 0
        const float t = xin[(uint32_t)k * 256u];
        x[k] = (uint32_t)k < in ? t : 0.0f;
    }
    // four neurons at a time: their scalar weight loads are in flight together and their four chains interleave (one wave per SIMD when the
    // batch is small: a single dependent chain would run at the latency of every instruction)
    for (uint32_t o0 = 0; o0 < out; o0 += 4u) {
        const float *w[4];
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) w[q] = W + (size_t)umin(o0 + q, out - 1u) * in;      // (a tail group repeats the last neuron: not stored)
#pragma unroll
        for (int g = 0; g < INB / 8; ++g) {
            if (8u * (uint32_t)g < in) {
                if (8u * (uint32_t)g + 8u <= in) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (uint32_t q = 0; q < 4u; ++q) acc[q] = __builtin_fmaf(w[q][8 * g + j], x[8 * g + j], acc[q]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t kk = 8u * (uint32_t)g + (uint32_t)j;
#pragma unroll
                        for (uint32_t q = 0; q < 4u; ++q) {
                            const float wk = w[q][kk < in ? kk : in - 1u];
                            acc[q] = __builtin_fmaf(kk < in ? wk : 0.0f, x[8 * g + j], acc[q]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            if (o0 + q < out) yout[(o0 + q) * 256u] = relu ? __builtin_fmaxf(acc[q], 0.0f) : acc[q];
        }
    }
}
__device__ __forceinline__ void dense_any(const float *__restrict__ W, uint32_t in, uint32_t out, bool relu, const float *xin, float *yout) {
    if (in <= 16u) dense_any_b<16>(W, in, out, relu, xin, yout);
    else if (in <= 32u) dense_any_b<32>(W, in, out, relu, xin, yout);
    else dense_any_b<64>(W, in, out, relu, xin, yout);
}

template <typename TT>
__global__ __launch_bounds__(256) void k_final_stage_any(FinalArgs a, AnyShape s) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t rows = s.rows;                      // rows of a ping-pong buffer: the widest layer, rounded up to the unroll bucket
    float *bufA = lds + threadIdx.x, *bufB = lds + rows * 256u + threadIdx.x, *accf = lds + 2u * rows * 256u + threadIdx.x;
    uint32_t n;
    const uint32_t wg = tile_id(a.rc);
    const bool ok = ray_of_lane(a.rc, wg, n);
    const uint32_t r = wg * 256u + threadIdx.x;
    const uint32_t Npad = a.rc.Npad;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    const uint32_t T = a.T, L = a.g.L, GEO = s.dg[s.ng] - 1u, NCOL = GEO + 16u;
    const float b0step = 1.0f / (float)T;
    auto bin_at = [&](uint32_t j) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + r];
        if (a.bins0_tab) return a.bins0_tab[(size_t)n * a.bins0_stride + j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };
    float dirn[3] = {rs.d[0], rs.d[1], rs.d[2]};     // normalised twice like the reference (renderer.py:294, sphere_harmonics.py:82)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float aa = dirn[0] * dirn[0], bb = dirn[1] * dirn[1], cc = dirn[2] * dirn[2];
        const float nrm = sqrtf((aa + bb) + cc);
        dirn[0] = dirn[0] / nrm; dirn[1] = dirn[1] / nrm; dirn[2] = dirn[2] / nrm;
    }
    for (uint32_t c = 0; c < GEO; ++c) accf[c * 256u] = 0.0f;
    float dep = 0.0f;
    double cum = 0.0, wsum = 0.0;
    const TT *table = reinterpret_cast<const TT *>(a.table);
    float bprev = bin_at(0);
    float rb_prev = real_bin(rs, bprev);
    if (ok && a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1)] = bprev;
    for (uint32_t j = 0; j < T; ++j) {
        const float bnext = bin_at(j + 1u);
        const float rb_next = real_bin(rs, bnext);
        const float tmid = (rb_next + rb_prev) / 2.0f;
        float p[3], x01[3];
        sample_x01(a.rc, rs, tmid, p, x01);
        // ---- grid features: gridencoder.cu:94-201 as k_grid_forward states it; four levels' rows requested before the first blend ----
        const bool oob = (x01[0] < 0.0f || x01[0] > 1.0f) || (x01[1] < 0.0f || x01[1] > 1.0f) || (x01[2] < 0.0f || x01[2] > 1.0f);
        for (uint32_t l0 = 0; l0 < L; l0 += 4u) {
            float pos[4][3], v[4][8][2];
#pragma unroll
            for (uint32_t gi = 0; gi < 4u; ++gi) {
                const uint32_t l = umin(l0 + gi, L - 1u);                   // (a group's tail repeats the last level: fetched, not stored)
                const uint32_t res = a.g.res[l], size = a.g.size[l], mode = a.g.mode[l];
                const TT *tab = table + (size_t)a.g.off[l] * 2u;
                float deriv[3];
                uint32_t cell[3];
                grid_locate<3>(x01, res, a.g.align_corners != 0, a.g.interp, pos[gi], deriv, cell);
#pragma unroll
                for (uint32_t idx = 0; idx < 8u; ++idx) {
                    uint32_t q[3];
#pragma unroll
                    for (uint32_t d = 0; d < 3u; ++d) q[d] = (idx & (1u << d)) ? umin(cell[d] + 1u, res - 1u) : cell[d];
                    load_row<TT, 2>(tab + (size_t)grid_row<3>(q, res, size, mode) * 2u, v[gi][idx]);
                }
            }
#pragma unroll
            for (uint32_t gi = 0; gi < 4u; ++gi) {
                float acc[2] = {0.0f, 0.0f};
#pragma unroll
                for (uint32_t idx = 0; idx < 8u; ++idx) {
                    float w = 1.0f;
#pragma unroll
                    for (uint32_t d = 0; d < 3u; ++d) w *= (idx & (1u << d)) ? pos[gi][d] : 1.0f - pos[gi][d];
                    acc[0] = __builtin_fmaf(w, v[gi][idx][0], acc[0]);
                    acc[1] = __builtin_fmaf(w, v[gi][idx][1], acc[1]);
                }
                if (l0 + gi < L) {
                    bufA[(2u * (l0 + gi)) * 256u] = oob ? 0.0f : acc[0];
                    bufA[(2u * (l0 + gi) + 1u) * 256u] = oob ? 0.0f : acc[1];
                }
            }
        }
        // ---- grid_mlp (network.py:146-153): the lane's own LDS columns, in -> out ping-pong ----
        float *xin = bufA, *xout = bufB;
        for (uint32_t i = 0; i < s.ng; ++i) {
            dense_any(s.wg[i], s.dg[i], s.dg[i + 1u], i + 1u < s.ng, xin, xout);
            float *t = xin; xin = xout; xout = t;
        }
        const float sigma = expf_det(xin[0]);                // network.py:151
        const float delta = rb_next - rb_prev;
        float ds = delta * sigma;
        if (a.rc.last_opaque && j == T - 1u) ds = __builtin_inff();
        const float alpha = 1.0f - expf_det(-ds);
        const float tr = expf_det(-(float)cum);
        float w = alpha * tr;
        if (w != w) w = 0.0f;
        cum += (double)ds;
        wsum += (double)w;
        dep = __builtin_fmaf(w, tmid, dep);
        for (uint32_t c = 0; c < GEO; ++c) accf[c * 256u] = __builtin_fmaf(w, xin[(1u + c) * 256u], accf[c * 256u]);
        if (a.w_out) a.w_out[(size_t)j * Npad + r] = w;
        if (ok) {
            if (a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1) + j + 1] = bnext;
            if (a.dbg_sigma) a.dbg_sigma[(size_t)n * T + j] = sigma;
            if (a.dbg_w) a.dbg_w[(size_t)n * T + j] = w;
            if (a.dbg_xyz) { float *q = a.dbg_xyz + ((size_t)n * T + j) * 3; q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
            if (a.dbg_geo) { float *q = a.dbg_geo + ((size_t)n * T + j) * GEO; for (uint32_t c = 0; c < GEO; ++c) q[c] = xin[(1u + c) * 256u]; }
        }
        if (a.stop_cum > 0.0f && __all((float)cum > a.stop_cum)) break;
        rb_prev = rb_next;
    }
    // ---- per-ray colour head: view_mlp([f_geo | SH(d) * sum w]) -> sigmoid -> + (1 - sum w) * bg (renderer.py:340-357) ----
    const float ws = (float)wsum;
    {
        float sh[16];
        sh_degree4(dirn[0], dirn[1], dirn[2], sh);
        for (uint32_t c = 0; c < GEO; ++c) bufA[c * 256u] = accf[c * 256u];
#pragma unroll
        for (uint32_t c = 0; c < 16u; ++c) bufA[(GEO + c) * 256u] = sh[c] * ws;
    }
    if (ok && a.dbg_fimg) for (uint32_t c = 0; c < NCOL; ++c) a.dbg_fimg[(size_t)n * a.fimg_stride + c] = bufA[c * 256u];
    float *xin = bufA, *xout = bufB;
    for (uint32_t i = 0; i < s.nv; ++i) {
        dense_any(s.wv[i], s.dv[i], s.dv[i + 1u], i + 1u < s.nv, xin, xout);
        float *t = xin; xin = xout; xout = t;
    }
    if (ok) {
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c) {
            const float sg = 1.0f / (1.0f + expf_det(-xin[c * 256u]));
            const float bgm = (1.0f - ws) * a.rc.bg;
            a.image[(size_t)n * a.istride + c] = sg + bgm;
        }
        a.depth[(size_t)n * a.sstride] = dep;
        a.wsum[(size_t)n * a.sstride] = ws;
    }
}

#ifdef SN_EXPERIMENTS
#include "render_experiments.inc"
#endif

// ------------------------------------------------------------------------------------------
// final stage with per-ray termination and wave-level compaction of live samples (opt-in: cfg->compact_live)
// ------------------------------------------------------------------------------------------
// k_final_stage walks every ray of a wave through all T samples in lock-step; a gather instruction, an MFMA or a vector
// instruction costs the same with 1 or 64 lanes active, so a ray that is already opaque (early_stop_eps), a ray that
// misses the aabb (renderer.py:133-135) or a lane beyond the image edge only saves time once the WHOLE wave is dead.
// Here a lane is no longer tied to a ray during the sample evaluation:
//   * home lane r keeps ray r's compositing state (optical depth, colour features, next unassigned sample);
//   * every iteration the wave ballots the rays that still want samples, ranks them with mbcnt (prefix count of the
//     ballot) and deals its 64 evaluation slots out to them round-robin: slot s evaluates sample j_r + s / L of the
//     (s mod L)-th live ray (ds_permute builds the compacted list, ds_bpermute hands out ray ids and cursors), up to
//     CMP_KMAX consecutive samples per ray and iteration -> the 64 rows of every MFMA tile and every gather instruction
//     are filled with live samples only;
//   * slots write (delta*sigma, geometry features, t_mid) to LDS, home lanes composite their own samples in ascending
//     order exactly like k_final_stage (fp64 optical depth, one fmaf chain per channel) and decide termination.
// The assignment of iteration i+1 is made before iteration i has been composited (its first gather group is in flight
// across the matrix-core phase, as in k_final_stage), so a ray that dies in iteration i still occupies its slots of
// iteration i+1; their results are dropped.  With nothing to skip (every ray live for all T samples) slot s IS ray s and
// the outputs are bit-identical to k_final_stage.  Per-ray state of the slots (origin, direction, spacing) sits in the 4
// padding dwords of the wave's two feature-slab images, the per-sample records reuse the slab rows: no extra LDS.
constexpr uint32_t CMP_KMAX = 4;

template <typename TT, int K>
__global__ __launch_bounds__(256, 2) void k_final_stage_cmp(FinalArgs a) {
    SN_POISON_ALL();
    constexpr int L = 16, GEO = 15, NSH = 16, NCOL = GEO + NSH, VH = 32, PG = 4, IN = 32;
    constexpr int VW0 = VH * PadIn<NCOL>::value, VW1 = VH * PadIn<VH>::value;
    constexpr int WAVE_SLAB = 2 * 64 * SLAB_STRIDE;
    static_assert(K >= PG && K <= 8, "FinalLv instantiation");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (uint32_t i = threadIdx.x; i < (uint32_t)PACK16_U4; i += 256u)
        reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.mlp_pack)[i];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    float *wave_base = lds + PACK_FLOATS + wave * WAVE_SLAB;
    uint32_t *slab_hi = reinterpret_cast<uint32_t *>(wave_base), *slab_lo = slab_hi + 64 * SLAB_STRIDE;
    float *fe = wave_base + lane;
    __syncthreads();
    clock_probe(0);

    uint32_t n;
    const uint32_t wg = tile_id(a.rc);
    const bool ok = ray_of_lane(a.rc, wg, n);
    const uint32_t col0 = wg * 256u + wave * 64u;                 // scratch column of the wave's lane 0
    const uint32_t Npad = a.rc.Npad, T = a.T;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    bool alive = ok;
    if (ray_misses(a.rc, rs)) alive = false;              // renderer.py:133-135; it is skipped here (weights 0)
    // ray table in the slab's padding columns (dwords 16..19 of each 20-dword row)
    *reinterpret_cast<float4 *>(slab_hi + lane * SLAB_STRIDE + 16) = make_float4(rs.o[0], rs.o[1], rs.o[2], rs.s_near);
    *reinterpret_cast<float4 *>(slab_lo + lane * SLAB_STRIDE + 16) = make_float4(rs.d[0], rs.d[1], rs.d[2], rs.s_far);
    __builtin_amdgcn_wave_barrier();
    const float b0step = 1.0f / (float)T;
    auto bin_col = [&](uint32_t j, uint32_t col) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + col];
        if (a.bins0_tab) return a.bins0_tab[j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };

    // ---- home-lane state ----
    float fimg[GEO];
#pragma unroll
    for (int c = 0; c < GEO; ++c) fimg[c] = 0.0f;
    float dep = 0.0f;
    double cum = 0.0, wsum = 0.0;
    uint32_t j_cur = 0;                                           // next sample of this ray not yet assigned to a slot

    // ---- assignment of the wave's 64 slots ----
    struct Assign { uint32_t L, rank, n; };                       // live rays (uniform); this ray's rank and sample count
    uint32_t s_ray = lane, s_j = 0;                               // slot view: ray (lane index inside the wave) and sample
    auto assign = [&](Assign &as) {
        const bool want = alive && j_cur < T;
        const uint64_t M = __ballot(want);
        const uint32_t Lc = (uint32_t)__popcll(M);
        as.L = Lc; as.rank = 0; as.n = 0;
        if (Lc == 0u) { s_ray = lane; s_j = T - 1u; return; }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
        const uint32_t dst = want ? rank : Lc + (lane - rank);    // a permutation: live rays first, in lane order
        const uint32_t list = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)lane);
        const uint32_t inv = (uint32_t)(65536.0f / (float)Lc) + 1u;   // (x * inv) >> 16 == x / Lc for x < 64, Lc <= 64
        const uint32_t q = (lane * inv) >> 16, rnk = lane - q * Lc;
        s_ray = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rnk << 2), (int)list);
        const uint32_t jr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_ray << 2), (int)j_cur);
        const uint32_t sj = jr + q;
        s_j = (q < CMP_KMAX && sj < T) ? sj : T - 1u;             // surplus slots redo an in-range sample; nobody reads them
        uint32_t cnt = (((63u - rank) * inv) >> 16) + 1u;         // slots rank, rank + L, ... below 64
        cnt = umin(umin(cnt, CMP_KMAX), T - umin(j_cur, T));
        as.rank = rank; as.n = want ? cnt : 0u;
        j_cur += as.n;
    };
    float tmid_n, delta_n, p_n[3], x01_n[3];
    bool last_n;
    auto slot_geometry = [&]() {
        const float4 A = *reinterpret_cast<const float4 *>(slab_hi + s_ray * SLAB_STRIDE + 16);
        const float4 B = *reinterpret_cast<const float4 *>(slab_lo + s_ray * SLAB_STRIDE + 16);
        RaySetup q;
        q.o[0] = A.x; q.o[1] = A.y; q.o[2] = A.z; q.s_near = A.w;
        q.d[0] = B.x; q.d[1] = B.y; q.d[2] = B.z; q.s_far = B.w;
        const uint32_t col = col0 + s_ray;
        const float rb_prev = real_bin(q, bin_col(s_j, col)), rb_next = real_bin(q, bin_col(s_j + 1u, col));
        tmid_n = (rb_next + rb_prev) / 2.0f;
        delta_n = rb_next - rb_prev;
        last_n = s_j == T - 1u;
        sample_x01(a.rc, q, tmid_n, p_n, x01_n);
    };

    Assign cur, nxt;
    GroupRegs<TT, 2, PG> g0;
    assign(cur);
    slot_geometry();
    bool fast_n = all_interior(a.lv, x01_n);
    issue_group_lv<TT, PG, K, 0, false>(a.lv, x01_n, g0);
    __builtin_amdgcn_sched_barrier(0);
    while (cur.L != 0u) {
        const float tmid = tmid_n, delta = delta_n;
        const bool last = last_n;
        const float x01[3] = {x01_n[0], x01_n[1], x01_n[2]};
        float h[16];
        uint32_t *row_hi = slab_hi + lane * SLAB_STRIDE, *row_lo = slab_lo + lane * SLAB_STRIDE;
        auto emit = [&](int l, const float (&acc)[2]) {
            uint32_t ph, pl;
            split2(acc[0], acc[1], ph, pl);
            row_hi[l] = ph;
            row_lo[l] = pl;
        };
        blend_group<TT, 2, PG, K, 0>(g0, emit);
        __builtin_amdgcn_sched_barrier(0);
        auto rest = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            static_for<1, L / PG>([&](auto gg) {
                constexpr int GRP = decltype(gg)::value;
                GroupRegs<TT, 2, PG> gr;
                issue_group_lv<TT, PG, K, GRP, FAST>(a.lv, x01, gr);
                __builtin_amdgcn_sched_barrier(0);
                blend_group<TT, 2, PG, K, GRP>(gr, emit);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (__builtin_expect(fast_n, 1)) rest(std::true_type{});
        else {
            rest(std::false_type{});
            const bool oob = (x01[0] < 0.0f || x01[0] > 1.0f) || (x01[1] < 0.0f || x01[1] > 1.0f) || (x01[2] < 0.0f || x01[2] > 1.0f);
            if (__builtin_expect(__any(oob), 0)) {
                if (oob) for (int l = 0; l < L; ++l) { row_hi[l] = 0u; row_lo[l] = 0u; }
            }
        }
        // next assignment (from the termination state of one iteration ago) + its first gather group
        assign(nxt);
        slot_geometry();
        fast_n = all_interior(a.lv, x01_n);
        issue_group_lv<TT, PG, K, 0, false>(a.lv, x01_n, g0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_wave_barrier();
        grid_mlp_mfma16(reinterpret_cast<const uint4 *>(lds) + opaque_zero(), slab_hi, slab_lo, h);
        __builtin_amdgcn_wave_barrier();
        {   // per-sample record of this slot: [delta*sigma | 15 geometry features] in its hi row, t_mid in its lo row
            const float sigma = expf_det(h[0]);
            float ds = delta * sigma;
            if (a.rc.last_opaque && last) ds = __builtin_inff();
            float4 *rec = reinterpret_cast<float4 *>(row_hi);
            rec[0] = make_float4(ds, h[1], h[2], h[3]);
            rec[1] = make_float4(h[4], h[5], h[6], h[7]);
            rec[2] = make_float4(h[8], h[9], h[10], h[11]);
            rec[3] = make_float4(h[12], h[13], h[14], h[15]);
            reinterpret_cast<float *>(row_lo)[0] = tmid;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- compositing by the home lanes, samples in ascending order (renderer.py:308-338) ----
#pragma unroll 1
        for (uint32_t k = 0; k < CMP_KMAX; ++k) {
            const bool mine = k < cur.n && alive;
            if (!__any(mine)) break;
            if (mine) {
                const uint32_t slot = cur.rank + k * cur.L;
                const float4 *rec = reinterpret_cast<const float4 *>(slab_hi + slot * SLAB_STRIDE);
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                const float tm = reinterpret_cast<const float *>(slab_lo + slot * SLAB_STRIDE)[0];
                const float g[GEO] = {r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
                const float ds = r0.x;
                const float alpha = 1.0f - expf_det(-ds);
                const float tr = expf_det(-(float)cum);
                float w = alpha * tr;
                if (w != w) w = 0.0f;
                cum += (double)ds;
                wsum += (double)w;
                dep = __builtin_fmaf(w, tm, dep);
#pragma unroll
                for (int c = 0; c < GEO; ++c) fimg[c] = __builtin_fmaf(w, g[c], fimg[c]);
                if (a.stop_cum > 0.0f && (float)cum > a.stop_cum) alive = false;   // remaining weights sum to < eps
            }
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
    }
    clock_probe(1);

    // ---- per-ray colour head (as in k_final_stage) ----
    __syncthreads();
    stage_weights<NCOL, VH>(lds, a.vw[0]);
    stage_weights<VH, VH>(lds + VW0, a.vw[1]);
    stage_weights<VH, 3>(lds + VW0 + VW1, a.vw[2]);
    __syncthreads();
    float dirn[3] = {rs.d[0], rs.d[1], rs.d[2]};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float aa = dirn[0] * dirn[0], bb = dirn[1] * dirn[1], cc = dirn[2] * dirn[2];
        const float nrm = sqrtf((aa + bb) + cc);
        dirn[0] = dirn[0] / nrm; dirn[1] = dirn[1] / nrm; dirn[2] = dirn[2] / nrm;
    }
    const float ws = (float)wsum;
    float sh[NSH];
    sh_degree4(dirn[0], dirn[1], dirn[2], sh);
#pragma unroll
    for (int c = 0; c < GEO; ++c) fe[c * 64] = fimg[c];
#pragma unroll
    for (int c = 0; c < NSH; ++c) fe[(GEO + c) * 64] = sh[c] * ws;
    static_assert(IN * 64 <= 2 * 64 * SLAB_STRIDE, "view MLP activations reuse the slab");
    dense_ldsw_col<NCOL, VH, 1>(lds, fe, fe, 64u);
    dense_ldsw_col<VH, VH, 1>(lds + VW0, fe, fe, 64u);
    float rgb[3];
    {
        float v2[VH];
#pragma unroll
        for (int k = 0; k < VH; ++k) v2[k] = fe[k * 64];
        dense_ldsw<VH, 3, 0>(lds + VW0 + VW1, v2, rgb);
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sg = 1.0f / (1.0f + expf_det(-rgb[c]));
            const float bgm = (1.0f - ws) * a.rc.bg;
            a.image[(size_t)n * a.istride + c] = sg + bgm;
        }
        a.depth[(size_t)n * a.sstride] = dep;
        a.wsum[(size_t)n * a.sstride] = ws;
        if (a.dbg_fimg) {
#pragma unroll
            for (int c = 0; c < GEO; ++c) a.dbg_fimg[(size_t)n * a.fimg_stride + c] = fimg[c];
#pragma unroll
            for (int c = 0; c < NSH; ++c) a.dbg_fimg[(size_t)n * a.fimg_stride + GEO + c] = sh[c] * ws;
        }
        if (a.head_rgbd) {
            float *q = a.head_rgbd + (size_t)n * a.fimg_stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = a.image[(size_t)n * a.istride + c];
            q[3] = dep;
        }
    }
}

// ------------------------------------------------------------------------------------------
// final stage, sample-parallel variant for small linear-order batches (the counterpart of k_prop_stage_sp)
// ------------------------------------------------------------------------------------------
// 4096 rays are 16 workgroups of k_final_stage, each wave alone on its SIMD for T serial samples of ~19 us.  Here
// lpr = 2^lpr_log2 lanes share a ray and evaluate spl consecutive samples each (lpr * spl >= T) with the same
// pipelined gather / split-f16 MFMA code; per-sample results (delta*sigma, 15 geometry features, t_mid) are parked
// in LDS.  The order-sensitive parts then run exactly as in the one-lane-per-ray kernel: the ray's first lane forms
// the fp64 optical-depth prefix, every lane turns its own samples into weights, and each output channel is one
// sample-ascending fmaf chain (channels dealt out to the ray's lanes).  The view MLP runs redundantly on all lanes of
// a ray; the first one stores.  Bit-identical to k_final_stage<MLP_F16X3>.  No early termination here.
constexpr int FSP_ROW = 65;                  // floats per (sample slot, channel) row: 64 lanes + 1 (bank spread)
constexpr int FSP_CH = 16;                   // 15 geometry features + t_mid
constexpr int FSP_DEPTH_ROW = 31;            // activation row that carries the depth (view MLP reads rows 0..30)
constexpr uint32_t FSP_MAX_SPL = 4;
static size_t final_sp_lds_floats(uint32_t spl) {
    return (size_t)PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE + 4 * 512 + 4 * (size_t)spl * FSP_CH * FSP_ROW;
}

template <typename TT, int K>
__global__ __launch_bounds__(256, 1) void k_final_stage_sp(FinalArgs a, uint32_t lpr_log2, uint32_t spl_log2) {
    SN_POISON_ALL();
    constexpr int L = 16, GEO = 15, NSH = 16, NCOL = GEO + NSH, VH = 32, PG = 4;
    constexpr int VW0 = VH * PadIn<NCOL>::value, VW1 = VH * PadIn<VH>::value;
    constexpr int WAVE_SLAB = 2 * 64 * SLAB_STRIDE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (uint32_t i = threadIdx.x; i < (uint32_t)PACK16_U4; i += 256u)
        reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.mlp_pack)[i];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, spl = 1u << spl_log2;
    float *wave_base = lds + PACK_FLOATS + wave * WAVE_SLAB;
    uint32_t *slab_hi = reinterpret_cast<uint32_t *>(wave_base), *slab_lo = slab_hi + 64 * SLAB_STRIDE;
    float *fe = wave_base + lane;                           // activation column of this lane (after the march)
    float *l_w = lds + PACK_FLOATS + 4 * WAVE_SLAB + wave * 512u;     // [ray][sample]: delta*sigma, then the weight
    float *l_cum = l_w + 256;                                         // [ray][sample]: (float) fp64 prefix of delta*sigma
    float *hs = lds + PACK_FLOATS + 4 * WAVE_SLAB + 4 * 512 + wave * (spl * FSP_CH * FSP_ROW);   // [slot][channel][lane]
    __syncthreads();

    const uint32_t lpr = 1u << lpr_log2, c = lane & (lpr - 1u), rw = lane >> lpr_log2, rpw = 64u >> lpr_log2;
    const uint32_t r = (blockIdx.x * 4u + wave) * rpw + rw;           // scratch column = ray index; the grid covers Npad
    const bool ok = r < a.rc.N;
    const uint32_t n = ok ? r : 0u;
    const uint32_t Npad = a.rc.Npad, T = a.T;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    const float b0step = 1.0f / (float)T;
    auto bin_at = [&](uint32_t j) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + r];
        if (a.bins0_tab) return a.bins0_tab[(size_t)n * a.bins0_stride + j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };
    float dirn[3] = {rs.d[0], rs.d[1], rs.d[2]};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float aa = dirn[0] * dirn[0], bb = dirn[1] * dirn[1], cc = dirn[2] * dirn[2];
        const float nrm = sqrtf((aa + bb) + cc);
        dirn[0] = dirn[0] / nrm; dirn[1] = dirn[1] / nrm; dirn[2] = dirn[2] / nrm;
    }
    const TT *table = reinterpret_cast<const TT *>(a.table);
    const uint32_t jbase = c << spl_log2, rbase = rw << (lpr_log2 + spl_log2);   // first sample of this lane; ray's LDS row
    if (ok && a.dbg_bins && c == 0u) a.dbg_bins[(size_t)n * (T + 1)] = bin_at(0);

    // ---- march over this lane's samples; slots past T redo sample T-1 and store nothing ----
    GroupRegs<TT, 2, PG> g0;
    uint32_t j_n = jbase < T ? jbase : T - 1u;
    float rb_prev_n = real_bin(rs, bin_at(j_n));
    float bnext_n = bin_at(j_n + 1u);
    float rb_next_n = real_bin(rs, bnext_n);
    float tmid_n = (rb_next_n + rb_prev_n) / 2.0f;
    float p_n[3], x01_n[3];
    sample_x01(a.rc, rs, tmid_n, p_n, x01_n);
    issue_group<TT, 2, PG, K, 0>(table, a.g, x01_n, g0, a.pairs);
    __builtin_amdgcn_sched_barrier(0);
    for (uint32_t i = 0; i < spl; ++i) {
        const uint32_t jr = jbase + i, j = j_n;
        const float bnext = bnext_n, rb_next = rb_next_n, rb_prev = rb_prev_n, tmid = tmid_n;
        const float p[3] = {p_n[0], p_n[1], p_n[2]}, x01[3] = {x01_n[0], x01_n[1], x01_n[2]};
        float h[16];
        uint32_t *row_hi = slab_hi + lane * SLAB_STRIDE, *row_lo = slab_lo + lane * SLAB_STRIDE;
        auto emit = [&](int l, const float (&acc)[2]) {
            uint32_t ph, pl;
            split2(acc[0], acc[1], ph, pl);
            row_hi[l] = ph;
            row_lo[l] = pl;
        };
        blend_group<TT, 2, PG, K, 0>(g0, emit);
        __builtin_amdgcn_sched_barrier(0);
        static_for<1, L / PG>([&](auto gg) {
            constexpr int GRP = decltype(gg)::value;
            GroupRegs<TT, 2, PG> gr;
            issue_group<TT, 2, PG, K, GRP>(table, a.g, x01, gr, a.pairs);
            __builtin_amdgcn_sched_barrier(0);
            blend_group<TT, 2, PG, K, GRP>(gr, emit);
            __builtin_amdgcn_sched_barrier(0);
        });
        {
            const bool oob = (x01[0] < 0.0f || x01[0] > 1.0f) || (x01[1] < 0.0f || x01[1] > 1.0f) || (x01[2] < 0.0f || x01[2] > 1.0f);
            if (__builtin_expect(__any(oob), 0)) {
                if (oob) for (int l = 0; l < L; ++l) { row_hi[l] = 0u; row_lo[l] = 0u; }
            }
        }
        {   // geometry and first gather group of this lane's next sample (the last slot re-issues its own)
            const uint32_t jq = jr + 1u;
            j_n = (i + 1u < spl && jq < T) ? jq : j;
            rb_prev_n = real_bin(rs, bin_at(j_n));
            bnext_n = bin_at(j_n + 1u);
            rb_next_n = real_bin(rs, bnext_n);
            tmid_n = (rb_next_n + rb_prev_n) / 2.0f;
            sample_x01(a.rc, rs, tmid_n, p_n, x01_n);
            issue_group<TT, 2, PG, K, 0>(table, a.g, x01_n, g0, a.pairs);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();
        grid_mlp_mfma16(reinterpret_cast<const uint4 *>(lds) + opaque_zero(), slab_hi, slab_lo, h);
        __builtin_amdgcn_wave_barrier();
        const float sigma = expf_det(h[0]);
        float ds = (rb_next - rb_prev) * sigma;
        if (a.rc.last_opaque && j == T - 1u) ds = __builtin_inff();
        if (jr < T) {
            l_w[rbase + jr] = ds;
            float *rec = hs + (size_t)(i * FSP_CH) * FSP_ROW + lane;
#pragma unroll
            for (int ch = 0; ch < GEO; ++ch) rec[ch * FSP_ROW] = h[1 + ch];
            rec[GEO * FSP_ROW] = tmid;
            if (ok) {
                if (a.dbg_bins) a.dbg_bins[(size_t)n * (T + 1) + jr + 1] = bnext;
                if (a.dbg_sigma) a.dbg_sigma[(size_t)n * T + jr] = sigma;
                if (a.dbg_xyz) { float *q = a.dbg_xyz + ((size_t)n * T + jr) * 3; q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
                if (a.dbg_geo) { float *q = a.dbg_geo + ((size_t)n * T + jr) * GEO;
#pragma unroll
                    for (int ch = 0; ch < GEO; ++ch) q[ch] = h[1 + ch]; }
            }
        }
    }
    __syncthreads();                          // per-sample records visible; the packed MLP weights are dead
    stage_weights<NCOL, VH>(lds, a.vw[0]);
    stage_weights<VH, VH>(lds + VW0, a.vw[1]);
    stage_weights<VH, 3>(lds + VW0 + VW1, a.vw[2]);
    if (c == 0u) {                            // optical depth before each sample: fp64 running sum, rounded per prefix
        double cum = 0.0;
        for (uint32_t j = 0; j < T; ++j) { l_cum[rbase + j] = (float)cum; cum += (double)l_w[rbase + j]; }
    }
    __syncthreads();
    for (uint32_t i = 0; i < spl; ++i) {      // weights of this lane's samples (renderer.py:308-325)
        const uint32_t j = jbase + i;
        if (j < T) {
            const float alpha = 1.0f - expf_det(-l_w[rbase + j]);
            const float tr = expf_det(-l_cum[rbase + j]);
            float w = alpha * tr;
            if (w != w) w = 0.0f;
            l_w[rbase + j] = w;
            if (a.w_out) a.w_out[(size_t)j * Npad + r] = w;
            if (ok && a.dbg_w) a.dbg_w[(size_t)n * T + j] = w;
        }
    }
    __syncthreads();
    // ---- compositing: one sample-ascending fmaf chain per channel, channels dealt out to the ray's lanes ----
    double wsum = 0.0;
    for (uint32_t j = 0; j < T; ++j) wsum += (double)l_w[rbase + j];
    float *ray_col = wave_base + (rw << lpr_log2);           // activation column of the ray's first lane (slab is dead)
    for (uint32_t ch = c; ch < (uint32_t)FSP_CH; ch += lpr) {
        float acc = 0.0f;
        for (uint32_t j = 0; j < T; ++j) {
            const uint32_t cj = j >> spl_log2, ij = j & (spl - 1u);
            acc = __builtin_fmaf(l_w[rbase + j], hs[(size_t)(ij * FSP_CH + ch) * FSP_ROW + (rw << lpr_log2) + cj], acc);
        }
        ray_col[(ch < (uint32_t)GEO ? ch : (uint32_t)FSP_DEPTH_ROW) * 64u] = acc;
    }
    const float ws = (float)wsum;
    if (c == 0u) {
        float sh[NSH];
        sh_degree4(dirn[0], dirn[1], dirn[2], sh);
#pragma unroll
        for (int k = 0; k < NSH; ++k) ray_col[(GEO + k) * 64] = sh[k] * ws;
    }
    __builtin_amdgcn_wave_barrier();          // the ray's lanes live in one wave; LDS serves a wave in order
    const float dep = ray_col[FSP_DEPTH_ROW * 64];
    if (ok && c == 0u && a.dbg_fimg) {
        for (int k = 0; k < NCOL; ++k) a.dbg_fimg[(size_t)n * a.fimg_stride + k] = ray_col[k * 64];
    }
    __builtin_amdgcn_wave_barrier();
    float rgb[3];
    dense_ldsw_col<NCOL, VH, 1>(lds, ray_col, fe, 64u);       // every lane of the ray: same inputs, own output column
    dense_ldsw_col<VH, VH, 1>(lds + VW0, fe, fe, 64u);
    {
        float v2[VH];
#pragma unroll
        for (int k = 0; k < VH; ++k) v2[k] = fe[k * 64];
        dense_ldsw<VH, 3, 0>(lds + VW0 + VW1, v2, rgb);
    }
    if (ok && c == 0u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sg = 1.0f / (1.0f + expf_det(-rgb[k]));
            const float bgm = (1.0f - ws) * a.rc.bg;
            a.image[(size_t)n * a.istride + k] = sg + bgm;
        }
        a.depth[(size_t)n * a.sstride] = dep;
        a.wsum[(size_t)n * a.sstride] = ws;
        if (a.head_rgbd) {
            float *q = a.head_rgbd + (size_t)n * a.fimg_stride;
#pragma unroll
            for (int k = 0; k < 3; ++k) q[k] = a.image[(size_t)n * a.istride + k];
            q[3] = dep;
        }
    }
}

// ------------------------------------------------------------------------------------------
// feature stage: f[n, :] = sum_j w[n,j] * feat_grid(xyz[n,j])   (renderer.py:301-302 + 361, the SAM head's f_sam)
// ------------------------------------------------------------------------------------------
// Runs after the final stage on the same lane -> ray mapping.  Sample positions are recomputed from the last
// stage's bins (scratch, coalesced) with the very same code the final stage uses, weights come from the final
// stage's scratch column; nothing per-sample is written.  A workgroup pass covers LG levels (blockIdx.y).
struct FeatArgs {
    RayCommon rc;
    GridLevels g;
    const void *table;
    uint32_t T;
    const float *bins_in;        // scratch [T+1][Npad] or NULL (single-stage)
    const float *bins0_tab;
    uint32_t bins0_stride;
    const float *w_in;           // scratch [T][Npad]
    float *out;                  // [N, L*C], rows out_stride floats apart
    uint32_t out_stride;         // L*C, or sn_render_io.head_stride
    uint32_t level0;             // first level of this launch (blockIdx.y counts passes of LG levels from here)
};

// wave-wide minimum / maximum of a float in six DPP steps (row_shr 1, 2, 4, 8 bring a row's extremum to its lane 15, row_bcast:15 / :31 carry
// it across the four rows to lane 63), returned wave-uniform.
template <bool MAXI>
__device__ __forceinline__ float wave_extremum(float v) {
    auto step = [&](auto ctrl_tag, auto rmask_tag) {
        constexpr int ctrl = decltype(ctrl_tag)::value, rmask = decltype(rmask_tag)::value;
        const float t = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), ctrl, rmask, 0xf, false));
        v = MAXI ? fmaxf(v, t) : fminf(v, t);
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

// PATCH (round 6; north_star "LDS staging of per-tile grid voxels"): on a DENSE level the 64 rays of a wave tile at one sample
// index occupy a handful of cells -- the bounding box of their vertices is 10-17 rows on levels 0-6 of the heads' grid
// (profiles/r06/gather_lines_per_level.txt) -- yet every lane issues its own 8 corner fetches (16 gather instructions per level with 32-byte
// rows, each at the texture path's per-instruction floor).  Here the wave fetches the box ONCE (lane v loads vertex v: one or two gather
// instructions), parks it in LDS and every lane reads its 8 corners from there.  Same rows, same blend: bit-identical.  A box beyond 64 vertices
// (wave-uniform test) and every hashed level take the ordinary path: on hashed levels a box fetches MORE lines than the direct gathers (same file).
// MEASURED SLOWER (profiles/r06/feat_patch_ab.json, same box): configs[2] 400x400 2.46 -> 2.64 ms (fp32 tables), 2.08 -> 2.22 (fp16); 800x800 6.81 -> 7.43 /
// 5.79 -> 6.32; one level per pass: 2.47 -> 2.53.  The dense levels' direct gathers touch 1.3-2.2 lines per instruction -- the texture path's best
// case -- while the staged form adds a box computation (six wave reductions per sample), an LDS write -> read round trip inside every level's
// dependent chain and twice the registers (192 vs 92: 2 instead of 5 waves per SIMD).  Third measured negative of LDS voxel staging on this chip
// (round 2: per-wave vertex cache in the last stage, round 3: LDS-resident level 0).  Opt-in: sn_render_tuning.feat_patch = 1.
constexpr uint32_t FEAT_PATCH_ROWS = 64;

template <typename TT, int C, int LG, bool PATCH = false>
__global__ __launch_bounds__(256) void k_feat_stage(FeatArgs a) {
    SN_POISON_ALL();
    __shared__ __attribute__((aligned(16))) TT patch_lds[PATCH ? 4 * LG * FEAT_PATCH_ROWS * C : 1];
    uint32_t n;
    const uint32_t wg = tile_id_x(a.rc, blockIdx.x, gridDim.x);
    const bool ok = ray_of_lane(a.rc, wg, n);
    const uint32_t r = wg * 256u + threadIdx.x;
    const uint32_t Npad = a.rc.Npad;
    RaySetup rs;
    setup_ray(a.rc, n, rs);
    const uint32_t T = a.T;
    const float b0step = 1.0f / (float)T;
    auto bin_at = [&](uint32_t j) -> float {
        if (a.bins_in) return a.bins_in[(size_t)j * Npad + r];
        if (a.bins0_tab) return a.bins0_tab[(size_t)n * a.bins0_stride + j];
        return linspace_at(0.0f, 1.0f, b0step, T + 1u, j);
    };
    const uint32_t l0 = a.level0 + blockIdx.y * LG;
    const TT *table = reinterpret_cast<const TT *>(a.table);
    const uint32_t lane = threadIdx.x & 63u;
    TT *my_patch = patch_lds + (PATCH ? (threadIdx.x >> 6) * (LG * FEAT_PATCH_ROWS * C) : 0u);
    float acc[LG][C];
#pragma unroll
    for (int i = 0; i < LG; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[i][c] = 0.0f;
    float rb_prev = real_bin(rs, bin_at(0));
    for (uint32_t j = 0; j < T; ++j) {
        const float rb_next = real_bin(rs, bin_at(j + 1u));
        const float tmid = (rb_next + rb_prev) / 2.0f;
        rb_prev = rb_next;
        const float w = a.w_in[(size_t)j * Npad + r];
        // a sample index at which all 64 rays of the wave carry weight exactly 0 (behind an opaque surface, in empty space) adds
        // fmaf(0, feature, acc) = acc: no position, no gathers.  Bit-identical (table values are finite).
        if (__all(w == 0.0f)) continue;
        float p[3], x01[3];
        sample_x01(a.rc, rs, tmid, p, x01);
        bool oob = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) if (x01[d] < 0.0f || x01[d] > 1.0f) oob = true;     // gridencoder.cu:105-130
        const float wz = oob ? 0.0f : w;
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        if constexpr (PATCH) {       // bounding box (in table coordinates) of the lanes whose sample counts; wave-uniform
            const bool act = wz != 0.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = wave_extremum<false>(act ? x01[d] : __builtin_inff());
                hi[d] = wave_extremum<true>(act ? x01[d] : -__builtin_inff());
            }
            if (!(lo[0] <= hi[0])) continue;              // no lane counts: every fmaf(0, feature, acc) would leave acc as it is
        }
        auto blend = [&](int i, const float (&ps)[3], const float (&cvi)[8][C]) {
            float feat[C];
#pragma unroll
            for (int c = 0; c < C; ++c) feat[c] = 0.0f;
#pragma unroll
            for (uint32_t idx = 0; idx < 8u; ++idx) {       // gridencoder.cu:171-192
                float cw = 1.0f;
#pragma unroll
                for (uint32_t d = 0; d < 3u; ++d) cw *= (idx & (1u << d)) ? ps[d] : 1.0f - ps[d];
#pragma unroll
                for (int c = 0; c < C; ++c) feat[c] = __builtin_fmaf(cw, cvi[idx][c], feat[c]);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) acc[i][c] = __builtin_fmaf(wz, feat[c], acc[i][c]);
        };
        if constexpr (!PATCH) {
            float pos[LG][3];
            float cv[LG][8][C];
#pragma unroll
            for (int i = 0; i < LG; ++i) {
                const uint32_t l = l0 + i;
                uint32_t cell[3], offs[8];
                locate_linear(x01, a.g.res[l], pos[i], cell);
                corner_offsets<-1, (uint32_t)(C * sizeof(TT))>(cell, a.g.res[l], a.g.size[l], a.g.mode[l], offs);
                const char *tab = reinterpret_cast<const char *>(table + (size_t)a.g.off[l] * C);
#pragma unroll
                for (int k = 0; k < 8; ++k) load_row<TT, C>(reinterpret_cast<const TT *>(tab + offs[k]), cv[i][k]);
            }
#pragma unroll
            for (int i = 0; i < LG; ++i) blend(i, pos[i], cv[i]);
        } else {
            // the boxes of all LG levels are fetched first (their gathers fly together), then every level is read back and blended
            uint32_t c0[LG][3], nd[LG][3];
            bool staged[LG];
#pragma unroll
            for (int i = 0; i < LG; ++i) {
                const uint32_t l = l0 + i;
                const uint32_t res = a.g.res[l];
                // vertices of the box: cells of lo .. hi (locate_linear is monotone in x01) plus the +1 corners, clamped like them (gridencoder.cu:182)
                const float rf = (float)res, top = (float)(res - 1u);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float p0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(lo[d], rf, -0.5f), 0.0f, top);
                    const float p1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(hi[d], rf, -0.5f), 0.0f, top);
                    c0[i][d] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p0);
                    const uint32_t c1 = umin((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p1) + 1u, res - 1u);
                    nd[i][d] = c1 - c0[i][d] + 1u;
                }
                const uint32_t nxy = nd[i][0] * nd[i][1], total = nxy * nd[i][2];
                const uint32_t mode = a.g.mode[l];
                staged[i] = (mode & 1u) == 0u && ((mode >> 4) & 15u) == 3u && ((mode >> 1) & 3u) == 0u && total <= FEAT_PATCH_ROWS;    // dense level, box fits
                if (staged[i]) {     // lane v fetches vertex v = ix + nx (iy + ny iz) of the box
                    TT *pl = my_patch + (uint32_t)i * (FEAT_PATCH_ROWS * C);
                    const char *tab = reinterpret_cast<const char *>(table + (size_t)a.g.off[l] * C);
                    const uint32_t iz = (uint32_t)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)nxy));
                    const uint32_t rem = lane - iz * nxy;
                    const uint32_t iy = (uint32_t)(((float)rem + 0.5f) * __builtin_amdgcn_rcpf((float)nd[i][0]));
                    const uint32_t ix = rem - iy * nd[i][0];
                    if (lane < total) {
                        const uint32_t row = (c0[i][0] + ix) + res * ((c0[i][1] + iy) + res * (c0[i][2] + iz));
                        const TT *src = reinterpret_cast<const TT *>(tab + (size_t)row * (C * sizeof(TT)));      // the row as it is stored
                        constexpr int RB = C * (int)sizeof(TT);
                        if constexpr (RB % 16 == 0) {
#pragma unroll
                            for (int q = 0; q < RB / 16; ++q) reinterpret_cast<uint4 *>(pl + lane * C)[q] = reinterpret_cast<const uint4 *>(src)[q];
                        } else {
#pragma unroll
                            for (int c = 0; c < C; ++c) pl[lane * C + c] = src[c];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < LG; ++i) {
                const uint32_t l = l0 + i;
                const uint32_t res = a.g.res[l];
                uint32_t cell[3];
                float ps[3];
                float cvi[8][C];
                locate_linear(x01, res, ps, cell);
                if (staged[i]) {
                    // this lane's corners inside the box (a lane that does not count may lie outside: any row will do, its weight is 0)
                    const TT *pl = my_patch + (uint32_t)i * (FEAT_PATCH_ROWS * C);
                    uint32_t q0[3], q1[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        q0[d] = umin(cell[d] - c0[i][d], nd[i][d] - 1u);
                        q1[d] = umin(q0[d] + (cell[d] < res - 1u ? 1u : 0u), nd[i][d] - 1u);
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 8u; ++k) {
                        const uint32_t v = ((k & 1u) ? q1[0] : q0[0]) + nd[i][0] * (((k & 2u) ? q1[1] : q0[1]) + nd[i][1] * ((k & 4u) ? q1[2] : q0[2]));
                        load_row<TT, C>(pl + v * C, cvi[k]);
                    }
                    blend(i, ps, cvi);
                } else {
                    uint32_t offs[8];
                    corner_offsets<-1, (uint32_t)(C * sizeof(TT))>(cell, res, a.g.size[l], a.g.mode[l], offs);
                    const char *tab = reinterpret_cast<const char *>(table + (size_t)a.g.off[l] * C);
#pragma unroll
                    for (int k = 0; k < 8; ++k) load_row<TT, C>(reinterpret_cast<const TT *>(tab + offs[k]), cvi[k]);
                    blend(i, ps, cvi);
                }
            }
            __builtin_amdgcn_wave_barrier();      // the next sample's box overwrites the patch: every lane has read its corners
        }
    }
    if (!ok) return;
    float *o = a.out + (size_t)n * a.out_stride + (size_t)l0 * C;
    const bool vec_ok = (a.out_stride & 3u) == 0u;          // (a [N, 163] head input: rows are not 16-byte aligned)
#pragma unroll
    for (int i = 0; i < LG; ++i) {
        if (C % 4 == 0 && vec_ok) {
#pragma unroll
            for (int q = 0; q < C / 4; ++q)
                reinterpret_cast<float4 *>(o + i * C)[q] = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) o[i * C + c] = acc[i][c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// ---- optional per-kernel timing (bench.py's roofline leg) ---------------------------------
// When enabled, every kernel launched by sn_rm_render_rays is bracketed by hipEvents recorded
// on the caller's stream; sn_rm_profile_read() synchronises them and returns, per kernel class,
// the launch count and the summed device time.  Off by default (no events, no overhead).
// pair rows (fp32: row, +x neighbour) / quad rows (fp16: row, +x, +y, +x+y) of the dense levels (see PairTab): one
// thread per (level < K, row); 16 bytes per vertex either way
template <typename T>
__global__ void k_pack_pairs(const T *__restrict__ table, GridLevels g, PairTab pt, uint32_t K) {
    SN_POISON_ALL();
    const uint32_t l = blockIdx.y;
    if (l >= K) return;
    const uint32_t res = g.res[l], rows = res * res * res;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    // vertex (x, y, z) -> its row through the level's OWN index function (gridencoder.cu:45-79): for a dense level that is i itself;
    // a hashed level that is "densified" for this render (densify_levels below) is looked up through the hash, once per vertex, here
    const uint32_t x = i % res, y = (i / res) % res, z = i / (res * res);
    const uint32_t x1 = x + 1u < res ? x + 1u : x, y1 = y + 1u < res ? y + 1u : y;      // gridencoder.cu:182: the +1 neighbour is clamped to res-1
    const uint32_t size = g.size[l], mode = g.mode[l];
    auto row = [&](uint32_t xx, uint32_t yy) { const uint32_t p[3] = {xx, yy, z}; return grid_row<3>(p, res, size, mode); };
    const T *tab = table + (size_t)g.off[l] * 2u;
    if constexpr (sizeof(T) == 4) {
        const float2 a = reinterpret_cast<const float2 *>(tab)[row(x, y)], b = reinterpret_cast<const float2 *>(tab)[row(x1, y)];
        reinterpret_cast<float4 *>(const_cast<void *>(pt.base))[pt.off[l] + i] = make_float4(a.x, a.y, b.x, b.y);
    } else {
        const uint32_t *tw = reinterpret_cast<const uint32_t *>(tab);
        reinterpret_cast<uint4 *>(const_cast<void *>(pt.base))[pt.off[l] + i] = make_uint4(tw[row(x, y)], tw[row(x1, y)], tw[row(x, y1)], tw[row(x1, y1)]);
    }
}

// pair-row offsets of the dense prefix; returns the total number of pair rows (0 if the grid has no usable prefix)
static uint32_t pair_layout(const GridLevels &g, int K, uint32_t (&off)[8]) {
    uint32_t total = 0;
    for (int l = 0; l < 8; ++l) off[l] = 0;
    if (K <= 0 || K > 8 || g.C != 2) return 0;
    for (int l = 0; l < K; ++l) { off[l] = total; total += g.res[l] * g.res[l] * g.res[l]; }
    return total;
}

// FinalLv of a grid whose levels are "K dense (packed as pair / quad rows in pt), then hashed, equal-sized and contiguous".
// Returns false when the grid does not have that shape (the generic instantiation then runs).
static bool build_final_lv(const GridLevels &g, int K, const PairTab &pt, const void *table, uint32_t row_bytes, FinalLv &lv) {
    memset(&lv, 0, sizeof(lv));
    if (g.L > 16 || (uint32_t)K >= g.L || g.C != 2 || K < 1 || K > 8 || pt.base == nullptr) return false;
    const uint32_t size = g.size[K];
    if ((size & (size - 1)) != 0 || (uint64_t)size * row_bytes > (1u << 24)) return false;      // 24-bit multiplies cover the mask
    for (uint32_t l = (uint32_t)K; l < g.L; ++l) {
        if (g.size[l] != size || (l + 1 < g.L && g.off[l + 1] != g.off[l] + size) || g.res[l] >= (1u << 24)) return false;
        lv.h_off[l] = (g.off[l] - g.off[K]) * row_bytes;
    }
    lv.h_mask = (size - 1u) * row_bytes;
    lv.hash_base = reinterpret_cast<const char *>(table) + (size_t)g.off[K] * row_bytes;
    lv.pair_base = reinterpret_cast<const char *>(pt.base);
    for (uint32_t l = 0; l < g.L; ++l) { lv.res_f[l] = (float)g.res[l]; lv.top_f[l] = (float)(g.res[l] - 1u); }
    for (uint32_t l = 0; l < (uint32_t)K; ++l) {
        const uint32_t res = g.res[l];
        lv.d_sy[l] = res * 16u; lv.d_sz[l] = res * res * 16u;
        lv.d_ylim[l] = (res - 1u) * lv.d_sy[l]; lv.d_zlim[l] = (res - 1u) * lv.d_sz[l];
        lv.d_off[l] = pt.off[l] * 16u;
    }
    // one and a half cells of the coarsest hashed level: fl(x * res_l - 0.5) stays inside [0.99, res_l - 1.99] on every hashed level
    lv.in_lo = 1.5f / (float)g.res[K];
    lv.in_hi = 1.0f - 1.5f / (float)g.res[K];
    return true;
}

static thread_local sn_launch_info g_launch = {};

enum { PK_PACK = 0, PK_PROP0, PK_PROP1, PK_PROP2, PK_FINAL, PK_FEAT, PK_CLASSES };
struct ProfSpan { hipEvent_t a, b; int cls; };
static bool g_prof_on = false;
static std::vector<ProfSpan> g_prof;

struct ProfScope {
    hipStream_t st; int idx = -1;
    ProfScope(hipStream_t s, int cls) : st(s) {
        if (!g_prof_on) return;
        ProfSpan sp; sp.cls = cls;
        if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) return;
        (void)hipEventRecord(sp.a, st);
        g_prof.push_back(sp); idx = (int)g_prof.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, st); }
};

static int to_levels(GridLevels *g, const sn_grid_desc *d) {
    SN_REQUIRE(d->D == 3, "render: grids must be 3-D (got D=%u)", d->D);
    return build_grid_levels(g, d->offsets, d->D, d->C, d->L, d->S, d->H, d->gridtype, (int)d->align_corners, d->interp);
}

// number of leading dense levels if the grid is "dense prefix, hashed tail" (every grid the reference builds is:
// resolution grows with the level), else -1 (generic run-time level kind)
static int dense_prefix(const GridLevels &g) {
    uint32_t k = 0;
    while (k < g.L && (g.mode[k] & 1u) == 0u) ++k;
    for (uint32_t l = k; l < g.L; ++l) if ((g.mode[l] & 1u) == 0u) return -1;
    if (k == g.L) return -1;   // the paired x-loads of dense levels rely on a hashed level following the dense prefix
    return (int)k;
}

static bool mlp_is(const sn_mlp_desc *m, uint32_t nl, const uint32_t *dims) {
    if (m->num_layers != nl || m->activation != 0 || m->skip_mask != 0) return false;
    for (uint32_t l = 0; l <= nl; ++l) if (m->dims[l] != dims[l]) return false;
    for (uint32_t l = 0; l < nl; ++l) if (m->bias[l] != nullptr || m->weight[l] == nullptr) return false;
    return true;
}

static uint32_t chunk_rays(uint32_t N, uint32_t W) {
    // keep a launch's scratch bounded; image mode splits on 16-row boundaries
    const uint32_t cap = 1u << 21;
    if (N <= cap) return N;
    if (W) { uint32_t rows = (cap / W) & ~15u; if (rows < 16) rows = 16; return rows * W; }
    return cap;
}

// rays per render_rays chunk up to which the stages run sample-parallel (sn_render_tuning: 0 = the default, < 0 = never)
static uint32_t prop_sp_max_rays(const sn_render_cfg *cfg) {
    const int32_t v = cfg->tuning.prop_sp_max_rays;
    return v == 0 ? 32768u : v < 0 ? 0u : (uint32_t)v;        // tools/prop_sp_ab.py: faster up to 32k rays, slower from 64k
}

// Two samples per lane at once in the proposal stages (bit-neutral): pays where the launch leaves the SIMDs short of waves.
// Measured (profiles/r06/prop_pair_ab.json, [128,64,32], proposal stages only): a wave alone on its SIMD issues one vector instruction per
// ~7 cycles whatever its instruction-level parallelism, so the gain is what the interleaved gathers hide -- 64-256 workgroups -11 %,
// 361 -7 %, 484 -4 %, 625 +-1 %, 1406 and more +5 ... +17 % (3 waves per SIMD instead of 5).
static bool prop_pair_auto(uint32_t workgroups_in_flight) { return workgroups_in_flight <= 512u; }

static uint32_t final_sp_max_rays(const sn_render_cfg *cfg) {
    const int32_t v = cfg->tuning.final_sp_max_rays;
    return v == 0 ? 16384u : v < 0 ? 0u : (uint32_t)v;
}

// the linear-tail form of the default final stage (third layer's geometry rows applied once per ray) is used wherever no
// per-sample tensor leaves the kernel; tuning.per_sample_form keeps the per-sample form (bit-identical to the compacting /
// several-lanes-per-ray / role-split kernels, which all evaluate the third layer per sample)
static bool lt_enabled(const sn_render_cfg *cfg) { return cfg->tuning.per_sample_form == 0; }

// the role-split final stage (k_final_stage_rs, experiments builds) where it applies
static bool rs_enabled(const sn_render_cfg *cfg) {
#ifdef SN_EXPERIMENTS
    return cfg->tuning.experiment == SN_EXP_ROLE_SPLIT;
#else
    (void)cfg;
    return false;
#endif
}

static uint32_t tile_log2w(const sn_render_cfg *cfg) { return (cfg && cfg->tuning.wave_tile >= 1 && cfg->tuning.wave_tile <= 5) ? (uint32_t)cfg->tuning.wave_tile : 3u; }
static uint32_t blocks_for(uint32_t n, uint32_t W, uint32_t tlw) {
    if (W) {                                                      // four wave tiles of 2^tlw x 2^(6 - tlw) pixels each (ray_of_lane)
        const uint32_t rows = (n + W - 1) / W, tw = 1u << tlw, th = 64u >> tlw;
        return div_up(((W + tw - 1u) >> tlw) * ((rows + th - 1u) / th), 4u);
    }
    return div_up(n, 256);
}

static size_t stage_scratch_floats(const sn_render_cfg *cfg, uint32_t Npad) {
    size_t f = 0;
    for (uint32_t k = 0; k + 1 < cfg->num_stages; ++k) f += (size_t)cfg->num_steps[k] * Npad;          // weights
    for (uint32_t k = 1; k < cfg->num_stages; ++k) f += (size_t)(cfg->num_steps[k] + 1) * Npad;       // bins
    if (cfg->with_feat) f += (size_t)cfg->num_steps[cfg->num_stages - 1] * Npad;                      // last-stage weights
    return f;
}


// Row bands on two streams (round 4, tuning.band_streams).  A schedule with proposal stages is three kernels of different character --
// the proposal stages are bound by the vector ALU, the last stage by the texture path and the matrix cores -- that a single stream runs one
// after the other, each with its own ramp and tail.  Rendered as TWO row bands on two streams the kernels of one band overlap those of the
// other (measured through captured graphs, tools/band_streams_ab.py: 800x800 [128,64,32] 4.36 -> 4.07 ms fp32, 3.85 -> 3.71 ms fp16; 1600x1600
// 14.76 -> 14.04 / 13.43 -> 12.98; nothing at 400x400, where the stages' latency floors decide).  Same kernels over sub-ranges of the tiles:
// bit-identical images.  The second stream is owned by the library (one per device), forked from and joined to the caller's stream with
// events inside the call, so the call stays asynchronous, ordered on the caller's stream and capturable into a HIP graph.
struct BandPlan { uint32_t chunk, slots; };
static BandPlan band_plan(const sn_render_cfg *cfg, uint32_t N, uint32_t W) {
    BandPlan p;
    p.chunk = chunk_rays(N, W);
    p.slots = 1;
    const int mode = cfg ? cfg->tuning.band_streams : 1;
    if (mode == 1 || W == 0 || !cfg || cfg->num_stages < 2 || N % W != 0) return p;
    // automatic: from 768 workgroups; with the in-render feature stage -- a fourth kernel, bound by the texture path alone -- from 512
    // (400x400 + SAM-feature head, BASELINE configs[2]: 3.00 -> 2.82 ms, 2.58 -> 2.41 with fp16 tables).  Without it (profiles/r06/band_small_ab.json,
    // fp32 / fp16 tables): 361 workgroups -2 / -4 %, 484 -0 / -3 %, 625 (400x400) +0.5 / -2 %, 784 -5 / -8 %, 1024 -6 / -9 %, 1444 -4 / -7 %, 2500 -7 / -4 %:
    // the last stage holds 512 workgroups at a time, and a band's share of a partly filled last round overlaps the other band's proposal stages.
    if (mode == 0 && blocks_for(N, W, tile_log2w(cfg)) < (cfg->with_feat ? 512u : 768u)) return p;
    const uint32_t rows = N / W;
    const uint32_t nbands = mode > 2 ? (uint32_t)mode : 2u;              // band_streams = K > 2: K bands, dealt alternately to the two streams (A/B; measured: see DESIGN section 7)
    const uint32_t half_rows = (((rows + nbands - 1u) / nbands + 15u) / 16u) * 16u;   // whole 16-row tile rows; the first bands take the odd ones
    if (half_rows == 0u || half_rows >= rows) return p;
    const uint64_t c = (uint64_t)half_rows * W;
    if (c < p.chunk) p.chunk = (uint32_t)c;                               // (images beyond the scratch cap keep its chunks and alternate them)
    p.slots = 2;
    return p;
}

// One library-owned side stream per (device, CALLER stream): two host threads that render on different streams of a device get different side
// streams (a shared one would serialise them, and a thread that forks it while capturing a graph would pull the other thread's launches into its
// capture -- advisor, round 4); callers that share a stream are ordered by that stream anyway.
static hipStream_t band_side_stream(hipStream_t caller) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, hipStream_t> streams;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, caller);
    auto it = streams.find(key);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    streams[key] = s;
    return s;
}

// fork on construction (the side stream waits for everything the caller's stream holds so far), join on destruction -- on every path out
// of sn_rm_render_rays, error returns included (an unjoined fork would invalidate a stream capture)
struct BandFork {
    hipStream_t main_st, side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    BandFork(hipStream_t m, bool want) : main_st(m) {
        if (!want) return;
        hipStream_t s = band_side_stream(m);
        if (!s) return;
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) { ev_fork = nullptr; return; }
        if (hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ev_fork); ev_fork = ev_join = nullptr; return; }
        if (hipEventRecord(ev_fork, main_st) != hipSuccess || hipStreamWaitEvent(s, ev_fork, 0) != hipSuccess) {
            (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); ev_fork = ev_join = nullptr; return;
        }
        side = s;
    }
    ~BandFork() {
        if (!side) return;
        (void)hipEventRecord(ev_join, side);
        (void)hipStreamWaitEvent(main_st, ev_join, 0);
        (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join);      // (released by the runtime once the recorded work has passed)
    }
};
}  // namespace sn

using namespace sn;

extern "C" {

int sn_build_flags(void) {
    int f = 0;
#ifdef SN_EXPERIMENTS
    f |= SN_BUILD_EXPERIMENTS;
#endif
#ifdef SN_POISON_LDS
    f |= SN_BUILD_POISON_LDS;
#endif
    return f;
}

int sn_rm_last_launch_info(sn_launch_info *info) {
    SN_REQUIRE(info, "last_launch_info: NULL output");
    *info = g_launch;
    return SN_OK;
}

void sn_rm_profile_enable(int on) {
    g_prof_on = on != 0;
    {   // arm / disarm the final stage's clock probe (see clock_probe)
        const int flag = g_prof_on ? 1 : 0;
        const unsigned long long zero[4] = {0, 0, 0, 0};
        if (g_prof_on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk_buf), zero, sizeof(zero));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk_on), &flag, sizeof(flag));
    }
    if (!g_prof_on) {
        for (auto &sp : g_prof) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
        g_prof.clear();
    }
}

int sn_rm_profile_shader_clock(float *shader_mhz, float *probe_ms) {
    SN_REQUIRE(shader_mhz, "profile_shader_clock: NULL output");
    *shader_mhz = 0.0f;
    if (probe_ms) *probe_ms = 0.0f;
    unsigned long long h[4];
    SN_HIP_OK(hipDeviceSynchronize());
    SN_HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_buf), sizeof(h)));
    int dev = 0, wall_khz = 0;
    SN_HIP_OK(hipGetDevice(&dev));
    SN_HIP_OK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
    if (h[3] <= h[1] || h[2] <= h[0] || wall_khz <= 0) return SN_OK;       // no final-stage launch since enable
    const double wall_s = (double)(h[3] - h[1]) / ((double)wall_khz * 1e3);
    *shader_mhz = (float)((double)(h[2] - h[0]) / wall_s / 1e6);
    if (probe_ms) *probe_ms = (float)(wall_s * 1e3);
    return SN_OK;
}

int sn_rm_profile_read(float *ms_per_class, int32_t *launches_per_class, int n_classes) {
    SN_REQUIRE(ms_per_class && launches_per_class && n_classes >= PK_CLASSES, "profile_read: need %d classes", (int)PK_CLASSES);
    for (int i = 0; i < n_classes; ++i) { ms_per_class[i] = 0.0f; launches_per_class[i] = 0; }
    for (auto &sp : g_prof) {
        SN_HIP_OK(hipEventSynchronize(sp.b));
        float ms = 0.0f;
        SN_HIP_OK(hipEventElapsedTime(&ms, sp.a, sp.b));
        ms_per_class[sp.cls] += ms; launches_per_class[sp.cls] += 1;
        (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);
    }
    g_prof.clear();
    return SN_OK;
}

/* Diagnostics: occupancy-API answer (workgroups per CU) for the fused kernels at their launch LDS sizes.
 * out[0] = k_prop_stage<float>, out[1] = k_final_stage f16x3 <float>, out[2] = f32-MFMA <float>; lds[i] = dynamic LDS bytes used. */
int sn_rm_debug_occupancy(int32_t *out, int32_t *lds, int n) {
    SN_REQUIRE(out && lds && n >= 3, "debug_occupancy: need 3 slots");
    int v = 0;
    SN_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_prop_stage<float, 5, 2, 16, 3>, 256, 0));
    out[0] = v; lds[0] = 0;
    size_t l1 = (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float);
    SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<float, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1));
    SN_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_final_stage<float, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5>, 256, l1));
    out[1] = v; lds[1] = (int32_t)l1;
    size_t l2 = (size_t)(PACK_FLOATS + 4 * 32 * 64) * sizeof(float);
    SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<float, 16, 2, 64, 64, 16, 32, MLP_F32, -1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
    SN_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_final_stage<float, 16, 2, 64, 64, 16, 32, MLP_F32, -1>, 256, l2));
    out[2] = v; lds[2] = (int32_t)l2;
    return SN_OK;
}

// Densified levels (round 3).  The first hashed levels of the main grid have few enough vertices (102^3, 148^3) to be re-laid out per
// render call like the dense ones: one 16-byte pair / quad row per VERTEX, fetched through the hash once per vertex by k_pack_pairs
// instead of once per sample-corner by the march.  The final stage then reads them with 4 (fp32) / 2 (fp16) aligned, spatially coherent
// gathers instead of 8 scattered ones -- 12 fewer gather instructions per wave-sample of 98 -- at the price of a 69 MB pack per call
// (~60 us), so only for renders with enough samples to pay for it.  Same values, same arithmetic: bit-identical.
// SN_RENDER_DENSIFY: unset = automatic, 0 = never, 1 / 2 = force that many levels.
constexpr uint64_t DENSIFY_MIN_SAMPLES = 64ull << 20;       // rays x samples of the last stage (the 69 MB pack costs ~55 us more than the plain one and buys 1.6 % of the kernel: break-even near 45 M)
constexpr uint64_t DENSIFY_MAX_VERTICES = 6ull << 20;       // 96 MB of 16-byte rows
// mode = sn_render_tuning.densify: 0 automatic, 1 never, 2 whenever the kernel exists
static int densify_levels(const GridLevels &g, int K, uint64_t samples, bool f16, int mode) {
    if (K != 5 || g.L != 16 || g.C != 2) return K;           // (the K + 2 = 7 instantiation is built for the main grid's shape)
    // automatic for fp16 tables only: 800x800 [128] 6.35 -> 6.22 ms incl. the pack; with fp32 tables the K = 7 instantiation spills 14
    // registers and the call comes out 0.4 % slower (6.957 -> 6.983 ms)
    const bool want = mode == 2 || (mode == 0 && samples >= DENSIFY_MIN_SAMPLES && f16);
    if (!want) return K;                                     // one extra level alone is not instantiated
    uint64_t v = 0;
    for (int l = 0; l < K + 2; ++l) {
        if ((uint64_t)g.res[l] * g.res[l] * 16u >= (1u << 24)) return K;      // dense byte strides go through 24-bit multiplies
        v += (uint64_t)g.res[l] * g.res[l] * g.res[l];
    }
    return v <= DENSIFY_MAX_VERTICES ? K + 2 : K;
}

// floats of a grid's packed pair / quad rows; densify_samples != 0: the main grid, whose levels 5-6 may be densified for this call
// (sized by the same rule the launch applies: table dtype and tuning.densify)
static size_t pair_floats_of(const sn_grid_desc *d, uint64_t densify_samples = 0, int densify_mode = 1) {
    GridLevels g;
    if (d->D != 3 || d->C != 2 || build_grid_levels(&g, d->offsets, d->D, d->C, d->L, d->S, d->H, d->gridtype, (int)d->align_corners, d->interp) != SN_OK) return 0;
    if (!levels_fast(g)) return 0;
    uint32_t off[8];
    const int K = dense_prefix(g);
    return (size_t)pair_layout(g, densify_samples ? densify_levels(g, K, densify_samples, d->table_dtype == SN_F16, densify_mode) : K, off) * 4u;
}
static size_t pair_region_floats(const sn_render_cfg *cfg, uint64_t main_samples) {
    size_t f = pair_floats_of(&cfg->grid, main_samples, cfg->tuning.densify);
    for (uint32_t k = 0; k + 1 < cfg->num_stages && k < SN_MAX_STAGES; ++k) f += pair_floats_of(&cfg->prop_grid[k]);
    return f;
}

size_t sn_rm_render_workspace_bytes(const sn_render_cfg *cfg, uint32_t N, uint32_t tile_w) {
    if (!cfg || N == 0) return (size_t)PACK_FLOATS * sizeof(float);
    const BandPlan bp = band_plan(cfg, N, tile_w);
    const uint32_t nc = bp.chunk < N ? bp.chunk : N;
    const size_t npad = (size_t)blocks_for(nc, tile_w, tile_log2w(cfg)) * 256u;
    const uint64_t samples = (uint64_t)N * cfg->num_steps[cfg->num_stages ? cfg->num_stages - 1 : 0];
    return ((size_t)bp.slots * stage_scratch_floats(cfg, (uint32_t)npad) + (size_t)PACK_FLOATS + pair_region_floats(cfg, samples)) * sizeof(float);
}

int sn_rm_render_rays(const sn_render_cfg *cfg, const sn_render_io *io, sn_stream_t stream) {
    SN_REQUIRE(cfg && io, "render_rays: cfg/io is NULL");
    if (io->N == 0) return SN_OK;   // empty batch: nothing to launch, pointers may be NULL
    g_launch = sn_launch_info{};
    SN_REQUIRE(io->rays_o && io->rays_d, "render_rays: rays must be device pointers");
    if (io->skip_final) SN_REQUIRE(cfg->num_stages >= 2 && io->bins[cfg->num_stages - 1] && !cfg->with_feat,
                                   "render_rays: skip_final needs >= 2 stages, io->bins[last] for the resampled bins, and no feature stage");
    else SN_REQUIRE(io->image && io->depth && io->weights_sum, "render_rays: outputs must be device pointers");
    SN_REQUIRE(io->out_stride == 0u || io->out_stride >= 5u, "render_rays: out_stride %u must be 0 (dense outputs) or at least 5 floats (rgb | depth | weights_sum per row)", io->out_stride);
    SN_REQUIRE(io->head_stride == 0u || (io->f_image && io->head_stride >= 31u + 4u && !io->skip_final),
               "render_rays: head_stride %u needs f_image and room for f_image | rgb | depth (>= 35 floats per row)", io->head_stride);
    SN_REQUIRE(cfg->tuning.wave_tile >= 0 && cfg->tuning.wave_tile <= 5, "render_rays: tuning.wave_tile %d outside 0..5", cfg->tuning.wave_tile);
    SN_REQUIRE(!(rs_enabled(cfg) && cfg->tuning.wave_tile != 0 && cfg->tuning.wave_tile != 3), "render_rays: the role-split experiment is built for 8x8 wave tiles (tuning.wave_tile 0 or 3)");
    const uint32_t S = cfg->num_stages;
    SN_REQUIRE(S >= 1 && S <= SN_MAX_STAGES, "render_rays: num_stages=%u outside 1..%d", S, SN_MAX_STAGES);
    for (uint32_t k = 0; k < S; ++k) SN_REQUIRE(cfg->num_steps[k] >= 1, "render_rays: num_steps[%u] must be >= 1", k);
    SN_REQUIRE(cfg->sh_degree == 4, "render_rays: fused path is built for SH degree 4 (network.py:97), got %u", cfg->sh_degree);
    if (io->N == 0) return SN_OK;
    if (io->tile_w) SN_REQUIRE(io->N % io->tile_w == 0, "render_rays: N=%u is not a multiple of tile_w=%u", io->N, io->tile_w);
    // per-ray tables: a row holds T+1 values (sanerf_hip.h); a shorter stride would read past the caller's rows
    SN_REQUIRE(io->bins0_ray_stride == 0 || (io->bins0_table && io->bins0_ray_stride >= cfg->num_steps[0] + 1u),
               "render_rays: bins0_ray_stride=%u needs bins0_table and at least num_steps[0]+1=%u values per ray", io->bins0_ray_stride, cfg->num_steps[0] + 1u);
    for (uint32_t k = 1; k < S; ++k)
        SN_REQUIRE(io->u_ray_stride[k] == 0 || (io->u_table[k] && io->u_ray_stride[k] >= cfg->num_steps[k] + 1u),
                   "render_rays: u_ray_stride[%u]=%u needs u_table[%u] and at least num_steps[%u]+1=%u values per ray", k, io->u_ray_stride[k], k, k, cfg->num_steps[k] + 1u);
    const hipStream_t st_main = (hipStream_t)stream;
    hipStream_t st = st_main;      // (the chunk loop below shadows it with the chunk's own stream when the image is rendered as bands on two streams)

    // ---- which kernel instantiations does this configuration map to? ----
    static const uint32_t d_prop[3] = {10, 16, 1};
    static const uint32_t d_main[4] = {32, 64, 64, 16};
    static const uint32_t d_view[4] = {31, 32, 32, 3};
    GridLevels gl_prop[SN_MAX_STAGES], gl_main;
    for (uint32_t k = 0; k + 1 < S; ++k) {
        int rc = to_levels(&gl_prop[k], &cfg->prop_grid[k]);
        if (rc) return rc;
        if (!(cfg->prop_grid[k].L == 5 && cfg->prop_grid[k].C == 2 && mlp_is(&cfg->prop_mlp[k], 2, d_prop) && levels_fast(gl_prop[k]))) {
            set_error("render_rays: proposal stage %u is not the L=5,F=2 grid + 10-16-1 bias-free ReLU MLP this build fuses (network.py:135-143)", k);
            return SN_ERR_UNSUPPORTED;
        }
    }
    {
        int rc = to_levels(&gl_main, &cfg->grid);
        if (rc) return rc;
    }
    const bool is_main = cfg->grid.L == 16 && cfg->grid.C == 2 && mlp_is(&cfg->grid_mlp, 3, d_main) && mlp_is(&cfg->view_mlp, 3, d_view) && levels_fast(gl_main);
    // any other field of the same structure (sizes at run time: k_final_stage_any; BASELINE configs[0] is one)
    AnyShape any_shape;
    memset(&any_shape, 0, sizeof(any_shape));
    bool is_any = false;
    if (!is_main) {
        const sn_mlp_desc *gm = &cfg->grid_mlp, *vm = &cfg->view_mlp;
        auto plain = [](const sn_mlp_desc *m) {
            if (m->num_layers < 1u || m->num_layers > ANY_LAYERS || m->activation != 0u || m->skip_mask != 0u) return false;
            for (uint32_t l = 0; l < m->num_layers; ++l) if (m->bias[l] != nullptr || m->weight[l] == nullptr) return false;
            for (uint32_t l = 0; l <= m->num_layers; ++l) if (m->dims[l] < 1u || m->dims[l] > ANY_W) return false;
            return true;
        };
        is_any = cfg->grid.D == 3 && cfg->grid.C == 2 && cfg->grid.L >= 1 && cfg->grid.L * 2u <= ANY_W && plain(gm) && plain(vm) &&
                 gm->dims[0] == cfg->grid.L * 2u && gm->dims[gm->num_layers] >= 2u && gm->dims[gm->num_layers] - 1u <= ANY_GEO &&
                 cfg->sh_degree == 4u && vm->dims[0] == gm->dims[gm->num_layers] - 1u + 16u && vm->dims[vm->num_layers] == 3u;
        if (is_any) {
            any_shape.ng = gm->num_layers; any_shape.nv = vm->num_layers;
            for (uint32_t l = 0; l <= gm->num_layers; ++l) any_shape.dg[l] = gm->dims[l];
            for (uint32_t l = 0; l <= vm->num_layers; ++l) any_shape.dv[l] = vm->dims[l];
            for (uint32_t l = 0; l < gm->num_layers; ++l) any_shape.wg[l] = gm->weight[l];
            for (uint32_t l = 0; l < vm->num_layers; ++l) any_shape.wv[l] = vm->weight[l];
            uint32_t wmax = 0;
            for (uint32_t l = 0; l <= gm->num_layers; ++l) wmax = gm->dims[l] > wmax ? gm->dims[l] : wmax;
            for (uint32_t l = 0; l <= vm->num_layers; ++l) wmax = vm->dims[l] > wmax ? vm->dims[l] : wmax;
            any_shape.rows = wmax <= 16u ? 16u : wmax <= 32u ? 32u : 64u;
        }
    }
    if (!is_main && !is_any) {
        set_error("render_rays: the field is neither the L=16,F=2 grid + 32-64-64-16 / 31-32-32-3 MLPs (network.py:93-98) nor of the shape the "
                  "size-agnostic final stage takes (3-D grid with level_dim 2 and <= 32 levels' worth of 64 features, bias-free ReLU MLPs of "
                  "<= 4 layers and <= 64 neurons, <= 31 geometry channels, degree-4 SH)");
        return SN_ERR_UNSUPPORTED;
    }
    if (is_any && (cfg->compact_live || cfg->with_feat)) {
        set_error("render_rays: compact_live / the in-render feature stage are instantiated for the reference network's sizes only");
        return SN_ERR_UNSUPPORTED;
    }
    GridLevels gl_feat;
    int feat_lg = 2;   // levels per workgroup pass of the feature stage (more passes = more workgroups, fewer registers)
    if (cfg->with_feat) {
        SN_REQUIRE(io->f_feat, "render_rays: with_feat is set but io->f_feat is NULL");
        SN_REQUIRE(cfg->feat_grid.embeddings && table_aligned(cfg->feat_grid.embeddings) && table_aligned(io->f_feat),
                   "render_rays: feature grid table / f_feat must be 16-byte aligned device pointers");
        int rc = to_levels(&gl_feat, &cfg->feat_grid);
        if (rc) return rc;
        if (!levels_fast(gl_feat) || !(gl_feat.C == 2 || gl_feat.C == 4 || gl_feat.C == 8)) {
            set_error("render_rays: feature grid must be a hash grid with level_dim 2/4/8, align_corners=False, linear "
                      "interpolation (network.py:103); use sn_rm_grid_composite for other grids");
            return SN_ERR_UNSUPPORTED;
        }
        if (cfg->tuning.feat_levels) feat_lg = cfg->tuning.feat_levels;
        if (feat_lg != 1 && feat_lg != 2 && feat_lg != 4) feat_lg = 2;
        while (gl_feat.L % (uint32_t)feat_lg) feat_lg >>= 1;
    }
    // tuning.mlp_mode: SN_MLP_AUTO (fp16 hi/lo split on the matrix cores, fp32 accumulate, unless cfg->mlp_exact_fp32), SN_MLP_F16X3 (forced),
    //                  SN_MLP_MFMA32 (exact fp32 v_mfma_f32_32x32x2_f32), SN_MLP_VALU (vector-ALU fallback)
    int mlp_mode = cfg->mlp_exact_fp32 ? MLP_F32 : MLP_F16X3;
    int mlp_np = 3;                // products per split multiply (SN_MLP_F16X2 / SN_MLP_F16X1: opt-in reduced forms of the plain linear-tail launch with fp16 tables)
    switch (cfg->tuning.mlp_mode) {
        case SN_MLP_F16X1: mlp_mode = MLP_F16X3; mlp_np = 1; break;
        case SN_MLP_AUTO: break;
        case SN_MLP_F16X3: mlp_mode = MLP_F16X3; break;
        case SN_MLP_MFMA32: mlp_mode = MLP_F32; break;
        case SN_MLP_VALU: mlp_mode = MLP_VALU; break;
        default: set_error("render_rays: unknown tuning.mlp_mode=%d", cfg->tuning.mlp_mode); return SN_ERR_INVALID;
    }
    if (cfg->tuning.experiment != SN_EXP_NONE && !(sn_build_flags() & SN_BUILD_EXPERIMENTS)) {
        set_error("render_rays: tuning.experiment=%d needs a library built with -DSN_EXPERIMENTS (make exp)", cfg->tuning.experiment);
        return SN_ERR_UNSUPPORTED;
    }
    const bool use_mfma = mlp_mode != MLP_VALU;

    SN_REQUIRE(io->workspace != nullptr, "render_rays: workspace is NULL");
    SN_REQUIRE(table_aligned(io->workspace), "render_rays: workspace must be 16-byte aligned");
    float *pack = reinterpret_cast<float *>(io->workspace);
    const uint64_t main_samples = (uint64_t)io->N * cfg->num_steps[S - 1];
    const size_t pair_floats = pair_region_floats(cfg, main_samples);       // (as sn_rm_render_workspace_bytes sized it)
    // will the last stage run as the plain linear-tail kernel?  Only that one has the instantiation that reads densified levels.
    const bool plain_lt = mlp_mode == MLP_F16X3 && lt_enabled(cfg) && !rs_enabled(cfg) && !cfg->with_feat && !cfg->compact_live && !io->skip_final &&
                          !(cfg->early_stop_eps > 0.0f && cfg->early_stop_eps < 1.0f) && dense_prefix(gl_main) == 5 &&
                          !(io->bins[S - 1] || io->weights[S - 1] || io->sigmas[S - 1] || io->xyzs_last || io->geo_feat_last);
    const int Kv_main = plain_lt ? densify_levels(gl_main, 5, main_samples, cfg->grid.table_dtype == SN_F16, cfg->tuning.densify) : dense_prefix(gl_main);
    float *pair_mem = pack + PACK_FLOATS;
    float *scratch = pair_mem + pair_floats;
    const size_t head_floats = (size_t)PACK_FLOATS + pair_floats;
    const size_t scratch_floats_avail = io->workspace_bytes / sizeof(float) > head_floats ? io->workspace_bytes / sizeof(float) - head_floats : 0;
    PairTab pairs;
    pairs.base = nullptr;
    for (int l = 0; l < 8; ++l) pairs.off[l] = 0;
    if (use_mfma && !is_any) {
        ProfScope ps(st, PK_PACK);
        if (mlp_mode == MLP_F16X3)
            hipLaunchKernelGGL(k_pack_grid_mlp_f16, dim3(PACK16_VECS * 64 / 256), dim3(256), 0, st, cfg->grid_mlp.weight[0],
                               cfg->grid_mlp.weight[1], cfg->grid_mlp.weight[2], reinterpret_cast<uint4 *>(pack));
        else
            hipLaunchKernelGGL(k_pack_grid_mlp, dim3(PACK_FLOATS / 256), dim3(256), 0, st, cfg->grid_mlp.weight[0], cfg->grid_mlp.weight[1], cfg->grid_mlp.weight[2], pack);
        SN_LAUNCH_CHECK("k_pack_grid_mlp");
    }
    // aligned x-pair rows of every grid's dense levels (PairTab), re-packed on every call: the call stays stateless
    PairTab prop_pairs[SN_MAX_STAGES];
    {
        float *cursor = pair_mem;
        auto pack_pairs = [&](const sn_grid_desc *d, const GridLevels &gl, PairTab &pt, int Kp, size_t fl) -> int {
            pt.base = nullptr;
            for (int l = 0; l < 8; ++l) pt.off[l] = 0;
            if (fl == 0 || Kp <= 0) return SN_OK;
            pair_layout(gl, Kp, pt.off);
            pt.base = cursor;
            cursor += fl;
            uint32_t max_rows = 0;
            for (int l = 0; l < Kp; ++l) { const uint32_t r3 = gl.res[l] * gl.res[l] * gl.res[l]; if (r3 > max_rows) max_rows = r3; }
            const dim3 gp(div_up(max_rows, 256), (uint32_t)Kp);
            if (d->table_dtype == SN_F16) hipLaunchKernelGGL(k_pack_pairs<__half>, gp, dim3(256), 0, st, (const __half *)d->embeddings, gl, pt, (uint32_t)Kp);
            else hipLaunchKernelGGL(k_pack_pairs<float>, gp, dim3(256), 0, st, (const float *)d->embeddings, gl, pt, (uint32_t)Kp);
            SN_LAUNCH_CHECK("k_pack_pairs");
            return SN_OK;
        };
        ProfScope ps(st, PK_PACK);
        // (the main grid's region is sized for the densified layout whenever the sample count allows it; levels 5-6 are packed only
        //  when the kernel that reads them will run)
        int rcp = pack_pairs(&cfg->grid, gl_main, pairs, is_any ? 0 : Kv_main, pair_floats_of(&cfg->grid, main_samples, cfg->tuning.densify));
        if (rcp) return rcp;
        for (uint32_t k = 0; k + 1 < S; ++k) {
            rcp = pack_pairs(&cfg->prop_grid[k], gl_prop[k], prop_pairs[k], dense_prefix(gl_prop[k]), pair_floats_of(&cfg->prop_grid[k]));
            if (rcp) return rcp;
        }
    }

    // opt-in compaction: k_final_stage_cmp replaces the last stage when nothing per-sample leaves the call (those tensors
    // must be complete) and the table has the FinalLv shape; the proposal stages then drop waves of missed rays as well
    bool use_cmp = false;
    if (cfg->compact_live && mlp_mode == MLP_F16X3 && !cfg->with_feat && !io->xyzs_last && !io->geo_feat_last && !io->skip_final &&
        io->bins0_ray_stride == 0) {
        bool any_dbg = false;
        for (uint32_t k = 1; k < S; ++k) any_dbg = any_dbg || io->u_ray_stride[k] != 0;      // per-ray jitter tables: the default kernels
        for (uint32_t k = 0; k < S; ++k) any_dbg = any_dbg || io->bins[k] || io->weights[k] || io->sigmas[k] || io->inds[k];
        FinalLv probe;
        use_cmp = !any_dbg && dense_prefix(gl_main) == 5 &&
                  build_final_lv(gl_main, 5, pairs, cfg->grid.embeddings, 2u * (cfg->grid.table_dtype == SN_F16 ? 2u : 4u), probe);
    }
    const uint32_t W = io->tile_w;
    const BandPlan bplan = band_plan(cfg, io->N, W);
    const uint32_t chunk = bplan.chunk;
    // two bands on two streams: everything above (weight and pair packs) is on the caller's stream, the fork waits for it
    BandFork fork(st_main, bplan.slots == 2u && chunk < io->N);
    uint32_t chunk_index = 0;
    for (uint32_t first = 0; first < io->N; first += chunk, ++chunk_index) {
        const uint32_t n = (io->N - first) < chunk ? (io->N - first) : chunk;
        const uint32_t nblk = blocks_for(n, W, tile_log2w(cfg));
        const uint32_t Npad = nblk * 256u;
        const uint32_t slot = fork.side ? (chunk_index & 1u) : 0u;
        hipStream_t st = slot ? fork.side : st_main;                        // shadows the function-level `st`: this chunk's kernels and profile spans
        const size_t slot_floats = stage_scratch_floats(cfg, blocks_for(chunk < io->N ? chunk : io->N, W, tile_log2w(cfg)) * 256u);
        if ((size_t)(fork.side ? 2u : 1u) * slot_floats > scratch_floats_avail || stage_scratch_floats(cfg, Npad) > slot_floats) {
            set_error("render_rays: workspace too small (%zu bytes, need %zu)", io->workspace_bytes,
                      ((size_t)bplan.slots * slot_floats + head_floats) * sizeof(float));
            return SN_ERR_WORKSPACE;
        }
        RayCommon rc;
        rc.rays_o = io->rays_o + (size_t)first * 3; rc.rays_d = io->rays_d + (size_t)first * 3;
        rc.cnf = io->cam_near_far ? io->cam_near_far + (size_t)first * 2 : nullptr;
        rc.N = n; rc.W = W; rc.rows = W ? n / W : 0; rc.Npad = Npad;
        for (int i = 0; i < 6; ++i) rc.aabb[i] = cfg->aabb[i];
        rc.min_near = cfg->min_near; rc.bound = cfg->bound; rc.contract = cfg->contract;
        rc.last_opaque = cfg->last_sample_opaque; rc.bg = cfg->bg_color;
        {
            int e = 0;
            const float m = frexpf(2.0f * cfg->bound, &e);
            rc.inv_den = (m == 0.5f && e > -100 && e < 100) ? 1.0f / (2.0f * cfg->bound) : 0.0f;
        }
        {   // tuning.linear_tile_order disables the XCD-aware tile order (A/B switch)
            rc.xcd_swizzle = cfg->tuning.linear_tile_order == 0;
            rc.tlw = tile_log2w(cfg);
            // compaction: workgroups differ in cost by the number of live rays of their tile; a contiguous tile range per
            // XCD would leave the XCDs that own empty image bands idle (measured on the small-aabb scene: 4.4 instead of
            // 3.0 ms), so tiles go round-robin over the XCDs in dispatch order
            if (use_cmp) rc.xcd_swizzle = 0;
        }

        // scratch carve-up
        float *w_scr[SN_MAX_STAGES] = {nullptr}, *b_scr[SN_MAX_STAGES] = {nullptr};
        float *cur = scratch + (size_t)slot * slot_floats;                    // (bands in flight on two streams: a scratch region each)
        for (uint32_t k = 0; k + 1 < S; ++k) { w_scr[k] = cur; cur += (size_t)cfg->num_steps[k] * Npad; }
        for (uint32_t k = 1; k < S; ++k) { b_scr[k] = cur; cur += (size_t)(cfg->num_steps[k] + 1) * Npad; }
        if (cfg->with_feat) { w_scr[S - 1] = cur; cur += (size_t)cfg->num_steps[S - 1] * Npad; }

        for (uint32_t k = 0; k + 1 < S; ++k) {
            PropArgs pa;
            pa.rc = rc; pa.g = gl_prop[k]; pa.table = cfg->prop_grid[k].embeddings;
            pa.w0 = cfg->prop_mlp[k].weight[0]; pa.w1 = cfg->prop_mlp[k].weight[1];
            pa.T = cfg->num_steps[k]; pa.Tn = cfg->num_steps[k + 1];
            pa.bins_in = b_scr[k];
            pa.bins0_stride = io->bins0_ray_stride; pa.u_stride = io->u_ray_stride[k + 1];
            pa.bins0_tab = (k == 0 && io->bins0_table) ? io->bins0_table + (size_t)first * pa.bins0_stride : nullptr;
            pa.u_tab = io->u_table[k + 1] ? io->u_table[k + 1] + (size_t)first * pa.u_stride : nullptr;
            pa.dbg_bins_next = (io->skip_final && k + 2 == S && io->bins[S - 1]) ? io->bins[S - 1] + (size_t)first * (cfg->num_steps[S - 1] + 1) : nullptr;
            pa.w_scr = w_scr[k]; pa.bins_out = b_scr[k + 1];
            pa.dbg_bins = io->bins[k] ? io->bins[k] + (size_t)first * (pa.T + 1) : nullptr;
            pa.dbg_w = io->weights[k] ? io->weights[k] + (size_t)first * pa.T : nullptr;
            pa.dbg_sigma = io->sigmas[k] ? io->sigmas[k] + (size_t)first * pa.T : nullptr;
            pa.dbg_inds = io->inds[k + 1] ? io->inds[k + 1] + (size_t)first * (pa.Tn + 1) : nullptr;
            pa.pairs = prop_pairs[k];
            pa.skip_miss = use_cmp ? 1 : 0;
            {
                ProfScope ps(st, PK_PROP0 + (int)k);
                const int K = dense_prefix(gl_prop[k]);
                const bool h16 = cfg->prop_grid[k].table_dtype != SN_F32;
                // few rays in linear order (training batches): 8 lanes per ray instead of one (k_prop_stage_sp)
                const bool sp = W == 0 && n <= prop_sp_max_rays(cfg) && pa.T <= SP_MAX_T;
                // lanes per ray of the small-batch kernel: the fewest that still give >= 512 workgroups (tuning.prop_sp_lanes forces 8 / 16 / 32)
                uint32_t lpr = 32u;      // (measured, profiles/r06/prop_sp_lanes_ab.json: 32 lanes are the fastest from 1024 to 32768 rays)
                if (cfg->tuning.prop_sp_lanes == 8 || cfg->tuning.prop_sp_lanes == 16 || cfg->tuning.prop_sp_lanes == 32) lpr = (uint32_t)cfg->tuning.prop_sp_lanes;
                const uint32_t nblk_sp = Npad / (256u / lpr);   // every scratch column, like the one-lane-per-ray launch
                // two samples of a ray at once (k_prop_stage<..., UN = 2>): tuning.prop_pair 2 = always, 1 = never, 0 = automatic
                const bool pair = cfg->tuning.prop_pair == 2 ||
                                  (cfg->tuning.prop_pair == 0 && prop_pair_auto(fork.side ? blocks_for(io->N, W, tile_log2w(cfg)) : nblk));   // (two bands run side by side)
#define SN_LAUNCH_PROP_SP(TT_, KK)                                                                                     \
                do {                                                                                               \
                    if (lpr == 8u) hipLaunchKernelGGL((k_prop_stage_sp<TT_, 5, 2, 16, KK, 8>), dim3(nblk_sp), dim3(256), 0, st, pa);        \
                    else if (lpr == 16u) hipLaunchKernelGGL((k_prop_stage_sp<TT_, 5, 2, 16, KK, 16>), dim3(nblk_sp), dim3(256), 0, st, pa); \
                    else hipLaunchKernelGGL((k_prop_stage_sp<TT_, 5, 2, 16, KK, 32>), dim3(nblk_sp), dim3(256), 0, st, pa);                 \
                } while (0)
#define SN_LAUNCH_PROP(KK)                                                                                         \
                do {                                                                                               \
                    if (sp && h16) SN_LAUNCH_PROP_SP(__half, KK);                                                  \
                    else if (sp) SN_LAUNCH_PROP_SP(float, KK);                                                     \
                    else if (h16 && pair) hipLaunchKernelGGL((k_prop_stage<__half, 5, 2, 16, KK, 2>), dim3(nblk), dim3(256), 0, st, pa); \
                    else if (pair) hipLaunchKernelGGL((k_prop_stage<float, 5, 2, 16, KK, 2>), dim3(nblk), dim3(256), 0, st, pa); \
                    else if (h16) hipLaunchKernelGGL((k_prop_stage<__half, 5, 2, 16, KK>), dim3(nblk), dim3(256), 0, st, pa); \
                    else hipLaunchKernelGGL((k_prop_stage<float, 5, 2, 16, KK>), dim3(nblk), dim3(256), 0, st, pa);     \
                } while (0)
                if (K == 3) SN_LAUNCH_PROP(3);          // prop0: res 16, 27, 46 dense (network.py:135)
                else if (K == 2) SN_LAUNCH_PROP(2);     // prop1: res 16, 32 dense (network.py:140)
                else SN_LAUNCH_PROP(-1);
#undef SN_LAUNCH_PROP_SP
#undef SN_LAUNCH_PROP
            }
            SN_LAUNCH_CHECK("k_prop_stage");
        }
        FinalArgs fa;
        fa.rc = rc; fa.g = gl_main; fa.table = cfg->grid.embeddings; fa.mlp_pack = pack;
        for (int l = 0; l < 3; ++l) { fa.w[l] = cfg->grid_mlp.weight[l]; fa.vw[l] = cfg->view_mlp.weight[l]; }
        fa.T = cfg->num_steps[S - 1];
        if (io->skip_final) continue;                      // proposal stages only: their last resampled bins went to io->bins[S-1]
        fa.bins_in = b_scr[S - 1]; fa.bins0_stride = io->bins0_ray_stride;
        fa.bins0_tab = (S == 1 && io->bins0_table) ? io->bins0_table + (size_t)first * fa.bins0_stride : nullptr;
        fa.sh_degree = cfg->sh_degree;
        fa.istride = io->out_stride ? io->out_stride : 3u; fa.sstride = io->out_stride ? io->out_stride : 1u;
        fa.image = io->image + (size_t)first * fa.istride; fa.depth = io->depth + (size_t)first * fa.sstride; fa.wsum = io->weights_sum + (size_t)first * fa.sstride;
        fa.dbg_bins = io->bins[S - 1] ? io->bins[S - 1] + (size_t)first * (fa.T + 1) : nullptr;
        fa.dbg_w = io->weights[S - 1] ? io->weights[S - 1] + (size_t)first * fa.T : nullptr;
        fa.dbg_sigma = io->sigmas[S - 1] ? io->sigmas[S - 1] + (size_t)first * fa.T : nullptr;
        fa.dbg_xyz = io->xyzs_last ? io->xyzs_last + (size_t)first * fa.T * 3 : nullptr;
        fa.dbg_geo = io->geo_feat_last ? io->geo_feat_last + (size_t)first * fa.T * 15 : nullptr;
        fa.fimg_stride = io->head_stride ? io->head_stride : 31u;
        fa.dbg_fimg = io->f_image ? io->f_image + (size_t)first * fa.fimg_stride : nullptr;
        fa.head_rgbd = (io->head_stride && io->f_image) ? fa.dbg_fimg + 31 : nullptr;
        fa.pairs = pairs;
        const bool lv_ok = build_final_lv(gl_main, dense_prefix(gl_main), pairs, cfg->grid.embeddings,
                                          2u * (cfg->grid.table_dtype == SN_F16 ? 2u : 4u), fa.lv);
        fa.w_out = cfg->with_feat ? w_scr[S - 1] : nullptr;
        // early termination is honoured only when nothing per-sample leaves the kernel (those tensors would be left
        // unwritten past the stop)
        const bool per_sample_out = fa.dbg_bins || fa.dbg_w || fa.dbg_sigma || fa.dbg_xyz || fa.dbg_geo || fa.w_out;
        fa.stop_cum = (cfg->early_stop_eps > 0.0f && cfg->early_stop_eps < 1.0f && !per_sample_out) ? -logf(cfg->early_stop_eps) : 0.0f;
        const bool f16 = cfg->grid.table_dtype == SN_F16;
        // what runs as the last stage (sn_rm_last_launch_info): gather instructions of one wave-sample = dense levels as packed rows
        // (fp32 tables: pair rows, 4 loads; fp16: quad rows, 2 loads) + 8 corners per hashed level
        auto lv_gathers = [&](int kd) -> uint32_t { return (uint32_t)kd * (f16 ? 2u : 4u) + (uint32_t)(16 - kd) * 8u; };
        auto note = [&](const char *name, uint32_t wgs, size_t lds, int dense, uint32_t gpw) {
            snprintf(g_launch.final_kernel, sizeof(g_launch.final_kernel), "%s", name);
            g_launch.workgroups = wgs; g_launch.lds_bytes = (uint32_t)lds; g_launch.dense_levels = (uint32_t)(dense < 0 ? 0 : dense);
            g_launch.gathers_per_wave_sample = gpw; g_launch.launches += 1u;
        };
        if (is_any) {     // sizes at run time
            ProfScope ps_final(st, PK_FINAL);
            note("k_final_stage_any", nblk, (size_t)(2u * any_shape.rows + (any_shape.dg[any_shape.ng] - 1u)) * 256u * sizeof(float), 0, cfg->grid.L * 8u);
            const uint32_t geo = any_shape.dg[any_shape.ng] - 1u;
            fa.dbg_geo = io->geo_feat_last ? io->geo_feat_last + (size_t)first * fa.T * geo : nullptr;
            fa.fimg_stride = io->head_stride ? io->head_stride : geo + 16u;
            fa.dbg_fimg = io->f_image ? io->f_image + (size_t)first * fa.fimg_stride : nullptr;
            fa.head_rgbd = nullptr;
            const size_t lds_bytes = (size_t)(2u * any_shape.rows + geo) * 256u * sizeof(float);
            if (f16) {
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_any<__half>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL(k_final_stage_any<__half>, dim3(nblk), dim3(256), lds_bytes, st, fa, any_shape);
            } else {
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_any<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL(k_final_stage_any<float>, dim3(nblk), dim3(256), lds_bytes, st, fa, any_shape);
            }
            SN_LAUNCH_CHECK("k_final_stage_any");
            continue;
        }
        {
        ProfScope ps_final(st, PK_FINAL);
#define SN_LAUNCH_FINAL_T(TT_, MODE_, KK, AUX_, LDS_FLOATS)                                                                      \
        do {                                                                                                                 \
            const size_t lds_bytes = (size_t)(LDS_FLOATS) * sizeof(float);                                                   \
            SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<TT_, 16, 2, 64, 64, 16, 32, MODE_, KK, AUX_>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                      \
            hipLaunchKernelGGL((k_final_stage<TT_, 16, 2, 64, 64, 16, 32, MODE_, KK, AUX_>), dim3(nblk), dim3(256), lds_bytes, st, fa); \
        } while (0)
#define SN_LAUNCH_FINAL(MODE_, KK, LDS_FLOATS)                                                                                  \
        do {                                                                                                                 \
            if (f16) SN_LAUNCH_FINAL_T(__half, MODE_, KK, true, LDS_FLOATS); else SN_LAUNCH_FINAL_T(float, MODE_, KK, true, LDS_FLOATS); \
        } while (0)
#define SN_LAUNCH_FINAL_AUX(MODE_, KK, LDS_FLOATS)   /* default mode only: plain instantiation unless the extras are on */ \
        do {                                                                                                                 \
            if (aux) SN_LAUNCH_FINAL(MODE_, KK, LDS_FLOATS);                                                                   \
            else if (f16) SN_LAUNCH_FINAL_T(__half, MODE_, KK, false, LDS_FLOATS);                                             \
            else SN_LAUNCH_FINAL_T(float, MODE_, KK, false, LDS_FLOATS);                                                       \
        } while (0)
        const bool aux = fa.w_out != nullptr || fa.stop_cum > 0.0f;
        // exact early-out of the LAST stage (linear-tail instantiations): opt-in (tuning.exact_early_out = 2).  Measured on the opaque field
        // (profiles/r04/bench_configs.json): behind proposal stages the last stage's 32 samples sit around the surface and the test buys nothing
        // (1.953 vs 1.952 ms -- the proposal stages' own early-out, always on, is what takes that render from 3.19 to 1.95 ms) while it costs a
        // semi-transparent scene 0.2-0.3 %; a single-stage schedule gains (6.07 -> 4.43 ms) but its kernel is the bench line's (0.5-1 % cost).
        const bool eo = cfg->tuning.exact_early_out == 2;
#ifdef SN_EXPERIMENTS
        // SN_EXP_FINAL_ONE_WG (round 6): the last stage asks for 84 KiB of LDS, so only ONE of its workgroups fits a CU and half of every SIMD's
        // registers stay free for the proposal-stage workgroups of the OTHER row band (tuning.band_streams): vector-ALU-bound waves beside
        // texture-bound ones on the same SIMD instead of on different CUs
        const size_t lds_pad_exp = cfg->tuning.experiment == SN_EXP_FINAL_ONE_WG ? (size_t)(84 * 1024) - (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float) : 0u;
#else
        const size_t lds_pad_exp = 0u;
#endif
        constexpr int VIEW_W = 32 * 32 + 32 * 32 + 3 * 32;     // padded view_mlp rows
        static_assert(VIEW_W <= PACK_FLOATS, "view weights overlay the packed MLP weights");
        const int Kmain = !lv_ok ? -1 : dense_prefix(gl_main);   // the K = 5 instantiations read FinalLv
        // few rays in linear order: lanes share rays (k_final_stage_sp); fewer samples per lane while CUs would idle
        const bool final_sp = W == 0 && mlp_mode == MLP_F16X3 && fa.stop_cum == 0.0f && !use_cmp && n <= final_sp_max_rays(cfg) &&
                              fa.T <= 64u * FSP_MAX_SPL;
        if (final_sp) {
            auto lpr_log2_of = [&](uint32_t sl) { const uint32_t need = div_up(fa.T, 1u << sl); uint32_t l2 = 0; while ((1u << l2) < need) ++l2; return l2; };
            uint32_t spl_log2 = 2;
            while (spl_log2 > 0 && div_up(fa.T, 1u << (spl_log2 - 1)) <= 64u && (((size_t)Npad << lpr_log2_of(spl_log2)) >> 8) < SN_FINAL_SP_MIN_BLOCKS) --spl_log2;
            const uint32_t lpr_log2 = lpr_log2_of(spl_log2);
            const uint32_t blocks = (uint32_t)(((size_t)Npad << lpr_log2) >> 8);
            const size_t lds_bytes = final_sp_lds_floats(1u << spl_log2) * sizeof(float);
#define SN_LAUNCH_FINAL_SP(TT_, KK)                                                                                              \
            do {                                                                                                             \
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_sp<TT_, KK>),                    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                  \
                hipLaunchKernelGGL((k_final_stage_sp<TT_, KK>), dim3(blocks), dim3(256), lds_bytes, st, fa, lpr_log2, spl_log2); \
            } while (0)
            if (Kmain == 5) { if (f16) SN_LAUNCH_FINAL_SP(__half, 5); else SN_LAUNCH_FINAL_SP(float, 5); }
            else { if (f16) SN_LAUNCH_FINAL_SP(__half, -1); else SN_LAUNCH_FINAL_SP(float, -1); }
#undef SN_LAUNCH_FINAL_SP
            note("k_final_stage_sp", blocks, lds_bytes, Kmain, Kmain == 5 ? lv_gathers(5) : 16u * 8u);
        } else if (use_cmp) {
            // per-ray termination + live-sample compaction (k_final_stage_cmp); f_image stays available
            const size_t lds_bytes = (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float);
            note("k_final_stage_cmp", nblk, lds_bytes, 5, lv_gathers(5));
            if (f16) {
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_cmp<__half, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL((k_final_stage_cmp<__half, 5>), dim3(nblk), dim3(256), lds_bytes, st, fa);
            } else {
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_cmp<float, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL((k_final_stage_cmp<float, 5>), dim3(nblk), dim3(256), lds_bytes, st, fa);
            }
#ifdef SN_EXPERIMENTS
        } else if (mlp_mode == MLP_F16X3 && Kmain == 5 && !aux && !per_sample_out && rs_enabled(cfg)) {
            // role-split waves (k_final_stage_rs): 768-thread workgroups over 32x16-pixel tiles (512 rays in linear order)
            const uint32_t nblk_rs = W ? ((W + 31u) >> 5) * ((rc.rows + 15u) >> 4) : div_up(n, (uint32_t)(RS_PROD * 64));
            const size_t lds_bytes = (size_t)RS_LDS_FLOATS * sizeof(float);
            note("k_final_stage_rs", nblk_rs, lds_bytes, 5, lv_gathers(5));
            if (f16) {
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_rs<__half, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL((k_final_stage_rs<__half, 5>), dim3(nblk_rs), dim3(RS_THREADS), lds_bytes, st, fa);
            } else {
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage_rs<float, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL((k_final_stage_rs<float, 5>), dim3(nblk_rs), dim3(RS_THREADS), lds_bytes, st, fa);
            }
#endif
        } else if (mlp_mode == MLP_F16X3 && Kmain == 5 && !(fa.dbg_bins || fa.dbg_w || fa.dbg_sigma || fa.dbg_xyz || fa.dbg_geo) && lt_enabled(cfg)) {
            // linear tail: layer 3 off the matrix cores (per-sample geometry features are not available in this form)
#define SN_LAUNCH_FINAL_LT_E(TT_, AUX_, EO_)                                                                                  \
            do {                                                                                                             \
                const size_t lds_bytes = (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float) + lds_pad_exp;     \
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<TT_, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5, AUX_, true, false, EO_>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                  \
                hipLaunchKernelGGL((k_final_stage<TT_, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5, AUX_, true, false, EO_>), dim3(nblk), dim3(256), lds_bytes, st, fa); \
            } while (0)
#define SN_LAUNCH_FINAL_LT(TT_, AUX_) do { if (eo) SN_LAUNCH_FINAL_LT_E(TT_, AUX_, true); else SN_LAUNCH_FINAL_LT_E(TT_, AUX_, false); } while (0)
            // SN_RENDER_L0=1 (opt-in): fp16 tables whose level 0 fits 16 KiB keep that level LDS-resident.  Bit-identical and measured
            // SLOWER than the texture path (800x800 [128]: 6.44 -> 6.58 ms, profiles/r03/ab_round3_experiments.txt): with fp16 tables the
            // kernel is not bound by the texture addressers, and the 8 ds_read_b32 + swizzled slab addressing cost more than 2 gathers save
            const uint64_t rows0 = (uint64_t)gl_main.res[0] * gl_main.res[0] * gl_main.res[0];
#ifdef SN_EXPERIMENTS
            const bool l0 = f16 && rows0 <= (uint64_t)L0_MAX_ROWS && cfg->tuning.experiment == SN_EXP_LDS_LEVEL0;
#else
            const bool l0 = false;
            (void)rows0;
#endif
#ifdef SN_EXPERIMENTS
#define SN_LAUNCH_FINAL_LT_L0(AUX_)                                                                                            \
            do {                                                                                                             \
                const size_t lds_bytes = (size_t)(PACK_FLOATS + 4 * 2 * 64 * 16 + L0_MAX_ROWS) * sizeof(float);              \
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<__half, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5, AUX_, true, true>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                  \
                hipLaunchKernelGGL((k_final_stage<__half, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5, AUX_, true, true>), dim3(nblk), dim3(256), lds_bytes, st, fa); \
            } while (0)
#endif
            FinalArgs fa7 = fa;
            const bool dens = Kv_main == 7 && !aux && !l0 && !final_sp &&
                              build_final_lv(gl_main, 7, pairs, cfg->grid.embeddings, 2u * (f16 ? 2u : 4u), fa7.lv);
            if (!l0) note(dens ? "k_final_stage<lt,K=7>" : aux ? "k_final_stage<lt,K=5,aux>" : "k_final_stage<lt,K=5>", nblk,
                          (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float), dens ? 7 : 5, lv_gathers(dens ? 7 : 5));
            if (dens) {     // levels 5 and 6 densified for this call (densify_levels)
                const size_t lds_bytes = (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float);
#define SN_LAUNCH_FINAL_K7_NP(TT_, EO_, NP_)                                                                                   \
                do {                                                                                                         \
                    SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<TT_, 16, 2, 64, 64, 16, 32, MLP_F16X3, 7, false, true, false, EO_, NP_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
                    hipLaunchKernelGGL((k_final_stage<TT_, 16, 2, 64, 64, 16, 32, MLP_F16X3, 7, false, true, false, EO_, NP_>), dim3(nblk), dim3(256), lds_bytes, st, fa7); \
                } while (0)
#define SN_LAUNCH_FINAL_K7(TT_, EO_) SN_LAUNCH_FINAL_K7_NP(TT_, EO_, 3)
                if (f16 && !eo && mlp_np == 1) SN_LAUNCH_FINAL_K7_NP(__half, false, 1);
                else if (f16) { if (eo) SN_LAUNCH_FINAL_K7(__half, true); else SN_LAUNCH_FINAL_K7(__half, false); }
                else { if (eo) SN_LAUNCH_FINAL_K7(float, true); else SN_LAUNCH_FINAL_K7(float, false); }
#undef SN_LAUNCH_FINAL_K7
#undef SN_LAUNCH_FINAL_K7_NP
            }
#ifdef SN_EXPERIMENTS
            else if (l0) { note("k_final_stage<lt,K=5,lds-level0>", nblk, (size_t)(PACK_FLOATS + 4 * 2 * 64 * 16 + L0_MAX_ROWS) * sizeof(float), 5, lv_gathers(5) - 2u);
                           if (aux) SN_LAUNCH_FINAL_LT_L0(true); else SN_LAUNCH_FINAL_LT_L0(false); }
#endif
            else if (aux) { if (f16) SN_LAUNCH_FINAL_LT(__half, true); else SN_LAUNCH_FINAL_LT(float, true); }
            else if (f16 && !eo && mlp_np < 3) {
                const size_t lds_bytes = (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float);
                SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_final_stage<__half, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5, false, true, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                hipLaunchKernelGGL((k_final_stage<__half, 16, 2, 64, 64, 16, 32, MLP_F16X3, 5, false, true, false, false, 1>), dim3(nblk), dim3(256), lds_bytes, st, fa);
            }
            else { if (f16) SN_LAUNCH_FINAL_LT(__half, false); else SN_LAUNCH_FINAL_LT(float, false); }
#ifdef SN_EXPERIMENTS
#undef SN_LAUNCH_FINAL_LT_L0
#endif
#undef SN_LAUNCH_FINAL_LT
#undef SN_LAUNCH_FINAL_LT_E
        } else if (mlp_mode == MLP_F16X3) {
            note(Kmain == 5 ? "k_final_stage<per-sample,K=5>" : "k_final_stage<per-sample,generic>", nblk, (size_t)(PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE) * sizeof(float),
                 Kmain, Kmain == 5 ? lv_gathers(5) : (uint32_t)dense_prefix(gl_main) * 4u + (16u - (uint32_t)dense_prefix(gl_main)) * 8u);
            if (Kmain == 5) SN_LAUNCH_FINAL_AUX(MLP_F16X3, 5, PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE);     // 72 KiB; main grid: levels 0-4 dense
            else SN_LAUNCH_FINAL_AUX(MLP_F16X3, -1, PACK_FLOATS + 4 * 2 * 64 * SLAB_STRIDE);
        } else if (mlp_mode == MLP_F32) { note("k_final_stage<mfma32>", nblk, (size_t)(PACK_FLOATS + 4 * 32 * 64) * sizeof(float), -1, 16u * 8u);
                                          SN_LAUNCH_FINAL(MLP_F32, -1, PACK_FLOATS + 4 * 32 * 64); }   // 64 KiB
        else { note("k_final_stage<valu>", nblk, (size_t)(2 * 64 * 256 + VIEW_W) * sizeof(float), -1, 16u * 8u);
               SN_LAUNCH_FINAL(MLP_VALU, -1, 2 * 64 * 256 + VIEW_W); }                                  // 136 KiB
#undef SN_LAUNCH_FINAL_AUX
#undef SN_LAUNCH_FINAL
#undef SN_LAUNCH_FINAL_T
        SN_LAUNCH_CHECK("k_final_stage");
        }
        if (cfg->with_feat) {
            FeatArgs ft;
            ft.rc = rc; ft.g = gl_feat; ft.table = cfg->feat_grid.embeddings; ft.T = cfg->num_steps[S - 1];
            ft.bins_in = b_scr[S - 1]; ft.bins0_stride = io->bins0_ray_stride;
            ft.bins0_tab = (S == 1 && io->bins0_table) ? io->bins0_table + (size_t)first * ft.bins0_stride : nullptr;
            ft.w_in = w_scr[S - 1];
            ft.out_stride = io->head_stride ? io->head_stride : gl_feat.L * gl_feat.C;
            ft.out = io->f_feat + (size_t)first * ft.out_stride;
            const bool h16 = cfg->feat_grid.table_dtype == SN_F16;
            ProfScope ps_feat(st, PK_FEAT);
            // dense prefix of the feature grid: its passes run the LDS-patch instantiation (k_feat_stage<..., PATCH = true>), the hashed rest the plain one
            uint32_t n_dense = 0;
            while (n_dense < gl_feat.L && (gl_feat.mode[n_dense] & 1u) == 0u) ++n_dense;
            const uint32_t lg_u = (uint32_t)feat_lg;
            const uint32_t patch_levels = cfg->tuning.feat_patch != 1 ? 0u : ((n_dense + lg_u - 1u) / lg_u) * lg_u > gl_feat.L ? gl_feat.L : ((n_dense + lg_u - 1u) / lg_u) * lg_u;
#define SN_LAUNCH_FEAT(CC, LGG)                                                                                       \
            do {                                                                                                      \
                if (patch_levels) {                                                                                   \
                    ft.level0 = 0;                                                                                    \
                    const dim3 fg(nblk, patch_levels / LGG);                                                          \
                    if (h16) hipLaunchKernelGGL((k_feat_stage<__half, CC, LGG, true>), fg, dim3(256), 0, st, ft);      \
                    else hipLaunchKernelGGL((k_feat_stage<float, CC, LGG, true>), fg, dim3(256), 0, st, ft);           \
                }                                                                                                     \
                if (patch_levels < gl_feat.L) {                                                                       \
                    ft.level0 = patch_levels;                                                                         \
                    const dim3 fg(nblk, (gl_feat.L - patch_levels) / LGG);                                            \
                    if (h16) hipLaunchKernelGGL((k_feat_stage<__half, CC, LGG>), fg, dim3(256), 0, st, ft);            \
                    else hipLaunchKernelGGL((k_feat_stage<float, CC, LGG>), fg, dim3(256), 0, st, ft);                 \
                }                                                                                                     \
            } while (0)
            if (gl_feat.C == 8) { if (feat_lg == 4) SN_LAUNCH_FEAT(8, 4); else if (feat_lg == 2) SN_LAUNCH_FEAT(8, 2); else SN_LAUNCH_FEAT(8, 1); }
            else if (gl_feat.C == 4) { if (feat_lg == 4) SN_LAUNCH_FEAT(4, 4); else if (feat_lg == 2) SN_LAUNCH_FEAT(4, 2); else SN_LAUNCH_FEAT(4, 1); }
            else { if (feat_lg == 4) SN_LAUNCH_FEAT(2, 4); else if (feat_lg == 2) SN_LAUNCH_FEAT(2, 2); else SN_LAUNCH_FEAT(2, 1); }
#undef SN_LAUNCH_FEAT
            SN_LAUNCH_CHECK("k_feat_stage");
        }
    }
    return SN_OK;
}

}  // extern "C"
