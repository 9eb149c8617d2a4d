// sn_common.h — shared host/device helpers of libsanerf_hip.so (gfx950 only).
//
// Numerics contract: the library is compiled with -ffp-contract=off and
// every fused multiply-add is an explicit fmaf, exp is sn::expf_det (not the libm/ocml
// one), division and sqrt are IEEE-rounded, prefix sums that feed sample indices
// accumulate in fp64.  The same recipe is followed by the CPU oracle, which is what makes
// integer outputs (sample indices) bit-comparable.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sanerf_hip.h"

namespace sn {

// ---- host side error plumbing -------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define SN_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::sn::set_error(__VA_ARGS__);     \
            return SN_ERR_INVALID;            \
        }                                     \
    } while (0)

#define SN_HIP_OK(call)                                   \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return ::sn::hip_fail(e__, #call); \
    } while (0)

#define SN_LAUNCH_CHECK(name)                                        \
    do {                                                             \
        hipError_t e__ = hipGetLastError();                          \
        if (e__ != hipSuccess) return ::sn::hip_fail(e__, name);     \
    } while (0)

static inline uint32_t div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

// ---- per-level table of a multiresolution grid, passed to kernels by value -----------------
// Built on the host from (offsets, S, H): resolution follows gridencoder.cu:133 in fp32,
// dense-vs-hash follows the stride walk of gridencoder.cu:66-76.
struct GridLevels {
    uint32_t res[SN_MAX_LEVELS];   // kernel-side resolution
    uint32_t size[SN_MAX_LEVELS];  // rows in the level (hashmap_size)
    uint32_t off[SN_MAX_LEVELS];   // first row of the level
    uint32_t mode[SN_MAX_LEVELS];  // bit0: hashed; bits1-2: modulo kind (0 none,1 pow2 mask,2 generic); bits4-7: dims in the dense walk
    uint32_t L, D, C;
    uint32_t gridtype, align_corners, interp;
};

int build_grid_levels(GridLevels *g, const int32_t *offsets_host, uint32_t D, uint32_t C, uint32_t L,
                      float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp);
uint32_t level_resolution(uint32_t level, float S, uint32_t H);

// ---- debug build -DSN_POISON_LDS (`make poison`): every kernel fills the workgroup's whole LDS allocation (static + dynamic, the size is
// read from the dispatch packet) with signalling NaNs before its first statement, so that a read-before-write -- invisible when the
// previous workgroup happened to leave finite values behind, e.g. a stale row multiplied by a zero weight (the round-3 bug of
// k_final_stage_any) -- turns into NaN in the result.  The fuzzers and the GPU tests are run under this build once per round.
#ifdef SN_POISON_LDS
__device__ __forceinline__ void poison_lds_all() {
    typedef __attribute__((address_space(4))) const uint32_t cst_u32;
    const uint32_t bytes = ((cst_u32 *)__builtin_amdgcn_dispatch_ptr())[7];      // hsa_kernel_dispatch_packet_t::group_segment_size (byte 28)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32 *base = (lds_u32 *)0;
    const uint32_t nthreads = blockDim.x * blockDim.y * blockDim.z, tid = (threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x;
    for (uint32_t i = tid; i < bytes / 4u; i += nthreads) base[i] = 0x7fa00000u;
    __syncthreads();
}
#define SN_POISON_ALL() ::sn::poison_lds_all()
#else
#define SN_POISON_ALL() do { } while (0)
#endif

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float expf_det(float x) {
    // branch-free: the polynomial runs on the clamped argument; scaling is ONE v_ldexp_f32, which also produces the
    // overflow (+inf from 89 -> k = 128) and underflow (0 from -104 -> k = -150) cases; NaN is selected at the end.
    // Same recipe as the oracle's orc_expf (oracle/oracle.c).
    const float xc = __builtin_amdgcn_fmed3f(x, -104.0f, 89.0f);
    const float k = __builtin_rintf(xc * 1.44269502162933349609375f);
    float r = __builtin_fmaf(k, -0.693145751953125f, xc);
    r = __builtin_fmaf(k, -1.42860676533018704503775e-06f, r);
    float p = 1.98756915e-4f;
    p = __builtin_fmaf(p, r, 1.39819995e-3f);
    p = __builtin_fmaf(p, r, 8.33345205e-3f);
    p = __builtin_fmaf(p, r, 4.16657962e-2f);
    p = __builtin_fmaf(p, r, 1.66666657e-1f);
    p = __builtin_fmaf(p, r, 5.00000000e-1f);
    const float r2 = r * r;
    p = __builtin_fmaf(p, r2, r);
    p = p + 1.0f;
    const float y = __builtin_amdgcn_ldexpf(p, (int)k);
    return x != x ? x : y;
}

template <typename T> __device__ __forceinline__ float table_ld(const T *p);
template <> __device__ __forceinline__ float table_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float table_ld<__half>(const __half *p) { return __half2float(*p); }

// One table row (C features) with the widest loads its size allows.  Rows are C*sizeof(T) bytes and the table
// base is at least 16-byte aligned (checked on the host, table_aligned()), so a row of 8/16/32+ bytes is naturally
// aligned for dwordx2/dwordx4: 1-2 vector loads instead of C scalar ones per corner.
template <typename T, int C>
__device__ __forceinline__ void load_row(const T *__restrict__ row, float (&v)[C]) {
    constexpr int BYTES = C * (int)sizeof(T);
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
        for (int q = 0; q < BYTES / 16; ++q) {
            const uint4 t = reinterpret_cast<const uint4 *>(row)[q];
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (sizeof(T) == 4) v[q * 4 + i] = __uint_as_float(w[i]);
                else {
                    const __half2 h = *reinterpret_cast<const __half2 *>(&w[i]);
                    v[q * 8 + 2 * i] = __low2float(h); v[q * 8 + 2 * i + 1] = __high2float(h);
                }
            }
        }
    } else if constexpr (BYTES == 8) {
        const uint2 t = *reinterpret_cast<const uint2 *>(row);
        if constexpr (sizeof(T) == 4) { v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); }
        else {
            const __half2 a = *reinterpret_cast<const __half2 *>(&t.x), b = *reinterpret_cast<const __half2 *>(&t.y);
            v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
        }
    } else if constexpr (BYTES == 4 && sizeof(T) == 2) {
        const __half2 a = *reinterpret_cast<const __half2 *>(row);
        v[0] = __low2float(a); v[1] = __high2float(a);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = table_ld<T>(row + c);
    }
}

static inline bool table_aligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// gridencoder.cu:45-79 — row of a grid vertex inside its level.
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(const uint32_t (&p)[D], uint32_t res, uint32_t size, uint32_t mode) {
    uint32_t idx;
    if (mode & 1u) {
        constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
        idx = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) idx ^= p[d] * primes[d];
    } else {
        const uint32_t nd = (mode >> 4) & 15u;
        idx = 0;
        uint32_t stride = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if (d < nd) { idx += p[d] * stride; stride *= res; }
        }
    }
    const uint32_t mk = (mode >> 1) & 3u;
    if (mk == 1u) idx &= (size - 1u);
    else if (mk == 2u) idx %= size;
    return idx;
}

// gridencoder.cu:137-159 — cell + fractional position of x01 inside a level.
template <uint32_t D>
__device__ __forceinline__ void grid_locate(const float (&x01)[D], uint32_t res, bool align_corners, uint32_t interp,
                                            float (&pos)[D], float (&deriv)[D], uint32_t (&cell)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        float p;
        if (align_corners) {
            p = x01[d] * (float)(res - 1);
            cell[d] = umin((uint32_t)floorf(p), res - 2);
        } else {
            p = __builtin_fmaf(x01[d], (float)res, -0.5f);
            p = fminf(fmaxf(p, 0.0f), (float)(res - 1));
            cell[d] = (uint32_t)floorf(p);
        }
        p -= (float)cell[d];
        if (interp == 1u) {
            deriv[d] = 6 * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            deriv[d] = 1.0f;
        }
        pos[d] = p;
    }
}

// nerf/renderer.py:60-69
__device__ __forceinline__ void contract3(float &x, float &y, float &z) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    float mag = ax; int idx = 0;
    if (ay > mag) { mag = ay; idx = 1; }
    if (az > mag) { mag = az; idx = 2; }
    if (ax != ax || ay != ay || az != az) mag = __builtin_nanf("");
    if (mag < 1.0f) return;
    const float inv = 1.0f / mag;
    const float big = (2.0f - inv) / mag;
    x = x * (idx == 0 ? big : inv);
    y = y * (idx == 1 ? big : inv);
    z = z * (idx == 2 ? big : inv);
}

// nerf/renderer.py:249-252
__device__ __forceinline__ float spacing_fn(float x) { return x < 1.0f ? x / 2.0f : 1.0f - 1.0f / (2.0f * x); }
__device__ __forceinline__ float spacing_inv(float x) { return x < 0.5f ? 2.0f * x : 1.0f / (2.0f - 2.0f * x); }

// nerf/renderer.py:122-139 for one ray
__device__ __forceinline__ void near_far_one(const float (&o)[3], const float (&d)[3], const float (&aabb)[6],
                                             float min_near, float &near, float &far) {
    near = -__builtin_inff(); far = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float den = d[k] + 1e-15f;
        const float tmin = (aabb[k] - o[k]) / den;
        const float tmax = (aabb[3 + k] - o[k]) / den;
        const float lo = tmin < tmax ? tmin : tmax;
        const float hi = tmin > tmax ? tmin : tmax;
        near = (lo > near || lo != lo) ? lo : near;
        far = (hi < far || hi != hi) ? hi : far;
    }
    if (far < near) { near = 1e9f; far = 1e9f; }
    if (near < min_near) near = min_near;
}

// torch.linspace value i of `steps` (scalar aten recipe: two roundings, no fma)
__device__ __forceinline__ float linspace_at(float start, float end, float step, uint32_t steps, uint32_t i) {
    if (i < steps / 2) { const float m = step * (float)i; return start + m; }
    const float m = step * (float)(steps - i - 1);
    return end - m;
}

// ---- fast-path level addressing shared by the fused kernels (render.hip, heads.hip) ---------------
// Byte offsets (from the level's base) of the 8 corners of one cell.
// KIND: 0 = dense level, 1 = hashed level (compile-time, from the kernel's dense-prefix length K),
//      -1 = decided at run time by a wave-uniform select (generic instantiation).
// No per-level branch either way: a branch is a basic-block boundary per level and pins the march
// to 8 gathers in flight per lane.  The row stride in bytes is folded into the (wave-uniform)
// multipliers: ((x ^ y*P1 ^ z*P2) & m) * s == ((x*s) ^ (y*P1*s) ^ (z*P2*s)) & (m*s) for s a power of two.
// Fast-path assumptions checked on the host (levels_fast): hashed levels have a power-of-two size,
// dense levels index all three dimensions and need no modulo, align_corners = False, linear interp.
template <int KIND, uint32_t STRIDE_BYTES>
__device__ __forceinline__ void corner_offsets(const uint32_t (&cell)[3], uint32_t res, uint32_t size, uint32_t mode,
                                               uint32_t (&offs)[8]) {
    static_assert((STRIDE_BYTES & (STRIDE_BYTES - 1)) == 0, "row stride must be a power of two");
    const bool hashed = KIND == 1 || (KIND == -1 && (mode & 1u) != 0u);
    const uint32_t my = (hashed ? 2654435761u : res) * STRIDE_BYTES;          // gridencoder.cu:49 primes / :66-70 strides
    const uint32_t mz = (hashed ? 805459861u : res * res) * STRIDE_BYTES;
    const uint32_t mask = hashed ? (size - 1u) * STRIDE_BYTES : 0xffffffffu;
    const uint32_t x0 = cell[0], y0 = cell[1], z0 = cell[2];
    const uint32_t top = res - 1u;
    // the +1 neighbour is clamped to res-1 (gridencoder.cu:182): its term is the base term plus one multiplier,
    // or the base term itself at the border -- an add and a select instead of a second quarter-rate v_mul_lo_u32
    uint32_t Y0, Z0;
    if constexpr (KIND == 0) { Y0 = __umul24(y0, my); Z0 = __umul24(z0, mz); }   // full rate; operands bounded by levels_fast()
    else if constexpr (KIND == 1) {
        // only the bits under the mask survive and levels_fast() bounds mask < 2^24: the low 24 bits of (y * P * stride)
        // depend on the low 24 bits of the factors only -> full-rate v_mul_u32_u24 instead of the quarter-rate v_mul_lo_u32
        Y0 = __umul24(y0, my & 0xffffffu); Z0 = __umul24(z0, mz & 0xffffffu);
    } else { Y0 = y0 * my; Z0 = z0 * mz; }
    const uint32_t X0 = x0 * STRIDE_BYTES;
    const uint32_t X1 = x0 < top ? X0 + STRIDE_BYTES : X0;
    const uint32_t Y1 = y0 < top ? Y0 + (KIND == 1 ? (my & 0xffffffu) : my) : Y0;
    const uint32_t Z1 = z0 < top ? Z0 + (KIND == 1 ? (mz & 0xffffffu) : mz) : Z0;
    if constexpr (KIND == 1) {
        // (X ^ Y ^ Z) & m == (X & m) ^ (Y & m) ^ (Z & m): masking the 6 partial terms replaces 8 per-corner ANDs
        const uint32_t X0m = X0 & mask, X1m = X1 & mask, Y0m = Y0 & mask, Y1m = Y1 & mask, Z0m = Z0 & mask, Z1m = Z1 & mask;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) offs[i] = ((i & 1u) ? X1m : X0m) ^ ((i & 2u) ? Y1m : Y0m) ^ ((i & 4u) ? Z1m : Z0m);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            const uint32_t X = (i & 1u) ? X1 : X0, Y = (i & 2u) ? Y1 : Y0, Z = (i & 4u) ? Z1 : Z0;
            if constexpr (KIND == 0) offs[i] = X + Y + Z;
            else offs[i] = (hashed ? (X ^ Y ^ Z) : (X + Y + Z)) & mask;
        }
    }
}

// gridencoder.cu:145-149 for align_corners = False, linear interpolation (what the fused kernels support)
__device__ __forceinline__ void locate_linear(const float (&x01)[3], uint32_t res, float (&pos)[3], uint32_t (&cell)[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = __builtin_fmaf(x01[d], (float)res, -0.5f);
        p = __builtin_amdgcn_fmed3f(p, 0.0f, (float)(res - 1u));      // = min(max(p, 0), res-1) in one instruction (NaN -> 0 like fmaxf)
        cell[d] = (uint32_t)p;                     // p >= 0: truncation == floor (one v_cvt_u32_f32)
        pos[d] = __builtin_amdgcn_fractf(p);       // p - floor(p), exact for p >= 0 (one v_fract_f32)
    }
}

// fused kernels assume: hashed levels have power-of-two size; dense levels walk all 3 dims, no modulo
static inline bool levels_fast(const GridLevels &g) {
    for (uint32_t l = 0; l < g.L; ++l) {
        const uint32_t mode = g.mode[l], mk = (mode >> 1) & 3u, nd = (mode >> 4) & 15u;
        if (mode & 1u) { if (mk != 1u || (uint64_t)g.size[l] * g.C * 4u > (1u << 24) || g.res[l] >= (1u << 24)) return false; }   // 24-bit hash multiplies
        else if (mk != 0u || nd != 3u) return false;
        else if ((uint64_t)g.res[l] * g.res[l] * 16u >= (1u << 24)) return false;   // dense strides go through 24-bit multiplies
    }
    return g.align_corners == 0 && g.interp == 0;
}


extern int g_bin_pull;        // grid_binned.hip (entry form of the binned grid backward; set by sn_debug_set in experiments builds)

}  // namespace sn
