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

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float expf_det(float x) {
    // branch-free: the polynomial runs on the clamped argument and the three special cases are selected at the end
    // (early returns compile to exec-mask branches that serialise the three exps of a march step)
    const float xc = fminf(fmaxf(x, -104.0f), 89.0f);
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
    const int ki = (int)k;
    const int k1 = ki / 2;
    const int k2 = ki - k1;
    const float s1 = __int_as_float((k1 + 127) << 23);
    const float s2 = __int_as_float((k2 + 127) << 23);
    float y = (p * s1) * s2;
    y = x < -103.97208404541015625f ? 0.0f : y;
    y = x > 88.72283935546875f ? __builtin_inff() : y;
    return x != x ? x : y;
}

template <typename T> __device__ __forceinline__ float table_ld(const T *p);
template <> __device__ __forceinline__ float table_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float table_ld<__half>(const __half *p) { return __half2float(*p); }

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

}  // namespace sn
