// raymarch.hip — stand-alone ray-marching operators for gfx950.
//
// The reference has no native twin for these (its README mentions a `raymarching`
// extension that is not in the tree); each kernel replaces a block of torch ops in
// nerf/utils.py:get_rays and nerf/renderer.py (near_far_from_aabb, contract, sample_pdf,
// the sigma->weights scan and the weighted sums).  One lane per ray; tensors keep the
// reference's [N, T] row-major layout so they interoperate with torch code on either side.
// The fused renderer (render.hip) inlines the same arithmetic with a [T, N] scratch layout.
#include "sn_common.h"
#include "sh_basis.inc"

#include <float.h>

namespace sn {

struct Pose { float m[16]; };
struct Aabb { float v[6]; };

// nerf/utils.py:201-205, 269-287
__global__ __launch_bounds__(256) void k_generate_rays(Pose pose, float fx, float fy, float cx, float cy,
                                                       uint32_t W, uint32_t first, uint32_t count,
                                                       float *__restrict__ rays_o, float *__restrict__ rays_d) {
    SN_POISON_ALL();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t n = first + t;
    const uint32_t row = n / W, col = n - row * W;
    const float i = (float)col + 0.5f, j = (float)row + 0.5f;
    const float xs = (i - cx) / fx;
    const float ys = -(j - cy) / fy;
    const float zs = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = xs * pose.m[k * 4 + 0];
        acc = __builtin_fmaf(ys, pose.m[k * 4 + 1], acc);
        acc = __builtin_fmaf(zs, pose.m[k * 4 + 2], acc);
        rays_d[(size_t)t * 3 + k] = acc;
        rays_o[(size_t)t * 3 + k] = pose.m[k * 4 + 3];
    }
}

// nerf/utils.py:209-287 for a drawn subset of pixels: ray n looks through flat pixel index inds[n] (row-major, H x W) of
// camera poses[n] (or the one shared camera), intrinsics likewise -- the per-ray cameras of provider.py:908-913
// (`random_image_batch`: `poses = self.poses[index]` with one image index per ray).  Same arithmetic as k_generate_rays.
__global__ __launch_bounds__(256) void k_rays_from_pixels(const float *__restrict__ poses, uint32_t pose_stride,
                                                          const float *__restrict__ intr, uint32_t intr_stride,
                                                          const int64_t *__restrict__ inds, uint32_t W, uint32_t N,
                                                          float *__restrict__ rays_o, float *__restrict__ rays_d) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *m = poses + (size_t)n * pose_stride;        // 4x4 row-major cam2world
    const float *k4 = intr + (size_t)n * intr_stride;        // fx, fy, cx, cy
    const int64_t ind = inds[n];
    const uint32_t row = (uint32_t)(ind / W), col = (uint32_t)(ind - (int64_t)row * W);
    const float i = (float)col + 0.5f, j = (float)row + 0.5f;
    const float xs = (i - k4[2]) / k4[0];
    const float ys = -(j - k4[3]) / k4[1];
    const float zs = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = xs * m[k * 4 + 0];
        acc = __builtin_fmaf(ys, m[k * 4 + 1], acc);
        acc = __builtin_fmaf(zs, m[k * 4 + 2], acc);
        rays_d[(size_t)n * 3 + k] = acc;
        rays_o[(size_t)n * 3 + k] = m[k * 4 + 3];
    }
}

__global__ __launch_bounds__(256) void k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                  Aabb ab, float min_near, uint32_t N,
                                                  float *__restrict__ nears, float *__restrict__ fars) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float o[3] = {rays_o[(size_t)n * 3], rays_o[(size_t)n * 3 + 1], rays_o[(size_t)n * 3 + 2]};
    const float d[3] = {rays_d[(size_t)n * 3], rays_d[(size_t)n * 3 + 1], rays_d[(size_t)n * 3 + 2]};
    float near, far;
    near_far_one(o, d, ab.v, min_near, near, far);
    nears[n] = near; fars[n] = far;
}

__global__ __launch_bounds__(256) void k_contract(const float *__restrict__ x, uint32_t N, float *__restrict__ z) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float a = x[(size_t)n * 3], b = x[(size_t)n * 3 + 1], c = x[(size_t)n * 3 + 2];
    contract3(a, b, c);
    z[(size_t)n * 3] = a; z[(size_t)n * 3 + 1] = b; z[(size_t)n * 3 + 2] = c;
}

// device numerics primitives exposed element-wise for the parity tests (sn_debug_eval)
__global__ __launch_bounds__(256) void k_debug_eval(int op, const float *__restrict__ a, const float *__restrict__ b, uint32_t n,
                                                    float *__restrict__ y) {
    SN_POISON_ALL();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], w = b ? b[i] : 0.0f;
    float r;
    switch (op) {
        case 0: r = expf_det(x); break;
        case 1: r = x / w; break;                       // IEEE-rounded division as compiled (-fhip-fp32-correctly-rounded-divide-sqrt)
        case 2: r = spacing_fn(x); break;
        case 3: r = spacing_inv(x); break;
        default: r = 0.0f;
    }
    y[i] = r;
}

__device__ __forceinline__ float nan_to_num(float v) {
    if (v != v) return 0.0f;
    if (v == __builtin_inff()) return FLT_MAX;
    if (v == -__builtin_inff()) return -FLT_MAX;
    return v;
}

constexpr uint32_t RM_WAVE_PER_RAY_MAX = 65536;     // rays up to which sample_pdf / weights_from_sigma run one wave per ray

// nerf/renderer.py:84-119.  cdf and u are both non-decreasing, so searchsorted(right=True)
// is one merge pass; the cdf is a running fp64 sum rounded to fp32 per prefix (= torch.cumsum
// on CPU) and the normaliser is the fp64-accumulated sum.
__global__ __launch_bounds__(256) void k_sample_pdf(const float *__restrict__ bins, const float *__restrict__ weights,
                                                    uint32_t N, uint32_t T0, uint32_t T, const float *__restrict__ u,
                                                    uint32_t u_stride, float *__restrict__ out_bins, int32_t *__restrict__ inds) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *w = weights + (size_t)n * T0;
    const float *b = bins + (size_t)n * (T0 + 1);
    double acc = 0;
    for (uint32_t i = 0; i < T0; ++i) acc += (double)(w[i] + 0.01f);
    const float wsum = (float)acc;
    const float ustart = (float)(0.5 / T), uend = (float)(1 - 0.5 / T);
    const float ustep = T > 1 ? (uend - ustart) / (float)(T - 1) : 0.0f;

    uint32_t i = 0;          // next cdf index to test
    acc = 0;
    float c_prev = 0.0f;     // cdf[i-1]
    float c_cur = 0.0f;      // cdf[i]   (cdf[0] = 0)
    float b_prev = b[0], b_cur = b[0];
    for (uint32_t j = 0; j < T; ++j) {
        const float uj = u ? u[(size_t)n * u_stride + j] : linspace_at(ustart, uend, ustep, T, j);
        while (i <= T0 && c_cur <= uj) {   // advance: cdf[i] <= u
            c_prev = c_cur; b_prev = b_cur;
            ++i;
            if (i <= T0) {
                const float pdf = (w[i - 1] + 0.01f) / wsum;
                acc += (double)pdf;
                const float c = (float)acc;
                c_cur = c > 1.0f ? 1.0f : c;
                b_cur = b[i];
            }
        }
        // ind = i ; below = clamp(i-1, 0, T0) ; above = clamp(i, 0, T0)
        float c0, c1, b0, b1;
        if (i == 0) { c0 = c_cur; b0 = b_cur; c1 = c_cur; b1 = b_cur; }
        else if (i > T0) { c0 = c_prev; b0 = b_prev; c1 = c_prev; b1 = b_prev; }
        else { c0 = c_prev; b0 = b_prev; c1 = c_cur; b1 = b_cur; }
        float t = nan_to_num((uj - c0) / (c1 - c0));
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        const float m = t * (b1 - b0);
        out_bins[(size_t)n * T + j] = b0 + m;
        if (inds) inds[(size_t)n * T + j] = (int32_t)i;
    }
}

// nerf/renderer.py:308-325
__global__ __launch_bounds__(256) void k_weights(const float *__restrict__ real_bins, const float *__restrict__ sigmas,
                                                 uint32_t N, uint32_t T, int last_opaque, float *__restrict__ weights) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *rb = real_bins + (size_t)n * (T + 1);
    const float *sg = sigmas + (size_t)n * T;
    double cum = 0;
    float prev = rb[0];
    for (uint32_t j = 0; j < T; ++j) {
        const float next = rb[j + 1];
        const float delta = next - prev;
        prev = next;
        float ds = delta * sg[j];
        if (last_opaque && j == T - 1) ds = __builtin_inff();
        const float alpha = 1.0f - expf_det(-ds);
        const float tr = expf_det(-(float)cum);
        float w = alpha * tr;
        if (w != w) w = 0.0f;
        weights[(size_t)n * T + j] = w;
        cum += (double)ds;
    }
}

// The same two operators with one WAVE per ray, for the few thousand rays of a training step (one lane per ray is 16
// workgroups of serial T-step loops: 60-130 us per call).  Loads, exponentials, divisions and the output search are
// spread over the lanes; the fp64 running sums keep the sequential order of the kernels above (every lane adds the
// same values in the same order), so the results are bit-identical.
__global__ __launch_bounds__(256) void k_weights_wave(const float *__restrict__ real_bins, const float *__restrict__ sigmas,
                                                      uint32_t N, uint32_t T, int last_opaque, float *__restrict__ weights) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (n >= N) return;
    const float *rb = real_bins + (size_t)n * (T + 1);
    const float *sg = sigmas + (size_t)n * T;
    double cum = 0.0;
    for (uint32_t j0 = 0; j0 < T; j0 += 64u) {
        const uint32_t j = j0 + lane;
        const bool live = j < T;
        float ds = live ? (rb[j + 1] - rb[j]) * sg[j] : 0.0f;
        if (live && last_opaque && j == T - 1u) ds = __builtin_inff();
        const float add = (live && ds != __builtin_inff()) ? ds : 0.0f;      // +inf is the last sample: nothing follows it
        double excl = 0.0;
#pragma unroll
        for (uint32_t q = 0; q < 64u; ++q) {
            const float d = __shfl(add, q);
            if (lane == q) excl = cum;
            cum += (double)d;
        }
        if (live) {
            const float alpha = 1.0f - expf_det(-ds);
            const float tr = expf_det(-(float)excl);
            float w = alpha * tr;
            if (w != w) w = 0.0f;
            weights[(size_t)n * T + j] = w;
        }
    }
}

__global__ __launch_bounds__(256) void k_sample_pdf_wave(const float *__restrict__ bins, const float *__restrict__ weights,
                                                         uint32_t N, uint32_t T0, uint32_t T, const float *__restrict__ u,
                                                         uint32_t u_stride, float *__restrict__ out_bins, int32_t *__restrict__ inds) {
    SN_POISON_ALL();
    extern __shared__ float pdf_lds[];                   // per wave: cdf[T0+1] | bins[T0+1]
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_raw = blockIdx.x * 4u + wave;
    const uint32_t n = n_raw < N ? n_raw : N - 1u;       // spare waves redo the last ray (no early exit: plain loops below)
    float *cdf = pdf_lds + (size_t)wave * 2u * (T0 + 1u), *bl = cdf + (T0 + 1u);
    const float *w = weights + (size_t)n * T0;
    const float *b = bins + (size_t)n * (T0 + 1);
    for (uint32_t i = lane; i < T0; i += 64u) cdf[i + 1u] = w[i] + 0.01f;
    for (uint32_t i = lane; i <= T0; i += 64u) bl[i] = b[i];
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    for (uint32_t i = 0; i < T0; ++i) acc += (double)cdf[i + 1u];                 // every lane, same order
    const float wsum = (float)acc;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < T0; i += 64u) cdf[i + 1u] = cdf[i + 1u] / wsum;    // pdf
    __builtin_amdgcn_wave_barrier();
    acc = 0.0;
    for (uint32_t i = 0; i < T0; ++i) {                                           // running fp64 sum, rounded per prefix
        acc += (double)cdf[i + 1u];
        const float c = (float)acc;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0u) cdf[i + 1u] = c > 1.0f ? 1.0f : c;
    }
    if (lane == 0u) cdf[0] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    if (n_raw >= N) return;
    const float ustart = (float)(0.5 / T), uend = (float)(1 - 0.5 / T);
    const float ustep = T > 1 ? (uend - ustart) / (float)(T - 1) : 0.0f;
    for (uint32_t j = lane; j < T; j += 64u) {
        const float uj = u ? u[(size_t)n * u_stride + j] : linspace_at(ustart, uend, ustep, T, j);
        uint32_t lo = 0u, hi = T0 + 1u;                                           // searchsorted(cdf, u, right=True)
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1u; else hi = mid;
        }
        const uint32_t i = lo;
        float c0, c1, b0, b1;
        if (i == 0u) { c0 = c1 = cdf[0]; b0 = b1 = bl[0]; }
        else if (i > T0) { c0 = c1 = cdf[T0]; b0 = b1 = bl[T0]; }
        else { c0 = cdf[i - 1u]; c1 = cdf[i]; b0 = bl[i - 1u]; b1 = bl[i]; }
        float t = nan_to_num((uj - c0) / (c1 - c0));
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        const float m = t * (b1 - b0);
        out_bins[(size_t)n * T + j] = b0 + m;
        if (inds) inds[(size_t)n * T + j] = (int32_t)i;
    }
}

// Backward of k_weights w.r.t. sigma (the bin edges carry no gradient: sample_pdf's output is not differentiated and
// the encoders have no input gradient).  w_j = (1 - e^{-ds_j}) T_j with T_j = e^{-sum_{i<j} ds_i}, ds_j = delta_j sigma_j:
//   dL/dds_i = g_i e^{-ds_i} T_i - sum_{j>i} g_j w_j,   dL/dsigma_i = delta_i dL/dds_i,
// the opaque last sample (ds = inf) is a constant (torch: cat([ds[:-1], inf])), NaN weights pass no gradient
// (nan_to_num).  One wave per ray: lane l owns samples l, l+64, ...; the prefix of ds (fp64) and the suffix of g*w are
// wave scans carried from one 64-sample segment to the next.
// T <= 256 (LONG = false): the per-lane terms of all four segments stay in registers between the sweeps.  LONG: any T -- the forward sweep keeps only
// the fp64 prefix at the start of every segment (LDS, one double per segment and wave), the backward sweep evaluates a segment's terms again
// from it: the same operations on the same values, so both instantiations give the same bits.
template <bool LONG>
__global__ __launch_bounds__(256) void k_weights_backward(const float *__restrict__ real_bins, const float *__restrict__ sigmas,
                                                          const float *__restrict__ grad_w, uint32_t N, uint32_t T, int last_opaque,
                                                          float *__restrict__ grad_sigmas) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(8))) double wb_lds[];      // LONG: [4 waves][segments] prefix of ds at the segment's start
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (n >= N) return;                                                 // (no workgroup barrier below)
    const float *rb = real_bins + (size_t)n * (T + 1);
    const float *sg = sigmas + (size_t)n * T;
    const float *gw = grad_w + (size_t)n * T;
    float *gs = grad_sigmas + (size_t)n * T;
    constexpr uint32_t MAXSEG = 4;                      // !LONG: T <= 256; loops fully unrolled so the per-lane arrays stay in registers
    const uint32_t nseg = (T + 63u) / 64u;
    // terms of segment s given the prefix of ds before it; returns the segment's sum of ds (the scan's last lane)
    auto segment = [&](uint32_t s, double carry, float &gwv, float &dterm, float &delta) -> double {
        const uint32_t j = s * 64u + lane;
        const bool live = j < T;
        const float d = live ? rb[j + 1] - rb[j] : 0.0f;
        float ds = live ? d * sg[j] : 0.0f;
        const bool opaque = last_opaque && j == T - 1u;
        double incl = (live && !opaque) ? (double)ds : 0.0;      // the opaque sample is last: nothing comes after it
#pragma unroll
        for (uint32_t k = 1; k < 64u; k <<= 1) { const double v = __shfl_up(incl, k); if (lane >= k) incl += v; }
        const double excl = carry + incl - ((live && !opaque) ? (double)ds : 0.0);
        const double total = __shfl(incl, 63);
        if (opaque) ds = __builtin_inff();
        const float tr = expf_det(-(float)excl);
        const float e = expf_det(-ds);
        const float w = (1.0f - e) * tr;
        const bool bad = w != w;
        const float g = live ? gw[j] : 0.0f;
        gwv = (live && !bad) ? g * w : 0.0f;
        dterm = (live && !bad && !opaque) ? g * e * tr : 0.0f;
        delta = (live && !opaque) ? d : 0.0f;
        return total;
    };
    // suffix of g*w inside a segment + the later segments' sum; writes the segment's gradients
    auto finish = [&](uint32_t s, float gwv, float dterm, float delta, float &tail) {
        float incl = gwv;
#pragma unroll
        for (uint32_t k = 1; k < 64u; k <<= 1) { const float v = __shfl_down(incl, k); if (lane + k < 64u) incl += v; }
        const float after = tail + incl - gwv;          // strictly later samples
        tail += __shfl(incl, 0);
        const uint32_t j = s * 64u + lane;
        if (j < T) gs[j] = delta * (dterm - after);
    };
    if constexpr (!LONG) {
        float gwv[MAXSEG], dterm[MAXSEG], delta[MAXSEG];
        double carry = 0.0;
#pragma unroll
        for (uint32_t s = 0; s < MAXSEG; ++s) {
            gwv[s] = dterm[s] = delta[s] = 0.0f;
            if (s >= nseg) continue;
            carry += segment(s, carry, gwv[s], dterm[s], delta[s]);
        }
        float tail = 0.0f;                              // sum over later segments
#pragma unroll
        for (int s = (int)MAXSEG - 1; s >= 0; --s) {
            if ((uint32_t)s >= nseg) continue;
            finish((uint32_t)s, gwv[s], dterm[s], delta[s], tail);
        }
    } else {
        double *pre = wb_lds + (size_t)(threadIdx.x >> 6) * nseg;
        double carry = 0.0;
        for (uint32_t s = 0; s < nseg; ++s) {
            float a, b, c;
            if (lane == 0u) pre[s] = carry;
            carry += segment(s, carry, a, b, c);
        }
        __builtin_amdgcn_wave_barrier();
        float tail = 0.0f;
        for (uint32_t s = nseg; s-- > 0u;) {
            float gwv, dterm, delta;
            segment(s, pre[s], gwv, dterm, delta);
            finish(s, gwv, dterm, delta, tail);
        }
    }
}

// renderer.py:277-285 for one stage, no gradient anywhere on this chain: bins -> real_bins, mid-points, positions
// (contracted if asked).  Same arithmetic as the fused renderer's real_bin / sample position.
__global__ __launch_bounds__(256) void k_sample_positions(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                          const float *__restrict__ nears, const float *__restrict__ fars,
                                                          const float *__restrict__ bins, uint32_t N, uint32_t T, int contract, float grid_bound,
                                                          float *__restrict__ real_bins, float *__restrict__ rays_t, float *__restrict__ xyzs) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)N * (T + 1)) return;
    const uint32_t n = (uint32_t)(t / (T + 1)), j = (uint32_t)(t - (uint64_t)n * (T + 1));
    const float s_near = spacing_fn(nears[n]), s_far = spacing_fn(fars[n]);
    auto rbin = [&](float b) { const float a = s_near * (1.0f - b); const float c = s_far * b; return spacing_inv(a + c); };
    const float r0 = rbin(bins[t]);
    real_bins[t] = r0;
    if (j == T) return;
    const float r1 = rbin(bins[t + 1]);
    const float tmid = (r1 + r0) / 2.0f;
    const size_t o = (size_t)n * T + j;
    rays_t[o] = tmid;
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float m = rays_d[(size_t)n * 3 + k] * tmid; p[k] = rays_o[(size_t)n * 3 + k] + m; }
    if (contract) contract3(p[0], p[1], p[2]);
    if (grid_bound > 0.0f) {                     // gridencoder/grid.py:156: (x + bound) / (2 * bound), the encoder's own first step
        const float two_b = 2.0f * grid_bound;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = (p[k] + grid_bound) / two_b;
    }
    xyzs[o * 3 + 0] = p[0]; xyzs[o * 3 + 1] = p[1]; xyzs[o * 3 + 2] = p[2];
}

// Training-time jitter of a stage's sampling positions from ONE uniform random tensor (the reference draws torch.rand_like per stage):
//   kind 0  renderer.py:262-270   out[n,i] = clamp(linspace(0, 1, T)[i] + (r[n,i] - 0.5) / (T - 1), 0, 1)      stage-0 bins, T = num_steps[0] + 1
//   kind 1  renderer.py:97-102    out[n,i] = linspace(0.5/T, 1 - 0.5/T, T)[i] + (r[n,i] - 0.5) / T              sample_pdf's u
// r == NULL: no jitter (perturb=False): the plain linspace rows.
__global__ __launch_bounds__(256) void k_jitter(const float *__restrict__ r, uint32_t N, uint32_t T, int kind, float *__restrict__ out) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)N * T) return;
    const uint32_t i = (uint32_t)(t % T);
    const float u = r ? r[t] - 0.5f : 0.0f;
    if (kind == 0) {
        const float step = T > 1u ? 1.0f / (float)(T - 1u) : 0.0f;
        float v = linspace_at(0.0f, 1.0f, step, T, i);
        if (r) { v = v + u / (float)(T - 1u); v = fminf(fmaxf(v, 0.0f), 1.0f); }
        out[t] = v;
    } else {
        const float lo = (float)(0.5 / T), hi = (float)(1 - 0.5 / T);        // Python doubles rounded once, as torch.linspace receives them
        const float step = T > 1u ? (hi - lo) / (float)(T - 1u) : 0.0f;
        float v = linspace_at(lo, hi, step, T, i);
        if (r) v = v + u / (float)T;
        out[t] = v;
    }
}

// Per-ray head of the training path (renderer.py:327-347 + network.py:164-170) without per-sample colour tensors:
//   weights_sum = sum_t w;  depth = sum_t w t_mid;  f_image[0:15] = sum_t w raw[t, 1:16];  f_image[15:31] = SH4(d / |d|) * weights_sum
// (the direction is constant along a ray, so sum_t w_t SH(d) = SH(d) sum_t w_t -- the reference evaluates SH per sample, renderer.py:293-295).
// raw [N,T,16] is grid_mlp's output ([sigma_raw | geo_feat]); 16 lanes per ray, lane c owns channel c (lane 0: weights_sum and depth).
__global__ __launch_bounds__(256) void k_ray_composite(const float *__restrict__ weights, const float *__restrict__ rays_t, const float *__restrict__ raw,
                                                       const float *__restrict__ rays_d, uint32_t N, uint32_t T, float *__restrict__ wsum,
                                                       float *__restrict__ depth, float *__restrict__ f_image) {
    SN_POISON_ALL();
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, n_raw = gid >> 4, c = gid & 15u;
    const uint32_t n = n_raw < N ? n_raw : N - 1u;
    const float *w = weights + (size_t)n * T, *tm = rays_t + (size_t)n * T, *rw = raw + (size_t)n * T * 16u + c;
    float acc = 0.0f, dsum = 0.0f;
    if (c == 0u) {
        for (uint32_t j = 0; j < T; ++j) { const float wj = w[j]; acc += wj; dsum = __builtin_fmaf(wj, tm[j], dsum); }
    } else {
        for (uint32_t j = 0; j < T; ++j) acc = __builtin_fmaf(w[j], rw[(size_t)j * 16u], acc);
    }
    const float ws = __shfl(acc, (int)(threadIdx.x & 48u), 64);            // lane 0 of the ray's 16-lane group
    float x = rays_d[(size_t)n * 3], y = rays_d[(size_t)n * 3 + 1], z = rays_d[(size_t)n * 3 + 2];
    const float inv = 1.0f / sqrtf(x * x + y * y + z * z);                 // renderer.py:294
    x *= inv; y *= inv; z *= inv;
    float sh[16];
    {
        const unsigned C = 4u;
        SN_SH_POWERS
        (void)x4; (void)x5; (void)x6; (void)x7; (void)y4; (void)y5; (void)y6; (void)y7; (void)z4; (void)z5; (void)z6; (void)z7;
        SN_SH_VALUES(sh);
    }
    float mine = sh[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mine = c == (uint32_t)k ? sh[k] : mine;
    if (n_raw >= N) return;
    float *f = f_image + (size_t)n * 31u;
    if (c == 0u) { wsum[n] = acc; depth[n] = dsum; }
    else f[c - 1u] = acc;
    if (c < 16u) f[15u + c] = mine * ws;
}

// Backward of k_ray_composite, one thread per sample:
//   d/d w[n,t]      = sum_c g_f[c] raw[n,t,1+c] + (g_wsum + sum_k g_f[15+k] SH_k) + g_depth t_mid[n,t]
//   d/d raw[n,t,1+c] = w[n,t] g_f[c];   d/d raw[n,t,0] = 0
__global__ __launch_bounds__(256) void k_ray_composite_backward(const float *__restrict__ weights, const float *__restrict__ rays_t,
                                                                const float *__restrict__ raw, const float *__restrict__ rays_d,
                                                                const float *__restrict__ g_wsum, const float *__restrict__ g_depth,
                                                                const float *__restrict__ g_f, uint32_t N, uint32_t T,
                                                                float *__restrict__ g_weights, float *__restrict__ g_raw) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint64_t)N * T) return;
    const uint32_t n = (uint32_t)(t / T);
    float x = rays_d[(size_t)n * 3], y = rays_d[(size_t)n * 3 + 1], z = rays_d[(size_t)n * 3 + 2];
    const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
    x *= inv; y *= inv; z *= inv;
    float sh[16];
    {
        const unsigned C = 4u;
        SN_SH_POWERS
        (void)x4; (void)x5; (void)x6; (void)x7; (void)y4; (void)y5; (void)y6; (void)y7; (void)z4; (void)z5; (void)z6; (void)z7;
        SN_SH_VALUES(sh);
    }
    const float *gf = g_f ? g_f + (size_t)n * 31u : nullptr;
    float k_ray = g_wsum ? g_wsum[n] : 0.0f;
    if (gf) {
#pragma unroll
        for (int k = 0; k < 16; ++k) k_ray = __builtin_fmaf(gf[15 + k], sh[k], k_ray);
    }
    const float4 *rp = reinterpret_cast<const float4 *>(raw + t * 16u);
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    const float rv[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
    const float w = weights[t];
    float gw = k_ray;
    float go[16];
    go[0] = 0.0f;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        const float g = gf ? gf[c] : 0.0f;
        gw = __builtin_fmaf(g, rv[1 + c], gw);
        go[1 + c] = w * g;
    }
    if (g_depth) gw = __builtin_fmaf(g_depth[n], rays_t[t], gw);
    g_weights[t] = gw;
    float4 *op = reinterpret_cast<float4 *>(g_raw + t * 16u);
    op[0] = float4{go[0], go[1], go[2], go[3]}; op[1] = float4{go[4], go[5], go[6], go[7]};
    op[2] = float4{go[8], go[9], go[10], go[11]}; op[3] = float4{go[12], go[13], go[14], go[15]};
}

__global__ __launch_bounds__(256) void k_zero16(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) p[i] = uint4{0u, 0u, 0u, 0u};
}

// Inter-level proposal loss of one proposal stage against the final stage (nerf/renderer.py:30-57), one wave per ray:
//   cum = [0, cumsum(w)];  lo_j = clamp(searchsorted(b[:-1], rb_j, right) - 1, 0, T-1);  hi_j = clamp(searchsorted(b[1:], rb_{j+1}, right), 0, T-1)
//   bound_j = cum[hi_j + 1] - cum[lo_j];   term_j = max(rw_j - bound_j, 0)^2 / (rw_j + 1e-8)
// forward: loss_ray[n] = sum_j term_j (the caller divides the grand total by N*Tr = torch's .mean());
// backward: bound_j = sum of w_i over lo_j <= i <= hi_j and lo, hi are non-decreasing in j (rb is sorted), so the js that
// contain a given i form one contiguous range [ja, jb]:  dL/dw_i = G[jb+1] - G[ja] with G the prefix sum of
// g_j = -2 max(rw_j - bound_j, 0) / (rw_j + 1e-8)  -- two binary searches per i, no atomics (deterministic).
// The reference's own proposal bins/weights of the final stage (rb, rw) are detached there too.
// LONG (T or Tr beyond what four waves' LDS holds): the per-wave arrays live in a workspace in memory and a wave walks rays n, n + waves, ...;
// the same operations in the same order, so both instantiations give the same bits.
template <bool BACKWARD, bool LONG>
__global__ __launch_bounds__(256) void k_proposal_loss(const float *__restrict__ bins, const float *__restrict__ weights,
                                                       const float *__restrict__ ref_bins, const float *__restrict__ ref_w,
                                                       uint32_t N, uint32_t T, uint32_t Tr, float scale, const float *__restrict__ scale_dev,
                                                       float *__restrict__ out, double *__restrict__ workspace) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(8))) double pl_lds[];
    const float sc = scale_dev ? scale * scale_dev[0] : scale;      // (scale 1 and no device factor: the plain value / gradient, exact)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // lanes of a wave exchange values through the arrays: LDS operations of a wave execute in order; in memory (LONG) the writes are fenced first
    auto wsync = [&]() { if constexpr (LONG) __threadfence_block(); __builtin_amdgcn_wave_barrier(); };
  for (uint32_t n_raw = blockIdx.x * 4u + wave; LONG ? n_raw < N : true; n_raw += gridDim.x * 4u) {
    const uint32_t n = n_raw < N ? n_raw : N - 1u;           // spare waves redo the last ray and store nothing
    // per wave: cum[T+1] (fp64: bound = cum[hi+1] - cum[lo] cancels), G[Tr+1] (fp64, backward), then the fp32 / int arrays
    const uint32_t nd = (T + 1u) + (BACKWARD ? Tr + 1u : 0u);
    const uint32_t nf = (T + 1u) + (BACKWARD ? 3u * Tr : 0u);
    const uint32_t per_wave_d = nd + (nf + 1u) / 2u;         // in doubles
    double *cum = (LONG ? workspace + (size_t)blockIdx.x * 4u * per_wave_d : pl_lds) + (size_t)wave * per_wave_d, *G = cum + (T + 1u);
    float *b = reinterpret_cast<float *>(cum + nd);
    float *g = b + (T + 1u);
    int32_t *lo_s = reinterpret_cast<int32_t *>(g + Tr), *hi_s = lo_s + Tr;
    const float *w = weights + (size_t)n * T;
    const float *rb = ref_bins + (size_t)n * (Tr + 1);
    const float *rw = ref_w + (size_t)n * Tr;
    for (uint32_t i = lane; i <= T; i += 64u) b[i] = bins[(size_t)n * (T + 1) + i];
    for (uint32_t i = lane; i < T; i += 64u) cum[i + 1u] = (double)w[i];
    wsync();
    {
        double acc = 0.0;
        for (uint32_t i = 0; i < T; ++i) {                   // every lane, same order; lane 0 writes the prefixes back
            acc += cum[i + 1u];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0u) cum[i + 1u] = acc;
        }
        if (lane == 0u) cum[0] = 0.0;
    }
    wsync();
    auto upper = [&](const float *a, uint32_t len, float v) {   // number of a[0..len) <= v
        uint32_t lo = 0u, hi = len;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1u; else hi = mid; }
        return lo;
    };
    float part = 0.0f;
    for (uint32_t j = lane; j < Tr; j += 64u) {
        int32_t lo = (int32_t)upper(b, T, rb[j]) - 1;
        lo = lo < 0 ? 0 : (lo > (int32_t)T - 1 ? (int32_t)T - 1 : lo);
        int32_t hi = (int32_t)upper(b + 1, T, rb[j + 1u]);
        hi = hi > (int32_t)T - 1 ? (int32_t)T - 1 : hi;
        const float r = rw[j];
        const float diff = (float)((double)r - (cum[hi + 1] - cum[lo]));
        const float d = diff > 0.0f ? diff : 0.0f;
        const float den = r + 1e-8f;
        if constexpr (BACKWARD) { g[j] = -2.0f * d / den; lo_s[j] = lo; hi_s[j] = hi; }
        else part += d * d / den;
    }
    if constexpr (!BACKWARD) {
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) part += __shfl_xor(part, k);
        if (lane == 0u && n_raw < N) out[n] = part * sc;
    } else {
        wsync();
        double acc = 0.0;
        for (uint32_t j = 0; j < Tr; ++j) {
            if (lane == 0u) G[j] = acc;
            acc += (double)g[j];
        }
        if (lane == 0u) G[Tr] = acc;
        wsync();
        if (n_raw >= N) return;
        for (uint32_t i = lane; i < T; i += 64u) {
            uint32_t a0 = 0u, a1 = Tr;                       // ja = first j with hi_j >= i
            while (a0 < a1) { const uint32_t mid = (a0 + a1) >> 1; if (hi_s[mid] >= (int32_t)i) a1 = mid; else a0 = mid + 1u; }
            uint32_t c0 = 0u, c1 = Tr;                       // jb + 1 = number of j with lo_j <= i
            while (c0 < c1) { const uint32_t mid = (c0 + c1) >> 1; if (lo_s[mid] <= (int32_t)i) c0 = mid + 1u; else c1 = mid; }
            out[(size_t)n * T + i] = (c0 > a0 ? (float)(G[c0] - G[a0]) : 0.0f) * sc;
        }
    }
    if constexpr (!LONG) break;
    wsync();                                                 // LONG: the next ray reuses the arrays
  }
}

// Distortion loss of Mip-NeRF 360 on the final stage's normalised bins (nerf/renderer.py:17-27; the reference calls the
// third-party torch_efficient_distloss.eff_distloss(w, m, interval), whose value is
//   mean_rays[ (1/3) sum_i w_i^2 d_i + 2 sum_{i>j} w_i w_j (m_i - m_j) ]   with m = interval mid-points, d = lengths,
// i.e. sum_ij w_i w_j |m_i - m_j| for sorted bins).  One wave per ray, lane k owns samples k, k+64, ...: the inner sum
// S_k = sum_j w_j |m_k - m_j| is evaluated directly (T terms per sample, no prefix-sum cancellation), which gives the
// value and the gradient at once:  loss_ray = (1/3) sum_k w_k^2 d_k + sum_k w_k S_k,  dloss_ray/dw_k = (2/3) w_k d_k + 2 S_k.
// LONG (T beyond what four waves' LDS holds): weights and mid-points come from memory in the inner loop (wave-uniform addresses) -- the same
// arithmetic in the same order, so both instantiations give the same bits.
template <bool LONG>
__global__ __launch_bounds__(256) void k_distort_loss(const float *__restrict__ bins, const float *__restrict__ weights, uint32_t N,
                                                      uint32_t T, float *__restrict__ loss_per_ray, float *__restrict__ grad_w) {
    SN_POISON_ALL();
    extern __shared__ float dl_lds[];                    // per wave: w[T] | m[T]   (LONG: unused)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_raw = blockIdx.x * 4u + wave;
    const uint32_t n = n_raw < N ? n_raw : N - 1u;
    float *w = dl_lds + (size_t)wave * 2u * T, *m = w + T;
    const float *b = bins + (size_t)n * (T + 1);
    const float *wg = weights + (size_t)n * T;
    if constexpr (!LONG) {
        for (uint32_t i = lane; i < T; i += 64u) {
            const float b0 = b[i], d = b[i + 1u] - b0;
            w[i] = wg[i];
            m[i] = b0 + d / 2.0f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    float part = 0.0f;
    for (uint32_t k = lane; k < T; k += 64u) {
        const float dk = b[k + 1u] - b[k];
        const float mk = LONG ? b[k] + dk / 2.0f : m[k], wk = LONG ? wg[k] : w[k];
        float S = 0.0f;
        if constexpr (LONG) {
            float bj = b[0];
            for (uint32_t j = 0; j < T; ++j) {
                const float bn = b[j + 1u];
                const float mj = bj + (bn - bj) / 2.0f;
                S = __builtin_fmaf(wg[j], fabsf(mk - mj), S);
                bj = bn;
            }
        } else {
            for (uint32_t j = 0; j < T; ++j) S = __builtin_fmaf(w[j], fabsf(mk - m[j]), S);
        }
        part += wk * wk * dk / 3.0f + wk * S;
        if (n_raw < N) grad_w[(size_t)n * T + k] = 2.0f * wk * dk / 3.0f + 2.0f * S;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) part += __shfl_xor(part, k);
    if (lane == 0u && n_raw < N) loss_per_ray[n] = part;
}

// out[n,k] = sum_t w[n,t] * v[n,t,k], sequential fmaf over t (renderer.py:333-338,361,384)
__global__ __launch_bounds__(256) void k_composite(const float *__restrict__ weights, const float *__restrict__ values,
                                                   uint32_t N, uint32_t T, uint32_t K, float *__restrict__ out) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)N * K) return;
    const uint32_t n = (uint32_t)(t / K), k = (uint32_t)(t - (uint64_t)n * K);
    const float *w = weights + (size_t)n * T;
    const float *v = values + (size_t)n * T * K + k;
    float acc = 0;
    for (uint32_t j = 0; j < T; ++j) acc = __builtin_fmaf(w[j], v[(size_t)j * K], acc);
    out[t] = acc;
}

__global__ __launch_bounds__(256) void k_composite_backward(const float *__restrict__ weights, const float *__restrict__ grad_out,
                                                            uint32_t N, uint32_t T, uint32_t K, float *__restrict__ grad_values) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)N * T * K) return;
    const uint32_t k = (uint32_t)(t % K);
    const uint64_t nt = t / K;
    const uint32_t n = (uint32_t)(nt / T);
    grad_values[t] = weights[nt] * grad_out[(size_t)n * K + k];
}


// Mask-field NLL of one ray (nerf/trainer.py:419-428): p = softmax(logits); c = clamp(p[label], eps, 1 - eps); loss = -log(c).
// One lane per ray, K <= 32 logits in registers; value and d loss / d logits in one pass (the clamp passes no gradient where it binds, as
// torch.clamp's backward).  Replaces softmax -> clamp -> gather -> log -> neg and their five backward kernels.
__global__ __launch_bounds__(256) void k_mask_nll(const float *__restrict__ logits, const int64_t *__restrict__ labels, uint32_t N, uint32_t K, float eps,
                                                  float *__restrict__ loss, float *__restrict__ grad_logits) {
    SN_POISON_ALL();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *row = logits + (size_t)n * K;
    float mx = row[0];
    for (uint32_t k = 1; k < K; ++k) mx = fmaxf(mx, row[k]);
    float sum = 0.0f;
    for (uint32_t k = 0; k < K; ++k) sum += expf(row[k] - mx);           // torch's softmax: exp(x - max) / sum
    const int64_t y = labels[n];
    const bool valid = y >= 0 && y < (int64_t)K;
    const float py = valid ? expf(row[(uint32_t)y] - mx) / sum : 1.0f;
    const float c = fminf(fmaxf(py, eps), 1.0f - eps);
    loss[n] = valid ? -logf(c) : 0.0f;
    if (grad_logits) {
        const bool pass = valid && py >= eps && py <= 1.0f - eps;         // clamp backward: gradient where min <= x <= max
        const float g = pass ? -1.0f / c : 0.0f;                           // d loss / d p_y
        for (uint32_t k = 0; k < K; ++k) {
            const float pk = expf(row[k] - mx) / sum;
            grad_logits[(size_t)n * K + k] = g * py * ((k == (uint32_t)y ? 1.0f : 0.0f) - pk);    // softmax backward for a one-hot upstream
        }
    }
}

}  // namespace sn

using namespace sn;

extern "C" {

int sn_rm_generate_rays(const float *pose_host, float fx, float fy, float cx, float cy,
                        uint32_t H, uint32_t W, uint32_t row_begin, uint32_t row_end,
                        float *rays_o, float *rays_d, sn_stream_t stream) {
    SN_REQUIRE(pose_host && rays_o && rays_d, "generate_rays: NULL pointer");
    SN_REQUIRE(row_begin <= row_end && row_end <= H, "generate_rays: rows [%u,%u) outside image height %u", row_begin, row_end, H);
    Pose p;
    for (int i = 0; i < 16; ++i) p.m[i] = pose_host[i];
    const uint32_t first = row_begin * W, count = (row_end - row_begin) * W;
    if (count == 0) return SN_OK;
    hipLaunchKernelGGL(k_generate_rays, dim3(div_up(count, 256)), dim3(256), 0, (hipStream_t)stream, p, fx, fy, cx, cy, W, first, count, rays_o, rays_d);
    SN_LAUNCH_CHECK("k_generate_rays");
    return SN_OK;
}

int sn_rm_rays_from_pixels(const float *poses, uint32_t n_poses, const float *intrinsics, uint32_t n_intrinsics, const int64_t *inds,
                           uint32_t W, uint32_t N, float *rays_o, float *rays_d, sn_stream_t stream) {
    SN_REQUIRE(poses && intrinsics && inds && rays_o && rays_d, "rays_from_pixels: NULL pointer");
    SN_REQUIRE((n_poses == 1 || n_poses == N) && (n_intrinsics == 1 || n_intrinsics == N),
               "rays_from_pixels: %u poses / %u intrinsics for %u rays (each must be 1 or N)", n_poses, n_intrinsics, N);
    SN_REQUIRE(W >= 1, "rays_from_pixels: W must be >= 1");
    if (N == 0) return SN_OK;
    hipLaunchKernelGGL(k_rays_from_pixels, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, poses, n_poses == 1 ? 0u : 16u,
                       intrinsics, n_intrinsics == 1 ? 0u : 4u, inds, W, N, rays_o, rays_d);
    SN_LAUNCH_CHECK("k_rays_from_pixels");
    return SN_OK;
}

int sn_rm_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb_host,
                             float min_near, uint32_t N, float *nears, float *fars, sn_stream_t stream) {
    SN_REQUIRE(rays_o && rays_d && aabb_host && nears && fars, "near_far_from_aabb: NULL pointer");
    Aabb ab;
    for (int i = 0; i < 6; ++i) ab.v[i] = aabb_host[i];
    if (N == 0) return SN_OK;
    hipLaunchKernelGGL(k_near_far, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, ab, min_near, N, nears, fars);
    SN_LAUNCH_CHECK("k_near_far");
    return SN_OK;
}

int sn_rm_contract(const float *x, uint32_t N, float *z, sn_stream_t stream) {
    SN_REQUIRE(x && z, "contract: NULL pointer");
    if (N == 0) return SN_OK;
    hipLaunchKernelGGL(k_contract, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, x, N, z);
    SN_LAUNCH_CHECK("k_contract");
    return SN_OK;
}

int sn_debug_eval(int op, const float *a, const float *b, uint32_t n, float *y, sn_stream_t stream) {
    SN_REQUIRE(a && y && op >= 0 && op <= 3, "debug_eval: bad arguments");
    SN_REQUIRE(b || op != 1, "debug_eval: op %d needs two operands", op);
    if (n == 0) return SN_OK;
    hipLaunchKernelGGL(k_debug_eval, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, n, y);
    SN_LAUNCH_CHECK("k_debug_eval");
    return SN_OK;
}

int sn_rm_sample_pdf(const float *bins, const float *weights, uint32_t N, uint32_t T0, uint32_t T,
                     const float *u, uint32_t u_stride, float *out_bins, int32_t *inds, sn_stream_t stream) {
    SN_REQUIRE(bins && weights && out_bins, "sample_pdf: NULL pointer");
    SN_REQUIRE(T0 >= 1 && T >= 1, "sample_pdf: T0=%u T=%u must be >= 1", T0, T);
    SN_REQUIRE(u_stride == 0 || u_stride == T, "sample_pdf: u_stride must be 0 (shared table) or T (per-ray rows)");
    if (N == 0) return SN_OK;
    const size_t wave_lds = (size_t)4 * 2 * (T0 + 1) * sizeof(float);
    if (N <= RM_WAVE_PER_RAY_MAX && wave_lds <= 64 * 1024)       // few rays: one wave per ray (bit-identical)
        hipLaunchKernelGGL(k_sample_pdf_wave, dim3(div_up(N, 4)), dim3(256), wave_lds, (hipStream_t)stream, bins, weights, N, T0, T, u, u_stride, out_bins, inds);
    else
        hipLaunchKernelGGL(k_sample_pdf, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, bins, weights, N, T0, T, u, u_stride, out_bins, inds);
    SN_LAUNCH_CHECK("k_sample_pdf");
    return SN_OK;
}

int sn_rm_weights_from_sigma(const float *real_bins, const float *sigmas, uint32_t N, uint32_t T,
                             int last_sample_opaque, float *weights, sn_stream_t stream) {
    SN_REQUIRE(real_bins && sigmas && weights, "weights_from_sigma: NULL pointer");
    if (N == 0 || T == 0) return SN_OK;
    if (N <= RM_WAVE_PER_RAY_MAX)
        hipLaunchKernelGGL(k_weights_wave, dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, real_bins, sigmas, N, T, last_sample_opaque, weights);
    else
        hipLaunchKernelGGL(k_weights, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, real_bins, sigmas, N, T, last_sample_opaque, weights);
    SN_LAUNCH_CHECK("k_weights");
    return SN_OK;
}

int sn_rm_weights_from_sigma_backward(const float *real_bins, const float *sigmas, const float *grad_weights, uint32_t N, uint32_t T,
                                      int last_sample_opaque, float *grad_sigmas, sn_stream_t stream) {
    SN_REQUIRE(real_bins && sigmas && grad_weights && grad_sigmas, "weights_from_sigma_backward: NULL pointer");
    if (N == 0 || T == 0) return SN_OK;
    SN_REQUIRE(T <= 131072u, "weights_from_sigma_backward: at most 131072 samples per ray (got %u)", T);
    if (T <= 256u)           // the per-lane terms of the four segments in registers
        hipLaunchKernelGGL(k_weights_backward<false>, dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, real_bins, sigmas, grad_weights, N, T,
                           last_sample_opaque, grad_sigmas);
    else                     // any length: segment prefixes in LDS, the terms evaluated again on the way back (same bits)
        hipLaunchKernelGGL(k_weights_backward<true>, dim3(div_up(N, 4)), dim3(256), (size_t)4 * div_up(T, 64) * sizeof(double), (hipStream_t)stream,
                           real_bins, sigmas, grad_weights, N, T, last_sample_opaque, grad_sigmas);
    SN_LAUNCH_CHECK("k_weights_backward");
    return SN_OK;
}

int sn_rm_sample_positions_ex(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *bins,
                              uint32_t N, uint32_t T, int contract, float grid_bound, float *real_bins, float *rays_t, float *xyzs, sn_stream_t stream) {
    SN_REQUIRE(rays_o && rays_d && nears && fars && bins && real_bins && rays_t && xyzs, "sample_positions: NULL pointer");
    SN_REQUIRE(grid_bound >= 0.0f, "sample_positions: grid_bound must be >= 0 (0 = positions as they are)");
    if (N == 0) return SN_OK;
    hipLaunchKernelGGL(k_sample_positions, dim3(div_up((uint64_t)N * (T + 1), 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, nears, fars,
                       bins, N, T, contract, grid_bound, real_bins, rays_t, xyzs);
    SN_LAUNCH_CHECK("k_sample_positions");
    return SN_OK;
}

int sn_rm_sample_positions(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *bins,
                           uint32_t N, uint32_t T, int contract, float *real_bins, float *rays_t, float *xyzs, sn_stream_t stream) {
    return sn_rm_sample_positions_ex(rays_o, rays_d, nears, fars, bins, N, T, contract, 0.0f, real_bins, rays_t, xyzs, stream);
}

int sn_rm_jitter(const float *uniform, uint32_t N, uint32_t T, int kind, float *out, sn_stream_t stream) {
    if (N == 0) return SN_OK;
    SN_REQUIRE(out, "jitter: NULL pointer");
    SN_REQUIRE(kind == 0 || kind == 1, "jitter: kind 0 (stage-0 bins) or 1 (sample_pdf u), got %d", kind);
    SN_REQUIRE(T >= 1, "jitter: T >= 1");
    if (N == 0) return SN_OK;
    hipLaunchKernelGGL(k_jitter, dim3(div_up((uint64_t)N * T, 256)), dim3(256), 0, (hipStream_t)stream, uniform, N, T, kind, out);
    SN_LAUNCH_CHECK("k_jitter");
    return SN_OK;
}

int sn_rm_ray_composite(const float *weights, const float *rays_t, const float *raw, const float *rays_d, uint32_t N, uint32_t T,
                        float *weights_sum, float *depth, float *f_image, sn_stream_t stream) {
    if (N == 0) return SN_OK;
    SN_REQUIRE(weights && rays_t && raw && rays_d && weights_sum && depth && f_image, "ray_composite: NULL pointer");
    hipLaunchKernelGGL(k_ray_composite, dim3(div_up((uint64_t)N * 16u, 256)), dim3(256), 0, (hipStream_t)stream, weights, rays_t, raw, rays_d, N, T,
                       weights_sum, depth, f_image);
    SN_LAUNCH_CHECK("k_ray_composite");
    return SN_OK;
}

int sn_rm_ray_composite_backward(const float *weights, const float *rays_t, const float *raw, const float *rays_d, const float *grad_weights_sum,
                                 const float *grad_depth, const float *grad_f_image, uint32_t N, uint32_t T, float *grad_weights, float *grad_raw,
                                 sn_stream_t stream) {
    if (N == 0 || T == 0) return SN_OK;
    SN_REQUIRE(weights && rays_t && raw && rays_d && grad_weights && grad_raw, "ray_composite_backward: NULL pointer");
    SN_REQUIRE(table_aligned(raw) && table_aligned(grad_raw), "ray_composite_backward: raw / grad_raw must be 16-byte aligned");
    if (N == 0 || T == 0) return SN_OK;
    hipLaunchKernelGGL(k_ray_composite_backward, dim3(div_up((uint64_t)N * T, 256)), dim3(256), 0, (hipStream_t)stream, weights, rays_t, raw, rays_d,
                       grad_weights_sum, grad_depth, grad_f_image, N, T, grad_weights, grad_raw);
    SN_LAUNCH_CHECK("k_ray_composite_backward");
    return SN_OK;
}

int sn_zero(void *ptr, size_t bytes, sn_stream_t stream) {
    SN_REQUIRE(ptr || bytes == 0, "zero: NULL pointer");
    SN_REQUIRE(table_aligned(ptr) && bytes % 16u == 0, "zero: pointer and size must be multiples of 16 bytes");
    if (bytes == 0) return SN_OK;
    const size_t n16 = bytes / 16u;
    uint32_t blocks = div_up(n16, 256 * 4);
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(k_zero16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint4 *>(ptr), n16);
    SN_LAUNCH_CHECK("k_zero16");
    return SN_OK;
}

// bytes one wave's arrays take (mirrors the carve-up in k_proposal_loss)
static size_t proposal_wave_bytes(uint32_t T, uint32_t Tr, bool backward) {
    const size_t nd = (size_t)(T + 1) + (backward ? (size_t)Tr + 1 : 0), nf = (size_t)(T + 1) + (backward ? 3 * (size_t)Tr : 0);
    return (nd + (nf + 1) / 2) * sizeof(double);
}
// workgroups of the long-ray launch: every wave of it owns a set of arrays in the workspace; at most 64 MiB of it, at least one workgroup
static uint32_t proposal_long_blocks(uint32_t N, uint32_t T, uint32_t Tr, bool backward) {
    const size_t per_block = 4 * proposal_wave_bytes(T, Tr, backward);
    size_t blocks = ((size_t)64 << 20) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    const uint32_t need = div_up(N, 4);
    return need < blocks ? need : (uint32_t)blocks;
}

size_t sn_rm_proposal_loss_workspace_bytes(uint32_t N, uint32_t T, uint32_t Tr, int backward) {
    if (N == 0 || (T <= 512u && Tr <= 512u)) return 0;
    return (size_t)proposal_long_blocks(N, T, Tr, backward != 0) * 4 * proposal_wave_bytes(T, Tr, backward != 0);
}

int sn_rm_proposal_loss_long(const float *bins, const float *weights, const float *ref_bins, const float *ref_weights, uint32_t N, uint32_t T,
                             uint32_t Tr, float scale, const float *scale_dev, float *loss_per_ray, float *grad_weights, void *workspace,
                             size_t workspace_bytes, sn_stream_t stream) {
    SN_REQUIRE(bins && weights && ref_bins && ref_weights, "proposal_loss: NULL pointer");
    SN_REQUIRE((loss_per_ray != nullptr) != (grad_weights != nullptr), "proposal_loss: pass exactly one of loss_per_ray (forward) / grad_weights (backward)");
    SN_REQUIRE(T >= 1 && Tr >= 1 && T <= (1u << 24) && Tr <= (1u << 24), "proposal_loss: 1..2^24 samples per ray (got %u, %u)", T, Tr);
    if (N == 0) return SN_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool backward = grad_weights != nullptr;
    if (T <= 512u && Tr <= 512u) {               // a ray's arrays in LDS, four rays per workgroup
        const size_t lds = 4 * proposal_wave_bytes(T, Tr, backward);
        if (!backward) hipLaunchKernelGGL((k_proposal_loss<false, false>), dim3(div_up(N, 4)), dim3(256), lds, st, bins, weights, ref_bins, ref_weights, N, T, Tr, scale, scale_dev, loss_per_ray, (double *)nullptr);
        else hipLaunchKernelGGL((k_proposal_loss<true, false>), dim3(div_up(N, 4)), dim3(256), lds, st, bins, weights, ref_bins, ref_weights, N, T, Tr, scale, scale_dev, grad_weights, (double *)nullptr);
    } else {                                     // any length: the arrays in the caller's workspace, waves walk the rays
        const size_t need = sn_rm_proposal_loss_workspace_bytes(N, T, Tr, backward);
        SN_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 7u) == 0,
                   "proposal_loss: %u / %u samples per ray need an 8-byte aligned workspace of %zu bytes (sn_rm_proposal_loss_workspace_bytes), got %zu", T, Tr, need, workspace_bytes);
        const uint32_t blocks = proposal_long_blocks(N, T, Tr, backward);
        if (!backward) hipLaunchKernelGGL((k_proposal_loss<false, true>), dim3(blocks), dim3(256), 0, st, bins, weights, ref_bins, ref_weights, N, T, Tr, scale, scale_dev, loss_per_ray, (double *)workspace);
        else hipLaunchKernelGGL((k_proposal_loss<true, true>), dim3(blocks), dim3(256), 0, st, bins, weights, ref_bins, ref_weights, N, T, Tr, scale, scale_dev, grad_weights, (double *)workspace);
    }
    SN_LAUNCH_CHECK("k_proposal_loss");
    return SN_OK;
}

int sn_rm_proposal_loss_scaled(const float *bins, const float *weights, const float *ref_bins, const float *ref_weights, uint32_t N, uint32_t T,
                               uint32_t Tr, float scale, const float *scale_dev, float *loss_per_ray, float *grad_weights, sn_stream_t stream) {
    SN_REQUIRE(T <= 512 && Tr <= 512, "proposal_loss: more than 512 samples per ray (got %u, %u) take a workspace: sn_rm_proposal_loss_long", T, Tr);
    return sn_rm_proposal_loss_long(bins, weights, ref_bins, ref_weights, N, T, Tr, scale, scale_dev, loss_per_ray, grad_weights, nullptr, 0, stream);
}

int sn_rm_proposal_loss(const float *bins, const float *weights, const float *ref_bins, const float *ref_weights, uint32_t N, uint32_t T,
                        uint32_t Tr, float *loss_per_ray, float *grad_weights, sn_stream_t stream) {
    return sn_rm_proposal_loss_scaled(bins, weights, ref_bins, ref_weights, N, T, Tr, 1.0f, nullptr, loss_per_ray, grad_weights, stream);
}

int sn_rm_distort_loss(const float *bins, const float *weights, uint32_t N, uint32_t T, float *loss_per_ray, float *grad_weights,
                       sn_stream_t stream) {
    SN_REQUIRE(bins && weights && loss_per_ray && grad_weights, "distort_loss: NULL pointer");
    SN_REQUIRE(T >= 1, "distort_loss: at least one sample per ray (got %u)", T);
    if (N == 0) return SN_OK;
    if (T <= 2048u)          // a ray's weights and mid-points in LDS (four waves: 64 KiB at T = 2048)
        hipLaunchKernelGGL(k_distort_loss<false>, dim3(div_up(N, 4)), dim3(256), (size_t)4 * 2 * T * sizeof(float), (hipStream_t)stream, bins, weights, N, T,
                           loss_per_ray, grad_weights);
    else                     // any length: the inner loop reads them from memory (same arithmetic, same bits)
        hipLaunchKernelGGL(k_distort_loss<true>, dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, bins, weights, N, T, loss_per_ray, grad_weights);
    SN_LAUNCH_CHECK("k_distort_loss");
    return SN_OK;
}

int sn_rm_composite(const float *weights, const float *values, uint32_t N, uint32_t T, uint32_t K,
                    float *out, sn_stream_t stream) {
    SN_REQUIRE(weights && values && out, "composite: NULL pointer");
    if (N == 0 || K == 0) return SN_OK;
    hipLaunchKernelGGL(k_composite, dim3(div_up((uint64_t)N * K, 256)), dim3(256), 0, (hipStream_t)stream, weights, values, N, T, K, out);
    SN_LAUNCH_CHECK("k_composite");
    return SN_OK;
}

int sn_rm_composite_backward(const float *weights, const float *grad_out, uint32_t N, uint32_t T, uint32_t K,
                             float *grad_values, sn_stream_t stream) {
    SN_REQUIRE(weights && grad_out && grad_values, "composite_backward: NULL pointer");
    if (N == 0 || K == 0 || T == 0) return SN_OK;
    hipLaunchKernelGGL(k_composite_backward, dim3(div_up((uint64_t)N * T * K, 256)), dim3(256), 0, (hipStream_t)stream, weights, grad_out, N, T, K, grad_values);
    SN_LAUNCH_CHECK("k_composite_backward");
    return SN_OK;
}

int sn_rm_mask_nll(const float *logits, const int64_t *labels, uint32_t N, uint32_t K, float eps, float *loss_per_ray, float *grad_logits,
                   sn_stream_t stream) {
    SN_REQUIRE(logits && labels && loss_per_ray, "mask_nll: NULL pointer");
    SN_REQUIRE(K >= 1, "mask_nll: at least one instance logit");
    if (N == 0) return SN_OK;
    hipLaunchKernelGGL(k_mask_nll, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, N, K, eps, loss_per_ray, grad_logits);
    SN_LAUNCH_CHECK("k_mask_nll");
    return SN_OK;
}

}  // extern "C"
