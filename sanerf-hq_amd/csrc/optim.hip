// optim.hip — dense Adam over a parameter tensor in ONE pass (gfx950).
//
// The reference trains with torch.optim.Adam(model.get_params(lr), eps=1e-15) (main.py:283): dense state, every row of
// every hash table moves each step by its decaying momentum, touched or not (gridencoder/grid.py:83 hands autograd a
// dense zeros_like gradient).  A lazy / touched-rows-only update is a different optimiser, so what can be saved is the
// memory traffic of the update itself: torch's default (foreach) implementation walks the 160 MiB mask-grid state in
// ~20 multi-tensor kernels (0.77 ms per step, profiles/r02/kernel_stats_train_mask.txt); the arithmetic below is the
// single-tensor recipe of torch/optim/adam.py (_single_tensor_adam) applied once per element with 16-byte accesses:
// 4 streams read (param, grad, exp_avg, exp_avg_sq), 3 written.  Elements whose gradient and both moments are exactly
// zero -- table rows no sample has reached yet -- are left untouched (their update is exactly 0): no stores for them.
//
// Opt-in LAZY mode (flags & SN_ADAM_LAZY; SURVEY 8 f2 "fused Adam over touched table rows only"): elements whose gradient is
// exactly zero in THIS step are skipped altogether -- moments do not decay, the parameter does not coast on its momentum -- the
// semantics of torch.optim.SparseAdam with the dense gradient's non-zeros as the sparse pattern.  A different optimiser from
// the reference's (hence opt-in): a hash table of which a 4096-ray batch touches a few percent then costs a read of the
// gradient stream plus the touched elements instead of seven full streams.
#include "sn_common.h"

namespace sn {

struct AdamArgs {
    float *p, *g, *m, *v;
    uint64_t n;
    float one_minus_beta1, beta2, one_minus_beta2, step_size, inv_bc2_sqrt, eps, weight_decay;
    int zero_grad, maximize, lazy;
    const float *step_dev;        // capturable form: the step count lives on the device (a captured HIP graph replays this launch with other counts)
    double lr, beta1_d, beta2_d;  // ... and the two bias corrections are formed from it in the kernel, in double like the host path
};

__device__ __forceinline__ bool adam_one(float &p, float g, float &m, float &v, const AdamArgs &a) {
    if (a.maximize) g = -g;
    if (a.weight_decay != 0.0f) g = g + a.weight_decay * p;              // grad.add(param, alpha=weight_decay)
    if (g == 0.0f && (a.lazy || (m == 0.0f && v == 0.0f))) return false; // update is exactly zero -- or, lazy: element not touched this step
    m = m + a.one_minus_beta1 * (g - m);                                 // exp_avg.lerp_(grad, 1 - beta1), weight < 0.5 branch
    const float gg = g * g;
    v = v * a.beta2 + a.one_minus_beta2 * gg;                            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;               // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - a.step_size * (m / denom);                                   // param.addcdiv_(exp_avg, denom, value=-step_size)
    return true;
}

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    SN_POISON_ALL();
    if (a.step_dev) {
        const double step = (double)*a.step_dev;
        const double bc1 = 1.0 - pow(a.beta1_d, step), bc2 = 1.0 - pow(a.beta2_d, step);
        a.step_size = (float)(a.lr / bc1);
        a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    }
    const uint64_t nq = a.n >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
        float4 p = reinterpret_cast<float4 *>(a.p)[i], m = reinterpret_cast<float4 *>(a.m)[i], v = reinterpret_cast<float4 *>(a.v)[i];
        const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
        bool any = adam_one(p.x, g.x, m.x, v.x, a);
        any |= adam_one(p.y, g.y, m.y, v.y, a);
        any |= adam_one(p.z, g.z, m.z, v.z, a);
        any |= adam_one(p.w, g.w, m.w, v.w, a);
        if (any) {
            reinterpret_cast<float4 *>(a.p)[i] = p; reinterpret_cast<float4 *>(a.m)[i] = m; reinterpret_cast<float4 *>(a.v)[i] = v;
            if (a.zero_grad) reinterpret_cast<float4 *>(a.g)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    // tail (n not a multiple of 4)
    const uint64_t t = (nq << 2) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < a.n) {
        float p = a.p[t], m = a.m[t], v = a.v[t];
        if (adam_one(p, a.g[t], m, v, a)) {
            a.p[t] = p; a.m[t] = m; a.v[t] = v;
            if (a.zero_grad) a.g[t] = 0.0f;
        }
    }
}

}  // namespace sn

using namespace sn;

extern "C" int sn_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, double lr, double beta1, double beta2,
                            double eps, double weight_decay, uint32_t step, const float *step_device, int maximize, int flags, sn_stream_t stream) {
    if (n == 0) return SN_OK;
    SN_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: param/grad/exp_avg/exp_avg_sq must be device pointers");
    SN_REQUIRE(table_aligned(param) && table_aligned(grad) && table_aligned(exp_avg) && table_aligned(exp_avg_sq), "adam_step: tensors must be 16-byte aligned");
    SN_REQUIRE(step >= 1 || step_device, "adam_step: step counts from 1");
    SN_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && lr >= 0.0 && eps >= 0.0, "adam_step: invalid hyper-parameters");
    // scalars exactly as torch/optim/adam.py computes them: Python doubles (hence double arguments: 1 - beta2 formed from a
    // float beta2 is off by 1e-5 relative), rounded to fp32 where they meet the tensors
    const double hstep = step_device ? 1.0 : (double)step;             // (device step: the kernel recomputes both corrections)
    const double bc1 = 1.0 - pow(beta1, hstep), bc2 = 1.0 - pow(beta2, hstep);
    AdamArgs a;
    a.step_dev = step_device; a.lr = lr; a.beta1_d = beta1; a.beta2_d = beta2;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.one_minus_beta1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.one_minus_beta2 = (float)(1.0 - beta2);
    a.step_size = (float)(lr / bc1);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = (float)eps; a.weight_decay = (float)weight_decay; a.zero_grad = flags & SN_ADAM_ZERO_GRAD; a.maximize = maximize;
    a.lazy = (flags & SN_ADAM_LAZY) ? 1 : 0;
    SN_REQUIRE(!(a.lazy && weight_decay != 0.0), "adam_step: lazy mode is defined for weight_decay = 0 (a decayed zero gradient is not a skipped element)");
    const uint64_t nq = n >> 2;
    uint64_t blocks = (nq + 255) / 256;
    if (blocks > 256u * 16u) blocks = 256u * 16u;                         // 16 workgroups per CU, grid-stride beyond
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_adam, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, a);
    SN_LAUNCH_CHECK("k_adam");
    return SN_OK;
}
