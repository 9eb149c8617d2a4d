// encoders.hip — spherical-harmonics and frequency encoders for gfx950.
//
// Replaces shencoder/src/shencoder.cu (kernel_sh, kernel_sh_backward) and
// freqencoder/src/freqencoder.cu (kernel_freq, kernel_freq_backward).
// The SH basis is not a transcription of the reference's table: sh_basis.inc is generated
// by tools/gen_sh.py from the closed-form definition of real spherical harmonics (same
// polynomials, monomial evaluation order).
#include "sn_common.h"
#include "sh_basis.inc"

namespace sn {

__device__ __forceinline__ void sh_values(float x, float y, float z, uint32_t C, float *o) {
    SN_SH_POWERS
    (void)x7; (void)y7; (void)z7;
    SN_SH_VALUES(o);
}

// one lane per direction; outputs staged through registers and written as contiguous rows
__global__ __launch_bounds__(256) void k_sh_forward(const float *__restrict__ inputs, float *__restrict__ outputs,
                                                    uint32_t B, uint32_t C, float *__restrict__ dy_dx) {
    SN_POISON_ALL();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t C2 = C * C;
    const float x = inputs[(size_t)b * 3 + 0], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
    float *o = outputs + (size_t)b * C2;
    SN_SH_POWERS
    (void)x7; (void)y7; (void)z7;
    SN_SH_VALUES(o);
    if (dy_dx) {  // layout [B, 3, C2] (shencoder.cu:126-128)
        float *gx = dy_dx + (size_t)b * 3 * C2;
        float *gy = gx + C2;
        float *gz = gy + C2;
        SN_SH_DX(gx);
        SN_SH_DY(gy);
        SN_SH_DZ(gz);
    }
}

// shencoder.cu:358-382 — accumulates into grad_inputs
__global__ __launch_bounds__(256) void k_sh_backward(const float *__restrict__ grad, const float *__restrict__ dy_dx,
                                                     float *__restrict__ grad_inputs, uint32_t B, uint32_t D, uint32_t C2) {
    SN_POISON_ALL();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *g = grad + (size_t)b * C2;
    const float *dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
    float r = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ++ch) r = __builtin_fmaf(g[ch], dd[ch], r);
    grad_inputs[t] = r;
}

// freqencoder.cu:30-58: one lane per output element.  Values are the precise sin/cos of
// x*2^f (what FreqEncoder_torch computes, encoding.py:34-40); the CUDA kernel's
// __sinf(v + pi/2) fast-math form is a lower-accuracy evaluation of the same numbers.
__global__ __launch_bounds__(256) void k_freq_forward(const float *__restrict__ inputs, uint32_t B, uint32_t D, uint32_t C,
                                                      float *__restrict__ outputs) {
    SN_POISON_ALL();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
    const float *in = inputs + (size_t)b * D;
    if (c < D) { outputs[t] = in[c]; return; }
    const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
    const float v = scalbnf(in[d], (int)freq);
    outputs[t] = (col & 1u) ? cosf(v) : sinf(v);
}

// freqencoder.cu:63-94
__global__ __launch_bounds__(256) void k_freq_backward(const float *__restrict__ grad, const float *__restrict__ outputs,
                                                       uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                                       float *__restrict__ grad_inputs) {
    SN_POISON_ALL();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *g = grad + (size_t)b * C, *o = outputs + (size_t)b * C;
    float r = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; ++f) {
        r += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = r;
}

}  // namespace sn

using namespace sn;

extern "C" {

int sn_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree,
                         float *dy_dx, sn_stream_t stream) {
    SN_REQUIRE(inputs && outputs, "sh_encode_forward: NULL device pointer");
    SN_REQUIRE(D == 3, "SH encoder only support input dim == 3");
    SN_REQUIRE(degree >= 1 && degree <= 8, "SH encoder only supports degree in [1, 8]");
    if (B == 0) return SN_OK;
    hipLaunchKernelGGL(k_sh_forward, dim3(div_up(B, 256)), dim3(256), 0, (hipStream_t)stream, inputs, outputs, B, degree, dy_dx);
    SN_LAUNCH_CHECK("k_sh_forward");
    return SN_OK;
}

int sn_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t degree,
                          const float *dy_dx, float *grad_inputs, sn_stream_t stream) {
    (void)inputs;
    SN_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: NULL device pointer");
    SN_REQUIRE(D == 3, "SH encoder only support input dim == 3");
    SN_REQUIRE(degree >= 1 && degree <= 8, "SH encoder only supports degree in [1, 8]");
    if (B == 0) return SN_OK;
    hipLaunchKernelGGL(k_sh_backward, dim3(div_up((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, grad, dy_dx,
                       grad_inputs, B, D, degree * degree);
    SN_LAUNCH_CHECK("k_sh_backward");
    return SN_OK;
}

int sn_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                           float *outputs, sn_stream_t stream) {
    SN_REQUIRE(inputs && outputs, "freq_encode_forward: NULL device pointer");
    SN_REQUIRE(C == D + 2 * D * deg, "freq_encode_forward: output_dim %u != D + 2*D*deg = %u", C, D + 2 * D * deg);
    if (B == 0) return SN_OK;
    hipLaunchKernelGGL(k_freq_forward, dim3(div_up((uint64_t)B * C, 256)), dim3(256), 0, (hipStream_t)stream, inputs, B, D, C, outputs);
    SN_LAUNCH_CHECK("k_freq_forward");
    return SN_OK;
}

int sn_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg,
                            uint32_t C, float *grad_inputs, sn_stream_t stream) {
    SN_REQUIRE(grad && outputs && grad_inputs, "freq_encode_backward: NULL device pointer");
    SN_REQUIRE(C == D + 2 * D * deg, "freq_encode_backward: output_dim %u != D + 2*D*deg = %u", C, D + 2 * D * deg);
    if (B == 0) return SN_OK;
    hipLaunchKernelGGL(k_freq_backward, dim3(div_up((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, grad, outputs, B, D, deg, C, grad_inputs);
    SN_LAUNCH_CHECK("k_freq_backward");
    return SN_OK;
}

}  // extern "C"
