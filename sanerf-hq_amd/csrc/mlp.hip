// mlp.hip — wide (256-neuron) skip-connection perceptrons of the feature heads on the matrix cores (gfx950).
//
//   sn_mlp_wide_forward    SkipConnMLP [+ LayerNorm]   (nerf/network.py:31-66, 101-128; renderer.py:359-385)
//
// samvit_mlp (163 -> 256 x4 -> 256, skip at layer 2, LayerNorm) runs once per ray and mask_mlp
// (143 -> 256 -> 256 -> n_inst) once per SAMPLE; in the reference each layer is a GEMM launch plus an
// activation launch with [rows, 256] round trips through memory.  Here one kernel carries a tile of
// 128 rows through all layers:
//
//   * transposed formulation H^T = W * X^T on v_mfma_f32_32x32x16_f16: A = weights (32 output
//     neurons x 16 inputs), B = activations (16 inputs x 32 rows), fp32 accumulate.  A wave owns 32
//     rows and ALL 256 outputs of a layer (8 accumulator tiles = 128 registers);
//   * fp32 accuracy from fp16 operands: x = hi + lo (two halves), three products hi*hi + hi*lo + lo*hi
//     (the lo*lo term is below fp32 rounding) -- the same recipe as the radiance MLP in render.hip;
//   * the accumulator layout of a 32x32 tile (reg r of lane l = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31])
//     is a valid B-operand layout, so a layer's outputs feed the next layer from registers: only the
//     weights' k order is permuted, once, by k_pack_mlp_wide;
//   * weights stream through LDS in 16 KiB chunks (1 k-step x 8 output tiles x (hi, lo) x 64 lanes x 16 B) by
//     LDS-DMA, a ring of 4 running 3 chunks ahead, shared by the 4 waves of the workgroup: 1 read per 128 rows.
//
// Packed weight stream: for each layer, its k-steps ("chunks"); h-input k-steps first (the previous
// layer's 256 outputs, 16 k-steps), then x-input k-steps (layer 0 and skip layers; input width padded
// to a multiple of 16).  Chunk = [mt 8][hi|lo][lane 64] uint4.
#include "sn_common.h"
#include <string.h>
#include <type_traits>
#include <utility>

namespace sn {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int WIDE = 256;                 // hidden width this build instantiates
constexpr int WIDE_MT = WIDE / 32;        // output tiles of a hidden layer
constexpr int WIDE_HKS = WIDE / 16;       // k-steps that consume a hidden layer
constexpr int WIDE_ROWS = 128;            // rows per workgroup (4 waves x 32)
constexpr int WIDE_CHUNK_U4 = WIDE_MT * 2 * 64;       // uint4 per chunk (one k-step, all 8 output tiles) = 16 KiB
constexpr int WIDE_NBUF = 4;                           // LDS ring: chunk g in buffer g % 4, DMA runs 3 chunks ahead

struct WideLayer {
    uint32_t uses_h, x_ks;                // h-input (previous layer) present; number of x-input k-steps
    uint32_t mt;                          // output tiles (ceil(out / 32))
    uint32_t out;                         // true output width
    uint32_t w_off;                       // first uint4 of the layer in the packed stream
    uint32_t has_bias;
    uint32_t narrow;                      // last layer with <= 64 outputs fed by h only: 4 k-steps x 1 tile pair per chunk (see run_chunk4)
};

struct WideArgs {
    const float *x;                       // [N, din]
    float *out;                           // [N, dout]
    const uint4 *pack;
    const float *bias[SN_MAX_LAYERS];
    const float *ln_w, *ln_b;             // LayerNorm over the last layer's outputs, or NULL
    float ln_eps;
    uint32_t N, din, nl, leaky, total_chunks, xs, out_lds;
    WideLayer layer[SN_MAX_LAYERS];
    // XMODE 3 (sn_rm_mask_head): the input row of sample n is cat([grid(xyz[n]), extra[n]]) and is never materialised
    const float *xyz, *extra, *wts;       // [N,3] positions, [N,E] appended channels, [N] compositing weights (N = rays * T)
    const float *table;                   // grid rows, C = 8 floats (fp32)
    uint32_t T, E;                        // samples per ray; appended channels (<= 16)
    float bound, inv_den;                 // x01 = (xyz + bound) / (2 bound); inv_den = 1 / (2 bound) when that is exact, else 0
    GridLevels g;
    // XMODE 4 (sn_mlp_wide_backward): "layer" b multiplies by W^T of forward layer nl-1-b; its 256 outputs are masked with the
    // activation derivative taken from the forward's saved outputs and written out (the weight gradients need them)
    const float *mask[SN_MAX_LAYERS];     // [N, 256] post-activation output of the forward layer that FED forward layer nl-1-b
    float *dump[SN_MAX_LAYERS];           // [N, 256] out: gradient w.r.t. that layer's pre-activation
    // sign bits of a hidden layer's outputs, 16 bytes per (row, half-wave): bit 16 t + r of lane (row, half) = (output register r of tile t > 0), i.e.
    // neuron 32 t + (r & 3) + 8 (r >> 2) + 4 half.  Written by the SAVE forward (k_mlp_wide_j), read by the backward (k_mlp_wide<5>) INSTEAD of the
    // [N, 256] fp32 outputs: the branch of a LeakyReLU / ReLU unit is all the backward data path needs of them (1 KiB -> 32 bytes per row and layer)
    uint32_t *bits[SN_MAX_LAYERS];
};

struct PackArgs {
    const float *w[SN_MAX_LAYERS];        // nn.Linear.weight [out, in]
    uint32_t in_dim[SN_MAX_LAYERS];       // row length of w[l]
    uint32_t din, nl;
    WideLayer layer[SN_MAX_LAYERS];
    uint4 *pack;
    uint32_t transposed;                  // 1: w[l] is read as its transpose (backward pass): element [m][k] = w[l][k * in_dim[l] + m]
    uint32_t pair_ks;                     // k_mlp_wide_j<3>: the first pair_ks input k-steps of layer 0 hold (half-wave h, slot i) = level 2 kx + (i >> 2),
                                          // channel 4 h + (i & 3) of the C = 8 grid instead of input column 16 kx + 8 h + i (see issue_pair)
    uint32_t pair_levels;                 // k_pack_mlp16: levels of that grid (its k-steps carry four levels each; mlp16.inc)
};

__device__ __forceinline__ void split2h(float a, float b, uint32_t &hi, uint32_t &lo) {
    const half2_t h = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    // x - hi (exact in fp32) straight from the packed halves: v_fma_mix_f32 widens its f16 operand for free
    // (render.hip:split2 -- two v_cvt_f32_f16 + a packed subtract otherwise)
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(b));
    const half2_t l = {(_Float16)l0, (_Float16)l1};
    lo = __builtin_bit_cast(uint32_t, l);
}

__device__ int g_wide_overflow = 0;      // sticky: a k_mlp_wide launch produced a non-finite output row (see the epilogue)
#ifndef SN_WIDE_TRACE
#define SN_WIDE_TRACE 0      // diagnostics build: workgroup 0 / wave 0 records the shader-cycle counter at every chunk (sn_mlp_wide_debug_trace)
#endif
#if SN_WIDE_TRACE
__device__ unsigned long long g_wide_trace[256];
__device__ __forceinline__ void wide_trace(uint32_t slot) {
    // SN_WIDE_TRACE_MID: a workgroup from the middle of the launch (warm instruction cache, neighbours at work) instead of workgroup 0
#if defined(SN_WIDE_TRACE_MID)
    if (blockIdx.x == gridDim.x / 2u && threadIdx.x == 0 && slot < 256u) g_wide_trace[slot] = __builtin_readcyclecounter();
#else
    if (blockIdx.x == 0 && threadIdx.x == 0 && slot < 256u) g_wide_trace[slot] = __builtin_readcyclecounter();
#endif
}
#else
__device__ __forceinline__ void wide_trace(uint32_t) {}
#endif


// one thread per (layer, k-step, output tile, lane): 8 weights -> (hi, lo) uint4
__global__ void k_pack_mlp_wide(PackArgs a) {
    SN_POISON_ALL();
    const uint32_t layer = blockIdx.y;
    const WideLayer L = a.layer[layer];
    const uint32_t nks = (L.uses_h ? WIDE_HKS : 0u) + L.x_ks;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nks * (uint32_t)WIDE_MT * 64u) return;
    const uint32_t lane = t & 63u, mt = (t >> 6) % (uint32_t)WIDE_MT, ks = (t >> 6) / (uint32_t)WIDE_MT;
    const uint32_t m = mt * 32u + (lane & 31u), hi = lane >> 5;
    const float *W = a.w[layer];
    const uint32_t row = a.in_dim[layer];
    float v[8];
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) {
        uint32_t col;
        bool valid = m < L.out;
        if (L.uses_h && ks < (uint32_t)WIDE_HKS) {     // previous layer's neuron held by register 8*(ks&1)+i of half-wave hi
            const uint32_t r = 8u * (ks & 1u) + i;
            col = 32u * (ks >> 1) + (r & 3u) + 8u * (r >> 2) + 4u * hi;
        } else {
            const uint32_t kx = ks - (L.uses_h ? (uint32_t)WIDE_HKS : 0u);
            const uint32_t c = (layer == 0u && kx < a.pair_ks) ? 16u * kx + 8u * (i >> 2) + 4u * hi + (i & 3u) : 16u * kx + 8u * hi + i;
            valid = valid && c < a.din;
            col = (L.uses_h ? (uint32_t)WIDE : 0u) + c;   // skip layers see cat([h, x]) (network.py:61-63)
        }
        v[i] = valid ? (a.transposed ? W[(size_t)col * row + m] : W[(size_t)m * row + col]) : 0.0f;
    }
    uint4 ph, pl;
    split2h(v[0], v[1], ph.x, pl.x); split2h(v[2], v[3], ph.y, pl.y);
    split2h(v[4], v[5], ph.z, pl.z); split2h(v[6], v[7], ph.w, pl.w);
    size_t base = (size_t)L.w_off + (size_t)ks * WIDE_MT * 128u + (size_t)mt * 128u;   // [k-step][mt][hi|lo][lane]: one chunk per k-step
    if (L.narrow) {                                                        // [chunk = ks / 4][slot = ks % 4][mt 0..1][hi|lo][lane]
        if (mt >= 2u) return;
        base = (size_t)L.w_off + (size_t)(ks >> 2) * WIDE_CHUNK_U4 + (size_t)(ks & 3u) * 256u + (size_t)mt * 128u;
    }
    a.pack[base + lane] = ph;
    a.pack[base + 64u + lane] = pl;
}

__device__ __forceinline__ floatx16 mfma3h(const uint4 &ah, const uint4 &al, const uint4 &bh, const uint4 &bl, floatx16 acc) {
    const half8_t Ah = __builtin_bit_cast(half8_t, ah), Al = __builtin_bit_cast(half8_t, al);
    const half8_t Bh = __builtin_bit_cast(half8_t, bh), Bl = __builtin_bit_cast(half8_t, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc, 0, 0, 0);   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc, 0, 0, 0);
    return acc;
}

// LDS-DMA of 16 bytes per lane (1 KiB per wave) issued through inline assembly: the compiler then does not see an
// LDS write in flight and does not put `s_waitcnt vmcnt(0)` in front of every following ds_read (it cannot prove
// the buffers distinct), which would make the copy synchronous.  The consumer waits explicitly (dma_wait) before
// the workgroup barrier that publishes the buffer.  Extra outstanding loads only make the compiler's own
// vmcnt(N) waits conservative (the counter retires in order), never early.
__device__ __forceinline__ void dma16(const void *gptr, uint32_t lds_byte_offset_uniform) {
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_byte_offset_uniform);
    // M0 = LDS destination base.  The SALU write of M0 needs ONE wait state before an LDS-DMA reads it (ISA "manually inserted wait states": S_MOV M0 ->
    // LDS-direct / buffer...lds): without the s_nop the request can go out with the PREVIOUS base -- the other ring buffer -- whenever the wave
    // issues the two instructions back to back, i.e. depending on what its SIMD neighbour is doing.  M0 is the compiler's: saved and restored.
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gptr), "s"(base) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }

// LDS: [ weight ring 4 x 16 KiB | biases nl x 256 floats | the tile's inputs ]
// XMODE 2: the tile's inputs (128 consecutive rows = one contiguous block of memory) are copied verbatim into LDS by
//          LDS-DMA, row stride = input width (odd widths -- 163, 143 -- are bank-conflict free);
// XMODE 1: copied with ordinary loads into rows padded to xs floats (16-byte aligned, conflict-free b128 reads);
// XMODE 0: too wide for LDS: read from memory per k-step (clamped, branch-free).
template <int XMODE>
__global__ __launch_bounds__(256, 1) void k_mlp_wide(WideArgs a) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(16))) uint4 lds_w[];        // WIDE_NBUF x WIDE_CHUNK_U4, then floats
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t half = lane >> 5;
    const uint32_t n = blockIdx.x * WIDE_ROWS + wave * 32u + (lane & 31u);
    const bool ok = n < a.N;
    const float *xrow = a.x + (size_t)(ok ? n : 0u) * a.din;
    float *lds_bias = reinterpret_cast<float *>(lds_w + WIDE_NBUF * WIDE_CHUNK_U4);      // [nl][WIDE], zero where absent
    float *lds_x = lds_bias + SN_MAX_LAYERS * WIDE;                               // [128][xs]
    const uint32_t xs = a.xs;
    for (uint32_t i = tid; i < a.nl * (uint32_t)WIDE; i += 256u) {
        const uint32_t l = i / (uint32_t)WIDE, m = i % (uint32_t)WIDE;
        const float *b = a.bias[l];
        lds_bias[i] = (b != nullptr && m < a.layer[l].out) ? b[m] : 0.0f;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    if constexpr (XMODE == 2) {
        // 16-byte pieces, 1 KiB per wave instruction, round-robin over the 4 waves.  Addresses are clamped to the last
        // whole 16 bytes of the array: rows past N receive (finite or not) garbage that only reaches their own,
        // never stored, outputs.  Asynchronous; waited for (in issue order) just before the barrier that publishes lds_x.
        const uint32_t tile_bytes = (uint32_t)WIDE_ROWS * a.din * 4u;
        const uint64_t arr_bytes = (uint64_t)a.N * a.din * 4u;
        const uint64_t tile0 = (uint64_t)blockIdx.x * tile_bytes;
        const uint32_t lds_x_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void *)lds_x);
        for (uint32_t piece = wave; piece * 1024u < tile_bytes; piece += 4u) {
            uint64_t off = tile0 + (uint64_t)piece * 1024u + lane * 16u;
            off = off + 16u <= arr_bytes ? off : arr_bytes - 16u;
            dma16(reinterpret_cast<const char *>(a.x) + off, lds_x_off + piece * 1024u);
        }
    }
    if constexpr (XMODE == 1) {
        // the tile's inputs -> LDS, zero padded to xs columns: wave w copies rows 32w..32w+31, 64 consecutive columns
        // per load (coalesced), 8 rows in flight
        const uint32_t row0 = blockIdx.x * WIDE_ROWS + wave * 32u;
        for (uint32_t c = lane; c < xs; c += 64u) {
#pragma unroll
            for (uint32_t rb = 0; rb < 32u; rb += 8u) {
                float t[8];
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) {
                    const uint32_t r = row0 + rb + k;
                    const uint32_t rr = r < a.N ? r : a.N - 1u, cc = c < a.din ? c : a.din - 1u;   // in range: no branch around the load
                    const float v = a.x[(size_t)rr * a.din + cc];
                    t[k] = (r < a.N && c < a.din) ? v : 0.0f;
                }
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) lds_x[(wave * 32u + rb + k) * xs + c] = t[k];
            }
        }
        if (tid < 32u) lds_x[(uint32_t)WIDE_ROWS * xs + tid] = 0.0f;         // slack read by the last row's last k-step
    }

    // XMODE 3: per-level constants where a lane can index them by ITS level (the two half-waves work on different levels)
    uint32_t *lds_lv = reinterpret_cast<uint32_t *>(lds_x);             // [4][SN_MAX_LEVELS]: res, size, mode, off
    float x01[3] = {0.0f, 0.0f, 0.0f};
    bool x_oob = false;
    if constexpr (XMODE == 3) {
        if (tid < (uint32_t)SN_MAX_LEVELS) {
            lds_lv[tid] = a.g.res[tid]; lds_lv[SN_MAX_LEVELS + tid] = a.g.size[tid];
            lds_lv[2 * SN_MAX_LEVELS + tid] = a.g.mode[tid]; lds_lv[3 * SN_MAX_LEVELS + tid] = a.g.off[tid];
        }
        const float *p = a.xyz + (size_t)(ok ? n : a.N - 1u) * 3u;
#pragma unroll
        for (int d = 0; d < 3; ++d) {   // grid.py:156
            const float t = p[d] + a.bound;
            x01[d] = a.inv_den != 0.0f ? t * a.inv_den : t / (2.0f * a.bound);
            x_oob = x_oob || x01[d] < 0.0f || x01[d] > 1.0f;                // gridencoder.cu:105-130: zeros outside [0,1]
        }
    }

    floatx16 acc[WIDE_MT];
    uint4 hbh[WIDE_HKS], hbl[WIDE_HKS];                                  // previous layer's outputs as B operands
#pragma unroll
    for (int k = 0; k < WIDE_HKS; ++k) { hbh[k] = make_uint4(0, 0, 0, 0); hbl[k] = make_uint4(0, 0, 0, 0); }

    // ---- weight stream: chunk g = k-step g of the concatenated layers (output tiles beyond a narrow last layer are
    //      zero padded by the packer), in LDS buffer g % WIDE_NBUF.  LDS-DMA (global_load_lds_dwordx4: the wave's
    //      64 lanes land 1 KiB contiguously at a wave-uniform LDS base, no staging registers) runs 3 chunks ahead:
    //      a chunk computes in ~770 cycles, an L2 fetch takes 1-2 thousand ----
    uint32_t g = 0;                        // chunks consumed so far
    const uint32_t total_chunks = a.total_chunks;
    const uint32_t lds_w_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void *)lds_w) + wave * 1024u;
    constexpr uint32_t CHUNK_BYTES = WIDE_CHUNK_U4 * sizeof(uint4);
    constexpr int PIECES = WIDE_CHUNK_U4 / 256;                          // DMA instructions per wave per chunk
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)WIDE_NBUF - 1u; ++c) {
        if (c < total_chunks) {
            const uint4 *src = a.pack + (size_t)c * WIDE_CHUNK_U4 + tid;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) dma16(src + i * 256, lds_w_off + c * CHUNK_BYTES + (uint32_t)i * 4096u);
        }
    }
    if constexpr (XMODE == 2) {
        // The input tile's DMA pieces were issued before the weight pieces and the counter retires in order: allow
        // only the weight pieces to be outstanding.  (Without this wait the first k-step's x_operand could read rows
        // whose pieces -- issued by another wave -- had not landed: seen as 2 wrong rows, the tile's last, in one of
        // a few 160 000-row runs.)
        if (total_chunks >= (uint32_t)WIDE_NBUF - 1u) asm volatile("s_waitcnt vmcnt(12)" : : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        static_assert(PIECES * (WIDE_NBUF - 1) == 12, "vmcnt immediate = weight pieces in flight");
    }
    __syncthreads();                       // publishes lds_bias / lds_x

    // one chunk = one k-step x 8 output tiles; the B operand is supplied by the caller
    // npairs: output-tile pairs the layer really has (a narrow last layer -- the mask head's 256 -> n_inst -- skips the
    // padded tiles: wave-uniform)
    // Software pipeline ACROSS chunks (round 3; cycle trace of workgroup 0 in profiles/r03/ab_round3_experiments.txt: a chunk took 1150
    // shader cycles against 768 of matrix-pipe time -- every chunk began with "wait for the DMA, workgroup barrier, read the first tile
    // pair's operands from LDS" fully exposed).  Now the synchronisation for chunk g+1 and the LDS reads of its first tile pair are issued
    // before the LAST pair's MFMAs of chunk g (192 cycles of matrix pipe cover the LDS latency), so a chunk starts with its operands
    // in registers.  Buffer (g+4) % 4 == g % 4 is refilled by DMA only after every wave has waited for its own LDS reads of chunk g
    // (lgkmcnt(0)) and passed that barrier.
    uint4 pah[2], pal[2];                  // (hi, lo) A operands of the first tile pair of the chunk about to start
    auto sync_and_prefetch = [&](uint32_t gn) {        // make chunk gn readable and refill the buffer that is now free
        const uint32_t later = total_chunks - 1u - gn;  // chunks issued after gn: their pieces may stay in flight
        if (later >= 2u) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : : : "memory");
        else if (later == 1u) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : : : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
        static_assert(PIECES == 4 && WIDE_NBUF == 4, "the vmcnt immediates above assume 4 pieces per chunk, 3 chunks ahead");
        __syncthreads();                   // every wave's pieces of chunk gn are in LDS; buffer (gn+3)%4 (chunk gn-1) has been read by all
        if (gn + 3u < total_chunks) {
            const uint4 *nsrc = a.pack + (size_t)(gn + 3u) * WIDE_CHUNK_U4 + tid;
            const uint32_t ndst = lds_w_off + ((gn + 3u) % (uint32_t)WIDE_NBUF) * CHUNK_BYTES;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) dma16(nsrc + i * 256, ndst + (uint32_t)i * 4096u);
        }
    };
    auto prefetch_first_pair = [&](uint32_t gn) {      // (reads a ring buffer even past the last chunk: harmless, never used)
        const uint4 *buf = lds_w + (gn % (uint32_t)WIDE_NBUF) * WIDE_CHUNK_U4 + lane;
        pah[0] = buf[0]; pal[0] = buf[64]; pah[1] = buf[128]; pal[1] = buf[192];
    };
    sync_and_prefetch(0u);
    prefetch_first_pair(0u);
    // tile by tile: the three products of a tile issue back to back on ONE accumulator (the matrix pipe forwards it: no accumulator
    // read / write-back between them), the next tile's two operand reads ride between them
    auto run_chunk = [&](const uint4 &bh, const uint4 &bl, uint32_t /*npairs*/) {
        wide_trace(g);
        const uint4 *buf = lds_w + (g % (uint32_t)WIDE_NBUF) * WIDE_CHUNK_U4 + lane;
        const half8_t Bh = __builtin_bit_cast(half8_t, bh), Bl = __builtin_bit_cast(half8_t, bl);
        uint4 ah[2], al[2];
        ah[0] = pah[0]; al[0] = pal[0]; ah[1] = pah[1]; al[1] = pal[1];     // tiles 0 and 1 were fetched under the previous chunk
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt) {
            const int cur = mt & 1;
            const half8_t Ah = __builtin_bit_cast(half8_t, ah[cur]), Al = __builtin_bit_cast(half8_t, al[cur]);
            floatx16 c = acc[mt];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, c, 0, 0, 0);   // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, c, 0, 0, 0);
            acc[mt] = c;
            if (mt + 2 < WIDE_MT) { ah[cur] = buf[(mt + 2) * 128]; al[cur] = buf[(mt + 2) * 128 + 64]; }
            else if (mt + 2 == WIDE_MT) {
                if (g + 1u < total_chunks) sync_and_prefetch(g + 1u);
                __builtin_amdgcn_sched_barrier(0);
                prefetch_first_pair(g + 1u);
            }
            if (mt + 2 == WIDE_MT) {       // four reads (the next chunk's first two tiles) ride on this tile's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ++g;
    };
    auto run_chunk4 = [&](const uint4 (&bh)[4], const uint4 (&bl)[4]) {
        wide_trace(g);
        const uint4 *buf = lds_w + (g % (uint32_t)WIDE_NBUF) * WIDE_CHUNK_U4 + lane;
        floatx16 c0 = acc[0], c1 = acc[1];
        uint4 ah[2][2], al[2][2];
        ah[0][0] = pah[0]; al[0][0] = pal[0]; ah[0][1] = pah[1]; al[0][1] = pal[1];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int cur = sl & 1, nxt = cur ^ 1;
            if (sl < 3) {
                ah[nxt][0] = buf[(sl + 1) * 256]; al[nxt][0] = buf[(sl + 1) * 256 + 64];
                ah[nxt][1] = buf[(sl + 1) * 256 + 128]; al[nxt][1] = buf[(sl + 1) * 256 + 192];
            } else {
                if (g + 1u < total_chunks) sync_and_prefetch(g + 1u);
                __builtin_amdgcn_sched_barrier(0);
                prefetch_first_pair(g + 1u);
            }
            const half8_t A0h = __builtin_bit_cast(half8_t, ah[cur][0]), A0l = __builtin_bit_cast(half8_t, al[cur][0]);
            const half8_t A1h = __builtin_bit_cast(half8_t, ah[cur][1]), A1l = __builtin_bit_cast(half8_t, al[cur][1]);
            const half8_t Bh = __builtin_bit_cast(half8_t, bh[sl]), Bl = __builtin_bit_cast(half8_t, bl[sl]);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0l, Bh, c0, 0, 0, 0);   // small terms first, as in run_chunk
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1l, Bh, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0h, Bl, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1h, Bl, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0h, Bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1h, Bh, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[0] = c0; acc[1] = c1;
        ++g;
    };

    // XMODE 4: the upstream gradient rows span many orders of magnitude (a sample's weight multiplies its row) and values
    // below 2^-14 would lose their lo half to fp16 subnormals, so every row is scaled by a power of two that brings its
    // largest entry to [1, 2) -- exact -- and the outputs are scaled back, exactly, on the way out
    float row_scale = 1.0f, row_unscale = 1.0f;
    constexpr bool BWD = XMODE == 4 || XMODE == 5;       // 5: the activation branches come from sign bits (a.bits) instead of the saved outputs (a.mask)
    if constexpr (BWD) {
        float mx = 0.0f;
        for (uint32_t c = 8u * half; c < a.din; c += 16u) {
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t cc = c + i;
                const float t = xrow[cc < a.din ? cc : a.din - 1u];
                mx = fmaxf(mx, cc < a.din ? fabsf(t) : 0.0f);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const uint32_t bits = __float_as_uint(mx);
        int e = (int)((bits >> 23) & 255u) - 127;
        if (mx > 0.0f && mx < __builtin_inff() && e > -100 && e < 100) {
            row_scale = __uint_as_float((uint32_t)(127 - e) << 23);
            row_unscale = __uint_as_float((uint32_t)(127 + e) << 23);
        }
    }

    auto x_operand = [&](uint32_t kx, uint4 &bh, uint4 &bl) {            // x[n][16 kx + 8 half + 0..7], zero padded
        float v[8];
        const uint32_t c0 = 16u * kx + 8u * half;
        if constexpr (XMODE == 2) {
            const float *row = lds_x + (wave * 32u + (lane & 31u)) * a.din + c0;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) { const float t = row[i]; v[i] = c0 + i < a.din ? t : 0.0f; }   // past the row: next row / slack
        } else if constexpr (XMODE == 1) {
            const float4 p = *reinterpret_cast<const float4 *>(lds_x + (wave * 32u + (lane & 31u)) * xs + c0);
            const float4 q = *reinterpret_cast<const float4 *>(lds_x + (wave * 32u + (lane & 31u)) * xs + c0 + 4u);
            v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) v[i] = c0 + i < a.din ? v[i] : 0.0f;   // columns past xs alias the next row
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t c = c0 + i;
                const float t = xrow[c < a.din ? c : a.din - 1u];         // always in range: no divergent branch around the load
                v[i] = (ok && c < a.din) ? t * row_scale : 0.0f;         // row_scale = 1 outside the backward mode
            }
        }
        split2h(v[0], v[1], bh.x, bl.x); split2h(v[2], v[3], bh.y, bl.y);
        split2h(v[4], v[5], bh.z, bl.z); split2h(v[6], v[7], bh.w, bl.w);
    };

    // XMODE 3: k-step kx of the input is levels 2 kx (low half-wave) and 2 kx + 1 (high half-wave) of the C = 8 grid: every
    // lane interpolates ONE level of its own sample -- its 8 features are exactly the lane's 8 B-operand values -- from 8
    // corner rows of 32 bytes (two 16-byte loads each; arithmetic as grid.hip:k_grid_forward, gridencoder.cu:94-201).
    // The gathers of k-step kx + 1 are in flight across the 24 MFMAs of k-step kx.
    struct LevelRegs { float pos[3]; float cv[8][8]; };
    auto issue_level = [&](uint32_t kx, LevelRegs &r) {
        const uint32_t level = umin(2u * kx + half, a.g.L - 1u);
        const uint32_t res = lds_lv[level], size = lds_lv[SN_MAX_LEVELS + level], mode = lds_lv[2 * SN_MAX_LEVELS + level];
        const float *tab = a.table + (size_t)lds_lv[3 * SN_MAX_LEVELS + level] * 8u;
        float deriv[3];
        uint32_t cell[3];
        grid_locate<3>(x01, res, a.g.align_corners != 0, a.g.interp, r.pos, deriv, cell);
#pragma unroll
        for (uint32_t idx = 0; idx < 8u; ++idx) {
            uint32_t p[3];
#pragma unroll
            for (uint32_t d = 0; d < 3u; ++d) p[d] = (idx & (1u << d)) ? umin(cell[d] + 1u, res - 1u) : cell[d];
            load_row<float, 8>(tab + (size_t)grid_row<3>(p, res, size, mode) * 8u, r.cv[idx]);
        }
    };
    auto blend_level = [&](uint32_t kx, const LevelRegs &r, uint4 &bh, uint4 &bl) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 0.0f;
#pragma unroll
        for (uint32_t idx = 0; idx < 8u; ++idx) {
            float w = 1.0f;
#pragma unroll
            for (uint32_t d = 0; d < 3u; ++d) w *= (idx & (1u << d)) ? r.pos[d] : 1.0f - r.pos[d];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = __builtin_fmaf(w, r.cv[idx][c], v[c]);
        }
        const bool zero = x_oob || 2u * kx + half >= a.g.L;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = zero ? 0.0f : v[c];
        split2h(v[0], v[1], bh.x, bl.x); split2h(v[2], v[3], bh.y, bl.y);
        split2h(v[4], v[5], bh.z, bl.z); split2h(v[6], v[7], bh.w, bl.w);
    };
    auto extra_operand = [&](uint4 &bh, uint4 &bl) {                     // the appended channels: extra[n][8 half + 0..7], zero padded
        float v[8];
        const float *row = a.extra + (size_t)(ok ? n : a.N - 1u) * a.E;
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            const uint32_t c = 8u * half + i;
            const float t = row[c < a.E ? c : a.E - 1u];
            v[i] = c < a.E ? t : 0.0f;
        }
        split2h(v[0], v[1], bh.x, bl.x); split2h(v[2], v[3], bh.y, bl.y);
        split2h(v[4], v[5], bh.z, bl.z); split2h(v[6], v[7], bh.w, bl.w);
    };

    const float act_slope = a.leaky ? 0.01f : 0.0f;
    for (uint32_t l = 0; l < a.nl; ++l) {
        const WideLayer L = a.layer[l];
        wide_trace(128u + l);
        // bias -> accumulator init (register r of this lane is neuron 32 mt + (r&3) + 8 (r>>2) + 4 half)
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4 *>(lds_bias + l * (uint32_t)WIDE + 32u * mt + 8u * q + 4u * half);
                acc[mt][4 * q] = b.x; acc[mt][4 * q + 1] = b.y; acc[mt][4 * q + 2] = b.z; acc[mt][4 * q + 3] = b.w;
            }
        }
        const uint32_t npairs = (L.mt + 1u) >> 1;
        if (L.narrow) {
#pragma unroll
            for (int c = 0; c < WIDE_HKS / 4; ++c) {
                const uint4 bh4[4] = {hbh[4 * c], hbh[4 * c + 1], hbh[4 * c + 2], hbh[4 * c + 3]};
                const uint4 bl4[4] = {hbl[4 * c], hbl[4 * c + 1], hbl[4 * c + 2], hbl[4 * c + 3]};
                run_chunk4(bh4, bl4);
            }
        } else if (L.uses_h) {
#pragma unroll
            for (int k = 0; k < WIDE_HKS; ++k) run_chunk(hbh[k], hbl[k], npairs);
        }
        if constexpr (XMODE == 3) {
            if (L.x_ks) {
                const uint32_t gks = (a.g.L + 1u) >> 1;                   // k-steps that carry grid levels; one more for the extras
                LevelRegs lr;
                issue_level(0u, lr);
                for (uint32_t k = 0; k < gks; ++k) {
                    uint4 bh, bl;
                    blend_level(k, lr, bh, bl);
                    if (k + 1u < gks) issue_level(k + 1u, lr);
                    run_chunk(bh, bl, npairs);
                }
                if (a.E) {
                    uint4 bh, bl;
                    extra_operand(bh, bl);
                    run_chunk(bh, bl, npairs);
                }
            }
        } else {
        for (uint32_t k = 0; k < L.x_ks; ++k) {
            uint4 bh, bl;
            x_operand(k, bh, bl);
            run_chunk(bh, bl, npairs);
        }
        }
        if constexpr (BWD) {
            if (l + 1u < a.nl) {
                // derivative of the activation (network.py:65-66; leaky_relu / relu keep the sign, so the saved OUTPUT tells the
                // branch, as in torch's in-place backward), gradient w.r.t. the pre-activation written out, then split
                const float slope = a.leaky ? 0.01f : 0.0f;
                const float *mrow = XMODE == 5 ? nullptr : a.mask[l] + (size_t)(ok ? n : a.N - 1u) * WIDE;
                uint4 sb = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (XMODE == 5) sb = *reinterpret_cast<const uint4 *>(a.bits[l] + ((size_t)(ok ? n : a.N - 1u) * 2u + half) * 4u);
                const uint32_t sw[4] = {sb.x, sb.y, sb.z, sb.w};
                float *drow = a.dump[l] + (size_t)(ok ? n : 0u) * WIDE;
#pragma unroll
                for (int mt = 0; mt < WIDE_MT; ++mt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t m0 = 32u * mt + 8u * q + 4u * half;
                        float4 h;
                        if constexpr (XMODE == 5) {       // bit 16 mt + r, r = 4 q + i
                            const uint32_t w = sw[mt >> 1] >> (16 * (mt & 1) + 4 * q);
                            h = make_float4((w & 1u) ? 1.0f : 0.0f, (w & 2u) ? 1.0f : 0.0f, (w & 4u) ? 1.0f : 0.0f, (w & 8u) ? 1.0f : 0.0f);
                        } else h = *reinterpret_cast<const float4 *>(mrow + m0);
                        float4 v;
                        v.x = acc[mt][4 * q + 0] * (h.x > 0.0f ? 1.0f : slope);
                        v.y = acc[mt][4 * q + 1] * (h.y > 0.0f ? 1.0f : slope);
                        v.z = acc[mt][4 * q + 2] * (h.z > 0.0f ? 1.0f : slope);
                        v.w = acc[mt][4 * q + 3] * (h.w > 0.0f ? 1.0f : slope);
                        acc[mt][4 * q + 0] = v.x; acc[mt][4 * q + 1] = v.y; acc[mt][4 * q + 2] = v.z; acc[mt][4 * q + 3] = v.w;
                        if (ok) *reinterpret_cast<float4 *>(drow + m0) = make_float4(v.x * row_unscale, v.y * row_unscale, v.z * row_unscale, v.w * row_unscale);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < WIDE_MT; ++mt) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        uint4 &bh = hbh[2 * mt + hf], &bl = hbl[2 * mt + hf];
                        split2h(acc[mt][8 * hf + 0], acc[mt][8 * hf + 1], bh.x, bl.x); split2h(acc[mt][8 * hf + 2], acc[mt][8 * hf + 3], bh.y, bl.y);
                        split2h(acc[mt][8 * hf + 4], acc[mt][8 * hf + 5], bh.z, bl.z); split2h(acc[mt][8 * hf + 6], acc[mt][8 * hf + 7], bh.w, bl.w);
                    }
                }
            }
        } else
        if (l + 1u < a.nl) {
            // activation (network.py:65-66) + split into the next layer's B operands
#pragma unroll
            for (int mt = 0; mt < WIDE_MT; ++mt) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        // t > 0 ? t : t * slope (slope 0.01 or 0) as a multiply and a max: slope < 1, so t * slope > t exactly
                        // when t < 0 (same product, same result; -0 instead of +0 for ReLU of a negative, which the split maps to 0 too)
                        const float t = acc[mt][8 * hf + i];
                        v[i] = __builtin_fmaxf(t, t * act_slope);
                    }
                    uint4 &bh = hbh[2 * mt + hf], &bl = hbl[2 * mt + hf];
                    split2h(v[0], v[1], bh.x, bl.x); split2h(v[2], v[3], bh.y, bl.y);
                    split2h(v[4], v[5], bh.z, bl.z); split2h(v[6], v[7], bh.w, bl.w);
                }
            }
        }
    }

    // ---- epilogue: optional LayerNorm over the row, then 16-byte stores (4 consecutive neurons per register quad) ----
    const WideLayer LL = a.layer[a.nl - 1u];
    {   // range check of the split-fp16 arithmetic: an activation beyond 65504 turned into inf in a hi half and reaches
        // the last layer as inf / NaN.  0 * x is NaN exactly for those.  Sticky flag, read by sn_mlp_wide_overflow().
        float chk = 0.0f;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(acc[mt][r], 0.0f, chk);
        if (ok && chk != chk) g_wide_overflow = 1;
    }
    wide_trace(160u);                      // all layers done; what follows is the output epilogue
    if constexpr (XMODE == 3) {
        // renderer.py:384: out[ray, m] = sum_t w[ray, t] * logits[ray, t, m].  Rows are samples in [ray][t] order and T divides
        // 128 or is a multiple of 32 that divides 128 (checked on the host): a ray's samples sit in T consecutive lanes of one
        // half-wave image (T <= 32) or in T / 32 whole waves of this workgroup.  Fixed reduction tree: deterministic.
        const float w = ok ? a.wts[n] : 0.0f;
        const uint32_t Tl = a.T < 32u ? a.T : 32u;                        // lanes (rows) of one wave that share a ray
        float sum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = w * acc[0][r];
#pragma unroll
            for (uint32_t d = 1; d < 32u; d <<= 1) { const float o = __shfl_xor(t, (int)d, 32); t = d < Tl ? t + o : t; }
            sum[r] = t;
        }
        const uint32_t j = lane & 31u;
        float *part = reinterpret_cast<float *>(lds_w);                  // [4 waves][32 neurons]: the weight ring is dead
        if (a.T > 32u) {
            __syncthreads();
            if (j == 0u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) part[wave * 32u + (r & 3) + 8u * (r >> 2) + 4u * half] = sum[r];
            }
            __syncthreads();
            const uint32_t wpr = a.T >> 5;                                // waves per ray: 2 or 4
            if (j == 0u && (wave % wpr) == 0u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = (r & 3) + 8u * (r >> 2) + 4u * half;
                    float t = sum[r];
                    for (uint32_t q = 1; q < wpr; ++q) t += part[(wave + q) * 32u + m];
                    sum[r] = t;
                }
            }
            if ((wave % wpr) != 0u) return;
        }
        if (ok && (j % Tl) == 0u) {
            float *orow = a.out + (size_t)(n / a.T) * LL.out;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = (r & 3) + 8u * (r >> 2) + 4u * half;
                if (m < LL.out) orow[m] = sum[r];
            }
        }
        return;
    }
    float mean = 0.0f, rstd = 1.0f;
    if (a.ln_w) {
        float s = 0.0f;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = 32u * mt + (r & 3) + 8u * (r >> 2) + 4u * half;
                s += m < LL.out ? acc[mt][r] : 0.0f;
            }
        s += __shfl_xor(s, 32);
        mean = s / (float)LL.out;
        float q = 0.0f;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = 32u * mt + (r & 3) + 8u * (r >> 2) + 4u * half;
                const float d = acc[mt][r] - mean;
                q += m < LL.out ? d * d : 0.0f;
            }
        q += __shfl_xor(q, 32);
        rstd = 1.0f / sqrtf(q / (float)LL.out + a.ln_eps);
    }
    if (a.out_lds) {
        // full-width output through LDS (the weight ring and the input tile are dead): a wave parks its 32 rows x 256
        // outputs, then writes whole rows -- 1 KiB per store instruction instead of 64 scattered 16-byte pieces
        constexpr uint32_t RS = WIDE + 4;                                 // row stride in floats (16-byte aligned, conflict-free)
        __syncthreads();
        float *reg = reinterpret_cast<float *>(lds_w) + wave * (32u * RS + 64u);
        float *stat = reg + 32u * RS;                                     // [32][2] mean, rstd
        const uint32_t r = lane & 31u;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                *reinterpret_cast<float4 *>(reg + r * RS + 32u * mt + 8u * qd + 4u * half) =
                    make_float4(acc[mt][4 * qd], acc[mt][4 * qd + 1], acc[mt][4 * qd + 2], acc[mt][4 * qd + 3]);
        if (half == 0u) { stat[2u * r] = mean; stat[2u * r + 1u] = rstd; }
        float4 w4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f), b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (a.ln_w) { w4 = reinterpret_cast<const float4 *>(a.ln_w)[lane]; b4 = reinterpret_cast<const float4 *>(a.ln_b)[lane]; }
        const uint32_t row0 = blockIdx.x * WIDE_ROWS + wave * 32u;
#pragma unroll 8
        for (uint32_t i = 0; i < 32u; ++i) {
            if (row0 + i >= a.N) break;
            float4 v = *reinterpret_cast<const float4 *>(reg + i * RS + lane * 4u);
            const float mu = stat[2u * i], rs = stat[2u * i + 1u];
            v.x = (v.x - mu) * rs * w4.x + b4.x; v.y = (v.y - mu) * rs * w4.y + b4.y;
            v.z = (v.z - mu) * rs * w4.z + b4.z; v.w = (v.w - mu) * rs * w4.w + b4.w;
            reinterpret_cast<float4 *>(a.out + (size_t)(row0 + i) * WIDE)[lane] = v;
        }
        return;
    }
    if (!ok) return;
    float *orow = a.out + (size_t)n * LL.out;
#pragma unroll
    for (int mt = 0; mt < WIDE_MT; ++mt) {
        if ((uint32_t)mt >= LL.mt) break;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const uint32_t m0 = 32u * mt + 8u * qd + 4u * half;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = acc[mt][4 * qd + i] * row_unscale;       // 1 outside the backward mode
                if (a.ln_w) {              // wave-uniform; index clamped so that the loads need no per-lane branch
                    const uint32_t mi = m0 + i < LL.out ? m0 + i : LL.out - 1u;
                    t = (t - mean) * rstd * a.ln_w[mi] + a.ln_b[mi];
                }
                v[i] = t;
            }
            if ((LL.out & 3u) == 0u && m0 + 3u < LL.out) {
                *reinterpret_cast<float4 *>(orow + m0) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (m0 + i < LL.out) orow[m0 + i] = v[i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_mlp_wide_j -- the same computation with every operand made JUST IN TIME (round 3, second half).
//
// In k_mlp_wide one in-order wave per SIMD alternates between matrix work and vector work: after a layer's last k-step it
// turns all 128 accumulators into the next layer's 128 B-operand registers (activation + fp16 hi/lo split, ~640 vector
// instructions = 5300 cycles per layer change with the matrix pipe idle), an input k-step is read and split in front of its
// 24 MFMAs, and the fused mask head interpolates a grid level (~300 vector instructions) in front of each of its first nine
// k-steps (cycle trace: 4500 cycles per gather k-step against 768 of matrix-pipe time).  Here the only thing a layer's LAST
// k-step does with a finished accumulator tile is move it, raw, into 128 ordinary registers (`prev`: 16 moves per tile, behind
// the MFMAs of the following tiles), and the B operand of k-step k+1 -- activation + hi/lo split of eight of those values, or
// eight input values, or a blended grid level -- is produced BETWEEN the MFMAs of k-step k: the vector work rides in the shadow of
// the matrix pipe, and a layer starts from its bias through the C operand of each tile's first MFMA (no accumulator
// initialisation).  The arithmetic per output is unchanged (same products, same order): results are bit-identical to k_mlp_wide.
// (A first version kept two accumulator sets used by alternate layers; the register allocator answered the 256 live accumulators
// with whole-set copies at every layer boundary: 9-10 thousand cycles per layer change, profiles/r03/ab_round3_experiments.txt.)
// ------------------------------------------------------------------------------------------
#ifndef SN_WIDE_JV
#define SN_WIDE_JV 8         // vector instructions the scheduler may place behind each MFMA of a tile
#endif

template <bool B> struct bool_tag { static constexpr bool value = B; };
template <int N> struct int_tag { static constexpr int value = N; };
// compile-time loop: f(int_tag<0>{}) ... f(int_tag<N-1>{}).  (`#pragma unroll` is a request: the 16 k-steps of a layer came back as a
// loop unrolled by 8 with a run-time index into `prev`, which put the array in scratch memory.)
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(int_tag<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// four LDS-DMA pieces of one wave with ONE m0 write and one address: the instruction offset applies to the global and to the LDS address
// alike, so a wave copies 4 KiB that are contiguous in both (k_mlp_wide gives a wave four pieces 4 KiB apart: four m0 writes, four
// addresses -- ~100 cycles of issue per k-step on an in-order wave)
__device__ __forceinline__ void dma16x4(const void *gptr, uint32_t lds_byte_offset_uniform) {
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_byte_offset_uniform);
    uint32_t keep;            // (wait state after the M0 write, M0 saved and restored: see dma16)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(base) : "memory");
}

template <int XMODE, bool N1 = false, int SAVE = 0>   // SAVE 1 (sn_mlp_wide_forward_train_f16x3): every hidden layer's post-activation output is also written to
                                           // a.dump[layer] ([N, 256] fp32), and its signs to a.bits[layer], as its tiles leave the accumulators: what the backward
                                           // pass needs.  SAVE 2 (sn_mlp_wide_backward_bits): the BACKWARD data path on this kernel -- the "MLP" is the transposed
                                           // one in reverse layer order, a tile leaving the accumulators is multiplied by the activation's derivative (from
                                           // a.bits) and written to a.dump (the gradient w.r.t. that layer's pre-activation), no activation on the way to the
                                           // next layer, rows scaled by a power of two on the way in and back on the way out (as k_mlp_wide<4>).
                                           // N1: the narrow last layer has <= 32 outputs -- ONE output tile per k-step instead of the padded pair (fused mask head:
                                           // n_inst = 2; a template so that the two forms of the layer are never both in one kernel: accumulators that meet at a
                                           // control-flow merge get copied wholesale)
__global__ __launch_bounds__(256, 1) void k_mlp_wide_j(WideArgs a) {
    SN_POISON_ALL();
    static_assert(XMODE >= 0 && XMODE <= 3, "the backward mode keeps k_mlp_wide");
    extern __shared__ __attribute__((aligned(16))) uint4 lds_w[];        // WIDE_NBUF x WIDE_CHUNK_U4, then floats
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t half = lane >> 5;
    // XMODE 0..2: one tile of 128 consecutive rows per workgroup.  XMODE 3 (fused mask head): a workgroup owns 32 CONSECUTIVE RAYS and walks
    // their T samples in T / 4 tiles -- wave w of tile p takes sample t = 4 p + w of every ray, lanes n and n + 32 share ray n -- so that a
    // gather instruction fetches the rows of 32 neighbouring pixels at ONE depth (they share fine-level lines) instead of 32 samples along
    // one ray (which share none: every 32-byte row cost a 128-byte line from beyond the L2, and that traffic bound the first layer)
    uint32_t n = blockIdx.x * WIDE_ROWS + wave * 32u + (lane & 31u);
    bool ok = n < a.N;
    const float *xrow = a.x + (size_t)(ok ? n : 0u) * a.din;
    const uint32_t ntiles = XMODE == 3 ? a.T >> 2 : 1u;
    float row_scale = 1.0f, row_unscale = 1.0f;            // SAVE 2: see k_mlp_wide<4>
    uint32_t sbits[4] = {0u, 0u, 0u, 0u};                  // SAVE 1: sign bits of the layer whose tiles are leaving (this lane's 8 tiles x 16 registers); SAVE 2: those read
    if constexpr (SAVE == 2) {
        float mx = 0.0f;
        for (uint32_t c = 8u * half; c < a.din; c += 16u) {
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t cc = c + i;
                const float t = xrow[cc < a.din ? cc : a.din - 1u];
                mx = fmaxf(mx, cc < a.din ? fabsf(t) : 0.0f);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const uint32_t mbits = __float_as_uint(mx);
        const int e = (int)((mbits >> 23) & 255u) - 127;
        if (mx > 0.0f && mx < __builtin_inff() && e > -100 && e < 100) {
            row_scale = __uint_as_float((uint32_t)(127 - e) << 23);
            row_unscale = __uint_as_float((uint32_t)(127 + e) << 23);
        }
        if (a.nl > 1u) {
            const uint4 sb = *reinterpret_cast<const uint4 *>(a.bits[0] + ((size_t)(ok ? n : a.N - 1u) * 2u + half) * 4u);
            sbits[0] = sb.x; sbits[1] = sb.y; sbits[2] = sb.z; sbits[3] = sb.w;
        }
    }
    float *lds_bias = reinterpret_cast<float *>(lds_w + WIDE_NBUF * WIDE_CHUNK_U4);      // [nl][WIDE], zero where absent
    float *lds_x = lds_bias + SN_MAX_LAYERS * WIDE;                               // [128][xs]
    const uint32_t xs = a.xs;
    for (uint32_t i = tid; i < a.nl * (uint32_t)WIDE; i += 256u) {
        const uint32_t l = i / (uint32_t)WIDE, m = i % (uint32_t)WIDE;
        const float *b = a.bias[l];
        lds_bias[i] = (b != nullptr && m < a.layer[l].out) ? b[m] : 0.0f;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    if constexpr (XMODE == 2) {          // see k_mlp_wide
        const uint32_t tile_bytes = (uint32_t)WIDE_ROWS * a.din * 4u;
        const uint64_t arr_bytes = (uint64_t)a.N * a.din * 4u;
        const uint64_t tile0 = (uint64_t)blockIdx.x * tile_bytes;
        const uint32_t lds_x_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void *)lds_x);
        for (uint32_t piece = wave; piece * 1024u < tile_bytes; piece += 4u) {
            uint64_t off = tile0 + (uint64_t)piece * 1024u + lane * 16u;
            off = off + 16u <= arr_bytes ? off : arr_bytes - 16u;
            dma16(reinterpret_cast<const char *>(a.x) + off, lds_x_off + piece * 1024u);
        }
    }
    if constexpr (XMODE == 1) {
        const uint32_t row0 = blockIdx.x * WIDE_ROWS + wave * 32u;
        for (uint32_t c = lane; c < xs; c += 64u) {
#pragma unroll
            for (uint32_t rb = 0; rb < 32u; rb += 8u) {
                float t[8];
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) {
                    const uint32_t r = row0 + rb + k;
                    const uint32_t rr = r < a.N ? r : a.N - 1u, cc = c < a.din ? c : a.din - 1u;
                    const float v = a.x[(size_t)rr * a.din + cc];
                    t[k] = (r < a.N && c < a.din) ? v : 0.0f;
                }
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) lds_x[(wave * 32u + rb + k) * xs + c] = t[k];
            }
        }
        if (tid < 32u) lds_x[(uint32_t)WIDE_ROWS * xs + tid] = 0.0f;
    }
    uint32_t *lds_lv = reinterpret_cast<uint32_t *>(lds_x);             // XMODE 3: [4][SN_MAX_LEVELS]: res, size, mode, off
    float x01[3] = {0.0f, 0.0f, 0.0f};
    bool x_oob = false;
    float ev[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};     // XMODE 3: this lane's 8 appended channels (one load per row, up front)
    if constexpr (XMODE == 3) {
        if (tid < (uint32_t)SN_MAX_LEVELS) {
            lds_lv[tid] = a.g.res[tid]; lds_lv[SN_MAX_LEVELS + tid] = a.g.size[tid];
            lds_lv[2 * SN_MAX_LEVELS + tid] = a.g.mode[tid]; lds_lv[3 * SN_MAX_LEVELS + tid] = a.g.off[tid];
        }
    }
    auto load_sample = [&](uint32_t tile) {          // XMODE 3: row of (ray, t) for this lane, its position and appended channels
        const uint32_t ray = blockIdx.x * 32u + (lane & 31u), t = 4u * tile + wave;
        ok = ray < a.N / a.T;
        n = (ok ? ray : 0u) * a.T + t;
        const float *p = a.xyz + (size_t)n * 3u;
        x_oob = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {   // grid.py:156
            const float v = p[d] + a.bound;
            x01[d] = a.inv_den != 0.0f ? v * a.inv_den : v / (2.0f * a.bound);
            x_oob = x_oob || x01[d] < 0.0f || x01[d] > 1.0f;                // gridencoder.cu:105-130: zeros outside [0,1]
        }
        if (a.E) {
            const float *row = a.extra + (size_t)n * a.E;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t c = 8u * half + i;
                const float v = row[c < a.E ? c : a.E - 1u];
                ev[i] = c < a.E ? v : 0.0f;
            }
        }
    };

    floatx16 acc[WIDE_MT];
    float prev[16 * WIDE_MT];              // the previous layer's outputs before the activation: prev[16 t + r] = register r of its tile t
    uint32_t g = 0;                        // chunks consumed so far
    const uint32_t tile_chunks = a.total_chunks;
    const uint32_t total_chunks = tile_chunks * ntiles;     // the workgroup's whole stream: the packed weights, once per tile, back to back
    const uint32_t lds_w_off4 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void *)lds_w) + wave * 4096u;   // a wave's 4 KiB of every chunk
    constexpr uint32_t CHUNK_BYTES = WIDE_CHUNK_U4 * sizeof(uint4);
    constexpr int PIECES = WIDE_CHUNK_U4 / 256;
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)WIDE_NBUF - 1u; ++c) {
        if (c < total_chunks) {
            dma16x4(a.pack + (size_t)(c % tile_chunks) * WIDE_CHUNK_U4 + wave * 256u + lane, lds_w_off4 + c * CHUNK_BYTES);
        }
    }
    if constexpr (XMODE == 2) {          // the input tile's pieces were issued first and the counter retires in order (k_mlp_wide)
        if (total_chunks >= (uint32_t)WIDE_NBUF - 1u) asm volatile("s_waitcnt vmcnt(12)" : : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        static_assert(PIECES * (WIDE_NBUF - 1) == 12, "vmcnt immediate = weight pieces in flight");
    }
    __syncthreads();                       // publishes lds_bias / lds_x / lds_lv

    // ---- weight ring: as k_mlp_wide (SN_WIDE_PIPE form).  `EXTRA` = loads of this wave that were issued after the DMA pieces of
    //      chunk gn and may stay in flight across the wait (the 16 grid gathers of the fused mask head): the counter retires
    //      in order, so they are simply allowed for in the immediate ----
    uint4 pah[2], pal[2];
    auto sync_and_prefetch = [&](uint32_t gn, auto extra_tag) {
        constexpr int EXTRA = decltype(extra_tag)::value;
        static_assert(EXTRA == 0 || EXTRA == 16, "vmcnt immediates are spelled out below");
        const uint32_t later = total_chunks - 1u - gn;
        if constexpr (EXTRA == 0) {
            if (later >= 2u) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : : : "memory");
            else if (later == 1u) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : : : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
        } else {
            if (later >= 2u) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" : : : "memory");
            else if (later == 1u) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" : : : "memory");
            else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" : : : "memory");
        }
        static_assert(PIECES == 4 && WIDE_NBUF == 4, "the vmcnt immediates above assume 4 pieces per chunk, 3 chunks ahead");
        __syncthreads();
        if (gn + 3u < total_chunks) {
            dma16x4(a.pack + (size_t)((gn + 3u) % tile_chunks) * WIDE_CHUNK_U4 + wave * 256u + lane, lds_w_off4 + ((gn + 3u) % (uint32_t)WIDE_NBUF) * CHUNK_BYTES);
        }
    };
    auto prefetch_first_pair = [&](uint32_t gn) {
        const uint4 *buf = lds_w + (gn % (uint32_t)WIDE_NBUF) * WIDE_CHUNK_U4 + lane;
        pah[0] = buf[0]; pal[0] = buf[64]; pah[1] = buf[128]; pal[1] = buf[192];
    };
    sync_and_prefetch(0u, int_tag<0>{});
    prefetch_first_pair(0u);

    const float act_slope = a.leaky ? 0.01f : 0.0f;
    auto act = [&](float t) { if constexpr (SAVE == 2) return t; else return __builtin_fmaxf(t, t * act_slope); };   // k_mlp_wide: multiply + max (network.py:65-66); backward: escape_tile applied the derivative
    auto bias_tile = [&](uint32_t l, int mt, floatx16 &b) {      // register r of this lane = neuron 32 mt + (r&3) + 8 (r>>2) + 4 half
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(lds_bias + l * (uint32_t)WIDE + 32u * mt + 8u * q + 4u * half);
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
    };

    // one chunk = one k-step x 8 output tiles.  FIRST: the layer's first k-step -- the accumulators start from the bias, handed to each
    // tile's first MFMA as its C operand.  prep(mt): the mt-th piece of whatever makes the NEXT k-step's B operand; it is emitted
    // inside tile mt's scheduling region and dealt out behind that tile's MFMAs.
    // ESC: the layer's last k-step -- finished tiles leave the accumulators for `prev` two tiles behind the matrix pipe (tiles 6 and 7 in
    // the next layer's FIRST chunk, before it overwrites them).  The empty asm makes the moved value opaque: without it the compiler keeps
    // reading the accumulator itself, whose life then overlaps the next layer's.
    auto escape_tile = [&](auto tc, uint32_t lsrc) {       // lsrc: the layer whose outputs these are
        constexpr int t = decltype(tc)::value;
        if constexpr (SAVE == 2) {  // backward: times the activation's derivative (bit 16 t + r), parked for the next layer, written out unscaled
            const float slope = a.leaky ? 0.01f : 0.0f;
            const bool hid = lsrc < a.nl - 1u;
            const uint32_t w16 = sbits[t >> 1] >> (16 * (t & 1));
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                float v = acc[t][r] * ((!hid || ((w16 >> r) & 1u)) ? 1.0f : slope);
                asm("" : "+v"(v));
                prev[16 * t + r] = v;
            });
            if (hid) {
                if (ok) {
                    float *drow = a.dump[lsrc] + (size_t)n * WIDE + 32u * t + 4u * half;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4 *>(drow + 8 * q) = make_float4(prev[16 * t + 4 * q] * row_unscale, prev[16 * t + 4 * q + 1] * row_unscale,
                                                                              prev[16 * t + 4 * q + 2] * row_unscale, prev[16 * t + 4 * q + 3] * row_unscale);
                }
                if constexpr (t == WIDE_MT - 1) {           // the layer's last tile has left: fetch the next hidden layer's bits (used from its last k-step on)
                    if (lsrc + 2u < a.nl) {
                        const uint4 sb = *reinterpret_cast<const uint4 *>(a.bits[lsrc + 1u] + ((size_t)(ok ? n : a.N - 1u) * 2u + half) * 4u);
                        sbits[0] = sb.x; sbits[1] = sb.y; sbits[2] = sb.z; sbits[3] = sb.w;
                    }
                }
            }
            return;
        }
        static_for<16>([&](auto rc) { constexpr int r = decltype(rc)::value; float v = acc[t][r]; asm("" : "+v"(v)); prev[16 * t + r] = v; });
        if constexpr (SAVE == 1) {  // register r of this lane = neuron 32 t + (r & 3) + 8 (r >> 2) + 4 half of row n
            if (ok && lsrc < a.nl - 1u) {                     // (the first layer's first chunk "escapes" tiles of a layer that does not exist: lsrc = ~0)
                float *drow = a.dump[lsrc] + (size_t)n * WIDE + 32u * t + 4u * half;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(drow + 8 * q) = make_float4(act(prev[16 * t + 4 * q]), act(prev[16 * t + 4 * q + 1]), act(prev[16 * t + 4 * q + 2]), act(prev[16 * t + 4 * q + 3]));
                // the units' branches as bits (WideArgs::bits): the output is positive exactly when the pre-activation is
                uint32_t b16 = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) b16 |= (prev[16 * t + r] > 0.0f ? 1u : 0u) << r;
                if constexpr ((t & 1) == 0) sbits[t >> 1] = b16; else sbits[t >> 1] |= b16 << 16;
                if constexpr (t == WIDE_MT - 1) {               // tile 7 is the last one of a layer to leave the accumulators
                    if (a.bits[lsrc]) *reinterpret_cast<uint4 *>(a.bits[lsrc] + ((size_t)n * 2u + half) * 4u) = make_uint4(sbits[0], sbits[1], sbits[2], sbits[3]);
                }
            }
        } else (void)lsrc;
    };
    auto chunk8 = [&](auto first_tag, auto esc_tag, auto extra_tag, uint32_t l, const uint32_t (&ob_h)[4], const uint32_t (&ob_l)[4], auto &&prep) {
        constexpr bool FIRST = decltype(first_tag)::value, ESC = decltype(esc_tag)::value;
        if (g < 128u) wide_trace(g);
        const uint4 *buf = lds_w + (g % (uint32_t)WIDE_NBUF) * WIDE_CHUNK_U4 + lane;
        const half8_t Bh = __builtin_bit_cast(half8_t, make_uint4(ob_h[0], ob_h[1], ob_h[2], ob_h[3]));
        const half8_t Bl = __builtin_bit_cast(half8_t, make_uint4(ob_l[0], ob_l[1], ob_l[2], ob_l[3]));
        uint4 ah[2], al[2];
        ah[0] = pah[0]; al[0] = pal[0]; ah[1] = pah[1]; al[1] = pal[1];
        floatx16 bias[2];
        if constexpr (FIRST) { bias_tile(l, 0, bias[0]); bias_tile(l, 1, bias[1]); }
        static_for<WIDE_MT>([&](auto mtc) {
            constexpr int mt = decltype(mtc)::value, cur = mt & 1;
            const half8_t Ah = __builtin_bit_cast(half8_t, ah[cur]), Al = __builtin_bit_cast(half8_t, al[cur]);
            floatx16 c;
            if constexpr (FIRST) c = bias[cur]; else c = acc[mt];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, c, 0, 0, 0);   // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, c, 0, 0, 0);
            acc[mt] = c;
            if constexpr (mt + 2 < WIDE_MT) {
                ah[cur] = buf[(mt + 2) * 128]; al[cur] = buf[(mt + 2) * 128 + 64];
                if constexpr (FIRST) bias_tile(l, mt + 2, bias[cur]);
            } else if constexpr (mt + 2 == WIDE_MT) {
                if (g + 1u < total_chunks) sync_and_prefetch(g + 1u, extra_tag);
                __builtin_amdgcn_sched_barrier(0);
                prefetch_first_pair(g + 1u);
            }
            if constexpr (FIRST && mt < 2) escape_tile(int_tag<6 + mt>{}, l - 1u);
            if constexpr (ESC && mt >= 2) escape_tile(int_tag<mt - 2>{}, l);
            prep(mtc);
            if constexpr (mt + 2 == WIDE_MT) {       // the tile's MFMAs were fenced in front of the synchronisation; what follows rides on tile 7
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, FIRST ? 3 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SN_WIDE_JV, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, FIRST ? 3 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SN_WIDE_JV, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SN_WIDE_JV, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        ++g;
    };

    // ---- operand makers ----
    auto h_piece = [&](auto kc, auto pc, uint32_t (&nbh)[4], uint32_t (&nbl)[4]) {      // pair p (0..3) of h k-step k
        constexpr int k = decltype(kc)::value, p = decltype(pc)::value;
        split2h(act(prev[8 * k + 2 * p]), act(prev[8 * k + 2 * p + 1]), nbh[p], nbl[p]);
    };
    // x[n][16 kx + 8 half + 0..7], zero padded (XMODE 0..2)
    auto x_values = [&](uint32_t kx, float (&v)[8]) {
        const uint32_t c0 = 16u * kx + 8u * half;
        if constexpr (XMODE == 2) {
            const float *row = lds_x + (wave * 32u + (lane & 31u)) * a.din + c0;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) { const float t = row[i]; v[i] = c0 + i < a.din ? t : 0.0f; }
        } else if constexpr (XMODE == 1) {
            const float4 p = *reinterpret_cast<const float4 *>(lds_x + (wave * 32u + (lane & 31u)) * xs + c0);
            const float4 q = *reinterpret_cast<const float4 *>(lds_x + (wave * 32u + (lane & 31u)) * xs + c0 + 4u);
            v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) v[i] = c0 + i < a.din ? v[i] : 0.0f;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t c = c0 + i;
                const float t = xrow[c < a.din ? c : a.din - 1u];
                v[i] = (ok && c < a.din) ? t : 0.0f;
            }
        }
        if constexpr (SAVE == 2) {
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) v[i] *= row_scale;
        }
    };
    // operand = input k-step kxn: loads at tile 0, split in tiles 2..5
    float xv[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    auto prep_x = [&](auto mtc, uint32_t kxn, uint32_t (&nbh)[4], uint32_t (&nbl)[4]) {
        constexpr int mt = decltype(mtc)::value;
        if constexpr (XMODE != 3 && mt == 0) x_values(kxn, xv);
        if constexpr (mt >= 2 && mt < 6) split2h(xv[2 * (mt - 2)], xv[2 * (mt - 2) + 1], nbh[mt - 2], nbl[mt - 2]);
    };
    // operand = the next layer's first h k-step, from tile 0 as escaped at tile 2 of an ESC chunk
    auto prep_h0 = [&](auto mtc, uint32_t (&nbh)[4], uint32_t (&nbl)[4]) {
        constexpr int mt = decltype(mtc)::value;
        if constexpr (mt >= 2 && mt < 6) h_piece(int_tag<0>{}, int_tag<mt - 2>{}, nbh, nbl);
    };
    // the last h k-step of a layer is followed EITHER by input k-step 0 (skip layers) OR by the next layer: both operands are formed and a
    // wave-uniform select picks (no branch inside a chunk: one scheduling region per tile)
    auto prep_sel = [&](auto mtc, bool use_x, uint32_t (&nbh)[4], uint32_t (&nbl)[4]) {
        constexpr int mt = decltype(mtc)::value;
        if constexpr (XMODE != 3 && mt == 0) x_values(0u, xv);
        if constexpr (mt >= 2 && mt < 6) {
            constexpr int p = mt - 2;
            const float h0 = act(prev[2 * p]), h1 = act(prev[2 * p + 1]);
            split2h(use_x ? xv[2 * p] : h0, use_x ? xv[2 * p + 1] : h1, nbh[p], nbl[p]);
        }
    };

    // XMODE 3: input k-step kx carries levels A = 2 kx and B = 2 kx + 1 of the C = 8 grid.  In k_mlp_wide<3> the low half-wave interpolates
    // level A and the high one level B of the SAME 32 samples, every lane fetching both 16-byte halves of its 8 corner rows: 16 gather
    // instructions per k-step that each touch up to 64 rows -- and those k-steps are bound by the texture path (cycle trace: 9-12 thousand
    // cycles per k-step with random positions, 4 waves on one address path).  Here lanes n and n + 32 SHARE the rows of sample n: each fetches
    // ONE half (channels 4 h .. 4 h + 3) of the 8 corner rows of BOTH levels, so a gather instruction touches 32 rows instead of 64 and no
    // data has to be exchanged; only the bookkeeping is split -- the low half-wave locates level A, the high one level B, and one
    // v_permlane32_swap per value hands row offsets and blend weights to the other half.  The operand's slots become (h, i) = level
    // A / B (i >> 2), channel 4 h + (i & 3): a permutation of the first layer's input columns that k_pack_mlp_wide applies (pair_ks).
    struct PairRegs { float wA[8], wB[8]; float4 rA[8], rB[8]; };
    auto issue_pair = [&](uint32_t kx, PairRegs &r) {
        const uint32_t level = umin(2u * kx + half, a.g.L - 1u);
        const uint32_t res = lds_lv[level], size = lds_lv[SN_MAX_LEVELS + level], mode = lds_lv[2 * SN_MAX_LEVELS + level];
        const uint32_t base = lds_lv[3 * SN_MAX_LEVELS + level];
        // Addresses without branches (round 4): the host admits only grids of the fast-path shape (levels_fast: hashed levels of power-of-two
        // size, dense levels over all three dimensions, align_corners = False, linear interpolation), so the 8 row offsets are 6 partial
        // terms combined by xor / add (corner_offsets, as in the fused render stages) instead of 8 calls of the generic grid_row, whose
        // hash / dense / modulo branches diverge between the half-waves (they work on different levels): ~560 -> ~90 vector instructions
        // per k-step and no basic-block boundaries between the gathers.  Same rows, same weights ((wx wy) wz).
        float pos[3];
        uint32_t cell[3], offs[8];
        locate_linear(x01, res, pos, cell);
        corner_offsets<-1, 32u>(cell, res, size, mode, offs);           // bytes from the level's first row
        const float wx[2] = {1.0f - pos[0], pos[0]}, wy[2] = {1.0f - pos[1], pos[1]}, wz[2] = {1.0f - pos[2], pos[2]};
        const float wxy[4] = {wx[0] * wy[0], wx[1] * wy[0], wx[0] * wy[1], wx[1] * wy[1]};
        const char *tabh = reinterpret_cast<const char *>(a.table) + 16u * half;                         // this lane's half of every row
#pragma unroll
        for (uint32_t idx = 0; idx < 8u; ++idx) {
            const float w = wxy[idx & 3u] * wz[idx >> 2];
            const uint32_t off = base * 32u + offs[idx];                  // bytes from the table's start (< 2^32: checked on the host)
            const auto po = __builtin_amdgcn_permlane32_swap(off, off, false, false);                                   // [A | A], [B | B]
            const auto pw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
            r.wA[idx] = __uint_as_float(pw[0]); r.wB[idx] = __uint_as_float(pw[1]);
            r.rA[idx] = *reinterpret_cast<const float4 *>(tabh + po[0]);
            r.rB[idx] = *reinterpret_cast<const float4 *>(tabh + po[1]);
        }
    };
    auto blend_pair = [&](const PairRegs &r, int first, int last, float (&v)[8]) {
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
            if (idx < first || idx >= last) continue;
            const float ra[4] = {r.rA[idx].x, r.rA[idx].y, r.rA[idx].z, r.rA[idx].w}, rb[4] = {r.rB[idx].x, r.rB[idx].y, r.rB[idx].z, r.rB[idx].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = __builtin_fmaf(r.wA[idx], ra[c], idx == 0 ? 0.0f : v[c]);
                v[4 + c] = __builtin_fmaf(r.wB[idx], rb[c], idx == 0 ? 0.0f : v[4 + c]);
            }
        }
    };

    uint32_t bh[4], bl[4], nbh[4] = {0u, 0u, 0u, 0u}, nbl[4] = {0u, 0u, 0u, 0u};   // operand of the chunk about to run / of the one after it
    const uint32_t nl = a.nl;

    float out_acc[16];                     // XMODE 3: sum over this wave's samples of weight x logit, per ray (neurons as tile 0's registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) out_acc[r] = 0.0f;
    float wt = 0.0f;                       // XMODE 3: this lane's compositing weight in the current tile
    if constexpr (XMODE == 3) ok = blockIdx.x * 32u + (lane & 31u) < a.N / a.T;     // (a workgroup whose tiles are all skipped never runs load_sample)
    for (uint32_t tile = 0; tile < ntiles; ++tile) {
    if constexpr (XMODE == 3) {
        // A tile whose 128 samples all carry weight exactly 0 (behind an opaque surface the transmittance has underflowed, in empty space
        // alpha = 1 - exp(-0) = 0) adds w * logit = 0 to every ray: it is skipped whole -- no gathers, no MLP.  Bit-identical as long as
        // the skipped logits would have been finite (0 * inf = NaN in the reference's sum).  The weight stream needs no adjustment: every
        // tile consumes the same chunk sequence, so the chunks prefetched for "the next tile" serve the tile after it just as well.
        const uint32_t ray_ = blockIdx.x * 32u + (lane & 31u), t_ = 4u * tile + wave;
        const bool ok_ = ray_ < a.N / a.T;
        wt = ok_ ? a.wts[(size_t)ray_ * a.T + t_] : 0.0f;
        if (__syncthreads_or(wt != 0.0f) == 0) continue;
        load_sample(tile);
    }
    // ---- layer 0: input k-steps only; accumulators start from the bias the plain way (once per tile) ----
    {
        const WideLayer L = a.layer[0];
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt) bias_tile(0u, mt, acc[mt]);
        if (tile == 0u) wide_trace(128u);
        if constexpr (XMODE == 3) {
            const uint32_t gks = (a.g.L + 1u) >> 1;                       // k-steps that carry grid levels; one more for the appended channels
            const uint32_t K = gks + (a.E ? 1u : 0u);
            PairRegs S;
            const uint32_t Lg = a.g.L;
            issue_pair(0u, S);
            {
                float v[8];
                blend_pair(S, 0, 8, v);
                const bool zA = x_oob, zB = x_oob || 1u >= Lg;
#pragma unroll
                for (int p = 0; p < 4; ++p) { const bool z = p < 2 ? zA : zB; split2h(z ? 0.0f : v[2 * p], z ? 0.0f : v[2 * p + 1], bh[p], bl[p]); }
            }
            issue_pair(1u, S);
            float bv[8];
            // chunk c < K-1: blend k-step c+1 (requested during chunk c-1) behind tiles 0..2, request the rows of k-step c+2 into the same
            // registers behind tile 3 (the k-step after the last level one is the appended channels); the last chunk is peeled: it hands
            // the layer over.  One register set: the texture path, not the latency of one wave, bounds these k-steps (4 waves share it)
            for (uint32_t c = 0; c + 1u < K; ++c) {
                const bool lev = c + 1u < gks;
                chunk8(bool_tag<false>{}, bool_tag<false>{}, int_tag<16>{}, 0u, bh, bl, [&](auto mtc) {
                    constexpr int mt = decltype(mtc)::value;
                    if constexpr (mt == 0) blend_pair(S, 0, 3, bv);
                    if constexpr (mt == 1) blend_pair(S, 3, 6, bv);
                    if constexpr (mt == 2) blend_pair(S, 6, 8, bv);
                    if constexpr (mt == 3) issue_pair(c + 2u, S);
                    if constexpr (mt == 5) {
                        const bool zA = x_oob || 2u * (c + 1u) >= Lg, zB = x_oob || 2u * (c + 1u) + 1u >= Lg;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const bool z = p < 2 ? zA : zB;
                            const float l0 = z ? 0.0f : bv[2 * p], l1 = z ? 0.0f : bv[2 * p + 1];
                            split2h(lev ? l0 : ev[2 * p], lev ? l1 : ev[2 * p + 1], nbh[p], nbl[p]);
                        }
                    }
                });
#pragma unroll
                for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
            }
            chunk8(bool_tag<false>{}, bool_tag<true>{}, int_tag<16>{}, 0u, bh, bl, [&](auto mtc) { prep_h0(mtc, nbh, nbl); });
#pragma unroll
            for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
        } else {
            {
                float v[8];
                x_values(0u, v);
#pragma unroll
                for (int p = 0; p < 4; ++p) split2h(v[2 * p], v[2 * p + 1], bh[p], bl[p]);
            }
            for (uint32_t k = 0; k + 1u < L.x_ks; ++k) {
                chunk8(bool_tag<false>{}, bool_tag<false>{}, int_tag<0>{}, 0u, bh, bl, [&](auto mtc) { prep_x(mtc, k + 1u, nbh, nbl); });
#pragma unroll
                for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
            }
            chunk8(bool_tag<false>{}, bool_tag<true>{}, int_tag<0>{}, 0u, bh, bl, [&](auto mtc) { prep_h0(mtc, nbh, nbl); });
#pragma unroll
            for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
        }
    }

    // ---- layers 1 .. nl-1 ----
    for (uint32_t l = 1; l < nl; ++l) {
        const WideLayer L = a.layer[l];
        if (tile == 0u) wide_trace(128u + l);
        if (L.narrow) {
            {
            // last layer with <= 64 outputs: 4 chunks of 4 k-steps x 1 tile pair (k_pack_mlp_wide); the operand of k-step k+1 is split
            // between the six MFMAs of k-step k.  Tiles 6 and 7 of the previous layer leave the accumulators first.
            escape_tile(int_tag<6>{}, l - 1u); escape_tile(int_tag<7>{}, l - 1u);
            floatx16 c0, c1;
            bias_tile(l, 0, c0); bias_tile(l, 1, c1);
            static_for<WIDE_HKS / 4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (g < 128u) wide_trace(g);
                const uint4 *buf = lds_w + (g % (uint32_t)WIDE_NBUF) * WIDE_CHUNK_U4 + lane;
                uint4 ah[2][2], al[2][2];
                ah[0][0] = pah[0]; al[0][0] = pal[0]; ah[0][1] = pah[1]; al[0][1] = pal[1];
                static_for<4>([&](auto slc) {
                    constexpr int sl = decltype(slc)::value, k = 4 * c + sl, cb = sl & 1, nb = cb ^ 1;
                    if constexpr (sl < 3) {
                        ah[nb][0] = buf[(sl + 1) * 256]; al[nb][0] = buf[(sl + 1) * 256 + 64];
                        if constexpr (!N1) { ah[nb][1] = buf[(sl + 1) * 256 + 128]; al[nb][1] = buf[(sl + 1) * 256 + 192]; }
                    } else {
                        if (g + 1u < total_chunks) sync_and_prefetch(g + 1u, int_tag<0>{});
                        __builtin_amdgcn_sched_barrier(0);
                        prefetch_first_pair(g + 1u);
                    }
                    const half8_t A0h = __builtin_bit_cast(half8_t, ah[cb][0]), A0l = __builtin_bit_cast(half8_t, al[cb][0]);
                    const half8_t A1h = __builtin_bit_cast(half8_t, ah[cb][1]), A1l = __builtin_bit_cast(half8_t, al[cb][1]);
                    const half8_t Bh = __builtin_bit_cast(half8_t, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                    const half8_t Bl = __builtin_bit_cast(half8_t, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                    if constexpr (N1) {       // <= 32 outputs: tile 1 is padding (c1 keeps its zero bias; its weight rows are never read)
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0l, Bh, c0, 0, 0, 0);   // small terms first
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0h, Bl, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0h, Bh, c0, 0, 0, 0);
                    } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0l, Bh, c0, 0, 0, 0);   // small terms first
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1l, Bh, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0h, Bl, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1h, Bl, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0h, Bh, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1h, Bh, c1, 0, 0, 0);
                    }
                    if constexpr (k + 1 < WIDE_HKS) static_for<4>([&](auto pc) { h_piece(int_tag<k + 1>{}, pc, nbh, nbl); });
#pragma unroll
                    for (int i = 0; i < (N1 ? 3 : 6); ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < (N1 ? 2 : 4)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, N1 ? 2 * SN_WIDE_JV : SN_WIDE_JV, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
                });
                ++g;
            });
            acc[0] = c0; acc[1] = c1;
            }
        } else {
            const bool has_x = L.x_ks != 0u;
            static_for<WIDE_HKS>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                auto prep_next_h = [&](auto mtc) {
                    constexpr int mt = decltype(mtc)::value;
                    if constexpr (mt >= 2 && mt < 6) h_piece(int_tag<k + 1>{}, int_tag<mt - 2>{}, nbh, nbl);
                };
                if constexpr (k == 0) chunk8(bool_tag<true>{}, bool_tag<false>{}, int_tag<0>{}, l, bh, bl, prep_next_h);
                else if constexpr (k + 1 < WIDE_HKS) chunk8(bool_tag<false>{}, bool_tag<false>{}, int_tag<0>{}, l, bh, bl, prep_next_h);
                else chunk8(bool_tag<false>{}, bool_tag<true>{}, int_tag<0>{}, l, bh, bl, [&](auto mtc) { prep_sel(mtc, has_x, nbh, nbl); });
#pragma unroll
                for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
            });
            if (has_x) {
                for (uint32_t k = 0; k + 1u < L.x_ks; ++k) {
                    chunk8(bool_tag<false>{}, bool_tag<false>{}, int_tag<0>{}, l, bh, bl, [&](auto mtc) { prep_x(mtc, k + 1u, nbh, nbl); });
#pragma unroll
                    for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
                }
                chunk8(bool_tag<false>{}, bool_tag<true>{}, int_tag<0>{}, l, bh, bl, [&](auto mtc) { prep_h0(mtc, nbh, nbl); });
#pragma unroll
                for (int p = 0; p < 4; ++p) { bh[p] = nbh[p]; bl[p] = nbl[p]; }
            }
        }
    }

    // ---- epilogue (k_mlp_wide's) ----
    {   // range check of the split-fp16 arithmetic: an activation beyond 65504 turned into inf in a hi half and reaches
        // the last layer as inf / NaN.  0 * x is NaN exactly for those.  Sticky flag, read by sn_mlp_wide_overflow().
        float chk = 0.0f;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(acc[mt][r], 0.0f, chk);
        if (ok && chk != chk) g_wide_overflow = 1;
    }
    if constexpr (XMODE == 3) {            // renderer.py:384: out[ray, m] = sum_t w[ray, t] * logits[ray, t, m]: this wave's t of this tile
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float t = wt * acc[0][r]; out_acc[r] = out_acc[r] + t; }
    }
    }                                      // tiles
    const WideLayer LL = a.layer[a.nl - 1u];
    wide_trace(160u);                      // all layers done; what follows is the output epilogue
    if constexpr (XMODE == 3) {
        // the four waves hold different samples of the same 32 rays: fixed-order sum through LDS (the weight ring is dead), wave 0 stores
        float *part = reinterpret_cast<float *>(lds_w);                  // [4 waves][32 rays][32 neurons]
        const uint32_t j = lane & 31u;
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");             // (skipped tiles leave prefetched weight chunks in flight: they must land before the ring is reused)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wave * 32u + j) * 32u + (r & 3) + 8u * (r >> 2) + 4u * half] = out_acc[r];
        __syncthreads();
        if (wave == 0u && ok) {
            float *orow = a.out + (size_t)(blockIdx.x * 32u + j) * LL.out;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = (r & 3) + 8u * (r >> 2) + 4u * half;
                const float t = ((part[(0u * 32u + j) * 32u + m] + part[(1u * 32u + j) * 32u + m]) + part[(2u * 32u + j) * 32u + m]) + part[(3u * 32u + j) * 32u + m];
                if (m < LL.out) orow[m] = t;
            }
        }
        return;
    }
    float mean = 0.0f, rstd = 1.0f;
    if (a.ln_w) {
        float s = 0.0f;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = 32u * mt + (r & 3) + 8u * (r >> 2) + 4u * half;
                s += m < LL.out ? acc[mt][r] : 0.0f;
            }
        s += __shfl_xor(s, 32);
        mean = s / (float)LL.out;
        float q = 0.0f;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = 32u * mt + (r & 3) + 8u * (r >> 2) + 4u * half;
                const float d = acc[mt][r] - mean;
                q += m < LL.out ? d * d : 0.0f;
            }
        q += __shfl_xor(q, 32);
        rstd = 1.0f / sqrtf(q / (float)LL.out + a.ln_eps);
    }
    if (a.out_lds) {
        // full-width output through LDS (the weight ring and the input tile are dead): a wave parks its 32 rows x 256
        // outputs, then writes whole rows -- 1 KiB per store instruction instead of 64 scattered 16-byte pieces
        constexpr uint32_t RS = WIDE + 4;                                 // row stride in floats (16-byte aligned, conflict-free)
        __syncthreads();
        float *reg = reinterpret_cast<float *>(lds_w) + wave * (32u * RS + 64u);
        float *stat = reg + 32u * RS;                                     // [32][2] mean, rstd
        const uint32_t r = lane & 31u;
#pragma unroll
        for (int mt = 0; mt < WIDE_MT; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                *reinterpret_cast<float4 *>(reg + r * RS + 32u * mt + 8u * qd + 4u * half) =
                    make_float4(acc[mt][4 * qd] * row_unscale, acc[mt][4 * qd + 1] * row_unscale, acc[mt][4 * qd + 2] * row_unscale, acc[mt][4 * qd + 3] * row_unscale);   // (row_unscale = 1 outside the backward mode)
        if (half == 0u) { stat[2u * r] = mean; stat[2u * r + 1u] = rstd; }
        float4 w4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f), b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (a.ln_w) { w4 = reinterpret_cast<const float4 *>(a.ln_w)[lane]; b4 = reinterpret_cast<const float4 *>(a.ln_b)[lane]; }
        const uint32_t row0 = blockIdx.x * WIDE_ROWS + wave * 32u;
#pragma unroll 8
        for (uint32_t i = 0; i < 32u; ++i) {
            if (row0 + i >= a.N) break;
            float4 v = *reinterpret_cast<const float4 *>(reg + i * RS + lane * 4u);
            const float mu = stat[2u * i], rs = stat[2u * i + 1u];
            v.x = (v.x - mu) * rs * w4.x + b4.x; v.y = (v.y - mu) * rs * w4.y + b4.y;
            v.z = (v.z - mu) * rs * w4.z + b4.z; v.w = (v.w - mu) * rs * w4.w + b4.w;
            reinterpret_cast<float4 *>(a.out + (size_t)(row0 + i) * WIDE)[lane] = v;
        }
        return;
    }
    if (!ok) return;
    float *orow = a.out + (size_t)n * LL.out;
#pragma unroll
    for (int mt = 0; mt < WIDE_MT; ++mt) {
        if ((uint32_t)mt >= LL.mt) break;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const uint32_t m0 = 32u * mt + 8u * qd + 4u * half;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = acc[mt][4 * qd + i];
                if constexpr (SAVE == 2) t *= row_unscale;
                if (a.ln_w) {              // wave-uniform; index clamped so that the loads need no per-lane branch
                    const uint32_t mi = m0 + i < LL.out ? m0 + i : LL.out - 1u;
                    t = (t - mean) * rstd * a.ln_w[mi] + a.ln_b[mi];
                }
                v[i] = t;
            }
            if ((LL.out & 3u) == 0u && m0 + 3u < LL.out) {
                *reinterpret_cast<float4 *>(orow + m0) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (m0 + i < LL.out) orow[m0 + i] = v[i];
            }
        }
    }
}

#include "mlp_f32.inc"   // k_mlp_f32_train: the training forward in true fp32 on the matrix cores, activations fused, hidden outputs saved (round 5)
#include "mlp16.inc"     // k_mask16: the fused mask head on 16-row tiles, two waves per SIMD, tile pipeline (round 5)

// Which forward kernel: k_mlp_wide_j (operands just in time; SAM head MLP 0.449 -> 0.433 ms, mask MLP 0.119 -> 0.114 ms, 400x400 mask render
// 7.49 -> 7.13 ms; bit-identical).  The superseded k_mlp_wide forward modes are compiled only into experiments builds (-DSN_EXPERIMENTS),
// where sn_debug_set("wide_jit", 0) selects them for an A/B.  (With fewer than 4 samples per ray the fused mask head keeps k_mlp_wide<3>;
// the backward-data pass is k_mlp_wide<4>.)
#ifdef SN_EXPERIMENTS
static int g_wide_jit = 1;
static int g_wide_narrow1 = 1;       // sn_debug_set("wide_narrow1", 0): the fused mask head's last layer as a padded tile PAIR as before (A/B)
static int g_wide_bwd_j = 0;         // sn_debug_set("wide_bwd_j", 1): sn_mlp_wide_backward_bits on k_mlp_wide_j<.., 2> instead of k_mlp_wide<5> (A/B: measured equal, 0.213 vs 0.212 ms,
                                     // bit-identical -- the pass is bound by its 348 MB of output, not by the skeleton; profiles/r05/mlp_bwd_ab.txt)
static int g_mask_head16 = 8;        // sn_debug_set("mask_head16", 0): the fused mask head on k_mlp_wide_j<3> (rounds 3-4) also where k_mask16 (mlp16.inc) takes it (A/B)
static bool wide_jit(int xmode) { (void)xmode; return g_wide_jit != 0; }
#else
static constexpr bool wide_jit(int) { return true; }
static constexpr int g_wide_narrow1 = 1;
static constexpr int g_mask_head16 = 8;
static constexpr int g_wide_bwd_j = 0;

#endif

extern "C" int sn_debug_set(const char *key, int value) {
    SN_REQUIRE(key, "debug_set: NULL key");
#ifdef SN_EXPERIMENTS
    if (strcmp(key, "wide_jit") == 0) { g_wide_jit = value; return SN_OK; }
    if (strcmp(key, "wide_narrow1") == 0) { g_wide_narrow1 = value; return SN_OK; }
    if (strcmp(key, "mask_head16") == 0) { g_mask_head16 = value; return SN_OK; }
    if (strcmp(key, "wide_bwd_j") == 0) { g_wide_bwd_j = value; return SN_OK; }
    if (strcmp(key, "bin_pull") == 0) { g_bin_pull = value; return SN_OK; }          // grid_binned.hip: entries as products (0) or references (1); -1 = by C
    set_error("debug_set: unknown key '%s'", key);
    return SN_ERR_INVALID;
#else
    (void)value;
    set_error("debug_set('%s'): this library was built without -DSN_EXPERIMENTS (make exp)", key);
    return SN_ERR_UNSUPPORTED;
#endif
}

static int wide_plan(const sn_mlp_desc *m, WideLayer *layers, size_t *total_u4) {
    SN_REQUIRE(m->num_layers >= 1 && m->num_layers <= SN_MAX_LAYERS, "mlp_wide: num_layers=%u outside 1..%d", m->num_layers, SN_MAX_LAYERS);
    SN_REQUIRE(m->activation <= 1u, "mlp_wide: activation must be 0 (relu) or 1 (leaky_relu 0.01)");
    const uint32_t din = m->dims[0], nl = m->num_layers;
    SN_REQUIRE(din >= 1 && din <= 1024, "mlp_wide: input width %u outside 1..1024", din);
    SN_REQUIRE((m->skip_mask & 1u) == 0u, "mlp_wide: a skip connection into layer 0 is not meaningful");
    size_t off = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        SN_REQUIRE(m->weight[l], "mlp_wide: layer %u has no weight", l);
        if (l + 1 < nl) SN_REQUIRE(m->dims[l + 1] == (uint32_t)WIDE, "mlp_wide: hidden width must be %d (layer %u has %u)", WIDE, l, m->dims[l + 1]);
        else SN_REQUIRE(m->dims[l + 1] >= 1 && m->dims[l + 1] <= (uint32_t)WIDE, "mlp_wide: output width %u outside 1..%d", m->dims[l + 1], WIDE);
        WideLayer &L = layers[l];
        const bool skip = ((m->skip_mask >> l) & 1u) != 0u;
        L.uses_h = l > 0 ? 1u : 0u;
        L.x_ks = (l == 0 || skip) ? div_up(din, 16) : 0u;
        L.out = m->dims[l + 1];
        L.mt = div_up(L.out, 32);
        L.has_bias = m->bias[l] ? 1u : 0u;
        L.w_off = (uint32_t)off;
        L.narrow = (l + 1 == nl && L.uses_h && L.x_ks == 0u && L.mt <= 2u) ? 1u : 0u;
        off += L.narrow ? (size_t)(WIDE_HKS / 4) * WIDE_CHUNK_U4 : (size_t)((L.uses_h ? WIDE_HKS : 0) + L.x_ks) * WIDE_CHUNK_U4;
    }
    SN_REQUIRE(off < (1ull << 31), "mlp_wide: packed weights too large");
    *total_u4 = off;
    return SN_OK;
}

// chunk stream of k_mask16 (mlp16.inc): k-steps of 32 inputs, one 32 KiB chunk each; grid_levels > 0: layer 0's input is grid_levels x 8 grid
// features (four levels per k-step) followed by `appended` channels in k-steps of their own (fused mask head)
static int wide16_plan(const sn_mlp_desc *m, uint32_t grid_levels, uint32_t appended, WideLayer *layers, size_t *total_u4) {
    SN_REQUIRE(m->num_layers >= 1 && m->num_layers <= SN_MAX_LAYERS, "mlp16: num_layers=%u outside 1..%d", m->num_layers, SN_MAX_LAYERS);
    SN_REQUIRE(m->activation <= 1u, "mlp16: activation must be 0 (relu) or 1 (leaky_relu 0.01)");
    const uint32_t din = m->dims[0], nl = m->num_layers;
    SN_REQUIRE(din >= 1 && din <= 1024, "mlp16: input width %u outside 1..1024", din);
    SN_REQUIRE((m->skip_mask & 1u) == 0u, "mlp16: a skip connection into layer 0 is not meaningful");
    const uint32_t x_ks = grid_levels ? div_up(grid_levels, 4) + div_up(appended, 32) : div_up(din, 32);
    size_t off = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        SN_REQUIRE(m->weight[l], "mlp16: layer %u has no weight", l);
        if (l + 1 < nl) SN_REQUIRE(m->dims[l + 1] == (uint32_t)WIDE, "mlp16: hidden width must be %d (layer %u has %u)", WIDE, l, m->dims[l + 1]);
        else SN_REQUIRE(m->dims[l + 1] >= 1 && m->dims[l + 1] <= (uint32_t)WIDE, "mlp16: output width %u outside 1..%d", m->dims[l + 1], WIDE);
        WideLayer &L = layers[l];
        const bool skip = ((m->skip_mask >> l) & 1u) != 0u;
        L.uses_h = l > 0 ? 1u : 0u;
        L.x_ks = (l == 0 || skip) ? x_ks : 0u;
        L.out = m->dims[l + 1];
        L.mt = div_up(L.out, 16);
        L.has_bias = m->bias[l] ? 1u : 0u;
        L.w_off = (uint32_t)off;
        L.narrow = (l + 1 == nl && L.uses_h && L.x_ks == 0u && L.out <= 16u) ? 1u : 0u;
        off += L.narrow ? (size_t)W16_CHUNK_U4 : (size_t)((L.uses_h ? W16_HKS : 0) + L.x_ks) * W16_CHUNK_U4;
    }
    SN_REQUIRE(off < (1ull << 31), "mlp16: packed weights too large");
    *total_u4 = off;
    return SN_OK;
}

}  // namespace sn

using namespace sn;

#if SN_WIDE_TRACE
extern "C" int sn_mlp_wide_debug_trace(unsigned long long *out, int n) {
    SN_HIP_OK(hipDeviceSynchronize());
    SN_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_trace), sizeof(unsigned long long) * (size_t)(n < 256 ? n : 256)));
    return SN_OK;
}
#endif

extern "C" int sn_mlp_wide_overflow(int32_t *flag) {
    SN_REQUIRE(flag, "mlp_wide_overflow: NULL output");
    int v = 0;
    const int zero = 0;
    SN_HIP_OK(hipDeviceSynchronize());
    SN_HIP_OK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wide_overflow), sizeof(v)));
    SN_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_wide_overflow), &zero, sizeof(zero)));
    *flag = v;
    return SN_OK;
}

extern "C" size_t sn_mlp_wide_workspace_bytes(const sn_mlp_desc *mlp) {
    WideLayer layers[SN_MAX_LAYERS];
    size_t u4 = 0;
    if (!mlp || wide_plan(mlp, layers, &u4) != SN_OK) return 0;
    return u4 * sizeof(uint4);
}

static int wide_forward_impl(const sn_mlp_desc *mlp, const float *ln_weight, const float *ln_bias, float ln_eps,
                             const float *x, uint32_t N, float *out, void *workspace, size_t workspace_bytes,
                             sn_stream_t stream, float *const *hidden, uint32_t *const *sign_bits = nullptr, const sn_mlp_desc *bwd_of = nullptr) {
    // bwd_of: `mlp` is the TRANSPOSED perceptron of *bwd_of in reverse layer order (sn_mlp_wide_backward_bits): weights are packed transposed, hidden[l] receives
    // the gradient w.r.t. the pre-activation that follows backward layer l, sign_bits[l] are READ (k_mlp_wide_j<.., 2>)
    SN_REQUIRE(mlp, "mlp_wide: mlp is NULL");
    if (N == 0) return SN_OK;
    SN_REQUIRE(x && out && workspace, "mlp_wide: x/out/workspace must be device pointers");
    SN_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), "mlp_wide: LayerNorm needs both weight and bias");
    SN_REQUIRE(table_aligned(workspace) && table_aligned(out), "mlp_wide: workspace/out must be 16-byte aligned");
    PackArgs pa;
    size_t u4 = 0;
    int rc = wide_plan(mlp, pa.layer, &u4);
    if (rc) return rc;
    SN_REQUIRE(workspace_bytes >= u4 * sizeof(uint4), "mlp_wide: workspace too small (%zu bytes, need %zu)", workspace_bytes, u4 * sizeof(uint4));
    const uint32_t nl = mlp->num_layers, din = mlp->dims[0];
    hipStream_t st = (hipStream_t)stream;
    pa.din = din; pa.nl = nl; pa.pack = reinterpret_cast<uint4 *>(workspace); pa.transposed = bwd_of ? 1 : 0; pa.pair_ks = 0;
    uint32_t max_threads = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        pa.w[l] = mlp->weight[l];
        pa.in_dim[l] = bwd_of ? bwd_of->dims[nl - 1 - l]                      // columns of the forward layer's [out, in] weight
                              : (l == 0 ? din : (uint32_t)WIDE) + ((l > 0 && ((mlp->skip_mask >> l) & 1u)) ? din : 0u);
        SN_REQUIRE(l == 0 || mlp->dims[l] == (uint32_t)WIDE, "mlp_wide: dims[%u] must be %d", l, WIDE);
        const uint32_t nks = (pa.layer[l].uses_h ? WIDE_HKS : 0) + pa.layer[l].x_ks;
        const uint32_t th = nks * (uint32_t)WIDE_MT * 64u;
        if (th > max_threads) max_threads = th;
    }
    hipLaunchKernelGGL(k_pack_mlp_wide, dim3(div_up(max_threads, 256), nl), dim3(256), 0, st, pa);
    SN_LAUNCH_CHECK("k_pack_mlp_wide");
    WideArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.x = x; wa.out = out; wa.pack = pa.pack; wa.ln_w = ln_weight; wa.ln_b = ln_bias; wa.ln_eps = ln_eps;
    if (hidden) {
        for (uint32_t l = 0; l + 1 < nl; ++l) {
            SN_REQUIRE(hidden[l] && table_aligned(hidden[l]), "mlp_wide_forward_train: hidden[%u] must be a 16-byte aligned device pointer", l);
            wa.dump[l] = hidden[l];
            if (bwd_of) SN_REQUIRE(sign_bits && sign_bits[l], "mlp_wide_backward_bits: sign bits of a hidden layer are missing");
            if (sign_bits && sign_bits[l]) {
                SN_REQUIRE(table_aligned(sign_bits[l]), "mlp_wide_forward_train: sign_bits[%u] must be 16-byte aligned", l);
                wa.bits[l] = sign_bits[l];
            }
        }
    }
    wa.N = N; wa.din = din; wa.nl = nl; wa.leaky = mlp->activation; wa.total_chunks = (uint32_t)(u4 / WIDE_CHUNK_U4);
    for (uint32_t l = 0; l < nl; ++l) { wa.bias[l] = mlp->bias[l]; wa.layer[l] = pa.layer[l]; }
    // XMODE 1 row stride: a multiple of 4 floats (16-byte reads) with stride/4 odd (the 32 rows of a wave then start in
    // distinct bank groups: conflict-free ds_read_b128)
    uint32_t xs = (din + 3u) & ~3u;
    if (((xs >> 2) & 1u) == 0u) xs += 4u;
    const size_t lds_fixed = (size_t)WIDE_NBUF * WIDE_CHUNK_U4 * sizeof(uint4) + (size_t)SN_MAX_LAYERS * WIDE * sizeof(float);
    const size_t lds_x1 = ((size_t)WIDE_ROWS * xs + 32u) * sizeof(float);       // + slack: the last k-step of the last row reads past xs
    const size_t lds_x2 = (size_t)WIDE_ROWS * din * sizeof(float) + 1024u;      // whole 1 KiB DMA pieces + slack
    const size_t lds_cap = 160u * 1024u;
    // DMA mode: rows must be an odd number of floats apart (bank conflicts) and the array a whole number of 16-byte
    // pieces unless the grid has no partial last tile
    const bool dma_ok = (din & 1u) && lds_fixed + lds_x2 <= lds_cap && (uint64_t)N * din >= 4u &&
                        ((((uint64_t)N * din) & 3u) == 0u || (N % WIDE_ROWS) == 0u);
    const int xmode = dma_ok ? 2 : (lds_fixed + lds_x1 <= lds_cap ? 1 : 0);
    size_t lds = lds_fixed + (xmode == 2 ? lds_x2 : xmode == 1 ? lds_x1 : 0u);
    const size_t out_need = 4u * (32u * (WIDE + 4u) + 64u) * sizeof(float);      // the transposed output tile of 4 waves
    wa.out_lds = (mlp->dims[nl] == (uint32_t)WIDE && lds_cap >= out_need) ? 1u : 0u;
    if (wa.out_lds && lds < out_need) lds = out_need;
    wa.xs = xs;
#ifdef SN_EXPERIMENTS
#define SN_MLP_LAUNCH_OLD(MODE)                                                                                       \
    do {                                                                                                              \
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_mlp_wide<MODE>, dim3(div_up(N, WIDE_ROWS)), dim3(256), lds, st, wa);                     \
    } while (0)
#else
#define SN_MLP_LAUNCH_OLD(MODE) do { } while (0)
#endif
#define SN_MLP_LAUNCH(MODE)                                                                                           \
    do {                                                                                                              \
        if (wide_jit(MODE)) {                                                                                         \
            SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide_j<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k_mlp_wide_j<MODE>, dim3(div_up(N, WIDE_ROWS)), dim3(256), lds, st, wa);               \
        } else SN_MLP_LAUNCH_OLD(MODE);                                                                               \
    } while (0)
#define SN_MLP_LAUNCH_SAVE(MODE)                                                                                      \
    do {                                                                                                              \
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide_j<MODE, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_mlp_wide_j<MODE, false, 1>), dim3(div_up(N, WIDE_ROWS)), dim3(256), lds, st, wa);    \
    } while (0)
#define SN_MLP_LAUNCH_BWD(MODE)                                                                                       \
    do {                                                                                                              \
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide_j<MODE, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_mlp_wide_j<MODE, false, 2>), dim3(div_up(N, WIDE_ROWS)), dim3(256), lds, st, wa);       \
    } while (0)
#ifdef SN_EXPERIMENTS
    if (bwd_of) { if (xmode == 2) SN_MLP_LAUNCH_BWD(2); else if (xmode == 1) SN_MLP_LAUNCH_BWD(1); else SN_MLP_LAUNCH_BWD(0); }
    else
#else
    SN_REQUIRE(!bwd_of, "mlp_wide: the backward on k_mlp_wide_j is compiled into experiments builds only");
#endif
    if (hidden) { if (xmode == 2) SN_MLP_LAUNCH_SAVE(2); else if (xmode == 1) SN_MLP_LAUNCH_SAVE(1); else SN_MLP_LAUNCH_SAVE(0); }
    else if (xmode == 2) SN_MLP_LAUNCH(2); else if (xmode == 1) SN_MLP_LAUNCH(1); else SN_MLP_LAUNCH(0);
#undef SN_MLP_LAUNCH
#undef SN_MLP_LAUNCH_SAVE
#undef SN_MLP_LAUNCH_BWD
    SN_LAUNCH_CHECK("k_mlp_wide");
    return SN_OK;
}

extern "C" int sn_mlp_wide_forward(const sn_mlp_desc *mlp, const float *ln_weight, const float *ln_bias, float ln_eps,
                                   const float *x, uint32_t N, float *out, void *workspace, size_t workspace_bytes,
                                   sn_stream_t stream) {
    return wide_forward_impl(mlp, ln_weight, ln_bias, ln_eps, x, N, out, workspace, workspace_bytes, stream, nullptr);
}

// The training forward on the inference kernel (k_mlp_wide_j: split-fp16 x3 products, fp32 accumulation, ~2^-22 per product) with every
// hidden layer's post-activation output saved as its tiles leave the accumulators.  Measured against the reference's gradients
// (tests/golden/train_c5.npz) the three forwards -- BLAS fp32, fp32 MFMA, this -- give the SAME errors to three digits (7.8e-4 / 1.6e-4 /
// 8.9e-4 relative L2 for the first two weight matrices and the table rows: tools/train_fwd_modes_err.py): the error against the fixture comes
// from elsewhere (the frozen field's 1e-5), not from how the mask MLP's pre-activations are rounded.
extern "C" int sn_mlp_wide_forward_train_f16x3(const sn_mlp_desc *mlp, const float *x, uint32_t N, float *const *hidden, uint32_t *const *sign_bits,
                                               float *out, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    SN_REQUIRE(mlp && mlp->num_layers >= 1, "mlp_wide_forward_train_f16x3: mlp is NULL");
    SN_REQUIRE(mlp->skip_mask == 0u, "mlp_wide_forward_train_f16x3: skip connections are not supported (use the torch layers)");
    SN_REQUIRE(mlp->num_layers == 1 || hidden, "mlp_wide_forward_train_f16x3: hidden is NULL");
    static float *const no_hidden[SN_MAX_LAYERS] = {};
    return wide_forward_impl(mlp, nullptr, nullptr, 0.0f, x, N, out, workspace, workspace_bytes, stream, mlp->num_layers == 1 ? no_hidden : hidden, sign_bits);
}

// Training forward in one kernel, true fp32 on the matrix cores (mlp_f32.inc): opt-in, see the measurements there.
extern "C" int sn_mlp_wide_forward_train(const sn_mlp_desc *mlp, const float *x, uint32_t N, float *const *hidden, float *out, sn_stream_t stream) {
    SN_REQUIRE(mlp, "mlp_wide_forward_train: mlp is NULL");
    const uint32_t nl = mlp->num_layers;
    SN_REQUIRE(nl >= 1 && nl <= SN_MAX_LAYERS, "mlp_wide_forward_train: num_layers=%u outside 1..%d", nl, SN_MAX_LAYERS);
    SN_REQUIRE(mlp->activation <= 1u, "mlp_wide_forward_train: activation must be 0 (relu) or 1 (leaky_relu 0.01)");
    SN_REQUIRE(mlp->skip_mask == 0u, "mlp_wide_forward_train: skip connections are not supported (use the torch layers)");
    SN_REQUIRE(mlp->dims[0] >= 1 && mlp->dims[0] <= 256u, "mlp_wide_forward_train: input width %u outside 1..256", mlp->dims[0]);
    SN_REQUIRE(mlp->dims[nl] >= 1 && mlp->dims[nl] <= 256u, "mlp_wide_forward_train: output width %u outside 1..256", mlp->dims[nl]);
    if (N == 0) return SN_OK;
    SN_REQUIRE(x && out && (nl == 1 || hidden), "mlp_wide_forward_train: NULL pointer");
    F32Args a{};
    a.x = x; a.n_rows = N; a.K0 = mlp->dims[0]; a.nl = nl; a.leaky = mlp->activation;
    for (uint32_t l = 0; l < nl; ++l) {
        SN_REQUIRE(mlp->weight[l] && !mlp->bias[l], "mlp_wide_forward_train: layer %u needs a weight and no bias", l);
        if (l + 1 < nl) {
            SN_REQUIRE(mlp->dims[l + 1] == 256u, "mlp_wide_forward_train: hidden width must be 256 (layer %u has %u)", l, mlp->dims[l + 1]);
            SN_REQUIRE(hidden[l], "mlp_wide_forward_train: hidden[%u] is NULL", l);
        }
        a.w[l] = mlp->weight[l];
        a.N[l] = mlp->dims[l + 1];
        a.h[l] = l + 1 < nl ? hidden[l] : out;
    }
    SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_f32_train), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F32_LDS_BYTES));
    hipLaunchKernelGGL(k_mlp_f32_train, dim3(div_up(N, (uint32_t)F32_ROWS)), dim3(F32_THREADS), F32_LDS_BYTES, (hipStream_t)stream, a);
    SN_LAUNCH_CHECK("k_mlp_f32_train");
    return SN_OK;
}

// Backward-data pass of a 256-wide perceptron in one kernel (the autograd of nerf/network.py:31-66 for the per-sample mask
// head during training, trainer.py:401-428): grad_in = ((grad_out W_{nl-1}) * act'(h_{nl-2}) ... W_0), with the gradient
// w.r.t. every hidden pre-activation written out for the weight-gradient kernels (sn_linear_wgrad).  The same machinery
// as the forward -- transposed formulation, split-fp16 products with fp32 accumulation, weights streamed through LDS --
// run over the transposed weights in reverse layer order; the activation step is the mask from the forward's saved
// outputs, so no rounding difference can flip a LeakyReLU branch.  (Rounds 2-4 held a split-fp16 FORWARD to be ruled out by the same
// argument; measured against the reference's gradients in round 5 it is not -- sn_mlp_wide_forward_train_f16x3 above)
extern "C" size_t sn_mlp_wide_backward_workspace_bytes(const sn_mlp_desc *mlp) {
    if (!mlp || mlp->num_layers < 1 || mlp->num_layers > SN_MAX_LAYERS) return 0;
    sn_mlp_desc b = *mlp;
    const uint32_t nl = mlp->num_layers;
    for (uint32_t l = 0; l <= nl; ++l) b.dims[l] = mlp->dims[nl - l];
    for (uint32_t l = 0; l < nl; ++l) { b.weight[l] = mlp->weight[nl - 1 - l]; b.bias[l] = nullptr; }
    b.skip_mask = 0;
    return sn_mlp_wide_workspace_bytes(&b);
}

static int wide_backward_impl(const sn_mlp_desc *mlp, const float *grad_out, const float *const *hidden, const uint32_t *const *sign_bits, uint32_t N,
                              float *grad_in, float *const *grad_hidden, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    SN_REQUIRE(mlp, "mlp_wide_backward: mlp is NULL");
    if (N == 0) return SN_OK;
    const uint32_t nl = mlp->num_layers;
    SN_REQUIRE(nl >= 2 && nl <= SN_MAX_LAYERS, "mlp_wide_backward: num_layers=%u outside 2..%d", nl, SN_MAX_LAYERS);
    SN_REQUIRE(grad_out && (hidden || sign_bits) && grad_hidden && grad_in && workspace, "mlp_wide_backward: NULL pointer");
    SN_REQUIRE(table_aligned(workspace) && table_aligned(grad_in), "mlp_wide_backward: workspace/grad_in must be 16-byte aligned");
    if (mlp->skip_mask != 0u) { set_error("mlp_wide_backward: skip layers are not supported (use autograd's GEMMs)"); return SN_ERR_UNSUPPORTED; }
    sn_mlp_desc b = *mlp;                                                   // the backward pass as an MLP over the transposed weights
    for (uint32_t l = 0; l <= nl; ++l) b.dims[l] = mlp->dims[nl - l];
    for (uint32_t l = 0; l < nl; ++l) { b.weight[l] = mlp->weight[nl - 1 - l]; b.bias[l] = nullptr; }
    PackArgs pa;
    size_t u4 = 0;
    int rc = wide_plan(&b, pa.layer, &u4);
    if (rc) return rc;
    SN_REQUIRE(workspace_bytes >= u4 * sizeof(uint4), "mlp_wide_backward: workspace too small (%zu bytes, need %zu)", workspace_bytes, u4 * sizeof(uint4));
    hipStream_t st = (hipStream_t)stream;
    pa.din = b.dims[0]; pa.nl = nl; pa.pack = reinterpret_cast<uint4 *>(workspace); pa.transposed = 1; pa.pair_ks = 0;
    uint32_t max_threads = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        pa.w[l] = b.weight[l];
        pa.in_dim[l] = mlp->dims[nl - 1 - l];                                // columns of the forward layer's [out, in] weight
        const uint32_t nks = (pa.layer[l].uses_h ? WIDE_HKS : 0) + pa.layer[l].x_ks;
        const uint32_t th = nks * (uint32_t)WIDE_MT * 64u;
        if (th > max_threads) max_threads = th;
    }
    hipLaunchKernelGGL(k_pack_mlp_wide, dim3(div_up(max_threads, 256), nl), dim3(256), 0, st, pa);
    SN_LAUNCH_CHECK("k_pack_mlp_wide");
    WideArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.x = grad_out; wa.out = grad_in; wa.pack = pa.pack;
    wa.N = N; wa.din = b.dims[0]; wa.nl = nl; wa.leaky = mlp->activation; wa.total_chunks = (uint32_t)(u4 / WIDE_CHUNK_U4);
    for (uint32_t l = 0; l < nl; ++l) wa.layer[l] = pa.layer[l];
    for (uint32_t l = 0; l + 1 < nl; ++l) {                                   // after backward layer l: forward layer nl-2-l's output
        SN_REQUIRE(grad_hidden[nl - 2 - l] && table_aligned(grad_hidden[nl - 2 - l]), "mlp_wide_backward: grad_hidden[%u] must be a 16-byte aligned device pointer", nl - 2 - l);
        if (sign_bits) {
            SN_REQUIRE(sign_bits[nl - 2 - l] && table_aligned(sign_bits[nl - 2 - l]), "mlp_wide_backward_bits: sign_bits[%u] must be a 16-byte aligned device pointer", nl - 2 - l);
            wa.bits[l] = const_cast<uint32_t *>(sign_bits[nl - 2 - l]);
        } else {
            SN_REQUIRE(hidden[nl - 2 - l] && table_aligned(hidden[nl - 2 - l]), "mlp_wide_backward: hidden[%u] must be a 16-byte aligned device pointer", nl - 2 - l);
            wa.mask[l] = hidden[nl - 2 - l];
        }
        wa.dump[l] = grad_hidden[nl - 2 - l];
    }
    const size_t lds = (size_t)WIDE_NBUF * WIDE_CHUNK_U4 * sizeof(uint4) + (size_t)SN_MAX_LAYERS * WIDE * sizeof(float) + 256u;
    if (sign_bits) {
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mlp_wide<5>, dim3(div_up(N, WIDE_ROWS)), dim3(256), lds, st, wa);
    } else {
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mlp_wide<4>, dim3(div_up(N, WIDE_ROWS)), dim3(256), lds, st, wa);
    }
    SN_LAUNCH_CHECK("k_mlp_wide<4>");
    return SN_OK;
}

extern "C" int sn_mlp_wide_backward(const sn_mlp_desc *mlp, const float *grad_out, const float *const *hidden, uint32_t N,
                                    float *grad_in, float *const *grad_hidden, void *workspace, size_t workspace_bytes,
                                    sn_stream_t stream) {
    return wide_backward_impl(mlp, grad_out, hidden, nullptr, N, grad_in, grad_hidden, workspace, workspace_bytes, stream);
}

// The same pass with the units' branches read from the sign bits the SAVE forward wrote (sn_mlp_wide_forward_train_f16x3: 32 bytes per row
// and layer) instead of the [N, 256] fp32 outputs: 268 MB less to read for the mask head's 131 072 rows, and no 1 KiB-per-row loads whose
// latency this kernel (one wave per SIMD) cannot hide.
extern "C" int sn_mlp_wide_backward_bits(const sn_mlp_desc *mlp, const float *grad_out, const uint32_t *const *sign_bits, uint32_t N,
                                         float *grad_in, float *const *grad_hidden, void *workspace, size_t workspace_bytes,
                                         sn_stream_t stream) {
    SN_REQUIRE(sign_bits, "mlp_wide_backward_bits: sign_bits is NULL");
    if (g_wide_bwd_j == 0) return wide_backward_impl(mlp, grad_out, nullptr, sign_bits, N, grad_in, grad_hidden, workspace, workspace_bytes, stream);   // k_mlp_wide<5>: the product path
    // experiments builds: the same pass on k_mlp_wide_j (operands just in time, tiles leave the accumulators under the next MFMAs): the transposed perceptron in
    // reverse layer order.  Measured equal to k_mlp_wide<5> (see g_wide_bwd_j).
    SN_REQUIRE(mlp, "mlp_wide_backward_bits: mlp is NULL");
    if (N == 0) return SN_OK;
    const uint32_t nl = mlp->num_layers;
    SN_REQUIRE(nl >= 2 && nl <= SN_MAX_LAYERS, "mlp_wide_backward_bits: num_layers=%u outside 2..%d", nl, SN_MAX_LAYERS);
    SN_REQUIRE(grad_out && grad_hidden && grad_in && workspace, "mlp_wide_backward_bits: NULL pointer");
    if (mlp->skip_mask != 0u) { set_error("mlp_wide_backward_bits: skip layers are not supported (use autograd's GEMMs)"); return SN_ERR_UNSUPPORTED; }
    sn_mlp_desc b = *mlp;
    for (uint32_t l = 0; l <= nl; ++l) b.dims[l] = mlp->dims[nl - l];
    for (uint32_t l = 0; l < nl; ++l) { b.weight[l] = mlp->weight[nl - 1 - l]; b.bias[l] = nullptr; }
    float *dump_r[SN_MAX_LAYERS] = {};
    uint32_t *bits_r[SN_MAX_LAYERS] = {};
    for (uint32_t l = 0; l + 1 < nl; ++l) {                                   // after backward layer l: forward layer nl-2-l's output
        SN_REQUIRE(grad_hidden[nl - 2 - l] && sign_bits[nl - 2 - l], "mlp_wide_backward_bits: grad_hidden[%u] / sign_bits[%u] is NULL", nl - 2 - l, nl - 2 - l);
        dump_r[l] = grad_hidden[nl - 2 - l];
        bits_r[l] = const_cast<uint32_t *>(sign_bits[nl - 2 - l]);
    }
    return wide_forward_impl(&b, nullptr, nullptr, 0.0f, grad_out, N, grad_in, workspace, workspace_bytes, stream, dump_r, bits_r, mlp);
}

// Mask head in one kernel (renderer.py:304-305, 376-385): k_mlp_wide<3> builds each sample's MLP input -- the C = 8 hash-grid
// levels of its position and the appended geometry channels -- in registers, straight into the B operands of the first
// layer, and composites the per-sample logits with the ray's weights in its epilogue.  Neither the [N*T, 143] input nor
// the [N*T, n_inst] logits exist in memory.
extern "C" size_t sn_rm_mask_head_workspace_bytes(const sn_mlp_desc *mlp) {
    // whichever skeleton takes the call: k_mlp16's stream pads the input k-steps to 32 columns (at most one more k-step per 4 levels + the appended ones)
    const size_t a = sn_mlp_wide_workspace_bytes(mlp);
    WideLayer layers[SN_MAX_LAYERS];
    size_t u4 = 0;
    if (!mlp || a == 0 || wide16_plan(mlp, 0u, 0u, layers, &u4) != SN_OK) return a;
    const size_t b = (u4 + 2u * (size_t)W16_CHUNK_U4) * sizeof(uint4);
    return a > b ? a : b;
}

extern "C" int sn_rm_mask_head(const float *xyzs, const float *extra, const float *weights, uint32_t N, uint32_t T, uint32_t E,
                               float bound, const sn_grid_desc *grid, const sn_mlp_desc *mlp, float *out,
                               void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    SN_REQUIRE(grid && mlp, "mask_head: grid/mlp is NULL");
    if (N == 0) return SN_OK;
    SN_REQUIRE(xyzs && weights && out && workspace && (extra || E == 0), "mask_head: xyzs/extra/weights/out/workspace must be device pointers");
    SN_REQUIRE(table_aligned(workspace) && grid->embeddings && table_aligned(grid->embeddings), "mask_head: workspace / table must be 16-byte aligned");
    SN_REQUIRE(bound > 0.0f, "mask_head: bound must be positive");
    if (!(grid->D == 3 && grid->C == 8 && grid->table_dtype == SN_F32 && grid->L >= 1 && grid->L <= SN_MAX_LEVELS && E <= 16u)) {
        set_error("mask_head: fused path needs a 3-D fp32 grid with level_dim 8 (network.py:104) and <= 16 appended channels "
                  "(got D=%u C=%u dtype=%d E=%u)", grid->D, grid->C, (int)grid->table_dtype, E);
        return SN_ERR_UNSUPPORTED;
    }
    if (!(T >= 1 && T <= 128u && (T & (T - 1u)) == 0u)) {
        set_error("mask_head: samples per ray must be a power of two <= 128 for the in-kernel compositing (got %u)", T);
        return SN_ERR_UNSUPPORTED;
    }
    const uint32_t nl = mlp->num_layers;
    SN_REQUIRE(mlp->dims[0] == grid->L * 8u + E, "mask_head: mlp input width %u != %u grid features + %u appended", mlp->dims[0], grid->L * 8u, E);
    SN_REQUIRE((grid->L & 1u) == 0u || E == 0u, "mask_head: an odd number of levels cannot be followed by appended channels");
    SN_REQUIRE((uint64_t)grid->offsets[grid->L] * 8u < (1ull << 32), "mask_head: table too large for 32-bit row offsets");
    PackArgs pa;
    size_t u4 = 0;
    int rc = wide_plan(mlp, pa.layer, &u4);
    if (rc) return rc;
    if (g_mask_head16 != 0) {
        // k_mask16 (mlp16.inc: 16-row tiles, two waves per SIMD, the tile ahead's first-layer operands made under the hidden layer) takes the
        // reference's mask head -- network.py:118-123: three bias-free layers, n_inst <= 16 outputs, <= 16 appended channels behind <= 16 levels,
        // T >= 4 samples per ray, a grid of the fast-path shape; every other stack keeps k_mlp_wide_j<3> / k_mlp_wide<3> below
        GridLevels gl16;
        rc = build_grid_levels(&gl16, grid->offsets, grid->D, grid->C, grid->L, grid->S, grid->H, grid->gridtype, (int)grid->align_corners, grid->interp);
        if (rc) return rc;
        bool any_bias = false;
        for (uint32_t l = 0; l < nl; ++l) any_bias = any_bias || mlp->bias[l] != nullptr;
        const bool fit16 = g_mask_head16 != 0 && nl == 3u && !any_bias && E <= 16u && mlp->dims[nl] <= 16u && mlp->skip_mask == 0u && T >= 4u && E <= 32u && grid->L <= 16u && levels_fast(gl16) &&
                           (uint64_t)grid->offsets[grid->L] * 32u < (1ull << 32);
        if (fit16) {
            PackArgs p16;
            size_t n16 = 0;
            rc = wide16_plan(mlp, grid->L, E, p16.layer, &n16);
            if (rc) return rc;
            SN_REQUIRE(workspace_bytes >= n16 * sizeof(uint4), "mask_head: workspace too small (%zu bytes, need %zu)", workspace_bytes, n16 * sizeof(uint4));
            SN_REQUIRE((uint64_t)N * T < (1ull << 32), "mask_head: N*T does not fit 32 bits");
            hipStream_t st = (hipStream_t)stream;
            p16.din = mlp->dims[0]; p16.nl = nl; p16.pack = reinterpret_cast<uint4 *>(workspace); p16.transposed = 0;
            p16.pair_ks = div_up(grid->L, 4); p16.pair_levels = grid->L;
            uint32_t max_threads = 0;
            for (uint32_t l = 0; l < nl; ++l) {
                p16.w[l] = mlp->weight[l];
                p16.in_dim[l] = l == 0 ? mlp->dims[0] : (uint32_t)WIDE;
                const uint32_t th = ((p16.layer[l].uses_h ? W16_HKS : 0) + p16.layer[l].x_ks) * (uint32_t)W16_MT * 64u;
                if (th > max_threads) max_threads = th;
            }
            hipLaunchKernelGGL(k_pack_mlp16, dim3(div_up(max_threads, 256), nl), dim3(256), 0, st, p16);
            SN_LAUNCH_CHECK("k_pack_mlp16");
            WideArgs wa;
            memset(&wa, 0, sizeof(wa));
            wa.out = out; wa.pack = p16.pack;
            wa.N = N * T; wa.din = mlp->dims[0]; wa.nl = nl; wa.leaky = mlp->activation; wa.total_chunks = (uint32_t)(n16 / W16_CHUNK_U4);
            for (uint32_t l = 0; l < nl; ++l) { wa.bias[l] = mlp->bias[l]; wa.layer[l] = p16.layer[l]; }
            wa.xyz = xyzs; wa.extra = extra; wa.wts = weights; wa.table = reinterpret_cast<const float *>(grid->embeddings);
            wa.T = T; wa.E = E; wa.bound = bound;
            int e = 0;
            const float mant = frexpf(2.0f * bound, &e);
            wa.inv_den = (mant == 0.5f && e > -100 && e < 100) ? 1.0f / (2.0f * bound) : 0.0f;
            wa.g = gl16;
            // weight ring (2 chunks), staged grid operands of the tile ahead [4][hi | lo][512] uint4, the narrow layer's resident weights, biases, level table
            // ... the appended channels' operand [hi | lo][256] uint4, the level table + the live-tile word
            const size_t lds16 = (size_t)W16Cfg<8>::NBUF * W16_CHUNK_U4 * sizeof(uint4) + 4u * 2u * 512u * sizeof(uint4) + (size_t)W16_HKS * 128u * sizeof(uint4) +
                                 512u * sizeof(uint4) + 4u * SN_MAX_LEVELS * sizeof(uint32_t) + 16u;
            SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mask16<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
            hipLaunchKernelGGL((k_mask16<8>), dim3(div_up(N, 32u)), dim3(512), lds16, st, wa);
            SN_LAUNCH_CHECK("k_mask16");
            return SN_OK;
        }
    }
    if (mlp->dims[nl] > 32u || mlp->skip_mask != 0u) {
        set_error("mask_head: fused path composites at most 32 outputs and no skip layers (got %u outputs, skip mask %u)", mlp->dims[nl], mlp->skip_mask);
        return SN_ERR_UNSUPPORTED;
    }
    SN_REQUIRE(workspace_bytes >= u4 * sizeof(uint4), "mask_head: workspace too small (%zu bytes, need %zu)", workspace_bytes, u4 * sizeof(uint4));
    const uint64_t rows64 = (uint64_t)N * T;
    SN_REQUIRE(rows64 < (1ull << 32), "mask_head: N*T does not fit 32 bits");
    const uint32_t rows = (uint32_t)rows64, width = mlp->dims[0];
    hipStream_t st = (hipStream_t)stream;
    pa.din = width; pa.nl = nl; pa.pack = reinterpret_cast<uint4 *>(workspace); pa.transposed = 0;
    GridLevels gl_probe;
    rc = build_grid_levels(&gl_probe, grid->offsets, grid->D, grid->C, grid->L, grid->S, grid->H, grid->gridtype, (int)grid->align_corners, grid->interp);
    if (rc) return rc;
    // (T is a power of two: the ray-major tiling needs 4 samples of a ray per tile; its branch-free addressing needs the fast-path grid shape)
    const bool jit3 = wide_jit(3) && T >= 4u && levels_fast(gl_probe) && (uint64_t)grid->offsets[grid->L] * 32u < (1ull << 32);
    pa.pair_ks = jit3 ? (grid->L + 1u) >> 1 : 0u;
    uint32_t max_threads = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        pa.w[l] = mlp->weight[l];
        pa.in_dim[l] = l == 0 ? width : (uint32_t)WIDE;
        const uint32_t nks = (pa.layer[l].uses_h ? WIDE_HKS : 0) + pa.layer[l].x_ks;
        const uint32_t th = nks * (uint32_t)WIDE_MT * 64u;
        if (th > max_threads) max_threads = th;
    }
    hipLaunchKernelGGL(k_pack_mlp_wide, dim3(div_up(max_threads, 256), nl), dim3(256), 0, st, pa);
    SN_LAUNCH_CHECK("k_pack_mlp_wide");
    WideArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.out = out; wa.pack = pa.pack;
    wa.N = rows; wa.din = width; wa.nl = nl; wa.leaky = mlp->activation; wa.total_chunks = (uint32_t)(u4 / WIDE_CHUNK_U4);
    for (uint32_t l = 0; l < nl; ++l) { wa.bias[l] = mlp->bias[l]; wa.layer[l] = pa.layer[l]; }
    // <= 32 outputs (the reference's n_inst = 2): the narrow last layer multiplies ONE output tile per k-step instead of the padded pair
    const bool narrow1 = jit3 && nl >= 2u && pa.layer[nl - 1u].narrow != 0u && mlp->dims[nl] <= 32u && g_wide_narrow1 != 0;
    wa.xyz = xyzs; wa.extra = extra; wa.wts = weights; wa.table = reinterpret_cast<const float *>(grid->embeddings);
    wa.T = T; wa.E = E; wa.bound = bound;
    {
        int e = 0;
        const float m = frexpf(2.0f * bound, &e);
        wa.inv_den = (m == 0.5f && e > -100 && e < 100) ? 1.0f / (2.0f * bound) : 0.0f;
    }
    rc = build_grid_levels(&wa.g, grid->offsets, grid->D, grid->C, grid->L, grid->S, grid->H, grid->gridtype, (int)grid->align_corners, grid->interp);
    if (rc) return rc;
    const size_t lds = (size_t)WIDE_NBUF * WIDE_CHUNK_U4 * sizeof(uint4) + (size_t)SN_MAX_LAYERS * WIDE * sizeof(float) + 4u * SN_MAX_LEVELS * sizeof(uint32_t);
    if (jit3) {          // a workgroup = 32 consecutive rays, all their samples (T / 4 tiles of 128 rows)
        if (narrow1) {
            SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide_j<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_mlp_wide_j<3, true>), dim3(div_up(N, 32u)), dim3(256), lds, st, wa);
        } else {
            SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide_j<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_mlp_wide_j<3>, dim3(div_up(N, 32u)), dim3(256), lds, st, wa);
        }
    } else {
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_wide<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mlp_wide<3>, dim3(div_up(rows, WIDE_ROWS)), dim3(256), lds, st, wa);
    }
    SN_LAUNCH_CHECK("k_mlp_wide<3>");
    return SN_OK;
}

// ------------------------------------------------------------------------------------------
// weight gradient of a tiny linear layer: dw[n][k] = sum_m dy[m][n] * x[m][k]
// ------------------------------------------------------------------------------------------
// The radiance / proposal MLPs are 10..64 wide and see 1e5..5e5 samples per training step, so their weight
// gradients are [<=64 x <=64] outputs of a 1e5-long reduction: the BLAS heuristics pick 16x16x512 tiles that take
// ~0.3 ms each (1.3 ms of a 6.4 ms RGB-mode step).  Here a workgroup takes a slab of 128 rows through LDS (all its
// loads in flight at once), every thread keeps a 4x4 register tile of the output, and the slabs are summed by a
// second kernel in a fixed order (deterministic, unlike an atomic accumulation).
namespace sn {

constexpr uint32_t WG_ROWS = 128;    // rows per LDS tile, loaded in one go (many loads in flight)
constexpr uint32_t WG_MAX_SLABS = 1024;   // workgroups = partial results; a workgroup walks tiles blockIdx, blockIdx + grid, ...

__global__ __launch_bounds__(256) void k_linear_wgrad_partial(const float *__restrict__ x, const float *__restrict__ dy, uint32_t M,
                                                              uint32_t K, uint32_t N, float *__restrict__ partial) {
    SN_POISON_ALL();
    __shared__ __attribute__((aligned(16))) float sx[WG_ROWS][64 + 4];
    __shared__ __attribute__((aligned(16))) float sy[WG_ROWS][64 + 4];
    const uint32_t t = threadIdx.x, tn = t >> 4, tk = t & 15u;          // 16 x 16 threads, 4 x 4 outputs each
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    const uint32_t ntiles = (M + WG_ROWS - 1u) / WG_ROWS;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // at most WG_MAX_SLABS workgroups: short second pass
    const uint32_t row0 = tile * WG_ROWS;
    const uint32_t rows = M - row0 < WG_ROWS ? M - row0 : WG_ROWS;
    __syncthreads();
    {   // thread -> (column t % 64, rows t / 64 + 4 i): coalesced along a row, all 2 x 32 loads issued before the first use
        const uint32_t col = t & 63u, r0 = t >> 6;
        float vx[WG_ROWS / 4], vy[WG_ROWS / 4];
#pragma unroll
        for (uint32_t i = 0; i < WG_ROWS / 4u; ++i) {
            const uint32_t r = r0 + 4u * i;
            const uint32_t rr = r < rows ? r : rows - 1u;                  // in range: no branch around the loads
            vx[i] = x[(size_t)(row0 + rr) * K + (col < K ? col : K - 1u)];
            vy[i] = dy[(size_t)(row0 + rr) * N + (col < N ? col : N - 1u)];
        }
#pragma unroll
        for (uint32_t i = 0; i < WG_ROWS / 4u; ++i) {
            const uint32_t r = r0 + 4u * i;
            sx[r][col] = (r < rows && col < K) ? vx[i] : 0.0f;
            sy[r][col] = (r < rows && col < N) ? vy[i] : 0.0f;
        }
    }
    __syncthreads();
#pragma unroll 8
    for (uint32_t r = 0; r < WG_ROWS; ++r) {
        const float4 a = *reinterpret_cast<const float4 *>(&sy[r][4u * tn]);
        const float4 b = *reinterpret_cast<const float4 *>(&sx[r][4u * tk]);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(av[i], bv[j], acc[i][j]);
    }
    }
    float *out = partial + (size_t)blockIdx.x * N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t n = 4u * tn + i, k = 4u * tk + j;
            if (n < N && k < K) out[n * K + k] = acc[i][j];
        }
}

// fold step of k_linear_wgrad_rows: v[0..CNT) per lane, lanes differing in bit D exchange halves; `base` = index of
// v[0] in the output once the lane bits seen so far have chosen their halves.  D = 0: every lane of a group of equal
// `base` holds the finished sums of its CNT outputs; the group's first lane stores them.
template <int CNT, int D>
__device__ __forceinline__ void wgrad_fold(float *v, uint32_t lane, uint32_t base, float *__restrict__ out) {
    if constexpr (D == 0) {
        // lanes that took the same halves at every halving stage are equal here; plain stages left all their lanes equal too
#pragma unroll
        for (int i = 0; i < CNT; ++i) out[base + i] = v[i];      // same value from every lane of the group: benign
    } else if constexpr (CNT % 2 == 0) {
        constexpr int H = CNT / 2;
        const bool up = (lane & (uint32_t)D) != 0u;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float lo = v[i], hi = v[i + H];
            const float r = __shfl_xor(up ? lo : hi, D);
            v[i] = (up ? hi : lo) + r;
        }
        wgrad_fold<H, D / 2>(v, lane, base + (up ? (uint32_t)H : 0u), out);
    } else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) v[i] += __shfl_xor(v[i], D);
        wgrad_fold<CNT, D / 2>(v, lane, base, out);
    }
}

// The proposal MLPs' two layers (10 -> 16 -> 1) see 262 144 / 524 288 samples per step; the tile kernel above keeps
// 12 or 4 of its 256 threads busy on such outputs (69 us for the larger one: 0.8 TB/s).  Here a lane owns whole rows:
// it reads its row of x and of dy (consecutive lanes = consecutive rows = contiguous memory) and keeps the full N x K
// outer-product sum in registers; a wave takes `rows_per_wave` rows, folds its 64 lanes with a butterfly and writes
// one partial result; k_linear_wgrad_sum adds the waves in a fixed order.
template <int K, int N>
__global__ __launch_bounds__(256) void k_linear_wgrad_rows(const float *__restrict__ x, const float *__restrict__ dy, uint32_t M,
                                                           uint32_t rows_per_wave, float *__restrict__ partial) {
    SN_POISON_ALL();
    const uint32_t lane = threadIdx.x & 63u, wg = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t m0 = (uint64_t)wg * rows_per_wave;
    if (m0 >= M) return;
    float acc[N][K];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[n][k] = 0.0f;
    auto load_row = [&](uint64_t m, float (&xv)[K], float (&dv)[N]) {
        if constexpr (K % 2 == 0) {
#pragma unroll
            for (int k = 0; k < K; k += 2) { const float2 t = *reinterpret_cast<const float2 *>(x + m * K + k); xv[k] = t.x; xv[k + 1] = t.y; }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) xv[k] = x[m * K + k];
        }
        if constexpr (N % 4 == 0) {
#pragma unroll
            for (int n = 0; n < N; n += 4) { const float4 t = *reinterpret_cast<const float4 *>(dy + m * N + n); dv[n] = t.x; dv[n + 1] = t.y; dv[n + 2] = t.z; dv[n + 3] = t.w; }
        } else {
#pragma unroll
            for (int n = 0; n < N; ++n) dv[n] = dy[m * N + n];
        }
    };
    // two rows per trip, both rows' loads issued before the first product (a row past the end is read at row M-1 and
    // multiplied by zero: no branch between the loads)
    for (uint32_t r = lane; r < rows_per_wave; r += 128u) {
        const uint64_t ma = m0 + r, mb = ma + 64u;
        const bool va = ma < M, vb = mb < M && r + 64u < rows_per_wave;
        float xa[K], da[N], xb[K], db[N];
        load_row(va ? ma : (uint64_t)M - 1u, xa, da);
        load_row(vb ? mb : (uint64_t)M - 1u, xb, db);
        const float sa = va ? 1.0f : 0.0f, sb = vb ? 1.0f : 0.0f;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float ya = da[n] * sa, yb = db[n] * sb;
#pragma unroll
            for (int k = 0; k < K; ++k) acc[n][k] = __builtin_fmaf(yb, xb[k], __builtin_fmaf(ya, xa[k], acc[n][k]));
        }
    }
    // Fold the 64 lanes.  A plain butterfly is 6 dependent shuffles per output (960 for the 16 x 10 layer, latency-bound:
    // 40 us); instead each stage also halves what a lane is responsible for -- the lane whose bit is clear keeps the
    // first half of its values and receives the partner's first half, the other lane the second half -- so a stage costs
    // CNT/2 independent shuffles: 80 + 40 + 20 + 10 + 5 (+ 5 plain ones once the count is odd) = 160.
    float *out = partial + (size_t)wg * (N * K);
    float *v = &acc[0][0];
    wgrad_fold<N * K, 32>(v, lane, 0u, out);
}

// dw[i] = sum over slabs, in a fixed order: LPO lanes per output stride over the slabs, then a shuffle tree.  The small
// layers have <= 4096 outputs and up to 1024 slabs: with 16 lanes per output each lane walked 64 dependent-latency
// loads (20 us per layer, seven layers per RGB-mode step); 64 lanes per output walk 16.
template <uint32_t LPO>
__global__ __launch_bounds__(256) void k_linear_wgrad_sum(const float *__restrict__ partial, uint32_t nslab, uint32_t NK, float *__restrict__ dw) {
    SN_POISON_ALL();
    constexpr uint32_t OPB = 256u / LPO;                 // outputs per workgroup
    const uint32_t o = blockIdx.x * OPB + threadIdx.x / LPO, s = threadIdx.x % LPO;
    float v = 0.0f;
    if (o < NK)
        for (uint32_t b = s; b < nslab; b += LPO) v += partial[(size_t)b * NK + o];
#pragma unroll
    for (int d = (int)LPO / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, LPO);
    if (o < NK && s == 0u) dw[o] = v;
}


// the same sum for large outputs (NK % 4 == 0): 16 lanes x float4 = 256 contiguous bytes per slab, 16 slab groups per
// workgroup each walking its slabs in order, then the 16 group sums are added in order through LDS
__global__ __launch_bounds__(256) void k_linear_wgrad_sum4(const float4 *__restrict__ partial, uint32_t nslab, uint32_t NK4, float4 *__restrict__ dw) {
    SN_POISON_ALL();
    __shared__ float4 red[16][16];
    const uint32_t ol = threadIdx.x & 15u, sg = threadIdx.x >> 4, o = blockIdx.x * 16u + ol;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (o < NK4)
        for (uint32_t b = sg; b < nslab; b += 16u) {
            const float4 p = partial[(size_t)b * NK4 + o];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
    red[sg][ol] = v;
    __syncthreads();
    if (sg == 0u && o < NK4) {
        float4 t = red[0][ol];
#pragma unroll
        for (int g = 1; g < 16; ++g) { const float4 p = red[g][ol]; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
        dw[o] = t;
    }
}

// ---- the same reduction for wide layers (mask / SAM head MLPs: up to 256 outputs, any fan-in) on the matrix cores ----
// dw = dy^T x is a [N x K] result of an M-long reduction (M = 131072 samples per mask-mode step): BLAS picks 32x64
// macro-tiles, i.e. 32 workgroups walking all of M (0.39 ms per 256 x 256 layer, 0.31 ms for the 2 x 256 one).  Here
// the reduction is split instead: a workgroup owns a slab of rows and the whole output (4 waves x 64 output rows x up
// to 256 columns = all 256 accumulator registers of a lane), feeds v_mfma_f32_32x32x2_f32 straight from coalesced
// global loads (lane l supplies A[i = l%32][k = l/32] = dy[m0 + l/32][i] and B[k][j = l%32] = x[m0 + l/32][j]: rows
// of the row-major operands, no transpose, no LDS), and writes its partial result; k_linear_wgrad_sum adds the slabs
// in a fixed order.  True fp32 products and accumulation.
//   VEC : x rows and dy rows are read as float4 / float2; the MFMA row / column index is then a permutation of the
//         real one (tile column jj of block cb <-> real column 4*jj + cb%4 of a 128-wide group), undone on store.
//   FLAT: N <= 32 (the 2-output last layer): one row block, the four waves split the columns instead.

//   MSPLIT: N, K <= 64 (the radiance MLP's layers): one wave holds the whole 64 x 64 output, so the four waves of a
//         workgroup split the slab's ROWS instead and each writes its own partial result (4 per workgroup).
template <int CBW, bool VEC, bool FLAT, bool MSPLIT = false>
__global__ __launch_bounds__(256, 1) void k_linear_wgrad_mfma(const float *__restrict__ x, const float *__restrict__ dy, uint32_t M,
                                                              uint32_t K, uint32_t N, uint32_t rows_per_slab, float *__restrict__ partial) {
    SN_POISON_ALL();
    constexpr int RBW = FLAT ? 1 : 2;
    // steps (of 2 rows) whose operands are in flight ahead of the matrix cores: ~8k clocks of MFMA work, the latency
    // of a loaded HBM system (4 steps = 4k clocks left the 256 x 256 case waiting on loads: 193 us instead of ~110)
    constexpr int WGM_PF = FLAT ? 32 : (CBW <= 4 ? 16 : 8);
    static_assert(!VEC || (CBW % 4 == 0 && !FLAT), "vector loads cover four column blocks / two row blocks at a time");
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, lo = lane & 31u, hi = lane >> 5;
    static_assert(!MSPLIT || (!FLAT && !VEC && CBW == 2), "row-split form: 64 x 64 outputs, scalar loads");
    const uint32_t r0 = (FLAT || MSPLIT) ? 0u : 64u * wave;                            // first output row of this wave
    const uint32_t c0 = MSPLIT ? 0u : (FLAT ? (blockIdx.y * 4u + wave) : blockIdx.y) * (CBW * 32u);  // first column of this wave
    const uint32_t slab = MSPLIT ? blockIdx.x * 4u + wave : blockIdx.x;                // rows_per_slab is per wave when MSPLIT
    const uint64_t mb64 = (uint64_t)slab * rows_per_slab;
    const uint32_t m_begin = mb64 < M ? (uint32_t)mb64 : M;
    const uint32_t m_end = mb64 + rows_per_slab < M ? m_begin + rows_per_slab : M;
    struct Step { float a[RBW]; float b[CBW]; };
    // Output rows >= N / columns >= K are never stored, so their operands may be anything finite or not: addresses
    // are clamped into the arrays and nothing is selected after the load (a select per loaded value made the
    // compiler wait for every load of an unrolled group at once: 193 us instead of ~120 for the 256 x 256 layer).
    // Only the reduction index needs real masking, and only in the tail of a slab (MASK).
    auto load = [&](Step &s, uint32_t t, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const uint32_t m = m_begin + 2u * t + hi;
        const bool mv = !MASK || m < m_end;
        const size_t mr = mv ? m : m_begin;
        if constexpr (VEC) {
            const uint32_t i = r0 + 2u * lo;
            const float2 v = *reinterpret_cast<const float2 *>(dy + mr * N + (i < N ? i : N - 2u));
            s.a[0] = mv ? v.x : 0.0f; s.a[1] = mv ? v.y : 0.0f;
#pragma unroll
            for (int g = 0; g < CBW / 4; ++g) {
                const uint32_t j = c0 + 128u * g + 4u * lo;
                const float4 w = *reinterpret_cast<const float4 *>(x + mr * K + (j < K ? j : K - 4u));
                s.b[4 * g + 0] = mv ? w.x : 0.0f; s.b[4 * g + 1] = mv ? w.y : 0.0f;
                s.b[4 * g + 2] = mv ? w.z : 0.0f; s.b[4 * g + 3] = mv ? w.w : 0.0f;
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                const uint32_t i = r0 + 32u * rb + lo;
                const float v = dy[mr * N + (i < N ? i : N - 1u)];
                s.a[rb] = mv ? v : 0.0f;
            }
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                const uint32_t j = c0 + 32u * cb + lo;
                const float v = x[mr * K + (j < K ? j : K - 1u)];
                s.b[cb] = mv ? v : 0.0f;
            }
        }
    };
    floatx16 acc[RBW][CBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[rb][cb] = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto fma_step = [&](const Step &cur) {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[rb], cur.b[cb], acc[rb][cb], 0, 0, 0);
    };
    const uint32_t rows = m_end > m_begin ? m_end - m_begin : 0u;
    const uint32_t nmain = (rows / 2u) / WGM_PF * WGM_PF;        // whole steps taken by the pipelined loop
    if (nmain) {
        Step ring[WGM_PF];
#pragma unroll
        for (int u = 0; u < WGM_PF; ++u) load(ring[u], (uint32_t)u, std::false_type{});
        for (uint32_t t = 0; t < nmain; t += WGM_PF) {
#pragma unroll
            for (int u = 0; u < WGM_PF; ++u) {
                // the scheduler would sink each load to its first use (no prefetch left): pin the order
                fma_step(ring[u]);
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t tn = t + WGM_PF + u;
                load(ring[u], tn < nmain ? tn : nmain - 1u, std::false_type{});     // last round: a valid step again, unused
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (uint32_t t = nmain; 2u * t < rows; ++t) {               // fewer than WGM_PF steps, the last one maybe half
        Step cur;
        load(cur, t, std::true_type{});
        fma_step(cur);
    }
    // accumulator register r of lane l holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
    float *out = partial + (size_t)slab * N * K;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t rho = (r & 3) + 8 * (r >> 2) + 4 * hi;
            if constexpr (VEC) {
                const uint32_t i = r0 + 2u * rho + rb;
#pragma unroll
                for (int g = 0; g < CBW / 4; ++g) {
                    const uint32_t j = c0 + 128u * g + 4u * lo;
                    if (i < N && j < K)
                        *reinterpret_cast<float4 *>(out + (size_t)i * K + j) =
                            make_float4(acc[rb][4 * g][r], acc[rb][4 * g + 1][r], acc[rb][4 * g + 2][r], acc[rb][4 * g + 3][r]);
                }
            } else {
                const uint32_t i = r0 + 32u * rb + rho;
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) {
                    const uint32_t j = c0 + 32u * cb + lo;
                    if (i < N && j < K) out[(size_t)i * K + j] = acc[rb][cb][r];
                }
            }
        }
}

constexpr uint32_t WGM_MAX_N = 256;      // 4 waves x 64 output rows
static uint32_t wgm_rows_per_slab(uint32_t M) {      // one slab per CU, at least 64 rows, an even count
    uint32_t r = div_up(M ? M : 1u, 256u);
    if (r < 64u) r = 64u;
    return (r + 1u) & ~1u;
}
}  // namespace sn

extern "C" size_t sn_linear_wgrad_workspace_bytes(uint32_t M, uint32_t K, uint32_t N) {
    if (K == 0 || N == 0 || N > sn::WGM_MAX_N) return 0;
    if (K > 64 || N > 64) return (size_t)sn::div_up(M ? M : 1u, sn::wgm_rows_per_slab(M)) * N * K * sizeof(float);
    const uint32_t tiles = sn::div_up(M ? M : 1u, sn::WG_ROWS);
    return (size_t)((tiles < sn::WG_MAX_SLABS ? tiles : sn::WG_MAX_SLABS) + 3u) * N * K * sizeof(float);   // + 3: whole workgroups of 4 wave-slabs
}

extern "C" int sn_linear_wgrad(const float *x, const float *dy, uint32_t M, uint32_t K, uint32_t N, float *dw,
                               void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    SN_REQUIRE(K >= 1 && N >= 1 && N <= sn::WGM_MAX_N, "linear_wgrad: built for layers of up to %u outputs (got %u x %u)", sn::WGM_MAX_N, N, K);
    SN_REQUIRE(dw, "linear_wgrad: dw is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) { SN_HIP_OK(hipMemsetAsync(dw, 0, (size_t)N * K * sizeof(float), st)); return SN_OK; }
    SN_REQUIRE(x && dy && workspace, "linear_wgrad: x/dy/workspace must be device pointers");
    const size_t need = sn_linear_wgrad_workspace_bytes(M, K, N);
    SN_REQUIRE(workspace_bytes >= need, "linear_wgrad: workspace too small (%zu bytes, need %zu)", workspace_bytes, need);
    if (K > 64 || N > 64) {      // wide layer: matrix cores, split over row slabs
        const uint32_t rps = sn::wgm_rows_per_slab(M), nslab = sn::div_up(M, rps);
        float *part = reinterpret_cast<float *>(workspace);
        const bool vec = K % 4 == 0 && N % 2 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 8 == 0) && ((uintptr_t)part % 16 == 0);
#define SN_WGM(CBW, VEC, FLAT, GY) hipLaunchKernelGGL((sn::k_linear_wgrad_mfma<CBW, VEC, FLAT>), dim3(nslab, GY), dim3(256), 0, st, x, dy, M, K, N, rps, part)
        if (N <= 32) SN_WGM(2, false, true, sn::div_up(K, 256));
        else if (vec) { if (K <= 128) SN_WGM(4, true, false, 1); else SN_WGM(8, true, false, sn::div_up(K, 256)); }   // (two half-width workgroups per CU: slower)
        else if (K <= 64) SN_WGM(2, false, false, 1);
        else if (K <= 128) SN_WGM(4, false, false, 1);
        else if (K <= 160) SN_WGM(5, false, false, 1);            // (the mask head's first layer: 143 inputs)
        else if (K <= 192) SN_WGM(6, false, false, 1);
        else SN_WGM(8, false, false, sn::div_up(K, 256));
#undef SN_WGM
        SN_LAUNCH_CHECK("k_linear_wgrad_mfma");
        if ((N * K) % 4u == 0 && (uintptr_t)dw % 16 == 0 && (uintptr_t)part % 16 == 0)
            hipLaunchKernelGGL(sn::k_linear_wgrad_sum4, dim3(sn::div_up(N * K / 4u, 16)), dim3(256), 0, st, reinterpret_cast<const float4 *>(part), nslab,
                               N * K / 4u, reinterpret_cast<float4 *>(dw));
        else
            hipLaunchKernelGGL(sn::k_linear_wgrad_sum<16>, dim3(sn::div_up(N * K, 16)), dim3(256), 0, st, part, nslab, N * K, dw);
        SN_LAUNCH_CHECK("k_linear_wgrad_sum");
        return SN_OK;
    }
    const uint32_t tiles = sn::div_up(M, sn::WG_ROWS);
    if (((K == 10 && N == 16) || (K == 16 && N == 1)) && (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0) {
        // the proposal MLPs' layers: one lane per row, whole output in registers (at most min(tiles, 1024) partial results)
        uint32_t rpw = sn::div_up(M, sn::WG_MAX_SLABS);
        rpw = rpw < sn::WG_ROWS ? sn::WG_ROWS : ((rpw + 63u) & ~63u);
        const uint32_t nwave = sn::div_up(M, rpw);
        float *part = reinterpret_cast<float *>(workspace);
        if (K == 10) hipLaunchKernelGGL((sn::k_linear_wgrad_rows<10, 16>), dim3(sn::div_up(nwave, 4)), dim3(256), 0, st, x, dy, M, rpw, part);
        else hipLaunchKernelGGL((sn::k_linear_wgrad_rows<16, 1>), dim3(sn::div_up(nwave, 4)), dim3(256), 0, st, x, dy, M, rpw, part);
        SN_LAUNCH_CHECK("k_linear_wgrad_rows");
        hipLaunchKernelGGL(sn::k_linear_wgrad_sum<64>, dim3(sn::div_up(N * K, 4)), dim3(256), 0, st, part, nwave, N * K, dw);
        SN_LAUNCH_CHECK("k_linear_wgrad_sum");
        return SN_OK;
    }
    if (M >= 16384u) {
        // <= 64 x 64 over many rows (the radiance MLP's layers): matrix cores, each wave its own slab of rows
        uint32_t rpw = sn::div_up(M, sn::WG_MAX_SLABS);
        rpw = rpw < sn::WG_ROWS ? sn::WG_ROWS : ((rpw + 1u) & ~1u);
        const uint32_t nblk = sn::div_up(sn::div_up(M, rpw), 4u), nparts = nblk * 4u;
        float *part = reinterpret_cast<float *>(workspace);
        hipLaunchKernelGGL((sn::k_linear_wgrad_mfma<2, false, false, true>), dim3(nblk), dim3(256), 0, st, x, dy, M, K, N, rpw, part);
        SN_LAUNCH_CHECK("k_linear_wgrad_mfma");
        hipLaunchKernelGGL(sn::k_linear_wgrad_sum<64>, dim3(sn::div_up(N * K, 4)), dim3(256), 0, st, part, nparts, N * K, dw);
        SN_LAUNCH_CHECK("k_linear_wgrad_sum");
        return SN_OK;
    }
    const uint32_t nslab = tiles < sn::WG_MAX_SLABS ? tiles : sn::WG_MAX_SLABS;
    hipLaunchKernelGGL(sn::k_linear_wgrad_partial, dim3(nslab), dim3(256), 0, st, x, dy, M, K, N, reinterpret_cast<float *>(workspace));
    SN_LAUNCH_CHECK("k_linear_wgrad_partial");
    hipLaunchKernelGGL(sn::k_linear_wgrad_sum<64>, dim3(sn::div_up(N * K, 4)), dim3(256), 0, st, reinterpret_cast<const float *>(workspace), nslab, N * K, dw);
    SN_LAUNCH_CHECK("k_linear_wgrad_sum");
    return SN_OK;
}
