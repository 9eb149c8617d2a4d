// grid_binned.hip — gradient scatter of the grid encoder without per-corner global atomics and without a sort
// (sn_grid_encode_backward_binned; the reference's scatter is gridencoder.cu:252-349, grid.py:71-95).
//
// Why not the reference's way: device-scope fp32 atomics on MI355X retire at ~2.1e10 lane-ops/s whatever the address
// pattern (tools/ubench/atomics.hip), so 2^D * C atomics per sample-level cost >= 6.7 ms for the mask-field step of
// BASELINE configs[4] (134 M atomics).  Rounds 1-3 sorted the (row, contribution) pairs with a library radix sort; that
// sort was the largest single item of both training steps (0.41 of 2.33 ms, 1.2 of 3.34 ms).  A full ordering is more than
// the scatter needs: a table row only has to meet all of its contributions in ONE place.  LDS is that place:
//
//   a level's rows are cut into BINS of 2^shift[level] consecutive rows, sized on the host so that a bin expects about half
//   of what one workgroup can hold in LDS (E_CAP = 16384 / C contributions): 4 rows per bin on level 0 of the mask grid
//   (256 contributions per row), 512 rows on its hashed levels (2 per row);
//   1. k_bin_count    per (level, block of samples): rows of the 2^D corners -> LDS histogram over the level's bins ->
//                     one global add per non-empty (block, bin)
//   2. k_bin_plan     one workgroup: exclusive scan of the (level, bin) counts -> entry offsets, write cursors and the
//                     work list of step 4 (a bin with more than E_CAP entries is split into several items)
//   3. k_bin_scatter  same geometry as 1: every pair takes a slot of its bin (rank inside the block from an LDS counter,
//                     the block's base from ONE global add per (block, bin)) and writes (row inside the bin: 2 bytes,
//                     w_corner * grad[sample, level, 0..C): the product the reference hands to atomicAdd,
//                     gridencoder.cu:340)
//   4. k_bin_accum    one workgroup per work item: a counting sort of the item's entries by row INSIDE LDS (two integer
//                     LDS atomics per entry), then every row is summed by the thread(s) that own it and written with a
//                     plain store -- or, for an item of a split bin, its partial sums to a slab
//   5. k_bin_merge    split bins only: sums the items' slabs in a fixed order and stores the rows
//
// The first version of step 4 added every contribution into an LDS accumulator with ds_add_f32 and took 0.83 ms for the mask
// grid's 16.8 M pairs: tools/ubench/lds_atomics.hip shows why (profiles/r04/ubench_lds_atomics.txt) -- a wave-wide ds_add_f32
// occupies the CU's LDS for ~194 cycles WHATEVER the address pattern (3 cycles per lane), a ds_add_u32 for <= 32.  Hence
// integer atomics to sort, plain LDS traffic to sum.
// One partition pass instead of three radix passes plus a segmented reduction; every table row has exactly one writer.
// The order in which a row's contributions are added is not fixed (slots are handed out by atomics), so sums may differ in
// the last bits between runs -- as with the reference's atomicAdd.  No host synchronisation, every launch has a static
// grid: the whole backward can sit inside a captured HIP graph.
// grad_embeddings must be zero-initialised by the caller (as for the atomic path, grid.py:83); rows that receive no
// contribution are not written.
#include "sn_common.h"

namespace sn {

constexpr uint32_t BIN_FLOATS = 16128;          // contribution floats one work item holds in LDS (63 KiB; with the counters and the small static arrays just under 80 KiB): E_CAP = 16128 / C entries
constexpr uint32_t BIN_ROWS_MAX = 4096;         // rows per bin (LDS counters of k_bin_accum: 16 KiB; with the 64 KiB above two workgroups per CU)
constexpr uint32_t BIN_MAX_PER_LEVEL = 4096;    // LDS histogram of the count / scatter kernels
constexpr uint32_t SPT = 4;                     // samples per thread of the count / scatter kernels (1024 samples = 8192 pairs per block)
constexpr uint32_t PLAN_THREADS = 1024;

__host__ __device__ constexpr uint32_t e_cap(uint32_t C) { return BIN_FLOATS / C; }

struct BinGeom {
    uint32_t shift[SN_MAX_LEVELS];   // log2(rows per bin) of each level
    uint32_t nb[SN_MAX_LEVELS];      // bins of each level
    uint32_t nbs;                    // stride of the [level][bin] arrays = max bins per level
    uint32_t ecap;                   // entries per work item (E_CAP of this C)
    uint32_t C;
};

struct BinHdr { uint32_t n_items, n_shared_items, n_shared_bins, n_entries; };
struct BinItem { uint32_t level_bin, begin, end, slab_off; };   // slab_off = first float of the item's partial-sum slab, or ~0u: the item owns its bin
struct BinShared { uint32_t level_bin, slab_off, n_slabs, pad; };   // the bin's slabs are consecutive: slab_off + k * (rows per bin * C)

template <uint32_t D>
__device__ __forceinline__ bool pair_rows(const float *__restrict__ inputs, uint32_t b, const GridLevels &g, uint32_t level,
                                          uint32_t (&row)[1u << D], float (&w)[1u << D]) {
    float x01[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        x01[d] = inputs[(size_t)b * D + d];
        oob |= (x01[d] < 0 || x01[d] > 1);
    }
    if (oob) return false;                                        // gridencoder.cu:290: no gradient outside [0,1]
    const uint32_t res = g.res[level], size = g.size[level], mode = g.mode[level];
    float pos[D], deriv[D];
    uint32_t cell[D];
    grid_locate<D>(x01, res, g.align_corners != 0, g.interp, pos, deriv, cell);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        uint32_t p[D];
        float ww = 1.0f;   // gridencoder.cu:315-327
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            const bool up = (idx & (1u << d)) != 0u;
            p[d] = up ? umin(cell[d] + 1, res - 1) : cell[d];
            ww *= up ? pos[d] : 1 - pos[d];
        }
        row[idx] = grid_row<D>(p, res, size, mode);
        w[idx] = ww;
    }
    return true;
}

template <uint32_t D>
__global__ __launch_bounds__(256) void k_bin_count(const float *__restrict__ inputs, uint32_t B, GridLevels g, BinGeom bg,
                                                   uint32_t *__restrict__ counts) {
    SN_POISON_ALL();
    __shared__ uint32_t hist[BIN_MAX_PER_LEVEL];
    const uint32_t level = blockIdx.y, nb = bg.nb[level], shift = bg.shift[level];
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) hist[i] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t s = 0; s < SPT; ++s) {
        const uint32_t b = blockIdx.x * (256u * SPT) + s * 256u + threadIdx.x;
        uint32_t row[1u << D];
        float w[1u << D];
        if (b < B && pair_rows<D>(inputs, b, g, level, row, w)) {
#pragma unroll
            for (uint32_t i = 0; i < (1u << D); ++i) atomicAdd(&hist[row[i] >> shift], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) {
        const uint32_t c = hist[i];
        if (c) atomicAdd(&counts[level * bg.nbs + i], c);
    }
}

// exclusive scan over all (level, bin) counts; emits write cursors and the work lists.  Items of split bins come first
// in the list.  One workgroup; thread t owns the run of bins [t * per, (t + 1) * per); five running sums (entries, items of split
// bins, items of whole bins, split bins, slab floats) are scanned with wave shuffles + one hop through LDS.
__global__ __launch_bounds__(PLAN_THREADS) void k_bin_plan(const uint32_t *__restrict__ counts, uint32_t *__restrict__ cursor, uint32_t total_bins,
                                                           BinHdr *__restrict__ hdr, BinItem *__restrict__ items, BinShared *__restrict__ shared_bins,
                                                           BinGeom bg) {
    SN_POISON_ALL();
    constexpr uint32_t NW = PLAN_THREADS / 64u;
    __shared__ uint32_t s_w[NW][5];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, cap = bg.ecap, nbs = bg.nbs;
    const uint32_t per = (total_bins + PLAN_THREADS - 1u) / PLAN_THREADS;
    const uint32_t lo = t * per < total_bins ? t * per : total_bins, hi = lo + per < total_bins ? lo + per : total_bins;
    uint32_t v[5] = {0u, 0u, 0u, 0u, 0u};            // entries, items of split bins, items of whole bins, split bins, slab floats
    for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {
        uint32_t c8[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) c8[j] = i0 + j < hi ? counts[i0 + j] : 0u;       // eight independent loads in flight
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t c = c8[j];
            v[0] += c;
            if (c > cap) { const uint32_t k = (c + cap - 1u) / cap; v[1] += k; ++v[3]; v[4] += k * (bg.C << bg.shift[(i0 + j) / nbs]); }
            else if (c) ++v[2];
        }
    }
    uint32_t inc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) inc[q] = v[q];
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
#pragma unroll
        for (int q = 0; q < 5; ++q) { const uint32_t u = __shfl_up(inc[q], d); if (lane >= d) inc[q] += u; }
    }
    if (lane == 63u) {
#pragma unroll
        for (int q = 0; q < 5; ++q) s_w[wave][q] = inc[q];
    }
    __syncthreads();
    uint32_t base[5], tot[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        uint32_t bsum = 0, all = 0;
        for (uint32_t w2 = 0; w2 < NW; ++w2) { const uint32_t x = s_w[w2][q]; if (w2 < wave) bsum += x; all += x; }
        base[q] = bsum + inc[q] - v[q];              // exclusive prefix of this thread's run
        tot[q] = all;
    }
    uint32_t off = base[0], ish = base[1], iex = tot[1] + base[2], isb = base[3], isl = base[4];
    for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {
        uint32_t c8[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) c8[j] = i0 + j < hi ? counts[i0 + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t i = i0 + j, c = c8[j];
            if (i >= hi) break;
            cursor[i] = off;
            const uint32_t level = i / nbs, bin = i - level * nbs, lb = (level << 16) | bin;
            if (c > cap) {
                const uint32_t k = (c + cap - 1u) / cap, slab = bg.C << bg.shift[level];
                shared_bins[isb++] = BinShared{lb, isl, k, 0u};
                for (uint32_t q = 0; q < k; ++q) {
                    const uint32_t b0 = off + q * cap, b1 = (q + 1u == k) ? off + c : b0 + cap;
                    items[ish++] = BinItem{lb, b0, b1, isl};
                    isl += slab;
                }
            } else if (c) {
                items[iex++] = BinItem{lb, off, off + c, 0xffffffffu};
            }
            off += c;
        }
    }
    if (t == 0u) *hdr = BinHdr{tot[1] + tot[2], tot[1], tot[3], tot[0]};
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_bin_scatter(const float *__restrict__ inputs, const float *__restrict__ grad, uint32_t B,
                                                     GridLevels g, BinGeom bg, int layout, uint32_t *__restrict__ cursor,
                                                     uint16_t *__restrict__ ekey, float *__restrict__ econtrib) {
    SN_POISON_ALL();
    constexpr uint32_t NC = 1u << D;
    __shared__ uint32_t hist[BIN_MAX_PER_LEVEL];                 // phase 1: pairs of this block per bin; phase 2: the block's first slot per bin
    const uint32_t level = blockIdx.y, nb = bg.nb[level], shift = bg.shift[level];
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) hist[i] = 0u;
    __syncthreads();
    uint32_t row[SPT][NC], rank[SPT][NC];
    float w[SPT][NC];
    bool live[SPT];
#pragma unroll
    for (uint32_t s = 0; s < SPT; ++s) {
        const uint32_t b = blockIdx.x * (256u * SPT) + s * 256u + threadIdx.x;
        live[s] = b < B && pair_rows<D>(inputs, b, g, level, row[s], w[s]);
        if (live[s]) {
#pragma unroll
            for (uint32_t i = 0; i < NC; ++i) rank[s][i] = atomicAdd(&hist[row[s][i] >> shift], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) {
        const uint32_t c = hist[i];
        if (c) hist[i] = atomicAdd(&cursor[level * bg.nbs + i], c);
    }
    __syncthreads();
    const uint32_t mask = (1u << shift) - 1u;
#pragma unroll
    for (uint32_t s = 0; s < SPT; ++s) {
        if (!live[s]) continue;
        const uint32_t b = blockIdx.x * (256u * SPT) + s * 256u + threadIdx.x;
        float gs[C];
        load_row<float, (int)C>(layout == SN_LAYOUT_LBC ? grad + ((size_t)level * B + b) * C : grad + ((size_t)b * g.L + level) * C, gs);
#pragma unroll
        for (uint32_t i = 0; i < NC; ++i) {
            const size_t slot = (size_t)hist[row[s][i] >> shift] + rank[s][i];
            ekey[slot] = (uint16_t)(row[s][i] & mask);
            float *dst = econtrib + slot * C;
            const float ww = w[s][i];
            if constexpr (C % 4 == 0) {
#pragma unroll
                for (uint32_t q = 0; q < C / 4; ++q)
                    reinterpret_cast<float4 *>(dst)[q] = make_float4(ww * gs[4 * q], ww * gs[4 * q + 1], ww * gs[4 * q + 2], ww * gs[4 * q + 3]);
            } else if constexpr (C == 2) {
                *reinterpret_cast<float2 *>(dst) = make_float2(ww * gs[0], ww * gs[1]);
            } else {
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) dst[c] = ww * gs[c];
            }
        }
    }
}

template <uint32_t C>
__device__ __forceinline__ void store_row(float *dst, const float (&v)[C]) {
    if constexpr (C % 4 == 0) {
#pragma unroll
        for (uint32_t q = 0; q < C / 4; ++q) reinterpret_cast<float4 *>(dst)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) dst[c] = v[c];
    }
}

// One work item = up to E_CAP entries of one bin.  LDS: end[rows of the bin] (u32) | sorted[E_CAP][C] (fp32).
//   a. end[r] = number of the item's entries on row r                       (ds_add_u32)
//   b. exclusive scan in place                                             -> end[r] = first sorted slot of row r
//   c. every entry takes slot end[row]++ and parks its contribution there   (ds_add_rtn_u32; afterwards end[r] = one past row r's last slot)
//   d. rows are summed from LDS by the thread(s) that own them: one thread per row when the bin has >= 256 rows, else 256 / rows threads per
//      row (coarse levels: a bin of 4 rows holds ~1000 entries) folded with wave shuffles and, beyond 64 threads per row, one LDS hop.
template <uint32_t C>
__global__ __launch_bounds__(256, 2) void k_bin_accum(const BinHdr *__restrict__ hdr, const BinItem *__restrict__ items, GridLevels g, BinGeom bg,
                                                      const uint16_t *__restrict__ ekey, const float *__restrict__ econtrib,
                                                      float *__restrict__ slabs, float *__restrict__ grad_table) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    if (blockIdx.x >= hdr->n_items) return;
    const BinItem it = items[blockIdx.x];
    const uint32_t level = it.level_bin >> 16, bin = it.level_bin & 0xffffu, shift = bg.shift[level], brows = 1u << shift;
    uint32_t *end = lds_u;
    float *sorted = reinterpret_cast<float *>(lds_u + BIN_ROWS_MAX);
    __shared__ uint32_t wsum[4];
    __shared__ float red[4][C];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // all of this thread's entries (keys and contributions: EPT * C = 63 floats) are requested before anything waits on them:
    // one exposed memory latency per item instead of one per phase
    constexpr uint32_t EPT = (e_cap(C) + 255u) / 256u;
    uint32_t key[EPT];
    float val[EPT][C];
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) {
        const uint32_t e = it.begin + j * 256u + tid;
        const bool in = e < it.end;
        key[j] = in ? (uint32_t)ekey[e] : 0xffffffffu;
        load_row<float, (int)C>(econtrib + (size_t)(in ? e : it.begin) * C, val[j]);
    }
    for (uint32_t i = tid; i < brows; i += 256u) end[i] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) if (key[j] != 0xffffffffu) atomicAdd(&end[key[j]], 1u);
    __syncthreads();
    {   // exclusive scan over brows counters: thread t owns the run [t * per, (t + 1) * per)
        const uint32_t per = (brows + 255u) >> 8, lo = umin(tid * per, brows), hi = umin(lo + per, brows);
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; ++i) sum += end[i];
        uint32_t inc = sum;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up(inc, d); if (lane >= d) inc += v; }
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = inc - sum;
        for (uint32_t w2 = 0; w2 < wave; ++w2) base += wsum[w2];
        for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = end[i]; end[i] = base; base += c; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) {
        if (key[j] == 0xffffffffu) continue;
        const uint32_t slot = atomicAdd(&end[key[j]], 1u);
        store_row<C>(sorted + (size_t)slot * C, val[j]);
    }
    __syncthreads();
    const uint32_t row0 = bin << shift, rows = umin(brows, g.size[level] - row0);
    const bool to_slab = it.slab_off != 0xffffffffu;
    float *out = to_slab ? slabs + it.slab_off : grad_table + ((size_t)g.off[level] + row0) * C;     // slab: [row][C] like the table
    if (brows >= 256u) {
        for (uint32_t r = tid; r < rows; r += 256u) {
            const uint32_t s0 = r ? end[r - 1u] : 0u, s1 = end[r];
            if (s1 == s0 && !to_slab) continue;
            float acc[C];
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
            for (uint32_t s = s0; s < s1; ++s) {
                float v[C];
                load_row<float, (int)C>(sorted + (size_t)s * C, v);
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) acc[c] += v[c];
            }
            store_row<C>(out + (size_t)r * C, acc);
        }
    } else {
        const uint32_t tpr_log2 = 8u - shift, tpr = 1u << tpr_log2;           // threads per row
        const uint32_t r = tid >> tpr_log2, sub = tid & (tpr - 1u);
        const uint32_t s0 = r ? end[r - 1u] : 0u, s1 = end[r];
        float acc[C];
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
        for (uint32_t s = s0 + sub; s < s1; s += tpr) {
            float v[C];
            load_row<float, (int)C>(sorted + (size_t)s * C, v);
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) acc[c] += v[c];
        }
        const uint32_t in_wave = tpr < 64u ? tpr : 64u;
        for (uint32_t d = in_wave >> 1; d >= 1u; d >>= 1) {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) acc[c] += __shfl_xor(acc[c], d);
        }
        if (tpr > 64u) {                                                        // 1 or 2 rows per bin: the row's waves meet in LDS
            if (lane == 0u) {
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) red[wave][c] = acc[c];
            }
            __syncthreads();
            const uint32_t wpr = tpr >> 6, w0 = r * wpr;                         // waves per row, first wave of this row
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) { float t2 = 0.0f; for (uint32_t k = 0; k < wpr; ++k) t2 += red[w0 + k][c]; acc[c] = t2; }
        }
        if (sub == 0u && r < rows && (s1 > s0 || to_slab)) store_row<C>(out + (size_t)r * C, acc);
    }
}

template <uint32_t C>
__global__ __launch_bounds__(256) void k_bin_merge(const BinHdr *__restrict__ hdr, const BinShared *__restrict__ shared_bins, GridLevels g, BinGeom bg,
                                                   const float *__restrict__ slabs, float *__restrict__ grad_table) {
    SN_POISON_ALL();
    const uint32_t nsb = hdr->n_shared_bins;
    for (uint32_t sbi = blockIdx.x; sbi < nsb; sbi += gridDim.x) {
        const BinShared sb = shared_bins[sbi];
        const uint32_t level = sb.level_bin >> 16, bin = sb.level_bin & 0xffffu, shift = bg.shift[level], brows = 1u << shift;
        const uint32_t row0 = bin << shift, rows = umin(brows, g.size[level] - row0);
        float *base = grad_table + ((size_t)g.off[level] + row0) * C;
        const uint32_t stride = brows * C;
        for (uint32_t i = threadIdx.x; i < rows * C; i += 256u) {               // one float per thread: sum over the bin's slabs in slab order
            float v = 0.0f;
            for (uint32_t k = 0; k < sb.n_slabs; ++k) v += slabs[(size_t)sb.slab_off + (size_t)k * stride + i];
            if (v != 0.0f) base[i] = v;
        }
    }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// rows per bin of every level: about half of an item's capacity in expected entries (density = pairs per row of the level), so that
// uneven occupancy rarely splits a bin; power of two; at most BIN_ROWS_MAX rows and BIN_MAX_PER_LEVEL bins
static bool bin_geometry(BinGeom *bg, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t max_level, uint64_t *bins_used, uint64_t *rows_total) {
    bg->ecap = e_cap(C); bg->C = C; bg->nbs = 0;
    *bins_used = 0; *rows_total = 0;
    for (uint32_t l = 0; l < SN_MAX_LEVELS; ++l) { bg->shift[l] = 0; bg->nb[l] = 0; }
    const uint32_t rmax = bg->ecap < BIN_ROWS_MAX ? bg->ecap : BIN_ROWS_MAX;
    for (uint32_t l = 0; l < max_level; ++l) {
        const uint64_t size = (uint64_t)(offsets_host[l + 1] - offsets_host[l]);
        if (size == 0 || size >= (1ull << 31)) return false;
        const double density = (double)B * (double)(1u << D) / (double)size;
        double target = (double)bg->ecap / (2.0 * density);
        uint32_t s = 0;
        while ((2u << s) <= rmax && (double)(2u << s) <= target) ++s;
        while (((size + (1ull << s) - 1) >> s) > BIN_MAX_PER_LEVEL) { if ((2u << s) > rmax) return false; ++s; }
        bg->shift[l] = s;
        bg->nb[l] = (uint32_t)((size + (1ull << s) - 1) >> s);
        if (bg->nb[l] > bg->nbs) bg->nbs = bg->nb[l];
        *bins_used += bg->nb[l];
        *rows_total += size;
    }
    return bg->nbs > 0;
}

struct BinLayout {
    size_t counts, cursor, hdr, items, shared_bins, ekey, econtrib, slabs, total;
    uint32_t max_items, max_shared_bins, total_bins;
};

// worst-case sizes: items <= n / E_CAP + bins; split bins <= n / E_CAP; slab floats <= 2 x the table's floats (a split
// bin's items hold >= E_CAP entries each, and rows per bin <= E_CAP / (2 density))
static BinLayout bin_layout(uint64_t n, uint32_t C, uint32_t levels, uint32_t nbs, uint64_t bins_used, uint64_t rows_total) {
    BinLayout l;
    const uint32_t cap = e_cap(C);
    l.total_bins = levels * nbs;
    l.max_items = (uint32_t)(n / cap + bins_used + 1);
    l.max_shared_bins = (uint32_t)(n / cap + 1);
    size_t o = 0;
    l.counts = o; o += align256((size_t)l.total_bins * 4);
    l.cursor = o; o += align256((size_t)l.total_bins * 4);
    l.hdr = o; o += 256;
    l.items = o; o += align256((size_t)l.max_items * sizeof(BinItem));
    l.shared_bins = o; o += align256((size_t)l.max_shared_bins * sizeof(BinShared));
    l.ekey = o; o += align256((size_t)n * 2);
    l.econtrib = o; o += align256((size_t)n * C * 4);
    l.slabs = o; o += align256((size_t)(2 * rows_total * C + 2 * (uint64_t)BIN_FLOATS) * 4);
    l.total = o + 256;
    return l;
}

}  // namespace sn

using namespace sn;

extern "C" {

size_t sn_grid_backward_binned_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, const int32_t *offsets_host) {
    if (D < 2 || D > 3 || C == 0 || C > 32 || (C & (C - 1)) != 0 || L == 0 || L > SN_MAX_LEVELS || !offsets_host) return 0;
    if (max_level > L) max_level = L;
    const uint64_t n = (uint64_t)B * max_level * (1u << D);
    if (n == 0 || n >= (1ull << 31)) return 0;
    BinGeom bg;
    uint64_t bins_used = 0, rows_total = 0;
    if (!bin_geometry(&bg, offsets_host, B, D, C, max_level, &bins_used, &rows_total)) return 0;   // a level beyond 4096 bins of 4096 rows: use the atomic path
    if (2 * rows_total * C + 2 * (uint64_t)BIN_FLOATS >= (1ull << 32)) return 0;                      // slab offsets are 32-bit
    return bin_layout(n, C, max_level, bg.nbs, bins_used, rows_total).total;
}

int sn_grid_encode_backward_binned(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                   float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    if (B == 0 || max_level == 0) return SN_OK;
    SN_REQUIRE(grad && inputs && grad_embeddings && workspace, "grid_encode_backward_binned: NULL device pointer");
    SN_REQUIRE(layout == SN_LAYOUT_LBC || layout == SN_LAYOUT_BLC, "grid_encode_backward_binned: bad layout %d", layout);
    if (D != 3 && D != 2) { set_error("grid_encode_backward_binned: D=%u not instantiated (use sn_grid_encode_backward)", D); return SN_ERR_UNSUPPORTED; }
    if (C == 0 || C > 32 || (C & (C - 1)) != 0) { set_error("grid_encode_backward_binned: C=%u not instantiated", C); return SN_ERR_UNSUPPORTED; }
    GridLevels g;
    int rc = build_grid_levels(&g, offsets_host, D, C, L, S, H, gridtype, align_corners, interp);
    if (rc) return rc;
    if (max_level > L) max_level = L;
    const uint64_t n64 = (uint64_t)B * max_level * (1u << D);
    SN_REQUIRE(n64 < (1ull << 31), "grid_encode_backward_binned: %llu contributions exceed 2^31", (unsigned long long)n64);
    BinGeom bg;
    uint64_t bins_used = 0, rows_total = 0;
    if (!bin_geometry(&bg, offsets_host, B, D, C, max_level, &bins_used, &rows_total) || 2 * rows_total * C + 2 * (uint64_t)BIN_FLOATS >= (1ull << 32)) {
        set_error("grid_encode_backward_binned: a level needs more than %u bins of %u rows (or the table exceeds 2^31 floats); use sn_grid_encode_backward",
                  BIN_MAX_PER_LEVEL, BIN_ROWS_MAX);
        return SN_ERR_UNSUPPORTED;
    }
    const BinLayout lay = bin_layout(n64, C, max_level, bg.nbs, bins_used, rows_total);
    SN_REQUIRE(table_aligned(grad) && table_aligned(workspace), "grid_encode_backward_binned: grad / workspace must be 16-byte aligned");
    if (workspace_bytes < lay.total) { set_error("grid_encode_backward_binned: workspace too small (%zu bytes, need %zu)", workspace_bytes, lay.total); return SN_ERR_WORKSPACE; }
    char *w = reinterpret_cast<char *>(workspace);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + lay.counts), *cursor = reinterpret_cast<uint32_t *>(w + lay.cursor);
    BinHdr *hdr = reinterpret_cast<BinHdr *>(w + lay.hdr);
    BinItem *items = reinterpret_cast<BinItem *>(w + lay.items);
    BinShared *shared_bins = reinterpret_cast<BinShared *>(w + lay.shared_bins);
    uint16_t *ekey = reinterpret_cast<uint16_t *>(w + lay.ekey);
    float *econtrib = reinterpret_cast<float *>(w + lay.econtrib), *slabs = reinterpret_cast<float *>(w + lay.slabs);
    hipStream_t st = (hipStream_t)stream;
    SN_HIP_OK(hipMemsetAsync(counts, 0, (size_t)lay.total_bins * 4, st));
    const dim3 gs(div_up(B, 256u * SPT), max_level), blk(256);
    if (D == 3) hipLaunchKernelGGL((k_bin_count<3>), gs, blk, 0, st, inputs, B, g, bg, counts);
    else hipLaunchKernelGGL((k_bin_count<2>), gs, blk, 0, st, inputs, B, g, bg, counts);
    SN_LAUNCH_CHECK("k_bin_count");
    hipLaunchKernelGGL(k_bin_plan, dim3(1), dim3(PLAN_THREADS), 0, st, counts, cursor, lay.total_bins, hdr, items, shared_bins, bg);
    SN_LAUNCH_CHECK("k_bin_plan");
    const size_t lds = (size_t)(BIN_ROWS_MAX + BIN_FLOATS) * sizeof(float);      // 80 KiB: two workgroups per CU
    const dim3 ga(lay.max_items), gm(lay.max_shared_bins < 2048u ? lay.max_shared_bins : 2048u);
#define SN_BIN_C(DD, CC)                                                                                                         \
    do {                                                                                                                         \
        hipLaunchKernelGGL((k_bin_scatter<DD, CC>), gs, blk, 0, st, inputs, grad, B, g, bg, layout, cursor, ekey, econtrib);     \
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accum<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_bin_accum<CC>), ga, blk, lds, st, hdr, items, g, bg, ekey, econtrib, slabs, grad_embeddings);      \
        hipLaunchKernelGGL((k_bin_merge<CC>), gm, blk, 0, st, hdr, shared_bins, g, bg, slabs, grad_embeddings);                  \
    } while (0)
#define SN_BIN_D(DD)                                                                                                             \
    switch (C) { case 1: SN_BIN_C(DD, 1); break; case 2: SN_BIN_C(DD, 2); break; case 4: SN_BIN_C(DD, 4); break;                 \
                 case 8: SN_BIN_C(DD, 8); break; case 16: SN_BIN_C(DD, 16); break; default: SN_BIN_C(DD, 32); break; }
    if (D == 3) { SN_BIN_D(3) } else { SN_BIN_D(2) }
#undef SN_BIN_D
#undef SN_BIN_C
    SN_LAUNCH_CHECK("k_bin_scatter / k_bin_accum / k_bin_merge");
    return SN_OK;
}

}  // extern "C"
