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

#ifndef SN_BIN_FLOATS
#define SN_BIN_FLOATS 12096
#define SN_BIN_ROWS_MAX 1024
#define SN_BIN_WGS 3
#define SN_BIN_FILL 1.3
#endif
constexpr uint32_t BIN_FLOATS = SN_BIN_FLOATS;          // contribution floats one work item holds in LDS (63 KiB; with the counters and the small static arrays just under 80 KiB): E_CAP = 16128 / C entries
constexpr uint32_t BIN_ROWS_MAX = SN_BIN_ROWS_MAX;         // rows per bin (LDS counters of k_bin_accum: 16 KiB; with the 64 KiB above two workgroups per CU)
constexpr uint32_t BIN_MAX_PER_LEVEL = 4096;    // LDS histogram of the count / scatter kernels
constexpr uint32_t SPT = 4;                     // samples per thread of the count / scatter kernels (1024 samples = 8192 pairs per block)
constexpr uint32_t PLAN_THREADS = 256;
constexpr uint32_t SMALL_BIN_FLOATS = 256;      // split bins up to this size flush with global atomics instead of slabs + merge

__host__ __device__ constexpr uint32_t e_cap(uint32_t C) { return BIN_FLOATS / C; }

#ifndef SN_BIN_TRACE
#define SN_BIN_TRACE 0       // diagnostics build: a few workgroups of k_bin_scatter / k_bin_accum record the shader clock at every phase (sn_bin_debug_trace)
#endif
#if SN_BIN_TRACE
__device__ unsigned long long g_bin_trace[2][16][16];     // [kernel][traced workgroup][phase]
__device__ __forceinline__ void bin_trace(uint32_t kernel, uint32_t wg, uint32_t n_wg, uint32_t slot) {
    if (threadIdx.x != 0 || slot >= 16u) return;
    const uint32_t stride = n_wg / 16u ? n_wg / 16u : 1u;
    if (wg % stride == 0u && wg / stride < 16u) g_bin_trace[kernel][wg / stride][slot] = __builtin_readcyclecounter();
}
#define BIN_TRACE(k, wg, n, slot) bin_trace(k, wg, n, slot)
#else
#define BIN_TRACE(k, wg, n, slot) ((void)0)
#endif

struct BinGeom {
    uint32_t shift[SN_MAX_LEVELS];   // log2(rows per bin) of each level
    uint32_t nb[SN_MAX_LEVELS];      // bins of each level
    uint32_t boff[SN_MAX_LEVELS + 1]; // first slot of each level in the bin-indexed arrays (counts, cursors): levels back to back
    uint32_t ecap;                   // entries per work item (E_CAP of this C)
    uint32_t C;
};

int g_bin_pull = -1;            // -1: by C (host code below); 0 / 1: sn_debug_set("bin_pull", v) in experiments builds (A/B)

struct BinHdr { uint32_t n_items, n_shared_items, n_shared_bins, n_entries; };
struct BinItem { uint32_t level_bin, begin, end, slab_off; };   // slab_off = first float of the item's partial-sum slab, or ~0u: the item owns its bin
struct BinShared { uint32_t level_bin, slab_off, n_slabs, item0, begin, end, pad0, pad1; };   // a split bin: n_slabs items (first one = item item0 of the work list) of E_CAP
                                                                                            // entries each over [begin, end); slabs consecutive: slab_off + k * (rows per bin * C)

// FAST (D = 3, grids of the fused kernels' shape -- levels_fast(): hashed levels of power-of-two size, dense levels over all three dimensions,
// align_corners = False, linear interpolation): the 8 rows come from 6 partial terms (corner_offsets: 2 full-rate 24-bit multiplies per
// level) instead of 8 calls of the generic grid_row (7 quarter-rate v_mul_lo_u32 per corner: 56 per sample-level).  Same rows, same weights.
template <uint32_t D, bool FAST>
__device__ __forceinline__ void pair_rows_at(const float (&x01)[D], const GridLevels &g, uint32_t level, uint32_t (&row)[1u << D], float (&w)[1u << D]);

template <uint32_t D, bool FAST>
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
    pair_rows_at<D, FAST>(x01, g, level, row, w);
    return true;
}

template <uint32_t D, bool FAST>
__device__ __forceinline__ void pair_rows_at(const float (&x01)[D], const GridLevels &g, uint32_t level, uint32_t (&row)[1u << D], float (&w)[1u << D]) {
    const uint32_t res = g.res[level], size = g.size[level], mode = g.mode[level];
    if constexpr (FAST && D == 3) {
        float pos[3];
        uint32_t cell[3];
        locate_linear(x01, res, pos, cell);
        if (mode & 1u) corner_offsets<1, 1u>(cell, res, size, mode, row);          // (the level is the block's: a uniform branch)
        else corner_offsets<0, 1u>(cell, res, size, mode, row);
        const float wx[2] = {1.0f - pos[0], pos[0]}, wy[2] = {1.0f - pos[1], pos[1]}, wz[2] = {1.0f - pos[2], pos[2]};
#pragma unroll
        for (uint32_t idx = 0; idx < 8u; ++idx) w[idx] = (wx[idx & 1u] * wy[(idx >> 1) & 1u]) * wz[idx >> 2];     // gridencoder.cu:315-327: ((1 wx) wy) wz
        return;
    } else {
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
    }
}

__global__ __launch_bounds__(256) void k_bin_zero(uint32_t *__restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = 0u;
}

template <uint32_t D, bool FAST>
__global__ __launch_bounds__(256) void k_bin_count(const float *__restrict__ inputs, uint32_t B, GridLevels g, BinGeom bg,
                                                   uint32_t *__restrict__ counts, uint32_t *__restrict__ blkcnt, uint32_t total_bins) {
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
        if (b < B && pair_rows<D, FAST>(inputs, b, g, level, row, w)) {
#pragma unroll
            for (uint32_t i = 0; i < (1u << D); ++i) atomicAdd(&hist[row[i] >> shift], 1u);
        }
    }
    __syncthreads();
    if (blkcnt) {                 // the block's histogram row, zeros included (k_bin_colscan turns the column of a bin into the blocks' first slots)
        for (uint32_t i = threadIdx.x; i < nb; i += 256u) blkcnt[(size_t)blockIdx.x * total_bins + bg.boff[level] + i] = hist[i];
        return;
    }
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) {
        const uint32_t c = hist[i];
        if (c) atomicAdd(&counts[bg.boff[level] + i], c);
    }
}

// Per-(block, bin) counts -> the block's first slot inside the bin (exclusive running sum down the column of the bin) and the bin's total.
// Why (round 5): handing out slots with one global atomic per non-empty (block, bin) -- 2 x 1 M atomics for the mask-field step, on 8 k
// addresses -- cost ~50 us in k_bin_count + the scatter at the measured ceiling of device-scope atomics (2.1e10 lane-ops/s); the matrix is
// 4 MB written, read, rewritten and read again.  A workgroup = 32 bins x 8 runs of blocks; coalesced over bins.
__global__ __launch_bounds__(256) void k_bin_colscan(uint32_t *__restrict__ blkcnt, uint32_t n_blk, uint32_t total_bins, uint32_t *__restrict__ counts) {
    __shared__ uint32_t part[8][32];
    const uint32_t bx = threadIdx.x & 31u, ch = threadIdx.x >> 5, bin = blockIdx.x * 32u + bx;
    const uint32_t per = (n_blk + 7u) >> 3, k0 = umin(ch * per, n_blk), k1 = umin(k0 + per, n_blk);
    uint32_t sum = 0;
    if (bin < total_bins) for (uint32_t k = k0; k < k1; ++k) sum += blkcnt[(size_t)k * total_bins + bin];
    part[ch][bx] = sum;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8u; ++q) { const uint32_t v = part[q][bx]; if (q < ch) base += v; all += v; }
    if (bin >= total_bins) return;
    if (ch == 0u) counts[bin] = all;
    for (uint32_t k = k0; k < k1; ++k) {
        uint32_t *p = blkcnt + (size_t)k * total_bins + bin;
        const uint32_t c = *p;
        *p = base;
        base += c;
    }
}

// Exclusive scan over all (level, bin) counts -> entry offsets, write cursors and the work lists (items of split bins first).  Five running
// sums: entries, items of split bins, items of whole bins, split bins, slab floats.  Two launches of PLAN_BLOCK_BINS-bin workgroups (one
// 1024-thread workgroup doing all of it kept a single CU busy for 55-70 us): k_bin_plan_sums leaves every workgroup's totals,
// k_bin_plan_emit turns them into its base, scans its own bins with wave shuffles and writes.  A thread owns PLAN_PER consecutive bins.
constexpr uint32_t PLAN_PER = 8, PLAN_BLOCK_BINS = PLAN_THREADS * PLAN_PER;

struct PlanTables { uint32_t boff[SN_MAX_LEVELS + 1], slab[SN_MAX_LEVELS]; };

// per-lane indexing of a kernel-argument array is a dependent memory load per access: the level table goes to LDS, and a thread's run of
// bins is consecutive, so its level only ever steps forward.  A split bin's items hand their partial sums over through slabs when the bin
// is large, and add them to the table with atomics when it is small (<= SMALL_BIN_FLOATS floats: coarse levels, where one row gets
// thousands of contributions and a bin splits into hundreds of items).
__device__ __forceinline__ void plan_tables(PlanTables &t, const BinGeom &bg) {
    if (threadIdx.x <= SN_MAX_LEVELS) t.boff[threadIdx.x] = bg.boff[threadIdx.x];
    if (threadIdx.x < SN_MAX_LEVELS) { const uint32_t f = bg.C << bg.shift[threadIdx.x]; t.slab[threadIdx.x] = f > SMALL_BIN_FLOATS ? f : 0u; }
    __syncthreads();
}

__device__ __forceinline__ void plan_thread_sums(const uint32_t *__restrict__ counts, uint32_t total_bins, const BinGeom &bg, const PlanTables &tb,
                                                 uint32_t lo, uint32_t (&c8)[PLAN_PER], uint32_t &lvl0, uint32_t (&v)[5]) {
#pragma unroll
    for (uint32_t j = 0; j < PLAN_PER; ++j) c8[j] = lo + j < total_bins ? counts[lo + j] : 0u;      // independent loads, in flight together
    lvl0 = 0;
    while (lvl0 + 1u < SN_MAX_LEVELS && lo >= tb.boff[lvl0 + 1u]) ++lvl0;
    uint32_t lvl = lvl0;
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = 0u;
#pragma unroll
    for (uint32_t j = 0; j < PLAN_PER; ++j) {
        const uint32_t c = c8[j];
        if (lo + j < total_bins) while (lo + j >= tb.boff[lvl + 1u]) ++lvl;
        v[0] += c;
        if (c > bg.ecap) { const uint32_t k = (c + bg.ecap - 1u) / bg.ecap; v[1] += k; ++v[3]; v[4] += k * tb.slab[lvl]; }
        else if (c) ++v[2];
    }
}

// workgroup-wide sums (total) and this thread's exclusive prefix (excl) of five values: wave shuffles + one hop through LDS
__device__ __forceinline__ void plan_block_scan(const uint32_t (&v)[5], uint32_t (&excl)[5], uint32_t (&total)[5]) {
    constexpr uint32_t NW = PLAN_THREADS / 64u;
    __shared__ uint32_t s_w[NW][5];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
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
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        uint32_t bsum = 0, all = 0;
        for (uint32_t w2 = 0; w2 < NW; ++w2) { const uint32_t x = s_w[w2][q]; if (w2 < wave) bsum += x; all += x; }
        excl[q] = bsum + inc[q] - v[q];
        total[q] = all;
    }
}

__global__ __launch_bounds__(PLAN_THREADS) void k_bin_plan_sums(const uint32_t *__restrict__ counts, uint32_t total_bins, BinGeom bg, uint32_t *__restrict__ block_sums) {
    SN_POISON_ALL();
    __shared__ PlanTables tb;
    plan_tables(tb, bg);
    uint32_t c8[PLAN_PER], lvl0, v[5], excl[5], total[5];
    plan_thread_sums(counts, total_bins, bg, tb, blockIdx.x * PLAN_BLOCK_BINS + threadIdx.x * PLAN_PER, c8, lvl0, v);
    plan_block_scan(v, excl, total);
    if (threadIdx.x < 5u) block_sums[blockIdx.x * 5u + threadIdx.x] = total[threadIdx.x];
}

__global__ __launch_bounds__(PLAN_THREADS) void k_bin_plan_emit(const uint32_t *__restrict__ counts, uint32_t *__restrict__ cursor, uint32_t total_bins,
                                                                const uint32_t *__restrict__ block_sums, BinHdr *__restrict__ hdr, BinItem *__restrict__ items,
                                                                BinShared *__restrict__ shared_bins, BinGeom bg, uint32_t slab_cap,
                                                                float *__restrict__ grad_table) {
    SN_POISON_ALL();
    __shared__ PlanTables tb;
    plan_tables(tb, bg);
    const uint32_t lo = blockIdx.x * PLAN_BLOCK_BINS + threadIdx.x * PLAN_PER;
    uint32_t c8[PLAN_PER], lvl, v[5], excl[5], total[5], base[5], all[5];
    plan_thread_sums(counts, total_bins, bg, tb, lo, c8, lvl, v);
    plan_block_scan(v, excl, total);
#pragma unroll
    for (int q = 0; q < 5; ++q) { base[q] = 0u; all[q] = 0u; }
    for (uint32_t bq = 0; bq < gridDim.x; ++bq) {              // (at most 128 workgroups' totals: uniform loads)
#pragma unroll
        for (int q = 0; q < 5; ++q) { const uint32_t x = block_sums[bq * 5u + q]; all[q] += x; if (bq < blockIdx.x) base[q] += x; }
    }
    // The slab region is sized from a proven worst case (slab_floats_bound), so this cannot trigger; if it ever did, no item is emitted (nothing
    // is written past the workspace) and the gradient's first element is poisoned so that the step fails loudly instead of silently.
    if (all[4] > slab_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0u) { *hdr = BinHdr{0u, 0u, 0u, 0u}; grad_table[0] = __builtin_nanf(""); }
        return;
    }
    uint32_t off = base[0] + excl[0], ish = base[1] + excl[1], iex = all[1] + base[2] + excl[2], isb = base[3] + excl[3], isl = base[4] + excl[4];
#pragma unroll
    for (uint32_t j = 0; j < PLAN_PER; ++j) {
        const uint32_t i = lo + j, c = c8[j];
        if (i >= total_bins) break;
        while (i >= tb.boff[lvl + 1u]) ++lvl;
        cursor[i] = off;
        const uint32_t lb = (lvl << 16) | (i - tb.boff[lvl]);
        if (c > bg.ecap) {                       // split bin: one record, its k items are implicit (k_bin_accum finds the record of item i by bisection)
            const uint32_t k = (c + bg.ecap - 1u) / bg.ecap, slab = tb.slab[lvl];
            shared_bins[isb++] = BinShared{lb, slab ? isl : 0xffffffffu, k, ish, off, off + c, 0u, 0u};
            ish += k;
            isl += k * slab;
        } else if (c) {
            items[iex++] = BinItem{lb, off, off + c, 0xffffffffu};
        }
        off += c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0u) *hdr = BinHdr{all[1] + all[2], all[1], all[3], all[0]};
}

template <uint32_t D, uint32_t C, bool FAST>
__global__ __launch_bounds__(256) void k_bin_scatter(const float *__restrict__ inputs, const float *__restrict__ grad, uint32_t B,
                                                     GridLevels g, BinGeom bg, int layout, uint32_t gstride, uint32_t *__restrict__ cursor,
                                                     const uint32_t *__restrict__ blkbase, uint32_t total_bins,
                                                     uint16_t *__restrict__ ekey, float *__restrict__ econtrib) {
    SN_POISON_ALL();
    constexpr uint32_t NC = 1u << D;
    __shared__ uint32_t hist[BIN_MAX_PER_LEVEL];                 // phase 1: pairs of this block per bin; phase 2: the block's first slot per bin
    const uint32_t level = blockIdx.y, nb = bg.nb[level], shift = bg.shift[level];
    const uint32_t t_wg = blockIdx.y * gridDim.x + blockIdx.x, t_n = gridDim.x * gridDim.y;
    (void)t_wg; (void)t_n;
    BIN_TRACE(0, t_wg, t_n, 0);
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) hist[i] = 0u;
    __syncthreads();
    uint32_t row[SPT][NC], rank[SPT][NC];
    float w[SPT][NC];
    bool live[SPT];
#pragma unroll
    for (uint32_t s = 0; s < SPT; ++s) {
        const uint32_t b = blockIdx.x * (256u * SPT) + s * 256u + threadIdx.x;
        live[s] = b < B && pair_rows<D, FAST>(inputs, b, g, level, row[s], w[s]);
        if (live[s]) {
#pragma unroll
            for (uint32_t i = 0; i < NC; ++i) rank[s][i] = atomicAdd(&hist[row[s][i] >> shift], 1u);
        }
    }
    BIN_TRACE(0, t_wg, t_n, 1);
    __syncthreads();
    BIN_TRACE(0, t_wg, t_n, 2);
    for (uint32_t i = threadIdx.x; i < nb; i += 256u) {
        const uint32_t c = hist[i];
        if (c) hist[i] = blkbase ? cursor[bg.boff[level] + i] + blkbase[(size_t)blockIdx.x * total_bins + bg.boff[level] + i]
                                 : atomicAdd(&cursor[bg.boff[level] + i], c);
    }
    __syncthreads();
    BIN_TRACE(0, t_wg, t_n, 3);
    const uint32_t mask = (1u << shift) - 1u;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t s = 0; s < SPT; ++s) {
        const uint32_t b = blockIdx.x * (256u * SPT) + s * 256u + threadIdx.x;
        float gs[C];
        if (live[s]) load_row<float, (int)C>(layout == SN_LAYOUT_LBC ? grad + ((size_t)level * B + b) * C : grad + (size_t)b * gstride + (size_t)level * C, gs);
        else {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) gs[c] = 0.0f;
        }
#pragma unroll
        for (uint32_t i = 0; i < NC; ++i) {
            const uint32_t slot = live[s] ? hist[row[s][i] >> shift] + rank[s][i] : 0xffffffffu;      // (n < 2^31: slots fit 32 bits)
            if (live[s]) ekey[slot] = (uint16_t)(row[s][i] & mask);
            const float ww = w[s][i];
            if constexpr (C == 8) {
                // a pair's 32-byte row leaves through TWO NEIGHBOURING LANES (16 bytes each): a store instruction then touches 32 distinct rows
                // instead of 64 -- what a scattered store costs follows the number of lines it touches (tools/ubench/gathers.hip).  Round k:
                // lane j writes half (j & 1) of the pair that lane 32 k + (j >> 1) owns.
                float v[8];
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c) v[c] = ww * gs[c];
#pragma unroll
                for (uint32_t k = 0; k < 2u; ++k) {
                    const int src = (int)(32u * k + (lane >> 1));
                    const uint32_t sl = (uint32_t)__shfl((int)slot, src);
                    float o[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) {
                        const float lo = __shfl(v[q], src), hi = __shfl(v[4 + q], src);
                        o[q] = (lane & 1u) ? hi : lo;
                    }
                    if (sl != 0xffffffffu)
                        *reinterpret_cast<float4 *>(econtrib + (size_t)sl * 8u + (lane & 1u) * 4u) = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else {
                if (!live[s]) continue;
                float *dst = econtrib + (size_t)slot * C;
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
#if SN_BIN_TRACE
    __builtin_amdgcn_s_waitcnt(0);
#endif
    BIN_TRACE(0, t_wg, t_n, 4);
}

// PULL form of step 3 (round 5): the entry is a REFERENCE -- {sample | row-in-bin << 22, corner weight}, 8 bytes -- instead of the 2-byte row
// and the C-float product.  k_bin_pull then fetches the sample's gradient row itself and multiplies.  Why: with C = 8 the products are 32
// bytes per pair -- 537 MB written and read again for the 16.8 M pairs of the mask-field step -- and the 2-byte rows, scattered one by one,
// each cost a 32-byte sector (counters: 921 MB written by k_bin_scatter against 570 MB of payload).  References are 134 MB each way.
// Scattered 8-byte stores would be amplified just like the rows, so the block sorts its references by bin in LDS first (ranks and a
// block-local scan of the histogram) and writes them out in runs: a (block, bin) run is consecutive in the bin's slot range, so a wave's
// store covers a few whole runs.  512 threads x 2 samples = the 1024 samples per block of k_bin_count (same blockIdx.x -> same matrix row).
constexpr uint32_t REF_THREADS = 512, REF_SPT = 256u * SPT / REF_THREADS;
constexpr uint32_t REF_WINDOW = 4096;                           // positions of the block's sorted order staged in LDS at a time
constexpr uint32_t REF_KEY_SHIFT = 22;                           // sample index below, row inside the bin above: B < 2^22, rows per bin <= 2^10
static_assert(BIN_ROWS_MAX <= (1u << (32u - REF_KEY_SHIFT)), "row-in-bin must fit above the sample index");

template <uint32_t D, bool FAST>
__global__ __launch_bounds__(REF_THREADS, 2) void k_bin_refs(const float *__restrict__ inputs, uint32_t B, GridLevels g, BinGeom bg,
                                                  uint32_t *__restrict__ cursor, const uint32_t *__restrict__ blkbase, uint32_t total_bins,
                                                  uint2 *__restrict__ eref) {
    SN_POISON_ALL();
    constexpr uint32_t NC = 1u << D, NPW = REF_WINDOW;
    static_assert(256u * SPT <= 1024u && BIN_MAX_PER_LEVEL <= 4096u && BIN_ROWS_MAX <= 1024u, "the LDS record packs bin (12 bits), row in bin (10) and sample in block (10)");
    __shared__ uint32_t hist[BIN_MAX_PER_LEVEL];                 // pairs of this block per bin -> the block's first slot of the bin
    __shared__ uint32_t lbase[BIN_MAX_PER_LEVEL];                // first position of the bin in the block's sorted order
    __shared__ uint32_t wsum[REF_THREADS / 64u];
    extern __shared__ __attribute__((aligned(16))) uint32_t rec[];   // [NPW] records (bin << 20 | row in bin << 10 | sample in block) then [NPW] weights: a window of the block's pairs sorted by bin
    float *wrec = reinterpret_cast<float *>(rec + NPW);
    const uint32_t level = blockIdx.y, nb = bg.nb[level], shift = bg.shift[level], kmask = (1u << shift) - 1u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t t_wg = blockIdx.y * gridDim.x + blockIdx.x, t_n = gridDim.x * gridDim.y;
    (void)t_wg; (void)t_n;
    BIN_TRACE(0, t_wg, t_n, 0);
    for (uint32_t i = tid; i < nb; i += REF_THREADS) hist[i] = 0u;
    __syncthreads();
    uint32_t row[REF_SPT][NC], rank[REF_SPT][NC];
    float w[REF_SPT][NC];
    bool live[REF_SPT];
#pragma unroll
    for (uint32_t s = 0; s < REF_SPT; ++s) {
        const uint32_t b = blockIdx.x * (256u * SPT) + s * REF_THREADS + tid;
        live[s] = b < B && pair_rows<D, FAST>(inputs, b, g, level, row[s], w[s]);
        if (live[s]) {
#pragma unroll
            for (uint32_t i = 0; i < NC; ++i) rank[s][i] = atomicAdd(&hist[row[s][i] >> shift], 1u);
        }
    }
    BIN_TRACE(0, t_wg, t_n, 1);
    __syncthreads();
    BIN_TRACE(0, t_wg, t_n, 2);
    uint32_t total;
    {   // block-local exclusive scan of the histogram (thread t owns a run of bins), and the block's slots from the bins' cursors
        const uint32_t per = (nb + REF_THREADS - 1u) / REF_THREADS, lo = umin(tid * per, nb), hi = umin(lo + per, nb);
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; ++i) sum += hist[i];
        uint32_t inc = sum;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up(inc, d); if (lane >= d) inc += v; }
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = inc - sum;
        total = 0;
#pragma unroll
        for (uint32_t w2 = 0; w2 < REF_THREADS / 64u; ++w2) { const uint32_t v = wsum[w2]; if (w2 < wave) base += v; total += v; }
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t c = hist[i];
            lbase[i] = base;
            base += c;
            if (c) hist[i] = blkbase ? cursor[bg.boff[level] + i] + blkbase[(size_t)blockIdx.x * total_bins + bg.boff[level] + i]
                                     : atomicAdd(&cursor[bg.boff[level] + i], c);
        }
    }
    __syncthreads();
    BIN_TRACE(0, t_wg, t_n, 3);
    // the sorted order leaves through a WINDOW of NPW positions at a time (two rounds for a full block): 32 KiB of records instead of 64, so that
    // two of these workgroups fit a CU (each is a chain of barrier-separated phases; alone on a CU nothing overlaps them)
    const uint32_t b0 = blockIdx.x * (256u * SPT);
    for (uint32_t p0 = 0; p0 < total; p0 += NPW) {
#pragma unroll
        for (uint32_t s = 0; s < REF_SPT; ++s) {
            if (!live[s]) continue;
#pragma unroll
            for (uint32_t i = 0; i < NC; ++i) {
                const uint32_t bn = row[s][i] >> shift, q = lbase[bn] + rank[s][i] - p0;
                if (q < NPW) {
                    rec[q] = (bn << 20) | ((row[s][i] & kmask) << 10) | (s * REF_THREADS + tid);
                    wrec[q] = w[s][i];
                }
            }
        }
        __syncthreads();
        const uint32_t nq = umin(NPW, total - p0);
        for (uint32_t q = tid; q < nq; q += REF_THREADS) {
            const uint32_t v = rec[q], bn = v >> 20;
            eref[hist[bn] + (p0 + q - lbase[bn])] = make_uint2((b0 + (v & 1023u)) | (((v >> 10) & 1023u) << REF_KEY_SHIFT), __float_as_uint(wrec[q]));
        }
        __syncthreads();
    }
#if SN_BIN_TRACE
    __builtin_amdgcn_s_waitcnt(0);
#endif
    BIN_TRACE(0, t_wg, t_n, 4);
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
// The phases are device functions shared by the two kernels that feed them: k_bin_accum (entries = products, one item per workgroup) and
// k_bin_pull (entries = references, persistent workgroups that fetch the next item's operands under the current item's phases).
struct BinPull { const float *grad; const uint2 *eref; uint32_t B; int layout; uint32_t gstride; };      // gstride: floats between samples' rows ([B, L*C] layout)

__device__ __forceinline__ BinItem bin_item_of(uint32_t i, const BinHdr &h, const BinItem *__restrict__ items, const BinShared *__restrict__ shared_bins, const BinGeom &bg) {
    if (i >= h.n_shared_items) return items[i];
    uint32_t lo = 0, hi = h.n_shared_bins;          // an item of a split bin: bisect the records' first-item indices (ascending)
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (shared_bins[mid].item0 <= i) lo = mid; else hi = mid; }
    const BinShared sb = shared_bins[lo];
    const uint32_t j = i - sb.item0, b0 = sb.begin + j * bg.ecap;
    return BinItem{sb.level_bin, b0, umin(b0 + bg.ecap, sb.end),
                   sb.slab_off == 0xffffffffu ? 0xfffffffeu : sb.slab_off + j * (bg.C << bg.shift[sb.level_bin >> 16])};   // ...fe: small split bin, atomics
}

// a + b: afterwards end[r] = first sorted slot of row r.  Starts with a barrier (a persistent workgroup's previous item is still being summed).
template <uint32_t EPT>
__device__ __forceinline__ void bin_count_scan(uint32_t *end, uint32_t brows, const uint32_t (&key)[EPT], uint32_t *wsum, uint32_t t_wg, uint32_t t_n) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    (void)t_wg; (void)t_n;
    __syncthreads();
    for (uint32_t i = tid; i < brows; i += 256u) end[i] = 0u;
    __syncthreads();
    BIN_TRACE(1, t_wg, t_n, 1);
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) if (key[j] != 0xffffffffu) atomicAdd(&end[key[j]], 1u);
    BIN_TRACE(1, t_wg, t_n, 2);
    __syncthreads();
    BIN_TRACE(1, t_wg, t_n, 3);
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
    BIN_TRACE(1, t_wg, t_n, 4);
}

// d (after a barrier): sums and stores.
template <uint32_t C>
__device__ __forceinline__ void bin_sum_store(const BinItem &it, const GridLevels &g, const BinGeom &bg, const uint32_t *end, const float *sorted,
                                              float (*red)[C], float *__restrict__ slabs, float *__restrict__ grad_table) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t level = it.level_bin >> 16, bin = it.level_bin & 0xffffu, shift = bg.shift[level], brows = 1u << shift;
    __syncthreads();
    const uint32_t row0 = bin << shift, rows = umin(brows, g.size[level] - row0);
    const bool to_slab = it.slab_off < 0xfffffffeu, atomic_out = it.slab_off == 0xfffffffeu;
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
            if (atomic_out) {
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) if (acc[c] != 0.0f) unsafeAtomicAdd(out + (size_t)r * C + c, acc[c]);
            } else store_row<C>(out + (size_t)r * C, acc);
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
        if (sub == 0u && r < rows && (s1 > s0 || to_slab)) {
            if (atomic_out) {                     // one of several items of a small bin: its partial row joins the others' in the table
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) if (acc[c] != 0.0f) unsafeAtomicAdd(out + (size_t)r * C + c, acc[c]);
            } else store_row<C>(out + (size_t)r * C, acc);
        }
    }
}

template <uint32_t C>
__global__ __launch_bounds__(256, SN_BIN_WGS) void k_bin_accum(const BinHdr *__restrict__ hdr, const BinItem *__restrict__ items, const BinShared *__restrict__ shared_bins,
                                                      GridLevels g, BinGeom bg,
                                                      const uint16_t *__restrict__ ekey, const float *__restrict__ econtrib,
                                                      float *__restrict__ slabs, float *__restrict__ grad_table) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    const BinHdr h = *hdr;
    if (blockIdx.x >= h.n_items) return;
    const BinItem it = bin_item_of(blockIdx.x, h, items, shared_bins, bg);
    BIN_TRACE(1, blockIdx.x, h.n_items, 0);
    const uint32_t level = it.level_bin >> 16, brows = 1u << bg.shift[level];
    uint32_t *end = lds_u;
    float *sorted = reinterpret_cast<float *>(lds_u + BIN_ROWS_MAX);
    __shared__ uint32_t wsum[4];
    __shared__ float red[4][C];
    const uint32_t tid = threadIdx.x;
    // all of this thread's entries (keys and contributions: EPT * C floats) are requested before anything waits on them:
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
    bin_count_scan<EPT>(end, brows, key, wsum, blockIdx.x, h.n_items);
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) {
        if (key[j] == 0xffffffffu) continue;
        const uint32_t slot = atomicAdd(&end[key[j]], 1u);
        store_row<C>(sorted + (size_t)slot * C, val[j]);
    }
    BIN_TRACE(1, blockIdx.x, h.n_items, 5);
    bin_sum_store<C>(it, g, bg, end, sorted, red, slabs, grad_table);
#if SN_BIN_TRACE
    __builtin_amdgcn_s_waitcnt(0);
#endif
    BIN_TRACE(1, blockIdx.x, h.n_items, 7);
}

// References in, rows out (entries made by k_bin_refs): per entry one gather of the sample's gradient row (C floats), one multiply by the
// stored corner weight -- the product the reference hands to atomicAdd (gridencoder.cu:340) -- then the same LDS phases as k_bin_accum.
// (A persistent, software-pipelined form -- next item's rows requested under the current item's summation -- was built and measured slower
// than letting three resident workgroups per CU overlap each other: 228 vs 207 us, profiles/r05/bin_stats_*.txt.)
template <uint32_t C>
__global__ __launch_bounds__(256, SN_BIN_WGS) void k_bin_pull(const BinHdr *__restrict__ hdr, const BinItem *__restrict__ items, const BinShared *__restrict__ shared_bins,
                                                     GridLevels g, BinGeom bg, BinPull pull, float *__restrict__ slabs, float *__restrict__ grad_table) {
    SN_POISON_ALL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    constexpr uint32_t EPT = (e_cap(C) + 255u) / 256u;
    const BinHdr h = *hdr;
    if (blockIdx.x >= h.n_items) return;
    const BinItem it = bin_item_of(blockIdx.x, h, items, shared_bins, bg);
    BIN_TRACE(1, blockIdx.x, h.n_items, 0);
    const uint32_t level = it.level_bin >> 16, brows = 1u << bg.shift[level];
    uint32_t *end = lds_u;
    float *sorted = reinterpret_cast<float *>(lds_u + BIN_ROWS_MAX);
    __shared__ uint32_t wsum[4];
    __shared__ float red[4][C];
    const uint32_t tid = threadIdx.x;
    uint2 ref[EPT];
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) {
        const uint32_t e = it.begin + j * 256u + tid;
        ref[j] = e < it.end ? pull.eref[e] : make_uint2(0xffffffffu, 0u);
    }
    uint32_t key[EPT];
    float val[EPT][C];
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) {
        const bool in = ref[j].x != 0xffffffffu;
        const uint32_t b = in ? ref[j].x & ((1u << REF_KEY_SHIFT) - 1u) : 0u;
        key[j] = in ? ref[j].x >> REF_KEY_SHIFT : 0xffffffffu;
        if (in) load_row<float, (int)C>(pull.layout == SN_LAYOUT_LBC ? pull.grad + ((size_t)level * pull.B + b) * C : pull.grad + (size_t)b * pull.gstride + (size_t)level * C, val[j]);
    }
    bin_count_scan<EPT>(end, brows, key, wsum, blockIdx.x, h.n_items);
#pragma unroll
    for (uint32_t j = 0; j < EPT; ++j) {
        if (key[j] == 0xffffffffu) continue;
        const uint32_t slot = atomicAdd(&end[key[j]], 1u);
        const float ww = __uint_as_float(ref[j].y);
        float v[C];
#pragma unroll
        for (uint32_t q = 0; q < C; ++q) v[q] = ww * val[j][q];
        store_row<C>(sorted + (size_t)slot * C, v);
    }
    BIN_TRACE(1, blockIdx.x, h.n_items, 5);
    bin_sum_store<C>(it, g, bg, end, sorted, red, slabs, grad_table);
#if SN_BIN_TRACE
    __builtin_amdgcn_s_waitcnt(0);
#endif
    BIN_TRACE(1, blockIdx.x, h.n_items, 7);
}

// A split bin's rows = the sum of its items' slabs.  Coarse bins are small (4 rows x 8 channels) and split into MANY items, fine bins large
// and split into few: the 256 threads are dealt out as (floats of the bin, rounded up to a power of two F <= 256) x (256 / F slab subsets);
// partial sums meet in LDS.  Slab order within a subset is fixed, so the result does not depend on scheduling.
template <uint32_t C>
__global__ __launch_bounds__(256) void k_bin_merge(const BinHdr *__restrict__ hdr, const BinShared *__restrict__ shared_bins, GridLevels g, BinGeom bg,
                                                   const float *__restrict__ slabs, float *__restrict__ grad_table) {
    SN_POISON_ALL();
    __shared__ float part[256];
    const uint32_t nsb = hdr->n_shared_bins;
    for (uint32_t sbi = blockIdx.x; sbi < nsb; sbi += gridDim.x) {
        const BinShared sb = shared_bins[sbi];
        if (sb.slab_off == 0xffffffffu) continue;                              // small bin: its items added their rows with atomics
        const uint32_t level = sb.level_bin >> 16, bin = sb.level_bin & 0xffffu, shift = bg.shift[level], brows = 1u << shift;
        const uint32_t row0 = bin << shift, rows = umin(brows, g.size[level] - row0);
        float *base = grad_table + ((size_t)g.off[level] + row0) * C;
        const uint32_t stride = brows * C, nf = rows * C;
        uint32_t F = 1;
        while (F < nf && F < 256u) F <<= 1;
        const uint32_t subsets = 256u / F, sub = threadIdx.x / F, fi = threadIdx.x & (F - 1u);
        for (uint32_t i0 = 0; i0 < nf; i0 += F) {
            const uint32_t i = i0 + fi;
            float v = 0.0f;
            if (i < nf) {
                const float *src = slabs + (size_t)sb.slab_off + i;
                for (uint32_t k0 = sub; k0 < sb.n_slabs; k0 += 8u * subsets) {          // eight slabs' loads in flight, summed in slab order
                    float t8[8];
#pragma unroll
                    for (uint32_t q = 0; q < 8u; ++q) { const uint32_t k = k0 + q * subsets; t8[q] = k < sb.n_slabs ? src[(size_t)k * stride] : 0.0f; }
#pragma unroll
                    for (uint32_t q = 0; q < 8u; ++q) v += t8[q];
                }
            }
            if (subsets > 1u) {
                part[threadIdx.x] = v;
                __syncthreads();
                if (sub == 0u) for (uint32_t q = 1; q < subsets; ++q) v += part[q * F + fi];
                __syncthreads();
            }
            if (sub == 0u && i < nf && v != 0.0f) base[i] = v;
        }
    }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// rows per bin of every level: about half of an item's capacity in expected entries (density = pairs per row of the level), so that
// uneven occupancy rarely splits a bin; power of two; at most BIN_ROWS_MAX rows and BIN_MAX_PER_LEVEL bins
// *slab_floats: worst case of the slab region.  A split bin holds c > E_CAP entries and becomes ceil(c / E_CAP) <= c / E_CAP + 1 items with one
// slab of (C << shift) floats each; a level holds P = B 2^D entries, so its split bins' items number <= P / E_CAP + min(bins, P / E_CAP).  (The
// round-4 bound, 2 x the table, assumed rows per bin <= E_CAP / (2 density); the 4096-bins-per-level cap raises the shift beyond that for large B.)
static bool bin_geometry(BinGeom *bg, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t max_level, uint64_t *bins_used, uint64_t *slab_floats) {
    bg->ecap = e_cap(C); bg->C = C;
    for (uint32_t l = 0; l <= SN_MAX_LEVELS; ++l) bg->boff[l] = 0;
    *bins_used = 0; *slab_floats = 0;
    const uint64_t pairs = (uint64_t)B << D;
    for (uint32_t l = 0; l < SN_MAX_LEVELS; ++l) { bg->shift[l] = 0; bg->nb[l] = 0; }
    const uint32_t rmax = bg->ecap < BIN_ROWS_MAX ? bg->ecap : BIN_ROWS_MAX;
    for (uint32_t l = 0; l < max_level; ++l) {
        const uint64_t size = (uint64_t)(offsets_host[l + 1] - offsets_host[l]);
        if (size == 0 || size >= (1ull << 31)) return false;
        const double density = (double)B * (double)(1u << D) / (double)size;
        double target = (double)bg->ecap / (SN_BIN_FILL * density);
        uint32_t s = 0;
        while ((2u << s) <= rmax && (double)(2u << s) <= target) ++s;
        while (((size + (1ull << s) - 1) >> s) > BIN_MAX_PER_LEVEL) { if ((2u << s) > rmax) return false; ++s; }
        bg->shift[l] = s;
        bg->nb[l] = (uint32_t)((size + (1ull << s) - 1) >> s);
        bg->boff[l + 1] = bg->boff[l] + bg->nb[l];
        *bins_used += bg->nb[l];
        const uint64_t slab = (uint64_t)C << s, by_entries = pairs / bg->ecap;
        if (slab > SMALL_BIN_FLOATS) *slab_floats += (by_entries + (by_entries < bg->nb[l] ? by_entries : bg->nb[l])) * slab;
    }
    for (uint32_t l = max_level; l < SN_MAX_LEVELS; ++l) bg->boff[l + 1] = bg->boff[max_level];
    return bg->boff[max_level] > 0;
}

struct BinLayout {
    size_t counts, cursor, hdr, block_sums, items, shared_bins, ekey, econtrib, slabs, blkcnt, total;
    uint32_t n_blk;            // 0: slots from global atomics (the per-block matrix would be too large)
    uint32_t max_items, max_shared_bins, total_bins;
};

// worst-case sizes: items <= n / E_CAP + bins; split bins <= n / E_CAP; slab floats: bin_geometry's bound
constexpr uint64_t BLK_MATRIX_MAX = 16ull << 20;    // (blocks x bins) entries of the per-block count matrix: 64 MiB

static BinLayout bin_layout(uint64_t n, uint32_t C, uint64_t bins_used, uint64_t slab_floats, uint32_t B) {
    BinLayout l;
    const uint32_t cap = e_cap(C);
    l.total_bins = (uint32_t)bins_used;
    l.max_items = (uint32_t)(n / cap + bins_used + 1);
    l.max_shared_bins = (uint32_t)(n / cap + 1);
    size_t o = 0;
    l.counts = o; o += align256((size_t)l.total_bins * 4);
    l.cursor = o; o += align256((size_t)l.total_bins * 4);
    l.hdr = o; o += 256;
    l.block_sums = o; o += align256((size_t)(l.total_bins / (PLAN_THREADS * 8u) + 1) * 5 * 4);
    l.items = o; o += align256((size_t)l.max_items * sizeof(BinItem));
    l.shared_bins = o; o += align256((size_t)l.max_shared_bins * sizeof(BinShared));
    l.ekey = o; o += align256((size_t)n * 2);
    l.econtrib = o; o += align256((size_t)n * C * 4);
    l.slabs = o; o += align256((size_t)(slab_floats + 64) * 4);
    const uint64_t nblk = div_up(B, 256u * SPT);
    l.n_blk = nblk * bins_used <= BLK_MATRIX_MAX ? (uint32_t)nblk : 0u;
    l.blkcnt = o; o += align256((size_t)l.n_blk * l.total_bins * 4);
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
    uint64_t bins_used = 0, slab_floats = 0;
    if (!bin_geometry(&bg, offsets_host, B, D, C, max_level, &bins_used, &slab_floats)) return 0;   // a level beyond 4096 bins of 4096 rows: use the atomic path
    if (slab_floats >= 0xffffff00ull) return 0;                                                       // slab offsets are 32-bit: use the atomic path
    return bin_layout(n, C, bins_used, slab_floats, B).total;
}

int sn_grid_encode_backward_binned(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                   float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    return sn_grid_encode_backward_binned_rows(grad, 0u, inputs, offsets_host, grad_embeddings, B, D, C, L, max_level, S, H, gridtype, align_corners, interp,
                                               layout, workspace, workspace_bytes, stream);
}

int sn_grid_encode_backward_binned_rows(const float *grad, uint32_t grad_row_stride, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                        float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                        int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream) {
    if (B == 0 || max_level == 0) return SN_OK;
    SN_REQUIRE(grad_row_stride == 0u || (layout == SN_LAYOUT_BLC && grad_row_stride >= L * C), "grid_encode_backward_binned: grad_row_stride %u needs the [B, L*C] layout and >= L*C = %u floats", grad_row_stride, L * C);
    const uint32_t gstride = grad_row_stride ? grad_row_stride : L * C;
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
    uint64_t bins_used = 0, slab_floats = 0;
    if (!bin_geometry(&bg, offsets_host, B, D, C, max_level, &bins_used, &slab_floats) || slab_floats >= 0xffffff00ull) {
        set_error("grid_encode_backward_binned: a level needs more than %u bins of %u rows (or the split bins' slabs exceed 2^32 floats); use sn_grid_encode_backward",
                  BIN_MAX_PER_LEVEL, BIN_ROWS_MAX);
        return SN_ERR_UNSUPPORTED;
    }
    const BinLayout lay = bin_layout(n64, C, bins_used, slab_floats, B);
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
    uint32_t *blkcnt = lay.n_blk ? reinterpret_cast<uint32_t *>(w + lay.blkcnt) : nullptr;
    if (!blkcnt) hipLaunchKernelGGL(k_bin_zero, dim3(div_up(lay.total_bins, 256u)), dim3(256), 0, st, counts, lay.total_bins);   // (a kernel, not hipMemsetAsync: one node type in a captured graph)
    const dim3 gs(div_up(B, 256u * SPT), max_level), blk(256);
    const bool fast = D == 3 && levels_fast(g);          // branch-free row addressing (pair_rows)
    if (D == 3 && fast) hipLaunchKernelGGL((k_bin_count<3, true>), gs, blk, 0, st, inputs, B, g, bg, counts, blkcnt, lay.total_bins);
    else if (D == 3) hipLaunchKernelGGL((k_bin_count<3, false>), gs, blk, 0, st, inputs, B, g, bg, counts, blkcnt, lay.total_bins);
    else hipLaunchKernelGGL((k_bin_count<2, false>), gs, blk, 0, st, inputs, B, g, bg, counts, blkcnt, lay.total_bins);
    if (blkcnt) hipLaunchKernelGGL(k_bin_colscan, dim3(div_up(lay.total_bins, 32u)), dim3(256), 0, st, blkcnt, lay.n_blk, lay.total_bins, counts);
    SN_LAUNCH_CHECK("k_bin_count");
    uint32_t *block_sums = reinterpret_cast<uint32_t *>(w + lay.block_sums);
    const dim3 gp(div_up(lay.total_bins, PLAN_BLOCK_BINS));
    hipLaunchKernelGGL(k_bin_plan_sums, gp, dim3(PLAN_THREADS), 0, st, counts, lay.total_bins, bg, block_sums);
    hipLaunchKernelGGL(k_bin_plan_emit, gp, dim3(PLAN_THREADS), 0, st, counts, cursor, lay.total_bins, block_sums, hdr, items, shared_bins, bg, (uint32_t)slab_floats, grad_embeddings);
    SN_LAUNCH_CHECK("k_bin_plan");
    const size_t lds = (size_t)(BIN_ROWS_MAX + BIN_FLOATS) * sizeof(float);      // 80 KiB: two workgroups per CU
    const dim3 ga(lay.max_items), gm(lay.max_shared_bins < 2048u ? lay.max_shared_bins : 2048u);
    // entries as products (push) or as references (pull, k_bin_refs).  References win at every C measured (C = 8: 0.55 -> 0.33 ms, C = 2: 0.26 -> 0.235, 0.27 -> 0.26,
    // 0.16 -> 0.144 ms; profiles/r05/bin_stats_*.txt): even where a product is as short as a reference the sorted runs replace scattered 8 + 2 byte stores;
    // a reference holds the sample index in 22 bits
    const bool pull = (g_bin_pull < 0 ? true : g_bin_pull != 0) && C >= 2u && B < (1u << REF_KEY_SHIFT);     // (strictly below: sample 2^22 - 1 in row 1023 of a bin would encode the 'no entry' key 0xffffffff)   // (C >= 2: the references live in the products' region)
    uint2 *eref = reinterpret_cast<uint2 *>(econtrib);
    const BinPull bp{grad, eref, B, layout, gstride};
    constexpr size_t refs_lds = (size_t)REF_WINDOW * 2 * sizeof(uint32_t);
    const dim3 blk_refs(REF_THREADS);
#define SN_BIN_ACC(CC)                                                                                                           \
    do {                                                                                                                         \
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accum<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_bin_accum<CC>), ga, blk, lds, st, hdr, items, shared_bins, g, bg, ekey, econtrib, slabs, grad_embeddings); \
    } while (0)
#define SN_BIN_REFS(DD, FAST)                                                                                                    \
    do {                                                                                                                         \
        SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_refs<DD, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)refs_lds)); \
        hipLaunchKernelGGL((k_bin_refs<DD, FAST>), gs, blk_refs, refs_lds, st, inputs, B, g, bg, cursor, blkcnt, lay.total_bins, eref); \
    } while (0)
#define SN_BIN_C(DD, CC)                                                                                                         \
    do {                                                                                                                         \
        if (pull) {                                                                                                              \
            if (DD == 3 && fast) SN_BIN_REFS(DD, DD == 3); else SN_BIN_REFS(DD, false);                                          \
            SN_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_pull<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_bin_pull<CC>), ga, blk, lds, st, hdr, items, shared_bins, g, bg, bp, slabs, grad_embeddings);  \
        } else {                                                                                                                 \
            if (DD == 3 && fast) hipLaunchKernelGGL((k_bin_scatter<DD, CC, DD == 3>), gs, blk, 0, st, inputs, grad, B, g, bg, layout, gstride, cursor, blkcnt, lay.total_bins, ekey, econtrib); \
            else hipLaunchKernelGGL((k_bin_scatter<DD, CC, false>), gs, blk, 0, st, inputs, grad, B, g, bg, layout, gstride, cursor, blkcnt, lay.total_bins, ekey, econtrib); \
            SN_BIN_ACC(CC);                                                                                                      \
        }                                                                                                                        \
        hipLaunchKernelGGL((k_bin_merge<CC>), gm, blk, 0, st, hdr, shared_bins, g, bg, slabs, grad_embeddings);                  \
    } while (0)
#define SN_BIN_D(DD)                                                                                                             \
    switch (C) { case 1: SN_BIN_C(DD, 1); break; case 2: SN_BIN_C(DD, 2); break; case 4: SN_BIN_C(DD, 4); break;                 \
                 case 8: SN_BIN_C(DD, 8); break; case 16: SN_BIN_C(DD, 16); break; default: SN_BIN_C(DD, 32); break; }
    if (D == 3) { SN_BIN_D(3) } else { SN_BIN_D(2) }
#undef SN_BIN_D
#undef SN_BIN_C
#undef SN_BIN_ACC
#undef SN_BIN_REFS
    SN_LAUNCH_CHECK("k_bin_scatter / k_bin_accum / k_bin_merge");
    return SN_OK;
}

#if SN_BIN_TRACE
int sn_bin_debug_trace(unsigned long long *out) {
    SN_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bin_trace), sizeof(unsigned long long) * 2 * 16 * 16));
    return SN_OK;
}
#endif

}  // extern "C"
