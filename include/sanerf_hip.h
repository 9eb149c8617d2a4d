/*
 * sanerf_hip.h — C ABI of libsanerf_hip.so, the MI355X (gfx950) implementation of
 * SANeRF-HQ's volumetric-rendering hot path.
 *
 * Boundary contract
 *   - plain `extern "C"`, raw DEVICE pointers + sizes + a hipStream_t passed as void*;
 *     no torch / ATen types.  Small per-level / per-layer tables (offsets, dims) are
 *     HOST data, as noted per argument.
 *   - the caller allocates every output (the reference does the same:
 *     gridencoder/grid.py:49,55,83,86); functions write in place.
 *   - every function returns 0 on success or a negative sn_status; it never throws.
 *     sn_last_error() returns a thread-local message for the last failure
 *     (the reference raises RuntimeError through TORCH_CHECK, gridencoder.cu:15-18,392).
 *   - all work is enqueued on `stream` (the reference launches on the legacy default
 *     stream, gridencoder.cu:386); nothing synchronises.
 *
 * Each entry point names the reference interface (file:line under /root/reference)
 * it replaces.  INTEGRATION.md shows the ctypes binding the reference side would add.
 */
#ifndef SANERF_HIP_H
#define SANERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_MAX_LEVELS 32
#define SN_MAX_LAYERS 8
#define SN_MAX_STAGES 4
#define SN_ABI_VERSION 12

typedef void *sn_stream_t; /* hipStream_t */

typedef enum {
    SN_OK = 0,
    SN_ERR_INVALID = -1,     /* bad argument (null pointer, unsupported D/C, ...) */
    SN_ERR_UNSUPPORTED = -2, /* valid in the reference but not built here */
    SN_ERR_HIP = -3,         /* HIP runtime error (launch failed, no device) */
    SN_ERR_WORKSPACE = -4    /* workspace too small */
} sn_status;

enum { SN_F32 = 0, SN_F16 = 1 };                 /* table storage type */
enum { SN_LAYOUT_LBC = 0, SN_LAYOUT_BLC = 1 };   /* [L,B,C] (reference kernel) or [B,L*C] (what grid.py:63 returns) */

int sn_abi_version(void);
/* How this library was built: bit 0 (SN_BUILD_EXPERIMENTS) = -DSN_EXPERIMENTS, the measured-and-rejected kernel variants are compiled in
 * (sn_render_tuning.experiment, sn_debug_set); bit 1 (SN_BUILD_POISON_LDS) = -DSN_POISON_LDS, every kernel that uses LDS fills it with
 * signalling NaNs at entry (a read-before-write shows up as NaN in the result: the r03 bug class of k_final_stage_any).  The product
 * library (`make`) carries neither. */
#define SN_BUILD_EXPERIMENTS 1
#define SN_BUILD_POISON_LDS 2
int sn_build_flags(void);
/* Experiments builds only (SN_ERR_UNSUPPORTED otherwise): process-wide A/B switches of variants that are not reachable through a
 * descriptor.  Keys: "wide_jit" (0: the superseded k_mlp_wide forward instead of k_mlp_wide_j). */
int sn_debug_set(const char *key, int value);
const char *sn_last_error(void);
/* number of HIP devices visible; <0 on error.  Lets hosts fail loudly before any launch. */
int sn_device_count(void);

/* ------------------------------------------------------------------------------------------
 * gridencoder  — replaces gridencoder/src/gridencoder.h:12-16 (pybind: bindings.cpp:5-9)
 * inputs  [B,D] f32 in [0,1] (device);  embeddings [rows,C] (device, table_dtype);
 * offsets [L+1] int32 on the HOST (the reference keeps them on the device and re-reads them in
 * every thread; here the per-level table — resolution per gridencoder.cu:133, size, dense/hash —
 * is computed once on the host and passed by value);  S = (float)log2(per_level_scale);
 * outputs f32, layout per `layout`;  dy_dx [B,L,D,C] f32 or NULL.
 * ------------------------------------------------------------------------------------------ */
int sn_grid_encode_forward(const float *inputs, const void *embeddings, int table_dtype,
                           const int32_t *offsets_host, float *outputs,
                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                           float S, uint32_t H, float *dy_dx,
                           uint32_t gridtype, int align_corners, uint32_t interp,
                           int layout, sn_stream_t stream);
/* outputs [B, L*C + E] = cat([grid_encode(inputs) in the [B, L*C] layout, extra [B, E]], -1): the mask head's MLP input
 * (nerf/renderer.py:380) in one pass.  D = 3, C in {2, 4, 8}, forward only (inference). */
int sn_grid_encode_forward_cat(const float *inputs, const void *embeddings, int table_dtype, const int32_t *offsets_host,
                               const float *extra, uint32_t E, float *outputs, uint32_t B, uint32_t C, uint32_t L,
                               float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, sn_stream_t stream);
/* grad f32 in `layout`; grad_embeddings [rows,C] f32, zero-initialised by the caller
 * (grid.py:83); grad_inputs [B,D] f32 written when dy_dx != NULL. */
int sn_grid_encode_backward(const float *grad, const float *inputs, const void *embeddings, int table_dtype,
                            const int32_t *offsets_host, float *grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                            float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                            uint32_t gridtype, int align_corners, uint32_t interp,
                            int layout, sn_stream_t stream);
/* Same result as sn_grid_encode_backward (embedding gradient only; replaces the scatter of gridencoder.cu:252-349 / grid.py:71-95)
 * without the reference's atomics-per-corner: a level's rows are cut into bins sized to what a workgroup holds in LDS, the
 * (row, contribution) pairs are partitioned by bin in one pass and every bin is counting-sorted and summed in LDS by the workgroup(s)
 * that own it (grid_binned.hip; rounds 1-3 sorted the pairs with a library radix sort instead).  D in {2,3}; C a power of two <= 32;
 * B*max_level*2^D < 2^31; every level at most 4096 bins of 4096 rows -- sn_grid_backward_binned_workspace_bytes() returns 0 for shapes
 * outside that, which then take sn_grid_encode_backward.  workspace >= that many device bytes (bin counters, work lists, 2-byte row keys
 * and the B*max_level*2^D*C pre-multiplied contributions, partial-sum slabs of split bins <= twice the table), 16-byte aligned like grad;
 * grad_embeddings zero-initialised by the caller, rows without contribution are not written.  offsets_host = the L+1 level offsets
 * (host memory).  No host synchronisation, static grids (graph-capturable).  Sums are order-nondeterministic in the last bits, like the
 * reference's atomicAdd.  Several times faster than the atomic path at the sizes of the training steps. */
/* sn_grid_encode_backward_binned on a gradient whose rows are grad_row_stride floats apart ([B, L*C] layout only; 0 = L*C): the gradient of a
 * wider tensor that carries the grid features in its first L*C columns -- the mask head's MLP input cat([m_grid(x), geo_feat]) of
 * renderer.py:380 -- is read in place, without the slice copy.  Rows need only 4-byte alignment. */
int sn_grid_encode_backward_binned_rows(const float *grad, uint32_t grad_row_stride, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                        float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                        int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream);
size_t sn_grid_backward_binned_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, const int32_t *offsets_host);
int sn_grid_encode_backward_binned(const float *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                   float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   int layout, void *workspace, size_t workspace_bytes, sn_stream_t stream);
/* gridencoder.h:15 / grid.py:170-191 */
int sn_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                            const int32_t *offsets_host, float weight,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, uint32_t gridtype, int align_corners,
                            sn_stream_t stream);
/* gridencoder.h:16 / grid.py:193-204.  B = number of table rows. */
int sn_grad_weight_decay(const float *embeddings, float *grad, const int32_t *offsets_host,
                         float weight, uint32_t B, uint32_t C, uint32_t L, sn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * shencoder — replaces shencoder/src/shencoder.h:9-10.  inputs [B,3] f32 (unit vectors),
 * outputs [B,degree^2], dy_dx [B,3,degree^2] or NULL; degree in 1..8.
 * ------------------------------------------------------------------------------------------ */
int sn_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree,
                         float *dy_dx, sn_stream_t stream);
/* accumulates into grad_inputs [B,3] (zero-initialised by the caller, sphere_harmonics.py:50) */
int sn_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t degree,
                          const float *dy_dx, float *grad_inputs, sn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * freqencoder — replaces freqencoder/src/freqencoder.h:6-9.  C = D + 2*D*deg.
 * ------------------------------------------------------------------------------------------ */
int sn_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                           float *outputs, sn_stream_t stream);
int sn_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg,
                            uint32_t C, float *grad_inputs, sn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * raymarching — the reference has no native twin for these (README.md:32-34 mentions a
 * `raymarching` extension that is not in the tree); each replaces a block of torch ops.
 * ------------------------------------------------------------------------------------------ */
/* nerf/utils.py:201-205,269-287 (full image).  pose: 16 floats row-major cam2world, HOST. */
int sn_rm_generate_rays(const float *pose_host, float fx, float fy, float cx, float cy,
                        uint32_t H, uint32_t W, uint32_t row_begin, uint32_t row_end,
                        float *rays_o, float *rays_d, sn_stream_t stream);
/* nerf/utils.py:209-287 for a drawn pixel subset (a training step's rays): ray n looks through flat pixel index inds[n]
 * (int64, row-major in an image of width W) of camera poses[n].  poses: DEVICE, n_poses x 16 floats row-major cam2world,
 * n_poses = 1 (one camera for all rays) or N (one camera per ray: provider.py:908-913 `random_image_batch`);
 * intrinsics: DEVICE, n_intrinsics x (fx, fy, cx, cy), 1 or N.  No host round trip. */
int sn_rm_rays_from_pixels(const float *poses, uint32_t n_poses, const float *intrinsics, uint32_t n_intrinsics, const int64_t *inds,
                           uint32_t W, uint32_t N, float *rays_o, float *rays_d, sn_stream_t stream);
/* nerf/renderer.py:122-139.  aabb: 6 floats, HOST.  nears/fars [N]. */
int sn_rm_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb_host,
                             float min_near, uint32_t N, float *nears, float *fars, sn_stream_t stream);
/* nerf/renderer.py:60-69.  x,z [N,3]. */
int sn_rm_contract(const float *x, uint32_t N, float *z, sn_stream_t stream);
/* nerf/renderer.py:84-119.  bins [N,T0+1], weights [N,T0] -> out_bins [N,T];
 * inds int32 [N,T] or NULL (the torch.searchsorted result);
 * u: NULL (the linspace of renderer.py:98), a shared device table [T] (u_stride = 0), or per-ray
 * rows [N,T] (u_stride = T; what perturb=True needs, renderer.py:101-102). */
int sn_rm_sample_pdf(const float *bins, const float *weights, uint32_t N, uint32_t T0, uint32_t T,
                     const float *u, uint32_t u_stride, float *out_bins, int32_t *inds, sn_stream_t stream);
/* nerf/renderer.py:308-325.  real_bins [N,T+1], sigmas [N,T] -> weights [N,T]. */
int sn_rm_weights_from_sigma(const float *real_bins, const float *sigmas, uint32_t N, uint32_t T,
                             int last_sample_opaque, float *weights, sn_stream_t stream);
/* Backward of sn_rm_weights_from_sigma w.r.t. sigmas (what autograd derives from renderer.py:308-325; the bin edges
 * carry no gradient on this path).  grad_weights [N,T] -> grad_sigmas [N,T]; any T <= 131072 (T <= 256: one pass in registers; longer rays keep
 * the fp64 prefix of every 64-sample segment and evaluate the terms again on the way back -- the same bits). */
int sn_rm_weights_from_sigma_backward(const float *real_bins, const float *sigmas, const float *grad_weights, uint32_t N, uint32_t T,
                                      int last_sample_opaque, float *grad_sigmas, sn_stream_t stream);

/* Inter-level proposal loss of one proposal stage (nerf/renderer.py:30-57): bins [N,T+1], weights [N,T] of the proposal
 * stage against ref_bins [N,Tr+1], ref_weights [N,Tr] of the final stage (detached in the reference).
 * Forward (loss_per_ray != NULL, grad_weights NULL): loss_per_ray[n] = sum_j max(rw_j - bound_j, 0)^2 / (rw_j + 1e-8);
 * the reference's value is sum(loss_per_ray) / (N * Tr).  Backward (grad_weights != NULL, loss_per_ray NULL):
 * grad_weights[n,i] = d(sum_j term_j)/d w_i (scale by upstream / (N * Tr)).  Deterministic (no atomics). */
int sn_rm_proposal_loss(const float *bins, const float *weights, const float *ref_bins, const float *ref_weights, uint32_t N, uint32_t T,
                        uint32_t Tr, float *loss_per_ray, float *grad_weights, sn_stream_t stream);

/* Distortion loss (nerf/renderer.py:17-27 -> torch_efficient_distloss.eff_distloss, third party, not vendored: the
 * published value is mean over rays of (1/3) sum_i w_i^2 d_i + sum_ij w_i w_j |m_i - m_j|) on bins [N,T+1], weights [N,T]:
 * loss_per_ray [N] (the reference's value is its mean) and grad_weights [N,T] = d loss_per_ray / d w, in one pass. */
int sn_rm_distort_loss(const float *bins, const float *weights, uint32_t N, uint32_t T, float *loss_per_ray, float *grad_weights,
                       sn_stream_t stream);

/* Mask-field NLL per ray (nerf/trainer.py:419-428): loss_per_ray[n] = -log(clamp(softmax(logits[n, :])[labels[n]], eps, 1 - eps)) -- the
 * reference's `loss` before its .mean() -- and, when grad_logits != NULL, d loss_per_ray[n] / d logits[n, :] in the same pass (zero where
 * the clamp binds, like torch.clamp's backward).  logits [N,K] f32, labels [N] int64 (a label outside 0..K-1: loss 0, no gradient). */
int sn_rm_mask_nll(const float *logits, const int64_t *labels, uint32_t N, uint32_t K, float eps, float *loss_per_ray, float *grad_logits,
                   sn_stream_t stream);

/* One stage's sample geometry (renderer.py:277-285): bins [N,T+1] in [0,1] -> real_bins [N,T+1] (distances along the
 * ray through the Mip-360 spacing of nears/fars [N]), rays_t [N,T] (mid-points), xyzs [N,T,3] (positions, contracted
 * into [-2,2]^3 like sn_rm_contract if `contract`).  Nothing here is differentiated by the reference. */
int sn_rm_sample_positions(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *bins,
                           uint32_t N, uint32_t T, int contract, float *real_bins, float *rays_t, float *xyzs, sn_stream_t stream);

/* The same with the grid encoder's first step folded in (gridencoder/grid.py:156): grid_bound > 0 makes `xyzs` receive
 * (x + grid_bound) / (2 grid_bound), the unit-cube coordinates sn_grid_encode_forward takes; grid_bound = 0 = sn_rm_sample_positions. */
int sn_rm_sample_positions_ex(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *bins,
                              uint32_t N, uint32_t T, int contract, float grid_bound, float *real_bins, float *rays_t, float *xyzs,
                              sn_stream_t stream);

/* Training-time jitter of the sampling positions (perturb=True) from a caller-supplied uniform [0,1) tensor `uniform` [N,T] (NULL: none):
 *   kind 0 (renderer.py:262-270, stage-0 bins, T = num_steps[0] + 1): out = clamp(linspace(0, 1, T) + (uniform - 0.5) / (T - 1), 0, 1)
 *   kind 1 (renderer.py:97-102, sample_pdf's u):                       out = linspace(0.5/T, 1 - 0.5/T, T) + (uniform - 0.5) / T
 * out [N,T].  The reference draws torch.rand_like per stage; one draw for all stages, sliced, is statistically the same. */
int sn_rm_jitter(const float *uniform, uint32_t N, uint32_t T, int kind, float *out, sn_stream_t stream);

/* Per-ray sums of the training path in one pass (renderer.py:327-347 with network.py:164-170's colour = cat([geo_feat, SH(d)])):
 *   weights_sum[n] = sum_t w;  depth[n] = sum_t w rays_t;  f_image[n, 0:15] = sum_t w raw[n,t,1:16];  f_image[n, 15:31] = SH4(d/|d|) weights_sum
 * raw [N,T,16] = grid_mlp's output ([sigma_raw | 15 geometry channels], network.py:94,151-153), rays_d [N,3] un-normalised.
 * The backward entry returns d/d weights [N,T] and d/d raw [N,T,16] (channel 0 = 0) from the gradients of the three outputs
 * (each may be NULL = zero).  raw / grad_raw 16-byte aligned. */
int sn_rm_ray_composite(const float *weights, const float *rays_t, const float *raw, const float *rays_d, uint32_t N, uint32_t T,
                        float *weights_sum, float *depth, float *f_image, sn_stream_t stream);
int sn_rm_ray_composite_backward(const float *weights, const float *rays_t, const float *raw, const float *rays_d, const float *grad_weights_sum,
                                 const float *grad_depth, const float *grad_f_image, uint32_t N, uint32_t T, float *grad_weights, float *grad_raw,
                                 sn_stream_t stream);

/* sn_rm_proposal_loss with every written value multiplied by scale * (scale_dev ? *scale_dev : 1): the mean's 1 / (N Tr) and autograd's
 * incoming (device-resident) gradient scalar without extra elementwise passes. */
int sn_rm_proposal_loss_scaled(const float *bins, const float *weights, const float *ref_bins, const float *ref_weights, uint32_t N, uint32_t T,
                               uint32_t Tr, float scale, const float *scale_dev, float *loss_per_ray, float *grad_weights, sn_stream_t stream);

/* The same for ANY number of samples per ray (T, Tr <= 512 run exactly sn_rm_proposal_loss_scaled's launch and need no workspace; longer rays keep
 * their prefix sums and search tables in `workspace` -- 8-byte aligned, sn_rm_proposal_loss_workspace_bytes(N, T, Tr, backward) bytes, at most
 * 64 MiB -- and a wave walks several rays: the same operations in the same order, the same bits). */
size_t sn_rm_proposal_loss_workspace_bytes(uint32_t N, uint32_t T, uint32_t Tr, int backward);
int sn_rm_proposal_loss_long(const float *bins, const float *weights, const float *ref_bins, const float *ref_weights, uint32_t N, uint32_t T,
                             uint32_t Tr, float scale, const float *scale_dev, float *loss_per_ray, float *grad_weights, void *workspace,
                             size_t workspace_bytes, sn_stream_t stream);

/* Clears `bytes` bytes at `ptr` (both multiples of 16) with a kernel on `stream` -- capturable in a HIP graph, unlike a memset node whose
 * replays faulted once the allocator had reused the memory (round 4); the zeros_like of gridencoder/grid.py:83. */
int sn_zero(void *ptr, size_t bytes, sn_stream_t stream);

/* nerf/renderer.py:333-338,361,384: out[n,k] = sum_t weights[n,t] * values[n,t,k]  (K may be 1). */
int sn_rm_composite(const float *weights, const float *values, uint32_t N, uint32_t T, uint32_t K,
                    float *out, sn_stream_t stream);
/* backward of sn_rm_composite w.r.t. values: grad_values[n,t,k] = weights[n,t] * grad_out[n,k] */
int sn_rm_composite_backward(const float *weights, const float *grad_out, uint32_t N, uint32_t T, uint32_t K,
                             float *grad_values, sn_stream_t stream);

/* ---- fused whole-path render: nerf/renderer.py:221-357 + network.py:146-186 ---- */
typedef struct sn_grid_desc {
    const void *embeddings;                 /* device */
    int32_t     table_dtype;                /* SN_F32 / SN_F16 */
    int32_t     offsets[SN_MAX_LEVELS + 1]; /* host values */
    uint32_t    D, C, L;
    float       S;                          /* (float)log2(per_level_scale) */
    uint32_t    H;                          /* base resolution */
    uint32_t    gridtype, align_corners, interp;
} sn_grid_desc;

typedef struct sn_mlp_desc {
    const float *weight[SN_MAX_LAYERS];     /* device, [out,in] row-major = nn.Linear.weight */
    const float *bias[SN_MAX_LAYERS];       /* device or NULL */
    uint32_t     dims[SN_MAX_LAYERS + 1];
    uint32_t     num_layers;
    uint32_t     activation;                /* 0 relu, 1 leaky_relu(0.01) */
    uint32_t     skip_mask;
} sn_mlp_desc;

/* Kernel selection of sn_rm_render_rays, read from the caller's cfg on every call (ABI <= 6 read process-wide environment variables
 * instead).  All zero = the defaults.  None of these changes WHAT is computed beyond fp32 round-off; the notes say which are bit-neutral. */
enum { SN_MLP_AUTO = 0, SN_MLP_F16X3 = 1, SN_MLP_MFMA32 = 2, SN_MLP_VALU = 3, SN_MLP_F16X1 = 5 };
enum { SN_EXP_NONE = 0, SN_EXP_ROLE_SPLIT = 1, SN_EXP_LDS_LEVEL0 = 2, SN_EXP_FINAL_ONE_WG = 3 };
typedef struct sn_render_tuning {
    int32_t mlp_mode;            /* SN_MLP_AUTO: split-fp16 on the matrix cores unless cfg.mlp_exact_fp32; SN_MLP_F16X3 forces it (overrides the
                                  * range guard); SN_MLP_MFMA32: exact fp32 v_mfma_f32_32x32x2_f32; SN_MLP_VALU: vector-ALU fallback (A/B of the layouts);
                                  * SN_MLP_F16X1 (opt-in measurement mode, NOT fp32-class: RGB 3e-4 from the reference on the stress-init fixtures, over
                                  * the 1e-4 bar; honoured by the plain last-stage launch with fp16 tables, F16X3 elsewhere): one product per multiply --
                                  * plain fp16 operands, fp32 accumulation: what an autocast fp16 run of the reference multiplies */
    int32_t per_sample_form;     /* 1: the last stage evaluates the third MLP layer per sample everywhere (no "linear tail"): bit-identical to
                                  * the compacting / several-lanes-per-ray kernels, fp32 round-off away from the default */
    int32_t densify;             /* hashed levels 5-6 of the main grid re-laid out per call like dense levels (bit-neutral): 0 automatic (fp16
                                  * tables and >= 64 M last-stage samples), 1 never, 2 whenever the kernel exists for the call */
    int32_t linear_tile_order;   /* 1: workgroup b renders tile b (no XCD-aware remap; bit-neutral) */
    int32_t prop_sp_max_rays;    /* linear-order batches up to this many rays run the proposal stages with 8 lanes per ray (bit-neutral):
                                  * 0 default (32768), < 0 never */
    int32_t final_sp_max_rays;   /* the same for the last stage (several lanes per ray, per-sample form): 0 default (16384), < 0 never */
    int32_t feat_levels;         /* levels per workgroup pass of the feature stage: 0 default (2), 1, 2 or 4 (bit-neutral) */
    int32_t band_streams;        /* image mode, schedules with proposal stages: the image is rendered as two row bands whose kernels go to TWO HIP streams
                                  * (the caller's and a library-owned one, forked and joined with events inside the call: capturable), so that
                                  * the vector-ALU-bound proposal stages of one band overlap the texture-path-bound last stage of the other
                                  * (bit-neutral): 0 automatic (>= 768 workgroups: 800x800 [128,64,32] 4.36 -> 4.07 ms fp32, 3.85 -> 3.71 fp16; 448x448 2.10 -> 2.00, 1.88 -> 1.74;
                                  * >= 512 with the feature stage: 400x400 + SAM head 3.00 -> 2.82 ms), 1 never, 2 whenever the image has two bands
                                  * of whole tile rows, K > 2: K bands dealt alternately to the two streams (measured, profiles/r05/band_count_ab.txt: two bands
                                  * are the optimum at 400 / 800 / 1600 pixels -- 800x800 fp16: 3.84 none, 3.59 two, 4.18 four; 1600x1600: 13.36, 12.71, 12.70) */
    int32_t exact_early_out;     /* the last stage leaves the march once the transmittance of all 64 rays of a wave has underflowed to EXACTLY 0 (every later
                                  * weight is alpha * 0: bit-neutral; opaque scenes only): 2 = on, 0 / 1 = off (the default: behind proposal stages it buys nothing, in a
                                  * single-stage schedule 6.07 -> 4.43 ms on an opaque field at 0.5-1 % cost on a semi-transparent one).
                                  * The proposal stages always do (their remaining weights are written as 0 without evaluating the density). */
    int32_t wave_tile;           /* image mode: the 64 lanes of a wave cover 2^w x 2^(6-w) pixels: 0 default (8x8), 1..5 = w (2x32 ... 32x2); bit-neutral
                                  * (A/B of the lines a gather instruction touches: profiles/r05/tile_shape_ab.txt) */
    int32_t prop_sp_lanes;       /* small linear-order batches, proposal stages: lanes that share a ray: 0 automatic (32: the fastest from 1024 to
                                  * 32768 rays, profiles/r06/prop_sp_lanes_ab.json), or 8 / 16 / 32 (bit-neutral) */
    int32_t feat_patch;          /* feature stage, dense levels: 1 = a wave fetches the bounding box of its rays' vertices once into LDS and the lanes read their
                                  * corners there (north_star "LDS staging of per-tile grid voxels"; bit-neutral).  0 default = off: measured 3-9 % SLOWER than
                                  * the direct gathers (profiles/r06/feat_patch_ab.json) */
    int32_t prop_pair;           /* proposal stages (one lane per ray): a lane evaluates TWO consecutive samples at once -- their gathers and MLP chains interleave
                                  * (k_prop_stage<..., UN = 2>, 3 waves per SIMD instead of 5; bit-neutral): 0 automatic, 1 never, 2 always */
    int32_t experiment;          /* SN_EXP_*: variants that were built, verified bit-identical and measured SLOWER; honoured only by
                                  * a library built with -DSN_EXPERIMENTS (sn_build_flags), SN_ERR_UNSUPPORTED otherwise */
} sn_render_tuning;

typedef struct sn_render_cfg {
    uint32_t     num_stages;                /* len(opt.num_steps), 1..SN_MAX_STAGES */
    uint32_t     num_steps[SN_MAX_STAGES];
    sn_grid_desc prop_grid[SN_MAX_STAGES];  /* stages 0..num_stages-2 (network.py:131-143) */
    sn_mlp_desc  prop_mlp[SN_MAX_STAGES];
    /* The field.  Two last-stage kernels exist: the reference network's own sizes (network.py:93-98: L=16, level_dim 2,
     * 32-64-64-16 and 31-32-32-3 bias-free ReLU MLPs) run on the matrix cores; any OTHER field of the same structure
     * (renderer.py:221-357 is size-agnostic) -- 3-D grid with level_dim 2 and <= 64 features, bias-free ReLU MLPs of <= 4
     * layers and <= 64 neurons, grid_mlp -> [sigma_raw | <= 31 geometry channels], view_mlp input = geometry channels + 16 SH
     * values, 3 outputs -- runs in a size-agnostic kernel (fp32 fmaf chains).  Other fields: SN_ERR_UNSUPPORTED. */
    sn_grid_desc grid;                      /* network.py:93 */
    sn_mlp_desc  grid_mlp;                  /* network.py:94: -> [sigma_raw | geo_feat] */
    sn_mlp_desc  view_mlp;                  /* network.py:98: per ray, after compositing */
    uint32_t     sh_degree;                 /* network.py:97 */
    float        aabb[6];
    float        min_near;
    float        bound;                     /* grid bound: 2 when opt.contract */
    int32_t      contract;
    int32_t      last_sample_opaque;        /* opt.background == 'last_sample' */
    float        bg_color;
    /* optional feature stage (renderer.py:301-302 + 361): f_feat[n,:] = sum_j w[n,j] * feat_grid(xyz[n,j]) over the
     * last stage's samples -- the SAM head's f_sam with feat_grid = s_grid (network.py:103) */
    sn_grid_desc feat_grid;
    int32_t      with_feat;
    /* opt-in, NOT reference behaviour (the reference always evaluates every sample): in the last stage a wave (8x8
     * pixels) stops marching once the transmittance of all its rays is below this value; 0 = off (default).  Every
     * output changes by at most ~eps * |feature|.  Ignored when per-sample outputs or the feature stage are on. */
    float        early_stop_eps;
    /* Arithmetic of the 32-64-64-16 MLP on the matrix cores.  0 (default): fp16 hi/lo-split products with fp32 accumulation
     * (2^-22 per product) -- requires every weight and activation of that MLP to stay below 65504 in magnitude, which the
     * CALLER guarantees (the Python mirror derives a bound from max|table| and the weights' row norms and sets 1 when it
     * cannot).  1: exact fp32 v_mfma_f32_32x32x2_f32 (no range limit, ~2.2x slower final stage).  tuning.mlp_mode != SN_MLP_AUTO
     * overrides this field. */
    int32_t      mlp_exact_fp32;
    /* opt-in, NOT reference behaviour: the last stage runs with per-ray termination and wave-level compaction of live
     * samples (k_final_stage_cmp).  A ray stops taking samples once its transmittance is below early_stop_eps (if > 0),
     * rays that miss the aabb (renderer.py:133-135: near = far = 1e9) and lanes beyond the image edge take none, and each
     * wave deals its 64 evaluation slots (gather lanes, rows of the MFMA tile) out to the rays still live, found with a
     * ballot + prefix count per iteration.  With nothing to skip the outputs are bit-identical to the default kernel; a
     * terminated ray changes every output by at most ~eps * |feature|; a missed ray gets weights_sum = depth = 0 and the
     * background colour (the reference marches such rays through inf / NaN distances; with its default options every
     * weight comes out 0 as well).  Ignored (default kernel runs) when per-sample outputs or the feature stage are on,
     * for tables the FinalLv fast path does not cover, and in the exact-fp32 / VALU MLP modes. */
    int32_t      compact_live;
    sn_render_tuning tuning;
} sn_render_cfg;

typedef struct sn_render_io {
    /* inputs */
    const float *rays_o, *rays_d;           /* [N,3] device */
    const float *cam_near_far;              /* [N,2] device or NULL */
    uint32_t     N;
    uint32_t     tile_w;                    /* >0: rays are a row-major image of this width; lanes map to 8x8 pixel tiles */
    const float *bins0_table;               /* device [num_steps[0]+1], or per ray [N, bins0_ray_stride], or NULL (-> linspace recipe) */
    const float *u_table[SN_MAX_STAGES];    /* device [num_steps[k]+1] for k>=1, or per ray [N, u_ray_stride[k]], or NULL */
    uint32_t     bins0_ray_stride;          /* 0: bins0_table is one table shared by all rays; else floats per ray (>= num_steps[0]+1):
                                             * the per-ray perturbed bins of a training step (renderer.py:267-270) */
    uint32_t     u_ray_stride[SN_MAX_STAGES]; /* the same for u_table[k]: sample_pdf's perturbed u (renderer.py:101-102) */
    int32_t      skip_final;                /* != 0: run the proposal stages only and write the LAST stage's resampled bins to
                                             * bins[num_stages-1] ([N,T_last+1]); image / depth / weights_sum may be NULL.  A training
                                             * step whose proposal networks are not updated (trainer.py:372-373: 4 steps of 5 after step
                                             * 3000) takes its final-stage sample positions from here and differentiates only that stage */
    /* outputs */
    float       *image;                     /* [N,3] */
    float       *depth;                     /* [N] */
    float       *weights_sum;               /* [N] */
    uint32_t     out_stride;                /* 0: the three outputs above are dense arrays.  s >= 5: each is a COLUMN RANGE of a row-major buffer whose
                                             * rays are s floats apart (image[n*s + 0..2], depth[n*s], weights_sum[n*s]): a band that feeds the
                                             * image all-gather is written straight into its [N,5] payload (image = p, depth = p+3, weights_sum = p+4) */
    /* optional per-stage outputs, [N, ...] row-major like the reference tensors; NULL = not wanted */
    float       *bins[SN_MAX_STAGES];       /* [N,T_k+1] */
    float       *weights[SN_MAX_STAGES];    /* [N,T_k]   */
    float       *sigmas[SN_MAX_STAGES];     /* [N,T_k]   */
    int32_t     *inds[SN_MAX_STAGES];       /* [N,T_k+1], k>=1 */
    float       *xyzs_last;                 /* [N,T_last,3] contracted positions of the last stage */
    float       *geo_feat_last;             /* [N,T_last,geo] per-sample geometry features (feeds the mask head) */
    float       *f_image;                   /* [N, geo+sh] composited colour features (feeds the SAM head) */
    float       *f_feat;                    /* [N, L*C of cfg->feat_grid]; required when cfg->with_feat */
    uint32_t     head_stride;               /* 0: f_image / f_feat are dense arrays.  s > 0: both are COLUMN RANGES of one row-major [N, s] buffer -- the SAM head's
                                             * MLP input cat([f_sam, f_image, image, depth]) of renderer.py:366 written in place of a concatenation: rows are s floats
                                             * apart, and rgb | depth of ray n are written once more at f_image + n*s + 31 (4 floats).  The caller points f_feat at
                                             * column 0 and f_image at column L*C, s = L*C + 31 + 4 (163 for the reference network).  Needs f_image. */
    /* workspace */
    void        *workspace;                 /* device, >= sn_rm_render_workspace_bytes() */
    size_t       workspace_bytes;
} sn_render_io;

/* bytes of device workspace sn_rm_render_rays needs for N rays (tile_w as in sn_render_io) */
size_t sn_rm_render_workspace_bytes(const sn_render_cfg *cfg, uint32_t N, uint32_t tile_w);
int sn_rm_render_rays(const sn_render_cfg *cfg, const sn_render_io *io, sn_stream_t stream);
/* What the last sn_rm_render_rays call of THIS THREAD launched as its last stage (measurement: bench.py prices the gather stream of the
 * kernel that really ran).  final_kernel: "k_final_stage<lt,K=7>", "k_final_stage_sp", ...; dense_levels: levels of the main grid read
 * as packed pair / quad rows; gathers_per_wave_sample: 64-lane gather instructions one sample of one wave issues in that kernel. */
typedef struct sn_launch_info {
    char     final_kernel[64];
    uint32_t workgroups, lds_bytes, dense_levels, gathers_per_wave_sample, launches;
} sn_launch_info;
int sn_rm_last_launch_info(sn_launch_info *info);

/* Feature-head accumulation, nerf/renderer.py:301-302 + 361: out[n, :] = sum_j weights[n,j] * grid(xyzs[n,j])
 * = composite(weights, s_grid(xyzs, bound)) without materialising the [N*T, L*C] per-sample features.
 * xyzs [N,T,3] (contracted sample positions), weights [N,T], out [N, L*C]; tile_w > 0: rays are a row-major image
 * of that width (lanes map to 8x8 pixel tiles).  Forward only (inference); training goes through
 * sn_grid_encode_* + sn_rm_composite*. */
int sn_rm_grid_composite(const float *xyzs, const float *weights, uint32_t N, uint32_t T, float bound,
                         const sn_grid_desc *grid, uint32_t tile_w, float *out, sn_stream_t stream);

/* Wide perceptron of the feature heads on the matrix cores: nerf/network.py:31-66 (SkipConnMLP) followed by an
 * optional nn.LayerNorm (network.py:115) -- samvit_mlp per ray (renderer.py:359-374), mask_mlp per sample
 * (renderer.py:376-385).  x [N, dims[0]] -> out [N, dims[num_layers]].  Hidden widths must be 256, the output
 * width <= 256; bias optional per layer; activation 0 relu / 1 leaky_relu(0.01); skip_mask bit l: layer l sees
 * cat([h, x]).  fp32 in / out; products run as fp16 hi+lo splits with fp32 accumulation (activations must stay
 * inside the fp16 range, |v| < 65504).  workspace: >= sn_mlp_wide_workspace_bytes(), 16-byte aligned (holds the
 * re-ordered weights, rebuilt on every call: the call is stateless).  ln_weight/ln_bias NULL = no LayerNorm. */
size_t sn_mlp_wide_workspace_bytes(const sn_mlp_desc *mlp);
/* Range check of the split-fp16 arithmetic: sn_mlp_wide_forward raises a sticky device-side flag when an output row is not
 * finite (what an activation beyond the fp16 range produces).  Reads and clears it: *flag = 1 if any call since the last
 * read overflowed.  Synchronises the device. */
int sn_mlp_wide_overflow(int32_t *flag);
int sn_mlp_wide_forward(const sn_mlp_desc *mlp, const float *ln_weight, const float *ln_bias, float ln_eps,
                        const float *x, uint32_t N, float *out, void *workspace, size_t workspace_bytes,
                        sn_stream_t stream);

/* Mask head (nerf/renderer.py:304-305, 376-385; network.py:104, 118-123) in one kernel:
 *   out[n, :] = sum_t weights[n,t] * mask_mlp(cat([m_grid(xyzs[n,t]), extra[n,t]]))        out [N, dims[num_layers]]
 * xyzs [N,T,3] (contracted sample positions; x01 = (xyz + bound) / (2 bound) as gridencoder/grid.py:156), extra [N,T,E]
 * (the detached geometry features, E <= 16), weights [N,T].  The lanes of the matrix-core MLP kernel interpolate the grid
 * levels of their samples straight into the first layer's B operand and the epilogue composites the logits, so neither
 * the [N*T, L*8+E] input nor the [N*T, n_inst] logits are written.  (T >= 4: a workgroup walks the samples of 32
 * consecutive rays -- neighbouring rays at one depth share table lines; rays are expected in image order.)  Needs a 3-D fp32 grid with level_dim 8, hidden width
 * 256, <= 32 outputs, no skip layers, T a power of two <= 128; anything else returns SN_ERR_UNSUPPORTED and the caller
 * composes sn_grid_encode_forward_cat + sn_mlp_wide_forward + sn_rm_composite.  Inference only.  Range of the split-fp16
 * arithmetic as sn_mlp_wide_forward (the overflow flag is not raised by this entry).  workspace >=
 * sn_rm_mask_head_workspace_bytes(), 16-byte aligned. */
size_t sn_rm_mask_head_workspace_bytes(const sn_mlp_desc *mlp);
int sn_rm_mask_head(const float *xyzs, const float *extra, const float *weights, uint32_t N, uint32_t T, uint32_t E,
                    float bound, const sn_grid_desc *grid, const sn_mlp_desc *mlp, float *out,
                    void *workspace, size_t workspace_bytes, sn_stream_t stream);

/* Training-time forward of a bias-free perceptron with 256-wide hidden layers and no skip connections (nerf/network.py:31-66: F.linear +
 * activation per layer; the per-sample mask head in training, network.py:118-123, trainer.py:401-428) in ONE kernel, true fp32 products
 * and accumulation on the matrix cores (v_mfma_f32_32x32x2_f32; results differ from a BLAS GEMM only by summation order).
 * x [N, dims[0]] (dims[0] <= 256) -> out [N, dims[nl]] (<= 256, linear); hidden[l] [N, 256] for l < nl-1 receives the POST-activation output
 * of layer l -- what torch's in-place activation leaves for autograd, and what sn_mlp_wide_backward / sn_linear_wgrad read.
 * hidden is a host array of nl-1 device pointers.  All tensors row-major fp32, no alignment requirement beyond 4 bytes. */
int sn_mlp_wide_forward_train(const sn_mlp_desc *mlp, const float *x, uint32_t N, float *const *hidden, float *out, sn_stream_t stream);

/* The same forward on the inference kernel of sn_mlp_wide_forward (split-fp16 x3 products with fp32 accumulation on the matrix cores, ~2^-22
 * per product; any dims[0] <= 1024, biases allowed) with every hidden layer's post-activation output saved: hidden[l] [N, 256], 16-byte
 * aligned.  workspace: sn_mlp_wide_workspace_bytes(mlp), 16-byte aligned.  Activations must stay inside the fp16 range (sn_mlp_wide_overflow). */
int sn_mlp_wide_forward_train_f16x3(const sn_mlp_desc *mlp, const float *x, uint32_t N, float *const *hidden, uint32_t *const *sign_bits,
                                    float *out, void *workspace, size_t workspace_bytes, sn_stream_t stream);
/* sign_bits (NULL, or a host array of nl-1 device pointers, entries may be NULL): sign_bits[l] [N, 8] uint32, 16-byte aligned, receives one
 * bit per hidden unit of layer l (set = output > 0) in the kernels' register order -- an opaque companion of hidden[l] for
 * sn_mlp_wide_backward_bits, which then reads 32 bytes per row and layer instead of 1 KiB. */

/* Backward-data pass of a 256-wide perceptron without skip layers (the autograd of nerf/network.py:31-66 for the per-sample
 * mask head in training, trainer.py:401-428) in one kernel: grad_out [N, dims[nl]] -> grad_in [N, dims[0]], and for every
 * hidden layer l < nl-1 grad_hidden[l] [N, 256] = d loss / d (pre-activation of layer l) -- what sn_linear_wgrad needs.
 * hidden[l] [N, 256] = the forward's saved (post-activation) output of layer l: its sign selects the LeakyReLU / ReLU branch
 * exactly as torch's in-place backward does.  Products as split-fp16 with fp32 accumulation on the matrix cores; every row is
 * scaled by a power of two (exact) so that small gradient rows keep their precision.  hidden / grad_hidden are host arrays of
 * nl-1 device pointers. */
size_t sn_mlp_wide_backward_workspace_bytes(const sn_mlp_desc *mlp);
int sn_mlp_wide_backward_bits(const sn_mlp_desc *mlp, const float *grad_out, const uint32_t *const *sign_bits, uint32_t N,
                              float *grad_in, float *const *grad_hidden, void *workspace, size_t workspace_bytes, sn_stream_t stream);
int sn_mlp_wide_backward(const sn_mlp_desc *mlp, const float *grad_out, const float *const *hidden, uint32_t N,
                         float *grad_in, float *const *grad_hidden, void *workspace, size_t workspace_bytes, sn_stream_t stream);

/* The reference's SMALL bias-free ReLU perceptrons under training (nerf/network.py:9-29 `MLP` as instantiated at network.py:94, 98, 143:
 * grid_mlp 32-64-64-16, view_mlp 31-32-32-3, prop_mlp 10-16-1; also BASELINE configs[0]'s 16-32-16 / 31-32-3), one kernel per direction on
 * the matrix cores in true fp32 (v_mfma_f32_32x32x2_f32: exact products, fixed summation order), replacing F.linear + F.relu per layer.
 *   forward:  out [N, dL] = raw output; hidden[l] [N, d_{l+1}] = post-ReLU output of layer l (l < L-1), 16-byte aligned.
 *             act: SN_SMALL_ACT_TRUNC_EXP0  aux_out [N]     = exp(out[:, 0])                      (trunc_exp, activation.py:5-11, network.py:155,179)
 *                  SN_SMALL_ACT_SIGMOID_BG  aux_out [N, dL] = sigmoid(out) + (1 - aux_in[n]) * bg  (renderer.py:349-353; aux_in = weights_sum, dL <= 4)
 *   backward: the whole data path.  grad_out [N, dL] (gradient of `out`, may be NULL) and grad_aux (gradient of aux_out, may be NULL; the
 *             activation's derivative is applied here: trunc_exp's exp(clamp(x, -15, 15)), activation.py:13-17) are combined into
 *             grad_last [N, dL] (gradient of the last pre-activation), then grad_hidden[l] [N, d_{l+1}] = gradient of layer l's pre-activation
 *             (ReLU branch from the sign of hidden[l], as torch's in-place backward) and grad_in [N, d0] (NULL: not wanted).
 *             grad_aux_in [N] (act SIGMOID_BG, may be NULL) = gradient of aux_in.  Weight gradients: sn_linear_wgrad on
 *             (x, grad_hidden[0]), (hidden[l-1], grad_hidden[l]), (hidden[L-2], grad_last).
 * sn_mlp_small_supported: 1 if this build instantiates the descriptor's widths (others: SN_ERR_UNSUPPORTED, the caller keeps its own layers). */
enum { SN_SMALL_ACT_NONE = 0, SN_SMALL_ACT_TRUNC_EXP0 = 1, SN_SMALL_ACT_SIGMOID_BG = 2 };
int sn_mlp_small_supported(const sn_mlp_desc *mlp);
int sn_mlp_small_forward_train(const sn_mlp_desc *mlp, const float *x, uint32_t N, float *const *hidden, float *out, int32_t act,
                               const float *aux_in, float bg, float *aux_out, sn_stream_t stream);
int sn_mlp_small_backward(const sn_mlp_desc *mlp, const float *grad_out, const float *grad_aux, int32_t act, const float *out_raw, float bg,
                          const float *const *hidden, uint32_t N, float *grad_in, float *const *grad_hidden, float *grad_last, float *grad_aux_in,
                          sn_stream_t stream);

/* General fp32 matrix product on the matrix cores in true fp32 (v_mfma_f32_32x32x2_f32: exact products, one k-ascending chain per output element,
 * deterministic) -- the route of every layer shape the specialised kernels do not instantiate; replaces torch.nn.functional.linear / `@`
 * (nerf/network.py:9-66 with widths other than the reference's):
 *   c[i * c_row + j] = act(sum_k a[i * a_row + k * a_col] * b[k * b_row + j * b_col] + bias[j]),  i < M, j < N, k < K
 * One stride of a and one of b must be 1.  nn.Linear forward: a = x (K, 1), b = weight [N,K] (1, K), bias or NULL; input gradient: a = dy (N, 1),
 * b = weight (K, 1) with K and N swapped; weight gradient: a = dy (1, N) transposed, b = x (K, 1).  act: 0 none, 1 ReLU, 2 leaky ReLU (0.01). */
int sn_gemm_f32(const float *a, int64_t a_row, int64_t a_col, const float *b, int64_t b_row, int64_t b_col, const float *bias, int32_t act,
                uint32_t M, uint32_t N, uint32_t K, float *c, int64_t c_row, sn_stream_t stream);

/* Weight gradient of an nn.Linear over a training batch: dw[N,K] = dy[M,N]^T x[M,K], fp32, summed in a fixed order
 * (deterministic).  K, N <= 64 (the radiance / proposal MLPs, nerf/network.py:9-29): register-tiled VALU kernel.
 * Otherwise N <= 256, any K (the per-sample mask head and the SAM head, network.py:31-66): v_mfma_f32_32x32x2_f32 over
 * row slabs, one per CU.  workspace: >= sn_linear_wgrad_workspace_bytes() (0 = shape not supported), holds the
 * per-slab partial results. */
size_t sn_linear_wgrad_workspace_bytes(uint32_t M, uint32_t K, uint32_t N);
int sn_linear_wgrad(const float *x, const float *dy, uint32_t M, uint32_t K, uint32_t N, float *dw,
                    void *workspace, size_t workspace_bytes, sn_stream_t stream);

/* Dense Adam step of one parameter tensor in a single pass: the single-tensor recipe of torch.optim.Adam (the reference's
 * optimiser, main.py:283: Adam(params, eps=1e-15), betas (0.9, 0.999), no amsgrad) -- exp_avg.lerp_(g, 1-beta1);
 * exp_avg_sq = beta2 * exp_avg_sq + (1-beta2) g^2; param -= lr / (1-beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1-beta2^step) + eps)
 * -- with 16-byte accesses, 4 streams read and 3 written.  Same dense semantics (untouched rows keep moving by their
 * momentum); elements whose gradient and both moments are exactly zero are skipped (their update is exactly zero).
 * Hyper-parameters are doubles like the Python floats torch derives its scalars from.  step counts from 1.  step_device != NULL
 * (capturable form, like torch.optim.Adam(capturable=True)): the count is read from that device float when the kernel RUNS and `step` is
 * ignored, so that the launch can be captured in a HIP graph and replayed.
 * flags: SN_ADAM_ZERO_GRAD -- the gradient is cleared in the same pass (for callers that accumulate in place);
 *        SN_ADAM_LAZY -- opt-in touched-elements-only update (NOT the reference's optimiser): an element whose gradient
 *        is exactly zero in this step is skipped altogether (moments do not decay, the parameter does not move) -- the semantics of
 *        torch.optim.SparseAdam with the non-zeros of the dense gradient as the sparse pattern; requires weight_decay = 0. */
#define SN_ADAM_ZERO_GRAD 1
#define SN_ADAM_LAZY 2
int sn_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, double lr, double beta1, double beta2,
                 double eps, double weight_decay, uint32_t step, const float *step_device, int maximize, int flags, sn_stream_t stream);

/* Measurement hook (bench.py): bracket every kernel sn_rm_render_rays launches with hipEvents on the
 * caller's stream.  Classes: 0 weight pack, 1..3 proposal stage k, 4 final stage.  profile_read
 * synchronises, returns summed device milliseconds and launch counts per class, and resets. */
void sn_rm_profile_enable(int on);
int sn_rm_profile_read(float *ms_per_class, int32_t *launches_per_class, int n_classes);
/* Shader clock the last profiled final-stage launch ran at, MHz: workgroup 0 reads s_memtime (shader cycles) and
 * s_memrealtime (constant wall-clock rate) at entry and exit; shader_mhz = their ratio x the wall-clock rate.  probe_ms
 * (optional) = the wall time the probe spanned.  0 when no final stage ran since sn_rm_profile_enable(1).  Synchronises. */
int sn_rm_profile_shader_clock(float *shader_mhz, float *probe_ms);
/* Test hook: the library's device numerics primitives element-wise, y[i] = f(a[i][, b[i]]).  op 0: exp (the shared
 * deterministic recipe, oracle orc_expf); 1: a / b as compiled (IEEE-rounded); 2 / 3: the Mip-360 spacing function and its
 * inverse (renderer.py:249-252). */
int sn_debug_eval(int op, const float *a, const float *b, uint32_t n, float *y, sn_stream_t stream);
/* Diagnostics: occupancy-API workgroups/CU of the fused kernels (prop, final f16x3, final f32-MFMA) and their dynamic LDS bytes. */
int sn_rm_debug_occupancy(int32_t *out, int32_t *lds, int n);

#ifdef __cplusplus
}
#endif
#endif /* SANERF_HIP_H */
