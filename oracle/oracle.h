/*
 * oracle.h — CPU restatement of the SANeRF-HQ volumetric-rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sanerf-hq_amd/ may include, link or
 * call this.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.  See oracle/README.md for how the oracle is pinned.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  Layouts are the reference's own.
 */
#ifndef SANERF_ORACLE_H
#define SANERF_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 32
#define ORC_MAX_LAYERS 8
#define ORC_MAX_STAGES 4

/* ---- scalar helpers (exposed for tests) -------------------------------- */
float    orc_expf(float x);                 /* the build's specified fp32 exp */
float    orc_half_to_float(uint16_t h);
uint16_t orc_float_to_half(float f);
/* gridencoder.cu:133 — kernel-side level resolution, fp32 recipe */
uint32_t orc_level_resolution(uint32_t level, float S, uint32_t H);
/* gridencoder.cu:45-79 — table row of one grid vertex (dense walk while the stride fits, else the coherent hash; % size) */
uint32_t orc_grid_row(uint32_t gridtype, uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid, uint32_t D);

/* ---- gridencoder (gridencoder.cu) --------------------------------------
 * table_dtype: 0 = float32, 1 = float16 storage (arithmetic is fp32 either way)
 * outputs / grad are [L,B,C] float32 (gridencoder.cu:399), dy_dx is [B,L,D,C].   */
void orc_grid_encode_forward(const float *inputs, const void *embeddings, int table_dtype,
                             const int32_t *offsets, float *outputs,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                             float S, uint32_t H, float *dy_dx,
                             uint32_t gridtype, int align_corners, uint32_t interp);
void orc_grid_encode_backward(const float *grad, const float *inputs, const void *embeddings, int table_dtype,
                              const int32_t *offsets, float *grad_embeddings,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                              float S, uint32_t H, const float *dy_dx, float *grad_inputs,
                              uint32_t gridtype, int align_corners, uint32_t interp);
void orc_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                              const int32_t *offsets, float weight,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                              float S, uint32_t H, uint32_t gridtype, int align_corners);
void orc_grad_weight_decay(const float *embeddings, float *grad, const int32_t *offsets,
                           float weight, uint32_t B, uint32_t C, uint32_t L);

/* ---- shencoder (shencoder.cu) ------------------------------------------ */
void orc_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D,
                           uint32_t degree, float *dy_dx);
/* test helper: dy_dx of the same encoder by a 4th-order central difference of the fp64 forward (independent check of the closed forms) */
void orc_sh_dy_dx_fd(const float *inputs, uint32_t B, uint32_t D, uint32_t degree, float *dy_dx);
void orc_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D,
                            uint32_t degree, const float *dy_dx, float *grad_inputs);

/* ---- freqencoder (freqencoder.cu / encoding.py:6-44) ------------------- */
void orc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float *outputs);
void orc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D,
                              uint32_t deg, uint32_t C, float *grad_inputs);

/* ---- ray-marching pieces (nerf/utils.py, nerf/renderer.py) ------------- */
void orc_generate_rays(const float *pose /*4x4 row-major*/, float fx, float fy, float cx, float cy,
                       uint32_t H, uint32_t W, float *rays_o, float *rays_d);
void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                            float min_near, uint32_t N, float *nears, float *fars);
void orc_contract(const float *x, uint32_t N, float *z);
void orc_linspace(float start, float end, uint32_t steps, float *out);
/* sample_pdf (renderer.py:84-119).  bins [N,T0+1], weights [N,T0] -> out_bins [N,T];
 * inds (int32 [N,T]) optional; u (float [T]) optional table, NULL => orc_linspace recipe. */
void orc_sample_pdf(const float *bins, const float *weights, uint32_t N, uint32_t T0, uint32_t T,
                    const float *u, float *out_bins, int32_t *inds);
/* sigmas -> weights (renderer.py:308-325). real_bins [N,T+1], sigmas [N,T] -> weights [N,T] */
void orc_weights_from_sigma(const float *real_bins, const float *sigmas, uint32_t N, uint32_t T,
                            int last_sample_opaque, float *weights);

/* ---- tiny MLPs (network.py:9-66) --------------------------------------- */
typedef struct {
    const float *weight[ORC_MAX_LAYERS];  /* [out,in] row-major (nn.Linear.weight) */
    const float *bias[ORC_MAX_LAYERS];    /* NULL = no bias */
    uint32_t     dims[ORC_MAX_LAYERS + 1];/* dims[0]=input, dims[l+1]=output of layer l (layer fan-out) */
    uint32_t     num_layers;
    uint32_t     activation;              /* 0 relu (MLP), 1 leaky_relu 0.01 (SkipConnMLP) */
    uint32_t     skip_mask;               /* bit l set: layer l input = cat[h, x_in] (SkipConnMLP) */
} orc_mlp;
void orc_mlp_forward(const orc_mlp *m, const float *x, uint32_t B, float *y);

typedef struct {
    const void    *embeddings;
    int            table_dtype;           /* 0 f32, 1 f16 */
    int32_t        offsets[ORC_MAX_LEVELS + 1];
    uint32_t       D, C, L;
    float          S;                     /* (float)log2(per_level_scale) */
    uint32_t       H;                     /* base resolution */
    uint32_t       gridtype, align_corners, interp;
} orc_grid;

typedef struct {
    uint32_t num_stages;                  /* len(opt.num_steps) */
    uint32_t num_steps[ORC_MAX_STAGES];
    orc_grid prop_grid[ORC_MAX_STAGES];   /* stages 0..num_stages-2 */
    orc_mlp  prop_mlp[ORC_MAX_STAGES];
    orc_grid grid;                        /* main field */
    orc_mlp  grid_mlp;                    /* -> [sigma_raw, geo_feat...] */
    orc_mlp  view_mlp;                    /* [geo_feat, SH] -> rgb (per ray, after compositing) */
    uint32_t sh_degree;
    float    aabb[6];
    float    min_near;
    float    bound;                       /* grid bound (2 when contracted) */
    int      contract;
    int      last_sample_opaque;          /* opt.background == 'last_sample' */
    float    bg_color;
    /* optional heads (renderer.py:359-385) */
    int      with_sam;  orc_grid s_grid;  orc_mlp samvit_mlp; const float *ln_weight, *ln_bias; float ln_eps;
    int      with_mask; orc_grid m_grid;  orc_mlp mask_mlp;
} orc_render_cfg;

typedef struct {                          /* all optional (NULL = skip); layouts [N,...] row-major */
    float   *nears, *fars;                /* [N] */
    float   *bins[ORC_MAX_STAGES];        /* [N,T_k+1] normalised bins of stage k */
    float   *real_bins[ORC_MAX_STAGES];   /* [N,T_k+1] */
    float   *sigmas[ORC_MAX_STAGES];      /* [N,T_k] */
    float   *weights[ORC_MAX_STAGES];     /* [N,T_k] */
    int32_t *inds[ORC_MAX_STAGES];        /* [N,T_k+1], stage k>=1: searchsorted result that produced bins[k] */
    float   *xyzs_last;                   /* [N,T_last,3] contracted sample positions of the last stage */
    float   *f_image;                     /* [N, geo+sh] */
    const float *u_table[ORC_MAX_STAGES]; /* INPUT, optional: u used by sample_pdf producing bins[k] ([T_k+1]) */
    const float *bins0_table;             /* INPUT, optional: stage-0 bins ([T_0+1]) */
} orc_render_debug;

/* Whole path: renderer.py:221-385 (run) for perturb=False.
 * out: image [N,3], depth [N], weights_sum [N]; samvit [N,256] / mask_logits [N,n_inst] if heads on. */
void orc_render_rays(const orc_render_cfg *cfg, const float *rays_o, const float *rays_d, uint32_t N,
                     const float *cam_near_far /* [N,2] or NULL */,
                     float *image, float *depth, float *weights_sum,
                     float *samvit, float *mask_logits,
                     orc_render_debug *dbg);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
