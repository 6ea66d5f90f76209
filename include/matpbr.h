/*
 * matpbr.h -- C ABI of libmatpbr.so: MI355X (gfx950) differentiable PBR shading kernels.
 *
 * This is the drop-in boundary for the hot path of lez-s/Materialist.  Every
 * entry point names the reference interface it replaces (file:line into the reference tree).
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers owned by the caller (PyTorch), fp32, contiguous,
 *     in the reference's row-major HWC layout: a[B,H,W,3] r[B,H,W,1] m[B,H,W,1] n[B,H,W,3]
 *     (myutils/mi_plugin.py:1238-1241 holds the same maps as TensorXf [512,512,C]); B = `batch`
 *     independent images (the reference processes one image per process; B>1 is the build's batching).
 *   - `light` is [B, n_light, 3]: MATPBR_LIGHT_SH25 = 25 real SH coefficients per colour channel in
 *     the convention of myutils/computeSH.py:13-68 (index l(l+1)+m), directions mapped by
 *     theta = acos(y), phi = atan2(x,-z) (myutils/envmap_utils.py:29-36).
 *   - `stream` is the caller's hipStream_t (NULL = default stream).  Entry points only enqueue work:
 *     no allocation, no synchronisation, no global mutable state -> usable under hipGraph capture
 *     and re-entrant per stream.  (The ONE process-wide switch the library has is a measurement aid and lives in
 *     matpbr_experimental.h: matpbr_mlp_set_lds_dma, between main loops of the bf16 layer kernels that produce the same bits.)
 *   - Return value: 0 = MATPBR_OK, negative = error (matpbr_strerror); nothing throws.
 *   - `spp` (even, 2..MATPBR_MAX_SPP) is the reference's samples-per-pixel argument
 *     (inverse_img_w_mi.py:59,69,625): here it sizes the deterministic quadrature rules of the two BRDF lobes
 *     (spp = 64 -> 5 x 4 GGX half vectors + 3 x 6 cosine-weighted directions), chosen so that the
 *     error against the converged integral is below that of spp random BSDF samples.
 */
#ifndef MATPBR_H
#define MATPBR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MATPBR_VERSION 311 /* round 6: the coordinate MLP's entry points moved to matpbr_mlp.h, five unused ones removed; lazy state 120 B/pixel, folded
                              models 68 B/pixel in a new layout (opaque storage: size it with the *_bytes queries); 311: a folded phase stores no
                              render (`pred` is written by matpbr_brdf_phase_resolve) */
#define MATPBR_MAX_SPP 128
#define MATPBR_NSH 25

enum {
    MATPBR_OK = 0,
    MATPBR_ERR_INVALID_ARG = -1,
    MATPBR_ERR_UNSUPPORTED = -2,
    MATPBR_ERR_LAUNCH = -3,
    MATPBR_ERR_WORKSPACE = -4,
};

enum { MATPBR_LIGHT_SH25 = 0, MATPBR_LIGHT_SH9 = 1, MATPBR_LIGHT_ENV_TEXELS = 2 };

/* flags */
#define MATPBR_FLAG_CLAMP_PARAMS 1u /* maps are raw optimiser parameters: render clamp(a,0,1), clamp(r,.07,1), clamp(m,0,1)
                                       (inverse_img_w_mi.py:371-377) */
#define MATPBR_FLAG_ATTACHED_SAMPLING 2u /* matpbr_shade_bwd only: d_r is the derivative of the rendered value THROUGH the GGX quadrature
                                           nodes (their half-vector angles move with r, and with them wi, the weights and the radiance),
                                           the convention of the live reference (myutils/mi_plugin.py:227-230,1335-1341), instead of the
                                           stop-gradient convention of the default.  d_a, d_m, d_n, d_light do not change. */
#define MATPBR_FLAG_LAZY_FORCE 16u /* matpbr_shade_fwd_lazy: rebuild the model of every pixel (first render of a part; `lazy_state` is not read) */
#define MATPBR_FLAG_MODELS_READY 64u /* MatpbrBrdfPhase.flags: the caller has already built (and possibly edited) what the step with t == 1 would
                                        build -- the models in lazy_state (matpbr_shade_fwd_lazy with MATPBR_FLAG_LAZY_FORCE), or jac + s1cache of a
                                        part without MATPBR_PART_R (matpbr_shade_fwd_keep): pixels without geometry are given constant models */
#define MATPBR_FLAG_ROTATE_BEST 128u /* MatpbrBrdfPhase.flags, pred_next mode only: SaveBest without copies.  The live maps of the part and the render
                                        live in two buffers each -- (pa, best_a), (pr, best_r), (pm, best_m), (pred, pred_next) -- and an improving
                                        iteration declares the buffers it has just read "best" and writes the new values into the others (any other
                                        iteration updates in place); nothing is copied in the loop (20 of 217 B/pixel of an improving 'rm' iteration,
                                        32 of 150 in an 'a' part).  The caller does NOT swap pred / pred_next between steps and must call
                                        matpbr_brdf_phase_resolve before it reads pa / pr / pm / best_* / best_img / pred or starts the next part
                                        (a folded part stores no render at all: every reader of `pred` calls resolve, with or without this flag) */
#define MATPBR_FLAG_GENERIC_STEP 256u /* MatpbrBrdfPhase.flags: keep the pred_next step on the generic models even where the part has a folded form
                                        (`lazy_fold`): A/B measurements and tests of one form against the other */
#define MATPBR_FLAG_JAC32 512u     /* matpbr_shade_fwd_lazy: `jac16` receives the NINE fp32 planes of matpbr_shade_fwd_ex's `jac` instead (matpbr_plane9_bytes()):
                                        what matpbr_shade_bwd_jac reads -- the operator face (render_w_brdf under autograd, inverse_img_w_mi.py:69-80)
                                        differentiates with full-precision P and S0 - S1 (d out / d m is a difference of the two) */
#define MATPBR_FLAG_SHARE_GPU 2048u /* MatpbrBrdfPhase.flags: the folded step runs on at most 512 workgroups (two per CU) instead of 1024, leaving wave
                                       slots and registers on every CU to launches of OTHER streams -- for a batch cut into groups that step on streams
                                       of their own (loop.PipelinedBrdfPhase): one group's walk and statistics launches then run under another's step */
#define MATPBR_FLAG_JAC16 32u      /* matpbr_brdf_loss_bwd_jac: `jac` holds the half-precision planes written by matpbr_shade_fwd_lazy */
#define MATPBR_PART_A 2u            /* which maps a BRDF phase optimises (`optimize_part`, inverse_img_w_mi.py:343-357) */
#define MATPBR_PART_R 4u
#define MATPBR_PART_M 8u
#define MATPBR_PART_N 1024u         /* matpbr_brdf_normal_step only: the part moves the normal map ('n' in the part, use_mesh_normal False) */
/* floats per image in the loss statistics buffer (device memory, caller-owned, persistent across iterations):
 *  0 ratio  1 mse  2 l1  3 l1/mse  4 L1(a)  5 L1(r)  6 L1(m)  7 loss  8 improved(0/1)  9 best_mse (init +inf)
 * 10 early-stopping counter  11 early-stopping best  12 early-stopping has-best
 * 13 stopped: 0 running, 1 EarlyStopping fired in the last executed iteration (which still ran to its end, as the reference's loop
 *    does), 2 stopped before the last enqueued iteration (host test: > 0.5)  14 iterations run
 * 15 sum(gt) over the image (set by the caller; used by matpbr_brdf_phase_step) */
#define MATPBR_STATS_STRIDE 16

/* Pinhole camera of the reference: camera at the origin looking down -z, +y up
 * (inverse_img_w_mi.py:31-39, myutils/default_cam.json), focal = (W/2)/tan(fov_x/2),
 * principal point ((W-1)/2, (H-1)/2) (myutils/mesh_recon.py:17-25). */
typedef struct MatpbrCamera {
    float fov_x_deg; /* 35 in the reference */
} MatpbrCamera;

int matpbr_version(void);
const char* matpbr_strerror(int code);

/* Forward render.  Replaces the forward half of
 *   render_w_brdf(scene, albedo, roughness, metallic, normal, spp)   inverse_img_w_mi.py:69-80
 *   render_envmap(scene, envmap, spp)                                inverse_img_w_mi.py:59-67
 * i.e. mi.render(scene, params, spp) with MatDiffBSDF (myutils/mi_plugin.py:1229-1475) under the
 * build's deterministic definition.  out_rgb[B,H,W,3] = linear radiance.
 *
 * matpbr_shade_fwd_ex additionally takes / produces two per-pixel plane buffers ([9][B*H*W] floats, matpbr_plane9_bytes()):
 *   dcache (nullable, input)  the diffuse-lobe coefficients A0, A1, A2 (rgb each) of matpbr_diffuse_cache(): the diffuse lobe
 *                             integrates to a(1-m)(A0 + r A1 + r^2 A2) with coefficients that depend on (n, view, light) only, so
 *                             while light and geometric normals are fixed (a whole BRDF phase, inverse_img_w_mi.py:317-342) the
 *                             render evaluates the GGX-lobe samples only;
 *   jac (nullable, output)    P_c, S0_c - S1_c, d out_c / d r: what matpbr_shade_bwd_jac / matpbr_brdf_loss_bwd_jac need to form
 *                             the material gradients of THIS forward pass without walking any sample again. */
int matpbr_shade_fwd(const float* a, const float* r, const float* m, const float* n, const float* light,
                     int light_kind, int n_light, float* out_rgb, int H, int W, int batch, int spp,
                     const MatpbrCamera* cam, uint32_t flags, void* stream);
size_t matpbr_plane9_bytes(int H, int W, int batch);
int matpbr_shade_fwd_ex(const float* a, const float* r, const float* m, const float* n, const float* light, int light_kind,
                        int n_light, const float* dcache, float* out_rgb, float* jac, int H, int W, int batch, int spp,
                        const MatpbrCamera* cam, uint32_t flags, void* stream);
int matpbr_diffuse_cache(const float* n, const float* light, int light_kind, int n_light, float* dcache, int H, int W, int batch,
                         int spp, const MatpbrCamera* cam, void* stream);
/* While r, the shading normals and the light stay as they are (the parts of --opt_order without 'r', inverse_img_w_mi.py:317-342,
 * 497-504), the specular sums of every pixel are constants:
 *   matpbr_shade_fwd_keep     matpbr_shade_fwd_ex that also writes S1 into `s1` ([3][B*H*W] floats; jac required)
 *   matpbr_shade_fwd_cached   the render for new a, m from the planes `jac` (0-5) and `s1` of that call: the same fused operations in
 *                             the same order, bit-identical to walking the samples again; 44 + 36 B/pixel, no samples */
int matpbr_shade_fwd_keep(const float* a, const float* r, const float* m, const float* n, const float* light, int light_kind,
                          int n_light, const float* dcache, float* out_rgb, float* jac, float* s1, int H, int W, int batch, int spp,
                          const MatpbrCamera* cam, uint32_t flags, void* stream);
int matpbr_shade_fwd_cached(const float* a, const float* m, const float* jac, const float* s1, float* out_rgb, int H, int W, int batch,
                            uint32_t flags, void* stream);

/* The same render WITHOUT walking the GGX samples of every pixel in every iteration, for the parts of --opt_order that move the roughness
 * (inverse_img_w_mi.py:371-386, 493-515).  Light and shading normals are fixed during a BRDF phase (:317-342), so the specular sums of a
 * pixel are functions of its roughness alone, and Adam moves the roughness by 1e-4 .. 3e-4 per step.  `lazy_state`
 * (matpbr_lazy_state_bytes(); opaque, owned by the caller, must persist between calls) holds a local model per pixel: the sums at the
 * roughness r_ref they were last sampled at, their slopes in r, the r-derivatives of the backward convention, and a validity
 * interval around r_ref.  A call renders every pixel whose roughness is still inside its interval from the model (a streaming pass),
 * walks the samples of the others again (1-2 % of the pixels per iteration on recorded runs), rebuilds their models, and writes
 *   out_rgb  the render; differs from matpbr_shade_fwd_ex by < 1e-3 max(|exact|, mean|exact|) on every pixel (the intervals are built
 *            for a quarter of that; oracle/matpbr_oracle.c `lazy_refresh_pixel` is the specification, tests/test_gpu_lazy.py the gate)
 *   jac16    5 planes of B*H*W 32-bit words: half2 (P_c, S0_c - S1_c) x rgb, half2 (JR_0, JR_1), half2 (JR_2, 0), what
 *            matpbr_brdf_loss_bwd_jac(..., MATPBR_FLAG_JAC16) reads (matpbr_plane9_bytes() is room enough)
 *   sums     (nullable) [B][matpbr_lazy_sums_count()] partial sums of out_rgb, for mean(pred) (:388)
 * `dcache` = matpbr_diffuse_cache(n, light); flags: MATPBR_FLAG_CLAMP_PARAMS, MATPBR_FLAG_LAZY_FORCE (first call with this state, or after
 * light / normals / dcache changed).  The parity scale needs the mean radiance of the image: with `stats` (layout above, nullable) it is
 * 0.5 (stats[15] / stats[0]) / (3 H W) per image, else `floor` (> 0).  `tol` scales the interval tolerances (1 = as specified).
 *   matpbr_lazy_state_unpack  test / inspection: the models in the oracle's layout [B*H*W][22] (r_ref, lo, hi, rho, SD, S1, gSD, gS1,
 *                             dSD, dS1) and / or refreshed[B*H*W] = 1 for the pixels whose samples the last call walked
 *   matpbr_jac16_unpack       jac16 -> the nine fp32 planes of matpbr_shade_fwd_ex's `jac` */
size_t matpbr_lazy_state_bytes(int H, int W, int batch);
/* Storage of a part's FOLDED models (MatpbrBrdfPhase.lazy_fold; opaque, caller-owned, 64 B/pixel): inside one part of --opt_order
 * (inverse_img_w_mi.py:343-357) the maps the part does not move are constants and fold into the per-pixel models -- the albedo in parts of
 * r / m, roughness and metallic in part 'a' -- so the step reads 64 (24) instead of 80 + 12 bytes of model and map per pixel. */
size_t matpbr_lazy_fold_bytes(int H, int W, int batch);
int matpbr_lazy_sums_count(int H, int W);
int matpbr_shade_fwd_lazy(const float* a, const float* r, const float* m, const float* n, const float* light, int light_kind, int n_light,
                          const float* dcache, void* lazy_state, float* out_rgb, void* jac16, const float* stats, float* sums, int H, int W,
                          int batch, int spp, const MatpbrCamera* cam, uint32_t flags, float floor, float tol, void* stream);
int matpbr_lazy_state_unpack(const void* lazy_state, float* state28, int* refreshed, int H, int W, int batch, void* stream);
int matpbr_jac16_unpack(const void* jac16, float* jac, int H, int W, int batch, void* stream);

/* Backward render.  Replaces the AD pass that `loss.backward()` drives through dr.wrap_ad / mi.render
 * (inverse_img_w_mi.py:59,69,248,420,544).  Any of d_a/d_r/d_m (all three or none), d_n, d_light may be
 * NULL to skip that gradient.  d_light[B,n_light,3] is overwritten (not accumulated); it needs
 * `workspace` of matpbr_shade_bwd_workspace_bytes() bytes (device memory owned by the caller, contents
 * undefined on entry and exit: per-workgroup partial sums of the light gradient). */
int matpbr_shade_bwd(const float* a, const float* r, const float* m, const float* n, const float* light,
                     int light_kind, int n_light, const float* d_out_rgb, float* d_a, float* d_r, float* d_m,
                     float* d_n, float* d_light, void* workspace, size_t workspace_bytes, int H, int W, int batch,
                     int spp, const MatpbrCamera* cam, uint32_t flags, void* stream);
size_t matpbr_shade_bwd_workspace_bytes(int H, int W, int batch, int n_light);
/* The material gradients of a forward pass that wrote `jac`: one streaming pass, no samples (same a, r, m as that pass). */
int matpbr_shade_bwd_jac(const float* a, const float* r, const float* m, const float* jac, const float* d_out_rgb, float* d_a,
                         float* d_r, float* d_m, int H, int W, int batch, void* stream);

/* Fused pieces of hot loop B in `--model_name none` mode (inverse_img_w_mi.py:371-432): everything between the two
 * renders of one optimisation iteration, without leaving the GPU.
 *   matpbr_brdf_loss_stats     ratio = mean(gt)/mean(pred); MSE / L1 of (pred*ratio)^(1/2.2) against gt^(1/2.2); L1 of the clamped
 *                              parameter maps against their initial values; loss = 3 (L1/MSE) MSE + L1 + scale_delta * sum(L1 reg)
 *                              (:388-418); SaveBest's strict `<` on the MSE (myutils/misc.py:75) -> stats[B, MATPBR_STATS_STRIDE].
 *   matpbr_brdf_loss_bwd_jac   matpbr_shade_bwd_jac with d loss/d pred formed in-kernel from those statistics, the regulariser
 *                              gradients added, torch.clamp's gradient gating applied, and best_* (nullable) snapshotted when
 *                              stats says the iteration improved.  pa/pr/pm are the raw (unclamped) parameter maps that
 *                              matpbr_shade_fwd_ex rendered with MATPBR_FLAG_CLAMP_PARAMS; `jac` is that render's.
 *   matpbr_adam_step           torch.optim.Adam update of one tensor (:359); `step` is the 1-based iteration count. */
size_t matpbr_brdf_loss_workspace_bytes(int batch);
int matpbr_brdf_loss_stats(const float* pred, const float* gt, const float* gt_srgb, const float* pa, const float* pr,
                           const float* pm, const float* a0, const float* r0, const float* m0, float scale_delta,
                           float* stats, void* workspace, size_t workspace_bytes, int H, int W, int batch, uint32_t flags,
                           void* stream);   /* flags: MATPBR_PART_* of the maps being optimised (none set = all three; with MATPBR_PART_N: exactly the
                                                material maps named beside it) */
/* matpbr_brdf_loss_stats with the EarlyStopping state machine of matpbr_brdf_phase_step kept in `stats` (es_patience >= 0; myutils/misc.py:37-60)
 * and the per-iteration loss_mse in history[hist_len, B] (nullable): once stats[b][13] is set the image's statistics rest, its `improved` flag stays
 * down (no snapshot in matpbr_brdf_loss_bwd_jac) and the AdamW step with a statistics row (matpbr_mlp.h) rests too, so a caller whose iteration contains
 * kernels of its own (the PosMLP of --model_name pos_mlp, inverse_img_w_mi.py:471-566) may poll the flag every few iterations: the iterations
 * enqueued past the stop change nothing.  es_patience == 0 disables stopping (the state machine still counts iterations). */
int matpbr_brdf_loss_stats_es(const float* pred, const float* gt, const float* gt_srgb, const float* pa, const float* pr, const float* pm,
                              const float* a0, const float* r0, const float* m0, float scale_delta, float* stats, void* workspace,
                              size_t workspace_bytes, int H, int W, int batch, uint32_t flags, int es_patience, float es_min_delta, float* history,
                              int hist_len, void* stream);
int matpbr_brdf_loss_bwd_jac(const float* pa, const float* pr, const float* pm, const float* jac, const float* pred,
                             const float* gt_srgb, const float* stats, const float* a0, const float* r0, const float* m0,
                             float scale_delta, float* d_a, float* d_r, float* d_m, float* best_a, float* best_r, float* best_m,
                             float* best_img, int H, int W, int batch, uint32_t flags, void* stream);
int matpbr_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                     int step, void* stream);

/* One whole iteration of hot loop B (`model_name == 'none'`, inverse_img_w_mi.py:371-432) enqueued by a single call:
 * render(clamped params, + jac) -> loss statistics + SaveBest / EarlyStopping decisions -> streaming loss backward + Adam.
 * All pointers are device memory owned by the caller.  EarlyStopping (myutils/misc.py:37-60, es_patience > 0) runs on the
 * device: once stats[b][13] is set every kernel skips image b, so iterations may be enqueued ahead of the host's polling
 * without changing any decision.  `t` is the 1-based Adam step, `lr` the learning rate of this iteration. */
typedef struct MatpbrBrdfPhase {
    float *pa, *pr, *pm;                  /* raw parameter maps [B,H,W,3|1|1], updated in place */
    const float *n, *light;               /* shading-normal map [B,H,W,3], SH25 light [B,25,3] */
    const float *gt_srgb;                 /* target ^ (1/2.2) [B,H,W,3]; sum(gt) per image sits in stats[b][15] */
    const float *a0, *r0, *m0;            /* regulariser anchors = the maps at phase start (:196-201) */
    const float* dcache;                  /* matpbr_diffuse_cache(n, light) planes; NULL: recomputed inside every render */
    float* pred;                          /* [B,H,W,3] the iteration's render (scratch / output) */
    float* jac;                           /* matpbr_plane9_bytes() scratch */
    float *d_a, *d_r, *d_m;               /* gradients (output, nullable: the Adam update happens in the same pass) */
    float *adam_m[3], *adam_v[3];         /* Adam moments of a, r, m (needed for the maps in part_mask) */
    float *best_a, *best_r, *best_m, *best_img; /* SaveBest snapshot targets, nullable */
    float* stats;                         /* [B, MATPBR_STATS_STRIDE] */
    float* history;                       /* [hist_len, B] loss_mse of every iteration, nullable */
    void* workspace;
    size_t workspace_bytes;               /* >= matpbr_brdf_phase_workspace_bytes() */
    int H, W, batch, spp;
    float fov_x_deg, scale_delta;         /* 35, 0.1 in the reference */
    uint32_t part_mask;                   /* MATPBR_PART_A | _R | _M */
    int es_patience;                      /* 200 // loop_num (:361); <= 0 disables early stopping */
    float es_min_delta;                   /* 0.005 if 'a' in part else 0.001 (:360-363) */
    int hist_len;
    float* s1cache;                       /* nullable, 3 planes (a third of matpbr_plane9_bytes()): when part_mask has no MATPBR_PART_R the
                                             roughness, normals and light are constants of the part (:317-342): the step with t == 1 walks
                                             the samples and keeps the specular sums here and in `jac`; later steps combine them
                                             (bit-identical render, no samples).  The caller must not touch pr / jac / s1cache in between. */
    void* lazy_state;                     /* nullable, matpbr_lazy_state_bytes(): with it (and dcache) the render of every step of every part is
                                             matpbr_shade_fwd_lazy (the step with t == 1 builds the models) and `jac` holds its half-precision
                                             planes; a part without MATPBR_PART_R never re-samples a pixel after that (s1cache is then not used).
                                             NULL: every step walks the samples of every pixel (or combines s1cache). */
    float lazy_tol;                       /* scales the interval tolerances of the lazy render; <= 0: 1 */
    float* pred_next;                     /* nullable [B,H,W,3], with lazy_state: the step's last launch is the backward pass + Adam of this iteration AND
                                             the render of the next one from the updated parameters (maps, models and Adam state are read once per
                                             iteration; no jac planes).  On return `pred` holds this iteration's render as always and `pred_next` the
                                             next one (complete after the resampling stage); the caller SWAPS pred and pred_next before the
                                             next step, calls the steps with t = 1, 2, 3, ... and leaves workspace / lazy_state / pr / pm alone in between.
                                             The SaveBest snapshot of a map that the part does not optimise is not rewritten.
                                             An iteration is then two launches -- and a third in a part that moves the roughness: the pixels the backward
                                             launch lists (their roughness left their model's interval) are re-sampled by a launch of their own,
                                             MATPBR_STAGE_RESAMPLE -- : the partial sums of the loss statistics, and this launch, at whose head
                                             every workgroup folds them (fixed order) and forms the iteration's scalars from the SaveBest / EarlyStopping
                                             state of the iteration before, which lives in `workspace` in two alternating copies (read t-1, written t:
                                             the step with t = 1 copies `stats` in); `stats` is rewritten by one workgroup per image at every step. */
    uint32_t flags;                       /* MATPBR_FLAG_MODELS_READY; MATPBR_FLAG_ATTACHED_SAMPLING (pred_next mode only): d loss / d r through the GGX quadrature nodes -- the models'
                                             slopes are that derivative -- i.e. the live reference's gradient convention (myutils/mi_plugin.py:227-230,
                                             1335-1341) instead of the stop-gradient default */
    void* lazy_fold;                      /* nullable, matpbr_lazy_fold_bytes(), pred_next mode only: room for the part's folded models.  With it a part
                                             of r / m (d_a not requested) or part 'a' (d_r, d_m not requested) runs its iteration as ONE persistent
                                             streaming launch (at most 1024 workgroups, each streaming up to four 512-pixel blocks from two register
                                             sets) that also forms the NEXT iteration's loss statistics (round 6: no statistics launch after t = 1)
                                             and stores no render (`pred` is written by matpbr_brdf_phase_resolve; `pred_next` is scratch); in a part
                                             that moves the roughness the listed pixels are re-sampled by the MATPBR_STAGE_RESAMPLE launch behind it.
                                             136 ('rm') / 120 ('a') instead of 172 / 160 B/pixel.  The step with t == 1 derives the folded planes from
                                             the generic ones; the generic ones stay current (re-sampled pixels rewrite both).  Other parts, and
                                             MATPBR_FLAG_GENERIC_STEP, run the generic step. */
} MatpbrBrdfPhase;
/* A part of --opt_order that moves the normal map ('n', 'armn' under --model_name none with use_mesh_normal False; inverse_img_w_mi.py:356-432),
 * launch by launch without a framework in the iteration and with SaveBest / EarlyStopping on the device:
 *     matpbr_shade_fwd(ca, cr, cm, cn, light) -> pred                       the render under the current maps (:384-386)
 *     matpbr_brdf_loss_stats_es(pred, ..., pa, pr, pm, ...)                 ratio, losses, SaveBest / EarlyStopping decisions (:388-427)
 *     matpbr_brdf_loss_dpred(pred, gt_srgb, stats) -> d_pred                d loss / d pred from those statistics (:420)
 *     matpbr_shade_bwd(ca, cr, cm, cn, light, d_pred) -> d_a, d_r, d_m, d_n the backward render
 *     matpbr_brdf_normal_step(...)                                          regulariser gradients (incl. L1(normal, normal_ori), :410-411),
 *                                                                           clamp gating, NF.normalize's backward (:379), the snapshot of an
 *                                                                           improving iteration, Adam on the maps of the part (t = 1-based
 *                                                                           iteration, lr: StepLR on the host), and the maps of the next render
 * Images whose EarlyStopping fired in an earlier iteration rest in both entries (the firing iteration still updates, as in the reference).
 * `ln_part` (nullable, [batch][ceil(H W / 256)]): per-workgroup sums of |n - n0| -- stats' `loss` carries the three material regularisers,
 * the caller adds scale_delta sum / (3 H W) for the record (the optimisation does not read it). */
typedef struct MatpbrNormalStep {
    float *pa, *pr, *pm, *pn;              /* raw parameters [B,H,W,3|1|1|3]; the maps of the part are updated in place */
    float *ca, *cr, *cm, *cn;              /* what the renders take: clamp(pa,0,1), clamp(pr,.07,1), clamp(pm,0,1), normalize(pn); rewritten for the maps of the part */
    const float *d_a, *d_r, *d_m, *d_n;    /* matpbr_shade_bwd's gradients with respect to ca, cr, cm, cn (needed for the maps of the part) */
    const float *a0, *r0, *m0, *n0;        /* regulariser anchors (:189-201) */
    float *adam_m[4], *adam_v[4];          /* a, r, m, n */
    float *best_a, *best_r, *best_m, *best_n, *best_img;   /* nullable: SaveBest's snapshot (clamped maps, unit normals, pred ratio ^ (1/2.2)) */
    const float *pred, *stats;
    float* ln_part;
    int H, W, batch;
    uint32_t part_mask;                    /* MATPBR_PART_A | _R | _M | _N */
    float scale_delta;
} MatpbrNormalStep;
int matpbr_brdf_loss_dpred(const float* pred, const float* gt_srgb, const float* stats, float* d_pred, int H, int W, int batch, void* stream);
int matpbr_brdf_normal_step(const MatpbrNormalStep* step, int t, float lr, void* stream);

size_t matpbr_brdf_phase_workspace_bytes(int H, int W, int batch);
int matpbr_brdf_phase_step(const MatpbrBrdfPhase* phase, int t, float lr, void* stream);
/* After the step with t = t_done, put everything where the caller reads it.
 * MATPBR_FLAG_ROTATE_BEST: current parameters in pa / pr / pm, SaveBest's snapshots in best_a / best_r / best_m (maps of the part) and best_img
 * (= the tone-mapped render of the best iteration of this phase; untouched if none improved), the render of the CURRENT parameters (the next
 * iteration's) in `pred`.
 * A part with a folded form (`lazy_fold` given, no MATPBR_FLAG_GENERIC_STEP; with or without MATPBR_FLAG_ROTATE_BEST): its steps store NO render
 * (round 6: they form the next iteration's loss statistics themselves) -- `pred` is written HERE, from the per-pixel models at the current
 * parameters (the values the next step judges); under MATPBR_FLAG_ROTATE_BEST best_img is formed here too: from the models where they are exact in
 * the best values, from the renderer on SaveBest's maps where the roughness has moved (within the models' 1e-3 of what the copying form stores in
 * the improving iteration); `pred_next` is scratch.  The generic step leaves the best iteration's render in `pred_next` as before.
 * The phase may go on afterwards (steps t_done + 1, ...).  With t_done = 0, or a generic copying phase: nothing to do. */
int matpbr_brdf_phase_resolve(const MatpbrBrdfPhase* phase, int t_done, void* stream);
/* The same iteration stage by stage (profiling, and callers that interleave their own work): matpbr_brdf_phase_step enqueues all of them. */
#define MATPBR_STAGE_RENDER 1u   /* the render of the iteration (nothing to launch in the pred_next mode after t = 1) */
#define MATPBR_STAGE_STATS 2u    /* loss statistics, SaveBest / EarlyStopping decisions */
#define MATPBR_STAGE_BACKWARD 4u /* loss backward + Adam (+ the next iteration's render in the pred_next mode) */
#define MATPBR_STAGE_RESAMPLE 8u /* pred_next mode: the pixels the backward launch listed (their roughness left their model's interval) are re-sampled,
                                    their models rebuilt and their render written (nothing to launch in the other modes) */
int matpbr_brdf_phase_stages(const MatpbrBrdfPhase* phase, int t, float lr, uint32_t stages, void* stream);

/* One evaluation of hot loop A (inverse_img_w_mi.py:238-250) for a candidate light.  Materials and normals are fixed during the
 * phase (:216-220) and the render is linear in the light, so the phase works on the radiance transfer T of
 * matpbr_shade_transfer() (computed once at phase start): pred = T.light, loss = MSE + L1 on x^(1/2.2), SaveBest / EarlyStopping
 * decisions in `stats` (same layout; ratio = 1) and d loss / d light = T^T (d loss / d pred) -> d_light[B,25,3] in ONE pass over
 * T (300 B/pixel, HBM-bound).  The light's own parameterisation (envmap MLP or texels through softplus, then the SH projection)
 * and its optimiser stay with the caller, who back-propagates d_light through them.  `pred` (nullable) receives the render; the
 * best-so-far render is matpbr_relight(T, best light). */
size_t matpbr_env_phase_workspace_bytes(int H, int W, int batch);
int matpbr_env_phase_step(const float* T, const float* light, const float* gt_srgb, float* pred, float* d_light, float* stats,
                          float* history, int hist_len, int es_patience, float es_min_delta, void* workspace,
                          size_t workspace_bytes, int H, int W, int batch, void* stream);

/* One iteration of hot loop A with the `--model_name none` parameterisation of the light (inverse_img_w_mi.py:225-254 with the envmap MLP
 * replaced by its output activation: the 16 x 32 texels y[T, ldy] through a softplus) in THREE launches: the pass over the radiance transfer
 * under `light` (matpbr_env_phase_step's first kernel), one workgroup that folds its partial sums, commits SaveBest / EarlyStopping,
 * snapshots the best envmap (`best_env`; `first` != 0: unconditionally), back-propagates d_light through the SH projection and the softplus
 * and applies Adam (hyper[0] = lr, hyper[1] = step count, both in device memory), and the NEXT iteration's `env` = softplus(y) and
 * `light` = proj @ env.  = matpbr_env_phase_step + matpbr_select_improved + matpbr_env_project_bwd + Adam (matpbr_mlp.h: the AdamW step with weight_decay 0) + matpbr_env_project,
 * the same bits.  The caller runs matpbr_env_project once before the first iteration; T <= 1024 texels, one image. */
int matpbr_env_texel_phase_step(const float* T, const float* gt_srgb, float* pred, float* d_light, float* stats, float* history, int hist_len,
                                int es_patience, float es_min_delta, void* workspace, size_t workspace_bytes, int H, int W, float* y, int ldy,
                                const float* proj, float* env, float* best_env, float* light, float* g, float* adam_m, float* adam_v, float* hyper,
                                float beta1, float beta2, float eps, int n_texels, int first, void* stream);
/* The same two launches for the reference's parameterisation of the light (envhead.EnvMlpPhase: y[T, ldy] is the envmap MLP's output, `env` =
 * softplus(y) and `light` = proj @ env formed by matpbr_env_project before the call): the pass over the transfer, then one workgroup that folds,
 * commits SaveBest / EarlyStopping, snapshots the best envmap and back-propagates d_light through the projection and the softplus into d_y[T, ldy]
 * (rows padded with zeros), from where the caller's backward chain through the MLP starts.  = matpbr_env_phase_step + matpbr_select_improved +
 * matpbr_env_project_bwd, the same bits; d_y's row stride is ldy. */
int matpbr_env_mlp_phase_step(const float* T, const float* light, const float* gt_srgb, float* pred, float* d_light, float* stats, float* history,
                              int hist_len, int es_patience, float es_min_delta, void* workspace, size_t workspace_bytes, int H, int W, const float* y,
                              int ldy, const float* proj, const float* env, float* best_env, float* d_y, int n_texels, int first, void* stream);
/* The envmap head of hot loop A without a framework in between (inverse_img_w_mi.py:117-124,238-254): the 16x32 envmap is
 * softplus(envmap_net(start_envmap)) (mymodels/mlps.py:230-232) and the kernels integrate its SH projection.
 *   matpbr_env_project      env[T,3] = softplus(y[T, ldy]), light[25,3] = proj[25,T] env      (T <= 1024 texels)
 *   matpbr_env_project_bwd  d_y[T, ldg] = sigmoid(y) * (proj^T d_light) in columns 0..2, zero in the padding columns
 *   matpbr_select_improved  dst = src (n floats) when stats[8] (improved) is set or `first`: SaveBest's envmap snapshot (:247)
 * (the optimiser steps with their hyper-parameters in device memory live in matpbr_mlp.h with the network's other entry points: its AdamW
 * step with weight_decay 0 is torch.optim.Adam) */
int matpbr_env_project(const float* y, int ldy, const float* proj, float* env, float* light, int n_texels, void* stream);
int matpbr_env_project_bwd(const float* y, int ldy, const float* proj, const float* d_light, float* d_y, int ldg, int n_texels,
                           void* stream);
/* Other light parameterisations -> the [B,25,3] SH coefficients that the shading entry points take (MATPBR_LIGHT_SH25), and the
 * gradient back.  MATPBR_LIGHT_SH9: light [B,9,3], bands 0..2.  MATPBR_LIGHT_ENV_TEXELS: light [B, He*2He, 3], the equirectangular
 * texel map that is `emitter.data` of the reference scene (inverse_img_w_mi.py:63,217-219; 16 x 32), projected by midpoint quadrature
 * with the direction <-> texel mapping of myutils/envmap_utils.py:29-36 (= materialist_amd.sh.envmap_to_sh_matrix). */
int matpbr_light_to_sh25(const float* light, int light_kind, int n_light, float* sh25, int batch, void* stream);
int matpbr_light_to_sh25_bwd(const float* d_sh25, int light_kind, int n_light, float* d_light, int batch, void* stream);
int matpbr_select_improved(float* dst, const float* src, const float* stats, int first, long n, void* stream);
/* `--use_mask` (inverse_img_w_mi.py:379-381,509-511: `mat['roughness'][mask] = mat['roughness'][mask].mean()`, same for metallic) on one
 * [n] map per image (batch images, n = H*W), deterministic (fixed-order sums, one workgroup per image).
 *   forward  (gate == NULL): out[i] = mask[i] ? mean over the mask of clamp(in[j], lo, hi) : in[i]   (out may alias in);
 *   backward (gate != NULL): out[i] = mask[i] ? (lo <= gate[i] <= hi ? mean over the mask of in[j] : 0) : in[i], with in = d loss / d out of
 *                            the forward and gate = the forward's input: the gradient of every masked entry is the mean of the masked
 *                            gradients, through the clamp of its own input.
 * mask: one byte per pixel (0 / non-zero).  A mask without pixels leaves the map as it is.  Entries outside the mask are copied RAW in both
 * forms (lo / hi apply to the masked entries only: the consumers clamp again, as the reference's loop does, :372-381); out == in is supported
 * in both forms. */
int matpbr_masked_mean_fill(const float* in, const unsigned char* mask, const float* gate, float lo, float hi, float* out, long n, int batch,
                            void* stream);
/* Forward-only relighting (render_final.py:148-203 `render_w_mi`, :290-418 `rotate_envmap` / `render_rolling_envmap`).
 * The render is linear in the light, R = sum_k light[k] * T[k]:
 *   matpbr_shade_transfer  per-pixel transfer (d render / d light, both lobes) of the current materials into T (matpbr_transfer_bytes(); 300 B/pixel, tiled
 *                          [B][ceil(H*W/256)][75][256]: 75 = 25 coefficients x rgb; opaque to the caller), computed once
 *   matpbr_relight         out[F,H,W,3] for F lights [F,25,3] against one image's T (HBM-bound; up to 24 lights share one pass over T: hand it
 *                          the frames in batches of 24 -- 300/24 + 12 B/pixel per frame instead of 312) */
size_t matpbr_transfer_bytes(int H, int W, int batch);
int matpbr_shade_transfer(const float* a, const float* r, const float* m, const float* n, float* T, int H, int W, int batch,
                          int spp, const MatpbrCamera* cam, uint32_t flags, void* stream);
int matpbr_relight(const float* T, const float* lights, float* out_rgb, int H, int W, int n_frames, void* stream);

/* Plugin face, N independent lanes, AoS [N,3] vectors (the reference traces these over Dr.Jit arrays).
 *   matpbr_eval_brdf   = MatDiffBSDF.eval_pdf / eval_brdf     myutils/mi_plugin.py:1372-1427,1449-1460
 *                        (wi = light direction, wo = view direction; f already includes cos)
 *   matpbr_sample_brdf = MatDiffBSDF.sample / sample_brdf     myutils/mi_plugin.py:1296-1341,1429-1446
 *                        sample1[N] > 0.5 -> diffuse lobe; sample2[N,2]; weight = f/(pdf+1e-6) */
int matpbr_eval_brdf(const float* wi, const float* wo, const float* n, const float* a, const float* r,
                     const float* m, float* f, float* pdf, long N, void* stream);
int matpbr_eval_brdf_bwd(const float* wi, const float* wo, const float* n, const float* a, const float* r,
                         const float* m, const float* g /*[N,3] upstream*/, float* d_a, float* d_r, float* d_m,
                         float* d_n, long N, void* stream);
/* a1-a3 over N lanes -> out[N,4] = { D_GGX(cos1, r), G1_GGX_Schlick(cos1, r), G_Smith(cos1, cos2, r), fresnelSchlick(cos1, f0) }
 * (myutils/mi_plugin.py:89-97, 60-68, 70-76, 78-81). */
int matpbr_brdf_terms(const float* cos1, const float* cos2, const float* r, const float* f0, float* out, long N, void* stream);
int matpbr_sample_brdf(const float* sample1, const float* sample2, const float* wo, const float* n, const float* a,
                       const float* r, const float* m, float* wi, float* pdf, float* weight, long N, void* stream);
/* Attached sampling (a5): d/dr of what matpbr_sample_brdf returns, THROUGH the sampled direction and the pdf it divides by, as the
 * live reference differentiates it (myutils/mi_plugin.py:227-230 `mi_specular_sampler` is differentiable in the roughness, :1335-1341
 * the weight divides by an attached pdf).  d_wi[N,3], d_pdf[N], d_weight[N,3]; forward-mode derivatives lane by lane, pinned to the
 * reference's own autograd (tests/golden/sample_brdf_grad.npz).  The image kernels use the detached convention. */
int matpbr_sample_brdf_dr(const float* sample1, const float* sample2, const float* wo, const float* n, const float* a, const float* r,
                          const float* m, float* d_wi, float* d_pdf, float* d_weight, long N, void* stream);

/* Radiance of the SH light in N directions: L[N,3] = sum_k coef[k,:] Y_k(w)
 * (myutils/computeSH.py:165-224 `projection`). */
int matpbr_sh_eval(const float* w, const float* coef /*[25,3]*/, float* L, long N, void* stream);

/* Scene preparation (replaces the depth->mesh->face-normal route of load_estimated_mesh,
 * inverse_img_w_mi.py:30-56,721-727; myutils/mesh_recon.py:17-25,41-74): per-pixel geometric normal
 * of the depth heightfield, out_n[B,H,W,3]. */
int matpbr_normals_from_depth(const float* depth, float* out_n, int H, int W, int batch, const MatpbrCamera* cam,
                              void* stream);

/* HOST function (no GPU involved; host pointers): the reference's depth -> mesh conversion `depth_file_to_mesh` -> `detect_boundary_points`
 * (myutils/mesh_recon.py:41-74,86-331, called with minAngle 6 at inverse_img_w_mi.py:726) with its gap closing at depth discontinuities, and
 * the rotation into the renderer's frame (inverse_img_w_mi.py:727): the same sequential algorithm, vertex for vertex and triangle for
 * triangle (tests/golden/mesh_normals.npz).  depth[H,W] is the array handed to the mesher (2 max - prediction, 0 = no geometry).
 *   new_depth[H,W]          depth after the boundary pixels were pushed back (pass 2)
 *   vertices[2 H W x 3]     doubles: the H W grid vertices (row-major), then the duplicates; *n_vertices of them are valid
 *   triangles[2 (H-1)(W-1) x 3], *n_triangles
 *   normals[H W x 3]        (nullable) area-weighted normal of every grid vertex, towards the camera: the per-pixel geometric normal the
 *                           kernels shade with; zero where the vertex has no triangle (the camera ray sees the environment) */
int matpbr_depth_to_mesh_host(const float* depth, int H, int W, float fov_x_deg, float min_angle_deg, float* new_depth, double* vertices,
                              int* n_vertices, int* triangles, int* n_triangles, float* normals);

#ifdef __cplusplus
}
#endif
/* row f2 (the coordinate MLP's layer products, heads, optimiser steps): a header of its own, the same library */
#include "matpbr_mlp.h"
#endif /* MATPBR_H */
