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

#define MATPBR_VERSION 300 /* 0.3.0 */
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
                                        matpbr_brdf_phase_resolve before it reads pa / pr / pm / best_* / best_img / pred or starts the next part */
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
 * down (no snapshot in matpbr_brdf_loss_bwd_jac) and matpbr_adamw_step_snapshot_dev(..., stats) rests too, so a caller whose iteration contains
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
                                             of r / m (d_a not requested) or part 'a' (d_r, d_m not requested) runs its iteration as TWO launches: the
                                             statistics pass and one persistent streaming launch (at most 1024 workgroups, each streaming up to four
                                             512-pixel blocks from two register sets and walking the samples of the pixels it listed at its end: no
                                             MATPBR_STAGE_RESAMPLE launch); 144 ('rm') / 132 ('a') instead of 172 / 160 B/pixel.  The step with t == 1
                                             derives the folded planes from the generic ones; the generic ones stay current (re-sampled pixels rewrite
                                             both).  Other parts, and MATPBR_FLAG_GENERIC_STEP, run the generic step. */
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
/* MATPBR_FLAG_ROTATE_BEST: after the step with t = t_done, put everything where the copying form leaves it: current parameters in pa / pr / pm,
 * SaveBest's snapshots in best_a / best_r / best_m (maps of the part) and best_img (= the tone-mapped render of the best iteration of this
 * phase; untouched if none improved), the render of the CURRENT parameters (the next iteration's) in `pred`, the best iteration's in
 * `pred_next`.  The phase may go on afterwards (steps t_done + 1, ...).  Without the flag, or with t_done = 0: nothing to do. */
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
 * `light` = proj @ env.  = matpbr_env_phase_step + matpbr_select_improved + matpbr_env_project_bwd + matpbr_adam_step_dev + matpbr_env_project,
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
 *   matpbr_adam_step_dev    torch.optim.Adam on one flat buffer with hyper[0] = lr and hyper[1] = steps done so far in DEVICE memory
 *                           (the count is advanced by the call): usable inside a captured hipGraph, lr changed by writing hyper[0]
 *   matpbr_mlp_layer_bwd_input_w   matpbr_mlp_layer_bwd_input taking the layer's forward weight w[n_red, ldw] (no transposed copy);
 *                           point sets of at most 1024 rows (MATPBR_ERR_UNSUPPORTED beyond) */
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
int matpbr_adam_step_dev(float* p, const float* g, float* m, float* v, long n, float* hyper, float beta1, float beta2, float eps,
                         void* stream);
/* One backward step of a small network (M <= 1024 points: the 16 x 32 envmap MLP, mymodels/mlps.py:216-236 under autograd) in ONE launch,
 * given g = dL/d pre of layer l [M, n_red]:
 *   d_w[n_red, K]        = g^T x                               (weight gradient of layer l; x[M, K] its input)
 *   g_prev[M, n_prev]    = (g w) * c_prev, w[n_red, ldw] the layer's FORWARD weight (w == NULL: no input gradient, the first layer);
 *   colsum_out           = per-row-tile column sums of g_prev ([ceil(M/32)][256] floats: the next step's `colsum_in`, stride 256)
 *   d_bias[n_red]        = sum over `groups_in` rows of colsum_in (row stride colsum_stride): the bias gradient of layer l from the column
 *                          sums the step before left (for the output layer: colsum_in = g itself, stride ldg, groups_in = M); NULL: skipped.
 * colsum_out and colsum_in must be different buffers.  Deterministic. */
int matpbr_mlp_small_bwd_step(const float* g, int ldg, const float* w, int ldw, const float* c_prev, float* g_prev, int ldo, float* colsum_out,
                              int n_prev, const float* x, int ldx, float* d_w, int ldw_out, int K, const float* colsum_in, int colsum_stride,
                              int groups_in, float* d_bias, long M, int n_red, void* stream);
int matpbr_mlp_layer_bwd_input_w(const float* g, int ldg, const float* w, int ldw, const float* c_prev, float* g_prev, int ldo,
                                 float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red,
                                 void* stream);

/* Column sums of a row-major [M, N] fp32 matrix -> out[N]: the bias gradient of the PosMLP layers over M = H*W points
 * (mymodels/mlps.py:102-103 under autograd).  Deterministic two-pass; workspace of matpbr_column_sum_workspace_bytes(N). */
int matpbr_sin_bwd(const float* d_y, long ld_d, const float* pre, long ld_p, float* out, long M, int n, void* stream);
    /* out[M,n] (contiguous) = d_y[M,n (row stride ld_d)] * cos(pre[M,n (row stride ld_p)]): backward of the PosMLP sine layers */
size_t matpbr_column_sum_workspace_bytes(int N);
int matpbr_column_sum(const float* x, float* out, long M, int N, void* workspace, size_t workspace_bytes, void* stream);

/* Sine layers of the PosMLP (mymodels/mlps.py:102-103 `sin(linear(x))`, layer loop :216-229) and their backward, on the exact-f32
 * MFMA with the element-wise work in the GEMM epilogues (materialist_amd/csrc/posmlp_kernels.hip).  Row-major fp32; every
 * leading dimension is a multiple of 4 floats and every base pointer 16-byte aligned; N, K <= 256; M = H*W points.
 *   matpbr_mlp_layer_fwd         s_out = sin(x w^T + bias), c_out = cos(same) [both M x N, row stride ldo]; c_out == NULL: s_out = x w^T + bias
 *                                x [M, K] (stride ldx), w [N, K] (stride ldw) = the layer's `linear.weight`
 *   matpbr_mlp_layer_bwd_input   g_prev[M, n_prev] = (g wt^T) * c_prev;  g [M, n_red] (stride ldg) = dL/d pre of this layer,
 *                                wt [n_prev, n_red] (stride ldwt) = weight^T restricted to the inputs that come from the layer below,
 *                                c_prev = that layer's c_out (stride ldo, as g_prev); d_bias_prev[n_prev] (optional) = column sums of g_prev
 *   matpbr_mlp_layer_bwd_weight  d_w[N, K] (stride ldw) = g^T x;  g [M, N] (stride ldg), x [M, K] (stride ldx); deterministic (slab partials)
 *   matpbr_mlp_sincos / _mul     the same epilogues as stand-alone passes for the layers whose product stays in the BLAS
 * Output columns N .. min(ldo, 128 ceil(N/128)) of s_out / c_out / g_prev are scratch: the kernels may overwrite them (a skip
 * layer's x0 tail is therefore copied in after the call).  Pre-activations up to |x| ~ 1e5 keep sin/cos at 1.5 ulp. */
int matpbr_mlp_layer_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* s_out, float* c_out, int ldo, long M,
                         int N, int K, void* stream);
size_t matpbr_mlp_bwd_input_workspace_bytes(long M);
int matpbr_mlp_layer_bwd_input(const float* g, int ldg, const float* wt, int ldwt, const float* c_prev, float* g_prev, int ldo,
                               float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, void* stream);
size_t matpbr_mlp_bwd_weight_workspace_bytes(long M);
int matpbr_mlp_layer_bwd_weight(const float* g, int ldg, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                size_t workspace_bytes, long M, int N, int K, void* stream);
int matpbr_mlp_sincos(const float* pre, long ldp, float* s_out, long lds, float* c_out, long ldc, long M, int n, void* stream);
int matpbr_mlp_mul(const float* a, long lda, const float* b, long ldb, float* out, long ldo, long M, int n, void* stream);

/* The same sine-layer products on the bf16 matrix pipe with SPLIT OPERANDS (csrc/posmlp_kernels.hip, "bx" kernels): an f32 number is
 * the exact sum of three bf16 numbers and a product of two bf16 numbers is exact in f32, so x w^T = sum_ij x_i w_j^T with f32
 * accumulation; nprod = 9 keeps every partial product (the f32 product, exactly), nprod = 6 drops those below 2^-24 |x||w| (one f32
 * rounding).  The layer then runs at its HBM traffic instead of the f32-MFMA rate.
 *   matpbr_mlp_split_weights   w[N, K] (row stride ldw) -> wsplit (matpbr_mlp_wsplit_bytes(K); opaque), once per weight state; for the
 *                              backward product pass the transposed weight wt[n_prev, n_red] of matpbr_mlp_layer_bwd_input
 *   matpbr_mlp_layer_fwd_bx / _bwd_input_bx   as matpbr_mlp_layer_fwd / _bwd_input (sine layers: c_out required); M a multiple of 128,
 *                              256-wide output buffers (ldo >= 256), x / g readable up to the next multiple of 32 columns
 *                              (MATPBR_ERR_UNSUPPORTED otherwise: use the f32 entry points) */
size_t matpbr_mlp_wsplit_bytes(int K);
int matpbr_mlp_split_weights(const float* w, int ldw, int N, int K, void* wsplit, void* stream);
int matpbr_mlp_layer_fwd_bx(const float* x, int ldx, const void* wsplit, const float* bias, float* s_out, float* c_out, int ldo, long M,
                            int N, int K, int nprod, void* stream);
/* The backward pass INTO the first layer of the network on the split-operand kernel, without materialising dL/d pre of that layer
 * (G0 = (g wt) * cos(pre0), [M, n0]): its only consumers are formed in the epilogue --
 *   d_w0[k * ld_j + n * ld_c] = sum_m G0[m][n] x0[m][k]   (k < d0 <= 16: the network's input rows x0[M, ldx0 >= 16], zero beyond d0)
 *   d_bias0[n]                = sum_m G0[m][n]
 * = matpbr_mlp_layer_bwd_input_bx[_sgn] + matpbr_mlp_skinny_bwd_weight without the 268 MB store and re-read of G0 (512 x 512).
 * c_prev[M, ldc]: cos(pre0), or with sgn != 0 the sign-carrying sines of the first layer.  workspace: matpbr_mlp_bwd_input_workspace_bytes
 * (only with d_bias0); workspace2: matpbr_mlp_skinny_workspace_bytes(16).  M a multiple of 128.  Deterministic. */
int matpbr_mlp_first_layer_bwd_bx(const float* g, int ldg, const void* wtsplit, const float* c_prev, int ldc, int sgn, const float* x0, int ldx0,
                                  float* d_w0, long ld_j, long ld_c, int d0, float* d_bias0, void* workspace, size_t workspace_bytes, void* workspace2,
                                  size_t workspace2_bytes, long M, int n0, int n_red, int nprod, void* stream);
int matpbr_mlp_layer_bwd_input_bx(const float* g, int ldg, const void* wtsplit, const float* c_prev, float* g_prev, int ldo,
                                  float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red,
                                  int nprod, void* stream);
/* as matpbr_mlp_layer_bwd_weight with split operands (nprod 6 or 9): M a multiple of 16, ldg and ldx >= 256 (all 256 columns of
 * both operands are read; those at or beyond N / K may hold anything finite or not and are dropped). */
/* matpbr_mlp_layer_fwd / _fwd_bx for a skip layer (N < 256 outputs in a 256-wide buffer whose columns N.. hold x0, mymodels/mlps.py
 * :214-217): `tail` [M, ldt >= 256 - N] = those x0 values.  The layer kernel then stores whole 16-byte words (also over the tail) and
 * a second small launch rewrites the tail: guarding the one straddling word of every row inside the epilogue costs ~30 us per
 * layer at 512 x 512, the rewrite ~8.  The cosine buffer's tail is scratch.  tail == NULL: the columns N.. are left untouched. */
int matpbr_mlp_layer_fwd_tail(const float* x, int ldx, const float* w, int ldw, const float* bias, float* s_out, float* c_out, int ldo,
                              const float* tail, int ldt, long M, int N, int K, void* stream);
int matpbr_mlp_layer_fwd_bx_tail(const float* x, int ldx, const void* wsplit, const float* bias, float* s_out, float* c_out, int ldo,
                                 const float* tail, int ldt, long M, int N, int K, int nprod, void* stream);
/* matpbr_mlp_layer_fwd_bx of the LAST sine layer (N = 256 outputs) that also finishes the network: its epilogue forms the five
 * outputs of the output layer w_out[5, ldw_out >= 256], bias_out[5] for the rows it holds and runs the 'arm' head on them
 * (= matpbr_mlp_arm_head_fwd on s_out, without the pass over s_out). */
int matpbr_mlp_layer_fwd_bx_head(const float* x, int ldx, const void* wsplit, const float* bias, float* s_out, float* c_out, int ldo,
                                 const float* w_out, int ldw_out, const float* bias_out, const float* start, int lds, float* th,
                                 float* map_a, float* map_r, float* map_m, long M, int K, int nprod, void* stream);
/* ONE float per sine activation.  sin and cos of a pre-activation lie on the unit circle: the forward pass can store the sine with the SIGN of
 * the cosine in its last mantissa bit (the stored value moves by at most one ulp) and no cosines at all (a third of a 256-wide layer's
 * traffic); the backward pass rebuilds cos = sign * sqrt(1 - sin^2) where it multiplies by it.  The products stay f32-accurate; the cosine
 * factor of the backward pass carries |error| ~ 2e-7 / |cos| (rms relative error of a layer ~ 1e-5).
 *   matpbr_mlp_layer_fwd_bx / _bx_tail / _bx_head with c_out == NULL   write such sines
 *   matpbr_mlp_layer_fwd_sgn          the same for the thin first layer (K <= 16, image size), as matpbr_mlp_layer_fwd_tail
 *   matpbr_mlp_layer_bwd_input_bx_sgn / matpbr_mlp_layer_bwd_input_sgn   as matpbr_mlp_layer_bwd_input_bx / _bwd_input (n_red <= 16, image size) with
 *                                     `s_prev` = those sines of the layer below in place of its cosines */
int matpbr_mlp_layer_fwd_sgn(const float* x, int ldx, const float* w, int ldw, const float* bias, float* s_out, int ldo, const float* tail, int ldt,
                             long M, int N, int K, void* stream);
int matpbr_mlp_layer_bwd_input_sgn(const float* g, int ldg, const float* wt, int ldwt, const float* s_prev, float* g_prev, int ldo,
                                   float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, void* stream);
int matpbr_mlp_layer_bwd_input_bx_sgn(const float* g, int ldg, const void* wtsplit, const float* s_prev, float* g_prev, int ldo,
                                      float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red,
                                      int nprod, void* stream);
int matpbr_mlp_layer_bwd_weight_bx(const float* g, int ldg, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                   size_t workspace_bytes, long M, int N, int K, int nprod, void* stream);

/* The skinny ends of the coordinate MLP at image size, one streaming pass over the 256-wide matrix each (mymodels/mlps.py:219-236,
 * inverse_img_w_mi.py:493-496):
 *   matpbr_mlp_split_weights_t   matpbr_mlp_split_weights of the TRANSPOSE of w[K, ldw >= N] (element (n, k) = w[k * ldw + n]): the
 *                                forward weight as the operand of matpbr_mlp_layer_bwd_input_bx, no transposed copy
 *   matpbr_mlp_skinny_fwd        out[M, ldo][:, :J] = x[:, :K] w[J, :K]^T + bias, J in {3, 5, 8}, K a multiple of 4 (the zero-
 *                                initialised output layer)
 *   matpbr_mlp_arm_head_fwd      the same product for J = 5 followed by the 'arm' head: th[M,8] = tanh(.), u = 1.3 th + start[:, :5],
 *                                y = (clamp(u, 0, 1) + u) - u as rounded in fp32 (the straight-through clamp); map_a[M,3] = y[:, 0:3], map_r[M] = 0.93 y[:, 3] + 0.07, map_m[M] = y[:, 4]
 *                                (each map nullable: a map that the running part does not optimise keeps its fixed values)
 *   matpbr_mlp_arm_head_bwd      d_x[M,8] = d maps chained through the head (straight-through clamp, tanh'), zero where a
 *                                gradient pointer is null and in the padding columns 5..7
 *   matpbr_mlp_skinny_bwd_weight d_w[j * ld_j + c * ld_c] = sum_m s[m][j] b[m][c]  (j < J <= 16 columns of the skinny s whose rows
 *                                are padded to a multiple of 8 floats, c < C <= 256 columns of b[M, ldb >= 256]) and, when
 *                                d_bias is not null, d_bias[j] = sum_m s[m][j].  With (s, b) = (d_x, last hidden layer) this is
 *                                the output layer's gradient (ld_j = K, ld_c = 1); with (x0, d pre of the first layer) it is
 *                                the first layer's, stored transposed (ld_j = 1, ld_c = row stride of d_w).
 *   matpbr_adamw_step_dev        matpbr_adam_step_dev with torch.optim.AdamW's decoupled weight decay (:470) */
int matpbr_mlp_split_weights_t(const float* w, int ldw, int N, int K, void* wsplit, void* stream);
/* The FORWARD sine layers on two f16 pieces per operand (round 5; nprod = 3 of matpbr_mlp_layer_fwd_bx[_tail|_head]; mymodels/mlps.py:102-103,
 * :216-224).  Two round-to-nearest f16 pieces carry an f32 number to 2^-24 of its size (a rounded piece leaves a signed remainder), so three
 * f16 products p1 q1 + p1 q2 + p2 q1 with f32 accumulation are an f32-accurate product at half the matrix time of nprod = 6.  f16 has no
 * exponent range to spare: the weights are cut as 256 w (|w| < 255; the kernel scales the sums back) and the rows x must satisfy |x| <= 65504
 * (sines, coordinates and colours here); numbers below 2^-14 are carried to an absolute 3e-8.  The input-gradient and weight-gradient
 * products (loss gradients of any magnitude) stay on three bf16 pieces.
 *   matpbr_mlp_split_weights_fmt   matpbr_mlp_split_weights with flags = MATPBR_WSPLIT_TRANSPOSED | MATPBR_WSPLIT_F16X2; an F16X2 image is
 *                                  the operand of nprod = 3 ONLY (and a bf16 image of nprod 6 / 9 only); same buffer size */
#define MATPBR_WSPLIT_TRANSPOSED 1
#define MATPBR_WSPLIT_F16X2 2
int matpbr_mlp_split_weights_fmt(const float* w, int ldw, int N, int K, int flags, void* wsplit, void* stream);
/* The whole forward pass of the 'arm' coordinate MLP in ONE launch (round 5; csrc/posmlp_chain.hip; mymodels/mlps.py:211-236 with the skip
 * concatenations of :214-217 and the tanh head of :232-234, the maps of inverse_img_w_mi.py:493-496).  The products are formed transposed
 * (weights = A operand, rows = B operand), so a lane's sines of one layer are its share of the next layer's operand: between two layers the
 * activations stay in registers; every layer's sign-carrying sines are still WRITTEN once (the backward pass reads them), none is read.
 * Arithmetic as matpbr_mlp_layer_fwd_bx with nprod = 3 (two f16 pieces of 256 w and of the sines, three products, f32 accumulation), the
 * first layer (K = d0 <= 16) on the exact-f32 matrix instruction; = the layer-by-layer kernels to f32 rounding (the k order differs).
 *   matpbr_mlp_chain_images_bytes  size of the `images` buffer
 *   matpbr_mlp_chain_prep          once per weight state: w[0] [n[0], ldw >= d0], w[1..3] [n[l], ldw >= 256], w[4] [n[4] <= 8, ldw >= 256] and the
 *                                  five bias vectors -> images.  n[l] of a sine layer is 256, or 241 = 256 - 15 for a layer whose buffer ends in x0.
 *                                  In the same launch, what else an iteration prepares once per optimiser step (both nullable): bwd_images[0..2] =
 *                                  the MATPBR_WSPLIT_F16X2 | MATPBR_WSPLIT_TRANSPOSED images of (w[l][:, :n[l-1]])^T, l = 1..3, the operands of
 *                                  matpbr_mlp_layer_bwd_input_blk (= matpbr_mlp_split_weights_fmt, the same bits), and `zero_words` 32-bit zeros
 *                                  at `zero` (the gradient tiles' maxima, which their producers fill by atomic max)
 *   matpbr_mlp_chain_fwd           x0 [M, ldx0 >= 16] (zero beyond d0) -> s_out[0..3] [M, ldo >= 256] (columns n[l].. of a 241-wide layer = x0,
 *                                  written here), th [M, 8], the maps (each nullable) as matpbr_mlp_arm_head_fwd.  M a multiple of 128 */
size_t matpbr_mlp_chain_images_bytes(void);
int matpbr_mlp_chain_prep(const float* const* w, const int* ldw, const int* n, const float* const* bias, int d0, void* images, void* const* bwd_images,
                          void* zero, long zero_words, void* stream);
int matpbr_mlp_chain_fwd(const float* x0, int ldx0, const void* images, float* const* s_out, int ldo, const int* n, const float* start, int lds, float* th,
                         float* map_a, float* map_r, float* map_m, int n_head, long M, void* stream);
/* The BACKWARD products of the 256-wide layers on two f16 pieces (round 5; the autograd backward of mymodels/mlps.py:102-103, :216-224 as
 * driven by inverse_img_w_mi.py:493-547).  A loss gradient has no natural size, so the rows g travel in blocks: every 128-row tile (128
 * consecutive pixels) has ONE power-of-two exponent that brings its largest |g| to [2^13, 2^14), taken from `g_tile_max` -- [M / 128] f32
 * bit patterns of the tiles' largest |g|, which the kernel that PRODUCED g filled by atomic max into an array the caller zeroed (bit patterns
 * of magnitudes order as values: the result does not depend on the order of the adds).  Within a tile, elements down to 2^-16 of the largest
 * keep the 2^-24 relative accuracy of two pieces, smaller ones are carried to an absolute 2^-39 of it; against fp64 the products' error is
 * that of the three-bf16-piece form and of the exact-f32 kernels (tests/test_gpu_parity.py::test_block_scaled_f16_backward_products).
 * The weight operand is an MATPBR_WSPLIT_F16X2 image (of the transposed forward weight for the input gradient).
 *   matpbr_mlp_out_layer_bwd_tmax    matpbr_mlp_out_layer_bwd that also fills g_tile_max for the g_prev it writes
 *   matpbr_mlp_layer_bwd_input_blk   matpbr_mlp_layer_bwd_input_bx_sgn (sign-carrying sines below) on these pieces; out_tile_max (nullable):
 *                                    the tile maxima of the g_prev it writes, for the next product
 *   matpbr_mlp_first_layer_bwd_blk   matpbr_mlp_first_layer_bwd_bx likewise (sgn = 1)
 *   matpbr_mlp_layer_bwd_weight_blk  matpbr_mlp_layer_bwd_weight_bx likewise: x (sines, |x| <= 65504) as it is, g under one exponent per slab of
 *                                    rows (the largest of its tiles' maxima); M a multiple of 128
 * Every one of these ends in a small launch that folds per-workgroup partial sums (a weight gradient's 256 slabs, the column sums behind a bias
 * gradient) -- nine latency-bound launches per iteration whose results nothing but the optimiser reads.  `defer` (nullable: fold now) receives the
 * fold as a record instead (matpbr_mlp_first_layer_bwd_blk: two records; kind MATPBR_REDUCE_NONE where there is nothing to fold); the caller keeps
 * the workspaces of the deferred calls apart and untouched, and runs all records in ONE launch before its optimiser step:
 *   matpbr_mlp_reduce_jobs           at most 16 records; the same sums in the same order as the launches they replace (the same bits) */
#define MATPBR_REDUCE_NONE (-1)
#define MATPBR_REDUCE_WGRAD 0   /* src [groups][256 x 256] -> dst[n * n2 + k], n < n0, k < n1 */
#define MATPBR_REDUCE_COLSUM 1  /* src [groups][256] -> dst[c], c < n0 */
#define MATPBR_REDUCE_SKINNY 2  /* matpbr_mlp_skinny_bwd_weight's fold: src [groups][n0][256] -> dst[j ld_j + c ld_c] (j < n1, c < n2), src_b [groups][n0] -> dst_b[j],
                                   src_g [groups][256] -> dst_g[c] (c < n3; nullable) */
typedef struct MatpbrReduceJob {
    int kind, groups;
    const float *src, *src_b, *src_g;
    float *dst, *dst_b, *dst_g;
    int n0, n1, n2, n3;
    long ld_j, ld_c;
} MatpbrReduceJob;
int matpbr_mlp_reduce_jobs(const MatpbrReduceJob* jobs, int n_jobs, void* stream);
int matpbr_mlp_out_layer_bwd_tmax(const float* d_x, int ldd, const float* s_prev, const float* c_prev, int lds, const float* w_out, int ldw, float* g_prev,
                                  int ldg, void* g_tile_max, float* d_w, long ld_j, long ld_c, float* d_bias, float* d_bias_prev, void* workspace,
                                  size_t workspace_bytes, long M, int J, int n_prev, MatpbrReduceJob* defer, void* stream);
int matpbr_mlp_layer_bwd_input_blk(const float* g, int ldg, const void* g_tile_max, const void* wtsplit, const float* s_prev, float* g_prev, int ldo,
                                   void* out_tile_max, float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red,
                                   MatpbrReduceJob* defer, void* stream);
int matpbr_mlp_first_layer_bwd_blk(const float* g, int ldg, const void* g_tile_max, const void* wtsplit, const float* s_prev, int lds, const float* x0,
                                   int ldx0, float* d_w0, long ld_j, long ld_c, int d0, float* d_bias0, void* workspace, size_t workspace_bytes,
                                   void* workspace2, size_t workspace2_bytes, long M, int n0, int n_red, MatpbrReduceJob* defer2, void* stream);
int matpbr_mlp_layer_bwd_weight_blk(const float* g, int ldg, const void* g_tile_max, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                    size_t workspace_bytes, long M, int N, int K, MatpbrReduceJob* defer, void* stream);
/* up to 8 splits in one launch (host arrays of n_jobs entries; transposed[j] = the flags of matpbr_mlp_split_weights_fmt for job j:
 * 0 / 1 as before, + MATPBR_WSPLIT_F16X2 for the f16 form): the weights of every layer change together, once per optimiser step */
int matpbr_mlp_split_weights_multi(const float* const* w, const int* ldw, const int* N, const int* K, const int* transposed,
                                   void* const* wsplit, int n_jobs, void* stream);
int matpbr_mlp_skinny_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, long M, int J, int K,
                          void* stream);
int matpbr_mlp_arm_head_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* start, int lds, float* th,
                            float* map_a, float* map_r, float* map_m, long M, int K, void* stream);
int matpbr_mlp_arm_head_bwd(const float* g_a, const float* g_r, const float* g_m, const float* th, float* d_x, long M, void* stream);
size_t matpbr_mlp_skinny_workspace_bytes(int J);
/* The backward pass of the 'arm' network's OUTPUT layer (mymodels/mlps.py:233-236 under autograd) in one pass over the sines of the last
 * sine layer, given d_x[M, ldd >= 8] = dL/d(output pre-activations) (J <= 5 valid columns, matpbr_mlp_arm_head_bwd):
 *   d_w[j * ld_j + c * ld_c] = sum_m d_x[m][j] s_prev[m][c],  d_bias[j] = sum_m d_x[m][j]                  (the layer's own gradients)
 *   g_prev[m][n] = (sum_j d_x[m][j] w_out[j][n]) * cos(pre_prev[m][n]),  d_bias_prev[n] = sum_m g_prev[m][n], n < n_prev
 * with cos(pre_prev) = c_prev[M, lds] or, c_prev == NULL, rebuilt from the sign-carrying sines s_prev (matpbr_mlp_layer_fwd_sgn).
 * = matpbr_mlp_skinny_bwd_weight + matpbr_mlp_layer_bwd_input[_sgn] without reading the 256-wide matrix twice.  All matrices 256
 * columns wide in memory; workspace of matpbr_mlp_skinny_workspace_bytes(J); deterministic. */
int matpbr_mlp_out_layer_bwd(const float* d_x, int ldd, const float* s_prev, const float* c_prev, int lds, const float* w_out, int ldw, float* g_prev,
                             int ldg, float* d_w, long ld_j, long ld_c, float* d_bias, float* d_bias_prev, void* workspace, size_t workspace_bytes,
                             long M, int J, int n_prev, void* stream);
int matpbr_mlp_skinny_bwd_weight(const float* s, int lds, const float* b, int ldb, float* d_w, long ld_j, long ld_c, float* d_bias,
                                 void* workspace, size_t workspace_bytes, long M, int J, int C, void* stream);
int matpbr_adamw_step_dev(float* p, const float* g, float* m, float* v, long n, float* hyper, float beta1, float beta2, float eps,
                          float weight_decay, void* stream);
/* the same with SaveBest's weight snapshot in the same pass: best[i] = p[i] (the weights that produced this iteration's render)
 * when stats[8] (improved) is set, before p is updated (best nullable; best needs stats).  With `stats`, an image whose EarlyStopping
 * fired in an earlier iteration (stats[13] >= 2) rests: no update, no step count (the reference's loop has left by then, :250-254,548-555);
 * and the 1-based step of the bias corrections is the row's iteration counter stats[14] -- the caller's iteration commits its statistics
 * (matpbr_brdf_loss_stats / the phase steps) ONCE before this call and starts its optimiser with its statistics row -- hyper[1] is kept in step. */
int matpbr_adamw_step_snapshot_dev(float* p, const float* g, float* m, float* v, long n, float* hyper, float beta1, float beta2,
                                   float eps, float weight_decay, float* best, const float* stats, void* stream);

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
#endif /* MATPBR_H */
