/* matpbr_mlp.h -- the PosMLP side of the C ABI (mymodels/mlps.py:129-251, the coordinate network of
 * `--model_name pos_mlp`, inverse_img_w_mi.py:114-124,159-172,470-590), in a header of its own since round 6: include/matpbr.h keeps the
 * shading hot path (rows a1-a13, b), this file the layer products, heads and optimiser steps of the network that feeds it.  Same conventions
 * (device pointers, row-major fp32, hipStream_t as void*, int error codes, nothing allocated or synchronised inside an entry point);
 * included by matpbr.h, exported by the same libmatpbr.so. */
#ifndef MATPBR_MLP_H
#define MATPBR_MLP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

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
 *   matpbr_adamw_step_dev        torch.optim.AdamW on one flat buffer (weight_decay 0: torch.optim.Adam) with hyper[0] = lr and hyper[1] = steps done so far
 *                                in DEVICE memory (the count is advanced by the call): usable inside a captured hipGraph, lr changed by writing hyper[0] (:470) */
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
 *                                  written here), th [M, 8], the maps (each nullable) as matpbr_mlp_arm_head_fwd.  M a multiple of 128; n_head = 5 (the 'arm'
 *                                  head: anything else is MATPBR_ERR_INVALID_ARG) */
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

#ifdef __cplusplus
}
#endif
#endif /* MATPBR_MLP_H */
