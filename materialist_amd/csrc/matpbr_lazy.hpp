// matpbr_lazy.hpp -- hot loop B without walking the GGX samples of every pixel in every iteration (gfx950, wave64, fp32).
//
// In the parts of --opt_order that move the roughness (inverse_img_w_mi.py:371-386 / 493-515) light and shading normals are fixed
// (:317-342), so the specular sums S0, S1 of a pixel are functions of its roughness alone -- and Adam moves r by 1e-4 .. 3e-4 per
// step.  Each pixel keeps a LOCAL MODEL around the roughness r_ref its 20 samples were last walked at:
//     P, SD = S0 - S1, S1 at r_ref (fp32);  their slopes P', gSD, gS1 and A2 (P is a quadratic in r), the detached r-derivatives
//     dSD, dS1 of the backward convention AND THEIR slopes eSD, eS1 (the same one-sided difference: the samples at r + h are walked anyway),
//     so that d out / d r is first order in dr = r - r_ref like the render (fp16);  a validity interval [r_ref - lo, r_ref + hi].
// lazy_fwd_kernel renders out = a (1-m) P(r) + C0 SD(r) + S1(r) from the model -- a streaming kernel -- writes the half-precision
// jac planes of the backward pass, and lists the pixels that have left their interval; lazy_refresh_kernel walks the samples of the
// listed pixels only (1-2 % of the image per iteration on recorded runs), rebuilds their models and patches their render.
// oracle/matpbr_oracle.c (lazy_refresh_pixel / lazy_eval_pixel) is the specification: interval construction (step-size control for
// the smooth part, first-order prediction of the horizon / back-facing crossings of every sample for the kinks) is explained there.
// Gate (tests/test_gpu_lazy.py): |lazy - exact| <= 1e-3 max(|exact|, mean|exact|) on every pixel in every iteration.
#pragma once
#include "matpbr_shade.hpp"

namespace matpbr {

// ---- constants of the specification (oracle/matpbr_oracle.c LAZY_*) --------------------------------------------------------
constexpr float kLzH = 1e-3f, kLzRhoInit = 2e-3f, kLzRhoMin = 2.5e-4f, kLzRhoMax = 3e-2f, kLzTolS = 2.5e-4f, kLzTolK = 1.5e-4f;
constexpr float kLzKinkSafety = 0.8f, kLzMoved = 1e-4f;
// round 6: the radius also answers to the DERIVATIVE the model hands the backward pass.  At a refresh the old model's prediction of the detached
// derivative d out_c / d r at the new roughness (dSD + eSD dr, dS1 + eS1 dr) is compared with the walked one, relative to
// max(|d out_c / d r|, kLzJFloor x the parity floor), tolerance kLzTolJ -- the same controller, on the larger of the two normalised errors
constexpr float kLzTolJ = 5e-4f, kLzJFloor = 0.25f;
// ... and so do the kinks: beyond a crossing the model keeps extrapolating the crossing sample's share of the DERIVATIVE too -- its share of the sums
// times lam = d ln(weight)/dr -- which, relative to d out / d r, weighs ~50 times what its share of the value weighs relative to the render.  An
// interval ends where either costs its tolerance (kLzTolKJ of max(|d out_c / d r|, kLzJFloor x the parity floor) for the derivative).
constexpr float kLzTolKJ = 1e-3f;
__device__ __forceinline__ float lazy_rho_next(float rho, float adr, float e_s, float e_j, float tol_s, float tol_j) {
    const float ec = fmaxf(e_s * (1.0f / tol_s), e_j * (1.0f / tol_j));
    const float want = 0.9f * adr * rsq(fmaxf(ec, 1e-9f));
    return fminf(fmaxf(want, 0.5f * rho), 2.0f * rho);
}

// ---- state: 32-bit planes of B*P entries (a lane's two pixels are an 8-byte access, a wave's access is 512 contiguous bytes) ----
enum { kLzRref = 0, kLzLoHi = 1 /* half2 (lo, hi) */, kLzRho = 2, kLzP = 3, kLzSD = 6, kLzS1 = 9,
       kLzPk = 12 /* half2 (P', A2) */, kLzSk = 15 /* half2 (gSD, gS1) */, kLzDk = 18 /* half2 (dSD, dS1) */,
       kLzEk = 21 /* half2 (eSD, eS1): the slopes of dSD, dS1 in r (round 5: d out / d r to first order in r - r_ref) */,
       kLzD32 = 24 /* dSD_c at 24 + c, dS1_c at 27 + c in fp32: read by the NEXT refresh of the pixel only (the radius control compares the old model's
                      prediction of the derivative with the walked one: half-precision rounding of dSD, dS1 would drown the tolerance) -- no streaming kernel reads them */,
       kLzPlanes = 30 };
// jac16: 5 planes, half2 (P_c, SD_c) for c = 0..2, half2 (JR_0, JR_1), half2 (JR_2, 0)
constexpr int kJac16Planes = 5;
constexpr int kLazyBlockPixels = 2 * kBlock;

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) { return __builtin_bit_cast(uint32_t, h2{(_Float16)lo, (_Float16)hi}); }
__device__ __forceinline__ float h2_lo(uint32_t u) { return (float)__builtin_bit_cast(h2, u).x; }
__device__ __forceinline__ float h2_hi(uint32_t u) { return (float)__builtin_bit_cast(h2, u).y; }
__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// ---- folded models of the persistent step kernel (lazy_pstep_kernel) -------------------------------------------------------------
// A part of --opt_order moves some of the maps and leaves the others alone (inverse_img_w_mi.py:343-357); what it leaves alone folds into
// the pixel's model, and the step reads fewer planes:
//   kFoldXY (parts of r / m: the albedo is a constant of the part)   out_c = X_c(dr) + m Y_c(dr),   dr = r - r_ref,
//       X_c = a_c P_c + 0.04 SD_c + S1_c,   Y_c = (a_c - 0.04) SD_c - a_c P_c      (C0 = 0.04 (1 - m) + m a, :1412)
//       X_c(dr) = X0 + X1 dr + X2 dr^2,  Y_c(dr) = Y0 + Y1 dr - X2 dr^2  (X2 = a_c A2_c: P is an exact quadratic in r),
//       d out_c / d m = Y_c(dr);  d out_c / d r = JX_c + m JY_c with JX = JX0 + 2 X2 dr, JY = JY0 - 2 X2 dr: the stop-gradient convention
//       (JX0 = a dP + 0.04 dSD + dS1, JY0 = (a - 0.04) dSD - a dP); with attached sampling the models' slopes X1, Y1 take their place.
//       17 planes = 68 B/pixel (the generic model: 96 B/pixel + the 12 B/pixel of the albedo it is combined with).
//   kFoldGH (part 'a': roughness and metallic are constants of the part)   out_c = a_c G_c + H_c,   d out_c / d a_c = G_c,
//       G_c = (1 - m) P_c(dr) + m SD_c(dr),   H_c = 0.04 (1 - m) SD_c(dr) + S1_c(dr):  6 planes = 24 B/pixel, never re-sampled.
// The generic planes stay the specification (oracle/matpbr_oracle.c); the folded ones are derived from them by lazy_fold_kernel at the
// start of a part and rewritten together with them for every re-sampled pixel.
enum { kFoldNone = 0, kFoldXY = 1, kFoldGH = 2 };
enum { kFxRref = 0, kFxLoHi = 1 /* byte lo | byte hi (iv_pack) | half X2_0 */, kFxXY = 2 /* five words: xy_pack */, kFxS = 7 /* half2 (X1_c, Y1_c) */,
       kFxJ = 10 /* half2 (JA0_c, JY0_c) */, kFxQ = 13 /* half2 (X2_1, X2_2) */, kFxE = 14 /* half2 (JX1_c, JY1_c) at 14 + c */, kFxPlanes = 17 };
// X0_c, Y0_c: 24 bits each (sign, exponent, 15 mantissa bits, rounded to nearest: 2^-16 = 1.5e-5 of a render that is held to 1e-3), six of them in five
// words -- the top 24 bits of word k hold X0_0, Y0_0, X0_1, Y0_1, X0_2 in turn, the low bytes of words 0..2 the three bytes of Y0_2, the low byte
// of word 3 m_ref (see FoldXY), the low byte of word 4 is spare.  68 B/pixel with half-precision slopes of the derivative (round 5: 68 with e5m2).
__device__ __forceinline__ float xy_round(float x) { return as_f((as_u(x) + 0x80u) & 0xffffff00u); }
__device__ __forceinline__ void xy_pack(const float (&X0)[3], const float (&Y0)[3], uint32_t mcode, uint32_t (&w)[5]) {      // X0, Y0: xy_round'ed already
    const uint32_t y2 = as_u(Y0[2]);
    w[0] = as_u(X0[0]) | (y2 >> 24);
    w[1] = as_u(Y0[0]) | ((y2 >> 16) & 0xffu);
    w[2] = as_u(X0[1]) | ((y2 >> 8) & 0xffu);
    w[3] = as_u(Y0[1]) | (mcode & 0xffu);
    w[4] = as_u(X0[2]);
}
__device__ __forceinline__ void xy_unpack(const uint32_t (&w)[5], float (&X0)[3], float (&Y0)[3]) {
    X0[0] = as_f(w[0] & 0xffffff00u); Y0[0] = as_f(w[1] & 0xffffff00u); X0[1] = as_f(w[2] & 0xffffff00u); Y0[1] = as_f(w[3] & 0xffffff00u);
    X0[2] = as_f(w[4] & 0xffffff00u);
    Y0[2] = as_f((w[0] << 24) | ((w[1] & 0xffu) << 16) | ((w[2] & 0xffu) << 8));
}
// JX1, JY1: the slopes of the folded detached derivative (JX = JX0 + (2 X2 + JX1) dr, JY = JY0 + (JY1 - 2 X2) dr).  Round 5 carried them in eight
// bits (e5m2): a first-order term of 5 % of the derivative at the far end of an interval then comes with an error of up to 6e-3 of the derivative
// (tools/lazy_grad_diag.py), above the 1e-3 the gradients are held to -- half precision since round 6.  The interval's two lengths pay for four of
// the six bytes: eight bits each (iv_pack: four exponent and four mantissa bits, rounded DOWN -- an interval never widens; 3 % shorter on average).
constexpr int kIvBias = (127 - 21) << 4;        // code 0 = 2^-21 (4.8e-7: 1/200 of an Adam step; shorter intervals are carried as that), code 255 = 3.03e-2 >= kLzRhoMax
__device__ __forceinline__ uint32_t iv_pack(float x) {
    const int q = (int)(__builtin_bit_cast(uint32_t, x) >> 19) - kIvBias;
    return (uint32_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
}
__device__ __forceinline__ float iv_unpack(uint32_t code) { return __builtin_bit_cast(float, (code + (uint32_t)kIvBias) << 19); }
__device__ __forceinline__ float iv_round(float x) { return iv_unpack(iv_pack(x)); }      // the specification's interval lengths ARE these 256 values
__device__ __forceinline__ uint32_t pack_lohi_x2(float lo, float hi, float x2_0) {
    return iv_pack(lo) | (iv_pack(hi) << 8) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)x2_0) << 16);
}
enum { kFgG = 0, kFgH = 3, kFgPlanes = 6 };
// The slopes eSD, eS1 of the detached derivatives come from a one-sided difference over kLzH; a sample that crosses the horizon inside that
// stencil makes the difference a jump / h, not a slope.  Whatever they are, they may correct the derivative by at most half its size
// at the far end of the pixel's interval (|e| <= 0.5 max_c (|dSD_c| + |dS1_c|) / max(lo, hi)): the specification's LAZY_E_CAP.  (Round 5 stated the
// cap per channel: a channel whose dSD passes through zero -- a legitimate slope beside a small value -- lost it, 25 pixels of an image with
// derivative errors of 1-3 %: tools/lazy_grad_diag.py with DIAG_DUMP.)
__device__ __forceinline__ float lazy_e_cap(float e, float dmax, float lo, float hi) {      // dmax: the largest |dSD_c| + |dS1_c| of the pixel's channels
    const float lim = 0.5f * dmax / fmaxf(fmaxf(lo, hi), 1e-4f);
    return fminf(fmaxf(e, -lim), lim);
}
// kFxJ carries (JA0_c, JY0_c) with JA0 = JX0 + m_ref JY0, m_ref = the pixel's metallic when its model was built, in eight bits (code / 255: the low
// byte of xy_pack's fourth word): d out_c / d r = JA0 + (m - m_ref) JY0 + ..., so that the
// half-precision rounding of the two words is relative to the derivative itself, not to two terms that may cancel in JX0 + m JY0 (round 6).
struct FoldXY { float X0, Y0, X1, Y1, JX0 /* JA0 */, JY0, X2, JX1, JY1; };
__device__ __forceinline__ uint32_t mref_code(float m) { return (uint32_t)__builtin_rintf(fminf(fmaxf(m, 0.0f), 1.0f) * 255.0f); }
__device__ __forceinline__ float mref_of(uint32_t word) { return (float)(word & 0xffu) * (1.0f / 255.0f); }
__device__ __forceinline__ void fold_xy(float a, float P, float SD, float S1, float dP, float A2, float gSD, float gS1, float dSD, float dS1, float eSD, float eS1,
                                        uint32_t mcode, FoldXY& f) {
    const float am = a - 0.04f, naP = -(a * P), nadP = -(a * dP);
    f.X0 = xy_round(fmaf(a, P, fmaf(0.04f, SD, S1)));            // as the planes carry them (xy_pack): every writer renders from the stored values
    f.Y0 = xy_round(fmaf(am, SD, naP));
    f.X1 = fmaf(a, dP, fmaf(0.04f, gSD, gS1));
    f.Y1 = fmaf(am, gSD, nadP);
    f.JY0 = fmaf(am, dSD, nadP);
    f.JX0 = fmaf(mref_of(mcode), f.JY0, fmaf(a, dP, fmaf(0.04f, dSD, dS1)));
    f.X2 = a * A2;
    f.JX1 = fmaf(0.04f, eSD, eS1);
    f.JY1 = am * eSD;
}
constexpr int kTile = kBlock;          // pixels of a tile of lazy_pstep_kernel: one per thread
constexpr int kMaxTilesPerWg = 8;      // tiles a workgroup of lazy_pstep_kernel streams at most (its LDS lists are sized for them)
__host__ __device__ inline int lazy_tiles(long P) { return (int)((P + kTile - 1) / kTile); }
// [16 planes][walk sums: B * nblk int64].  The walk sums: the render of the pixels lazy_pwalk_kernel re-samples, summed per 512-pixel block in
// FIXED POINT (units of 2^-32) by integer atomics -- integer addition is associative, so the sum does not depend on which wave adds first
// (and is the same whatever the batch size); the statistics pass adds it to the block's floating-point sum of the streamed pixels.
// Behind them the WALK QUEUE of lazy_pstep_kernel / lazy_pwalk_kernel: per image kWalkShards flat lists of pixel indices, each with two
// counters (one per iteration parity: the step kernel of iteration t appends to the lists of parity t & 1 -- a wave reserves room for all its
// pixels with ONE atomic on the counter of its shard, when it has streamed its last tile -- and clears the other parity's counters).
constexpr double kWalkFix = 4294967296.0;
constexpr int kWalkShards = 32;
// The folded planes are stored TILE-major: the kFxPlanes words of 256 consecutive pixels (one tile of the persistent step) as kFxPlanes consecutive
// 1-KB rows -- a tile's model is one contiguous 17-KB read instead of seventeen 1-KB reads 8 MB apart (plane-major, through the first half of round 6).
// A plane's pointer is the address of its row in tile 0; fx_off(i) is the byte offset of pixel i (= b P + p) from there.
constexpr unsigned kFxTileBytes = (unsigned)kFxPlanes * 1024u;
inline size_t lazy_fold_planes_bytes(long P, int batch) { return (((size_t)batch * (size_t)P + 255) / 256) * (size_t)kFxTileBytes; }
__device__ __forceinline__ unsigned fx_off(unsigned i) { return (i >> 8) * kFxTileBytes + ((i & 255u) << 2); }
inline size_t fx_row_words(int k) { return (size_t)k * 256; }      // a plane's pointer: row k of tile 0 (in 32-bit words from the buffer's start)
// entries of one shard: the pixels of every wave that may append to it (wave v of the 4 gridDim.x of an image: shard v % kWalkShards; a wave
// owns 128 pixels of each of its workgroup's <= 4 blocks: <= 16 nblk + 576 pixels per shard whatever the launch geometry)
__host__ __device__ inline long walk_shard_cap(long P) { return (P + 2 * kBlock - 1) / (2 * kBlock) * (2 * kBlock / kWalkShards) + 1024; }
inline size_t lazy_fold_bytes(long P, int batch) {
    const size_t nblk = (size_t)((P + 2 * kBlock - 1) / (2 * kBlock));
    return lazy_fold_planes_bytes(P, batch) + (size_t)batch * nblk * sizeof(long long) + (size_t)batch * 2 * kWalkShards * sizeof(uint32_t) +
           (size_t)batch * kWalkShards * (size_t)walk_shard_cap(P) * sizeof(uint32_t);
}

__host__ __device__ inline int lazy_fwd_blocks(long P) { return (int)((P + kLazyBlockPixels - 1) / kLazyBlockPixels); }
// workgroups of the refresh kernel per image (each takes every lazy_groups()-th chunk of the image's work list; also the number of
// partial sums it contributes): enough to fill the GPU when a few per cent of a 512 x 512 image are listed, few enough to be cheap when none is
__host__ __device__ inline int lazy_groups(long P) { const int n = lazy_fwd_blocks(P) / 4; return n < 16 ? 16 : (n > 256 ? 256 : n); }
// [planes][counts: B * nblk u32][lists: B * nblk * 512 u16]
inline size_t lazy_planes_bytes(long P, int batch) { return (size_t)kLzPlanes * (size_t)batch * (size_t)P * 4; }
inline size_t lazy_counts_bytes(long P, int batch) { return (size_t)batch * (size_t)lazy_fwd_blocks(P) * 4; }
inline size_t lazy_lists_bytes(long P, int batch) { return (size_t)batch * (size_t)lazy_fwd_blocks(P) * kLazyBlockPixels * 2; }

// =================================================================================================
// streaming forward from the per-pixel models
// =================================================================================================
struct LazyFwdArgs {
    const float *a, *r, *m;
    const uint32_t* state;
    float* out;
    uint32_t* jac16;
    const float* stats;       // nullable: skip images whose EarlyStopping has fired
    float* block_sums;        // nullable [B][n_sums]: slot blockIdx.x = sum of the rgb this workgroup rendered (listed pixels excluded)
    uint32_t* counts;
    uint16_t* lists;
    int clamp, force, n_sums;
    int jac32;                // `jac16` holds NINE fp32 planes instead (P, SD, d out / d r: the layout of matpbr_shade_fwd_ex's jac)
};

__global__ __launch_bounds__(kBlock) void lazy_fwd_kernel(const LazyFwdArgs q, int P) {
    __shared__ float s_sum[4];
    __shared__ int s_cnt[4];
    const int b = blockIdx.y;
    if (q.stats && img_stopped(q.stats, b)) return;
    const long BP = (long)gridDim.y * P;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool act0 = q0 < P, two = q0 + 1 < P;
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    bool need0 = act0, need1 = two;
    float tot = 0.0f;
    if (!q.force) {
        f2 a[3], r = f2{q.r[i0], q.r[i1]}, m = f2{q.m[i0], q.m[i1]};
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = f2{q.a[i0 * 3 + c], q.a[i1 * 3 + c]};
        if (q.clamp) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a[c] = clamp2(a[c], 0.0f, 1.0f);
            r = clamp2(r, 0.07f, 1.0f);
            m = clamp2(m, 0.0f, 1.0f);
        }
        const uint32_t* __restrict__ S = q.state;
        const f2 dr = r - f2{as_f(S[kLzRref * BP + i0]), as_f(S[kLzRref * BP + i1])};
        const uint32_t lh0 = S[kLzLoHi * BP + i0], lh1 = S[kLzLoHi * BP + i1];
        need0 = act0 && !(dr.x >= -h2_lo(lh0) && dr.x <= h2_hi(lh0));
        need1 = two && !(dr.y >= -h2_lo(lh1) && dr.y <= h2_hi(lh1));
        const f2 omm = 1.0f - m;
        f2 JR[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f2 Pv = f2{as_f(S[(kLzP + c) * BP + i0]), as_f(S[(kLzP + c) * BP + i1])};
            const f2 SDv = f2{as_f(S[(kLzSD + c) * BP + i0]), as_f(S[(kLzSD + c) * BP + i1])};
            const f2 S1v = f2{as_f(S[(kLzS1 + c) * BP + i0]), as_f(S[(kLzS1 + c) * BP + i1])};
            const uint32_t pk0 = S[(kLzPk + c) * BP + i0], pk1 = S[(kLzPk + c) * BP + i1];
            const uint32_t sk0 = S[(kLzSk + c) * BP + i0], sk1 = S[(kLzSk + c) * BP + i1];
            const uint32_t dk0 = S[(kLzDk + c) * BP + i0], dk1 = S[(kLzDk + c) * BP + i1];
            const uint32_t ek0 = S[(kLzEk + c) * BP + i0], ek1 = S[(kLzEk + c) * BP + i1];
            const f2 dP0 = f2{h2_lo(pk0), h2_lo(pk1)}, A2 = f2{h2_hi(pk0), h2_hi(pk1)};
            const f2 Pc = vfma(vfma(A2, dr, dP0), dr, Pv);                        // P(r) = P + P' dr + A2 dr^2 (exact quadratic)
            const f2 dP = vfma(2.0f * A2, dr, dP0);
            const f2 SD = vfma(f2{h2_lo(sk0), h2_lo(sk1)}, dr, SDv);
            const f2 S1 = vfma(f2{h2_hi(sk0), h2_hi(sk1)}, dr, S1v);
            const f2 C0 = vfma(m, a[c], omm * 0.04f);                                // :1412
            const f2 rgb = vfma(a[c] * omm, Pc, vfma(C0, SD, S1));
            JR[c] = vfma(a[c] * omm, dP, vfma(C0, vfma(f2{h2_lo(ek0), h2_lo(ek1)}, dr, f2{h2_lo(dk0), h2_lo(dk1)}), vfma(f2{h2_hi(ek0), h2_hi(ek1)}, dr, f2{h2_hi(dk0), h2_hi(dk1)})));
            if (act0 && !need0) { q.out[i0 * 3 + c] = rgb.x; tot += rgb.x; }
            if (two && !need1) { q.out[i1 * 3 + c] = rgb.y; tot += rgb.y; }
            if (q.jac32) {
                if (act0 && !need0) { q.jac16[c * BP + i0] = as_u(Pc.x); q.jac16[(3 + c) * BP + i0] = as_u(SD.x); q.jac16[(6 + c) * BP + i0] = as_u(JR[c].x); }
                if (two && !need1) { q.jac16[c * BP + i1] = as_u(Pc.y); q.jac16[(3 + c) * BP + i1] = as_u(SD.y); q.jac16[(6 + c) * BP + i1] = as_u(JR[c].y); }
            } else {
                if (act0 && !need0) q.jac16[c * BP + i0] = pack_h2(Pc.x, SD.x);
                if (two && !need1) q.jac16[c * BP + i1] = pack_h2(Pc.y, SD.y);
            }
        }
        if (!q.jac32) {
            if (act0 && !need0) { q.jac16[3 * BP + i0] = pack_h2(JR[0].x, JR[1].x); q.jac16[4 * BP + i0] = pack_h2(JR[2].x, 0.0f); }
            if (two && !need1) { q.jac16[3 * BP + i1] = pack_h2(JR[0].y, JR[1].y); q.jac16[4 * BP + i1] = pack_h2(JR[2].y, 0.0f); }
        }
    }
    // the pixels that left their interval, compacted in a fixed order (wave, then first / second pixel of the lanes): the list's
    // order never enters a result, and it is reproducible all the same
    const unsigned long long b0 = __ballot(need0), b1 = __ballot(need1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n0 = __popcll(b0), nw = n0 + __popcll(b1);
    if (q.block_sums) tot = wave_sum_to_lane63(tot);
    if (lane == 63) { s_cnt[wave] = nw; s_sum[wave] = tot; }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) base += w < wave ? s_cnt[w] : 0;
    uint16_t* list = q.lists + ((long)b * gridDim.x + blockIdx.x) * kLazyBlockPixels;
    if (need0) list[base + __popcll(b0 & below)] = (uint16_t)(2 * threadIdx.x);
    if (need1) list[base + n0 + __popcll(b1 & below)] = (uint16_t)(2 * threadIdx.x + 1);
    if (threadIdx.x == 0) {
        q.counts[(long)b * gridDim.x + blockIdx.x] = (uint32_t)((s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]));
        if (q.block_sums) q.block_sums[(long)b * q.n_sums + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}

// ---- shared by the re-sampling code of both kernels below -------------------------------------------------------------------
__device__ __forceinline__ void pixel_set_r(Pixel& px, f2 r) {
    px.r = r;
    px.alpha2 = pow4(r);
    px.am1 = px.alpha2 - 1.0f;
    const f2 rp1 = r + 1.0f, k = (rp1 * rp1) * 0.125f;
    px.omk = 1.0f - k;
    px.kpe = k + 1e-6f;
    px.dk_dr = rp1 * 0.25f;
    px.g1v = rcp(vfma(px.NoV, px.omk, px.kpe));
}
// (selects, not branches on which bound to update: a reference picked by a branch sends lo / hi to scratch memory)
__device__ __forceinline__ void lazy_kink(float x, float xp, float J, float tol_k, float& lo, float& hi) {
    const bool valid = (J * kLzRhoMax > tol_k) && (fabsf(xp) > 1e-12f);
    const float dk = -x * rcp(xp) * kLzKinkSafety, ad = fabsf(dk) + tol_k * rcp(J);
    const bool both = fabsf(dk) < 2.0f * kLzH, up = dk > 0.0f;
    hi = (valid && (both || up)) ? fminf(hi, ad) : hi;
    lo = (valid && (both || !up)) ? fminf(lo, ad) : lo;
}
template <int LPI>
__device__ __forceinline__ float lane_group_sum(float v) {
#pragma unroll
    for (int msk = 1; msk < LPI; msk <<= 1) v += __shfl_xor(v, msk);
    return v;
}
template <int LPI>
__device__ __forceinline__ float lane_group_min(float v) {
#pragma unroll
    for (int msk = 1; msk < LPI; msk <<= 1) v = fminf(v, __shfl_xor(v, msk));
    return v;
}

// =================================================================================================
// one launch per iteration of hot loop B: backward of iteration t + Adam + forward of iteration t+1
// =================================================================================================
// With the models in HBM the render is a function of (a, r, m, model) that costs a few FMAs, so the streaming backward pass of
// iteration t (jac_bwd_kernel<FUSED>: d loss / d pred from pred_t, the target and the statistics; material gradients; regularisers;
// clamp gating; SaveBest snapshot; Adam) renders iteration t+1 from the parameters it has just updated while the pixel's model is
// still in registers: the maps, the models and the Adam state are read once per iteration, and the jac planes disappear (P, SD and
// d out / d r at r_t are re-formed from the model, which is what the forward of iteration t rendered from).  Pixels whose new
// roughness has left their interval are listed for lazy_refresh_kernel, which patches pred_{t+1} at the head of the next step.
// In parts that do not optimise the albedo the SaveBest snapshot of the albedo is not rewritten (it cannot have changed).
struct LazyStepArgs {
    JacBwdArgs j;             // as for jac_bwd_kernel<FUSED> (jac unused; pred = pred_t)
    uint32_t* plane[kLzPlanes];   // the model planes, one base pointer each (scalar operands of the loads)
    float* pred_next;         // [B,H,W,3] render of the updated parameters
    float* block_sums;        // [B][n_sums]: slot blockIdx.x
    const float *n, *dcache;  // shading normals, diffuse coefficients (re-sampling only)
    uint32_t* counts;         // per workgroup: how many of its pixels were re-sampled, and which (inspection: matpbr_lazy_state_unpack)
    uint16_t* lists;
    int n_sums;
    float tol;
    int attached;             // d out / d r through the GGX quadrature nodes: the models' slopes (MATPBR_FLAG_ATTACHED_SAMPLING) instead of dSD, dS1
    // The iteration's statistics are folded HERE (no loss_final2 launch): every workgroup folds the rows of partial sums that
    // loss_sums2_kernel<3> left for its image (fixed order: the same bits in every workgroup), forms the iteration's scalars from the OLD
    // SaveBest / EarlyStopping state (`state_old`) and uses them; workgroup 0 of the image also writes the NEW state to `state_new` and to
    // the caller's statistics rows (j.stats, written here).  Old and new are different buffers (the caller alternates them): a workgroup
    // that starts late still reads the state its siblings read.
    const float* fold_part;   // nullable: then the statistics were committed by loss_final2_kernel and j.stats is read as before
    int fold_rows;
    float* reg_sums;          // nullable [B][n_sums][3]: per-workgroup sums of |a - a0|, |r - r0|, |m - m0| at the UPDATED parameters (the next
                              // iteration's regulariser terms: its statistics pass then reads pred and the target only)
    int reg_from_part;        // the regulariser sums of THIS iteration are in the tail of fold_part (carried by the step before)
    const float* state_old;   // [B][kStateStride]
    float* state_new;         // [B][kStateStride]
    int rotate;               // MATPBR_FLAG_ROTATE_BEST: SaveBest without copies.  The live maps and the render live in two buffers each
                              // (j.pa / alt_a, ..., pred_buf[0] / pred_buf[1]); the state row says which holds the current values.  An improving
                              // iteration declares the buffer it has just read "best" and writes the new values into the other one; any
                              // other iteration updates in place.  matpbr_brdf_phase_resolve puts things where the caller expects them.
    float *alt_a, *alt_r, *alt_m;   // second buffer of each live map (the caller's best_*), null for a map the part does not move
    float* pred_buf[2];
    float* stats_out;         // [B][kStatsStride] the public rows
    float* history;
    int hist_len, batch, es_patience;
    float es_min_delta;
    // lazy_pstep_kernel (the folded, persistent form of the step): the folded planes, tiles per workgroup, tiles per image
    uint32_t* fplane[kFxPlanes];
    long long* walk_fix;      // [B][nblk] fixed-point sums of the re-sampled pixels' render (lazy_fold_bytes)
    int tiles_per_wg, n_tiles;
    uint32_t* walk_cnt;       // [B][2][kWalkShards]: entries of the walk queue's lists, per iteration parity
    uint32_t* walk_queue;     // [B][kWalkShards][walk_shard_cap(P)] pixel indices
    int walk_par;             // this iteration's parity
    // round 6: the statistics of iteration t + 1 are formed where its render is formed (no statistics launch behind the step, none of its 24 B/pixel).
    // With x0 = max(pred ratio_t, eps)^(1/2.2) (the exposure ratio of THIS iteration as the expansion point) and d0 = x0 - gt, every block leaves
    //     S = sum pred,  A = sum d0^2,  Bq = sum d0 x0,  Cq = sum x0^2,  L = sum |d0|,  Mq = sum sign(d0) x0        (Bq, Cq, Mq: pixels above eps only)
    // and the head of the next step, which knows ratio_{t+1} = sum gt / S, has with e = (ratio_{t+1} / ratio_t)^(1/2.2) - 1 (a few 1e-4)
    //     sum (x - gt)^2 = A + 2 e Bq + e^2 Cq   exactly,     sum |x - gt| = L + e Mq   up to the pixels whose sign flips inside e (a relative 1e-7).
    const float* rec_in;      // [B][nblk][9] per block: S, (A, Bq, Cq, L, Mq) of the streamed pixels, the three regulariser sums -- as the step BEFORE left them
    float* rec_out;           // ... and where this step leaves its own (the other of two sets: a launch's heads read while its first workgroups finish)
    int no_pred;              // round 6: the folded steps do not store the render they form (nothing of the loop reads it back: matpbr_brdf_phase_resolve evaluates the models for the caller)
    long long* walk_acc;      // [B][2][kWalkShards][6] (S, A, Bq, Cq, L, Mq) of the walked pixels in fixed point (kWalkFix), per iteration parity and queue shard: integer atomics, order-free
    int acc_mode;             // the head takes this iteration's statistics from block_sums / block_acc / walk_acc / reg_sums (t > 1) instead of fold_part's rows
};
// one channel of a freshly rendered pixel into the five sums above
__device__ __forceinline__ void loss_acc(float pred, float gt, float ratio, float (&acc)[5]) {
    const float x = pred * ratio;
    const bool live = x > kLossEps;
    const float x0 = pow_inv_gamma(fmaxf(x, kLossEps)), d0 = x0 - gt, xq = live ? x0 : 0.0f;
    acc[0] = fmaf(d0, d0, acc[0]);
    acc[1] = fmaf(d0, xq, acc[1]);
    acc[2] = fmaf(xq, xq, acc[2]);
    acc[3] += fabsf(d0);
    acc[4] += d0 > 0.0f ? xq : (d0 < 0.0f ? -xq : 0.0f);
}
// LDS exchange between the lanes of ONE wave: DS operations of a wave execute in order, so only the compiler has to be held back
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// what the lane that owns a pixel hands to the lanes that walk its samples
struct LazyRecord { float a[3], r, m, dr; };
constexpr int kResampleWaves = 256;    // waves (64-thread workgroups) of lazy_resample_kernel per image at most, eight listed pixels each per pass
constexpr int kRecStride = 7;         // floats per record: pixel, a (clamped, updated), r, m, r - r_ref (odd stride: conflict-free)
constexpr int kWalkVals = 26;         // per-sample contributions: S0, S1, dS0, dS1 at r (12), S0, S1 at r + dir h (6), interval lo / hi

// Loads / stores at (wave-uniform base pointer) + (32-bit per-lane byte offset): the form the hardware addresses directly
// (global_load ... v_off, s[base]); indexing 21 planes through 64-bit per-lane arithmetic costs more VALU issue slots than the
// arithmetic of the step itself.  Byte offsets stay below 2^32 (12 B x B*H*W: 357 M pixels).
__device__ __forceinline__ uint32_t ldu(const void* base, unsigned off) { return *(const uint32_t*)((const char*)base + off); }
__device__ __forceinline__ float ldf(const void* base, unsigned off) { return *(const float*)((const char*)base + off); }
__device__ __forceinline__ void stf(void* base, unsigned off, float v) { *(float*)((char*)base + off) = v; }
// 12 bytes per lane of an HWC map: global_load_dwordx3 / global_store_dwordx3 at (uniform base) + (32-bit lane offset)
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const void* base, unsigned off) { return *(const F3*)((const char*)base + off); }
__device__ __forceinline__ void st3(void* base, unsigned off, float x, float y, float z) { *(F3*)((char*)base + off) = F3{x, y, z}; }

// one pixel of lazy_step_kernel; returns whether the pixel's new roughness has left its model's interval, adds its render to `tot`
struct StepPtrs {             // where this image's iteration reads its parameters and writes the new ones and the next render (uniform)
    const float *a, *r, *m;
    float *pa, *pr, *pm, *pred_next;
};
__device__ __forceinline__ bool lazy_step_pixel(const LazyStepArgs& qs, const StepPtrs& sp, unsigned i, float ratio, float sr, bool improved, float& tot,
                                                float (&reg)[3], LazyRecord& rec) {
    if (qs.rotate) improved = false;      // no snapshot stores: the buffer just read IS the snapshot
    const JacBwdArgs& q = qs.j;
    const unsigned o1 = i * 4u, o3 = i * 12u;
    // HWC maps as ONE 12-byte access per lane (as the folded step): a third of the memory instructions of the albedo's streams, and Adam's
    // moments of the albedo in registers between their load and their store -- the 'arm' part of the 8 x 512^2 shard 140 -> 123 us per
    // iteration on one box (tools/arm_trace.py), the same bits
    const F3 ra3 = ld3(sp.a, o3), gt3 = ld3(q.gt_srgb, o3);
    const float ra[3] = {ra3.x, ra3.y, ra3.z}, rr = ldf(sp.r, o1), rm = ldf(sp.m, o1);
    const float gt[3] = {gt3.x, gt3.y, gt3.z};
    // the pixel's model
    const float rref = as_f(ldu(qs.plane[kLzRref], o1));
    const uint32_t lohi = ldu(qs.plane[kLzLoHi], o1);
    float Pv[3], SDv[3], S1v[3];
    uint32_t pk[3], sk[3], dk[3], ek[3];
    // a part that leaves the roughness alone never moves away from r_ref: the twelve planes of slopes (48 of the 92 B/pixel of a model)
    // multiply r - r_ref = 0 and are not read (uniform branch)
    const bool slopes = (q.part_mask & MATPBR_PART_R) != 0 || q.d_r != nullptr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Pv[c] = as_f(ldu(qs.plane[kLzP + c], o1)); SDv[c] = as_f(ldu(qs.plane[kLzSD + c], o1)); S1v[c] = as_f(ldu(qs.plane[kLzS1 + c], o1));
        pk[c] = sk[c] = dk[c] = ek[c] = 0u;
    }
    if (slopes) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pk[c] = ldu(qs.plane[kLzPk + c], o1); sk[c] = ldu(qs.plane[kLzSk + c], o1); dk[c] = ldu(qs.plane[kLzDk + c], o1);
            if (!qs.attached) ek[c] = ldu(qs.plane[kLzEk + c], o1);
        }
    }
    float a[3];
    const float r = fminf(fmaxf(rr, 0.07f), 1.0f), m = fminf(fmaxf(rm, 0.0f), 1.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = fminf(fmaxf(ra[c], 0.0f), 1.0f);
    // ---- backward of iteration t at (a, r, m): d loss / d pred of 3 (l1/mse) mse + l1 on xs = max(pred ratio, eps)^(1/2.2)  (:388-418)
    const float dr = r - rref, omm = 1.0f - m;
    float da[3], drr = 0.0f, dm = 0.0f, xs_keep[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // the render of iteration t is not read back: it IS the model at the current parameters (the expression below is the one that
        // wrote pred -- the step before, the re-sampling, or the first render -- on the same operands: the same bits)
        const float dP0 = h2_lo(pk[c]), A2 = h2_hi(pk[c]);
        const float Pc = fmaf(fmaf(A2, dr, dP0), dr, Pv[c]), dPc = fmaf(2.0f * A2, dr, dP0);
        const float SD = fmaf(h2_lo(sk[c]), dr, SDv[c]);
        const float C0 = fmaf(m, a[c], omm * 0.04f);
        const float prc = fmaf(a[c] * omm, Pc, fmaf(C0, SD, fmaf(h2_hi(sk[c]), dr, S1v[c])));
        const float x = prc * ratio;
        const float xc = fmaxf(x, kLossEps);
        const float xs = pow_inv_gamma(xc);
        const float d = xs - gt[c];
        const float dxs = x > kLossEps ? xs * rcp(xc) * (1.0f / 2.2f) : 0.0f;
        const float go = ratio * dxs * fmaf(6.0f * sr, d, fsign(d)) * q.inv_n3;
        xs_keep[c] = xs;
        // d out / d r: the stop-gradient convention (dSD, dS1) by default; with `attached` the derivative of the rendered value through
        // the sample directions, which is what the models' slopes are (the live reference's convention, mi_plugin.py:227-230,1335-1341)
        const float JR = fmaf(a[c] * omm, dPc, qs.attached ? fmaf(C0, h2_lo(sk[c]), h2_hi(sk[c]))
                                                            : fmaf(C0, fmaf(h2_lo(ek[c]), dr, h2_lo(dk[c])), fmaf(h2_hi(ek[c]), dr, h2_hi(dk[c]))));
        da[c] = go * fmaf(m, SD, omm * Pc);
        dm = fmaf(go, fmaf(a[c] - 0.04f, SD, -(a[c] * Pc)), dm);
        drr = fmaf(go, JR, drr);
    }
    float na[3] = {ra[0], ra[1], ra[2]}, nr = rr, nm = rm;     // the raw parameters after the step
    if (q.part_mask & MATPBR_PART_A) {
        const F3 a03 = ld3(q.a0, o3);
        const float a0v[3] = {a03.x, a03.y, a03.z};
        float mo[3] = {0.0f, 0.0f, 0.0f}, vo[3] = {0.0f, 0.0f, 0.0f}, gs[3];
        if (q.am[0]) {
            const F3 m3 = ld3(q.am[0], o3), v3 = ld3(q.av[0], o3);
            mo[0] = m3.x; mo[1] = m3.y; mo[2] = m3.z; vo[0] = v3.x; vo[1] = v3.y; vo[2] = v3.z;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a0c = a0v[c];
            float gsum = da[c] + q.scale_delta * q.inv_n3 * fsign(a[c] - a0c);                                                   // :398,418
            gsum = (ra[c] >= 0.0f && ra[c] <= 1.0f) ? gsum : 0.0f;                                                               // clamp backward
            gs[c] = gsum;
            if (q.am[0]) {                                     // adam_update on registers: the same operations
                const float mi = fmaf(q.b1, mo[c], (1.0f - q.b1) * gsum);
                const float vi = fmaf(q.b2, vo[c], (1.0f - q.b2) * gsum * gsum);
                mo[c] = mi; vo[c] = vi;
                na[c] = ra[c] - q.lr_over_bc1 * mi / fmaf(fsqrt(vi), q.inv_sqrt_bc2, q.eps);
            }
            reg[0] += fabsf(fminf(fmaxf(na[c], 0.0f), 1.0f) - a0c);
        }
        if (q.d_a) st3(q.d_a, o3, gs[0], gs[1], gs[2]);
        if (improved && q.best_a) st3(q.best_a, o3, a[0], a[1], a[2]);
        if (q.am[0]) {
            st3(q.am[0], o3, mo[0], mo[1], mo[2]);
            st3(q.av[0], o3, vo[0], vo[1], vo[2]);
            st3(sp.pa, o3, na[0], na[1], na[2]);
        }
    } else if (q.d_a) {
        st3(q.d_a, o3, (ra[0] >= 0.0f && ra[0] <= 1.0f) ? da[0] : 0.0f, (ra[1] >= 0.0f && ra[1] <= 1.0f) ? da[1] : 0.0f, (ra[2] >= 0.0f && ra[2] <= 1.0f) ? da[2] : 0.0f);
    }
    if (improved && q.best_img) st3(q.best_img, o3, xs_keep[0], xs_keep[1], xs_keep[2]);
    const float r0v = (q.part_mask & MATPBR_PART_R) ? ldf(q.r0, o1) : 0.0f, m0v = (q.part_mask & MATPBR_PART_M) ? ldf(q.m0, o1) : 0.0f;
    float gr = drr + ((q.part_mask & MATPBR_PART_R) ? q.scale_delta * q.inv_n1 * fsign(r - r0v) : 0.0f);
    float gm = dm + ((q.part_mask & MATPBR_PART_M) ? q.scale_delta * q.inv_n1 * fsign(m - m0v) : 0.0f);
    gr = (rr >= 0.07f && rr <= 1.0f) ? gr : 0.0f;
    gm = (rm >= 0.0f && rm <= 1.0f) ? gm : 0.0f;
    if (q.d_r) stf(q.d_r, o1, gr);
    if (q.d_m) stf(q.d_m, o1, gm);
    if (improved && q.best_r) stf(q.best_r, o1, r);
    if (improved && q.best_m) stf(q.best_m, o1, m);
    if ((q.part_mask & MATPBR_PART_R) && q.am[1]) { nr = adam_update(rr, gr, q.am[1], q.av[1], (long)i, q); stf(sp.pr, o1, nr); }
    if ((q.part_mask & MATPBR_PART_M) && q.am[2]) { nm = adam_update(rm, gm, q.am[2], q.av[2], (long)i, q); stf(sp.pm, o1, nm); }
    // ---- forward of iteration t+1 from the same model
    const float r1 = fminf(fmaxf(nr, 0.07f), 1.0f), m1 = fminf(fmaxf(nm, 0.0f), 1.0f), dr1 = r1 - rref, omm1 = 1.0f - m1;
    if (q.part_mask & MATPBR_PART_R) reg[1] += fabsf(r1 - r0v);
    if (q.part_mask & MATPBR_PART_M) reg[2] += fabsf(m1 - m0v);
    const bool need = !(dr1 >= -h2_lo(lohi) && dr1 <= h2_hi(lohi));
    rec.r = r1; rec.m = m1; rec.dr = dr1;
    float rgb3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a1 = fminf(fmaxf(na[c], 0.0f), 1.0f);
        const float Pc = fmaf(fmaf(h2_hi(pk[c]), dr1, h2_lo(pk[c])), dr1, Pv[c]);
        const float SD = fmaf(h2_lo(sk[c]), dr1, SDv[c]), S1 = fmaf(h2_hi(sk[c]), dr1, S1v[c]);
        const float C0 = fmaf(m1, a1, omm1 * 0.04f);
        const float rgb = fmaf(a1 * omm1, Pc, fmaf(C0, SD, S1));
        rec.a[c] = a1;
        rgb3[c] = rgb;
        if (!need) tot += rgb;
    }
    if (!need) st3(sp.pred_next, o3, rgb3[0], rgb3[1], rgb3[2]);
    return need;
}

// SH radiance with the (pre-normalised) coefficients in LDS: the walk below runs one sample per lane inside a streaming kernel and
// cannot afford the 76 VGPRs that hold the coefficients in shade_kernel
struct RadianceLdsUse {
    const float* c;
    f2* L;
    template <int K> __device__ __forceinline__ void operator()(f2 Bk) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) L[ch] = K == 0 ? f2{c[ch], c[ch]} : vfma(Bk, c[3 * K + ch], L[ch]);
    }
};

// A workgroup takes 512 consecutive pixels, thread t the pixels t and t + 256 of them (every load instruction of a wave covers 64
// consecutive pixels, the two pixels' independent chains interleave).  Pixels whose new roughness has left their model's interval
// -- a few of the 512 per iteration -- are re-sampled before the workgroup ends: eight lanes per pixel, four azimuths x (r, r + dir h:
// the one-sided difference that gives the slopes), each lane walking the rings of its azimuth, contributions folded by a fixed
// butterfly; waves without listed pixels leave at once.  The render this launch leaves behind is complete.
__global__ __launch_bounds__(kBlock, 4) void lazy_step_kernel(const LazyStepArgs qs, const float* __restrict__ light, const Geom g, const RuleTable tab) {
    __shared__ float s_sum[4];
    __shared__ int s_cnt[4];
    __shared__ float s_state[kStateStride];
    __shared__ float s_fold[4][6];
    const JacBwdArgs& q = qs.j;
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const int q0 = blockIdx.x * kLazyBlockPixels + threadIdx.x, q1 = q0 + kBlock;
    float ratio, sr, gt_sum;
    bool improved;
    if (qs.fold_part != nullptr) {
        const float* old = qs.state_old + b * kStateStride;
        if (old[kStStopped] > 0.5f) {                          // EarlyStopping fired in an earlier iteration (uniform): nothing to do
            if (blockIdx.x == 0 && threadIdx.x < kStateStride) {
                float v = old[threadIdx.x];
                if (threadIdx.x == kStStopped) v = 2.0f;
                if (threadIdx.x == kStImproved) v = 0.0f;
                qs.state_new[b * kStateStride + threadIdx.x] = v;
                if (threadIdx.x < kStatsStride) qs.stats_out[b * kStatsStride + threadIdx.x] = v;
            }
            return;
        }
        // fold: thread i takes row i (fold_rows <= kBlock), waves by a fixed DPP tree, the four wave totals in fixed order
        const float* rows = qs.fold_part + (long)b * step_part_stride(qs.fold_rows);
        float v[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = threadIdx.x; i < qs.fold_rows; i += kBlock) {
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] += rows[(long)i * 5 + k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float w = wave_sum_to_lane63(v[k]);
            if ((threadIdx.x & 63) == 63) s_fold[threadIdx.x >> 6][k] = w;
        }
        float st[kStatsStride];                                // thread 0: the old state in registers (sixteen independent loads, issued
        const float sp_total = rows[(long)qs.fold_rows * 5];   // before the barrier), the commit on registers, one burst of LDS writes
        float sel_old = 0.0f, bratio = -1.0f;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < kStatsStride; ++i) st[i] = old[i];
            sel_old = old[kStSel]; bratio = old[kStBestRatio];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float t[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) t[k] = (s_fold[0][k] + s_fold[1][k]) + (s_fold[2][k] + s_fold[3][k]);
            st[kStRatio] = st[kStGtSum] / sp_total;
            const float mse = t[0] * q.inv_n3, l1 = t[1] * q.inv_n3;
            if (qs.reg_from_part) { t[2] = rows[(long)qs.fold_rows * 5 + 1]; t[3] = rows[(long)qs.fold_rows * 5 + 2]; t[4] = rows[(long)qs.fold_rows * 5 + 3]; }
            const float la = (q.part_mask & MATPBR_PART_A) ? t[2] * q.inv_n3 : 0.0f;
            const float lr = (q.part_mask & MATPBR_PART_R) ? t[3] * q.inv_n1 : 0.0f;
            const float lm = (q.part_mask & MATPBR_PART_M) ? t[4] * q.inv_n1 : 0.0f;
            stats_commit(st, mse, l1, l1 / mse /* scale_raito, :411 */, la, lr, lm, q.scale_delta, qs.es_patience, qs.es_min_delta,
                         blockIdx.x == 0 ? qs.history : nullptr, qs.hist_len, qs.batch, b);
#pragma unroll
            for (int i = 0; i < kStatsStride; ++i) s_state[i] = st[i];
            const bool imp = st[kStImproved] > 0.5f && qs.rotate != 0;
            s_state[kStSel] = imp ? 1.0f - sel_old : sel_old;          // the buffer the new values go to: the other one after an improvement
            s_state[kStBestRatio] = imp ? st[kStRatio] : bratio;
            s_state[kStSelOld] = sel_old;
            s_state[kStSelOld + 1] = 0.0f;
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < kStateStride) {
            qs.state_new[b * kStateStride + threadIdx.x] = s_state[threadIdx.x];
            if (threadIdx.x < kStatsStride) qs.stats_out[b * kStatsStride + threadIdx.x] = s_state[threadIdx.x];
        }
        ratio = s_state[kStRatio];
        sr = s_state[kStSr];
        improved = s_state[kStImproved] > 0.5f;
        gt_sum = s_state[kStGtSum];
    } else {
        if (q.check_stop && img_stopped_before(q.stats, b)) return;
        ratio = q.stats[b * kStatsStride + kStRatio];
        sr = q.stats[b * kStatsStride + kStSr];
        improved = q.stats[b * kStatsStride + kStImproved] > 0.5f;
        gt_sum = q.stats[b * kStatsStride + kStGtSum];
    }
    float tot = 0.0f;
    bool need0 = false, need1 = false;
    LazyRecord rec0, rec1;
    float reg[3] = {0.0f, 0.0f, 0.0f};
    StepPtrs sp{q.a, q.r, q.m, q.pa, q.pr, q.pm, qs.pred_next};
    if (qs.rotate) {                                           // uniform per image: scalar selects of the base pointers
        const bool rd1 = __builtin_amdgcn_readfirstlane((int)(s_state[kStSelOld] > 0.5f)) != 0;
        const bool wr1 = __builtin_amdgcn_readfirstlane((int)(s_state[kStSel] > 0.5f)) != 0;
        if (qs.alt_a) { sp.a = rd1 ? qs.alt_a : q.pa; sp.pa = wr1 ? qs.alt_a : q.pa; }
        if (qs.alt_r) { sp.r = rd1 ? qs.alt_r : q.pr; sp.pr = wr1 ? qs.alt_r : q.pr; }
        if (qs.alt_m) { sp.m = rd1 ? qs.alt_m : q.pm; sp.pm = wr1 ? qs.alt_m : q.pm; }
        sp.pred_next = qs.pred_buf[wr1 ? 1 : 0];
    }
    if (q0 < P) need0 = lazy_step_pixel(qs, sp, (unsigned)(b * P + q0), ratio, sr, improved, tot, reg, rec0);
    if (q1 < P) need1 = lazy_step_pixel(qs, sp, (unsigned)(b * P + q1), ratio, sr, improved, tot, reg, rec1);
    if (qs.reg_sums) {                                         // fixed order: DPP tree per wave, the four waves in order
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w = wave_sum_to_lane63(reg[k]);
            if ((threadIdx.x & 63) == 63) s_fold[threadIdx.x >> 6][k] = w;
        }
    }
    const unsigned long long b0 = __ballot(need0), b1 = __ballot(need1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n0 = __popcll(b0);
    if (lane == 63) s_cnt[wave] = n0 + __popcll(b1);
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) base += w < wave ? s_cnt[w] : 0;
    const int count = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
    const int idx0 = base + __popcll(b0 & below), idx1 = base + n0 + __popcll(b1 & below);   // fixed order: wave, first / second pixel, lane
    if (qs.lists) {
        uint16_t* list = qs.lists + ((long)b * gridDim.x + blockIdx.x) * kLazyBlockPixels;
        if (need0) list[idx0] = (uint16_t)threadIdx.x;
        if (need1) list[idx1] = (uint16_t)(kBlock + threadIdx.x);
        if (threadIdx.x == 0) qs.counts[(long)b * gridDim.x + blockIdx.x] = (uint32_t)count;
    }
    tot = wave_sum_to_lane63(tot);
    if (lane == 63) s_sum[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) qs.block_sums[(long)b * qs.n_sums + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    if (qs.reg_sums && threadIdx.x < 3)
        qs.reg_sums[((long)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
            (s_fold[0][threadIdx.x] + s_fold[1][threadIdx.x]) + (s_fold[2][threadIdx.x] + s_fold[3][threadIdx.x]);
}

// The pixels the step kernel listed (their new roughness has left their model's interval: a few of a workgroup's 512 per iteration) are
// re-sampled by this launch, one workgroup per workgroup of the step kernel: eight lanes per pixel, four azimuths x (r, r + dir h: the
// one-sided difference that gives the slopes), each lane walking the rings of its azimuth, contributions folded by a fixed butterfly.
// It rebuilds their models, writes their render and leaves the sum of it beside the step kernel's sums (slot nblk + blockIdx.x of the
// image).  64-thread workgroups, kResampleWaves per image, eight items per wave and pass.  Inside the step kernel the same walk cost 29 of its 111 us at 8 x 512^2 (tools/step_parts_ab.sh): three
// quarters of the workgroups list a pixel or two, and a workgroup that walks holds its registers and LDS through two more memory round
// trips while nothing of it streams; here the walkers are a launch of their own and the step kernel is a pure streaming pass (80
// registers, no spills).
// One listed pixel re-sampled by the eight lanes `sub` = 0..7 of its group (four azimuths x (r, r + dir h: the one-sided difference that gives
// the slopes), each lane walking the rings of its azimuth, contributions folded by a fixed butterfly): rebuilds the pixel's model, writes its
// render into sp.pred_next and adds it to `tot` (lane sub == 0 of an item that exists; the other lanes return without side effects).
// Shared by lazy_resample_kernel (FOLD = false) and lazy_pwalk_kernel behind lazy_pstep_kernel (FOLD = true: the folded planes are rewritten
// too and the render is the folded expression).  Tables in LDS: light coefficients x basis normalisation, GGX rings, azimuths.
template <bool FOLD, bool LEAN = false>
__device__ __forceinline__ void resample_walk_pixel(const LazyStepArgs& qs, const StepPtrs& sp, const float* s_light, const float4* s_ring,
                                                    const float2* s_saz, const Geom& g, const RuleTable& tab, int b, int P, long BPl, int p,
                                                    bool item_ok, int sub, float floor_, float tol_k, float tol_s, float& tot, float* rgb_out = nullptr) {
    const JacBwdArgs& q = qs.j;
    (void)q;
    const int half = sub >> 2, azi = sub & 3;
    const unsigned i = (unsigned)(b * P + p), o1 = i * 4u, o3 = i * 12u, of = fx_off(i);
    float rc[7];
#pragma unroll
    for (int c = 0; c < 3; ++c) rc[1 + c] = fminf(fmaxf(ldf(sp.a, o3 + 4 * c), 0.0f), 1.0f);
    rc[4] = fminf(fmaxf(ldf(sp.r, o1), 0.07f), 1.0f);
    rc[5] = fminf(fmaxf(ldf(sp.m, o1), 0.0f), 1.0f);
    rc[6] = rc[4] - as_f(ldu(qs.plane[kLzRref], o1));
    const float rc_r = rc[4], mv = rc[5], dr = rc[6];
    const float rho_old = as_f(ldu(qs.plane[kLzRho], o1));
    // geometry of the pixel (the same at r and at r + dir h: plain floats), as load_pixel forms it
    float nn[3], ss[3], tt[3], vx, vy, vz, NoV;
    {
        float nv[3] = {ldf(qs.n, o3), ldf(qs.n, o3 + 4), ldf(qs.n, o3 + 8)};
        const float inl = rsq(fmaxf(dot3(nv, nv), 1e-30f));
#pragma unroll
        for (int c = 0; c < 3; ++c) nn[c] = nv[c] * inl;
        const float fi = (float)(p / g.W), fj = (float)(p % g.W);
        const float x = (g.cx - fj) * g.inv_f, y = (fi - g.cy) * g.inv_f, il = rsq(fmaf(x, x, fmaf(y, y, 1.0f)));
        const float wo[3] = {x * il, y * il, il};
        frame(nn, ss, tt);
        vx = dot3(ss, wo); vy = dot3(tt, wo); vz = dot3(nn, wo);
        NoV = fmaxf(vz, 0.0f);
    }
    float dir = dr < 0.0f ? -1.0f : 1.0f;
    if (rc_r + dir * kLzH > 1.0f || rc_r + dir * kLzH < 0.07f) dir = -dir;
    float C0[3], kd[3], Pc[3], dP[3], A2[3], iscale[3], ijscale[3], pSD[3], pS1[3], pdSD[3], pdS1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float A0 = ldf(qs.dcache + c * BPl, o1), A1 = ldf(qs.dcache + (3 + c) * BPl, o1);
        A2[c] = ldf(qs.dcache + (6 + c) * BPl, o1);
        const uint32_t sk = ldu(qs.plane[kLzSk + c], o1);      // what the old model predicts at the new roughness
        pSD[c] = fmaf(h2_lo(sk), dr, as_f(ldu(qs.plane[kLzSD + c], o1)));
        pS1[c] = fmaf(h2_hi(sk), dr, as_f(ldu(qs.plane[kLzS1 + c], o1)));
        const uint32_t ek = ldu(qs.plane[kLzEk + c], o1);      // ... and of the detached derivatives
        pdSD[c] = fmaf(h2_lo(ek), dr, as_f(ldu(qs.plane[kLzD32 + c], o1)));
        pdS1[c] = fmaf(h2_hi(ek), dr, as_f(ldu(qs.plane[kLzD32 + 3 + c], o1)));
        Pc[c] = fmaf(fmaf(A2[c], rc_r, A1), rc_r, A0);
        dP[c] = fmaf(2.0f * rc_r, A2[c], A1);
        kd[c] = rc[1 + c] * (1.0f - mv);
        C0[c] = fmaf(mv, rc[1 + c], (1.0f - mv) * 0.04f);
        iscale[c] = 1.0f / fmaxf(fabsf(fmaf(kd[c], Pc[c], fmaf(C0[c], pSD[c], pS1[c]))), floor_);
        // the derivative's scale, from the old model's prediction like the render's; premultiplied by tol_k / tol_kj: one lazy_kink call serves both
        ijscale[c] = (kLzTolK / kLzTolKJ) / fmaxf(fabsf(fmaf(kd[c], dP[c], fmaf(C0[c], pdSD[c], pdS1[c]))), kLzJFloor * floor_);
    }
    // ---- this lane's samples: one azimuth of every ring, at r (sub 0-3) or at r + dir h (sub 4-7)  (spec_ring / spec_sample /
    // spec_accumulate of matpbr_shade.hpp, one value per lane)
    float S0[3] = {0, 0, 0}, S1[3] = {0, 0, 0}, dS0[3] = {0, 0, 0}, dS1[3] = {0, 0, 0}, klo = 1e30f, khi = 1e30f;
    {
        const float rr = rc_r + (half ? dir * kLzH : 0.0f);
        const float alpha2 = pow4(rr), am1 = alpha2 - 1.0f, rp1 = rr + 1.0f, kk = (rp1 * rp1) * 0.125f;
        const float omk = 1.0f - kk, kpe = kk + 1e-6f, dk_dr = rp1 * 0.25f, g1v = rcp(fmaf(NoV, omk, kpe));
        const float four_over_r = 4.0f * rcp(rr), cv = dk_dr * g1v * (1.0f - NoV), r3x4 = 4.0f * rr * rr * rr, g1l0 = rcp(kpe);
        for (int ring = 0; ring < tab.nu_s; ++ring) {
            const float4 rg = s_ring[ring];
            const float rq = rcp(fmaf(am1, rg.x, 1.0f));
            const float cos2 = rq * rg.y, sin2 = (alpha2 * rg.x) * rq, ict = rsq(cos2);
            const float ct = cos2 * ict, st = sin2 * rsq(sin2), ringw = (g1v * rg.z) * ict, idq = rcp(rq + 1e-6f * rcp(alpha2));
            const float lam0 = fmaf(four_over_r, fmaf(rq * idq, -2.0f * rg.y, 1.0f), -cv);
            const float gq = r3x4 * rg.x * rq * cos2;      // d sin^2 theta_h / dr  (mi_specular_sampler :232-233)
            const float stp = 0.5f * gq * rcp(st), ctp = -0.5f * gq * rcp(ct);
            for (int j = azi; j < tab.nphi_s; j += 4) {
                const float2 az = s_saz[ring * kMaxAz + j];
                const float whx = st * az.x, why = st * az.y;
                const float d = fmaf(ct, vz, fmaf(why, vy, whx * vx)), d2 = d + d;
                const float wlx = fmaf(d2, whx, -vx), wly = fmaf(d2, why, -vy), wlz = fmaf(d2, ct, -vz);      // 2 (wo.wh) wh - wo  (:245)
                float wi[3], B[kNSH], L[3] = {0, 0, 0};
#pragma unroll
                for (int c = 0; c < 3; ++c) wi[c] = fmaf(wlz, nn[c], fmaf(wly, tt[c], wlx * ss[c]));
                const float NoL = fmaxf(wlz, 0.0f), dpos = fmaxf(d, 0.0f), g1l = rcp(fmaf(NoL, omk, kpe)), x5 = pow5(1.0f - dpos);
                const float wgt = (ringw * g1l) * (NoL * dpos);
                sh_poly(wi, B);
                // LEAN (a caller short of registers): the 75 coefficients are read from LDS per sample -- left to itself the compiler keeps them
                // in registers across the ring loop (they are loop invariants: 243 registers, the fastest walk).  Through an opaque ZERO OFFSET,
                // not an opaque pointer: the address must stay an LDS address -- a laundered pointer is a generic one, and every read becomes a
                // flat load with a full wait behind it
                unsigned zoff = 0u;
                if (LEAN) asm volatile("" : "+v"(zoff));
                const float* lp = s_light + (zoff << 2);
#pragma unroll
                for (int k = 0; k < kNSH; ++k) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) L[c] = fmaf(B[k], lp[3 * k + c], L[c]);
                }
                const float lam = fmaf(dk_dr * g1l, NoL - 1.0f, lam0);
                const float wx = wgt * x5, wl = wgt * lam, wlx5 = wl * x5;
                const float lamk1 = fabsf(lam0 - dk_dr * g1l0), lamk2 = fabsf(lam);      // |d ln(weight)/dr| at n.wi = 0 / at this sample
                float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    S0[c] = fmaf(wgt, L[c], S0[c]); S1[c] = fmaf(wx, L[c], S1[c]);
                    dS0[c] = fmaf(wl, L[c], dS0[c]); dS1[c] = fmaf(wlx5, L[c], dS1[c]);
                    const float aL = fabsf(L[c]), sc1 = fmaxf(iscale[c], lamk1 * ijscale[c]), sc2 = fmaxf(iscale[c], lamk2 * ijscale[c]);
                    m1 = fmaxf(m1, fmaf(1.0f - C0[c], x5, C0[c]) * aL * sc1);
                    m2 = fmaxf(m2, aL * sc2);
                }
                // where this sample's clamped variables n.wi and wo.h cross zero, to first order in r
                const float dp = fmaf(stp, fmaf(az.x, vx, az.y * vy), ctp * vz);
                const float wlzp = 2.0f * fmaf(dp, ct, d * ctp);
                lazy_kink(wlz, wlzp, ringw * g1l0 * dpos * m1 * fabsf(wlzp), tol_k, klo, khi);
                lazy_kink(d, dp, ringw * g1l * NoL * m2 * fabsf(dp), tol_k, klo, khi);
            }
        }
    }
    // fold over the four azimuth lanes with a fixed butterfly (the same tree for every pixel: reproducible)
    float fv[kWalkVals];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fv[c] = lane_group_sum<4>(S0[c]); fv[3 + c] = lane_group_sum<4>(S1[c]);
        fv[6 + c] = lane_group_sum<4>(dS0[c]); fv[9 + c] = lane_group_sum<4>(dS1[c]);
    }
    fv[18] = lane_group_min<4>(klo);
    fv[19] = lane_group_min<4>(khi);
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // the sums at r + dir h, from the lanes four further up
        fv[12 + c] = __shfl_down(fv[c], 4);
        fv[15 + c] = __shfl_down(fv[3 + c], 4);
        fv[20 + c] = __shfl_down(fv[6 + c], 4);      // dS0, dS1 at r + dir h: the slopes of the detached derivatives
        fv[23 + c] = __shfl_down(fv[9 + c], 4);
    }
    if (item_ok && sub == 0) {
        const float ih = dir * (1.0f / kLzH);
        float vSD[3], vS1[3], gSD[3], gS1[3], dSD[3], dS1v[3], eSD[3], eS1[3], x2h[3] = {0.0f, 0.0f, 0.0f}, jx1[3] = {0.0f, 0.0f, 0.0f}, jy1[3] = {0.0f, 0.0f, 0.0f};
        float fx0[3] = {0.0f, 0.0f, 0.0f}, fy0[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vSD[c] = fv[c] - fv[3 + c];
            vS1[c] = fv[3 + c];
            gSD[c] = ((fv[12 + c] - fv[15 + c]) - vSD[c]) * ih;
            gS1[c] = (fv[15 + c] - fv[3 + c]) * ih;
            dSD[c] = fv[6 + c] - fv[9 + c];
            dS1v[c] = fv[9 + c];
            eSD[c] = ((fv[20 + c] - fv[23 + c]) - dSD[c]) * ih;
            eS1[c] = (fv[23 + c] - fv[9 + c]) * ih;
        }
        float rho = rho_old;
        if (fabsf(dr) > kLzMoved) {   // step-size control on the measured extrapolation error of the render and of d out / d r
            float e = 0.0f, ej = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                e = fmaxf(e, fabsf(fmaf(C0[c], pSD[c] - vSD[c], pS1[c] - vS1[c])) * iscale[c]);
                const float jc = fmaf(kd[c], dP[c], fmaf(C0[c], dSD[c], dS1v[c]));
                ej = fmaxf(ej, fabsf(fmaf(C0[c], pdSD[c] - dSD[c], pdS1[c] - dS1v[c])) / fmaxf(fabsf(jc), kLzJFloor * floor_));
            }
            rho = lazy_rho_next(rho, fabsf(dr), e, ej, tol_s, tol_s * (kLzTolJ / kLzTolS));
        }
        rho = fminf(fmaxf(rho, kLzRhoMin), kLzRhoMax);
        const float dmax = fmaxf(fmaxf(fabsf(dSD[0]) + fabsf(dS1v[0]), fabsf(dSD[1]) + fabsf(dS1v[1])), fabsf(dSD[2]) + fabsf(dS1v[2]));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float es = lazy_e_cap(eSD[c], dmax, fminf(fv[18], rho), fminf(fv[19], rho));
            eS1[c] = lazy_e_cap(eS1[c], dmax, fminf(fv[18], rho), fminf(fv[19], rho));
            eSD[c] = es;
        }
        *(uint32_t*)((char*)qs.plane[kLzRref] + o1) = as_u(rc_r);
        *(uint32_t*)((char*)qs.plane[kLzLoHi] + o1) = pack_h2(iv_round(fminf(fv[18], rho)), iv_round(fminf(fv[19], rho)));
        *(uint32_t*)((char*)qs.plane[kLzRho] + o1) = as_u(rho);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            *(uint32_t*)((char*)qs.plane[kLzP + c] + o1) = as_u(Pc[c]);
            *(uint32_t*)((char*)qs.plane[kLzSD + c] + o1) = as_u(vSD[c]);
            *(uint32_t*)((char*)qs.plane[kLzS1 + c] + o1) = as_u(vS1[c]);
            *(uint32_t*)((char*)qs.plane[kLzPk + c] + o1) = pack_h2(dP[c], A2[c]);
            *(uint32_t*)((char*)qs.plane[kLzSk + c] + o1) = pack_h2(gSD[c], gS1[c]);
            *(uint32_t*)((char*)qs.plane[kLzDk + c] + o1) = pack_h2(dSD[c], dS1v[c]);
            *(uint32_t*)((char*)qs.plane[kLzEk + c] + o1) = pack_h2(eSD[c], eS1[c]);
            *(uint32_t*)((char*)qs.plane[kLzD32 + c] + o1) = as_u(dSD[c]);
            *(uint32_t*)((char*)qs.plane[kLzD32 + 3 + c] + o1) = as_u(dS1v[c]);
            float rgb = fmaf(kd[c], Pc[c], fmaf(C0[c], vSD[c], vS1[c]));
            if (FOLD) {      // the folded planes of lazy_pstep_kernel<kFoldXY> (fold_xy_*: one definition for the fold kernel and this one)
                FoldXY f;
                fold_xy(rc[1 + c], Pc[c], vSD[c], vS1[c], dP[c], A2[c], gSD[c], gS1[c], dSD[c], dS1v[c], eSD[c], eS1[c], mref_code(mv), f);
                x2h[c] = f.X2; jx1[c] = f.JX1; jy1[c] = f.JY1; fx0[c] = f.X0; fy0[c] = f.Y0;
                *(uint32_t*)((char*)qs.fplane[kFxS + c] + of) = pack_h2(f.X1, f.Y1);
                *(uint32_t*)((char*)qs.fplane[kFxJ + c] + of) = pack_h2(f.JX0, f.JY0);
                rgb = fmaf(mv, f.Y0, f.X0);
            }
            if (!(FOLD && qs.no_pred)) stf(sp.pred_next, o3 + 4 * c, rgb);
            tot += rgb;
            if (FOLD && rgb_out != nullptr) rgb_out[c] = rgb;      // (the caller forms the pixel's share of the next iteration's statistics)
        }
        if (FOLD) {
            uint32_t xw[5];
            xy_pack(fx0, fy0, mref_code(mv), xw);
#pragma unroll
            for (int k = 0; k < 5; ++k) *(uint32_t*)((char*)qs.fplane[kFxXY + k] + of) = xw[k];
            *(uint32_t*)((char*)qs.fplane[kFxRref] + of) = as_u(rc_r);
            // the interval as the generic plane holds it (its half-precision words): both forms list a pixel in the same iteration, up to iv_pack's rounding
            const uint32_t lh = pack_h2(iv_round(fminf(fv[18], rho)), iv_round(fminf(fv[19], rho)));
            *(uint32_t*)((char*)qs.fplane[kFxLoHi] + of) = pack_lohi_x2(h2_lo(lh), h2_hi(lh), x2h[0]);
            *(uint32_t*)((char*)qs.fplane[kFxQ] + of) = pack_h2(x2h[1], x2h[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) *(uint32_t*)((char*)qs.fplane[kFxE + c] + of) = pack_h2(jx1[c], jy1[c]);
        }
    }
}

__global__ __launch_bounds__(64) void lazy_resample_kernel(const LazyStepArgs qs, const float* __restrict__ light, const Geom g, const RuleTable tab) {
    __shared__ float s_light[kNL + 1];
    __shared__ float4 s_ring[kMaxRings];
    __shared__ float2 s_saz[kMaxRings * kMaxAz];
    const JacBwdArgs& q = qs.j;
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    // one round trip for everything that depends on nothing: the stop flag, the counts, the committed state, the tables
    extern __shared__ int s_pref[];                                        // [nblk + 1]
    const int nblk = qs.n_sums - (int)gridDim.x;
    const float stopped = qs.state_old[b * kStateStride + kStStopped];
    const float* st = qs.state_new + b * kStateStride;                     // the state the step kernel has just committed
    const float ratio = st[kStRatio], gt_sum = st[kStGtSum], sel_f = st[kStSel];
    const float lt0 = light[(long)b * kNL + threadIdx.x], lt1 = threadIdx.x + 64 < kNL ? light[(long)b * kNL + threadIdx.x + 64] : 0.0f;
    static_assert(kNL <= 128 && kMaxRings <= 64 && kMaxRings * kMaxAz <= 256, "table loads below");
    float2 saz_v[(kMaxRings * kMaxAz + 63) / 64];
#pragma unroll
    for (int u = 0; u < (kMaxRings * kMaxAz + 63) / 64; ++u) {
        const int i = (int)threadIdx.x + 64 * u;
        saz_v[u] = (&tab.saz[0][0])[i < kMaxRings * kMaxAz ? i : 0];
    }
    const float4 ring_v = tab.sring[threadIdx.x < kMaxRings ? threadIdx.x : 0];
    // The image's work list is the concatenation of its workgroups' lists: every wave forms the exclusive prefix of the counts in LDS
    // (2 KB at 512 x 512; no second launch) and takes the items 8 blockIdx.x + 8 gridDim.x k .. + 7, eight lanes each -- lists are short
    // but bursty (a few pixels of one neighbourhood cross together): the items are spread evenly whatever their lists.
    const int per = (nblk + 63) / 64, first = (int)threadIdx.x * per;
    int mine = 0;
    for (int j0 = 0; j0 < per; j0 += 8) {                                  // eight counts in flight at once (the loop of one: eight round trips)
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int blk = first + j0 + u;
            c[u] = (j0 + u < per && blk < nblk) ? (int)qs.counts[(long)b * nblk + (blk < nblk ? blk : 0)] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int blk = first + j0 + u;
            if (j0 + u < per && blk < nblk) s_pref[blk] = c[u];       // the raw counts: a lane's run is scanned linearly by whoever lands in it
            mine += c[u];
        }
    }
    if (stopped > 0.5f) return;                                            // the step kernel skipped the image (uniform)
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if ((int)threadIdx.x >= d) incl += v;
    }
    const int T = __shfl(incl, 63);                                        // all the image's items (uniform)
    if (T == 0 || (long)blockIdx.x * 8 >= T) {
        if (threadIdx.x == 0) qs.block_sums[(long)b * qs.n_sums + nblk + blockIdx.x] = 0.0f;
        return;
    }
    s_light[threadIdx.x] = lt0 * kShNorm[threadIdx.x / 3];
    if (threadIdx.x + 64 < kNL) s_light[threadIdx.x + 64] = lt1 * kShNorm[(threadIdx.x + 64) / 3];
#pragma unroll
    for (int u = 0; u < (kMaxRings * kMaxAz + 63) / 64; ++u)
        if ((int)threadIdx.x + 64 * u < kMaxRings * kMaxAz) s_saz[threadIdx.x + 64 * u] = saz_v[u];
    if (threadIdx.x < kMaxRings) s_ring[threadIdx.x] = ring_v;
    StepPtrs sp{q.pa, q.pr, q.pm, q.pa, q.pr, q.pm, qs.pred_next};          // the parameters as WRITTEN by the step (a map the part does not move: as read)
    if (qs.rotate) {
        const bool wr1 = __builtin_amdgcn_readfirstlane((int)(sel_f > 0.5f)) != 0;
        if (qs.alt_a) sp.a = wr1 ? qs.alt_a : q.pa;
        if (qs.alt_r) sp.r = wr1 ? qs.alt_r : q.pr;
        if (qs.alt_m) sp.m = wr1 ? qs.alt_m : q.pm;
        sp.pred_next = qs.pred_buf[wr1 ? 1 : 0];
    }
    const uint16_t* lists = qs.lists + (long)b * nblk * kLazyBlockPixels;
    const int lane = threadIdx.x;                                          // one wave per list: eight lanes per pixel, eight pixels per pass
    float tot = 0.0f;
    __syncthreads();
    {
        float floor_;
        {
            const float rt = ratio;
            floor_ = 0.5f * gt_sum / (3.0f * (float)P) / (rt > 0.0f ? rt : 1.0f);
        }
        const float tol_k = qs.tol * kLzTolK, tol_s = qs.tol * kLzTolS;
        const int sub = lane & 7;
        const long BPl = (long)gridDim.y * P;
        // eight lanes per pixel, eight pixels per wave and pass (one sample per lane -- 40 lanes per pixel, one pixel per wave -- was
        // slower: every wave pays the prefix over the image's counts)
        for (int ib = blockIdx.x * 8; ib < T; ib += gridDim.x * 8) {
            const int item = ib + (lane >> 3);
            const bool item_ok = item < T;
            // the pixel's record, as the step kernel formed it: the clamped parameters it has just written, the distance to the old model
            const int it = item_ok ? item : ib;
            // the list the item is in: the lane whose run of counts contains it (one ballot per pixel slot on the inclusive scan the lanes
            // hold), then a linear scan of that run's raw counts in LDS
            int run = 0, rel = 0;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int it_s = __shfl(it, 8 * sl);                            // uniform
                const unsigned long long above = __ballot(incl > it_s);         // lanes whose runs end beyond the item
                const int rn = above ? (int)__builtin_ctzll(above) : 63;
                const int ex = __shfl(incl - mine, rn);                         // items before that run
                if ((lane >> 3) == sl) { run = rn; rel = it_s - ex; }
            }
            int lo_b = run * per;
            for (int j = 0; j + 1 < per; ++j) {
                const int cj = s_pref[lo_b < nblk ? lo_b : nblk - 1];
                if (rel < cj) break;
                rel -= cj; ++lo_b;
            }
            const int p = lo_b * kLazyBlockPixels + (int)lists[(long)lo_b * kLazyBlockPixels + rel];
            resample_walk_pixel<false>(qs, sp, s_light, s_ring, s_saz, g, tab, b, P, BPl, p, item_ok, sub, floor_, tol_k, tol_s, tot);
        }
    }
    tot = wave_sum_to_lane63(tot);
    if (lane == 63) qs.block_sums[(long)b * qs.n_sums + nblk + blockIdx.x] = tot;
}

// =================================================================================================
// refresh: walk the samples of the listed pixels, rebuild their models, patch their render
// =================================================================================================
// Mapping: LPI lanes per listed pixel, lane `sub` takes the azimuths sub, sub + LPI, ... of every ring, and the two halves of every
// packed register hold the SAME pixel at r and at r + dir h (the one-sided difference that gives the slopes): the serial chain of a
// refresh is nu_s samples instead of 2 x nu_s x nphi_s, which matters because this kernel runs at a few waves per CU.
struct LazyRefreshArgs {
    const float *a, *r, *m, *n, *dcache;
    uint32_t* state;
    float* out;
    uint32_t* jac16;
    const float* stats;       // nullable; with it the parity floor is half the image's mean radiance, 0.5 (sum gt / ratio) / (3 P)
    float* block_sums;        // nullable [B][n_sums]: slot n_fwd + blockIdx.x
    const uint32_t* counts;
    const uint16_t* lists;
    int clamp, force, n_sums, n_fwd, nblk;
    float floor, tol;
    int jac32;                // as LazyFwdArgs
};

constexpr int kLazyMaxBlocks = 8192;   // forward workgroups per image whose counts fit the LDS prefix (512 x 8192 pixels = 2048 x 2048)
template <int LPI>
__global__ __launch_bounds__(kBlock, 2) void lazy_refresh_kernel(const LazyRefreshArgs q, const float* __restrict__ light, const Geom g,
                                                                 const RuleTable tab) {
    __shared__ float2 s_saz[kMaxRings * kMaxAz];
    __shared__ int s_pref[kLazyMaxBlocks + 1];      // exclusive prefix of the image's per-workgroup counts: the work list is their concatenation
    __shared__ int s_wave[4];
    __shared__ float s_sum[4];
    if (threadIdx.x < kMaxRings * kMaxAz) s_saz[threadIdx.x] = (&tab.saz[0][0])[threadIdx.x];
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const long BP = (long)gridDim.y * P;
    const bool stopped = q.stats && img_stopped(q.stats, b);
    // every workgroup scans all counts of its image (2 KB at 512 x 512): no second launch, no global prefix array to chase
    const int per = (q.nblk + kBlock - 1) / kBlock, first = threadIdx.x * per;
    int mine = 0;
    for (int j = 0; j < per; ++j) {
        const int blk = first + j;
        const int c = (blk < q.nblk && !stopped) ? (int)q.counts[(long)b * q.nblk + blk] : 0;
        if (blk < q.nblk) s_pref[blk] = mine;       // exclusive within the thread's run; the thread offset is added below
        mine += c;
    }
    int incl = mine;                                 // inclusive scan of the per-thread totals: wave (DPP-free, shuffles), then the 4 waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if ((threadIdx.x & 63) >= d) incl += v;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) woff += w < (int)(threadIdx.x >> 6) ? s_wave[w] : 0;
    const int excl = woff + incl - mine;
    for (int j = 0; j < per; ++j)
        if (first + j < q.nblk) s_pref[first + j] += excl;
    const int T = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
    if (threadIdx.x == 0) s_pref[q.nblk] = T;
    __syncthreads();
    float tot = 0.0f;
    constexpr int kItems = kBlock / LPI;
    if ((long)blockIdx.x * kItems < T) {
        float floor_ = q.floor;
        if (q.stats) {
            const float ratio = q.stats[b * kStatsStride + kStRatio];
            floor_ = 0.5f * q.stats[b * kStatsStride + kStGtSum] / (3.0f * (float)P) / (ratio > 0.0f ? ratio : 1.0f);
        }
        const float tol_k = q.tol * kLzTolK, tol_s = q.tol * kLzTolS;
        LightRegs lr;
        load_light_regs(lr, light + (long)b * kNL);
        const int sub = threadIdx.x % LPI, slot = threadIdx.x / LPI;
        for (int base = blockIdx.x * kItems; base < T; base += gridDim.x * kItems) {   // chunks of kItems items, strided over the workgroups
            const bool active = base + slot < T;
            const int it = active ? base + slot : T - 1;
            int lo_b = 0, hi_b = q.nblk;            // largest blk with s_pref[blk] <= it
            while (hi_b - lo_b > 1) {
                const int mid = (lo_b + hi_b) >> 1;
                if (s_pref[mid] <= it) lo_b = mid; else hi_b = mid;
            }
            const int blk = lo_b;
            const int p = blk * kLazyBlockPixels + (int)q.lists[((long)b * q.nblk + blk) * kLazyBlockPixels + (it - s_pref[blk])];
            const long i = (long)b * P + p;
            Pixel px;
            load_pixel(px, q.a, q.r, q.m, q.n, i, i, p, p, g, q.clamp != 0);
            const float rc = px.r.x, mv = px.m.x;
            // what the old model predicts at the new roughness: the parity scale of this refresh, and the measured extrapolation error
            const bool has_old = !q.force;
            float rho = kLzRhoInit, dr = 0.0f, pSD[3] = {0, 0, 0}, pS1[3] = {0, 0, 0}, pdSD[3] = {0, 0, 0}, pdS1[3] = {0, 0, 0};
            if (has_old) {
                dr = rc - as_f(q.state[kLzRref * BP + i]);
                rho = as_f(q.state[kLzRho * BP + i]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t sk = q.state[(kLzSk + c) * BP + i];
                    pSD[c] = fmaf(h2_lo(sk), dr, as_f(q.state[(kLzSD + c) * BP + i]));
                    pS1[c] = fmaf(h2_hi(sk), dr, as_f(q.state[(kLzS1 + c) * BP + i]));
                    const uint32_t ek = q.state[(kLzEk + c) * BP + i];
                    pdSD[c] = fmaf(h2_lo(ek), dr, as_f(q.state[(kLzD32 + c) * BP + i]));
                    pdS1[c] = fmaf(h2_hi(ek), dr, as_f(q.state[(kLzD32 + 3 + c) * BP + i]));
                }
            }
            float C0[3], kd[3], Pc[3], dP[3], A2[3], iscale[3], ijscale[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float A0 = q.dcache[c * BP + i], A1 = q.dcache[(3 + c) * BP + i];
                A2[c] = q.dcache[(6 + c) * BP + i];
                Pc[c] = fmaf(fmaf(A2[c], rc, A1), rc, A0);
                dP[c] = fmaf(2.0f * rc, A2[c], A1);
                kd[c] = px.a[c].x * (1.0f - mv);
                C0[c] = fmaf(mv, px.a[c].x, (1.0f - mv) * 0.04f);
                const float pred = has_old ? fabsf(fmaf(kd[c], Pc[c], fmaf(C0[c], pSD[c], pS1[c]))) : 0.0f;
                iscale[c] = 1.0f / fmaxf(pred, floor_);
                const float jpred = has_old ? fabsf(fmaf(kd[c], dP[c], fmaf(C0[c], pdSD[c], pdS1[c]))) : 0.0f;
                ijscale[c] = (kLzTolK / kLzTolKJ) / fmaxf(jpred, kLzJFloor * floor_);
            }
            float dir = (has_old && dr < 0.0f) ? -1.0f : 1.0f;
            if (rc + dir * kLzH > 1.0f || rc + dir * kLzH < 0.07f) dir = -dir;
            pixel_set_r(px, f2{rc, rc + dir * kLzH});

            f2 S0[3], S1[3], dS0[3], dS1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) S0[c] = S1[c] = dS0[c] = dS1[c] = f2{0.0f, 0.0f};
            float klo = 1e30f, khi = 1e30f;
            const f2 four_over_r = 4.0f * rcp(px.r);
            const f2 cv = px.dk_dr * px.g1v * (1.0f - px.NoV);
            const float g1l0 = rcp(px.kpe.x), r3x4 = 4.0f * rc * rc * rc;
            for (int k = 0; k < tab.nu_s; ++k) {
                const float4 rg = tab.sring[k];
                SpecRing R;
                spec_ring<true>(px, rg, R);
                const f2 lam0 = vfma(four_over_r, vfma(R.q * R.idq, -2.0f * rg.y, 1.0f), -cv);
                // d sin^2 theta_h / dr = -d cos^2 theta_h / dr = 4 r^3 u0 q cos^2 theta_h   (mi_specular_sampler :232-233)
                const float gq = r3x4 * rg.x * R.q.x * (R.q.x * rg.y);
                const float stp = 0.5f * gq * rcp(R.st.x), ctp = -0.5f * gq * rcp(R.ct.x);
                for (int j = sub; j < tab.nphi_s; j += LPI) {
                    const float2 az = s_saz[k * kMaxAz + j];
                    SpecSample sm;
                    spec_sample(px, R, az, sm);
                    const f2 wgt = (R.ringw * sm.g1l) * (sm.NoL * sm.dpos);
                    f2 L[3];
                    sh_radiance(lr, sm.wi, L);
                    const f2 wx = wgt * sm.x5;
                    const f2 lam = vfma(px.dk_dr * sm.g1l, sm.NoL - 1.0f, lam0);
                    const f2 wl = wgt * lam, wlx = wl * sm.x5;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        S0[c] = vfma(wgt, L[c], S0[c]);
                        S1[c] = vfma(wx, L[c], S1[c]);
                        dS0[c] = vfma(wl, L[c], dS0[c]);
                        dS1[c] = vfma(wlx, L[c], dS1[c]);
                    }
                    // where this sample's clamped variables n.wi and wo.h cross zero, to first order in r
                    const float dp = fmaf(stp, fmaf(az.x, px.vx.x, az.y * px.vy.x), ctp * px.vz.x);
                    const float wlzp = 2.0f * fmaf(dp, R.ct.x, sm.d.x * ctp);
                    const float lamk1 = fabsf(lam0.x - px.dk_dr.x * g1l0), lamk2 = fabsf(lam.x);
                    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float aL = fabsf(L[c].x), sc1 = fmaxf(iscale[c], lamk1 * ijscale[c]), sc2 = fmaxf(iscale[c], lamk2 * ijscale[c]);
                        m1 = fmaxf(m1, fmaf(1.0f - C0[c], sm.x5.x, C0[c]) * aL * sc1);
                        m2 = fmaxf(m2, aL * sc2);
                    }
                    lazy_kink(sm.wlz.x, wlzp, R.ringw.x * g1l0 * sm.dpos.x * m1 * fabsf(wlzp), tol_k, klo, khi);
                    lazy_kink(sm.d.x, dp, R.ringw.x * sm.g1l.x * sm.NoL.x * m2 * fabsf(dp), tol_k, klo, khi);
                }
            }
            klo = lane_group_min<LPI>(klo);
            khi = lane_group_min<LPI>(khi);
            float vSD[3], vS1[3], gSD[3], gS1[3], dSD[3], dS1v[3], eSD[3], eS1[3];
            const float ih = dir * (1.0f / kLzH);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float s0x = lane_group_sum<LPI>(S0[c].x), s0y = lane_group_sum<LPI>(S0[c].y);
                const float s1x = lane_group_sum<LPI>(S1[c].x), s1y = lane_group_sum<LPI>(S1[c].y);
                const float d0 = lane_group_sum<LPI>(dS0[c].x), d1 = lane_group_sum<LPI>(dS1[c].x);
                const float d0y = lane_group_sum<LPI>(dS0[c].y), d1y = lane_group_sum<LPI>(dS1[c].y);
                vSD[c] = s0x - s1x;
                vS1[c] = s1x;
                gSD[c] = ((s0y - s1y) - vSD[c]) * ih;
                gS1[c] = (s1y - s1x) * ih;
                dSD[c] = d0 - d1;
                dS1v[c] = d1;
                eSD[c] = ((d0y - d1y) - dSD[c]) * ih;
                eS1[c] = (d1y - d1) * ih;
            }
            if (active && sub == 0) {
                if (has_old && fabsf(dr) > kLzMoved) {   // step-size control on the measured extrapolation error of the render and of d out / d r
                    float e = 0.0f, ej = 0.0f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        e = fmaxf(e, fabsf(fmaf(C0[c], pSD[c] - vSD[c], pS1[c] - vS1[c])) * iscale[c]);
                        const float jc = fmaf(kd[c], dP[c], fmaf(C0[c], dSD[c], dS1v[c]));
                        ej = fmaxf(ej, fabsf(fmaf(C0[c], pdSD[c] - dSD[c], pdS1[c] - dS1v[c])) / fmaxf(fabsf(jc), kLzJFloor * floor_));
                    }
                    rho = lazy_rho_next(rho, fabsf(dr), e, ej, tol_s, tol_s * (kLzTolJ / kLzTolS));
                }
                rho = fminf(fmaxf(rho, kLzRhoMin), kLzRhoMax);
                const float dmax = fmaxf(fmaxf(fabsf(dSD[0]) + fabsf(dS1v[0]), fabsf(dSD[1]) + fabsf(dS1v[1])), fabsf(dSD[2]) + fabsf(dS1v[2]));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float es = lazy_e_cap(eSD[c], dmax, fminf(klo, rho), fminf(khi, rho));
                    eS1[c] = lazy_e_cap(eS1[c], dmax, fminf(klo, rho), fminf(khi, rho));
                    eSD[c] = es;
                }
                uint32_t* S = q.state;
                S[kLzRref * BP + i] = as_u(rc);
                S[kLzLoHi * BP + i] = pack_h2(iv_round(fminf(klo, rho)), iv_round(fminf(khi, rho)));   // rounded DOWN to the 256 lengths of iv_pack (exact in half precision)
                S[kLzRho * BP + i] = as_u(rho);
                float jr[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    S[(kLzP + c) * BP + i] = as_u(Pc[c]);
                    S[(kLzSD + c) * BP + i] = as_u(vSD[c]);
                    S[(kLzS1 + c) * BP + i] = as_u(vS1[c]);
                    S[(kLzPk + c) * BP + i] = pack_h2(dP[c], A2[c]);
                    S[(kLzSk + c) * BP + i] = pack_h2(gSD[c], gS1[c]);
                    S[(kLzDk + c) * BP + i] = pack_h2(dSD[c], dS1v[c]);
                    S[(kLzD32 + c) * BP + i] = as_u(dSD[c]);
                    S[(kLzD32 + 3 + c) * BP + i] = as_u(dS1v[c]);
                    S[(kLzEk + c) * BP + i] = pack_h2(eSD[c], eS1[c]);
                    const float rgb = fmaf(kd[c], Pc[c], fmaf(C0[c], vSD[c], vS1[c]));
                    q.out[i * 3 + c] = rgb;
                    tot += rgb;
                    jr[c] = fmaf(kd[c], dP[c], fmaf(C0[c], dSD[c], dS1v[c]));
                    if (q.jac16 && q.jac32) { q.jac16[c * BP + i] = as_u(Pc[c]); q.jac16[(3 + c) * BP + i] = as_u(vSD[c]); q.jac16[(6 + c) * BP + i] = as_u(jr[c]); }
                    else if (q.jac16) q.jac16[c * BP + i] = pack_h2(Pc[c], vSD[c]);
                }
                if (q.jac16 && !q.jac32) {
                    q.jac16[3 * BP + i] = pack_h2(jr[0], jr[1]);
                    q.jac16[4 * BP + i] = pack_h2(jr[2], 0.0f);
                }
            }
        }
    }
    if (q.block_sums) {
        tot = wave_sum_to_lane63(tot);
        if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) q.block_sums[(long)b * q.n_sums + q.n_fwd + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}

// ---- test / inspection helpers -------------------------------------------------------------------
// state -> the oracle's layout [B*P][28]: r_ref, lo, hi, rho, SD, S1, gSD, gS1, dSD, dS1, eSD, eS1; refreshed[B*P] (nullable) = 1 for the pixels
// of the last call's work lists
__global__ __launch_bounds__(kBlock) void lazy_unpack_kernel(const uint32_t* __restrict__ S, float* __restrict__ st, long BP) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= BP) return;
    float* o = st + i * 28;
    const uint32_t lh = S[kLzLoHi * BP + i];
    o[0] = as_f(S[kLzRref * BP + i]); o[1] = h2_lo(lh); o[2] = h2_hi(lh); o[3] = as_f(S[kLzRho * BP + i]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t sk = S[(kLzSk + c) * BP + i], dk = S[(kLzDk + c) * BP + i], ek = S[(kLzEk + c) * BP + i];
        o[22 + c] = h2_lo(ek); o[25 + c] = h2_hi(ek);
        o[4 + c] = as_f(S[(kLzSD + c) * BP + i]); o[7 + c] = as_f(S[(kLzS1 + c) * BP + i]);
        o[10 + c] = h2_lo(sk); o[13 + c] = h2_hi(sk); o[16 + c] = h2_lo(dk); o[19 + c] = h2_hi(dk);
    }
}
__global__ __launch_bounds__(kBlock) void lazy_refreshed_kernel(const uint32_t* __restrict__ counts, const uint16_t* __restrict__ lists,
                                                                int* __restrict__ refreshed, int P, int nblk) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const int n = (int)counts[(long)b * nblk + blk];
    for (int k = threadIdx.x; k < kLazyBlockPixels; k += kBlock) {
        const int p = blk * kLazyBlockPixels + k;
        if (p < P) refreshed[(long)b * P + p] = 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += kBlock) refreshed[(long)b * P + blk * kLazyBlockPixels + lists[((long)b * nblk + blk) * kLazyBlockPixels + k]] = 1;
}
__global__ __launch_bounds__(kBlock) void jac16_unpack_kernel(const uint32_t* __restrict__ j16, float* __restrict__ jac, long BP) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= BP) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t u = j16[c * BP + i];
        jac[c * BP + i] = h2_lo(u);
        jac[(3 + c) * BP + i] = h2_hi(u);
    }
    const uint32_t u3 = j16[3 * BP + i], u4 = j16[4 * BP + i];
    jac[6 * BP + i] = h2_lo(u3); jac[7 * BP + i] = h2_hi(u3); jac[8 * BP + i] = h2_lo(u4);
}

}  // namespace matpbr
