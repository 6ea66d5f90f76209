// matpbr_kernels.hip -- gfx950 kernels + the C ABI of include/matpbr.h.
//
// Image kernels: one thread per pixel, 256-thread workgroups (4 wave64), blockIdx.y = image of the batch.
//   * maps a/r/m/n are read once per pixel with 12-byte / 4-byte per-lane loads that tile the row-major
//     HWC arrays without gaps (every fetched byte is used); rgb / gradients are written the same way;
//   * the 25x3 SH coefficients of the image are staged once per workgroup in LDS, pre-multiplied by the
//     basis normalisation, so the per-sample radiance is 72 FMAs on raw polynomials;
//   * the deterministic sample set arrives in the kernel-argument segment (scalar loads, no VGPRs);
//   * the light gradient is reduced per wave with DPP, per workgroup through LDS, written as one
//     [75]-float partial per workgroup and summed by a second tiny kernel: no atomics, bit-reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/matpbr.h"
#include "matpbr_device.hpp"

using namespace matpbr;

namespace {

constexpr int kBlock = 256;
constexpr int kNL = kNSH * 3;  // 75 light scalars per image

// Deterministic BSDF-sample set: n = spp/2 points per lobe,
// u0_i = (i+.5)/n, u1_i = vdC2(i) + .5/m.  Values are computed on the host in double.  The table travels in the
// kernel-argument segment: wave-uniform scalar loads into SGPRs, which VALU instructions broadcast for free.
constexpr int kMaxHalf = MATPBR_MAX_SPP / 2;
struct SampleTable {
    float4 diff[kMaxHalf];  // cosine-weighted sample: local direction (x, y, z), unused
    float4 spec[kMaxHalf];  // GGX half-vector sample: u0, cos(phi), sin(phi), 1-u0
};

double vdc2(uint32_t i) {
    i = (i << 16) | (i >> 16);
    i = ((i & 0x55555555u) << 1) | ((i & 0xAAAAAAAAu) >> 1);
    i = ((i & 0x33333333u) << 2) | ((i & 0xCCCCCCCCu) >> 2);
    i = ((i & 0x0F0F0F0Fu) << 4) | ((i & 0xF0F0F0F0u) >> 4);
    i = ((i & 0x00FF00FFu) << 8) | ((i & 0xFF00FF00u) >> 8);
    return (double)i * 2.3283064365386963e-10;
}

void fill_sample_table(int spp, SampleTable& t) {
    const int n = spp / 2;
    int m = 1;
    while (m < n) m <<= 1;
    std::memset(&t, 0, sizeof(t));
    for (int i = 0; i < n; ++i) {
        double u0 = (i + 0.5) / n, u1 = vdc2((uint32_t)i) + 0.5 / m;
        double phi = 2.0 * M_PI * u1;
        double st = std::sqrt(u0), ct = std::sqrt(1.0 - u0);  // theta = asin(sqrt(u0))  (mi_plugin.py:265)
        t.diff[i] = make_float4((float)(st * std::cos(phi)), (float)(st * std::sin(phi)), (float)ct, 0.0f);
        t.spec[i] = make_float4((float)u0, (float)std::cos(phi), (float)std::sin(phi), (float)(1.0 - u0));
    }
}

struct Geom {
    int H, W, half;  // half = spp/2 samples per lobe
    float inv_f, cx, cy, inv_spp;
};

// ---- wave64 sum with DPP: row_shr 1,2,4,8 -> row totals in lane 15 of each row of 16,
// row_bcast:15 and row_bcast:31 fold the four rows; the total lands in lane 63.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
    int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111>(v);  // row_shr:1
    v = dpp_add<0x112>(v);  // row_shr:2
    v = dpp_add<0x114>(v);  // row_shr:4
    v = dpp_add<0x118>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return v;
}

// Per-lane set-up shared by forward and backward: TWO pixels (flattened indices 2q, 2q+1) in the halves of every f2.
struct Pixel {
    PixelConst<f2> pc;
    f2 n[3], wo[3], s[3], t[3];
    f2 vx, vy;   // view direction in the shading frame (its z component is pc.NoV_raw)
    f2 inv_len;  // 1/|n| of the stored normal
};

// CLAMP: the maps are the optimiser's raw parameters and the render uses clamp(a,0,1), clamp(r,.07,1), clamp(m,0,1)
// (inverse_img_w_mi.py:371-377); the raw values are returned so the backward pass can gate the gradient like torch.clamp.
struct RawParams { f2 a[3], r, m; };
__device__ __forceinline__ f2 clamp2(f2 x, float lo, float hi) { return vmin(vmax(x, lo), hi); }

template <bool CLAMP>
__device__ __forceinline__ void load_pixel(Pixel& px, const float* __restrict__ a, const float* __restrict__ r,
                                           const float* __restrict__ m, const float* __restrict__ n, long i0, long i1, int p0, int p1,
                                           const Geom& g, RawParams* raw = nullptr) {
    f2 av[3], nv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        av[c] = f2{a[i0 * 3 + c], a[i1 * 3 + c]};
        nv[c] = f2{n[i0 * 3 + c], n[i1 * 3 + c]};
    }
    f2 rv = f2{r[i0], r[i1]}, mv = f2{m[i0], m[i1]};
    if (CLAMP) {
        if (raw) {
#pragma unroll
            for (int c = 0; c < 3; ++c) raw->a[c] = av[c];
            raw->r = rv; raw->m = mv;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) av[c] = clamp2(av[c], 0.0f, 1.0f);
        rv = clamp2(rv, 0.07f, 1.0f);
        mv = clamp2(mv, 0.0f, 1.0f);
    }
    // shading normal = normalize(n map); the geometric normals and MaterialNet's are unit already
    px.inv_len = rsq(vmax(dot3v(nv, nv), 1e-30f));
#pragma unroll
    for (int c = 0; c < 3; ++c) px.n[c] = nv[c] * px.inv_len;
    // view direction of pixel (i,j): wo = -p/|p|, p = ((j-cx)/f, -(i-cy)/f, -1)   
    f2 fi = f2{(float)(p0 / g.W), (float)(p1 / g.W)}, fj = f2{(float)(p0 % g.W), (float)(p1 % g.W)};
    f2 x = (g.cx - fj) * g.inv_f, y = (fi - g.cy) * g.inv_f;
    f2 il = rsq(vfma(x, x, vfma(y, y, 1.0f)));
    px.wo[0] = x * il; px.wo[1] = y * il; px.wo[2] = il;
    frame(px.n, px.s, px.t);
    px.vx = dot3v(px.s, px.wo);
    px.vy = dot3v(px.t, px.wo);
    pixel_const(px.pc, av, rv, mv, dot3v(px.n, px.wo));
}

// One sample of the deterministic estimator for both pixels of the lane: direction wi, cosines, GGX denominator
// and, for the normal gradient, the tangential (shading-frame x,y) components of wi and of the half vector h.
struct Sample {
    f2 wi[3];
    f2 lwx, lwy, lhx, lhy;
    f2 NoL_raw, NoH, VoH, den;
    f2 nh_gate;  // > 0 where n.h > 0
};

template <bool WANT_H>
__device__ __forceinline__ void diffuse_sample(const Pixel& px, float lx, float ly, float lz, Sample& sm) {
    // mi_diffuse_sampler (mi_plugin.py:255-281): local (sin t cos p, sin t sin p, cos t) -> Frame3f(n).to_world
    to_world(px.s, px.t, px.n, lx, ly, lz, sm.wi);
    sm.NoL_raw = f2{lz, lz};  // n.wi for an orthonormal frame
    f2 wiwo = vfma(px.pc.NoV_raw, lz, vfma(px.vy, ly, px.vx * lx));
    f2 il = rsq(vmax(vfma(wiwo, 2.0f, 2.0f), 1e-30f));  // 1/|wi+wo|
    sm.VoH = vmax((1.0f + wiwo) * il, 0.0f);
    f2 nh = (px.pc.NoV_raw + lz) * il;
    sm.NoH = vmax(nh, 0.0f);
    sm.nh_gate = nh;
    // 1 - NoH^2 = h_x^2 + h_y^2 in the shading frame: no cancellation when the sample lands on the GGX peak
    f2 hx = (px.vx + lx) * il, hy = (px.vy + ly) * il;
    f2 sin2 = sel_pos(nh, vmin(vfma(hx, hx, hy * hy), 1.0f), 1.0f);
    sm.den = ggx_den_stable(px.pc, sin2);
    if (WANT_H) {
        sm.lwx = f2{lx, lx}; sm.lwy = f2{ly, ly};
        sm.lhx = hx; sm.lhy = hy;
    }
}

template <bool WANT_H>
__device__ __forceinline__ void specular_sample(const Pixel& px, float u0, float cphi, float sphi, float omu0, Sample& sm) {
    // mi_specular_sampler (mi_plugin.py:217-253): cos^2 t_h = (1-u0)/(u0(alpha2-1)+1), wi = reflect(wo, wh)
    f2 q = rcp(vfma(px.pc.am1, u0, 1.0f));
    f2 cos2 = vmax(q * omu0, 0.0f);
    f2 sin2 = vmax((px.pc.alpha2 * u0) * q, 0.0f);  // 1 - cos2 without cancellation
    f2 ct = fsqrt(cos2), st = fsqrt(sin2);
    const f2 whx = st * cphi, why = st * sphi;
    f2 d = vfma(ct, px.pc.NoV_raw, vfma(why, px.vy, whx * px.vx));  // wo.wh in the shading frame
    f2 d2 = 2.0f * d;
    f2 wlx = vfma(d2, whx, -px.vx), wly = vfma(d2, why, -px.vy), wlz = vfma(d2, ct, -px.pc.NoV_raw);  // wi = 2(wo.wh)wh - wo
    to_world(px.s, px.t, px.n, wlx, wly, wlz, sm.wi);
    sm.NoL_raw = wlz;       // n.wi
    sm.VoH = vabs(d);       // wo.h with h = sign(d) wh
    sm.NoH = sel_pos(d, ct, 0.0f);
    sm.nh_gate = sel_pos(d, ct, -1.0f);
    sm.den = sel_pos(d, ggx_den_stable(px.pc, sin2), 1.0f + 1e-6f);
    if (WANT_H) {
        f2 sg = sel_pos(d, 1.0f, -1.0f);
        sm.lhx = sg * whx; sm.lhy = sg * why;
        sm.lwx = wlx; sm.lwy = wly;
    }
}

// ---- SH coefficients: 75 wave-uniform scalars kept in 38 VGPR pairs ------------------------------------------
// A packed FMA needs the scalar c'[k][c] in both halves of a 64-bit operand.  hipcc materialises such a splat with
// a v_mov per use (VGPR) or spills the SGPR file (75 live scalars + sample table > 102 SGPRs), so the broadcast
// is spelled out: VOP3P op_sel/op_sel_hi pick the low or the high half of a register pair for BOTH lanes of the
// packed operation, letting one pair carry two different coefficients at zero extra instructions.
constexpr int kNPairs = (kNL + 1) / 2;
struct LightRegs { f2 c[kNPairs]; };

// c'[k][c] = coefficient * basis normalisation, so that the per-sample radiance is 72 FMAs on raw polynomials
__device__ __forceinline__ void load_light_regs(LightRegs& lr, const float* __restrict__ light) {
#pragma unroll
    for (int j = 0; j < kNPairs; ++j) {
        const int q0 = 2 * j, q1 = 2 * j + 1 < kNL ? 2 * j + 1 : 2 * j;
        lr.c[j] = f2{light[q0] * kShNorm[q0 / 3], light[q1] * kShNorm[q1 / 3]};
        asm volatile("" : "+v"(lr.c[j]));  // pin in VGPRs for the whole kernel
    }
}
template <int Q>
__device__ __forceinline__ void fma_bcast(f2& acc, f2 b, const LightRegs& lr) {  // acc += b * c'_Q (both halves)
    if (Q & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(b), "v"(lr.c[Q >> 1]));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(b), "v"(lr.c[Q >> 1]));
}
// out = b * c'_Q1 + c'_Q0 : opens the sum with the constant (k = 0) term at no extra instruction
template <int Q1, int Q0>
__device__ __forceinline__ f2 fma_bcast_init(f2 b, const LightRegs& lr) {
    f2 out;
    constexpr int s1 = Q1 & 1, s0 = Q0 & 1;
    if (s1 == 0 && s0 == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(b), "v"(lr.c[Q1 >> 1]), "v"(lr.c[Q0 >> 1]));
    if (s1 == 0 && s0 == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(out) : "v"(b), "v"(lr.c[Q1 >> 1]), "v"(lr.c[Q0 >> 1]));
    if (s1 == 1 && s0 == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(out) : "v"(b), "v"(lr.c[Q1 >> 1]), "v"(lr.c[Q0 >> 1]));
    if (s1 == 1 && s0 == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(out) : "v"(b), "v"(lr.c[Q1 >> 1]), "v"(lr.c[Q0 >> 1]));
    return out;
}
template <int K>
__device__ __forceinline__ void sh_term(const LightRegs& lr, f2 Bk, f2 L[3]) {
    if (K == 0) {
        // B_0 = 1: its term is the addend of the k = 1 FMA below
    } else if (K == 1) {
        L[0] = fma_bcast_init<3, 0>(Bk, lr); L[1] = fma_bcast_init<4, 1>(Bk, lr); L[2] = fma_bcast_init<5, 2>(Bk, lr);
    } else {
        fma_bcast<3 * K>(L[0], Bk, lr); fma_bcast<3 * K + 1>(L[1], Bk, lr); fma_bcast<3 * K + 2>(L[2], Bk, lr);
    }
}

// Order-4 SH basis polynomials of wi (both pixels), each consumed the moment it is produced by `use.template operator()<k>(B_k)`:
// only the ~14 shared monomials stay live, never 25 packed basis values.
template <class F>
__device__ __forceinline__ void sh_stream(const f2 w[3], F&& use) {
    const f2 X = -w[2], Y = w[0], Z = w[1];
    use.template operator()<0>(f2{1.0f, 1.0f});
    use.template operator()<1>(Y); use.template operator()<2>(Z); use.template operator()<3>(X);
    const f2 z2 = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z, y2 = Y * Y;
    const f2 d = vfma(X, X, -y2);
    use.template operator()<4>(xy); use.template operator()<5>(yz); use.template operator()<6>(vfma(z2, 3.0f, -1.0f));
    use.template operator()<7>(xz); use.template operator()<8>(d);
    const f2 t5 = vfma(z2, 5.0f, -1.0f);
    const f2 s3 = Y * vfma(3.0f * X, X, -y2), c3 = X * vfma(X, X, -3.0f * y2);
    use.template operator()<9>(s3); use.template operator()<10>(xy * Z); use.template operator()<11>(Y * t5);
    use.template operator()<12>(Z * (t5 - 2.0f)); use.template operator()<13>(X * t5); use.template operator()<14>(d * Z);
    use.template operator()<15>(c3);
    const f2 t7 = vfma(z2, 7.0f, -1.0f), t73 = t7 - 2.0f;
    use.template operator()<16>(xy * d); use.template operator()<17>(s3 * Z); use.template operator()<18>(xy * t7);
    use.template operator()<19>(yz * t73); use.template operator()<20>(vfma(vfma(z2, 35.0f, -30.0f), z2, 3.0f));
    use.template operator()<21>(xz * t73); use.template operator()<22>(d * t7); use.template operator()<23>(c3 * Z);
    use.template operator()<24>(vfma(d, d, -4.0f * (xy * xy)));
}

// radiance of the SH light for the lane's two pixels: L[c] = sum_k c'[k][c] B_k(wi)
struct RadianceUse {
    const LightRegs& lr;
    f2* L;
    template <int K> __device__ __forceinline__ void operator()(f2 Bk) { sh_term<K>(lr, Bk, L); }
};
__device__ __forceinline__ void sh_radiance(const LightRegs& lr, const f2 wi[3], f2 L[3]) { sh_stream(wi, RadianceUse{lr, L}); }

// =================================================================================================
// forward
// =================================================================================================
__device__ __forceinline__ void fwd_accumulate(const Pixel& px, const Sample& sm, const LightRegs& lr, f2 acc[3]) {
    BrdfState<f2> st;
    f2 f[3], pdf;
    brdf_core(px.pc, sm.NoL_raw, sm.NoH, sm.VoH, sm.den, st, f, pdf);
    // sample_brdf weight (mi_plugin.py:1335-1339): f/(pdf+1e-6) where pdf > 1e-6
    f2 ip = sel_pos(pdf - 1e-6f, rcp(pdf + 1e-6f), 0.0f);
    f2 L[3];
    sh_radiance(lr, sm.wi, L);
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = vfma(f[c] * ip, L[c], acc[c]);
}

// forward declaration of the statistics layout used by the fused optimisation step (defined with the loss kernels)
constexpr int kStatsStride = 16;
enum { kStRatio = 0, kStMse, kStL1, kStSr, kStLa, kStLr, kStLm, kStLoss, kStImproved, kStBest, kStEsCounter, kStEsBest, kStEsHas,
       kStStopped, kStIters, kStGtSum };

// SUMS (fused optimisation step): additionally writes the workgroup's sum of the rendered rgb (for mean(pred), :388) and
// skips images whose on-device EarlyStopping has fired.
template <bool CLAMP, bool SUMS>
__global__ __launch_bounds__(kBlock, 2) void shade_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                              const float* __restrict__ m, const float* __restrict__ n,
                                                              const float* __restrict__ light, float* __restrict__ out,
                                                              const Geom g, const SampleTable tab, const float* __restrict__ stats,
                                                              float* __restrict__ block_sums) {
    __shared__ float s_sum[4];
    const int b = blockIdx.y;
    if (SUMS && stats[b * kStatsStride + kStStopped] > 0.5f) return;
    const float* __restrict__ cp = light + (long)b * kNL;
    const int P = g.H * g.W;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    if (!SUMS && q0 >= P) return;
    const bool act0 = q0 < P, two = q0 + 1 < P;
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    Pixel px;
    load_pixel<CLAMP>(px, a, r, m, n, i0, i1, p0, p1, g);
    LightRegs lr;
    load_light_regs(lr, cp);

    f2 acc[3] = {f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}};
    // the table entry of sample s+1 is fetched (scalar load) while sample s is evaluated
    float4 t = tab.diff[0];
    for (int s = 0; s < g.half; ++s) {
        Sample sm;
        const float4 tn = s + 1 < g.half ? tab.diff[s + 1] : tab.spec[0];
        diffuse_sample<false>(px, t.x, t.y, t.z, sm);
        fwd_accumulate(px, sm, lr, acc);
        t = tn;
    }
    for (int s = 0; s < g.half; ++s) {
        Sample sm;
        const float4 tn = tab.spec[s + 1 < g.half ? s + 1 : s];
        specular_sample<false>(px, t.x, t.y, t.z, t.w, sm);
        fwd_accumulate(px, sm, lr, acc);
        t = tn;
    }
    float tot = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f2 v = acc[c] * g.inv_spp;
        if (act0) out[i0 * 3 + c] = v.x;
        if (two) out[i1 * 3 + c] = v.y;
        if (SUMS) tot += (act0 ? v.x : 0.0f) + (two ? v.y : 0.0f);
    }
    if (SUMS) {
        tot = wave_sum_to_lane63(tot);
        if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) block_sums[(long)b * gridDim.x + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}

// =================================================================================================
// precomputed radiance transfer (forward-only relighting, render_final.py:148-203,300-418)
// The render is linear in the light: R[c] = sum_k light[k][c] * T[k][c] with the per-pixel transfer
//   T[k][c] = kShNorm[k]/spp * sum_s w_s[c] * B_k(wi_s).
// T is computed once per material state (one pass per group of 9/8/8 basis functions keeps the 27 packed accumulators
// in registers); every relit frame is then 75 FMAs per pixel over 300 bytes -- HBM-bound, and with F lights per launch the
// T reads are amortised down to the 12-byte pixel write per frame.
// =================================================================================================
// Transfer layout: tiles of 256 consecutive pixels, each tile a contiguous [75][256] block (75 KB): the relight workgroup
// reads one contiguous block with lane-consecutive addresses (coalesced AND page-local; 75 planes 16 MB apart thrash the TLB).
__host__ __device__ inline long transfer_tiles(long P) { return (P + 255) / 256; }
__device__ __forceinline__ long transfer_index(int b, long p, int j, long P) {
    return (((long)b * transfer_tiles(P) + (p >> 8)) * kNL + j) * 256 + (p & 255);
}

template <int K0, int K1>
struct TransferUse {
    const f2* w;       // f*ip per channel for the two pixels
    f2* acc;           // [(K1-K0)*3]
    template <int K> __device__ __forceinline__ void operator()(f2 Bk) {
        if (K >= K0 && K < K1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[(K - K0) * 3 + c] = vfma(w[c], Bk, acc[(K - K0) * 3 + c]);
        }
    }
};

template <int K0, int K1>
__global__ __launch_bounds__(kBlock, 2) void shade_transfer_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                                   const float* __restrict__ m, const float* __restrict__ n,
                                                                   float* __restrict__ T, const Geom g, const SampleTable tab) {
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const int p0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    if (p0 >= P) return;
    const bool two = p0 + 1 < P;
    const int p1 = two ? p0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    Pixel px;
    load_pixel<false>(px, a, r, m, n, i0, i1, p0, p1, g);
    constexpr int NK = K1 - K0;
    f2 acc[NK * 3];
#pragma unroll
    for (int k = 0; k < NK * 3; ++k) acc[k] = f2{0.0f, 0.0f};
    float4 t = tab.diff[0];
    for (int s = 0; s < 2 * g.half; ++s) {
        Sample sm;
        const float4 tn = s + 1 < g.half ? tab.diff[s + 1] : tab.spec[s + 1 < 2 * g.half ? s + 1 - g.half : g.half - 1];
        if (s < g.half) diffuse_sample<false>(px, t.x, t.y, t.z, sm);
        else specular_sample<false>(px, t.x, t.y, t.z, t.w, sm);
        BrdfState<f2> st;
        f2 f[3], pdf;
        brdf_core(px.pc, sm.NoL_raw, sm.NoH, sm.VoH, sm.den, st, f, pdf);
        f2 ip = sel_pos(pdf - 1e-6f, rcp(pdf + 1e-6f), 0.0f);
        f2 w[3] = {f[0] * ip, f[1] * ip, f[2] * ip};
        sh_stream(sm.wi, TransferUse<K0, K1>{w, acc});
        t = tn;
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const float sc = kShNorm[K0 + k] * g.inv_spp;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f2 v = acc[k * 3 + c] * sc;
            const int j = (K0 + k) * 3 + c;
            T[transfer_index(b, p0, j, P)] = v.x;
            if (two) T[transfer_index(b, p1, j, P)] = v.y;
        }
    }
}

// out[f][p][c] = sum_k T[p][k][c] * light[f][k][c] for up to kRelightFrames lights per launch (lights in LDS, T read once)
constexpr int kRelightFrames = 8;
// the lights are read at wave-uniform addresses: scalar loads, SGPR operands of the FMAs (lights of frames >= n_frames must be readable)
__global__ __launch_bounds__(kBlock) void relight_kernel(const float* __restrict__ T, const float* __restrict__ L, float* __restrict__ out,
                                                         long P, int n_frames) {
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float acc[kRelightFrames][3];
#pragma unroll
    for (int f = 0; f < kRelightFrames; ++f) acc[f][0] = acc[f][1] = acc[f][2] = 0.0f;
    const float* tp = T + transfer_index(0, p, 0, P);
#pragma unroll 5
    for (int k = 0; k < kNSH; ++k) {
        const float t0 = tp[(k * 3) * 256], t1 = tp[(k * 3 + 1) * 256], t2 = tp[(k * 3 + 2) * 256];
#pragma unroll
        for (int f = 0; f < kRelightFrames; ++f) {
            const int fi = f < n_frames ? f : 0;     // uniform: stays a scalar load
            acc[f][0] = fmaf(t0, L[fi * kNL + k * 3], acc[f][0]);
            acc[f][1] = fmaf(t1, L[fi * kNL + k * 3 + 1], acc[f][1]);
            acc[f][2] = fmaf(t2, L[fi * kNL + k * 3 + 2], acc[f][2]);
        }
    }
#pragma unroll
    for (int f = 0; f < kRelightFrames; ++f) {
        if (f < n_frames) {
            float* o = out + ((long)f * P + p) * 3;
            o[0] = acc[f][0]; o[1] = acc[f][1]; o[2] = acc[f][2];
        }
    }
}

// =================================================================================================
// backward (sample directions and pdf are constants: stop-gradient, as in the reference's torch
// variants -- `D.data`, `alpha.data`, mi_plugin.py:179,366)
// =================================================================================================
template <bool WANT_LIGHT>
struct BwdAcc {
    BrdfGrad<f2> gr;
    f2 dnx, dny;  // gradient w.r.t. the unit normal, tangential components only
    float dc[WANT_LIGHT ? kNL : 1];
};

struct LightGradUse {   // dc[k][c] += gw[c] . B_k over the lane's two pixels, radiance alongside when materials need it
    const LightRegs& lr;
    const f2* gw;
    float* dc;
    f2* L;
    template <int K> __device__ __forceinline__ void operator()(f2 Bk) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dc[K * 3 + c] = fmaf(gw[c].y, Bk.y, fmaf(gw[c].x, Bk.x, dc[K * 3 + c]));
        if (L) sh_term<K>(lr, Bk, L);
    }
};

template <bool WANT_MAT, bool WANT_N, bool WANT_LIGHT>
__device__ __forceinline__ void bwd_accumulate(const Pixel& px, const Sample& sm, const LightRegs& lr, const f2 go[3],
                                               BwdAcc<WANT_LIGHT>& A) {
    BrdfState<f2> st;
    f2 f[3], pdf;
    brdf_core(px.pc, sm.NoL_raw, sm.NoH, sm.VoH, sm.den, st, f, pdf);
    f2 ip = sel_pos(pdf - 1e-6f, rcp(pdf + 1e-6f), 0.0f);
    f2 L[3];
    if (WANT_LIGHT) {
        f2 gw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gw[c] = (go[c] * f[c]) * ip;
        sh_stream(sm.wi, LightGradUse{lr, gw, A.dc, (WANT_MAT || WANT_N) ? L : nullptr});
    } else {
        sh_radiance(lr, sm.wi, L);
    }
    if (WANT_MAT || WANT_N) {
        f2 gg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gg[c] = (go[c] * L[c]) * ip;
        f2 gl, gh;
        brdf_core_grad<f2, WANT_N>(px.pc, st, gg, A.gr, gl, gh);
        if (WANT_N) {
            // dr.maximum(x, 0) passes the gradient where x > 0 (mi_plugin.py:1393-1396)
            gl = sel_pos(sm.NoL_raw, gl, 0.0f);
            gh = sel_pos(sm.nh_gate, gh, 0.0f);
            A.dnx = vfma(gl, sm.lwx, vfma(gh, sm.lhx, A.dnx));
            A.dny = vfma(gl, sm.lwy, vfma(gh, sm.lhy, A.dny));
        }
    }
}

// Fused BRDF-phase loss (hot loop B, `model_name == 'none'`, inverse_img_w_mi.py:371-420).  With FUSED the kernel
// takes the optimiser's raw parameter maps, forms d loss / d pred itself from the forward image, the gamma-2.2 target
// and the per-image statistics of matpbr_brdf_loss_stats, adds the L1 regularisers towards the initial maps, gates
// everything like torch.clamp's backward, and (when the statistics say this iteration is the best so far) snapshots
// the clamped maps and the gamma-2.2 render -- SaveBest without a host round trip.
struct FusedLoss {
    const float* pred;      // [B,H,W,3] linear render of this iteration (matpbr_shade_fwd with MATPBR_FLAG_CLAMP_PARAMS)
    const float* gt_srgb;   // [B,H,W,3] target ^ (1/2.2)
    const float* stats;     // [B,kStatsStride] from matpbr_brdf_loss_stats
    const float* a0; const float* r0; const float* m0;   // initial maps of the L1 regularisers (:398-409)
    float* best_a; float* best_r; float* best_m; float* best_img;  // nullable snapshot targets
    float scale_delta, inv_n3, inv_n1;
    unsigned part_mask;   // MATPBR_PART_A|R|M: which maps this phase optimises (their regularisers are active, :398-409)
    int check_stop;       // skip images whose on-device EarlyStopping has fired
};
constexpr float kLossEps = 1e-8f; // materialist_amd/loss.py _EPS: x^(1/2.2) has no gradient at exact zeros
__device__ __forceinline__ float fsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float pow_inv_gamma(float x) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.2f)); }

template <bool WANT_MAT, bool WANT_N, bool WANT_LIGHT, bool FUSED = false>
__global__ __launch_bounds__(kBlock, 2) void shade_bwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                              const float* __restrict__ m, const float* __restrict__ n,
                                                              const float* __restrict__ light, const float* __restrict__ d_out,
                                                              float* __restrict__ d_a, float* __restrict__ d_r,
                                                              float* __restrict__ d_m, float* __restrict__ d_n,
                                                              float* __restrict__ partials, const Geom g, const SampleTable tab,
                                                              const FusedLoss fl) {
    __shared__ float s_red[4][kNL + 1];
    const int b = blockIdx.y;
    if (FUSED && fl.check_stop && fl.stats[b * kStatsStride + kStStopped] > 0.5f) return;
    const float* __restrict__ cp = light + (long)b * kNL;
    const int P = g.H * g.W;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool act0 = q0 < P, two = q0 + 1 < P;
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    Pixel px;
    RawParams raw;
    load_pixel<FUSED>(px, a, r, m, n, i0, i1, p0, p1, g, &raw);
    LightRegs lr;
    if (WANT_MAT || WANT_N) load_light_regs(lr, cp);
    f2 go[3];
    float xs_keep[6];
    if (FUSED) {
        // d loss / d pred of  3 (l1/mse) mse + l1  on  xs = max(pred*ratio, eps)^(1/2.2)   (:388-418)
        const float ratio = fl.stats[b * kStatsStride + 0], sr = fl.stats[b * kStatsStride + 3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float gv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long ii = e ? i1 : i0;
                const bool on = e ? two : act0;
                float x = fl.pred[ii * 3 + c] * ratio;
                float xc = fmaxf(x, kLossEps);
                float xs = pow_inv_gamma(xc);
                float d = xs - fl.gt_srgb[ii * 3 + c];
                float dxs = x > kLossEps ? xs * rcp(xc) * (1.0f / 2.2f) : 0.0f;
                gv[e] = on ? ratio * dxs * fmaf(6.0f * sr, d, fsign(d)) * fl.inv_n3 * g.inv_spp : 0.0f;
                xs_keep[c * 2 + e] = xs;
            }
            go[c] = f2{gv[0], gv[1]};
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) go[c] = f2{act0 ? d_out[i0 * 3 + c] * g.inv_spp : 0.0f, two ? d_out[i1 * 3 + c] * g.inv_spp : 0.0f};
    }

    BwdAcc<WANT_LIGHT> A;
    brdf_grad_zero(A.gr);
    A.dnx = A.dny = f2{0.0f, 0.0f};
    if (WANT_LIGHT) {
#pragma unroll
        for (int k = 0; k < kNL; ++k) A.dc[k] = 0.0f;
    }

    float4 t = tab.diff[0];
    for (int s = 0; s < g.half; ++s) {
        Sample sm;
        const float4 tn = s + 1 < g.half ? tab.diff[s + 1] : tab.spec[0];
        diffuse_sample<WANT_N>(px, t.x, t.y, t.z, sm);
        bwd_accumulate<WANT_MAT, WANT_N, WANT_LIGHT>(px, sm, lr, go, A);
        t = tn;
    }
    for (int s = 0; s < g.half; ++s) {
        Sample sm;
        const float4 tn = tab.spec[s + 1 < g.half ? s + 1 : s];
        specular_sample<WANT_N>(px, t.x, t.y, t.z, t.w, sm);
        bwd_accumulate<WANT_MAT, WANT_N, WANT_LIGHT>(px, sm, lr, go, A);
        t = tn;
    }

    if (act0) {
        if (WANT_MAT && FUSED) {
            const bool improved = fl.stats[b * kStatsStride + 8] > 0.5f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (e && !two) break;
                const long ii = e ? i1 : i0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float pa = raw.a[c][e], ac = px.pc.a[c][e];
                    float gsum = A.gr.d_a[c][e] + ((fl.part_mask & MATPBR_PART_A) ? fl.scale_delta * fl.inv_n3 * fsign(ac - fl.a0[ii * 3 + c]) : 0.0f);   // :398,418
                    d_a[ii * 3 + c] = (pa >= 0.0f && pa <= 1.0f) ? gsum : 0.0f;                                 // clamp backward
                    if (improved && fl.best_a) fl.best_a[ii * 3 + c] = ac;
                    if (improved && fl.best_img) fl.best_img[ii * 3 + c] = xs_keep[c * 2 + e];
                }
                float pr = raw.r[e], rc = px.pc.r[e], pm = raw.m[e], mc = px.pc.m[e];
                float gr_ = A.gr.d_r[e] + ((fl.part_mask & MATPBR_PART_R) ? fl.scale_delta * fl.inv_n1 * fsign(rc - fl.r0[ii]) : 0.0f);
                float gm_ = A.gr.d_m[e] + ((fl.part_mask & MATPBR_PART_M) ? fl.scale_delta * fl.inv_n1 * fsign(mc - fl.m0[ii]) : 0.0f);
                d_r[ii] = (pr >= 0.07f && pr <= 1.0f) ? gr_ : 0.0f;
                d_m[ii] = (pm >= 0.0f && pm <= 1.0f) ? gm_ : 0.0f;
                if (improved && fl.best_r) fl.best_r[ii] = rc;
                if (improved && fl.best_m) fl.best_m[ii] = mc;
            }
        } else if (WANT_MAT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d_a[i0 * 3 + c] = A.gr.d_a[c].x;
                if (two) d_a[i1 * 3 + c] = A.gr.d_a[c].y;
            }
            d_r[i0] = A.gr.d_r.x;
            d_m[i0] = A.gr.d_m.x;
            if (two) { d_r[i1] = A.gr.d_r.y; d_m[i1] = A.gr.d_m.y; }
        }
        if (WANT_N) {
            // d/dn_hat = sum gl wi + gh h + gv wo.  Through n_hat = n/|n| only its tangential part survives:
            // d_n = (g - n_hat (n_hat.g)) / |n|, so g is accumulated in the shading frame's (s,t) plane directly and the
            // (huge, alternating-sign) radial parts of the GGX-peak terms never enter an fp32 sum.
            f2 gv = sel_pos(px.pc.NoV_raw, A.gr.dNoV, 0.0f);
            f2 dnx = vfma(gv, px.vx, A.dnx), dny = vfma(gv, px.vy, A.dny);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f2 v = vfma(px.s[c], dnx, px.t[c] * dny) * px.inv_len;
                d_n[i0 * 3 + c] = v.x;
                if (two) d_n[i1 * 3 + c] = v.y;
            }
        }
    }

    if (FUSED && !WANT_MAT && act0 && fl.best_img && fl.stats[b * kStatsStride + kStImproved] > 0.5f) {
        // env phase: SaveBest keeps the linear render (inverse_img_w_mi.py:247)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            fl.best_img[i0 * 3 + c] = fl.pred[i0 * 3 + c];
            if (two) fl.best_img[i1 * 3 + c] = fl.pred[i1 * 3 + c];
        }
    }
    if (WANT_LIGHT) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < kNL; ++k) {
            float v = wave_sum_to_lane63(A.dc[k]);
            if (lane == 63) s_red[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < kNL) {
            float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
            partials[((long)b * gridDim.x + blockIdx.x) * kNL + threadIdx.x] = v;
        }
    }
}

// =================================================================================================
// BRDF-phase loss statistics (inverse_img_w_mi.py:388-418) and the Adam update (torch.optim.Adam, :359)
// Two-pass reductions with fixed-order partial sums: bit-reproducible, no atomics.
// =================================================================================================
constexpr int kRedBlocks = 768;   // partial sums per image and pass (3 workgroups per CU: the passes are latency-bound)

__device__ __forceinline__ float block_sum(float v, float* s_buf) {   // all threads get the total
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) s_buf[wave] = v;
    __syncthreads();
    return (s_buf[0] + s_buf[1]) + (s_buf[2] + s_buf[3]);
}

// pass 1: sum(pred), sum(gt) -> ratio = mean(gt)/mean(pred)   (:388)
__global__ __launch_bounds__(kBlock) void loss_sums1_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            float* __restrict__ part, long n3) {
    __shared__ float s_buf[4];
    const int b = blockIdx.y;
    const float* p = pred + b * n3;
    const float* q = gt + b * n3;
    float sp = 0.0f, sg = 0.0f;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (long)gridDim.x * kBlock) { sp += p[i]; sg += q[i]; }
    sp = block_sum(sp, s_buf);
    sg = block_sum(sg, s_buf);
    if (threadIdx.x == 0) { part[((long)b * gridDim.x + blockIdx.x) * 2] = sp; part[((long)b * gridDim.x + blockIdx.x) * 2 + 1] = sg; }
}
__global__ __launch_bounds__(kBlock) void loss_final1_kernel(const float* __restrict__ part, float* __restrict__ stats, int nblk) {
    __shared__ float s_buf[4];
    const int b = blockIdx.x;
    float sp = 0.0f, sg = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += kBlock) { sp += part[((long)b * nblk + i) * 2]; sg += part[((long)b * nblk + i) * 2 + 1]; }
    sp = block_sum(sp, s_buf);
    sg = block_sum(sg, s_buf);
    if (threadIdx.x == 0) stats[b * kStatsStride + 0] = sg / sp;
}
// pass 2: sum (xs-gs)^2, sum |xs-gs| over [H,W,3]; sum |a-a0| over [H,W,3]; sum |r-r0|, |m-m0| over [H,W]
// FROM_FWD: ratio is formed here from the forward kernel's per-workgroup sums and the stored sum(gt) (no pass 1).
template <int MODE>   // 0: ratio from stats (piecewise API); 1: BRDF phase step (ratio from the forward sums); 2: env phase (ratio 1)
__global__ __launch_bounds__(kBlock) void loss_sums2_kernel(const float* __restrict__ pred, const float* __restrict__ gt_srgb,
                                                            const float* __restrict__ stats, const float* __restrict__ pa,
                                                            const float* __restrict__ a0, const float* __restrict__ pr,
                                                            const float* __restrict__ r0, const float* __restrict__ pm,
                                                            const float* __restrict__ m0, float* __restrict__ part, long n3, long n1,
                                                            const float* __restrict__ fwd_sums, int n_fwd) {
    __shared__ float s_buf[4];
    const int b = blockIdx.y;
    float ratio = 1.0f;
    if (MODE >= 1 && stats[b * kStatsStride + kStStopped] > 0.5f) return;
    if (MODE == 1) {
        float sp = 0.0f;
        for (int i = threadIdx.x; i < n_fwd; i += kBlock) sp += fwd_sums[(long)b * n_fwd + i];
        ratio = stats[b * kStatsStride + kStGtSum] / block_sum(sp, s_buf);
    } else if (MODE == 0) {
        ratio = stats[b * kStatsStride + kStRatio];
    }
    float s[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (long)gridDim.x * kBlock) {
        float xs = pow_inv_gamma(fmaxf(pred[b * n3 + i] * ratio, kLossEps));
        float d = xs - gt_srgb[b * n3 + i];
        s[0] = fmaf(d, d, s[0]);
        s[1] += fabsf(d);
        if (MODE != 2) s[2] += fabsf(fminf(fmaxf(pa[b * n3 + i], 0.0f), 1.0f) - a0[b * n3 + i]);
    }
    if (MODE != 2) {
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n1; i += (long)gridDim.x * kBlock) {
            s[3] += fabsf(fminf(fmaxf(pr[b * n1 + i], 0.07f), 1.0f) - r0[b * n1 + i]);
            s[4] += fabsf(fminf(fmaxf(pm[b * n1 + i], 0.0f), 1.0f) - m0[b * n1 + i]);
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float v = block_sum(s[k], s_buf);
        if (threadIdx.x == 0) part[((long)b * gridDim.x + blockIdx.x) * 5 + k] = v;
    }
}
// Per-image scalars of the iteration, SaveBest's decision, and (es_patience > 0) the EarlyStopping state machine of
// myutils/misc.py:37-60 kept on the device: once an image has stopped, every later kernel of the fused step skips it, so
// the host may enqueue iterations ahead and read the flag occasionally without changing any decision.
template <int MODE>
__global__ __launch_bounds__(kBlock) void loss_final2_kernel(const float* __restrict__ part, float* __restrict__ stats, int nblk,
                                                             float inv_n3, float inv_n1, float scale_delta, unsigned part_mask,
                                                             int es_patience, float es_min_delta, const float* __restrict__ fwd_sums,
                                                             int n_fwd, float* __restrict__ history, int hist_len, int batch) {
    __shared__ float s_buf[4];
    const int b = blockIdx.x;
    float* st = stats + b * kStatsStride;
    if (MODE >= 1 && st[kStStopped] > 0.5f) return;
    float s[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = threadIdx.x; i < nblk; i += kBlock) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += part[((long)b * nblk + i) * 5 + k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = block_sum(s[k], s_buf);
    float sp = 0.0f;
    if (MODE == 1) {
        for (int i = threadIdx.x; i < n_fwd; i += kBlock) sp += fwd_sums[(long)b * n_fwd + i];
        sp = block_sum(sp, s_buf);
    }
    if (threadIdx.x == 0) {
        if (MODE == 1) st[kStRatio] = st[kStGtSum] / sp;
        if (MODE == 2) st[kStRatio] = 1.0f;
        float mse = s[0] * inv_n3, l1 = s[1] * inv_n3;
        float la = (part_mask & MATPBR_PART_A) ? s[2] * inv_n3 : 0.0f;
        float lr = (part_mask & MATPBR_PART_R) ? s[3] * inv_n1 : 0.0f;
        float lm = (part_mask & MATPBR_PART_M) ? s[4] * inv_n1 : 0.0f;
        // scale_raito (:411), a constant of the backward pass; the env phase's loss is MSE + L1 (:244) = 3 (1/3) MSE + L1
        float sr = MODE == 2 ? (1.0f / 3.0f) : l1 / mse;
        st[kStMse] = mse; st[kStL1] = l1; st[kStSr] = sr; st[kStLa] = la; st[kStLr] = lr; st[kStLm] = lm;
        st[kStLoss] = 3.0f * sr * mse + l1 + scale_delta * (la + lr + lm);   // :412-414
        float best = st[kStBest];
        bool improved = mse < best;                            // SaveBest.update: strict < (myutils/misc.py:75)
        st[kStImproved] = improved ? 1.0f : 0.0f;
        st[kStBest] = improved ? mse : best;
        const int it = (int)st[kStIters];
        if (history && it < hist_len) history[(long)it * batch + b] = mse;
        st[kStIters] = (float)(it + 1);
        if (es_patience > 0) {                                 // EarlyStopping.__call__ (myutils/misc.py:51-60)
            if (st[kStEsHas] < 0.5f) { st[kStEsBest] = mse; st[kStEsHas] = 1.0f; }
            else if (mse > st[kStEsBest] * (1.0f - es_min_delta)) {
                float cnt = st[kStEsCounter] + 1.0f;
                st[kStEsCounter] = cnt;
                if (cnt >= (float)es_patience) st[kStStopped] = 1.0f;
            } else { st[kStEsBest] = mse; st[kStEsCounter] = 0.0f; }
        }
    }
}

// torch.optim.Adam (no weight decay, no amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(kBlock) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, long n, float lr_over_bc1, float b1, float b2, float eps,
                                                           float inv_sqrt_bc2) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        float gi = g[i];
        float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
        float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        m[i] = mi; v[i] = vi;
        p[i] -= lr_over_bc1 * mi / fmaf(fsqrt(vi), inv_sqrt_bc2, eps);
    }
}

// Adam for the selected parameter maps of a batch in one launch; blockIdx.y = image (stopped images are skipped),
// blockIdx.z = tensor (0 a, 1 r, 2 m).
struct Adam3 { float* p[3]; const float* g[3]; float* m[3]; float* v[3]; long n[3]; };
__global__ __launch_bounds__(kBlock) void adam3_kernel(const Adam3 t, const float* __restrict__ stats, unsigned part_mask, float lr_over_bc1,
                                                       float b1, float b2, float eps, float inv_sqrt_bc2) {
    const int b = blockIdx.y, z = blockIdx.z;
    if (!(part_mask & (MATPBR_PART_A << z))) return;
    if (stats[b * kStatsStride + kStStopped] > 0.5f) return;
    const long n = t.n[z], off = (long)b * n;
    float* p = t.p[z] + off; const float* g = t.g[z] + off; float* m = t.m[z] + off; float* v = t.v[z] + off;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        float gi = g[i];
        float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
        float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        m[i] = mi; v[i] = vi;
        p[i] -= lr_over_bc1 * mi / fmaf(fsqrt(vi), inv_sqrt_bc2, eps);
    }
}

// Column sums of a row-major [M, N] matrix (N <= 1024): the bias gradient of a Linear layer over M = H*W points.
// Pass 1: workgroup b sums rows [b*rows_per, ...) with thread t owning columns t, t+256, ... (coalesced row reads);
// pass 2 adds the per-workgroup partials in fixed order.  PyTorch's reduce_kernel and rocBLAS gemv both take
// milliseconds on this shape (262144 x 256); this is bandwidth-bound.
__global__ __launch_bounds__(kBlock) void colsum_pass1_kernel(const float* __restrict__ x, float* __restrict__ part, long M, int N,
                                                              long rows_per) {
    const long r0 = (long)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int c = threadIdx.x; c < N; c += kBlock) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        long r = r0;
        for (; r + 3 < r1; r += 4) {
            a0 += x[r * N + c]; a1 += x[(r + 1) * N + c]; a2 += x[(r + 2) * N + c]; a3 += x[(r + 3) * N + c];
        }
        for (; r < r1; ++r) a0 += x[r * N + c];
        part[(long)blockIdx.x * N + c] = (a0 + a1) + (a2 + a3);
    }
}
// pass 2: a workgroup owns 32 columns; its 8 thread rows split the partials, LDS folds them in fixed order
__global__ __launch_bounds__(kBlock) void colsum_pass2_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int N) {
    __shared__ float s_acc[8][33];
    const int cx = threadIdx.x & 31, cy = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float a = 0.0f;
    if (c < N)
        for (int b = cy; b < nblk; b += 8) a += part[(long)b * N + c];
    s_acc[cy][cx] = a;
    __syncthreads();
    if (cy == 0 && c < N) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_acc[k][cx];
        out[c] = t;
    }
}

// Backward of y = sin(pre): out[i][j] = d_y[i*ld_d + j] * cos(pre[i*ld_p + j]), contiguous [M, n] result
// (PosMLP hidden layers, mymodels/mlps.py:102-103; replaces a cos kernel + a mul kernel and takes the row-strided views
// that the skip-connection buffers produce).
__global__ __launch_bounds__(kBlock) void sin_bwd_kernel(const float* __restrict__ d_y, long ld_d, const float* __restrict__ pre, long ld_p,
                                                         float* __restrict__ out, long M, int n) {
    const long total = M * n;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long r = i / n;
        const int c = (int)(i - r * n);
        out[i] = d_y[r * ld_d + c] * cosf(pre[r * ld_p + c]);
    }
}

// d_light[b][k][c] = kShNorm[k] * sum over the image's workgroups of partials (fixed order -> reproducible)
__global__ __launch_bounds__(kBlock) void light_grad_finalize_kernel(const float* __restrict__ partials, float* __restrict__ d_light,
                                                                     int nblocks, const float* __restrict__ stats) {
    __shared__ float s_red[kBlock];
    const int b = blockIdx.y, k = blockIdx.x;  // one workgroup per light scalar
    if (stats && stats[b * kStatsStride + kStStopped] > 0.5f) {   // a stopped image contributes no gradient
        if (threadIdx.x == 0) d_light[(long)b * kNL + k] = 0.0f;
        return;
    }
    float v = 0.0f;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) v += partials[((long)b * nblocks + i) * kNL + k];
    s_red[threadIdx.x] = v;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) d_light[(long)b * kNL + k] = s_red[0] * kShNorm[k / 3];
}

// =================================================================================================
// plugin face: N independent lanes (literal restatement: raw n, h = normalize(wi+wo), all dots)
// =================================================================================================
struct Lane {
    PixelConst<float> pc;
    float NoL_raw, NoH, VoH, den, nh_raw;
    float h[3];
};
__device__ __forceinline__ void lane_setup(Lane& ln, const float wi[3], const float wo[3], const float n[3], const float a[3], float r,
                                           float m) {
    float h[3] = {wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]};
    float il = rsq(dot3(h, h));
#pragma unroll
    for (int c = 0; c < 3; ++c) ln.h[c] = h[c] * il;
    pixel_const<float>(ln.pc, a, r, m, dot3(n, wo));
    ln.NoL_raw = dot3(n, wi);
    ln.VoH = fmaxf(dot3(wo, ln.h), 0.0f);
    ln.nh_raw = dot3(n, ln.h);
    ln.NoH = fmaxf(ln.nh_raw, 0.0f);
    ln.den = ggx_den_literal(ln.pc, ln.NoH);
}

__global__ __launch_bounds__(kBlock) void eval_brdf_kernel(const float* __restrict__ wi, const float* __restrict__ wo,
                                                           const float* __restrict__ n, const float* __restrict__ a,
                                                           const float* __restrict__ r, const float* __restrict__ m,
                                                           float* __restrict__ f, float* __restrict__ pdf, long N) {
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= N) return;
    float wiv[3] = {wi[3 * k], wi[3 * k + 1], wi[3 * k + 2]}, wov[3] = {wo[3 * k], wo[3 * k + 1], wo[3 * k + 2]};
    float nv[3] = {n[3 * k], n[3 * k + 1], n[3 * k + 2]}, av[3] = {a[3 * k], a[3 * k + 1], a[3 * k + 2]};
    Lane ln;
    lane_setup(ln, wiv, wov, nv, av, r[k], m[k]);
    BrdfState<float> st;
    float fv[3], p;
    brdf_core(ln.pc, ln.NoL_raw, ln.NoH, ln.VoH, ln.den, st, fv, p);
#pragma unroll
    for (int c = 0; c < 3; ++c) f[3 * k + c] = fv[c];
    pdf[k] = p;
}

__global__ __launch_bounds__(kBlock) void eval_brdf_bwd_kernel(const float* __restrict__ wi, const float* __restrict__ wo,
                                                               const float* __restrict__ n, const float* __restrict__ a,
                                                               const float* __restrict__ r, const float* __restrict__ m,
                                                               const float* __restrict__ g, float* __restrict__ d_a,
                                                               float* __restrict__ d_r, float* __restrict__ d_m,
                                                               float* __restrict__ d_n, long N) {
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= N) return;
    float wiv[3] = {wi[3 * k], wi[3 * k + 1], wi[3 * k + 2]}, wov[3] = {wo[3 * k], wo[3 * k + 1], wo[3 * k + 2]};
    float nv[3] = {n[3 * k], n[3 * k + 1], n[3 * k + 2]}, av[3] = {a[3 * k], a[3 * k + 1], a[3 * k + 2]};
    float gv[3] = {g[3 * k], g[3 * k + 1], g[3 * k + 2]};
    Lane ln;
    lane_setup(ln, wiv, wov, nv, av, r[k], m[k]);
    BrdfState<float> st;
    float fv[3], p;
    brdf_core(ln.pc, ln.NoL_raw, ln.NoH, ln.VoH, ln.den, st, fv, p);
    BrdfGrad<float> o;
    brdf_grad_zero(o);
    float gl = 0.0f, gh = 0.0f;
    brdf_core_grad<float, true>(ln.pc, st, gv, o, gl, gh);
    gl = ln.NoL_raw > 0.0f ? gl : 0.0f;
    gh = ln.nh_raw > 0.0f ? gh : 0.0f;
    float gvv = ln.pc.NoV_raw > 0.0f ? o.dNoV : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d_a[3 * k + c] = o.d_a[c];
        d_n[3 * k + c] = fmaf(gl, wiv[c], fmaf(gvv, wov[c], gh * ln.h[c]));
    }
    d_r[k] = o.d_r;
    d_m[k] = o.d_m;
}

// a1-a3 as N-lane functions (myutils/mi_plugin.py:60-97): D_GGX(cos_h, r), G1_GGX_Schlick(cos, r), G_Smith(cos, cos2, r),
// fresnelSchlick(cos, f0) -> out[N,4]
__global__ __launch_bounds__(kBlock) void brdf_terms_kernel(const float* __restrict__ cos1, const float* __restrict__ cos2,
                                                            const float* __restrict__ r, const float* __restrict__ f0, float* __restrict__ out,
                                                            long N) {
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= N) return;
    out[4 * k + 0] = D_GGX(cos1[k], r[k]);
    out[4 * k + 1] = G1_GGX_Schlick(cos1[k], r[k]);
    out[4 * k + 2] = G_Smith(cos1[k], cos2[k], r[k]);
    out[4 * k + 3] = fresnelSchlick(cos1[k], f0[k]);
}

__global__ __launch_bounds__(kBlock) void sample_brdf_kernel(const float* __restrict__ sample1, const float* __restrict__ sample2,
                                                             const float* __restrict__ wo, const float* __restrict__ n,
                                                             const float* __restrict__ a, const float* __restrict__ r,
                                                             const float* __restrict__ m, float* __restrict__ wi,
                                                             float* __restrict__ pdf, float* __restrict__ weight, long N) {
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= N) return;
    float wov[3] = {wo[3 * k], wo[3 * k + 1], wo[3 * k + 2]};
    float nv[3] = {n[3 * k], n[3 * k + 1], n[3 * k + 2]}, av[3] = {a[3 * k], a[3 * k + 1], a[3 * k + 2]};
    float u0 = sample2[2 * k], u1 = sample2[2 * k + 1], rv = r[k];
    float s[3], t[3], wiv[3];
    frame(nv, s, t);
    float sp, cp;
    sincosf(2.0f * kPi * u1, &sp, &cp);
    float sin2_h = -1.0f, cos_h = 0.0f;  // exact 1-NoH^2 of a GGX-sampled half vector (unit n assumed)
    if (sample1[k] > 0.5f) {  // diffuse lobe (mi_plugin.py:1328-1329)
        float st_ = fsqrt(fmaxf(u0, 0.0f)), ct = fsqrt(fmaxf(1.0f - u0, 0.0f));
        to_world(s, t, nv, st_ * cp, st_ * sp, ct, wiv);
    } else {  // GGX lobe (mi_plugin.py:1330-1331)
        float alpha2 = pow4(rv);
        float q = rcp(fmaf(u0, alpha2 - 1.0f, 1.0f));
        float ct = fsqrt(fmaxf((1.0f - u0) * q, 0.0f)), st_ = fsqrt(fmaxf(u0 * alpha2 * q, 0.0f));
        float wh[3];
        to_world(s, t, nv, st_ * cp, st_ * sp, ct, wh);
        float d = 2.0f * dot3(wov, wh);
#pragma unroll
        for (int c = 0; c < 3; ++c) wiv[c] = fmaf(d, wh[c], -wov[c]);
        float il = rsq(dot3(wiv, wiv));
#pragma unroll
        for (int c = 0; c < 3; ++c) wiv[c] *= il;
        if (d > 0.0f) { sin2_h = u0 * alpha2 * q; cos_h = ct; }
    }
    Lane ln;
    lane_setup(ln, wiv, wov, nv, av, rv, m[k]);
    if (sin2_h >= 0.0f) {  // same value as the literal form, without the fp32 cancellation at the GGX peak
        ln.NoH = cos_h;
        ln.den = ggx_den_stable(ln.pc, sin2_h);
    }
    BrdfState<float> st;
    float fv[3], p;
    brdf_core(ln.pc, ln.NoL_raw, ln.NoH, ln.VoH, ln.den, st, fv, p);
    float ip = p > 1e-6f ? rcp(p + 1e-6f) : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wi[3 * k + c] = wiv[c];
        weight[3 * k + c] = fv[c] * ip;
    }
    pdf[k] = p > 0.0f ? p : 0.0f;
}

__global__ __launch_bounds__(kBlock) void sh_eval_kernel(const float* __restrict__ w, const float* __restrict__ coef,
                                                         float* __restrict__ L, long N) {
    __shared__ float s_c[kNL + 1];
    if (threadIdx.x < kNL) s_c[threadIdx.x] = coef[threadIdx.x] * kShNorm[threadIdx.x / 3];
    __syncthreads();
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= N) return;
    float wv[3] = {w[3 * k], w[3 * k + 1], w[3 * k + 2]};
    float B[kNSH];
    sh_poly(wv, B);
    float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < kNSH; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fmaf(s_c[i * 3 + c], B[i], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) L[3 * k + c] = acc[c];
}

// per-pixel geometric normal of the depth heightfield (oracle_normals_from_depth is the spec)
__global__ __launch_bounds__(kBlock) void normals_from_depth_kernel(const float* __restrict__ depth, float* __restrict__ out_n,
                                                                    const Geom g) {
    const int P = g.H * g.W;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const float* d = depth + (long)blockIdx.y * P;
    const int i = p / g.W, j = p % g.W;
    const int j0 = j > 0 ? j - 1 : j, j1 = j < g.W - 1 ? j + 1 : j;
    const int i0 = i > 0 ? i - 1 : i, i1 = i < g.H - 1 ? i + 1 : i;
    auto world = [&](int ii, int jj, float out[3]) {
        float dd = d[(long)ii * g.W + jj];
        out[0] = ((float)jj - g.cx) * g.inv_f * dd;
        out[1] = -((float)ii - g.cy) * g.inv_f * dd;
        out[2] = -dd;
    };
    float pl[3], pr[3], pu[3], pd[3], c[3];
    world(i, j0, pl); world(i, j1, pr); world(i0, j, pu); world(i1, j, pd); world(i, j, c);
    float dx[3] = {pr[0] - pl[0], pr[1] - pl[1], pr[2] - pl[2]};
    float dy[3] = {pd[0] - pu[0], pd[1] - pu[1], pd[2] - pu[2]};
    float nn[3] = {dx[1] * dy[2] - dx[2] * dy[1], dx[2] * dy[0] - dx[0] * dy[2], dx[0] * dy[1] - dx[1] * dy[0]};
    float l2 = dot3(nn, nn);
    float sgn = dot3(nn, c) > 0.0f ? -1.0f : 1.0f;
    long o = ((long)blockIdx.y * P + p) * 3;
    if (l2 > 0.0f) {
        float il = sgn * rsq(l2);
        out_n[o] = nn[0] * il; out_n[o + 1] = nn[1] * il; out_n[o + 2] = nn[2] * il;
    } else {
        out_n[o] = 0.0f; out_n[o + 1] = 0.0f; out_n[o + 2] = 1.0f;
    }
}

// ---- host helpers ------------------------------------------------------------------------------
bool make_geom(int H, int W, int spp, const MatpbrCamera* cam, Geom& g) {
    if (H <= 0 || W <= 0 || (long)H * W > 0x7fffffffL / 4) return false;
    float fov = cam ? cam->fov_x_deg : 35.0f;
    if (!(fov > 0.0f && fov < 179.0f)) return false;
    double f = (0.5 * W) / std::tan(0.5 * (double)fov * M_PI / 180.0);
    g.H = H; g.W = W; g.half = spp / 2;
    g.inv_f = (float)(1.0 / f);
    g.cx = 0.5f * (float)(W - 1);
    g.cy = 0.5f * (float)(H - 1);
    g.inv_spp = spp > 0 ? 1.0f / (float)spp : 0.0f;
    return true;
}
// two pixels per lane: a 256-thread workgroup covers 512 pixels
int grid_blocks(int H, int W) { return (int)(((long)H * W + 2 * kBlock - 1) / (2 * kBlock)); }
bool valid_spp(int spp) { return spp >= 2 && spp <= MATPBR_MAX_SPP && (spp % 2) == 0; }
int launch_status() { return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH; }

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int matpbr_version(void) { return MATPBR_VERSION; }

const char* matpbr_strerror(int code) {
    switch (code) {
        case MATPBR_OK: return "ok";
        case MATPBR_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size, or unsupported light kind)";
        case MATPBR_ERR_UNSUPPORTED: return "unsupported spp (must be even, 2..128)";
        case MATPBR_ERR_LAUNCH: return "HIP kernel launch failed";
        case MATPBR_ERR_WORKSPACE: return "workspace missing or smaller than matpbr_shade_bwd_workspace_bytes() (needed for d_light)";
        default: return "unknown matpbr error";
    }
}

int matpbr_shade_fwd(const float* a, const float* r, const float* m, const float* n, const float* light, int light_kind,
                     int n_light, float* out_rgb, int H, int W, int batch, int spp, const MatpbrCamera* cam, uint32_t flags,
                     void* stream) {
    if (!a || !r || !m || !n || !light || !out_rgb || batch <= 0) return MATPBR_ERR_INVALID_ARG;
    if (light_kind != MATPBR_LIGHT_SH25 || n_light != MATPBR_NSH) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(spp)) return MATPBR_ERR_UNSUPPORTED;
    Geom g;
    if (!make_geom(H, W, spp, cam, g)) return MATPBR_ERR_INVALID_ARG;
    SampleTable tab;
    fill_sample_table(spp, tab);
    dim3 grid((unsigned)grid_blocks(H, W), (unsigned)batch);
    if (flags & MATPBR_FLAG_CLAMP_PARAMS)
        hipLaunchKernelGGL((shade_fwd_kernel<true, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, a, r, m, n, light, out_rgb, g, tab,
                           (const float*)nullptr, (float*)nullptr);
    else
        hipLaunchKernelGGL((shade_fwd_kernel<false, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, a, r, m, n, light, out_rgb, g, tab,
                           (const float*)nullptr, (float*)nullptr);
    return launch_status();
}

size_t matpbr_shade_bwd_workspace_bytes(int H, int W, int batch, int n_light) {
    if (H <= 0 || W <= 0 || batch <= 0 || n_light <= 0) return 0;
    return (size_t)grid_blocks(H, W) * (size_t)batch * (size_t)n_light * 3 * sizeof(float);
}

int matpbr_shade_bwd(const float* a, const float* r, const float* m, const float* n, const float* light, int light_kind,
                     int n_light, const float* d_out_rgb, float* d_a, float* d_r, float* d_m, float* d_n, float* d_light,
                     void* workspace, size_t workspace_bytes, int H, int W, int batch, int spp, const MatpbrCamera* cam,
                     uint32_t flags, void* stream) {
    (void)flags;
    if (!a || !r || !m || !n || !light || !d_out_rgb || batch <= 0) return MATPBR_ERR_INVALID_ARG;
    if (light_kind != MATPBR_LIGHT_SH25 || n_light != MATPBR_NSH) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(spp)) return MATPBR_ERR_UNSUPPORTED;
    const bool want_mat = d_a || d_r || d_m;
    if (want_mat && !(d_a && d_r && d_m)) return MATPBR_ERR_INVALID_ARG;
    const bool want_n = d_n != nullptr, want_light = d_light != nullptr;
    if (!want_mat && !want_n && !want_light) return MATPBR_OK;
    Geom g;
    if (!make_geom(H, W, spp, cam, g)) return MATPBR_ERR_INVALID_ARG;
    if (want_light && (!workspace || workspace_bytes < matpbr_shade_bwd_workspace_bytes(H, W, batch, n_light)))
        return MATPBR_ERR_WORKSPACE;
    SampleTable tab;
    fill_sample_table(spp, tab);
    dim3 grid((unsigned)grid_blocks(H, W), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
#define MATPBR_LAUNCH_BWD(MAT, NRM, LGT) \
    hipLaunchKernelGGL((shade_bwd_kernel<MAT, NRM, LGT>), grid, dim3(kBlock), 0, st, a, r, m, n, light, d_out_rgb, d_a, d_r, d_m, d_n, part, g, tab, FusedLoss{})
    // The light gradient keeps 75 accumulators per lane and the material/normal gradients keep the 38 coefficient
    // pairs: together they exceed the 256-VGPR budget of two waves per SIMD, so a call that wants both runs two launches.
    switch ((want_mat ? 2 : 0) | (want_n ? 1 : 0)) {
        case 1: MATPBR_LAUNCH_BWD(false, true, false); break;
        case 2: MATPBR_LAUNCH_BWD(true, false, false); break;
        case 3: MATPBR_LAUNCH_BWD(true, true, false); break;
        default: break;
    }
    if (want_light) MATPBR_LAUNCH_BWD(false, false, true);
#undef MATPBR_LAUNCH_BWD
    if (hipGetLastError() != hipSuccess) return MATPBR_ERR_LAUNCH;
    if (want_light) {
        hipLaunchKernelGGL(light_grad_finalize_kernel, dim3(kNL, (unsigned)batch), dim3(kBlock), 0, st, part, d_light, (int)grid.x,
                           (const float*)nullptr);
    }
    return launch_status();
}

int matpbr_eval_brdf(const float* wi, const float* wo, const float* n, const float* a, const float* r, const float* m, float* f,
                     float* pdf, long N, void* stream) {
    if (!wi || !wo || !n || !a || !r || !m || !f || !pdf || N < 0) return MATPBR_ERR_INVALID_ARG;
    if (N == 0) return MATPBR_OK;
    hipLaunchKernelGGL(eval_brdf_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, wi, wo, n, a,
                       r, m, f, pdf, N);
    return launch_status();
}

int matpbr_eval_brdf_bwd(const float* wi, const float* wo, const float* n, const float* a, const float* r, const float* m,
                         const float* g, float* d_a, float* d_r, float* d_m, float* d_n, long N, void* stream) {
    if (!wi || !wo || !n || !a || !r || !m || !g || !d_a || !d_r || !d_m || !d_n || N < 0) return MATPBR_ERR_INVALID_ARG;
    if (N == 0) return MATPBR_OK;
    hipLaunchKernelGGL(eval_brdf_bwd_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, wi, wo,
                       n, a, r, m, g, d_a, d_r, d_m, d_n, N);
    return launch_status();
}

int matpbr_brdf_terms(const float* cos1, const float* cos2, const float* r, const float* f0, float* out, long N, void* stream) {
    if (!cos1 || !cos2 || !r || !f0 || !out || N < 0) return MATPBR_ERR_INVALID_ARG;
    if (N == 0) return MATPBR_OK;
    hipLaunchKernelGGL(brdf_terms_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, cos1, cos2, r, f0,
                       out, N);
    return launch_status();
}

int matpbr_sample_brdf(const float* sample1, const float* sample2, const float* wo, const float* n, const float* a, const float* r,
                       const float* m, float* wi, float* pdf, float* weight, long N, void* stream) {
    if (!sample1 || !sample2 || !wo || !n || !a || !r || !m || !wi || !pdf || !weight || N < 0) return MATPBR_ERR_INVALID_ARG;
    if (N == 0) return MATPBR_OK;
    hipLaunchKernelGGL(sample_brdf_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, sample1,
                       sample2, wo, n, a, r, m, wi, pdf, weight, N);
    return launch_status();
}

int matpbr_sh_eval(const float* w, const float* coef, float* L, long N, void* stream) {
    if (!w || !coef || !L || N < 0) return MATPBR_ERR_INVALID_ARG;
    if (N == 0) return MATPBR_OK;
    hipLaunchKernelGGL(sh_eval_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, w, coef, L, N);
    return launch_status();
}

int matpbr_normals_from_depth(const float* depth, float* out_n, int H, int W, int batch, const MatpbrCamera* cam, void* stream) {
    if (!depth || !out_n || batch <= 0) return MATPBR_ERR_INVALID_ARG;
    Geom g;
    if (!make_geom(H, W, 2, cam, g)) return MATPBR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(normals_from_depth_kernel, dim3((unsigned)((H * W + kBlock - 1) / kBlock), (unsigned)batch), dim3(kBlock), 0,
                       (hipStream_t)stream, depth, out_n, g);
    return launch_status();
}

size_t matpbr_brdf_loss_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * kRedBlocks * 5 * sizeof(float) : 0; }

static unsigned part_mask_of(uint32_t flags) {
    const unsigned all = MATPBR_PART_A | MATPBR_PART_R | MATPBR_PART_M;
    return (flags & all) ? (flags & all) : all;   // no part bit = all three maps (optimize_part 'arm')
}

int matpbr_brdf_loss_stats(const float* pred, const float* gt, const float* gt_srgb, const float* pa, const float* pr, const float* pm,
                           const float* a0, const float* r0, const float* m0, float scale_delta, float* stats, void* workspace,
                           size_t workspace_bytes, int H, int W, int batch, uint32_t flags, void* stream) {
    if (!pred || !gt || !gt_srgb || !pa || !pr || !pm || !a0 || !r0 || !m0 || !stats || H <= 0 || W <= 0 || batch <= 0)
        return MATPBR_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < matpbr_brdf_loss_workspace_bytes(batch)) return MATPBR_ERR_WORKSPACE;
    const long n1 = (long)H * W, n3 = n1 * 3;
    float* part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(kRedBlocks, (unsigned)batch);
    hipLaunchKernelGGL(loss_sums1_kernel, grid, dim3(kBlock), 0, st, pred, gt, part, n3);
    hipLaunchKernelGGL(loss_final1_kernel, dim3((unsigned)batch), dim3(kBlock), 0, st, part, stats, kRedBlocks);
    hipLaunchKernelGGL(loss_sums2_kernel<0>, grid, dim3(kBlock), 0, st, pred, gt_srgb, stats, pa, a0, pr, r0, pm, m0, part, n3, n1,
                       (const float*)nullptr, 0);
    hipLaunchKernelGGL(loss_final2_kernel<0>, dim3((unsigned)batch), dim3(kBlock), 0, st, part, stats, kRedBlocks, 1.0f / (float)n3,
                       1.0f / (float)n1, scale_delta, part_mask_of(flags), 0, 0.0f,
                       (const float*)nullptr, 0, (float*)nullptr, 0, batch);
    return launch_status();
}

int matpbr_shade_bwd_brdf_loss(const float* pa, const float* pr, const float* pm, const float* n, const float* light, int light_kind,
                               int n_light, const float* pred, const float* gt_srgb, const float* stats, const float* a0,
                               const float* r0, const float* m0, float scale_delta, float* d_a, float* d_r, float* d_m, float* best_a,
                               float* best_r, float* best_m, float* best_img, int H, int W, int batch, int spp, const MatpbrCamera* cam,
                               uint32_t flags, void* stream) {
    if (!pa || !pr || !pm || !n || !light || !pred || !gt_srgb || !stats || !a0 || !r0 || !m0 || !d_a || !d_r || !d_m || batch <= 0)
        return MATPBR_ERR_INVALID_ARG;
    if (light_kind != MATPBR_LIGHT_SH25 || n_light != MATPBR_NSH) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(spp)) return MATPBR_ERR_UNSUPPORTED;
    Geom g;
    if (!make_geom(H, W, spp, cam, g)) return MATPBR_ERR_INVALID_ARG;
    SampleTable tab;
    fill_sample_table(spp, tab);
    FusedLoss fl{pred, gt_srgb, stats, a0, r0, m0, best_a, best_r, best_m, best_img, scale_delta, 1.0f / (3.0f * (float)H * (float)W),
                 1.0f / ((float)H * (float)W), part_mask_of(flags), 0};
    dim3 grid((unsigned)grid_blocks(H, W), (unsigned)batch);
    hipLaunchKernelGGL((shade_bwd_kernel<true, false, false, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, pa, pr, pm, n, light,
                       (const float*)nullptr, d_a, d_r, d_m, (float*)nullptr, (float*)nullptr, g, tab, fl);
    return launch_status();
}

size_t matpbr_brdf_phase_workspace_bytes(int H, int W, int batch) {
    if (H <= 0 || W <= 0 || batch <= 0) return 0;
    return ((size_t)batch * grid_blocks(H, W) + (size_t)batch * kRedBlocks * 5) * sizeof(float);
}

int matpbr_brdf_phase_step(const MatpbrBrdfPhase* ph, int t, float lr, void* stream) {
    if (!ph || t < 1) return MATPBR_ERR_INVALID_ARG;
    const MatpbrBrdfPhase& q = *ph;
    if (!q.pa || !q.pr || !q.pm || !q.n || !q.light || !q.gt_srgb || !q.a0 || !q.r0 || !q.m0 || !q.pred || !q.d_a || !q.d_r || !q.d_m ||
        !q.stats || q.batch <= 0)
        return MATPBR_ERR_INVALID_ARG;
    for (int z = 0; z < 3; ++z)
        if ((q.part_mask & (MATPBR_PART_A << z)) && (!q.adam_m[z] || !q.adam_v[z])) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(q.spp)) return MATPBR_ERR_UNSUPPORTED;
    if (!q.workspace || q.workspace_bytes < matpbr_brdf_phase_workspace_bytes(q.H, q.W, q.batch)) return MATPBR_ERR_WORKSPACE;
    MatpbrCamera cam{q.fov_x_deg};
    Geom g;
    if (!make_geom(q.H, q.W, q.spp, &cam, g)) return MATPBR_ERR_INVALID_ARG;
    SampleTable tab;
    fill_sample_table(q.spp, tab);
    hipStream_t st = (hipStream_t)stream;
    const int nfwd = grid_blocks(q.H, q.W);
    float* fwd_sums = (float*)q.workspace;
    float* part = fwd_sums + (size_t)q.batch * nfwd;
    const long n1 = (long)q.H * q.W, n3 = n1 * 3;
    dim3 grid((unsigned)nfwd, (unsigned)q.batch);
    // 1. render with the clamped parameters (:371-386) + per-workgroup sums for mean(pred)
    hipLaunchKernelGGL((shade_fwd_kernel<true, true>), grid, dim3(kBlock), 0, st, q.pa, q.pr, q.pm, q.n, q.light, q.pred, g, tab, q.stats,
                       fwd_sums);
    // 2. loss statistics, SaveBest / EarlyStopping decisions (:388-418, misc.py:37-97)
    hipLaunchKernelGGL(loss_sums2_kernel<1>, dim3(kRedBlocks, (unsigned)q.batch), dim3(kBlock), 0, st, q.pred, q.gt_srgb, q.stats, q.pa, q.a0,
                       q.pr, q.r0, q.pm, q.m0, part, n3, n1, fwd_sums, nfwd);
    hipLaunchKernelGGL(loss_final2_kernel<1>, dim3((unsigned)q.batch), dim3(kBlock), 0, st, part, q.stats, kRedBlocks, 1.0f / (float)n3,
                       1.0f / (float)n1, q.scale_delta, q.part_mask, q.es_patience, q.es_min_delta, fwd_sums, nfwd, q.history, q.hist_len,
                       q.batch);
    // 3. backward of the loss through the render (:420), regularisers, clamp gating, best-so-far snapshot
    FusedLoss fl{q.pred, q.gt_srgb, q.stats, q.a0, q.r0, q.m0, q.best_a, q.best_r, q.best_m, q.best_img, q.scale_delta, 1.0f / (float)n3,
                 1.0f / (float)n1, q.part_mask, 1};
    hipLaunchKernelGGL((shade_bwd_kernel<true, false, false, true>), grid, dim3(kBlock), 0, st, q.pa, q.pr, q.pm, q.n, q.light,
                       (const float*)nullptr, q.d_a, q.d_r, q.d_m, (float*)nullptr, (float*)nullptr, g, tab, fl);
    // 4. Adam on the maps of this part (:359,429)
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const double bc1 = 1.0 - std::pow((double)b1, t), bc2 = 1.0 - std::pow((double)b2, t);
    Adam3 ad{{q.pa, q.pr, q.pm}, {q.d_a, q.d_r, q.d_m}, {q.adam_m[0], q.adam_m[1], q.adam_m[2]}, {q.adam_v[0], q.adam_v[1], q.adam_v[2]},
             {n3, n1, n1}};
    hipLaunchKernelGGL(adam3_kernel, dim3(768, (unsigned)q.batch, 3), dim3(kBlock), 0, st, ad, q.stats, q.part_mask, (float)(lr / bc1), b1, b2,
                       eps, (float)(1.0 / std::sqrt(bc2)));
    return launch_status();
}

size_t matpbr_env_phase_workspace_bytes(int H, int W, int batch) {
    if (H <= 0 || W <= 0 || batch <= 0) return 0;
    return ((size_t)batch * grid_blocks(H, W) * kNL + (size_t)batch * kRedBlocks * 5) * sizeof(float);
}

int matpbr_env_phase_step(const float* a, const float* r, const float* m, const float* n, const float* light, const float* gt_srgb,
                          float* pred, float* d_light, float* stats, float* best_img, float* history, int hist_len, int es_patience,
                          float es_min_delta, void* workspace, size_t workspace_bytes, int H, int W, int batch, int spp,
                          const MatpbrCamera* cam, void* stream) {
    if (!a || !r || !m || !n || !light || !gt_srgb || !pred || !d_light || !stats || batch <= 0) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(spp)) return MATPBR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < matpbr_env_phase_workspace_bytes(H, W, batch)) return MATPBR_ERR_WORKSPACE;
    Geom g;
    if (!make_geom(H, W, spp, cam, g)) return MATPBR_ERR_INVALID_ARG;
    SampleTable tab;
    fill_sample_table(spp, tab);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = grid_blocks(H, W);
    float* lpart = (float*)workspace;
    float* part = lpart + (size_t)batch * nblk * kNL;
    const long n1 = (long)H * W, n3 = n1 * 3;
    dim3 grid((unsigned)nblk, (unsigned)batch);
    // render under the candidate light (:240); images whose EarlyStopping fired are skipped by every kernel
    hipLaunchKernelGGL((shade_fwd_kernel<false, true>), grid, dim3(kBlock), 0, st, a, r, m, n, light, pred, g, tab, (const float*)stats, lpart);
    // loss = MSE + L1 on x^(1/2.2) (:241-245); SaveBest / EarlyStopping decisions (:247,250)
    hipLaunchKernelGGL(loss_sums2_kernel<2>, dim3(kRedBlocks, (unsigned)batch), dim3(kBlock), 0, st, (const float*)pred, gt_srgb,
                       (const float*)stats, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, part, n3, n1, (const float*)nullptr, 0);
    hipLaunchKernelGGL(loss_final2_kernel<2>, dim3((unsigned)batch), dim3(kBlock), 0, st, (const float*)part, stats, kRedBlocks,
                       1.0f / (float)n3, 1.0f / (float)n1, 0.0f, 0u, es_patience, es_min_delta, (const float*)nullptr, 0, history, hist_len, batch);
    // d loss / d light through the render (:248)
    FusedLoss fl{pred, gt_srgb, stats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, best_img, 0.0f, 1.0f / (float)n3, 1.0f / (float)n1, 0u, 1};
    hipLaunchKernelGGL((shade_bwd_kernel<false, false, true, true>), grid, dim3(kBlock), 0, st, a, r, m, n, light, (const float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, lpart, g, tab, fl);
    hipLaunchKernelGGL(light_grad_finalize_kernel, dim3(kNL, (unsigned)batch), dim3(kBlock), 0, st, (const float*)lpart, d_light, nblk,
                       (const float*)stats);
    return launch_status();
}

int matpbr_shade_transfer(const float* a, const float* r, const float* m, const float* n, float* T, int H, int W, int batch, int spp,
                          const MatpbrCamera* cam, uint32_t flags, void* stream) {
    (void)flags;
    if (!a || !r || !m || !n || !T || batch <= 0) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(spp)) return MATPBR_ERR_UNSUPPORTED;
    Geom g;
    if (!make_geom(H, W, spp, cam, g)) return MATPBR_ERR_INVALID_ARG;
    SampleTable tab;
    fill_sample_table(spp, tab);
    dim3 grid((unsigned)grid_blocks(H, W), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((shade_transfer_kernel<0, 9>), grid, dim3(kBlock), 0, st, a, r, m, n, T, g, tab);
    hipLaunchKernelGGL((shade_transfer_kernel<9, 17>), grid, dim3(kBlock), 0, st, a, r, m, n, T, g, tab);
    hipLaunchKernelGGL((shade_transfer_kernel<17, 25>), grid, dim3(kBlock), 0, st, a, r, m, n, T, g, tab);
    return launch_status();
}

size_t matpbr_transfer_bytes(int H, int W, int batch) {
    if (H <= 0 || W <= 0 || batch <= 0) return 0;
    return (size_t)batch * transfer_tiles((long)H * W) * kNL * 256 * sizeof(float);
}

int matpbr_relight(const float* T, const float* lights, float* out_rgb, int H, int W, int n_frames, void* stream) {
    if (!T || !lights || !out_rgb || H <= 0 || W <= 0 || n_frames <= 0) return MATPBR_ERR_INVALID_ARG;
    const long P = (long)H * W;
    hipStream_t st = (hipStream_t)stream;
    for (int f0 = 0; f0 < n_frames; f0 += kRelightFrames) {
        const int nf = n_frames - f0 < kRelightFrames ? n_frames - f0 : kRelightFrames;
        hipLaunchKernelGGL(relight_kernel, dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, T, lights + (long)f0 * kNL,
                           out_rgb + (long)f0 * P * 3, P, nf);
    }
    return launch_status();
}

int matpbr_sin_bwd(const float* d_y, long ld_d, const float* pre, long ld_p, float* out, long M, int n, void* stream) {
    if (!d_y || !pre || !out || M <= 0 || n <= 0 || ld_d < n || ld_p < n) return MATPBR_ERR_INVALID_ARG;
    const long total = M * n;
    unsigned blocks = (unsigned)std::min<long>((total + kBlock - 1) / kBlock, 256L * 16);
    hipLaunchKernelGGL(sin_bwd_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, d_y, ld_d, pre, ld_p, out, M, n);
    return launch_status();
}

constexpr int kColsumBlocks = 512;
size_t matpbr_column_sum_workspace_bytes(int N) { return N > 0 ? (size_t)kColsumBlocks * N * sizeof(float) : 0; }

int matpbr_column_sum(const float* x, float* out, long M, int N, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || M <= 0 || N <= 0) return MATPBR_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < matpbr_column_sum_workspace_bytes(N)) return MATPBR_ERR_WORKSPACE;
    const long rows_per = (M + kColsumBlocks - 1) / kColsumBlocks;
    const int nblk = (int)((M + rows_per - 1) / rows_per);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_pass1_kernel, dim3((unsigned)nblk), dim3(kBlock), 0, st, x, (float*)workspace, M, N, rows_per);
    hipLaunchKernelGGL(colsum_pass2_kernel, dim3((unsigned)((N + 31) / 32)), dim3(kBlock), 0, st, (const float*)workspace, out,
                       nblk, N);
    return launch_status();
}

int matpbr_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, int step,
                     void* stream) {
    if (!p || !g || !m || !v || n < 0 || step < 1) return MATPBR_ERR_INVALID_ARG;
    if (n == 0) return MATPBR_OK;
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    unsigned blocks = (unsigned)std::min<long>((n + kBlock - 1) / kBlock, 2048);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, p, g, m, v, n, (float)(lr / bc1), beta1, beta2,
                       eps, (float)(1.0 / std::sqrt(bc2)));
    return launch_status();
}

}  // extern "C"
