// matpbr_shade.hpp -- image kernels of the render R(a, r, m, n; light), gfx950, wave64, fp32.
//
// The estimator integrates the two lobes of MatDiffBSDF.eval_brdf (myutils/mi_plugin.py:1405-1415) separately, each with its own
// sampler of sample_brdf (:1296-1341):
//   diffuse  (cosine-weighted directions, mi_diffuse_sampler :255-281):  a (1-m) (A0 + r A1 + r^2 A2), a quadratic in r whose
//            coefficients depend on (n, wo, light) only -- constants of a BRDF phase, where light and geometric normals are fixed
//            (inverse_img_w_mi.py:317-342); they are computed in-kernel or read from a per-pixel cache (9 floats);
//   specular (GGX half vectors, mi_specular_sampler :217-253): D cancels between value and pdf; what is summed per sample is
//            g L(wi), g = G1(NoL) G1(NoV) NoL VoH / NoH, once plain (S0) and once times (1-VoH)^5 (S1): C0 S0 + (1-C0) S1.
// The material gradients are closed forms of {A, S0, S1, dS0/dr, dS1/dr} ("jac": 9 floats per pixel written by the forward
// pass), so the backward pass of hot loop B is a streaming kernel without any sample loop.
//
// Layout / mapping: one lane owns the two pixels 2q, 2q+1 of the flattened image (every per-pixel value is a genuine register
// pair for v_pk_*_f32, every per-sample value is wave-uniform and arrives as a scalar operand from the kernel-argument
// segment); 256-thread workgroups; blockIdx.y = image of the batch.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "matpbr_device.hpp"

namespace matpbr {

constexpr int kBlock = 256;
constexpr int kNL = kNSH * 3;  // 75 light scalars per image

// ---- quadrature rules (oracle_rule in oracle/matpbr_oracle.c is the specification) -----------------------------------------
// Per lobe a product rule over the sampler's (u0, u1) square: nu Gauss-Legendre rings x nphi equally spaced azimuths, ring k
// rotated by vdC_2(k)/nphi; the specular lobe places its nodes through u0 = 1 - (1-v)^2.  Everything is computed on the host
// in double and travels in the kernel-argument segment (scalar loads -> SGPR operands).
constexpr int kMaxRings = 8, kMaxAz = 8;
struct RuleTable {
    int nu_d, nphi_d, nu_s, nphi_s;
    float4 dring[kMaxRings];         // lz = cos(theta), w (per sample), w (1-lz)^5, 1/lz
    float4 dring2[kMaxRings];        // (1-lz)^5, (1-lz)^4, 0, 0
    float2 daz[kMaxRings][kMaxAz];   // (lx, ly) = sin(theta) (cos phi, sin phi): local direction of the cosine-weighted sample
    float4 sring[kMaxRings];         // u0, 1-u0, w (per sample), 0
    float2 saz[kMaxRings][kMaxAz];   // (cos phi, sin phi) of the half vector
};

// The per-sample table entries (azimuths) are read by every lane ONCE, lane i holding entry [i / kMaxAz][i % kMaxAz] of both
// lobes (kMaxRings * kMaxAz = 64 = one wave), and broadcast per sample with v_readlane: a scalar load per sample would expose
// its latency inside the sample loop (hipcc sinks such a load to its first use).
// MUST be loaded while every lane of the wave is still active (first statement of a kernel): v_readlane reads the register of
// a lane whatever EXEC says, so a lane that left early would otherwise hold garbage.
struct RuleRegs { float2 daz, saz; };
__device__ __forceinline__ void load_rule_regs(const RuleTable& tab, RuleRegs& rr) {
    static_assert(kMaxRings * kMaxAz == 64, "one table entry per lane of a wave64");
    const int lane = threadIdx.x & 63;
    rr.daz = (&tab.daz[0][0])[lane];
    rr.saz = (&tab.saz[0][0])[lane];
}
__device__ __forceinline__ float2 rule_entry(float2 v, int idx) {   // idx wave-uniform
    return make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), idx)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), idx)));
}

inline double vdc2_host(uint32_t i) {
    i = (i << 16) | (i >> 16);
    i = ((i & 0x55555555u) << 1) | ((i & 0xAAAAAAAAu) >> 1);
    i = ((i & 0x33333333u) << 2) | ((i & 0xCCCCCCCCu) >> 2);
    i = ((i & 0x0F0F0F0Fu) << 4) | ((i & 0xF0F0F0F0u) >> 4);
    i = ((i & 0x00FF00FFu) << 8) | ((i & 0xFF00FF00u) >> 8);
    return (double)i * 2.3283064365386963e-10;
}
inline void gauss_legendre01(int n, double* x, double* w) {   // ascending nodes on [0,1], weights sum to 1
    for (int i = 0; i < n; ++i) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 0; j < n; ++j) {
                double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-15) break;
        }
        x[n - 1 - i] = 0.5 * (z + 1.0);
        w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
    }
}
inline void rule_dims(int spp, int lobe, int& nu, int& nphi) {   // spp = 64 -> 5 x 4 specular, 3 x 6 diffuse
    const double q = 0.25 * spp, s = std::sqrt(q);
    if (lobe) {
        nu = (int)std::floor(s + 1.5);
        nphi = std::max(1, (int)std::floor(s + 0.5));
    } else {
        nu = std::max(1, (int)std::floor(0.75 * s + 0.5));
        nphi = 2 * nu;
    }
}
inline bool fill_rule_table(int spp, RuleTable& t) {
    std::memset(&t, 0, sizeof(t));
    rule_dims(spp, 0, t.nu_d, t.nphi_d);
    rule_dims(spp, 1, t.nu_s, t.nphi_s);
    if (t.nu_d > kMaxRings || t.nu_s > kMaxRings || t.nphi_d > kMaxAz || t.nphi_s > kMaxAz) return false;
    double x[kMaxRings], wx[kMaxRings];
    gauss_legendre01(t.nu_d, x, wx);
    for (int k = 0; k < t.nu_d; ++k) {
        const double u0 = x[k], st = std::sqrt(u0), ct = std::sqrt(1.0 - u0), w = wx[k] / t.nphi_d;   // theta = asin(sqrt(u0)), :265
        const double om = 1.0 - ct, p4 = om * om * om * om;
        t.dring[k] = make_float4((float)ct, (float)w, (float)(w * p4 * om), (float)(1.0 / ct));
        t.dring2[k] = make_float4((float)(p4 * om), (float)p4, 0.0f, 0.0f);
        const double off = vdc2_host((uint32_t)k) / t.nphi_d;
        for (int j = 0; j < t.nphi_d; ++j) {
            double v = (j + 0.5) / t.nphi_d + off;
            const double phi = 2.0 * M_PI * (v - std::floor(v));
            t.daz[k][j] = make_float2((float)(st * std::cos(phi)), (float)(st * std::sin(phi)));
        }
    }
    gauss_legendre01(t.nu_s, x, wx);
    for (int k = 0; k < t.nu_s; ++k) {
        const double v0 = x[k], u0 = 1.0 - (1.0 - v0) * (1.0 - v0), w = wx[k] * 2.0 * (1.0 - v0) / t.nphi_s;
        t.sring[k] = make_float4((float)u0, (float)((1.0 - v0) * (1.0 - v0)), (float)w, 0.0f);
        const double off = vdc2_host((uint32_t)k) / t.nphi_s;
        for (int j = 0; j < t.nphi_s; ++j) {
            double v = (j + 0.5) / t.nphi_s + off;
            const double phi = 2.0 * M_PI * (v - std::floor(v));
            t.saz[k][j] = make_float2((float)std::cos(phi), (float)std::sin(phi));
        }
    }
    return true;
}

struct Geom {
    int H, W;
    float inv_f, cx, cy;
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
__device__ __forceinline__ float block_sum(float v, float* s_buf) {   // all threads get the total (s_buf: 4 floats of LDS)
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) s_buf[wave] = v;
    __syncthreads();
    return (s_buf[0] + s_buf[1]) + (s_buf[2] + s_buf[3]);
}

// ---- per-lane set-up: TWO pixels (flattened indices 2q, 2q+1) in the halves of every f2 -----------------------------------
struct Pixel {
    f2 a[3], r, m;       // (clamped) materials
    f2 n[3], s[3], t[3]; // unit shading normal and its frame ([ext] mi.Frame3f)
    f2 vx, vy, vz;       // view direction in the shading frame; vz = n.wo (unclamped)
    f2 inv_len;          // 1/|n| of the stored normal
    f2 NoV, po;          // max(n.wo, 0), (1-NoV)^5                               (:1394,1407)
    f2 alpha2, am1;      // r^4, r^4 - 1                                          (:93-97)
    f2 omk, kpe, dk_dr;  // 1-k, k+1e-6, dk/dr; k = (r+1)^2/8                     (:64-67)
    f2 g1v;              // G1(NoV)                                               (:1411)
};
struct RawParams { f2 a[3], r, m; };
__device__ __forceinline__ f2 clamp2(f2 x, float lo, float hi) { return vmin(vmax(x, lo), hi); }

// clamp: the maps are the optimiser's raw parameters and the render uses clamp(a,0,1), clamp(r,.07,1), clamp(m,0,1)
// (inverse_img_w_mi.py:371-377); the raw values are returned so the backward pass can gate the gradient like torch.clamp.
__device__ __forceinline__ void load_pixel(Pixel& px, const float* __restrict__ a, const float* __restrict__ r, const float* __restrict__ m,
                                           const float* __restrict__ n, long i0, long i1, int p0, int p1, const Geom& g, bool clamp,
                                           RawParams* raw = nullptr) {
    f2 nv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        px.a[c] = f2{a[i0 * 3 + c], a[i1 * 3 + c]};
        nv[c] = f2{n[i0 * 3 + c], n[i1 * 3 + c]};
    }
    px.r = f2{r[i0], r[i1]};
    px.m = f2{m[i0], m[i1]};
    if (raw) {
#pragma unroll
        for (int c = 0; c < 3; ++c) raw->a[c] = px.a[c];
        raw->r = px.r; raw->m = px.m;
    }
    if (clamp) {
#pragma unroll
        for (int c = 0; c < 3; ++c) px.a[c] = clamp2(px.a[c], 0.0f, 1.0f);
        px.r = clamp2(px.r, 0.07f, 1.0f);
        px.m = clamp2(px.m, 0.0f, 1.0f);
    }
    // shading normal = normalize(n map); the geometric normals and MaterialNet's are unit already
    px.inv_len = rsq(vmax(dot3v(nv, nv), 1e-30f));
#pragma unroll
    for (int c = 0; c < 3; ++c) px.n[c] = nv[c] * px.inv_len;
    // view direction of pixel (i,j): wo = -p/|p|, p = ((j-cx)/f, -(i-cy)/f, -1)   
    f2 fi = f2{(float)(p0 / g.W), (float)(p1 / g.W)}, fj = f2{(float)(p0 % g.W), (float)(p1 % g.W)};
    f2 x = (g.cx - fj) * g.inv_f, y = (fi - g.cy) * g.inv_f;
    f2 il = rsq(vfma(x, x, vfma(y, y, 1.0f)));
    f2 wo[3] = {x * il, y * il, il};
    frame(px.n, px.s, px.t);
    px.vx = dot3v(px.s, wo);
    px.vy = dot3v(px.t, wo);
    px.vz = dot3v(px.n, wo);
    px.NoV = vmax(px.vz, 0.0f);
    px.po = pow5(1.0f - px.NoV);
    px.alpha2 = pow4(px.r);
    px.am1 = px.alpha2 - 1.0f;
    f2 rp1 = px.r + 1.0f;
    f2 k = (rp1 * rp1) * 0.125f;
    px.omk = 1.0f - k;
    px.kpe = k + 1e-6f;
    px.dk_dr = rp1 * 0.25f;
    px.g1v = rcp(vfma(px.NoV, px.omk, px.kpe));
}

// ---- SH coefficients: 75 wave-uniform scalars kept in 38 VGPR pairs ------------------------------------------
// A packed FMA needs the scalar c'[k][c] in both halves of a 64-bit operand.  hipcc materialises such a splat with
// a v_mov per use (VGPR) or spills the SGPR file (75 live scalars + rule table > 102 SGPRs), so the broadcast
// is spelled out: VOP3P op_sel/op_sel_hi pick the low or the high half of a register pair for BOTH lanes of the
// packed operation, letting one pair carry two different coefficients at zero extra instructions.
constexpr int kNPairs = (kNL + 1) / 2;
struct LightRegs { f2 c[kNPairs]; };

// c'[k][c] = coefficient * basis normalisation, so that the per-sample radiance is 72 FMAs on raw polynomials
// `light` must be a `const float* __restrict__` kernel parameter: the 75 wave-uniform reads then become scalar loads (five
// s_load_dwordx16) instead of 38 vector loads, and all of them are issued before the first value is pinned.
__device__ __forceinline__ void load_light_regs(LightRegs& lr, const float* __restrict__ light) {
    float raw[kNL + 1];
#pragma unroll
    for (int q = 0; q < kNL; ++q) raw[q] = light[q];
    raw[kNL] = raw[kNL - 1];
#pragma unroll
    for (int j = 0; j < kNPairs; ++j) {
        const int q0 = 2 * j, q1 = 2 * j + 1 < kNL ? 2 * j + 1 : 2 * j;
        lr.c[j] = f2{raw[q0] * kShNorm[q0 / 3], raw[2 * j + 1] * kShNorm[q1 / 3]};
    }
#pragma unroll
    for (int j = 0; j < kNPairs; ++j) asm volatile("" : "+v"(lr.c[j]));  // pin in VGPRs for the whole kernel
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
// the two lobes
// =================================================================================================
struct DiffuseCoef { f2 A0[3], A1[3], A2[3]; };

// A0, A1, A2 of the diffuse lobe: moments of the radiance over the cosine-weighted directions
//   M0 = sum w L, M1 = sum w p5 L, M2 = sum w u L, M3 = sum w u p5 L, M4 = sum w u^2 p5 L;  u = 1 + wi.wo, p5 = (1 - NoL)^5
//   F_out F_in = (1 + q po)(1 + q p5), q = r u - 1/2  =>  A0 = M0 (1 - po/2) + M1 (po/4 - 1/2), A1 = po M2 + (1-po) M3, A2 = po M4
__device__ __forceinline__ void diffuse_coef(const Pixel& px, const LightRegs& lr, const RuleTable& tab, const RuleRegs& rr, DiffuseCoef& A) {
    f2 M[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) M[i][c] = f2{0.0f, 0.0f};
    for (int k = 0; k < tab.nu_d; ++k) {
        const float4 rg = tab.dring[k];
        for (int j = 0; j < tab.nphi_d; ++j) {
            const float2 az = rule_entry(rr.daz, k * kMaxAz + j);
            f2 wi[3], L[3];
            to_world(px.s, px.t, px.n, az.x, az.y, rg.x, wi);
            sh_radiance(lr, wi, L);
            const f2 u = vfma(px.vz, rg.x, vfma(px.vy, az.y, vfma(px.vx, az.x, 1.0f)));
            const f2 t2 = u * rg.y, t3 = u * rg.z, t4 = t3 * u;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                M[0][c] = vfma(L[c], rg.y, M[0][c]);
                M[1][c] = vfma(L[c], rg.z, M[1][c]);
                M[2][c] = vfma(L[c], t2, M[2][c]);
                M[3][c] = vfma(L[c], t3, M[3][c]);
                M[4][c] = vfma(L[c], t4, M[4][c]);
            }
        }
    }
    const f2 e0 = vfma(px.po, -0.5f, 1.0f), e1 = vfma(px.po, 0.25f, -0.5f), ompo = 1.0f - px.po;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A.A0[c] = vfma(M[0][c], e0, M[1][c] * e1);
        A.A1[c] = vfma(M[2][c], px.po, M[3][c] * ompo);
        A.A2[c] = M[4][c] * px.po;
    }
}

// per-ring quantities of the GGX half-vector sampler (mi_specular_sampler :217-253): cos^2 t_h = (1-u0)/(u0(alpha2-1)+1)
struct SpecRing {
    f2 q, ct, st, ringw;   // 1/(u0 (alpha2-1) + 1), cos t_h = NoH, sin t_h, w G1(NoV) / NoH
    f2 idq;                // 1/(q + 1e-6/alpha2): D_GGX's regulariser, den = alpha2 q + 1e-6  (:95)
};
template <bool WANT_IDQ>
__device__ __forceinline__ void spec_ring(const Pixel& px, const float4 rg, SpecRing& R) {
    R.q = rcp(vfma(px.am1, rg.x, 1.0f));
    const f2 cos2 = R.q * rg.y;
    const f2 sin2 = (px.alpha2 * rg.x) * R.q;   // 1 - cos2 without cancellation
    const f2 ict = rsq(cos2);
    R.ct = cos2 * ict;
    R.st = sin2 * rsq(sin2);
    R.ringw = (px.g1v * rg.z) * ict;
    if (WANT_IDQ) R.idq = rcp(R.q + 1e-6f * rcp(px.alpha2));
}
struct SpecSample {
    f2 whx, why;          // half vector, tangential components (its normal component is R.ct)
    f2 d;                 // wo.wh
    f2 wlx, wly, wlz;     // wi = reflect(wo, wh) in the shading frame; wlz = n.wi
    f2 wi[3];
    f2 NoL, dpos, g1l, x5;
};
__device__ __forceinline__ void spec_sample(const Pixel& px, const SpecRing& R, const float2 az, SpecSample& sm) {
    sm.whx = R.st * az.x; sm.why = R.st * az.y;
    sm.d = vfma(R.ct, px.vz, vfma(sm.why, px.vy, sm.whx * px.vx));
    const f2 d2 = sm.d + sm.d;
    sm.wlx = vfma(d2, sm.whx, -px.vx); sm.wly = vfma(d2, sm.why, -px.vy); sm.wlz = vfma(d2, R.ct, -px.vz);   // 2 (wo.wh) wh - wo  (:245)
    to_world(px.s, px.t, px.n, sm.wlx, sm.wly, sm.wlz, sm.wi);
    sm.NoL = vmax(sm.wlz, 0.0f);     // a sample below the horizon (or with wo.wh <= 0) carries weight 0
    sm.dpos = vmax(sm.d, 0.0f);      // VoH
    sm.g1l = rcp(vfma(sm.NoL, px.omk, px.kpe));
    sm.x5 = pow5(1.0f - sm.dpos);
}

template <bool JAC>
struct SpecAcc { f2 S0[3], S1[3], dS0[JAC ? 3 : 1], dS1[JAC ? 3 : 1]; };

template <bool JAC>
__device__ __forceinline__ void spec_accumulate(const Pixel& px, const LightRegs& lr, const RuleTable& tab, const RuleRegs& rr,
                                                SpecAcc<JAC>& A) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A.S0[c] = A.S1[c] = f2{0.0f, 0.0f};
        if (JAC) A.dS0[c] = A.dS1[c] = f2{0.0f, 0.0f};
    }
    const f2 four_over_r = 4.0f * rcp(px.r);
    const f2 cv = px.dk_dr * px.g1v * (1.0f - px.NoV);
    for (int k = 0; k < tab.nu_s; ++k) {
        const float4 rg = tab.sring[k];
        SpecRing R;
        spec_ring<JAC>(px, rg, R);
        f2 lam0 = f2{0.0f, 0.0f};
        // d ln D/dr at a fixed direction = 4/r - 8 r^3 NoH^2/den = (4/r)(1 - 2 (1-u0) q/(q + 1e-6/alpha2)) on a GGX-sampled half vector
        if (JAC) lam0 = vfma(four_over_r, vfma(R.q * R.idq, -2.0f * rg.y, 1.0f), -cv);
        for (int j = 0; j < tab.nphi_s; ++j) {
            SpecSample sm;
            spec_sample(px, R, rule_entry(rr.saz, k * kMaxAz + j), sm);
            const f2 wgt = (R.ringw * sm.g1l) * (sm.NoL * sm.dpos);   // f cos / pdf without F_m: G1(NoL) G1(NoV) NoL VoH / NoH
            f2 L[3];
            sh_radiance(lr, sm.wi, L);
            const f2 wx = wgt * sm.x5;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                A.S0[c] = vfma(wgt, L[c], A.S0[c]);
                A.S1[c] = vfma(wx, L[c], A.S1[c]);
            }
            if (JAC) {
                const f2 lam = vfma(px.dk_dr * sm.g1l, sm.NoL - 1.0f, lam0);   // + d ln G/dr = -dk/dr (G1l (1-NoL) + G1v (1-NoV))
                const f2 wl = wgt * lam, wlx = wl * sm.x5;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    A.dS0[c] = vfma(wl, L[c], A.dS0[c]);
                    A.dS1[c] = vfma(wlx, L[c], A.dS1[c]);
                }
            }
        }
    }
}

// statistics layout used by the fused optimisation steps (include/matpbr.h MATPBR_STATS_STRIDE)
constexpr int kStatsStride = 16;
enum { kStRatio = 0, kStMse, kStL1, kStSr, kStLa, kStLr, kStLm, kStLoss, kStImproved, kStBest, kStEsCounter, kStEsBest, kStEsHas,
       kStStopped, kStIters, kStGtSum };
// The step kernel's private copy of a row (two alternating copies in the phase workspace) is four floats longer: with
// MATPBR_FLAG_ROTATE_BEST the parameters of the live maps and the render live in two buffers each (the caller's tensor and its SaveBest
// target) and kStSel says which of the two holds the CURRENT values (the other one holds the best so far, or nothing yet); kStBestRatio is
// the exposure ratio of the best iteration (< 0: no improvement in this phase yet), from which matpbr_brdf_phase_resolve forms best_img.
constexpr int kStateStride = 20;
enum { kStSel = 16, kStBestRatio = 17, kStSelOld = 18 /* LDS broadcast only */ };
// kStStopped: 0 running; 1 = EarlyStopping fired in this iteration (its backward / optimiser step still run, as in the reference's
// loop, which breaks after optimizer.step()); 2 = stopped in an earlier iteration (every kernel skips the image).
__device__ __forceinline__ bool img_stopped(const float* stats, int b) { return stats[b * kStatsStride + kStStopped] > 0.5f; }
__device__ __forceinline__ bool img_stopped_before(const float* stats, int b) { return stats[b * kStatsStride + kStStopped] > 1.5f; }

// Per-image scalars of the iteration, SaveBest's decision, and (es_patience > 0) the EarlyStopping state machine of
// myutils/misc.py:37-60 kept on the device: once an image has stopped, every later kernel of the fused step skips it, so
// the host may enqueue iterations ahead and read the flag occasionally without changing any decision.
// The iteration in which EarlyStopping fires still finishes (snapshot of a new best, optimiser step), as in the reference's
// loop, which tests early_stop after optimizer.step(): kStStopped goes 0 -> 1 here and 1 -> 2 on the next entry.
__device__ __forceinline__ bool stats_enter(float* st) {   // true: the image stopped earlier, nothing to do
    if (st[kStStopped] > 0.5f) {
        st[kStStopped] = 2.0f;
        st[kStImproved] = 0.0f;
        return true;
    }
    return false;
}
__device__ __forceinline__ void stats_commit(float* st, float mse, float l1, float sr, float la, float lr, float lm, float scale_delta,
                                             int es_patience, float es_min_delta, float* history, int hist_len, int batch, int b) {
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
// rows of partial sums the step kernel folds itself (one per workgroup of loss_sums2_kernel<3>), and the stride of an image's block
// One value for every batch size, so that an image's statistics do not depend on how many images run beside it (batch = stand-alone
// bit for bit).  Measured: one 512 x 512 image 128 / 256 / 384 / 768 rows -> 25.9 / 31.0 / 33.3 / 34.1 k it/s (the pass over pred is
// latency-bound: more rows win); eight images 96 / 192 / 256 / 384 / 768 -> 54.5 / 59.7 / 61.3 / 59.7 / 57.5 k image-it/s (every
// workgroup of the step kernel folds its image's rows: fewer rows win).
constexpr int kStepRows = 384;
__host__ __device__ inline long step_part_stride(int rows) { return (long)rows * 5 + 4; }

// =================================================================================================
// forward (+ jac) and the material backward of the operator face
// =================================================================================================
// jac planes [9][B*P]: 0-2 P_c = A0 + r A1 + r^2 A2, 3-5 SD_c = S0_c - S1_c, 6-8 JR_c = d out_c / d r
//   d out_c / d a_c = (1-m) P_c + m SD_c;   d out_c / d m = -a_c P_c + (a_c - 0.04) SD_c
// dcache planes [9][B*P]: A0 rgb, A1 rgb, A2 rgb
struct ShadeArgs {
    const float *a, *r, *m, *n;
    const float* dcache;      // nullable: diffuse coefficients computed in-kernel
    float* out;               // [B,H,W,3] linear radiance (forward)
    float* jac;               // nullable
    const float* d_out;       // non-null: material backward (d_a, d_r, d_m written instead of out)
    float *d_a, *d_r, *d_m;
    const float* stats;       // nullable: skip images whose EarlyStopping has fired
    float* block_sums;        // nullable: per-workgroup sum of the rendered rgb (for mean(pred), :388)
    int clamp;
    float* s1;                // nullable [3][B*P]: the specular sum S1 of this render (written by shade_kernel, read by shade_cached_kernel)
};

template <bool JAC>
__global__ __launch_bounds__(kBlock, 2) void shade_kernel(const ShadeArgs q, const float* __restrict__ light, const Geom g,
                                                          const RuleTable tab) {
    __shared__ float s_sum[4];
    RuleRegs rr;
    load_rule_regs(tab, rr);
    const int b = blockIdx.y;
    if (q.stats && img_stopped(q.stats, b)) return;
    const int P = g.H * g.W;
    const long BP = (long)gridDim.y * P;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    // (no early exit of the lanes beyond the image: the sample loop reads the rule table out of EVERY lane's registers, and hipcc is free to
    // sink their loads below a divergent return -- the lanes that left would then never have loaded their entries)
    const bool act0 = q0 < P, two = q0 + 1 < P;
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    Pixel px;
    load_pixel(px, q.a, q.r, q.m, q.n, i0, i1, p0, p1, g, q.clamp != 0);
    LightRegs lr;
    load_light_regs(lr, light + (long)b * kNL);

    DiffuseCoef A;
    if (q.dcache) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A.A0[c] = f2{q.dcache[c * BP + i0], q.dcache[c * BP + i1]};
            A.A1[c] = f2{q.dcache[(3 + c) * BP + i0], q.dcache[(3 + c) * BP + i1]};
            A.A2[c] = f2{q.dcache[(6 + c) * BP + i0], q.dcache[(6 + c) * BP + i1]};
        }
    } else {
        diffuse_coef(px, lr, tab, rr, A);
    }
    SpecAcc<JAC> S;
    spec_accumulate<JAC>(px, lr, tab, rr, S);

    const f2 omm = 1.0f - px.m;
    f2 rgb[3], Pc[3], SD[3], JR[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f2 C0 = vfma(px.m, px.a[c], omm * 0.04f);                               // :1412
        Pc[c] = vfma(vfma(A.A2[c], px.r, A.A1[c]), px.r, A.A0[c]);
        rgb[c] = vfma(px.a[c] * omm, Pc[c], vfma(C0, S.S0[c] - S.S1[c], S.S1[c]));
        if (JAC) {
            SD[c] = S.S0[c] - S.S1[c];
            const f2 dP = vfma(2.0f * px.r, A.A2[c], A.A1[c]);
            JR[c] = vfma(px.a[c] * omm, dP, vfma(C0, S.dS0[c] - S.dS1[c], S.dS1[c]));
        }
    }
    if (JAC && q.d_out) {   // material backward
        f2 dr = f2{0.0f, 0.0f}, dm = f2{0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f2 go = f2{act0 ? q.d_out[i0 * 3 + c] : 0.0f, two ? q.d_out[i1 * 3 + c] : 0.0f};
            const f2 da = go * vfma(px.m, SD[c], omm * Pc[c]);
            dm = vfma(go, vfma(px.a[c] - 0.04f, SD[c], -(px.a[c] * Pc[c])), dm);
            dr = vfma(go, JR[c], dr);
            if (act0) q.d_a[i0 * 3 + c] = da.x;
            if (two) q.d_a[i1 * 3 + c] = da.y;
        }
        if (act0) { q.d_r[i0] = dr.x; q.d_m[i0] = dm.x; }
        if (two) { q.d_r[i1] = dr.y; q.d_m[i1] = dm.y; }
        return;
    }
    float tot = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (act0) q.out[i0 * 3 + c] = rgb[c].x;
        if (two) q.out[i1 * 3 + c] = rgb[c].y;
        tot += (act0 ? rgb[c].x : 0.0f) + (two ? rgb[c].y : 0.0f);
        if (JAC && q.s1) {
            if (act0) q.s1[c * BP + i0] = S.S1[c].x;
            if (two) q.s1[c * BP + i1] = S.S1[c].y;
        }
        if (JAC && q.jac) {
            if (act0) { q.jac[c * BP + i0] = Pc[c].x; q.jac[(3 + c) * BP + i0] = SD[c].x; q.jac[(6 + c) * BP + i0] = JR[c].x; }
            if (two) { q.jac[c * BP + i1] = Pc[c].y; q.jac[(3 + c) * BP + i1] = SD[c].y; q.jac[(6 + c) * BP + i1] = JR[c].y; }
        }
    }
    if (q.block_sums) {
        tot = wave_sum_to_lane63(tot);
        if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) q.block_sums[(long)b * gridDim.x + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}

// The render of a part that leaves the roughness alone ('a', 'm', 'am' of --opt_order): with r, the normals and the light fixed,
// P = A0 + r A1 + r^2 A2, SD = S0 - S1 and S1 of every pixel are constants of the part, kept from its first render (jac planes 0-5,
// s1 planes), and out = a (1-m) P + (C0 SD + S1), C0 = 0.04 (1-m) + a m -- the same fused operations in the same order as
// shade_kernel, so the result is bit-identical to walking the 20 samples again.  44 + 36 B/pixel, no samples: HBM-bound.
__global__ __launch_bounds__(kBlock) void shade_cached_kernel(const ShadeArgs q, const Geom g) {
    __shared__ float s_sum[4];
    const int b = blockIdx.y;
    if (q.stats && img_stopped(q.stats, b)) return;
    const int P = g.H * g.W;
    const long BP = (long)gridDim.y * P;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool act0 = q0 < P, two = q0 + 1 < P;
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    f2 a[3], m = f2{q.m[i0], q.m[i1]};
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = f2{q.a[i0 * 3 + c], q.a[i1 * 3 + c]};
    if (q.clamp) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = clamp2(a[c], 0.0f, 1.0f);
        m = clamp2(m, 0.0f, 1.0f);
    }
    const f2 omm = 1.0f - m;
    float tot = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f2 Pc = f2{q.jac[c * BP + i0], q.jac[c * BP + i1]};
        const f2 SD = f2{q.jac[(3 + c) * BP + i0], q.jac[(3 + c) * BP + i1]};
        const f2 S1 = f2{q.s1[c * BP + i0], q.s1[c * BP + i1]};
        const f2 C0 = vfma(m, a[c], omm * 0.04f);
        const f2 rgb = vfma(a[c] * omm, Pc, vfma(C0, SD, S1));
        if (act0) q.out[i0 * 3 + c] = rgb.x;
        if (two) q.out[i1 * 3 + c] = rgb.y;
        tot += (act0 ? rgb.x : 0.0f) + (two ? rgb.y : 0.0f);
    }
    if (q.block_sums) {
        tot = wave_sum_to_lane63(tot);
        if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) q.block_sums[(long)b * gridDim.x + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}

// the diffuse coefficients of every pixel -> dcache planes [9][B*P] (once per BRDF phase)
__global__ __launch_bounds__(kBlock, 2) void diffuse_cache_kernel(const float* __restrict__ n, const float* __restrict__ light,
                                                                  float* __restrict__ dcache, const Geom g, const RuleTable tab) {
    RuleRegs rr;
    load_rule_regs(tab, rr);
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const long BP = (long)gridDim.y * P;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool act0 = q0 < P, two = q0 + 1 < P;            // lanes beyond the image stay (shade_kernel: the rule table lives in every lane)
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    // materials do not enter the coefficients: load the normal only
    Pixel px;
    f2 nv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nv[c] = f2{n[i0 * 3 + c], n[i1 * 3 + c]};
    px.inv_len = rsq(vmax(dot3v(nv, nv), 1e-30f));
#pragma unroll
    for (int c = 0; c < 3; ++c) px.n[c] = nv[c] * px.inv_len;
    f2 fi = f2{(float)(p0 / g.W), (float)(p1 / g.W)}, fj = f2{(float)(p0 % g.W), (float)(p1 % g.W)};
    f2 x = (g.cx - fj) * g.inv_f, y = (fi - g.cy) * g.inv_f;
    f2 il = rsq(vfma(x, x, vfma(y, y, 1.0f)));
    f2 wo[3] = {x * il, y * il, il};
    frame(px.n, px.s, px.t);
    px.vx = dot3v(px.s, wo); px.vy = dot3v(px.t, wo); px.vz = dot3v(px.n, wo);
    px.NoV = vmax(px.vz, 0.0f);
    px.po = pow5(1.0f - px.NoV);
    LightRegs lr;
    load_light_regs(lr, light + (long)b * kNL);
    DiffuseCoef A;
    diffuse_coef(px, lr, tab, rr, A);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (act0) { dcache[c * BP + i0] = A.A0[c].x; dcache[(3 + c) * BP + i0] = A.A1[c].x; dcache[(6 + c) * BP + i0] = A.A2[c].x; }
        if (two) { dcache[c * BP + i1] = A.A0[c].y; dcache[(3 + c) * BP + i1] = A.A1[c].y; dcache[(6 + c) * BP + i1] = A.A2[c].y; }
    }
}

// =================================================================================================
// streaming backward of hot loop B: material gradients from the forward's jac planes
// =================================================================================================
// FUSED (`model_name == 'none'`, inverse_img_w_mi.py:371-432): the kernel takes the optimiser's raw parameter maps, forms
// d loss / d pred itself from the forward image, the gamma-2.2 target and the per-image statistics, adds the L1 regularisers
// towards the initial maps, gates everything like torch.clamp's backward, snapshots the clamped maps and the gamma-2.2 render
// when the statistics say this iteration is the best so far (SaveBest without a host round trip) and applies the Adam update
// (torch.optim.Adam, :359,429) -- one pass over the maps, no gradient round trip through HBM unless d_* are requested.
struct JacBwdArgs {
    const float *a, *r, *m;   // maps the forward rendered (FUSED: raw parameters, clamped here like the forward did)
    const float* jac;
    const float* d_out;       // !FUSED: upstream gradient [B,H,W,3]
    float *d_a, *d_r, *d_m;   // FUSED: nullable
    // FUSED:
    const float *pred, *gt_srgb, *stats;
    const float *a0, *r0, *m0;
    float *best_a, *best_r, *best_m, *best_img;   // nullable
    float *pa, *pr, *pm;      // the same maps as a, r, m, writable (Adam)
    float *am[3], *av[3];     // Adam moments (nullable per map: no update)
    float scale_delta, inv_n3, inv_n1;
    unsigned part_mask;
    float lr_over_bc1, b1, b2, eps, inv_sqrt_bc2;
    int check_stop;
};
constexpr float kLossEps = 1e-8f;  // materialist_amd/loss.py _EPS: x^(1/2.2) has no gradient at exact zeros
__device__ __forceinline__ float fsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float pow_inv_gamma(float x) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.2f)); }
__device__ __forceinline__ float adam_update(float p, float gi, float* m, float* v, long i, const JacBwdArgs& q) {
    const unsigned off = (unsigned)i * 4u;     // uniform base + 32-bit lane offset (byte offsets below 2^32)
    float* mp = (float*)((char*)m + off);
    float* vp = (float*)((char*)v + off);
    const float mi = fmaf(q.b1, *mp, (1.0f - q.b1) * gi);
    const float vi = fmaf(q.b2, *vp, (1.0f - q.b2) * gi * gi);
    *mp = mi; *vp = vi;
    return p - q.lr_over_bc1 * mi / fmaf(fsqrt(vi), q.inv_sqrt_bc2, q.eps);
}

// J16: `jac` holds the five half-precision planes of the lazy forward (matpbr_lazy.hpp): half2 (P_c, SD_c) x 3, half2 (JR_0, JR_1), half2 (JR_2, 0)
typedef _Float16 jac_h2 __attribute__((ext_vector_type(2)));
template <bool FUSED, bool J16 = false>
__global__ __launch_bounds__(kBlock) void jac_bwd_kernel(const JacBwdArgs q, long P) {
    const int b = blockIdx.y;
    if (FUSED && q.check_stop && img_stopped_before(q.stats, b)) return;
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const long BP = (long)gridDim.y * P, i = (long)b * P + p;
    float a[3] = {q.a[i * 3], q.a[i * 3 + 1], q.a[i * 3 + 2]}, r = q.r[i], m = q.m[i];
    float ra[3] = {a[0], a[1], a[2]}, rr = r, rm = m;
    if (FUSED) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = fminf(fmaxf(a[c], 0.0f), 1.0f);
        r = fminf(fmaxf(r, 0.07f), 1.0f);
        m = fminf(fmaxf(m, 0.0f), 1.0f);
    }
    float go[3], xs_keep[3];
    if (FUSED) {
        // d loss / d pred of  3 (l1/mse) mse + l1  on  xs = max(pred*ratio, eps)^(1/2.2)   (:388-418)
        const float ratio = q.stats[b * kStatsStride + kStRatio], sr = q.stats[b * kStatsStride + kStSr];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = q.pred[i * 3 + c] * ratio;
            const float xc = fmaxf(x, kLossEps);
            const float xs = pow_inv_gamma(xc);
            const float d = xs - q.gt_srgb[i * 3 + c];
            const float dxs = x > kLossEps ? xs * rcp(xc) * (1.0f / 2.2f) : 0.0f;
            go[c] = ratio * dxs * fmaf(6.0f * sr, d, fsign(d)) * q.inv_n3;
            xs_keep[c] = xs;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) go[c] = q.d_out[i * 3 + c];
    }
    float da[3], dr = 0.0f, dm = 0.0f;
    const float omm = 1.0f - m;
    float jrv[3] = {0.0f, 0.0f, 0.0f};
    if (J16) {
        const jac_h2 u3 = __builtin_bit_cast(jac_h2, q.jac[3 * BP + i]), u4 = __builtin_bit_cast(jac_h2, q.jac[4 * BP + i]);
        jrv[0] = (float)u3.x; jrv[1] = (float)u3.y; jrv[2] = (float)u4.x;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float Pc, SD, JR;
        if (J16) {
            const jac_h2 u = __builtin_bit_cast(jac_h2, q.jac[c * BP + i]);
            Pc = (float)u.x; SD = (float)u.y; JR = jrv[c];
        } else {
            Pc = q.jac[c * BP + i]; SD = q.jac[(3 + c) * BP + i]; JR = q.jac[(6 + c) * BP + i];
        }
        da[c] = go[c] * fmaf(m, SD, omm * Pc);
        dm = fmaf(go[c], fmaf(a[c] - 0.04f, SD, -(a[c] * Pc)), dm);
        dr = fmaf(go[c], JR, dr);
    }
    if (!FUSED) {
#pragma unroll
        for (int c = 0; c < 3; ++c) q.d_a[i * 3 + c] = da[c];
        q.d_r[i] = dr;
        q.d_m[i] = dm;
        return;
    }
    const bool improved = q.stats[b * kStatsStride + kStImproved] > 0.5f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float gsum = da[c] + ((q.part_mask & MATPBR_PART_A) ? q.scale_delta * q.inv_n3 * fsign(a[c] - q.a0[i * 3 + c]) : 0.0f);   // :398,418
        gsum = (ra[c] >= 0.0f && ra[c] <= 1.0f) ? gsum : 0.0f;                                                               // clamp backward
        if (q.d_a) q.d_a[i * 3 + c] = gsum;
        if (improved && q.best_a) q.best_a[i * 3 + c] = a[c];
        if (improved && q.best_img) q.best_img[i * 3 + c] = xs_keep[c];
        if ((q.part_mask & MATPBR_PART_A) && q.am[0]) q.pa[i * 3 + c] = adam_update(ra[c], gsum, q.am[0], q.av[0], i * 3 + c, q);
    }
    float gr = dr + ((q.part_mask & MATPBR_PART_R) ? q.scale_delta * q.inv_n1 * fsign(r - q.r0[i]) : 0.0f);
    float gm = dm + ((q.part_mask & MATPBR_PART_M) ? q.scale_delta * q.inv_n1 * fsign(m - q.m0[i]) : 0.0f);
    gr = (rr >= 0.07f && rr <= 1.0f) ? gr : 0.0f;
    gm = (rm >= 0.0f && rm <= 1.0f) ? gm : 0.0f;
    if (q.d_r) q.d_r[i] = gr;
    if (q.d_m) q.d_m[i] = gm;
    if (improved && q.best_r) q.best_r[i] = r;
    if (improved && q.best_m) q.best_m[i] = m;
    if ((q.part_mask & MATPBR_PART_R) && q.am[1]) q.pr[i] = adam_update(rr, gr, q.am[1], q.av[1], i, q);
    if ((q.part_mask & MATPBR_PART_M) && q.am[2]) q.pm[i] = adam_update(rm, gm, q.am[2], q.av[2], i, q);
}

// =================================================================================================
// normal / light gradients of the operator face, and the radiance transfer: per-sample weights of both lobes
// (sample directions and pdfs are constants: stop-gradient, as in the reference's torch variants -- `D.data`, `alpha.data`,
// mi_plugin.py:179,366)
// =================================================================================================
struct LightGradUse {   // dc[k][c] += gw[c] . B_k over the lane's two pixels
    const f2* gw;
    float* dc;
    template <int K> __device__ __forceinline__ void operator()(f2 Bk) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dc[K * 3 + c] = fmaf(gw[c].y, Bk.y, fmaf(gw[c].x, Bk.x, dc[K * 3 + c]));
    }
};
template <int K0, int K1>
struct TransferUse {    // acc[(k-K0)*3+c] += w[c] B_k per pixel
    const f2* w;
    f2* acc;
    template <int K> __device__ __forceinline__ void operator()(f2 Bk) {
        if (K >= K0 && K < K1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[(K - K0) * 3 + c] = vfma(w[c], Bk, acc[(K - K0) * 3 + c]);
        }
    }
};

// Walks every sample of both lobes and hands (wi, per-channel weight wc[3] = d out_c / d L_c(wi)) to `use`; with WANT_N it also
// accumulates the gradient w.r.t. the unit normal in the shading frame's tangent plane (dnx, dny) for upstream go[3].
//   d/dn_hat = sum gl wi + gh h + gv wo.  Through n_hat = n/|n| only its tangential part survives, so it is accumulated in the
//   (s,t) plane directly and the (huge, alternating-sign) radial parts of the GGX-peak terms never enter an fp32 sum.
template <bool WANT_N, bool WANT_W, class Use>
__device__ __forceinline__ void walk_samples(const Pixel& px, const LightRegs& lr, const RuleTable& tab, const RuleRegs& rr, const f2 go[3],
                                             Use&& use, f2& dnx, f2& dny) {
    const f2 omm = 1.0f - px.m;
    f2 kd[3], C0[3], ga[3], gC[3], gD[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        kd[c] = px.a[c] * omm;
        C0[c] = vfma(px.m, px.a[c], omm * 0.04f);
        if (WANT_N) { ga[c] = go[c] * kd[c]; gC[c] = go[c] * C0[c]; gD[c] = go[c] - gC[c]; }
    }
    f2 gvd = f2{0.0f, 0.0f}, scs = f2{0.0f, 0.0f};
    // diffuse lobe: weight a (1-m) w F_out F_in
    for (int k = 0; k < tab.nu_d; ++k) {
        const float4 rg = tab.dring[k];
        const float4 rg2 = tab.dring2[k];
        for (int j = 0; j < tab.nphi_d; ++j) {
            const float2 az = rule_entry(rr.daz, k * kMaxAz + j);
            f2 wi[3];
            to_world(px.s, px.t, px.n, az.x, az.y, rg.x, wi);
            const f2 u = vfma(px.vz, rg.x, vfma(px.vy, az.y, vfma(px.vx, az.x, 1.0f)));
            const f2 qq = vfma(px.r, u, -0.5f);                              // F_D90 - 1  (:1406)
            const f2 Fo = vfma(qq, px.po, 1.0f), Fi = vfma(qq, rg2.x, 1.0f);   // :1407-1408
            if (WANT_W) {
                const f2 wff = (Fo * Fi) * rg.y;
                f2 wc[3] = {kd[0] * wff, kd[1] * wff, kd[2] * wff};
                use(wi, wc);
            }
            if (WANT_N) {
                f2 L[3];
                sh_radiance(lr, wi, L);
                const f2 sc = vfma(ga[2], L[2], vfma(ga[1], L[1], ga[0] * L[0])) * rg.y;
                const f2 dFi = (-5.0f * rg2.y) * qq;
                const f2 gl = (sc * Fo) * vfma(Fi, rg.w, dFi);              // d(F_in NoL)/dNoL / NoL_sampled
                dnx = vfma(gl, az.x, dnx);
                dny = vfma(gl, az.y, dny);
                gvd = vfma(sc * qq, Fi, gvd);                               // times -5 (1-NoV)^4 below
            }
        }
    }
    // specular lobe: weight g (C0 + (1-C0)(1-VoH)^5)
    const f2 ia2 = rcp(px.alpha2);
    for (int k = 0; k < tab.nu_s; ++k) {
        const float4 rg = tab.sring[k];
        SpecRing R;
        spec_ring<WANT_N>(px, rg, R);
        f2 gh = f2{0.0f, 0.0f};
        if (WANT_N) gh = (-4.0f * R.ct) * px.am1 * (ia2 * R.idq);           // d ln D/dNoH = -4 NoH (alpha2-1)/den
        for (int j = 0; j < tab.nphi_s; ++j) {
            SpecSample sm;
            spec_sample(px, R, rule_entry(rr.saz, k * kMaxAz + j), sm);
            const f2 wgt0 = (R.ringw * sm.g1l) * sm.dpos;                   // weight / NoL
            if (WANT_W) {
                const f2 wgt = wgt0 * sm.NoL;
                f2 wc[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) wc[c] = wgt * vfma(sm.x5, 1.0f - C0[c], C0[c]);
                use(sm.wi, wc);
            }
            if (WANT_N) {
                f2 L[3];
                sh_radiance(lr, sm.wi, L);
                const f2 sL0 = vfma(gC[2], L[2], vfma(gC[1], L[1], gC[0] * L[0]));
                const f2 sL1 = vfma(gD[2], L[2], vfma(gD[1], L[1], gD[0] * L[0]));
                f2 sc0 = wgt0 * vfma(sm.x5, sL1, sL0);
                sc0 = sel_pos(sm.wlz, sc0, 0.0f);
                const f2 sc = sc0 * sm.NoL;
                const f2 tl = sc0 * vfma(-(sm.NoL * sm.g1l), px.omk, 1.0f);   // sc (1/NoL - G1l (1-k))
                const f2 th = sc * gh;
                dnx = vfma(tl, sm.wlx, vfma(th, sm.whx, dnx));
                dny = vfma(tl, sm.wly, vfma(th, sm.why, dny));
                scs += sc;
            }
        }
    }
    if (WANT_N) {
        // through NoV = max(n.wo, 0): F_out of the diffuse lobe and G1(NoV) of the specular one
        const f2 omv = 1.0f - px.NoV;
        const f2 omv4 = (omv * omv) * (omv * omv);
        f2 gv = vfma(gvd, -5.0f * omv4, -(scs * (px.g1v * px.omk)));
        gv = sel_pos(px.vz, gv, 0.0f);
        dnx = vfma(gv, px.vx, dnx);
        dny = vfma(gv, px.vy, dny);
    }
}

struct NoUse { __device__ __forceinline__ void operator()(const f2*, const f2*) const {} };
struct LightUse {
    float* dc;
    const f2* go;
    __device__ __forceinline__ void operator()(const f2 wi[3], const f2 wc[3]) const {
        f2 gw[3] = {go[0] * wc[0], go[1] * wc[1], go[2] * wc[2]};
        sh_stream(wi, LightGradUse{gw, dc});
    }
};

template <bool WANT_N, bool WANT_LIGHT>
__global__ __launch_bounds__(kBlock, 2) void shade_bwd_nl_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                                 const float* __restrict__ m, const float* __restrict__ n,
                                                                 const float* __restrict__ light, const float* __restrict__ d_out,
                                                                 float* __restrict__ d_n, float* __restrict__ partials, const Geom g,
                                                                 const RuleTable tab) {
    __shared__ float s_red[4][kNL + 1];
    RuleRegs rr;
    load_rule_regs(tab, rr);
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool act0 = q0 < P, two = q0 + 1 < P;
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    Pixel px;
    load_pixel(px, a, r, m, n, i0, i1, p0, p1, g, false);
    LightRegs lr;
    if (WANT_N) load_light_regs(lr, light + (long)b * kNL);
    f2 go[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) go[c] = f2{act0 ? d_out[i0 * 3 + c] : 0.0f, two ? d_out[i1 * 3 + c] : 0.0f};
    f2 dnx = f2{0.0f, 0.0f}, dny = f2{0.0f, 0.0f};
    float dc[WANT_LIGHT ? kNL : 1];
    if (WANT_LIGHT) {
#pragma unroll
        for (int k = 0; k < kNL; ++k) dc[k] = 0.0f;
        walk_samples<WANT_N, true>(px, lr, tab, rr, go, LightUse{dc, go}, dnx, dny);
    } else {
        walk_samples<WANT_N, false>(px, lr, tab, rr, go, NoUse{}, dnx, dny);
    }
    if (WANT_N && act0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f2 v = vfma(px.s[c], dnx, px.t[c] * dny) * px.inv_len;
            d_n[i0 * 3 + c] = v.x;
            if (two) d_n[i1 * 3 + c] = v.y;
        }
    }
    if (WANT_LIGHT) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < kNL; ++k) {
            float v = wave_sum_to_lane63(dc[k]);
            if (lane == 63) s_red[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < kNL) {
            float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
            partials[((long)b * gridDim.x + blockIdx.x) * kNL + threadIdx.x] = v;
        }
    }
}

// d_light[b][k][c] = kShNorm[k] * sum over the image's workgroups of partials (fixed order -> reproducible)
__global__ __launch_bounds__(kBlock) void light_grad_finalize_kernel(const float* __restrict__ partials, float* __restrict__ d_light,
                                                                     int nblocks) {
    __shared__ float s_red[kBlock];
    const int b = blockIdx.y, k = blockIdx.x;  // one workgroup per light scalar
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
// precomputed radiance transfer (hot loop A, inverse_img_w_mi.py:236-254: materials and normals are fixed while the light is
// optimised; forward-only relighting, render_final.py:148-203,300-418)
// The render is linear in the light: R[c] = sum_k light[k][c] * T[k][c] with the per-pixel transfer
//   T[k][c] = kShNorm[k] * sum over the samples of both lobes of  (d out_c / d L_c(wi_s)) B_k(wi_s).
// T is computed once per material state (one pass per group of 9/8/8 basis functions keeps the 27 packed accumulators
// in registers); every render under a new light is then 75 FMAs per pixel over 300 bytes -- HBM-bound.
// =================================================================================================
// Transfer layout: tiles of 256 consecutive pixels, each tile a contiguous [75][256] block (75 KB): a workgroup
// reads one contiguous block with lane-consecutive addresses (coalesced AND page-local; 75 planes 16 MB apart thrash the TLB).
__host__ __device__ inline long transfer_tiles(long P) { return (P + 255) / 256; }
__device__ __forceinline__ long transfer_index(int b, long p, int j, long P) {
    return (((long)b * transfer_tiles(P) + (p >> 8)) * kNL + j) * 256 + (p & 255);
}

template <int K0, int K1>
struct TransferWalk {
    f2* acc;
    __device__ __forceinline__ void operator()(const f2 wi[3], const f2 wc[3]) const { sh_stream(wi, TransferUse<K0, K1>{wc, acc}); }
};

template <int K0, int K1>
__global__ __launch_bounds__(kBlock, 2) void shade_transfer_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                                   const float* __restrict__ m, const float* __restrict__ n,
                                                                   float* __restrict__ T, const Geom g, const RuleTable tab) {
    RuleRegs rr;
    load_rule_regs(tab, rr);
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const int q0 = 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool act0 = q0 < P, two = q0 + 1 < P;            // lanes beyond the image stay (shade_kernel: the rule table lives in every lane)
    const int p0 = act0 ? q0 : P - 1, p1 = two ? q0 + 1 : p0;
    const long i0 = (long)b * P + p0, i1 = (long)b * P + p1;
    Pixel px;
    load_pixel(px, a, r, m, n, i0, i1, p0, p1, g, false);
    constexpr int NK = K1 - K0;
    f2 acc[NK * 3];
#pragma unroll
    for (int k = 0; k < NK * 3; ++k) acc[k] = f2{0.0f, 0.0f};
    LightRegs lr;   // unused (no radiance is evaluated)
    f2 go[3], dnx, dny;
    walk_samples<false, true>(px, lr, tab, rr, go, TransferWalk<K0, K1>{acc}, dnx, dny);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const float sc = kShNorm[K0 + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f2 v = acc[k * 3 + c] * sc;
            const int j = (K0 + k) * 3 + c;
            if (act0) T[transfer_index(b, p0, j, P)] = v.x;
            if (two) T[transfer_index(b, p1, j, P)] = v.y;
        }
    }
}

// out[f][p][c] = sum_k T[p][k][c] * light[f][k][c] for up to NF lights per launch (T read once per launch: 300 B/pixel shared by NF frames of
// 12 B/pixel each -- sixteen frames per pass move 129 MB per 2048 x 2048 frame, eight 207, one 1.31 GB)
constexpr int kRelightFrames = 24;
// the lights are read at wave-uniform addresses: scalar loads, SGPR operands of the FMAs (lights of frames >= n_frames must be readable)
template <int NF>
__global__ __launch_bounds__(kBlock) void relight_kernel(const float* __restrict__ T, const float* __restrict__ L, float* __restrict__ out,
                                                         long P, int n_frames) {
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float acc[NF][3];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f][0] = acc[f][1] = acc[f][2] = 0.0f;
    const float* tp = T + transfer_index(0, p, 0, P);
#pragma unroll 5
    for (int k = 0; k < kNSH; ++k) {
        const float t0 = tp[(k * 3) * 256], t1 = tp[(k * 3 + 1) * 256], t2 = tp[(k * 3 + 2) * 256];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int fi = f < n_frames ? f : 0;     // uniform: stays a scalar load
            acc[f][0] = fmaf(t0, L[fi * kNL + k * 3], acc[f][0]);
            acc[f][1] = fmaf(t1, L[fi * kNL + k * 3 + 1], acc[f][1]);
            acc[f][2] = fmaf(t2, L[fi * kNL + k * 3 + 2], acc[f][2]);
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f < n_frames) {                               // one 12-byte store per lane (global_store_dwordx3)
            struct __attribute__((packed, aligned(4))) Rgb { float x, y, z; };
            *reinterpret_cast<Rgb*>(out + ((long)f * P + p) * 3) = Rgb{acc[f][0], acc[f][1], acc[f][2]};
        }
    }
}

// ---- hot loop A on the transfer: one pass over T per iteration ---------------------------------------------------------------
// pred = T . light, loss = MSE + L1 on x^(1/2.2) (:241-245) and d loss / d light = T^T (d loss / d pred) in the same pass:
// the env loss has no image-wide scale factor, so the per-pixel loss gradient is known the moment the pixel is rendered.
// Per workgroup: kEnvPart floats = 75 light-gradient partials + sum d^2 + sum |d|; env_final_kernel folds them in fixed order.
constexpr int kEnvPart = kNL + 2;
constexpr int kEnvTilesPerBlock = 4;
__global__ __launch_bounds__(kBlock) void env_prt_kernel(const float* __restrict__ T, const float* __restrict__ light,
                                                         const float* __restrict__ gt_srgb, float* __restrict__ pred,
                                                         const float* __restrict__ stats, float* __restrict__ part, long P, float inv_n3) {
    __shared__ float s_red[4][kEnvPart + 1];
    const int b = blockIdx.y;
    if (img_stopped(stats, b)) return;
    const float* __restrict__ L = light + (long)b * kNL;
    const long tiles = transfer_tiles(P);
    float acc[kEnvPart];
#pragma unroll
    for (int k = 0; k < kEnvPart; ++k) acc[k] = 0.0f;
    // the workgroup's four tiles, the NEXT tile's 75 coefficients and target requested before the current one is worked on (two sets of
    // registers, the loop unrolled: no copies) -- a tile was latency + work + issue in turn, 3 us each
    auto request = [&](float (&t)[kNL], float (&g)[3], int it) {
        const long tile = (long)blockIdx.x * kEnvTilesPerBlock + it;
        const long tc = tile < tiles ? tile : tiles - 1;
        const long p = tc * 256 + threadIdx.x, pc = p < P ? p : P - 1;
        const float* tp = T + (((long)b * tiles + tc) * kNL) * 256 + threadIdx.x;
#pragma unroll
        for (int j = 0; j < kNL; ++j) t[j] = tp[j * 256];
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = gt_srgb[((long)b * P + pc) * 3 + c];
    };
    auto work = [&](const float (&t)[kNL], const float (&g)[3], int it) {
        const long tile = (long)blockIdx.x * kEnvTilesPerBlock + it;
        const long p = tile * 256 + threadIdx.x;
        if (tile >= tiles || p >= P) return;
        float x[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < kNSH; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = fmaf(t[k * 3 + c], L[k * 3 + c], x[c]);
        }
        const long i = (long)b * P + p;
        float gq[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (pred) pred[i * 3 + c] = x[c];
            const float xc = fmaxf(x[c], kLossEps);
            const float xs = pow_inv_gamma(xc);
            const float d = xs - g[c];
            const float dxs = x[c] > kLossEps ? xs * rcp(xc) * (1.0f / 2.2f) : 0.0f;
            gq[c] = dxs * fmaf(2.0f, d, fsign(d)) * inv_n3;     // d (MSE + L1) / d pred
            acc[kNL] = fmaf(d, d, acc[kNL]);
            acc[kNL + 1] += fabsf(d);
        }
#pragma unroll
        for (int j = 0; j < kNL; ++j) acc[j] = fmaf(t[j], gq[j % 3], acc[j]);
    };
    static_assert(kEnvTilesPerBlock == 4, "the loop below is written out for four tiles");
    float ta[kNL], tb[kNL], ga[3], gb[3];
    request(ta, ga, 0);
    request(tb, gb, 1);
    __builtin_amdgcn_sched_barrier(0);
    work(ta, ga, 0);
    request(ta, ga, 2);
    __builtin_amdgcn_sched_barrier(0);
    work(tb, gb, 1);
    request(tb, gb, 3);
    __builtin_amdgcn_sched_barrier(0);
    work(ta, ga, 2);
    work(tb, gb, 3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kEnvPart; ++k) {
        float v = wave_sum_to_lane63(acc[k]);
        if (lane == 63) s_red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kEnvPart) {
        float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
        part[((long)b * gridDim.x + blockIdx.x) * kEnvPart + threadIdx.x] = v;
    }
}

}  // namespace matpbr
