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
// u0_i = (i+.5)/n, u1_i = vdC2(i) + .5/m.  Values are computed on the host in double.
struct SampleTable {
    float4 diff[MATPBR_MAX_SPP / 2];  // local direction (x,y,z) of the cosine-weighted sample, (1-z)^5
    float4 spec[MATPBR_MAX_SPP / 2];  // u0, cos(phi), sin(phi), 1-u0 of the GGX half-vector sample
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
        t.diff[i] = make_float4((float)(st * std::cos(phi)), (float)(st * std::sin(phi)), (float)ct, (float)std::pow(1.0 - ct, 5.0));
        t.spec[i] = make_float4((float)u0, (float)std::cos(phi), (float)std::sin(phi), (float)(1.0 - u0));
    }
}

struct Geom {
    int H, W, half;
    float inv_f, cx, cy, inv_spp;
};

// ---- wave64 sum with DPP: row_shr 1,2,3 / row_shr 4 / row_shr 8 -> row totals in lane 15 of each row of 16,
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

// Per-pixel set-up shared by forward and backward.
struct Pixel {
    PixelConst pc;
    float n[3], wo[3], s[3], t[3];
    float vx, vy;   // view direction in the shading frame (its z component is pc.NoV_raw)
    float inv_len;  // 1/|n| of the stored normal
};

__device__ __forceinline__ void load_pixel(Pixel& px, const float* __restrict__ a, const float* __restrict__ r,
                                           const float* __restrict__ m, const float* __restrict__ n, long idx, int i, int j,
                                           const Geom& g) {
    float av[3] = {a[idx * 3], a[idx * 3 + 1], a[idx * 3 + 2]};
    float nv[3] = {n[idx * 3], n[idx * 3 + 1], n[idx * 3 + 2]};
    float rv = r[idx], mv = m[idx];
    // shading normal = normalize(n map); the geometric normals and MaterialNet's are unit already
    px.inv_len = rsq(fmaxf(dot3(nv, nv), 1e-30f));
#pragma unroll
    for (int c = 0; c < 3; ++c) px.n[c] = nv[c] * px.inv_len;
    view_dir(i, j, g.inv_f, g.cx, g.cy, px.wo);
    frame(px.n, px.s, px.t);
    px.vx = dot3(px.s, px.wo);
    px.vy = dot3(px.t, px.wo);
    pixel_const(px.pc, av, rv, mv, dot3(px.n, px.wo));
}

// One sample of the deterministic estimator: direction wi, the cosines and the GGX denominator.
struct Sample {
    float wi[3];
    float lwx, lwy, lhx, lhy;  // tangential (shading-frame x,y) components of wi and of the half vector h
    float NoL_raw, NoH, VoH, den;
    bool nh_pos;
};

template <bool WANT_H>
__device__ __forceinline__ void diffuse_sample(const Pixel& px, const float4 tab, Sample& sm) {
    // mi_diffuse_sampler (mi_plugin.py:255-281): local (sin t cos p, sin t sin p, cos t) -> Frame3f(n).to_world
    to_world(px.s, px.t, px.n, tab.x, tab.y, tab.z, sm.wi);
    sm.NoL_raw = tab.z;  // n.wi for an orthonormal frame
    float wiwo = dot3(sm.wi, px.wo);
    float il = rsq(fmaxf(fmaf(2.0f, wiwo, 2.0f), 1e-30f));  // 1/|wi+wo|
    sm.VoH = fmaxf((1.0f + wiwo) * il, 0.0f);
    float nh = (tab.z + px.pc.NoV_raw) * il;
    sm.NoH = fmaxf(nh, 0.0f);
    sm.nh_pos = nh > 0.0f;
    // 1 - NoH^2 = h_x^2 + h_y^2 in the shading frame: no cancellation when the sample lands on the GGX peak
    float hx = tab.x + px.vx, hy = tab.y + px.vy;
    float sin2 = sm.nh_pos ? fminf(fmaf(hx, hx, hy * hy) * il * il, 1.0f) : 1.0f;
    sm.den = ggx_den(px.pc, sm.NoH, sin2);
    if (WANT_H) {
        sm.lwx = tab.x; sm.lwy = tab.y;
        sm.lhx = hx * il; sm.lhy = hy * il;
    }
}

template <bool WANT_H>
__device__ __forceinline__ void specular_sample(const Pixel& px, const float4 tab, Sample& sm) {
    // mi_specular_sampler (mi_plugin.py:217-253): cos^2 t_h = (1-u0)/(u0(alpha2-1)+1), wi = reflect(wo, wh)
    float q = rcp(fmaf(tab.x, px.pc.am1, 1.0f));
    float cos2 = fmaxf(tab.w * q, 0.0f);
    float sin2 = fmaxf(tab.x * px.pc.alpha2 * q, 0.0f);  // 1 - cos2 without cancellation
    float ct = fsqrt(cos2), st = fsqrt(sin2);
    float wh[3];
    const float whx = st * tab.y, why = st * tab.z;
    to_world(px.s, px.t, px.n, whx, why, ct, wh);
    float d = dot3(px.wo, wh);
#pragma unroll
    for (int c = 0; c < 3; ++c) sm.wi[c] = fmaf(2.0f * d, wh[c], -px.wo[c]);
    sm.NoL_raw = fmaf(2.0f * d, ct, -px.pc.NoV_raw);  // n.wi
    sm.VoH = fabsf(d);                                 // wo.h with h = sign(d) wh
    bool front = d > 0.0f;
    sm.NoH = front ? ct : 0.0f;
    sm.nh_pos = front && ct > 0.0f;
    sm.den = front ? ggx_den(px.pc, ct, sin2) : 1.0f + 1e-6f;
    if (WANT_H) {
        float sg = front ? 1.0f : -1.0f;
        sm.lhx = sg * whx; sm.lhy = sg * why;
        sm.lwx = fmaf(2.0f * d, whx, -px.vx); sm.lwy = fmaf(2.0f * d, why, -px.vy);
    }
}

__device__ __forceinline__ void stage_light(float* s_c, const float* __restrict__ light, int b) {
    if (threadIdx.x < kNL) s_c[threadIdx.x] = light[(long)b * kNL + threadIdx.x] * kShNorm[threadIdx.x / 3];
    __syncthreads();
}

// =================================================================================================
// forward
// =================================================================================================
__global__ __launch_bounds__(kBlock) void shade_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                           const float* __restrict__ m, const float* __restrict__ n,
                                                           const float* __restrict__ light, float* __restrict__ out,
                                                           const Geom g, const SampleTable tab) {
    __shared__ float s_c[kNL + 1];
    const int b = blockIdx.y;
    stage_light(s_c, light, b);
    const int P = g.H * g.W;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const long idx = (long)b * P + p;
    Pixel px;
    load_pixel(px, a, r, m, n, idx, p / g.W, p % g.W, g);

    float acc[3] = {0.0f, 0.0f, 0.0f};
    const int spp = 2 * g.half;
#pragma unroll 2
    for (int s = 0; s < spp; ++s) {
        Sample sm;
        if (s < g.half) diffuse_sample<false>(px, tab.diff[s], sm);
        else specular_sample<false>(px, tab.spec[s - g.half], sm);
        BrdfState st;
        float f[3], pdf;
        brdf_core(px.pc, sm.NoL_raw, sm.NoH, sm.VoH, sm.den, st, f, pdf);
        // sample_brdf weight (mi_plugin.py:1335-1339): f/(pdf+1e-6) where pdf > 1e-6
        float ip = pdf > 1e-6f ? rcp(pdf + 1e-6f) : 0.0f;
        float B[kNSH];
        sh_poly(sm.wi, B);
        float L[3] = {s_c[0], s_c[1], s_c[2]};
#pragma unroll
        for (int k = 1; k < kNSH; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) L[c] = fmaf(s_c[k * 3 + c], B[k], L[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fmaf(f[c] * ip, L[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[idx * 3 + c] = acc[c] * g.inv_spp;
}

// =================================================================================================
// backward (sample directions and pdf are constants: stop-gradient, as in the reference's torch
// variants -- `D.data`, `alpha.data`, mi_plugin.py:179,366)
// =================================================================================================
template <bool WANT_MAT, bool WANT_N, bool WANT_LIGHT>
__global__ __launch_bounds__(kBlock) void shade_bwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                           const float* __restrict__ m, const float* __restrict__ n,
                                                           const float* __restrict__ light, const float* __restrict__ d_out,
                                                           float* __restrict__ d_a, float* __restrict__ d_r,
                                                           float* __restrict__ d_m, float* __restrict__ d_n,
                                                           float* __restrict__ partials, const Geom g, const SampleTable tab) {
    __shared__ float s_c[kNL + 1];
    __shared__ float s_red[4][kNL + 1];
    const int b = blockIdx.y;
    stage_light(s_c, light, b);
    const int P = g.H * g.W;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    const bool active = p < P;
    const long idx = (long)b * P + (active ? p : P - 1);
    Pixel px;
    load_pixel(px, a, r, m, n, idx, (active ? p : P - 1) / g.W, (active ? p : P - 1) % g.W, g);
    float go[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) go[c] = active ? d_out[idx * 3 + c] * g.inv_spp : 0.0f;

    BrdfGrad gr;
#pragma unroll
    for (int c = 0; c < 3; ++c) gr.d_a[c] = 0.0f;
    gr.d_r = gr.d_m = gr.dNoL = gr.dNoV = gr.dNoH = 0.0f;
    float dnx = 0.0f, dny = 0.0f;  // gradient w.r.t. the unit normal, tangential components only (see below)
    float dc[WANT_LIGHT ? kNL : 1];
    if (WANT_LIGHT) {
#pragma unroll
        for (int k = 0; k < kNL; ++k) dc[k] = 0.0f;
    }

    const int spp = 2 * g.half;
    for (int s = 0; s < spp; ++s) {
        Sample sm;
        if (s < g.half) diffuse_sample<WANT_N>(px, tab.diff[s], sm);
        else specular_sample<WANT_N>(px, tab.spec[s - g.half], sm);
        BrdfState st;
        float f[3], pdf;
        brdf_core(px.pc, sm.NoL_raw, sm.NoH, sm.VoH, sm.den, st, f, pdf);
        float ip = pdf > 1e-6f ? rcp(pdf + 1e-6f) : 0.0f;
        float B[kNSH];
        sh_poly(sm.wi, B);
        if (WANT_LIGHT) {
            float gw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gw[c] = go[c] * f[c] * ip;
#pragma unroll
            for (int k = 0; k < kNSH; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dc[k * 3 + c] = fmaf(gw[c], B[k], dc[k * 3 + c]);
            }
        }
        if (WANT_MAT || WANT_N) {
            float L[3] = {s_c[0], s_c[1], s_c[2]};
#pragma unroll
            for (int k = 1; k < kNSH; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) L[c] = fmaf(s_c[k * 3 + c], B[k], L[c]);
            }
            float gg[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gg[c] = go[c] * L[c] * ip;
            BrdfGrad one;
#pragma unroll
            for (int c = 0; c < 3; ++c) one.d_a[c] = 0.0f;
            one.d_r = one.d_m = one.dNoL = one.dNoV = one.dNoH = 0.0f;
            brdf_core_grad<WANT_N>(px.pc, st, gg, one);
#pragma unroll
            for (int c = 0; c < 3; ++c) gr.d_a[c] += one.d_a[c];
            gr.d_r += one.d_r;
            gr.d_m += one.d_m;
            if (WANT_N) {
                // dr.maximum(x, 0) passes the gradient where x > 0 (mi_plugin.py:1393-1396)
                float gl = sm.NoL_raw > 0.0f ? one.dNoL : 0.0f;
                float gh = sm.nh_pos ? one.dNoH : 0.0f;
                gr.dNoV += one.dNoV;
                dnx = fmaf(gl, sm.lwx, fmaf(gh, sm.lhx, dnx));
                dny = fmaf(gl, sm.lwy, fmaf(gh, sm.lhy, dny));
            }
        }
    }

    if (active) {
        if (WANT_MAT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) d_a[idx * 3 + c] = gr.d_a[c];
            d_r[idx] = gr.d_r;
            d_m[idx] = gr.d_m;
        }
        if (WANT_N) {
            // d/dn_hat = sum gl wi + gh h + gv wo.  Through n_hat = n/|n| only its tangential part survives:
            // d_n = (g - n_hat (n_hat.g)) / |n|, so g is accumulated in the shading frame's (s,t) plane directly and the
            // (huge, alternating-sign) radial parts of the GGX-peak terms never enter an fp32 sum.
            float gv = px.pc.NoV_raw > 0.0f ? gr.dNoV : 0.0f;
            dnx = fmaf(gv, px.vx, dnx);
            dny = fmaf(gv, px.vy, dny);
#pragma unroll
            for (int c = 0; c < 3; ++c) d_n[idx * 3 + c] = fmaf(px.s[c], dnx, px.t[c] * dny) * px.inv_len;
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
// plugin face: N independent lanes (literal restatement: raw n, h = normalize(wi+wo), all dots)
// =================================================================================================
struct Lane {
    PixelConst pc;
    float NoL_raw, NoH, VoH, den, nh_raw;
    float h[3];
};
__device__ __forceinline__ void lane_setup(Lane& ln, const float wi[3], const float wo[3], const float n[3], const float a[3], float r,
                                           float m) {
    float h[3] = {wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]};
    float il = rsq(dot3(h, h));
#pragma unroll
    for (int c = 0; c < 3; ++c) ln.h[c] = h[c] * il;
    pixel_const(ln.pc, a, r, m, dot3(n, wo));
    ln.NoL_raw = dot3(n, wi);
    ln.VoH = fmaxf(dot3(wo, ln.h), 0.0f);
    ln.nh_raw = dot3(n, ln.h);
    ln.NoH = fmaxf(ln.nh_raw, 0.0f);
    ln.den = ggx_den(ln.pc, ln.NoH, -1.0f);
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
    BrdfState st;
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
    BrdfState st;
    float fv[3], p;
    brdf_core(ln.pc, ln.NoL_raw, ln.NoH, ln.VoH, ln.den, st, fv, p);
    BrdfGrad o;
#pragma unroll
    for (int c = 0; c < 3; ++c) o.d_a[c] = 0.0f;
    o.d_r = o.d_m = o.dNoL = o.dNoV = o.dNoH = 0.0f;
    brdf_core_grad<true>(ln.pc, st, gv, o);
    float gl = ln.NoL_raw > 0.0f ? o.dNoL : 0.0f, gvv = ln.pc.NoV_raw > 0.0f ? o.dNoV : 0.0f, gh = ln.nh_raw > 0.0f ? o.dNoH : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d_a[3 * k + c] = o.d_a[c];
        d_n[3 * k + c] = fmaf(gl, wiv[c], fmaf(gvv, wov[c], gh * ln.h[c]));
    }
    d_r[k] = o.d_r;
    d_m[k] = o.d_m;
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
        ln.den = ggx_den(ln.pc, cos_h, sin2_h);
    }
    BrdfState st;
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
    stage_light(s_c, coef, 0);
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
        case MATPBR_ERR_WORKSPACE: return "workspace missing or too small for the light gradient";
        default: return "unknown matpbr error";
    }
}

int matpbr_shade_fwd(const float* a, const float* r, const float* m, const float* n, const float* light, int light_kind,
                     int n_light, float* out_rgb, int H, int W, int batch, int spp, const MatpbrCamera* cam, uint32_t flags,
                     void* stream) {
    (void)flags;
    if (!a || !r || !m || !n || !light || !out_rgb || batch <= 0) return MATPBR_ERR_INVALID_ARG;
    if (light_kind != MATPBR_LIGHT_SH25 || n_light != MATPBR_NSH) return MATPBR_ERR_INVALID_ARG;
    if (!valid_spp(spp)) return MATPBR_ERR_UNSUPPORTED;
    Geom g;
    if (!make_geom(H, W, spp, cam, g)) return MATPBR_ERR_INVALID_ARG;
    SampleTable tab;
    fill_sample_table(spp, tab);
    dim3 grid((unsigned)((H * W + kBlock - 1) / kBlock), (unsigned)batch);
    hipLaunchKernelGGL(shade_fwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, a, r, m, n, light, out_rgb, g, tab);
    return launch_status();
}

size_t matpbr_shade_bwd_workspace_bytes(int H, int W, int batch, int n_light) {
    if (H <= 0 || W <= 0 || batch <= 0 || n_light <= 0) return 0;
    size_t nblocks = ((size_t)H * W + kBlock - 1) / kBlock;
    return nblocks * (size_t)batch * (size_t)n_light * 3 * sizeof(float);
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
    dim3 grid((unsigned)((H * W + kBlock - 1) / kBlock), (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
#define MATPBR_LAUNCH_BWD(MAT, NRM, LGT) \
    hipLaunchKernelGGL((shade_bwd_kernel<MAT, NRM, LGT>), grid, dim3(kBlock), 0, st, a, r, m, n, light, d_out_rgb, d_a, d_r, d_m, d_n, part, g, tab)
    const int sel = (want_mat ? 4 : 0) | (want_n ? 2 : 0) | (want_light ? 1 : 0);
    switch (sel) {
        case 1: MATPBR_LAUNCH_BWD(false, false, true); break;
        case 2: MATPBR_LAUNCH_BWD(false, true, false); break;
        case 3: MATPBR_LAUNCH_BWD(false, true, true); break;
        case 4: MATPBR_LAUNCH_BWD(true, false, false); break;
        case 5: MATPBR_LAUNCH_BWD(true, false, true); break;
        case 6: MATPBR_LAUNCH_BWD(true, true, false); break;
        default: MATPBR_LAUNCH_BWD(true, true, true); break;
    }
#undef MATPBR_LAUNCH_BWD
    if (hipGetLastError() != hipSuccess) return MATPBR_ERR_LAUNCH;
    if (want_light) {
        hipLaunchKernelGGL(light_grad_finalize_kernel, dim3(kNL, (unsigned)batch), dim3(kBlock), 0, st, part, d_light, (int)grid.x);
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

}  // extern "C"
