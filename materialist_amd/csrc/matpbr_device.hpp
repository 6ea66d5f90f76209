// matpbr_device.hpp -- device-side arithmetic of the PBR shading path (gfx950, wave64, fp32).
//
// Restates, for the GPU, the reference's BRDF helpers and MatDiffBSDF.eval_brdf / sample_brdf
// (myutils/mi_plugin.py:60-97,217-281,1296-1341,1372-1427) and the order-4 real SH convention of
// myutils/computeSH.py:13-68.  The same `brdf_core_*` functions serve the image kernels
// (matpbr_shade_fwd/bwd) and the N-lane plugin-face kernels (matpbr_eval_brdf, matpbr_sample_brdf),
// so the function-level golden tests exercise the code the render runs.
#pragma once
#include <hip/hip_runtime.h>

namespace matpbr {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr int kNSH = 25;

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float pow4(float x) { float x2 = x * x; return x2 * x2; }
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

// ---- a1-a3 as stand-alone functions (myutils/mi_plugin.py:60-97) -------------------------------
__device__ __forceinline__ float G1_GGX_Schlick(float NoV, float eta) {
    float k = eta + 1.0f;
    k = k * k * 0.125f;
    return rcp(fmaf(NoV, 1.0f - k, k + 1e-6f));
}
__device__ __forceinline__ float G_Smith(float NoV, float NoL, float eta) { return G1_GGX_Schlick(NoL, eta) * G1_GGX_Schlick(NoV, eta); }
__device__ __forceinline__ float fresnelSchlick(float VoH, float F0) { return fmaf(1.0f - F0, pow5(1.0f - VoH), F0); }
__device__ __forceinline__ float D_GGX(float cos_h, float eta) {
    float alpha2 = pow4(eta);
    float denom = fmaf(cos_h * cos_h, alpha2 - 1.0f, 1.0f) + 1e-6f;
    return alpha2 * kInvPi * rcp(denom * denom);
}

// ---- per-pixel constants of eval_brdf (everything that does not depend on the light direction) --
struct PixelConst {
    float a[3], kd[3], C0[3], omC0[3];  // albedo, a(1-m)/pi, C_0, 1-C_0        (:1405,1412)
    float r, m;
    float alpha2, am1, a2_over_pi;      // r^4, r^4-1, r^4/pi                    (:93-97)
    float omk, kpe, dk_dr;              // 1-k, k+1e-6, dk/dr; k=(r+1)^2/8       (:64-67)
    float NoV_raw, NoV, g1v, po;        // n.wo, max(.,0), G1(NoV), (1-NoV)^5    (:1394,1407,1411)
};

__device__ __forceinline__ void pixel_const(PixelConst& pc, const float a[3], float r, float m, float NoV_raw) {
    pc.r = r; pc.m = m;
    float omm = 1.0f - m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pc.a[c] = a[c];
        pc.kd[c] = a[c] * omm * kInvPi;
        pc.C0[c] = fmaf(m, a[c], omm * 0.04f);
        pc.omC0[c] = 1.0f - pc.C0[c];
    }
    pc.alpha2 = pow4(r);
    pc.am1 = pc.alpha2 - 1.0f;
    pc.a2_over_pi = pc.alpha2 * kInvPi;
    float k = (r + 1.0f) * (r + 1.0f) * 0.125f;
    pc.omk = 1.0f - k;
    pc.kpe = k + 1e-6f;
    pc.dk_dr = (r + 1.0f) * 0.25f;
    pc.NoV_raw = NoV_raw;
    pc.NoV = fmaxf(NoV_raw, 0.0f);
    pc.g1v = rcp(fmaf(pc.NoV, pc.omk, pc.kpe));
    pc.po = pow5(1.0f - pc.NoV);
}

// den = NoH^2 (alpha2-1) + 1 + 1e-6 of D_GGX (:95).  `one_m_NoH2` < 0 selects the literal form; otherwise the
// caller supplies 1-NoH^2 computed without cancellation (specular samples know sin^2(theta_h) exactly).
__device__ __forceinline__ float ggx_den(const PixelConst& pc, float NoH, float one_m_NoH2) {
    if (one_m_NoH2 >= 0.0f) return fmaf(pc.alpha2, 1.0f - one_m_NoH2, one_m_NoH2) + 1e-6f;
    return fmaf(NoH * NoH, pc.am1, 1.0f) + 1e-6f;
}

// Value of eval_brdf (f*cos, RGB) and the mixture pdf for already-clamped cosines (:1392-1415).
struct BrdfState {  // intermediates the backward pass reuses
    float NoL, NoH, VoH, iden, D, FDm1, pi5, Fi, Fo, g1l, G, x5, dsc, ssc;
};
__device__ __forceinline__ void brdf_core(const PixelConst& pc, float NoL_raw, float NoH, float VoH, float den,
                                          BrdfState& s, float f[3], float& pdf) {
    s.NoL = fmaxf(NoL_raw, 0.0f);
    s.NoH = NoH;
    s.VoH = VoH;
    s.iden = rcp(den);
    s.D = pc.a2_over_pi * s.iden * s.iden;                                  // :93-97
    pdf = fmaf(0.125f * s.D * NoH, rcp(fmaxf(VoH, 1e-6f)), (0.5f * kInvPi) * s.NoL);  // :1399-1401
    s.FDm1 = fmaf(2.0f * VoH * VoH, pc.r, -0.5f);                           // F_D90 - 1, :1406
    s.pi5 = pow5(1.0f - s.NoL);
    s.Fi = fmaf(s.FDm1, s.pi5, 1.0f);                                       // :1408
    s.Fo = fmaf(s.FDm1, pc.po, 1.0f);                                       // :1407
    s.g1l = rcp(fmaf(s.NoL, pc.omk, pc.kpe));
    s.G = s.g1l * pc.g1v;                                                   // :1411
    s.x5 = pow5(1.0f - VoH);
    s.dsc = s.Fo * s.Fi * s.NoL;                                            // :1409 without baseColor_d/pi
    s.ssc = 0.25f * s.D * s.G * s.NoL;                                      // :1414 without F_m
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = fmaf(s.ssc, fmaf(pc.omC0[c], s.x5, pc.C0[c]), pc.kd[c] * s.dsc);  // :1413,1415
}

// Gradient of f (RGB, upstream weights g) w.r.t. a, r, m and the three cosines; accumulates.
struct BrdfGrad { float d_a[3], d_r, d_m, dNoL, dNoV, dNoH; };
template <bool WANT_N>
__device__ __forceinline__ void brdf_core_grad(const PixelConst& pc, const BrdfState& s, const float g[3], BrdfGrad& o) {
    float omx5 = 1.0f - s.x5;
    float gd = 0.0f, gs = 0.0f, gm = 0.0f;
    float da_d = (1.0f - pc.m) * kInvPi * s.dsc;   // d f_d / d a
    float da_s = s.ssc * omx5 * pc.m;              // d f_s / d a (via C_0)
    float dm_s = s.ssc * omx5;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gd = fmaf(g[c], pc.kd[c], gd);
        gs = fmaf(g[c], fmaf(pc.omC0[c], s.x5, pc.C0[c]), gs);
        o.d_a[c] = fmaf(g[c], da_d + da_s, o.d_a[c]);
        gm = fmaf(g[c], fmaf(dm_s, pc.a[c] - 0.04f, -pc.a[c] * kInvPi * s.dsc), gm);
    }
    o.d_m += gm;
    // d/dr: F_D90 = .5 + 2 VoH^2 r ; D(alpha2 = r^4) ; G(k = (r+1)^2/8)
    float two_voh2 = 2.0f * s.VoH * s.VoH;
    float dFoFi = fmaf(pc.po, s.Fi, s.Fo * s.pi5) * two_voh2;
    float r3 = pc.r * pc.r * pc.r;
    float dD_dr = s.D * fmaf(-8.0f * r3 * s.NoH * s.NoH, s.iden, 4.0f * rcp(pc.r));
    float dG_dr = -pc.dk_dr * s.G * fmaf(s.g1l, 1.0f - s.NoL, pc.g1v * (1.0f - pc.NoV));
    float gsq = gs * 0.25f * s.NoL;
    o.d_r += fmaf(gd * s.NoL, dFoFi, gsq * fmaf(dD_dr, s.G, s.D * dG_dr));
    if (WANT_N) {
        float dFi = -s.FDm1 * 5.0f * pow4(1.0f - s.NoL);
        float dFo = -s.FDm1 * 5.0f * pow4(1.0f - pc.NoV);
        float dG_dNoL = -s.g1l * pc.omk * s.G;
        float dG_dNoV = -pc.g1v * pc.omk * s.G;
        o.dNoL += fmaf(gd * s.Fo, fmaf(dFi, s.NoL, s.Fi), gs * 0.25f * s.D * fmaf(dG_dNoL, s.NoL, s.G));
        o.dNoV += fmaf(gd * dFo, s.Fi * s.NoL, gsq * s.D * dG_dNoV);
        o.dNoH += gsq * s.G * (-4.0f * s.D * s.NoH * pc.am1 * s.iden);
    }
}

// ---- [ext] mi.Frame3f: Duff et al. 2017 branchless orthonormal basis (Mitsuba 3 coordinate_system) -------
__device__ __forceinline__ void frame(const float n[3], float s[3], float t[3]) {
    float sign = n[2] >= 0.0f ? 1.0f : -1.0f;
    float a = -rcp(sign + n[2]);
    float b = n[0] * n[1] * a;
    s[0] = fmaf(sign * n[0] * n[0], a, 1.0f); s[1] = sign * b; s[2] = -sign * n[0];
    t[0] = b; t[1] = fmaf(n[1] * n[1], a, sign); t[2] = -n[1];
}
__device__ __forceinline__ void to_world(const float s[3], const float t[3], const float n[3], float x, float y, float z, float out[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = fmaf(n[i], z, fmaf(t[i], y, s[i] * x));
}

// ---- order-4 real SH: Y_k = kShNorm[k] * B_k(X,Y,Z), (X,Y,Z) = (-z, x, y) of the world direction --------
// (theta = acos(y), phi = atan2(x,-z): myutils/envmap_utils.py:29-36; basis: myutils/computeSH.py:13-68)
__device__ __constant__ const float kShNorm[kNSH] = {
    0.28209479177387814f,
    -0.4886025119029199f, 0.4886025119029199f, -0.4886025119029199f,
    1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f,
    -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f,
    1.445305721320277f, -0.5900435899266435f,
    2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f, 0.10578554691520431f,
    -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

__device__ __forceinline__ void sh_poly(const float w[3], float B[kNSH]) {
    const float X = -w[2], Y = w[0], Z = w[1];
    const float z2 = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
    const float d = fmaf(X, X, -Y * Y);
    const float t5 = fmaf(5.0f, z2, -1.0f), t7 = fmaf(7.0f, z2, -1.0f), t73 = t7 - 2.0f;
    const float s3 = Y * fmaf(3.0f * X, X, -Y * Y), c3 = X * fmaf(X, X, -3.0f * Y * Y);
    B[0] = 1.0f;
    B[1] = Y; B[2] = Z; B[3] = X;
    B[4] = xy; B[5] = yz; B[6] = fmaf(3.0f, z2, -1.0f); B[7] = xz; B[8] = d;
    B[9] = s3; B[10] = xy * Z; B[11] = Y * t5; B[12] = Z * (t5 - 2.0f); B[13] = X * t5; B[14] = d * Z; B[15] = c3;
    B[16] = xy * d; B[17] = s3 * Z; B[18] = xy * t7; B[19] = yz * t73; B[20] = fmaf(fmaf(35.0f, z2, -30.0f), z2, 3.0f);
    B[21] = xz * t73; B[22] = d * t7; B[23] = c3 * Z; B[24] = fmaf(d, d, -4.0f * xy * xy);
}

// ---- view direction of pixel (i,j): wo = -p/|p|, p = ((j-cx)/f, -(i-cy)/f, -1)    ---------
__device__ __forceinline__ void view_dir(int i, int j, float inv_f, float cx, float cy, float wo[3]) {
    float x = (cx - (float)j) * inv_f, y = ((float)i - cy) * inv_f;
    float il = rsq(fmaf(x, x, fmaf(y, y, 1.0f)));
    wo[0] = x * il; wo[1] = y * il; wo[2] = il;
}

}  // namespace matpbr
