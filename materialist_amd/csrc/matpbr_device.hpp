// matpbr_device.hpp -- device-side arithmetic of the PBR shading path (gfx950, wave64, fp32).
//
// Restates, for the GPU, the reference's BRDF helpers and MatDiffBSDF.eval_brdf / sample_brdf
// (myutils/mi_plugin.py:60-97,217-281,1296-1341,1372-1427) and the order-4 real SH convention of
// myutils/computeSH.py:13-68.
//
// Everything per-sample is templated on the value type T:
//   T = float : one item per lane   -- the N-lane plugin-face kernels (matpbr_eval_brdf, matpbr_sample_brdf);
//   T = f2    : two PIXELS per lane -- the image kernels.  gfx950 issues one wave64 VALU instruction per ~4
//               cycles per SIMD whether it is v_fma_f32 or v_pk_fma_f32 (measured: tools/valu_rate.hip, 69 vs
//               116 TFLOP/s), so carrying two pixels in every 64-bit register pair nearly halves the issue
//               slots of every add/mul/fma of the estimator; only rcp/rsq/sqrt/max/select stay per-component.
//               Pixels (not samples) are paired so that every per-pixel quantity is a genuine pair and every
//               per-sample / per-image quantity is wave-uniform and can sit in SGPRs (free broadcast operand).
// The same `brdf_core*` code serves both, so the function-level golden tests exercise what the render runs.
#pragma once
#include <hip/hip_runtime.h>

namespace matpbr {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr int kNSH = 25;

// ---- scalar / packed helpers --------------------------------------------------------------------
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f2 rcp(f2 x) { return f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ f2 rsq(f2 x) { return f2{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ f2 fsqrt(f2 x) { return f2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, float b) { return f2{fmaxf(a.x, b), fmaxf(a.y, b)}; }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ f2 vmin(f2 a, float b) { return f2{fminf(a.x, b), fminf(a.y, b)}; }
__device__ __forceinline__ float vabs(float a) { return fabsf(a); }
__device__ __forceinline__ f2 vabs(f2 a) { return f2{fabsf(a.x), fabsf(a.y)}; }
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 vfma(f2 a, float b, f2 c) { return __builtin_elementwise_fma(a, f2{b, b}, c); }
__device__ __forceinline__ f2 vfma(float a, f2 b, f2 c) { return __builtin_elementwise_fma(f2{a, a}, b, c); }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, float c) { return __builtin_elementwise_fma(a, b, f2{c, c}); }
__device__ __forceinline__ f2 vfma(f2 a, float b, float c) { return __builtin_elementwise_fma(a, f2{b, b}, f2{c, c}); }
__device__ __forceinline__ f2 vfma(float a, f2 b, float c) { return __builtin_elementwise_fma(f2{a, a}, b, f2{c, c}); }
// where(x > 0, a, b) / where(x > thr, a, 0)
__device__ __forceinline__ float sel_pos(float x, float a, float b) { return x > 0.0f ? a : b; }
__device__ __forceinline__ f2 sel_pos(f2 x, f2 a, f2 b) { return f2{x.x > 0.0f ? a.x : b.x, x.y > 0.0f ? a.y : b.y}; }
__device__ __forceinline__ f2 sel_pos(f2 x, f2 a, float b) { return f2{x.x > 0.0f ? a.x : b, x.y > 0.0f ? a.y : b}; }
__device__ __forceinline__ f2 sel_pos(f2 x, float a, float b) { return f2{x.x > 0.0f ? a : b, x.y > 0.0f ? a : b}; }
__device__ __forceinline__ float hsum(float x) { return x; }
__device__ __forceinline__ float hsum(f2 x) { return x.x + x.y; }
template <class T> __device__ __forceinline__ T pow4(T x) { T x2 = x * x; return x2 * x2; }
template <class T> __device__ __forceinline__ T pow5(T x) { T x2 = x * x; return x2 * x2 * x; }
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
template <class T>
struct PixelConst {
    T a[3], kd[3], C0[3], omC0[3];  // albedo, a(1-m)/pi, C_0, 1-C_0        (:1405,1412)
    T r, m;
    T alpha2, am1, a2_over_pi;      // r^4, r^4-1, r^4/pi                    (:93-97)
    T omk, kpe, dk_dr;              // 1-k, k+1e-6, dk/dr; k=(r+1)^2/8       (:64-67)
    T NoV_raw, NoV, g1v, po;        // n.wo, max(.,0), G1(NoV), (1-NoV)^5    (:1394,1407,1411)
};

template <class T>
__device__ __forceinline__ void pixel_const(PixelConst<T>& pc, const T a[3], T r, T m, T NoV_raw) {
    pc.r = r; pc.m = m;
    T omm = 1.0f - m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pc.a[c] = a[c];
        pc.kd[c] = (a[c] * omm) * kInvPi;
        pc.C0[c] = vfma(m, a[c], omm * 0.04f);
        pc.omC0[c] = 1.0f - pc.C0[c];
    }
    pc.alpha2 = pow4(r);
    pc.am1 = pc.alpha2 - 1.0f;
    pc.a2_over_pi = pc.alpha2 * kInvPi;
    T rp1 = r + 1.0f;
    T k = (rp1 * rp1) * 0.125f;
    pc.omk = 1.0f - k;
    pc.kpe = k + 1e-6f;
    pc.dk_dr = rp1 * 0.25f;
    pc.NoV_raw = NoV_raw;
    pc.NoV = vmax(NoV_raw, 0.0f);
    pc.g1v = rcp(vfma(pc.NoV, pc.omk, pc.kpe));
    pc.po = pow5(1.0f - pc.NoV);
}

// den = NoH^2 (alpha2-1) + 1 + 1e-6 of D_GGX (:95).  The literal form loses its digits in fp32 on the GGX
// peak (NoH -> 1, alpha2 ~ 2e-5); where the caller knows 1-NoH^2 without cancellation it uses the second form,
// which is the same real number: alpha2 NoH^2 + (1 - NoH^2).
template <class T>
__device__ __forceinline__ T ggx_den_literal(const PixelConst<T>& pc, T NoH) { return vfma(NoH * NoH, pc.am1, 1.0f) + 1e-6f; }
template <class T>
__device__ __forceinline__ T ggx_den_stable(const PixelConst<T>& pc, T one_m_NoH2) { return vfma(one_m_NoH2, -pc.am1, pc.alpha2) + 1e-6f; }

// Value of eval_brdf (f*cos, RGB) and the mixture pdf for already-clamped cosines (:1392-1415).
template <class T>
struct BrdfState {  // intermediates the backward pass reuses
    T NoL, NoH, VoH, iden, D, FDm1, pi5, Fi, Fo, g1l, G, x5, dsc, ssc;
};
template <class T>
__device__ __forceinline__ void brdf_core(const PixelConst<T>& pc, T NoL_raw, T NoH, T VoH, T den, BrdfState<T>& s, T f[3], T& pdf) {
    s.NoL = vmax(NoL_raw, 0.0f);
    s.NoH = NoH;
    s.VoH = VoH;
    s.iden = rcp(den);
    s.D = pc.a2_over_pi * (s.iden * s.iden);                                            // :93-97
    pdf = vfma(0.125f * (s.D * NoH), rcp(vmax(VoH, 1e-6f)), (0.5f * kInvPi) * s.NoL);   // :1399-1401
    s.FDm1 = vfma((2.0f * pc.r) * VoH, VoH, -0.5f);                                     // F_D90 - 1, :1406
    s.pi5 = pow5(1.0f - s.NoL);
    s.Fi = vfma(s.FDm1, s.pi5, 1.0f);                                                   // :1408
    s.Fo = vfma(s.FDm1, pc.po, 1.0f);                                                   // :1407
    s.g1l = rcp(vfma(s.NoL, pc.omk, pc.kpe));
    s.G = s.g1l * pc.g1v;                                                               // :1411
    s.x5 = pow5(1.0f - VoH);
    s.dsc = s.Fo * s.Fi * s.NoL;                                                        // :1409 without baseColor_d/pi
    s.ssc = 0.25f * (s.D * s.G) * s.NoL;                                                // :1414 without F_m
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = vfma(s.ssc, vfma(s.x5, pc.omC0[c], pc.C0[c]), pc.kd[c] * s.dsc);  // :1413,1415
}

// Gradient of f (RGB, upstream weights g) w.r.t. a, r, m and the three cosines; accumulates into T-typed sums
// (packed: one partial sum per sample slot, added horizontally once per pixel).
template <class T>
struct BrdfGrad { T d_a[3], d_r, d_m, dNoV; };
template <class T>
__device__ __forceinline__ void brdf_grad_zero(BrdfGrad<T>& o) {
    o.d_a[0] = o.d_a[1] = o.d_a[2] = T(0.0f);
    o.d_r = o.d_m = o.dNoV = T(0.0f);
}
// gl / gh gate the NoL / NoH cosine gradients (dr.maximum passes the gradient where its argument is > 0)
template <class T, bool WANT_N>
__device__ __forceinline__ void brdf_core_grad(const PixelConst<T>& pc, const BrdfState<T>& s, const T g[3], BrdfGrad<T>& o, T& gl, T& gh) {
    T omx5 = 1.0f - s.x5;
    T gd = g[0] * pc.kd[0], gs = g[0] * vfma(s.x5, pc.omC0[0], pc.C0[0]);
    T da_d = ((1.0f - pc.m) * kInvPi) * s.dsc;   // d f_d / d a
    T dm_s = s.ssc * omx5;
    T da = vfma(dm_s, pc.m, da_d);               // + d f_s / d a (via C_0)
    T gm = g[0] * vfma(dm_s, pc.a[0] - 0.04f, (-pc.a[0] * kInvPi) * s.dsc);
    o.d_a[0] = vfma(g[0], da, o.d_a[0]);
#pragma unroll
    for (int c = 1; c < 3; ++c) {
        gd = vfma(g[c], pc.kd[c], gd);
        gs = vfma(g[c], vfma(s.x5, pc.omC0[c], pc.C0[c]), gs);
        o.d_a[c] = vfma(g[c], da, o.d_a[c]);
        gm = vfma(g[c], vfma(dm_s, pc.a[c] - 0.04f, (-pc.a[c] * kInvPi) * s.dsc), gm);
    }
    o.d_m += gm;
    // d/dr: F_D90 = .5 + 2 VoH^2 r ; D(alpha2 = r^4) ; G(k = (r+1)^2/8)
    T two_voh2 = 2.0f * (s.VoH * s.VoH);
    T dFoFi = vfma(s.Fi, pc.po, s.Fo * s.pi5) * two_voh2;
    T r3 = pc.r * pc.r * pc.r;
    T dD_dr = s.D * vfma((-8.0f * r3) * (s.NoH * s.NoH), s.iden, 4.0f * rcp(pc.r));
    T dG_dr = (-pc.dk_dr) * s.G * vfma(s.g1l, 1.0f - s.NoL, pc.g1v * (1.0f - pc.NoV));
    T gsq = (gs * 0.25f) * s.NoL;
    o.d_r += vfma(gd * s.NoL, dFoFi, gsq * vfma(dD_dr, s.G, s.D * dG_dr));
    if (WANT_N) {
        T dFi = (-5.0f * s.FDm1) * pow4(1.0f - s.NoL);
        T dFo = (-5.0f * pow4(1.0f - pc.NoV)) * s.FDm1;
        T dG_dNoL = (-pc.omk) * s.g1l * s.G;
        T dG_dNoV = (-pc.g1v * pc.omk) * s.G;
        gl = vfma(gd * s.Fo, vfma(dFi, s.NoL, s.Fi), (gs * 0.25f) * s.D * vfma(dG_dNoL, s.NoL, s.G));
        o.dNoV += vfma(gd * dFo, s.Fi * s.NoL, gsq * s.D * dG_dNoV);
        gh = gsq * s.G * ((-4.0f * pc.am1) * s.D * s.NoH * s.iden);
    }
}

// ---- [ext] mi.Frame3f: Duff et al. 2017 branchless orthonormal basis (Mitsuba 3 coordinate_system) -------
__device__ __forceinline__ float sign_ge0(float x) { return x >= 0.0f ? 1.0f : -1.0f; }
__device__ __forceinline__ f2 sign_ge0(f2 x) { return f2{x.x >= 0.0f ? 1.0f : -1.0f, x.y >= 0.0f ? 1.0f : -1.0f}; }
template <class T>
__device__ __forceinline__ void frame(const T n[3], T s[3], T t[3]) {
    T sign = sign_ge0(n[2]);
    T a = -rcp(sign + n[2]);
    T b = n[0] * n[1] * a;
    s[0] = vfma(sign * n[0] * n[0], a, 1.0f); s[1] = sign * b; s[2] = -sign * n[0];
    t[0] = b; t[1] = vfma(n[1] * n[1], a, sign); t[2] = -n[1];
}
// local (x,y,z) -> world with the frame (s,t,n); x,y,z may be wave-uniform scalars (sample table) or T
template <class T, class U>
__device__ __forceinline__ void to_world(const T s[3], const T t[3], const T n[3], U x, U y, U z, T out[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = vfma(n[i], z, vfma(t[i], y, s[i] * x));
}
template <class T>
__device__ __forceinline__ T dot3v(const T a[3], const T b[3]) { return vfma(a[2], b[2], vfma(a[1], b[1], a[0] * b[0])); }

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

template <class T>
__device__ __forceinline__ void sh_poly(const T w[3], T B[kNSH]) {
    const T X = -w[2], Y = w[0], Z = w[1];
    const T z2 = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z, y2 = Y * Y;
    const T d = vfma(X, X, -y2);
    const T t5 = vfma(z2, 5.0f, -1.0f), t7 = vfma(z2, 7.0f, -1.0f), t73 = t7 - 2.0f;
    const T s3 = Y * vfma(3.0f * X, X, -y2), c3 = X * vfma(X, X, -3.0f * y2);
    B[0] = T(1.0f);
    B[1] = Y; B[2] = Z; B[3] = X;
    B[4] = xy; B[5] = yz; B[6] = vfma(z2, 3.0f, -1.0f); B[7] = xz; B[8] = d;
    B[9] = s3; B[10] = xy * Z; B[11] = Y * t5; B[12] = Z * (t5 - 2.0f); B[13] = X * t5; B[14] = d * Z; B[15] = c3;
    B[16] = xy * d; B[17] = s3 * Z; B[18] = xy * t7; B[19] = yz * t73; B[20] = vfma(vfma(z2, 35.0f, -30.0f), z2, 3.0f);
    B[21] = xz * t73; B[22] = d * t7; B[23] = c3 * Z; B[24] = vfma(d, d, -4.0f * (xy * xy));
}

}  // namespace matpbr
