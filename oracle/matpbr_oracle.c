/*
 * matpbr_oracle.c -- CPU restatement of the reference's PBR shading path.          TEST INFRASTRUCTURE
 *
 * This file is the ORACLE: a plain-C restatement of the arithmetic of lez-s/Materialist's hot path
 * (citations are file:line into the reference tree).  It exists to check the HIP kernels.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
 * (materialist_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Parity pin: every function below that restates a reference function is checked against golden
 * vectors produced by importing the reference's own Python arithmetic under stub modules
 * (tests/golden/gen_golden.py writes the .npz fixtures; tests/test_oracle_golden.py).  The image-level
 * render (oracle_shade_fwd/bwd) has no reference counterpart that can run here or anywhere
 * deterministic (Mitsuba's stochastic path tracer): "parity unpinned" at the
 * mi.render boundary; it is the build's own deterministic definition composed
 * only of pinned functions, and its backward pass is pinned by finite differences.
 *
 * Precision: compiled twice from this one source, -DORACLE_REAL=double (the checker) and
 * -DORACLE_REAL=float with OpenMP (the timed "cpu_baseline" port).
 */
#include <tgmath.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif
typedef ORACLE_REAL real;

#define R(x) ((real)(x))
#define O_PI R(3.14159265358979323846264338327950288)

#define MATPBR_NSH 25
#define MATPBR_MAX_SPP 256

static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline real pow5(real x) { real x2 = x * x; return x2 * x2 * x; }

/* ------------------------------------------------------------------------------------------------
 * a2: G1_GGX_Schlick / G_Smith            myutils/mi_plugin.py:60-76
 * ---------------------------------------------------------------------------------------------- */
real oracle_G1_GGX_Schlick(real NoV, real eta) {
    real k = eta + R(1);
    k = k * k / R(8);
    real denom = NoV * (R(1) - k) + k + R(1e-6);
    return R(1) / denom;
}
real oracle_G_Smith(real NoV, real NoL, real eta) {
    return oracle_G1_GGX_Schlick(NoL, eta) * oracle_G1_GGX_Schlick(NoV, eta);
}
/* a3: fresnelSchlick                       myutils/mi_plugin.py:78-81 */
real oracle_fresnelSchlick(real VoH, real F0) {
    real x = pow5(R(1) - VoH);
    return F0 + (R(1) - F0) * x;
}
/* a1: D_GGX                                myutils/mi_plugin.py:89-97 */
real oracle_D_GGX(real cos_h, real eta) {
    real alpha = eta * eta;
    real alpha2 = alpha * alpha;
    real denom = (cos_h * cos_h * (alpha2 - R(1)) + R(1)) + R(1e-6);
    denom = O_PI * denom * denom;
    return alpha2 / denom;
}

/* ------------------------------------------------------------------------------------------------
 * a4: MatDiffBSDF.eval_brdf (disney_brdf=True branch)     myutils/mi_plugin.py:1372-1427
 *     wi = light direction, wo = view direction (F9), value already includes the cosine NoL.
 * ---------------------------------------------------------------------------------------------- */
void oracle_eval_brdf(const real wi[3], const real wo[3], const real n[3], const real a[3], real r, real m,
                      real f[3], real* pdf) {
    real h[3] = {wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]};
    real hl = sqrt(dot3(h, h));
    h[0] /= hl; h[1] /= hl; h[2] /= hl;                                   /* :1392 */
    real NoL = rmax(dot3(n, wi), R(0));                                   /* :1393 */
    real NoV = rmax(dot3(n, wo), R(0));
    real VoH = rmax(dot3(wo, h), R(0));
    real NoH = rmax(dot3(n, h), R(0));
    real D = oracle_D_GGX(NoH, r);                                        /* :1398 */
    real pdf_spec = D / (R(4) * rmax(VoH, R(1e-6))) * NoH;                /* :1399 */
    real pdf_diff = NoL / O_PI;                                           /* :1400 */
    *pdf = R(0.5) * pdf_spec + R(0.5) * pdf_diff;                         /* :1401 */
    real F_D90 = R(0.5) + R(2) * VoH * VoH * r;                           /* :1406 */
    real F_D_w_out = R(1) + (F_D90 - R(1)) * pow5(R(1) - NoV);            /* :1407 */
    real F_D_w_in = R(1) + (F_D90 - R(1)) * pow5(R(1) - NoL);             /* :1408 */
    real G = oracle_G_Smith(NoV, NoL, r);                                 /* :1411 */
    real x5 = pow5(R(1) - VoH);
    for (int c = 0; c < 3; ++c) {
        real baseColor_d = a[c] * (R(1) - m);                             /* :1405 */
        real brdf_diff = baseColor_d / O_PI * F_D_w_out * F_D_w_in * NoL; /* :1409 */
        real C_0 = (R(1) - m) * R(0.04) + m * a[c];                       /* :1412 */
        real F_m = C_0 + (R(1) - C_0) * x5;                               /* :1413 */
        real brdf_metal = D * G * F_m / R(4) * NoL;                       /* :1414 */
        f[c] = brdf_diff + brdf_metal;                                    /* :1415 */
    }
}

/* Analytic gradient of eval_brdf's value (not of pdf) w.r.t. a, r, m, n for an upstream RGB weight g:
 *   d_a[c] = g[c] * df[c]/da[c];  d_r = sum_c g[c] df[c]/dr;  d_m likewise;  d_n[3] = sum_c g[c] df[c]/dn.
 * Pinned against torch autograd of the reference function (tests/golden/eval_brdf.npz). */
void oracle_eval_brdf_grad(const real wi[3], const real wo[3], const real n[3], const real a[3], real r, real m,
                           const real g[3], real d_a[3], real* d_r, real* d_m, real d_n[3]) {
    real h[3] = {wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]};
    real hl = sqrt(dot3(h, h));
    h[0] /= hl; h[1] /= hl; h[2] /= hl;
    real nl = dot3(n, wi), nv = dot3(n, wo), nh = dot3(n, h);
    real NoL = rmax(nl, R(0)), NoV = rmax(nv, R(0)), NoH = rmax(nh, R(0));
    real VoH = rmax(dot3(wo, h), R(0));
    real alpha2 = r * r * r * r;
    real den = NoH * NoH * (alpha2 - R(1)) + R(1) + R(1e-6);
    real D = alpha2 / (O_PI * den * den);
    real dD_dalpha2 = R(1) / (O_PI * den * den) - R(2) * alpha2 * NoH * NoH / (O_PI * den * den * den);
    real dD_dr = dD_dalpha2 * R(4) * r * r * r;
    real dD_dNoH = -R(2) * alpha2 / (O_PI * den * den * den) * R(2) * NoH * (alpha2 - R(1));
    real k = (r + R(1)) * (r + R(1)) / R(8);
    real dk_dr = (r + R(1)) / R(4);
    real g1l = R(1) / (NoL * (R(1) - k) + k + R(1e-6));
    real g1v = R(1) / (NoV * (R(1) - k) + k + R(1e-6));
    real G = g1l * g1v;
    real dg1l_dk = -g1l * g1l * (R(1) - NoL), dg1v_dk = -g1v * g1v * (R(1) - NoV);
    real dG_dr = (dg1l_dk * g1v + g1l * dg1v_dk) * dk_dr;
    real dG_dNoL = -g1l * g1l * (R(1) - k) * g1v;
    real dG_dNoV = -g1v * g1v * (R(1) - k) * g1l;
    real FD90 = R(0.5) + R(2) * VoH * VoH * r;
    real po = pow5(R(1) - NoV), pi_ = pow5(R(1) - NoL);
    real Fo = R(1) + (FD90 - R(1)) * po, Fi = R(1) + (FD90 - R(1)) * pi_;
    real dFoFi_dr = (po * Fi + Fo * pi_) * R(2) * VoH * VoH;
    real q4o = (R(1) - NoV); q4o = q4o * q4o; q4o = q4o * q4o;
    real q4i = (R(1) - NoL); q4i = q4i * q4i; q4i = q4i * q4i;
    real dFo_dNoV = -(FD90 - R(1)) * R(5) * q4o;
    real dFi_dNoL = -(FD90 - R(1)) * R(5) * q4i;
    real x5 = pow5(R(1) - VoH);
    real dr_acc = R(0), dm_acc = R(0), dNoL = R(0), dNoV = R(0), dNoH = R(0);
    for (int c = 0; c < 3; ++c) {
        real kd = a[c] * (R(1) - m) / O_PI;
        real C0 = (R(1) - m) * R(0.04) + m * a[c];
        real Fm = C0 + (R(1) - C0) * x5;
        /* f = kd*Fo*Fi*NoL + D*G*Fm/4*NoL */
        real df_da = (R(1) - m) / O_PI * Fo * Fi * NoL + D * G / R(4) * NoL * (R(1) - x5) * m;
        real df_dm = -a[c] / O_PI * Fo * Fi * NoL + D * G / R(4) * NoL * (R(1) - x5) * (a[c] - R(0.04));
        real df_dr = kd * dFoFi_dr * NoL + (dD_dr * G + D * dG_dr) * Fm / R(4) * NoL;
        real df_dNoL = kd * Fo * (dFi_dNoL * NoL + Fi) + D * Fm / R(4) * (dG_dNoL * NoL + G);
        real df_dNoV = kd * dFo_dNoV * Fi * NoL + D * dG_dNoV * Fm / R(4) * NoL;
        real df_dNoH = dD_dNoH * G * Fm / R(4) * NoL;
        d_a[c] = g[c] * df_da;
        dr_acc += g[c] * df_dr;
        dm_acc += g[c] * df_dm;
        dNoL += g[c] * df_dNoL;
        dNoV += g[c] * df_dNoV;
        dNoH += g[c] * df_dNoH;
    }
    *d_r = dr_acc;
    *d_m = dm_acc;
    /* dr.maximum(x, 0): gradient passes where x > 0 */
    if (!(nl > R(0))) dNoL = R(0);
    if (!(nv > R(0))) dNoV = R(0);
    if (!(nh > R(0))) dNoH = R(0);
    for (int i = 0; i < 3; ++i) d_n[i] = dNoL * wi[i] + dNoV * wo[i] + dNoH * h[i];
}

/* ------------------------------------------------------------------------------------------------
 * [ext] mi.Frame3f(n): Mitsuba 3 coordinate_system() = Duff et al. 2017 branchless orthonormal basis.
 * Not in the reference tree (mitsuba==3.5.2 wheel); restated from the published algorithm.
 * ---------------------------------------------------------------------------------------------- */
void oracle_frame(const real n[3], real s[3], real t[3]) {
    real sign = n[2] >= R(0) ? R(1) : -R(1);
    real a = -R(1) / (sign + n[2]);
    real b = n[0] * n[1] * a;
    s[0] = R(1) + sign * n[0] * n[0] * a; s[1] = sign * b; s[2] = -sign * n[0];
    t[0] = b; t[1] = sign + n[1] * n[1] * a; t[2] = -n[1];
}
static inline void to_world(const real s[3], const real t[3], const real n[3], const real v[3], real out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = s[i] * v[0] + t[i] * v[1] + n[i] * v[2];
}

/* a5: mi_diffuse_sampler                   myutils/mi_plugin.py:255-281 */
void oracle_diffuse_sampler(real u0, real u1, const real n[3], real wi[3]) {
    real theta = asin(sqrt(rmax(u0, R(0))));
    real phi = R(2) * O_PI * u1;
    real l[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
    real s[3], t[3];
    oracle_frame(n, s, t);
    to_world(s, t, n, l, wi);
}
/* a5: mi_specular_sampler                  myutils/mi_plugin.py:217-253 */
void oracle_specular_sampler(real u0, real u1, real roughness, const real wo[3], const real n[3], real wi[3]) {
    real alpha = roughness * roughness;
    real cos_theta = sqrt(rmax((R(1) - u0) / (u0 * (alpha * alpha - R(1)) + R(1)), R(0)));
    real sin_theta = sqrt(rmax(R(0), R(1) - cos_theta * cos_theta));
    real phi = R(2) * O_PI * u1;
    real l[3] = {sin_theta * cos(phi), sin_theta * sin(phi), cos_theta};
    real s[3], t[3], wh[3];
    oracle_frame(n, s, t);
    to_world(s, t, n, l, wh);
    real d = R(2) * dot3(wo, wh);
    for (int i = 0; i < 3; ++i) wi[i] = d * wh[i] - wo[i];
    real len = sqrt(dot3(wi, wi));
    for (int i = 0; i < 3; ++i) wi[i] /= len;
}
/* a5: MatDiffBSDF.sample_brdf              myutils/mi_plugin.py:1296-1341
 *     sample1 > 0.5 -> diffuse lobe, else specular; weight = f*cos/(pdf+1e-6) where pdf > 1e-6. */
void oracle_sample_brdf(real sample1, real u0, real u1, const real wo[3], const real n[3], const real a[3], real r,
                        real m, real wi[3], real* pdf_out, real weight[3]) {
    if (sample1 > R(0.5)) oracle_diffuse_sampler(u0, u1, n, wi);
    else oracle_specular_sampler(u0, u1, r, wo, n, wi);
    real f[3], pdf;
    oracle_eval_brdf(wi, wo, n, a, r, m, f, &pdf);
    for (int c = 0; c < 3; ++c) weight[c] = pdf > R(1e-6) ? f[c] / (pdf + R(1e-6)) : R(0);
    *pdf_out = pdf > R(0) ? pdf : R(0);
}

/* ------------------------------------------------------------------------------------------------
 * a6: perspective_projection_matrix + mi_world_to_screen      myutils/mi_plugin.py:585-595,645-671
 *     view = inverse(to_world) with to_world = diag(-1,1,-1,1) (myutils/default_cam.json).
 * ---------------------------------------------------------------------------------------------- */
void oracle_world_to_screen(const real p[3], real fov_x_rad, real aspect, real near_, real far_, int width,
                            int height, real screen[2]) {
    real f = R(1) / tan(fov_x_rad / R(2));
    real cam[4] = {-p[0], p[1], -p[2], R(1)};          /* view_matrix @ (p,1) */
    real clip[4];
    clip[0] = f / aspect * cam[0];
    clip[1] = f * cam[1];
    clip[2] = (far_ + near_) / (near_ - far_) * cam[2] + (R(2) * far_ * near_) / (near_ - far_) * cam[3];
    clip[3] = -cam[2];
    real ndc0 = clip[0] / clip[3], ndc1 = clip[1] / clip[3];
    screen[0] = (ndc0 + R(1)) * R(0.5) * (real)width;  /* x_screen */
    screen[1] = (ndc1 + R(1)) * R(0.5) * (real)height; /* y_screen */
}

/* App. E / myutils/mesh_recon.py:17-25: f = (W/2)/tan(fov/2), c = (W-1)/2, (H-1)/2; pixel (i,j) centre
 * <-> world ((j-cx)/f*d, -(i-cy)/f*d, -d); view direction wo = -p/|p| (independent of d). */
void oracle_pixel_to_world(int i, int j, real depth, int H, int W, real fov_x_deg, real p[3]) {
    real f = (R(0.5) * (real)W) / tan(R(0.5) * fov_x_deg * O_PI / R(180));
    real cx = R(0.5) * (real)(W - 1), cy = R(0.5) * (real)(H - 1);
    p[0] = ((real)j - cx) / f * depth;
    p[1] = -((real)i - cy) / f * depth;
    p[2] = -depth;
}
void oracle_view_dir(int i, int j, int H, int W, real fov_x_deg, real wo[3]) {
    real p[3];
    oracle_pixel_to_world(i, j, R(1), H, W, fov_x_deg, p);
    real l = sqrt(dot3(p, p));
    for (int k = 0; k < 3; ++k) wo[k] = -p[k] / l;
}

/* ------------------------------------------------------------------------------------------------
 * a10: order-4 real spherical harmonics, 25 coefficients, index l(l+1)+m.
 *      Convention of myutils/computeSH.py:13-68 (associated Legendre with Condon-Shortley sign,
 *      K = sqrt((2l+1)(l-|m|)!/(4pi (l+|m|)!)), sqrt2 sin(|m|phi) / 1 / sqrt2 cos(m phi)).
 *      Angles <-> world direction follow myutils/envmap_utils.py:29-36 and computeSH.py:233-235:
 *      theta = acos(y) (pole = +y), phi = atan2(x, -z) (phi = 0 <-> camera-forward -z).
 * ---------------------------------------------------------------------------------------------- */
static real fact(int n) { real f = R(1); for (int i = 2; i <= n; ++i) f *= (real)i; return f; }
real oracle_sh_K(int l, int m) {
    if (m < 0) m = -m;
    return sqrt((real)(2 * l + 1) * fact(l - m) / fact(l + m) / R(4) / O_PI);
}
void oracle_sh_basis_angles(real theta, real phi, real Y[MATPBR_NSH]) {
    real ct = cos(theta), st = sin(theta);
    real P[5][5];
    P[0][0] = R(1);
    P[1][0] = ct; P[1][1] = -st;
    P[2][0] = R(0.5) * (R(3) * ct * ct - R(1)); P[2][1] = -R(3) * ct * st; P[2][2] = R(3) * st * st;
    P[3][0] = R(0.5) * (R(5) * ct * ct * ct - R(3) * ct);
    P[3][1] = -R(1.5) * (R(5) * ct * ct - R(1)) * st;
    P[3][2] = R(15) * ct * st * st;
    P[3][3] = -R(15) * st * st * st;
    P[4][0] = R(0.125) * (R(35) * ct * ct * ct * ct - R(30) * ct * ct + R(3));
    P[4][1] = -R(2.5) * (R(7) * ct * ct * ct - R(3) * ct) * st;
    P[4][2] = R(7.5) * (R(7) * ct * ct - R(1)) * st * st;
    P[4][3] = -R(105) * ct * st * st * st;
    P[4][4] = R(105) * st * st * st * st;
    const real s2 = sqrt(R(2));
    for (int l = 0; l <= 4; ++l)
        for (int m = -l; m <= l; ++m) {
            int am = m < 0 ? -m : m;
            real K = oracle_sh_K(l, am);
            real v = K * P[l][am];
            if (m < 0) v *= s2 * sin((real)am * phi);
            else if (m > 0) v *= s2 * cos((real)am * phi);
            Y[l * (l + 1) + m] = v;
        }
}
void oracle_dir_to_angles(const real w[3], real* theta, real* phi) {
    real y = w[1];
    if (y > R(1)) y = R(1);
    if (y < -R(1)) y = -R(1);
    *theta = acos(y);
    *phi = atan2(w[0], -w[2]);
}
void oracle_sh_basis_dir(const real w[3], real Y[MATPBR_NSH]) {
    real th, ph;
    oracle_dir_to_angles(w, &th, &ph);
    oracle_sh_basis_angles(th, ph, Y);
}
/* radiance L(w)[c] = sum_k coef[k][c] * Y_k(w)      (computeSH.py:165-224 `projection`) */
void oracle_sh_eval(const real w[3], const real* coef /*[25][3]*/, real L[3]) {
    real Y[MATPBR_NSH];
    oracle_sh_basis_dir(w, Y);
    L[0] = L[1] = L[2] = R(0);
    for (int k = 0; k < MATPBR_NSH; ++k)
        for (int c = 0; c < 3; ++c) L[c] += coef[k * 3 + c] * Y[k];
}

/* ------------------------------------------------------------------------------------------------
 * Sample sets.
 *  (1) Reference-literal estimator ("MIS", kept as the yardstick the production estimator is measured
 *      against): n = spp/2 points per lobe, u0_i = (i+0.5)/n, u1_i = vdC_2(i) + 0.5/m, m = 2^ceil(log2 n);
 *      samples [0, n) use the diffuse sampler (reference: sample1 > 0.5), [n, 2n) the specular one.
 *  (2) Production rule: per lobe a product rule over the sampler's own (u0, u1)
 *      square -- nu Gauss-Legendre nodes x nphi equally spaced azimuths, ring k rotated by vdC_2(k)/nphi;
 *      the specular lobe places its nodes v_k through u0 = 1 - (1 - v)^2 (weight 2 (1 - v)), which removes the
 *      square-root end-point behaviour of the integrand at u0 -> 1 (grazing half vectors).  Sizes follow `spp`:
 *      at the reference's spp = 64 (5 x 4 specular, 3 x 6 diffuse nodes) its error against the converged integral
 *      is below the reference-literal estimator's at the same spp (tests/test_estimator_accuracy.py).
 * ---------------------------------------------------------------------------------------------- */
static real vdc2(uint32_t i) {
    i = (i << 16) | (i >> 16);
    i = ((i & 0x55555555u) << 1) | ((i & 0xAAAAAAAAu) >> 1);
    i = ((i & 0x33333333u) << 2) | ((i & 0xCCCCCCCCu) >> 2);
    i = ((i & 0x0F0F0F0Fu) << 4) | ((i & 0xF0F0F0F0u) >> 4);
    i = ((i & 0x00FF00FFu) << 8) | ((i & 0xFF00FF00u) >> 8);
    return (real)((double)i * 2.3283064365386963e-10);
}
void oracle_sample_point(int spp, int i, real* u0, real* u1) {
    int n = spp / 2;
    int m = 1;
    while (m < n) m <<= 1;
    *u0 = ((real)i + R(0.5)) / (real)n;
    *u1 = vdc2((uint32_t)i) + R(0.5) / (real)m;
}

#define ORACLE_MAX_RULE 96
/* Gauss-Legendre nodes / weights on [0,1] (Newton iteration on P_n, ascending nodes, weights sum to 1) */
static void gauss_legendre01(int n, double* x, double* w) {
    for (int i = 0; i < n; ++i) {
        double z = cos(3.14159265358979323846 * ((double)i + 0.75) / ((double)n + 0.5)), pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 0; j < n; ++j) {
                double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j + 1.0) * z * p2 - (double)j * p3) / ((double)j + 1.0);
            }
            pp = (double)n * (z * p1 - p2) / (z * z - 1.0);
            double z1 = z;
            z = z1 - p1 / pp;
            if (fabs(z - z1) < 1e-15) break;
        }
        x[n - 1 - i] = 0.5 * (z + 1.0);
        w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
    }
}
/* spp -> (nu, nphi) of the specular (lobe 1) and diffuse (lobe 0) rules; spp = 64 -> 5 x 4 and 3 x 6 */
static double g_rule_power = 2.0;    /* specular lobe: u0 = 1 - (1-v)^2 at the Gauss-Legendre nodes v (other powers: studies only) */
void oracle_rule_power(double p) { g_rule_power = p; }
static int g_rule_override[2][2];   /* quadrature studies only (tools/quadrature_study.py): 0 = use the production sizes */
void oracle_rule_override(int lobe, int nu, int nphi) { g_rule_override[lobe][0] = nu; g_rule_override[lobe][1] = nphi; }
void oracle_rule_dims(int spp, int lobe, int* nu, int* nphi) {
    double q = 0.25 * (double)spp, s = sqrt(q);
    if (g_rule_override[lobe][0] > 0) { *nu = g_rule_override[lobe][0]; *nphi = g_rule_override[lobe][1]; return; }
    if (lobe) {
        *nu = (int)floor(s + 1.5);
        *nphi = (int)floor(s + 0.5);
        if (*nphi < 1) *nphi = 1;
    } else {
        *nu = (int)floor(0.75 * s + 0.5);
        if (*nu < 1) *nu = 1;
        *nphi = 2 * *nu;
    }
}
int oracle_rule(int spp, int lobe, real* u0, real* u1, real* w) {
    int nu, nphi;
    double x[ORACLE_MAX_RULE], wx[ORACLE_MAX_RULE];
    oracle_rule_dims(spp, lobe, &nu, &nphi);
    if (nu * nphi > ORACLE_MAX_RULE) return -1;
    gauss_legendre01(nu, x, wx);
    int c = 0;
    for (int k = 0; k < nu; ++k) {
        double off = (double)vdc2((uint32_t)k) / (double)nphi;
        for (int j = 0; j < nphi; ++j, ++c) {
            double v = ((double)j + 0.5) / (double)nphi + off;
            u0[c] = (real)x[k];
            u1[c] = (real)(v - floor(v));
            w[c] = (real)(wx[k] / (double)nphi);
            if (lobe && g_rule_power != 1.0) {
                u0[c] = (real)(1.0 - pow(1.0 - x[k], g_rule_power));
                w[c] = (real)(wx[k] * g_rule_power * pow(1.0 - x[k], g_rule_power - 1.0) / (double)nphi);
            }
        }
    }
    return c;
}

typedef struct Rules {   /* both lobes' rules of one spp, built once per image-level call */
    int nd, ns;
    real du0[ORACLE_MAX_RULE], du1[ORACLE_MAX_RULE], dw[ORACLE_MAX_RULE];
    real su0[ORACLE_MAX_RULE], su1[ORACLE_MAX_RULE], sw[ORACLE_MAX_RULE];
} Rules;
static void rules_init(Rules* q, int spp) {
    q->nd = oracle_rule(spp, 0, q->du0, q->du1, q->dw);
    q->ns = oracle_rule(spp, 1, q->su0, q->su1, q->sw);
}

/* ------------------------------------------------------------------------------------------------
 * Reference-literal estimator for one pixel (the round-1 definition of the render, now the yardstick):
 *   L_o = (1/spp) sum_s  w_s * L_SH(wi_s),   (wi_s, w_s) = sample_brdf(lobe_s, u_s; wo, n, a, r, m)
 * i.e. the BSDF-sampling arm of a depth-1 path (direct light from the environment, unshadowed) with
 * the one-sample-model mixture pdf of eval_brdf (:1398-1401) and the MC weight of sample_brdf (:1335-1341).
 * For spp -> infinity it converges to  I = integral of eval_brdf(wi) L(wi) dwi  over the hemisphere.
 * ---------------------------------------------------------------------------------------------- */
static void mis_pixel(const real wo[3], const real n[3], const real a[3], real r, real m, const real* coef,
                      int spp, real out[3]) {
    int half = spp / 2;
    out[0] = out[1] = out[2] = R(0);
    for (int s = 0; s < spp; ++s) {
        real u0, u1, wi[3], pdf, w[3], L[3];
        oracle_sample_point(spp, s < half ? s : s - half, &u0, &u1);
        oracle_sample_brdf(s < half ? R(1) : R(0), u0, u1, wo, n, a, r, m, wi, &pdf, w);
        oracle_sh_eval(wi, coef, L);
        for (int c = 0; c < 3; ++c) out[c] += w[c] * L[c];
    }
    for (int c = 0; c < 3; ++c) out[c] /= (real)spp;
}
/* Its backward with sample directions and pdf treated as constants (stop-gradient), the convention of the
 * reference's torch variants (`D.data`, `alpha.data`: myutils/mi_plugin.py:366,179): a consistent estimator of dI. */
static void mis_pixel_bwd(const real wo[3], const real n[3], const real a[3], real r, real m, const real* coef,
                          int spp, const real g_out[3], real d_a[3], real* d_r, real* d_m, real d_n[3],
                          real* d_coef /*[25][3] accumulated, may be NULL*/) {
    int half = spp / 2;
    d_a[0] = d_a[1] = d_a[2] = R(0);
    *d_r = R(0); *d_m = R(0);
    d_n[0] = d_n[1] = d_n[2] = R(0);
    for (int s = 0; s < spp; ++s) {
        real u0, u1, wi[3], f[3], pdf, Y[MATPBR_NSH], L[3] = {0, 0, 0};
        oracle_sample_point(spp, s < half ? s : s - half, &u0, &u1);
        if (s < half) oracle_diffuse_sampler(u0, u1, n, wi);
        else oracle_specular_sampler(u0, u1, r, wo, n, wi);
        oracle_eval_brdf(wi, wo, n, a, r, m, f, &pdf);
        if (!(pdf > R(1e-6))) continue;
        real inv = R(1) / ((pdf + R(1e-6)) * (real)spp);
        oracle_sh_basis_dir(wi, Y);
        for (int k = 0; k < MATPBR_NSH; ++k)
            for (int c = 0; c < 3; ++c) L[c] += coef[k * 3 + c] * Y[k];
        real g[3], ga[3], gr, gm, gn[3];
        for (int c = 0; c < 3; ++c) g[c] = g_out[c] * L[c] * inv;
        oracle_eval_brdf_grad(wi, wo, n, a, r, m, g, ga, &gr, &gm, gn);
        for (int c = 0; c < 3; ++c) { d_a[c] += ga[c]; d_n[c] += gn[c]; }
        *d_r += gr; *d_m += gm;
        if (d_coef)
            for (int k = 0; k < MATPBR_NSH; ++k)
                for (int c = 0; c < 3; ++c) d_coef[k * 3 + c] += g_out[c] * f[c] * inv * Y[k];
    }
}

/* ------------------------------------------------------------------------------------------------
 * a8: the production render  R(a, r, m, n; light)  for one pixel -- the same integral I, with the two
 * lobes of eval_brdf (:1405-1415) integrated separately, each with its own sampler of sample_brdf:
 *
 *   diffuse (:1405-1409), cosine-weighted directions wi_j (mi_diffuse_sampler, pdf = NoL/pi, weights w_j):
 *     I_d[c] = a_c (1-m) * sum_j w_j F_out F_in L_c(wi_j),   F_x = 1 + (F_D90 - 1)(1 - x)^5,
 *     F_D90 - 1 = 2 VoH^2 r - 1/2 = r (1 + wi.wo) - 1/2   (VoH^2 = (1 + wi.wo)/2 for h = normalize(wi + wo))
 *     => I_d[c] = a_c (1-m) (A0_c + r A1_c + r^2 A2_c): a quadratic in r whose coefficients depend on (n, wo, light)
 *     only.  With light and geometric normals fixed during a BRDF phase (F10) they are constants of the phase.
 *   specular (:1410-1414), GGX half vectors (mi_specular_sampler, pdf = D NoH / (4 VoH)): D cancels between the
 *     value and the pdf,  f cos / pdf = G1(NoL) G1(NoV) F_m NoL VoH / NoH,  F_m = C0 + (1 - C0)(1 - VoH)^5:
 *     I_s[c] = C0_c S0_c + (1 - C0_c) S1_c,  S0 = sum_j w_j g_j L(wi_j),  S1 = sum_j w_j g_j (1-VoH_j)^5 L(wi_j),
 *     g = G1(NoL) G1(NoV) NoL VoH / NoH on samples with wo.h > 0 and NoL > 0.
 * Backward: sample directions and pdfs are constants (stop-gradient), as before; then
 *     d ln(f_s)/dr at a fixed direction = 4/r - 8 r^3 NoH^2/den + d ln G / dr   (= (4/r)(2 u0 - 1) + d ln G/dr up to D_GGX's 1e-6)
 * so the material gradients are closed forms of {A, S0, S1, dS0/dr, dS1/dr}.
 * ---------------------------------------------------------------------------------------------- */
typedef struct SplitState {
    real P[3], dP[3];        /* A0 + r A1 + r^2 A2 and its r-derivative */
    real S0[3], S1[3], dS0[3], dS1[3];
} SplitState;

static void diffuse_local(real u0, real u1, real l[3]) {      /* mi_diffuse_sampler :255-281, local frame */
    real theta = asin(sqrt(rmax(u0, R(0))));
    real phi = R(2) * O_PI * u1;
    l[0] = sin(theta) * cos(phi); l[1] = sin(theta) * sin(phi); l[2] = cos(theta);
}
static real pow4r(real x) { real x2 = x * x; return x2 * x2; }

/* A0, A1, A2 of the diffuse lobe (per channel) */
static void diffuse_coef(const real wo[3], const real n[3], const real* coef, const Rules* q, real A[9]) {
    const real *u0 = q->du0, *u1 = q->du1, *w = q->dw;
    const int nd = q->nd;
    real s[3], t[3];
    oracle_frame(n, s, t);
    real po = pow5(R(1) - rmax(dot3(n, wo), R(0)));
    real M[5][3];
    memset(M, 0, sizeof(M));
    for (int j = 0; j < nd; ++j) {
        real l[3], wi[3], L[3];
        diffuse_local(u0[j], u1[j], l);
        to_world(s, t, n, l, wi);
        oracle_sh_eval(wi, coef, L);
        real u = R(1) + dot3(wi, wo), p5 = pow5(R(1) - l[2]);
        for (int c = 0; c < 3; ++c) {
            real x = w[j] * L[c];
            M[0][c] += x; M[1][c] += x * p5; M[2][c] += x * u; M[3][c] += x * u * p5; M[4][c] += x * u * u * p5;
        }
    }
    for (int c = 0; c < 3; ++c) {
        A[c] = M[0][c] * (R(1) - R(0.5) * po) + M[1][c] * (R(0.25) * po - R(0.5));
        A[3 + c] = po * M[2][c] + (R(1) - po) * M[3][c];
        A[6 + c] = po * M[4][c];
    }
}

/* GGX-sampled specular sample j in the shading frame of n: returns 0 when the sample carries no weight */
typedef struct SpecSample { real wi[3], wh[3], NoL, VoH, NoH, g1l, wgt, x5, lam; } SpecSample;
static int spec_sample(real u0, real u1, real w, const real wo[3], const real n[3], const real s[3], const real t[3], real r,
                       real g1v, real NoV, SpecSample* sp) {
    real alpha2 = pow4r(r);
    real q = R(1) / (u0 * (alpha2 - R(1)) + R(1));
    real ct = sqrt(rmax((R(1) - u0) * q, R(0))), st = sqrt(rmax(alpha2 * u0 * q, R(0)));   /* :232-233 */
    real phi = R(2) * O_PI * u1;
    real l[3] = {st * cos(phi), st * sin(phi), ct};
    to_world(s, t, n, l, sp->wh);
    real d = dot3(wo, sp->wh);
    for (int i = 0; i < 3; ++i) sp->wi[i] = R(2) * d * sp->wh[i] - wo[i];                     /* reflect, :245 */
    real nl = dot3(n, sp->wi);
    if (!(d > R(0)) || !(nl > R(0))) return 0;
    real k = (r + R(1)) * (r + R(1)) / R(8);
    sp->NoL = nl; sp->VoH = d; sp->NoH = ct;
    sp->g1l = oracle_G1_GGX_Schlick(nl, r);
    sp->wgt = w * sp->g1l * g1v * nl * d / ct;
    sp->x5 = pow5(R(1) - d);
    /* d ln(D G)/dr at a fixed direction; D = D_GGX of :89-97 including its 1e-6 regulariser:
     * d ln D/dr = 4/r - 8 r^3 NoH^2 / den  (= (4/r)(2 u0 - 1) up to the regulariser, since den = alpha2/(1 + u0 (alpha2 - 1)) here) */
    real den = ct * ct * (alpha2 - R(1)) + R(1) + R(1e-6);
    sp->lam = R(4) / r - R(8) * r * r * r * ct * ct / den - (r + R(1)) / R(4) * (sp->g1l * (R(1) - nl) + g1v * (R(1) - NoV));
    (void)k;
    return 1;
}

static void split_pixel_state(const real wo[3], const real n[3], real r, const real* coef, const Rules* q, const real* A_cached,
                              SplitState* st) {
    real A[9];
    if (A_cached) memcpy(A, A_cached, sizeof(A));
    else diffuse_coef(wo, n, coef, q, A);
    for (int c = 0; c < 3; ++c) {
        st->P[c] = A[c] + r * A[3 + c] + r * r * A[6 + c];
        st->dP[c] = A[3 + c] + R(2) * r * A[6 + c];
        st->S0[c] = st->S1[c] = st->dS0[c] = st->dS1[c] = R(0);
    }
    const real *u0 = q->su0, *u1 = q->su1, *w = q->sw;
    const int ns = q->ns;
    real s[3], t[3];
    oracle_frame(n, s, t);
    real NoV = rmax(dot3(n, wo), R(0)), g1v = oracle_G1_GGX_Schlick(NoV, r);
    for (int j = 0; j < ns; ++j) {
        SpecSample sp;
        real L[3];
        if (!spec_sample(u0[j], u1[j], w[j], wo, n, s, t, r, g1v, NoV, &sp)) continue;
        oracle_sh_eval(sp.wi, coef, L);
        for (int c = 0; c < 3; ++c) {
            real x = sp.wgt * L[c];
            st->S0[c] += x; st->S1[c] += x * sp.x5;
            st->dS0[c] += x * sp.lam; st->dS1[c] += x * sp.lam * sp.x5;
        }
    }
}
static void split_pixel(const real wo[3], const real n[3], const real a[3], real r, real m, const real* coef, const Rules* q,
                        const real* A_cached, real out[3]) {
    SplitState st;
    split_pixel_state(wo, n, r, coef, q, A_cached, &st);
    for (int c = 0; c < 3; ++c) {
        real C0 = (R(1) - m) * R(0.04) + m * a[c];                                            /* :1412 */
        out[c] = a[c] * (R(1) - m) * st.P[c] + C0 * st.S0[c] + (R(1) - C0) * st.S1[c];
    }
}

/* Backward of split_pixel (stop-gradient through directions and pdfs).  d_n is the gradient w.r.t. the unit normal in world
 * space (its radial part is removed by the caller); d_coef[25][3] is accumulated when not NULL. */
static void split_pixel_bwd(const real wo[3], const real n[3], const real a[3], real r, real m, const real* coef, const Rules* q,
                            const real g_out[3], real d_a[3], real* d_r, real* d_m, real d_n[3], real* d_coef) {
    SplitState st;
    split_pixel_state(wo, n, r, coef, q, NULL, &st);
    *d_r = R(0); *d_m = R(0);
    d_n[0] = d_n[1] = d_n[2] = R(0);
    real C0[3];
    for (int c = 0; c < 3; ++c) {
        C0[c] = (R(1) - m) * R(0.04) + m * a[c];
        real SD = st.S0[c] - st.S1[c];
        d_a[c] = g_out[c] * ((R(1) - m) * st.P[c] + m * SD);
        *d_m += g_out[c] * (-a[c] * st.P[c] + (a[c] - R(0.04)) * SD);
        *d_r += g_out[c] * (a[c] * (R(1) - m) * st.dP[c] + C0[c] * st.dS0[c] + (R(1) - C0[c]) * st.dS1[c]);
    }
    /* per-sample parts: normal and light gradients */
    const real *u0 = q->du0, *u1 = q->du1, *w = q->dw;
    real s[3], t[3];
    oracle_frame(n, s, t);
    real nv = dot3(n, wo), NoV = rmax(nv, R(0)), po = pow5(R(1) - NoV);
    real k = (r + R(1)) * (r + R(1)) / R(8), alpha2 = pow4r(r);
    const int nd = q->nd;
    for (int j = 0; j < nd; ++j) {
        real l[3], wi[3], Y[MATPBR_NSH], L[3] = {0, 0, 0};
        diffuse_local(u0[j], u1[j], l);
        to_world(s, t, n, l, wi);
        oracle_sh_basis_dir(wi, Y);
        for (int kk = 0; kk < MATPBR_NSH; ++kk)
            for (int c = 0; c < 3; ++c) L[c] += coef[kk * 3 + c] * Y[kk];
        real q = r * (R(1) + dot3(wi, wo)) - R(0.5);
        real p5 = pow5(R(1) - l[2]);
        real Fo = R(1) + q * po, Fi = R(1) + q * p5;
        real dFi = -R(5) * q * pow4r(R(1) - l[2]);
        real dFo = nv > R(0) ? -R(5) * q * pow4r(R(1) - NoV) : R(0);
        real sc = R(0);
        for (int c = 0; c < 3; ++c) sc += g_out[c] * a[c] * (R(1) - m) * L[c];
        sc *= w[j];
        real gl = Fo * (dFi + Fi / l[2]), gv = dFo * Fi;
        for (int i = 0; i < 3; ++i) d_n[i] += sc * (gl * wi[i] + gv * wo[i]);
        if (d_coef)
            for (int kk = 0; kk < MATPBR_NSH; ++kk)
                for (int c = 0; c < 3; ++c) d_coef[kk * 3 + c] += g_out[c] * a[c] * (R(1) - m) * w[j] * Fo * Fi * Y[kk];
    }
    const int ns = q->ns;
    u0 = q->su0; u1 = q->su1; w = q->sw;
    real g1v = oracle_G1_GGX_Schlick(NoV, r);
    for (int j = 0; j < ns; ++j) {
        SpecSample sp;
        real Y[MATPBR_NSH], L[3] = {0, 0, 0};
        if (!spec_sample(u0[j], u1[j], w[j], wo, n, s, t, r, g1v, NoV, &sp)) continue;
        oracle_sh_basis_dir(sp.wi, Y);
        for (int kk = 0; kk < MATPBR_NSH; ++kk)
            for (int c = 0; c < 3; ++c) L[c] += coef[kk * 3 + c] * Y[kk];
        real sc = R(0);
        for (int c = 0; c < 3; ++c) sc += g_out[c] * (C0[c] + (R(1) - C0[c]) * sp.x5) * L[c];
        sc *= sp.wgt;
        real den = sp.NoH * sp.NoH * (alpha2 - R(1)) + R(1) + R(1e-6);                       /* :95 */
        real gl = R(1) / sp.NoL - sp.g1l * (R(1) - k);
        real gv = nv > R(0) ? -g1v * (R(1) - k) : R(0);
        real gh = -R(4) * sp.NoH * (alpha2 - R(1)) / den;
        for (int i = 0; i < 3; ++i) d_n[i] += sc * (gl * sp.wi[i] + gv * wo[i] + gh * sp.wh[i]);
        if (d_coef)
            for (int kk = 0; kk < MATPBR_NSH; ++kk)
                for (int c = 0; c < 3; ++c) d_coef[kk * 3 + c] += g_out[c] * sp.wgt * (C0[c] + (R(1) - C0[c]) * sp.x5) * Y[kk];
    }
}

/* The image-level render shades with the unit normal n/|n| of the stored map (the reference's geometric
 * and MaterialNet normals are unit length already; dpt.py normalises its normal head). */
static real unit_normal(const real n[3], real nh[3]) {
    real l = sqrt(dot3(n, n));
    if (!(l > R(0))) { nh[0] = R(0); nh[1] = R(0); nh[2] = R(1); return R(1); }
    for (int c = 0; c < 3; ++c) nh[c] = n[c] / l;
    return l;
}

/* Forward with the sampling state frozen: directions and pdfs come from (n_s, r_s), the BRDF value from (n, a, r, m),
 * evaluated lobe by lobe with the literal formulas (no closed forms).  At (n_s, r_s) = (n, r) this equals split_pixel;
 * its derivative w.r.t. (a, r, m, n) at that point is what split_pixel_bwd returns, which the tests verify by finite
 * differences -- so the closed forms of the backward pass are checked against an independent evaluation. */
static void split_pixel_frozen(const real wo[3], const real n[3], const real a[3], real r, real m, const real n_s[3], real r_s,
                               const real* coef, const Rules* rq, real out[3]) {
    const real *u0 = rq->du0, *u1 = rq->du1, *w = rq->dw;
    real s[3], t[3];
    oracle_frame(n_s, s, t);
    real NoV = rmax(dot3(n, wo), R(0)), po = pow5(R(1) - NoV);
    out[0] = out[1] = out[2] = R(0);
    const int nd = rq->nd;
    for (int j = 0; j < nd; ++j) {
        real l[3], wi[3], L[3];
        diffuse_local(u0[j], u1[j], l);
        to_world(s, t, n_s, l, wi);
        oracle_sh_eval(wi, coef, L);
        real NoL = rmax(dot3(n, wi), R(0));
        real h[3] = {wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]};
        real hl = sqrt(dot3(h, h));
        real VoH = rmax(dot3(wo, h) / hl, R(0));
        real FD90 = R(0.5) + R(2) * VoH * VoH * r;                                           /* :1406 */
        real Fo = R(1) + (FD90 - R(1)) * po, Fi = R(1) + (FD90 - R(1)) * pow5(R(1) - NoL);  /* :1407-1408 */
        for (int c = 0; c < 3; ++c) {
            real val = a[c] * (R(1) - m) / O_PI * Fo * Fi * NoL;                              /* :1409 */
            out[c] += w[j] * val / (l[2] / O_PI) * L[c];                                      /* pdf_diff = NoL_s / pi, :1400 */
        }
    }
    const int ns = rq->ns;
    u0 = rq->su0; u1 = rq->su1; w = rq->sw;
    real a2s = pow4r(r_s);
    for (int j = 0; j < ns; ++j) {
        real q = R(1) / (u0[j] * (a2s - R(1)) + R(1));
        real ct = sqrt(rmax((R(1) - u0[j]) * q, R(0))), st = sqrt(rmax(a2s * u0[j] * q, R(0)));
        real phi = R(2) * O_PI * u1[j];
        real l[3] = {st * cos(phi), st * sin(phi), ct}, wh[3], wi[3], L[3];
        to_world(s, t, n_s, l, wh);
        real d = dot3(wo, wh);
        for (int i = 0; i < 3; ++i) wi[i] = R(2) * d * wh[i] - wo[i];
        if (!(d > R(0)) || !(dot3(n_s, wi) > R(0))) continue;
        real pdf = oracle_D_GGX(ct, r_s) * ct / (R(4) * d);                                         /* :1399 */
        oracle_sh_eval(wi, coef, L);
        real NoL = rmax(dot3(n, wi), R(0)), NoH = rmax(dot3(n, wh), R(0));
        real G = oracle_G_Smith(NoV, NoL, r);
        real x5 = pow5(R(1) - d);
        for (int c = 0; c < 3; ++c) {
            real C0 = (R(1) - m) * R(0.04) + m * a[c];
            real val = oracle_D_GGX(NoH, r) * G * (C0 + (R(1) - C0) * x5) / R(4) * NoL;            /* :1413-1414 */
            out[c] += w[j] * val / pdf * L[c];
        }
    }
}

/* ---- image-level drivers.  Layout = the reference's: row-major HWC float maps a[H,W,3] r[H,W,1] m[H,W,1] n[H,W,3]
 * (myutils/mi_plugin.py:1238-1241), light = SH coef [batch,25,3].  A window (h x w at (i0, j0) of an H x W image) or explicit
 * per-lane view directions select where the view direction comes from. */
typedef struct View { int H, W, i0, j0, w; real fov; const real* wo; } View;
static void view_of(const View* v, long p, real wo[3]) {
    if (v->wo) { for (int c = 0; c < 3; ++c) wo[c] = v->wo[p * 3 + c]; return; }
    oracle_view_dir(v->i0 + (int)(p / v->w), v->j0 + (int)(p % v->w), v->H, v->W, v->fov, wo);
}
static void fwd_driver(const real* a, const real* r, const real* m, const real* n, const real* light, real* out, long P, int batch,
                       int spp, const View* v, int kind, const real* dcache) {
    Rules q;
    rules_init(&q, spp);
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < P * batch; ++idx) {
        long b = idx / P, p = idx % P;
        real wo[3], nh[3];
        view_of(v, v->wo ? idx : p, wo);
        unit_normal(n + idx * 3, nh);
        const real* coef = light + b * MATPBR_NSH * 3;
        if (kind == 1) mis_pixel(wo, nh, a + idx * 3, r[idx], m[idx], coef, spp, out + idx * 3);
        else split_pixel(wo, nh, a + idx * 3, r[idx], m[idx], coef, &q, dcache ? dcache + idx * 9 : NULL, out + idx * 3);
    }
}
static void bwd_driver(const real* a, const real* r, const real* m, const real* n, const real* light, const real* d_out, real* d_a,
                       real* d_r, real* d_m, real* d_n, real* d_light, long P, int batch, int spp, const View* v, int kind) {
    Rules q;
    rules_init(&q, spp);
    if (d_light) memset(d_light, 0, sizeof(real) * (size_t)batch * MATPBR_NSH * 3);
    for (long b = 0; b < batch; ++b) {
#pragma omp parallel
        {
            real acc[MATPBR_NSH * 3];
            memset(acc, 0, sizeof(acc));
#pragma omp for schedule(static)
            for (long p = 0; p < P; ++p) {
                long idx = b * P + p;
                real wo[3], ga[3], gr, gm, gn[3], nh[3];
                view_of(v, v->wo ? idx : p, wo);
                real nlen = unit_normal(n + idx * 3, nh);
                const real* coef = light + b * MATPBR_NSH * 3;
                if (kind == 1) mis_pixel_bwd(wo, nh, a + idx * 3, r[idx], m[idx], coef, spp, d_out + idx * 3, ga, &gr, &gm, gn, d_light ? acc : NULL);
                else split_pixel_bwd(wo, nh, a + idx * 3, r[idx], m[idx], coef, &q, d_out + idx * 3, ga, &gr, &gm, gn, d_light ? acc : NULL);
                {   /* through n_hat = n/|n|: d_n = (g - n_hat (n_hat.g)) / |n| */
                    real pr = dot3(nh, gn);
                    for (int c = 0; c < 3; ++c) gn[c] = (gn[c] - nh[c] * pr) / nlen;
                }
                if (d_a) for (int c = 0; c < 3; ++c) d_a[idx * 3 + c] = ga[c];
                if (d_r) d_r[idx] = gr;
                if (d_m) d_m[idx] = gm;
                if (d_n) for (int c = 0; c < 3; ++c) d_n[idx * 3 + c] = gn[c];
            }
            if (d_light) {
#pragma omp critical
                for (int k = 0; k < MATPBR_NSH * 3; ++k) d_light[b * MATPBR_NSH * 3 + k] += acc[k];
            }
        }
    }
}

/* kind: 0 = production estimator, 1 = reference-literal MIS estimator */
void oracle_shade_fwd_kind(const real* a, const real* r, const real* m, const real* n, const real* light, real* out,
                           int H, int W, int batch, int spp, real fov_x_deg, int kind) {
    View v = {H, W, 0, 0, W, fov_x_deg, NULL};
    fwd_driver(a, r, m, n, light, out, (long)H * W, batch, spp, &v, kind, NULL);
}
void oracle_shade_fwd(const real* a, const real* r, const real* m, const real* n, const real* light, real* out,
                      int H, int W, int batch, int spp, real fov_x_deg) {
    oracle_shade_fwd_kind(a, r, m, n, light, out, H, W, batch, spp, fov_x_deg, 0);
}
/* The same render for an h x w window whose top-left pixel is (i0, j0) of an H x W image: view directions are those of
 * the full image (used to check large renders on a crop in seconds). */
void oracle_shade_fwd_win(const real* a, const real* r, const real* m, const real* n, const real* light, real* out,
                          int h, int w, int batch, int spp, real fov_x_deg, int H, int W, int i0, int j0, int kind) {
    View v = {H, W, i0, j0, w, fov_x_deg, NULL};
    fwd_driver(a, r, m, n, light, out, (long)h * w, batch, spp, &v, kind, NULL);
}
/* N independent lanes with explicit view directions and one light (quadrature studies, stress cases). */
void oracle_shade_fwd_lanes(const real* a, const real* r, const real* m, const real* n, const real* wo, const real* light,
                            real* out, long N, int spp, int kind) {
    View v = {0, 0, 0, 0, 1, R(0), wo};
    fwd_driver(a, r, m, n, light, out, N, 1, spp, &v, kind, NULL);
}
void oracle_shade_bwd_kind(const real* a, const real* r, const real* m, const real* n, const real* light,
                           const real* d_out, real* d_a, real* d_r, real* d_m, real* d_n /*nullable*/,
                           real* d_light /*nullable, [batch,25,3], overwritten*/, int H, int W, int batch, int spp,
                           real fov_x_deg, int kind) {
    View v = {H, W, 0, 0, W, fov_x_deg, NULL};
    bwd_driver(a, r, m, n, light, d_out, d_a, d_r, d_m, d_n, d_light, (long)H * W, batch, spp, &v, kind);
}
void oracle_shade_bwd(const real* a, const real* r, const real* m, const real* n, const real* light,
                      const real* d_out, real* d_a, real* d_r, real* d_m, real* d_n, real* d_light, int H, int W, int batch, int spp,
                      real fov_x_deg) {
    oracle_shade_bwd_kind(a, r, m, n, light, d_out, d_a, d_r, d_m, d_n, d_light, H, W, batch, spp, fov_x_deg, 0);
}
/* SURVEY.md 8b names the CPU twin of the C ABI matpbr_shade_{fwd,bwd}_cpu: host pointers, the argument meaning of
 * matpbr_shade_fwd / matpbr_shade_bwd for an SH25 light (include/matpbr.h).  They live in the ORACLE library, not in libmatpbr.so:
 * the product path has no CPU fallback. */
int matpbr_shade_fwd_cpu(const real* a, const real* r, const real* m, const real* n, const real* light, real* out_rgb, int H, int W,
                         int batch, int spp, real fov_x_deg) {
    if (!a || !r || !m || !n || !light || !out_rgb || H <= 0 || W <= 0 || batch <= 0 || spp < 2 || (spp & 1)) return -1;
    oracle_shade_fwd(a, r, m, n, light, out_rgb, H, W, batch, spp, fov_x_deg);
    return 0;
}
int matpbr_shade_bwd_cpu(const real* a, const real* r, const real* m, const real* n, const real* light, const real* d_out_rgb, real* d_a,
                         real* d_r, real* d_m, real* d_n, real* d_light, int H, int W, int batch, int spp, real fov_x_deg) {
    if (!a || !r || !m || !n || !light || !d_out_rgb || !d_a || !d_r || !d_m || H <= 0 || W <= 0 || batch <= 0 || spp < 2 || (spp & 1)) return -1;
    oracle_shade_bwd(a, r, m, n, light, d_out_rgb, d_a, d_r, d_m, d_n, d_light, H, W, batch, spp, fov_x_deg);
    return 0;
}
void oracle_shade_bwd_lanes(const real* a, const real* r, const real* m, const real* n, const real* wo, const real* light,
                            const real* d_out, real* d_a, real* d_r, real* d_m, real* d_n, real* d_light, long N, int spp, int kind) {
    View v = {0, 0, 0, 0, 1, R(0), wo};
    bwd_driver(a, r, m, n, light, d_out, d_a, d_r, d_m, d_n, d_light, N, 1, spp, &v, kind);
}
/* The diffuse-lobe coefficients A0, A1, A2 (rgb each) per pixel, dcache[B,H,W,9], and the render from them. */
void oracle_diffuse_cache(const real* n, const real* light, real* dcache, int H, int W, int batch, int spp, real fov_x_deg) {
    const long P = (long)H * W;
    Rules q;
    rules_init(&q, spp);
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < P * batch; ++idx) {
        long b = idx / P, p = idx % P;
        real wo[3], nh[3];
        oracle_view_dir((int)(p / W), (int)(p % W), H, W, fov_x_deg, wo);
        unit_normal(n + idx * 3, nh);
        diffuse_coef(wo, nh, light + b * MATPBR_NSH * 3, &q, dcache + idx * 9);
    }
}
void oracle_shade_fwd_cached(const real* a, const real* r, const real* m, const real* n, const real* light, const real* dcache,
                             real* out, int H, int W, int batch, int spp, real fov_x_deg) {
    View v = {H, W, 0, 0, W, fov_x_deg, NULL};
    fwd_driver(a, r, m, n, light, out, (long)H * W, batch, spp, &v, 0, dcache);
}
void oracle_shade_fwd_frozen(const real* a, const real* r, const real* m, const real* n, const real* n_s, const real* r_s,
                             const real* light, real* out, int H, int W, int batch, int spp, real fov_x_deg) {
    const long P = (long)H * W;
    Rules q;
    rules_init(&q, spp);
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < P * batch; ++idx) {
        long b = idx / P, p = idx % P;
        int i = (int)(p / W), j = (int)(p % W);
        real wo[3], nh[3], nsh[3];
        oracle_view_dir(i, j, H, W, fov_x_deg, wo);
        unit_normal(n + idx * 3, nh);
        unit_normal(n_s + idx * 3, nsh);
        split_pixel_frozen(wo, nh, a + idx * 3, r[idx], m[idx], nsh, r_s[idx], light + b * MATPBR_NSH * 3, &q, out + idx * 3);
    }
}
/* ------------------------------------------------------------------------------------------------
 * Lazy re-sampling of the specular sums in hot loop B (the parts of --opt_order that move the roughness,
 * inverse_img_w_mi.py:493-515 / 371-386).  This is the SPECIFICATION of materialist_amd/csrc/matpbr_lazy.hpp.
 *
 * With light and shading normals fixed during a BRDF phase (:317-342) the specular sums S0, S1 of a pixel are
 * functions of its roughness alone, and Adam moves r by at most lr ~ 3e-4 per step.  A pixel therefore keeps
 *     r_ref,  SD = S0 - S1 and S1 at r_ref,  their slopes gSD, gS1 (one-sided difference over LAZY_H in the direction
 *     the pixel is travelling),  the detached r-derivatives dSD, dS1 of the backward convention and THEIR
 *     slopes eSD, eS1 from the same one-sided difference (round 5: d out/d r is first order in dr like the render),
 * and renders  out = a (1-m) P(r) + C0 (SD + gSD dr) + (S1 + gS1 dr),  dr = r - r_ref,  with P(r) exact from the cached
 * diffuse coefficients, as long as r stays inside the pixel's validity interval [r_ref - lo, r_ref + hi].  Outside of it
 * the 20 GGX samples are walked again (refresh).  The interval is built so that the extrapolation stays well inside the
 * parity bar |lazy - exact| <= 1e-3 max(|exact|, mean|exact|):
 *   * smooth part: a radius rho, controlled like an ODE step size -- at a refresh the prediction of the old state is
 *     compared with the exact sums (relative error e at distance D), rho' = 0.9 D sqrt(tol_s / e), within [rho/2, 2 rho];
 *   * kinks: S0, S1 are C0 but not C1 in r where a sample's reflected direction crosses the horizon (NoL = max(n.wi, 0))
 *     or its half vector turns away from the view (VoH = max(wo.h, 0)).  Both crossings are predicted to first order
 *     from d(theta_h)/dr of the GGX sampler (:232-233); beyond a crossing the linear model is off by J |r - r_k| with
 *     J = the sample's weight per unit of the clamped variable times its r-derivative, so the interval ends
 *     tol_k / J behind the crossing (crossings too weak to matter inside rho_max are ignored).
 * Layout of a state: LAZY_NSTATE reals per pixel, indices LZ_*.
 * ---------------------------------------------------------------------------------------------- */
#define LAZY_NSTATE 28
enum { LZ_RREF = 0, LZ_LO, LZ_HI, LZ_RHO, LZ_SD = 4, LZ_S1 = 7, LZ_GSD = 10, LZ_GS1 = 13, LZ_DSD = 16, LZ_DS1 = 19, LZ_ESD = 22, LZ_ES1 = 25 };
#define LAZY_H R(1e-3)
#define LAZY_RHO_INIT R(2e-3)
#define LAZY_RHO_MIN R(2.5e-4)
#define LAZY_RHO_MAX R(3e-2)
#define LAZY_TOL_S R(2.5e-4)
#define LAZY_TOL_K R(1.5e-4)
#define LAZY_KINK_SAFETY R(0.8)
#define LAZY_MOVED R(1e-4)
/* round 6: the radius also answers to the derivative handed to the backward pass: the old state's prediction of d out_c / d r at the new roughness
 * (dSD + eSD dr, dS1 + eS1 dr) against the walked one, relative to max(|d out_c / d r|, LAZY_JFLOOR x the parity floor), tolerance LAZY_TOL_J */
#define LAZY_TOL_J R(5e-4)
#define LAZY_JFLOOR R(0.25)
/* ... and the kinks: beyond a crossing the model keeps extrapolating the crossing sample's share of the derivative too (its share of the sums times
 * lam = d ln(weight)/dr); an interval ends where either costs its tolerance (LAZY_TOL_KJ of max(|d out_c / d r|, LAZY_JFLOOR x the parity floor)) */
#define LAZY_TOL_KJ R(1e-3)
int oracle_lazy_nstate(void) { return LAZY_NSTATE; }

typedef struct LazySums { real S0[3], S1[3], dS0[3], dS1[3]; } LazySums;
typedef struct LazyKinks { real lo, hi; } LazyKinks;
static void lazy_kink(real x, real xp, real J, real tol_k, LazyKinks* k) {
    if (!(J * LAZY_RHO_MAX > tol_k)) return;               /* cannot cost tol_k anywhere within reach */
    if (!(fabs(xp) > R(1e-12))) return;
    real dk = -x / xp * LAZY_KINK_SAFETY, om = tol_k / J;
    if (fabs(dk) < R(2) * LAZY_H) {                        /* inside the slope stencil: the model mixes both branches */
        if (fabs(dk) + om < k->hi) k->hi = fabs(dk) + om;
        if (fabs(dk) + om < k->lo) k->lo = fabs(dk) + om;
    } else if (dk > R(0)) { if (dk + om < k->hi) k->hi = dk + om; }
    else { if (-dk + om < k->lo) k->lo = -dk + om; }
}
/* the specular sums at roughness r (all 20 samples); with `kinks` also the crossing scan, for which C0 / scale (per channel)
 * weigh a sample's contribution to the rendered value relative to the parity scale */
static void lazy_sums(const real wo[3], const real n[3], real r, const real* coef, const Rules* q, LazySums* S, LazyKinks* kinks,
                      const real C0[3], const real scale[3], const real jscale[3], real tol_k) {
    const real *u0 = q->su0, *u1 = q->su1, *w = q->sw;
    real s[3], t[3];
    oracle_frame(n, s, t);
    const real vx = dot3(s, wo), vy = dot3(t, wo), vz = dot3(n, wo);
    const real NoV = rmax(vz, R(0)), g1v = oracle_G1_GGX_Schlick(NoV, r);
    const real alpha2 = pow4r(r), kk = (r + R(1)) * (r + R(1)) / R(8);
    for (int c = 0; c < 3; ++c) S->S0[c] = S->S1[c] = S->dS0[c] = S->dS1[c] = R(0);
    for (int j = 0; j < q->ns; ++j) {
        SpecSample sp;
        real L[3];
        if (spec_sample(u0[j], u1[j], w[j], wo, n, s, t, r, g1v, NoV, &sp)) {
            oracle_sh_eval(sp.wi, coef, L);
            for (int c = 0; c < 3; ++c) {
                real x = sp.wgt * L[c];
                S->S0[c] += x; S->S1[c] += x * sp.x5;
                S->dS0[c] += x * sp.lam; S->dS1[c] += x * sp.lam * sp.x5;
            }
        }
        if (!kinks) continue;
        /* the sample's two clamped variables and their r-derivatives through theta_h(u0; r)  (:232-233) */
        const real qq = R(1) / (u0[j] * (alpha2 - R(1)) + R(1));
        const real c2 = (R(1) - u0[j]) * qq, s2 = alpha2 * u0[j] * qq;
        const real ct = sqrt(rmax(c2, R(0))), st = sqrt(rmax(s2, R(0)));
        const real g = R(4) * r * r * r * u0[j] * qq * c2;                 /* d sin^2/dr = -d cos^2/dr */
        const real stp = st > R(0) ? g / (R(2) * st) : R(0), ctp = ct > R(0) ? -g / (R(2) * ct) : R(0);
        const real phi = R(2) * O_PI * u1[j], T = cos(phi) * vx + sin(phi) * vy;
        const real d = st * T + ct * vz, dp = stp * T + ctp * vz;
        const real wlz = R(2) * d * ct - vz, wlzp = R(2) * (dp * ct + d * ctp);
        real l[3] = {st * cos(phi), st * sin(phi), ct}, wh[3], wi[3];
        to_world(s, t, n, l, wh);
        for (int i = 0; i < 3; ++i) wi[i] = R(2) * d * wh[i] - wo[i];
        oracle_sh_eval(wi, coef, L);
        const real dpos = rmax(d, R(0)), nlpos = rmax(wlz, R(0));
        const real g1l0 = R(1) / (kk + R(1e-6)), g1l = R(1) / (nlpos * (R(1) - kk) + kk + R(1e-6));
        const real x5 = pow5(R(1) - dpos);
        /* |d ln(weight)/dr| (spec_sample's lam) at n.wi = 0 and at this sample */
        const real lam_ring = R(4) / r - R(8) * r * r * r * c2 / (c2 * (alpha2 - R(1)) + R(1) + R(1e-6)) - (r + R(1)) / R(4) * g1v * (R(1) - NoV);
        const real lamk1 = fabs(lam_ring - (r + R(1)) / R(4) * g1l0), lamk2 = fabs(lam_ring - (r + R(1)) / R(4) * g1l * (R(1) - nlpos));
        const real rk = LAZY_TOL_K / LAZY_TOL_KJ;
        real J1 = R(0), J2 = R(0);
        for (int c = 0; c < 3; ++c) {
            const real F = C0[c] + (R(1) - C0[c]) * x5, aL = fabs(L[c]);
            const real K1 = w[j] * g1v * g1l0 * dpos / ct * F * aL, K2 = w[j] * g1v * g1l * nlpos / ct * aL;
            J1 = rmax(J1, K1 * rmax(R(1) / scale[c], lamk1 * rk / jscale[c]));
            J2 = rmax(J2, K2 * rmax(R(1) / scale[c], lamk2 * rk / jscale[c]));
        }
        lazy_kink(wlz, wlzp, J1 * fabs(wlzp), tol_k, kinks);
        lazy_kink(d, dp, J2 * fabs(dp), tol_k, kinks);
    }
}
static real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* round 6: an interval's two lengths are carried in eight bits each (four exponent, four mantissa bits: (16 + m) 2^(e - 25), 4.8e-7 .. 3.03e-2),
 * rounded DOWN -- an interval never widens (csrc/matpbr_lazy.hpp iv_pack / iv_unpack; lengths below 2^-21 are carried as 2^-21) */
static real iv_round_down(real x) {
    float f = (float)x;
    uint32_t u;
    if ((real)f > x) f = nextafterf(f, 0.0f);
    memcpy(&u, &f, 4);
    int q = (int)(u >> 19) - ((127 - 21) << 4);
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    u = (uint32_t)(q + ((127 - 21) << 4)) << 19;
    memcpy(&f, &u, 4);
    return (real)f;
}
/* refresh of one pixel at (clamped) roughness r: new state `st`, the exact render `out`.  The parity scale that weighs kinks and
 * the measured extrapolation error is taken from what is known BEFORE the samples are walked (so that one walk suffices): the old
 * state's prediction of the render at r, or the floor alone on a forced (first) refresh, whose intervals are <= rho_init anyway. */
static void lazy_refresh_pixel(const real wo[3], const real n[3], const real a[3], real r, real m, const real* coef, const Rules* q,
                               const real A[9], real floor_, real tol, const real* old, real* st, real out[3]) {
    real C0[3], P[3], scale[3], jscale[3], pSD[3], pS1[3];
    LazySums S, Sh;
    const real dr = old ? r - old[LZ_RREF] : R(0);
    for (int c = 0; c < 3; ++c) {
        C0[c] = (R(1) - m) * R(0.04) + m * a[c];
        P[c] = A[c] + r * A[3 + c] + r * r * A[6 + c];
        scale[c] = floor_;
        jscale[c] = LAZY_JFLOOR * floor_;
        if (old) {
            pSD[c] = old[LZ_SD + c] + old[LZ_GSD + c] * dr;
            pS1[c] = old[LZ_S1 + c] + old[LZ_GS1 + c] * dr;
            scale[c] = rmax(fabs(a[c] * (R(1) - m) * P[c] + C0[c] * pSD[c] + pS1[c]), floor_);
            const real pdSD = old[LZ_DSD + c] + old[LZ_ESD + c] * dr, pdS1 = old[LZ_DS1 + c] + old[LZ_ES1 + c] * dr;
            jscale[c] = rmax(fabs(a[c] * (R(1) - m) * (A[3 + c] + R(2) * r * A[6 + c]) + C0[c] * pdSD + pdS1), LAZY_JFLOOR * floor_);
        }
    }
    real dir = (old && dr < R(0)) ? R(-1) : R(1);
    if (r + dir * LAZY_H > R(1) || r + dir * LAZY_H < R(0.07)) dir = -dir;
    LazyKinks k = {R(1e30), R(1e30)};
    lazy_sums(wo, n, r, coef, q, &S, &k, C0, scale, jscale, tol * LAZY_TOL_K);
    lazy_sums(wo, n, r + dir * LAZY_H, coef, q, &Sh, NULL, NULL, NULL, NULL, R(0));
    for (int c = 0; c < 3; ++c) {
        st[LZ_SD + c] = S.S0[c] - S.S1[c];
        st[LZ_S1 + c] = S.S1[c];
        st[LZ_GSD + c] = ((Sh.S0[c] - Sh.S1[c]) - st[LZ_SD + c]) / (dir * LAZY_H);
        st[LZ_GS1 + c] = (Sh.S1[c] - S.S1[c]) / (dir * LAZY_H);
        st[LZ_DSD + c] = S.dS0[c] - S.dS1[c];
        st[LZ_DS1 + c] = S.dS1[c];
        st[LZ_ESD + c] = ((Sh.dS0[c] - Sh.dS1[c]) - st[LZ_DSD + c]) / (dir * LAZY_H);
        st[LZ_ES1 + c] = (Sh.dS1[c] - S.dS1[c]) / (dir * LAZY_H);
        out[c] = a[c] * (R(1) - m) * P[c] + C0[c] * st[LZ_SD + c] + st[LZ_S1 + c];
    }
    real rho = LAZY_RHO_INIT;
    if (old) {
        rho = old[LZ_RHO];
        if (fabs(dr) > LAZY_MOVED) {
            real e = R(0), ej = R(0);
            for (int c = 0; c < 3; ++c) {
                e = rmax(e, fabs(C0[c] * (pSD[c] - st[LZ_SD + c]) + (pS1[c] - st[LZ_S1 + c])) / scale[c]);
                const real pdSD = old[LZ_DSD + c] + old[LZ_ESD + c] * dr, pdS1 = old[LZ_DS1 + c] + old[LZ_ES1 + c] * dr;
                const real dP = A[3 + c] + R(2) * r * A[6 + c];
                const real jc = a[c] * (R(1) - m) * dP + C0[c] * st[LZ_DSD + c] + st[LZ_DS1 + c];
                ej = rmax(ej, fabs(C0[c] * (pdSD - st[LZ_DSD + c]) + (pdS1 - st[LZ_DS1 + c])) / rmax(fabs(jc), LAZY_JFLOOR * floor_));
            }
            const real ec = rmax(e / (tol * LAZY_TOL_S), ej / (tol * LAZY_TOL_J));
            const real want = R(0.9) * fabs(dr) / sqrt(rmax(ec, R(1e-9)));
            rho = clampr(want, R(0.5) * rho, R(2) * rho);
        }
    }
    rho = clampr(rho, LAZY_RHO_MIN, LAZY_RHO_MAX);
    st[LZ_RREF] = r; st[LZ_LO] = k.lo < rho ? k.lo : rho; st[LZ_HI] = k.hi < rho ? k.hi : rho; st[LZ_RHO] = rho;
    const real lo_cap = st[LZ_LO], hi_cap = st[LZ_HI];      /* LAZY_E_CAP below is stated on the lengths before rounding */
    st[LZ_LO] = iv_round_down(st[LZ_LO]); st[LZ_HI] = iv_round_down(st[LZ_HI]);
    /* LAZY_E_CAP: a sample that crosses the horizon inside the stencil makes the one-sided difference of a detached derivative a jump / h, not
     * a slope; whatever eSD, eS1 are, they correct the derivative by at most half its size at the far end of the interval */
    real dmax = R(0);      /* the pixel's largest |dSD_c| + |dS1_c| (round 6: per channel, a channel whose dSD passes through zero lost a legitimate slope) */
    for (int c = 0; c < 3; ++c) dmax = rmax(dmax, fabs(st[LZ_DSD + c]) + fabs(st[LZ_DS1 + c]));
    for (int c = 0; c < 3; ++c) {
        const real lim = R(0.5) * dmax / rmax(rmax(lo_cap, hi_cap), R(1e-4));
        st[LZ_ESD + c] = clampr(st[LZ_ESD + c], -lim, lim);
        st[LZ_ES1 + c] = clampr(st[LZ_ES1 + c], -lim, lim);
    }
}
/* the streaming evaluation from a state: render, jac (P, SD, d out/d r as the backward pass uses them), need = 1 when r has left the
 * validity interval (then out / jac are NOT to be used: the pixel must be refreshed) */
static int lazy_eval_pixel(const real a[3], real r, real m, const real A[9], const real* st, real out[3], real jac[9]) {
    const real dr = r - st[LZ_RREF];
    for (int c = 0; c < 3; ++c) {
        const real C0 = (R(1) - m) * R(0.04) + m * a[c];
        const real P = A[c] + r * A[3 + c] + r * r * A[6 + c], dP = A[3 + c] + R(2) * r * A[6 + c];
        const real SD = st[LZ_SD + c] + st[LZ_GSD + c] * dr, S1 = st[LZ_S1 + c] + st[LZ_GS1 + c] * dr;
        out[c] = a[c] * (R(1) - m) * P + C0 * SD + S1;
        jac[c] = P; jac[3 + c] = SD;
        jac[6 + c] = a[c] * (R(1) - m) * dP + C0 * (st[LZ_DSD + c] + st[LZ_ESD + c] * dr) + (st[LZ_DS1 + c] + st[LZ_ES1 + c] * dr);
    }
    return (dr < -st[LZ_LO] || dr > st[LZ_HI]) ? 1 : 0;
}
/* One lazy forward over N lanes / an image: evaluate from `state` (in/out, [.., LAZY_NSTATE]); lanes that left their interval --
 * or all of them when `force` is set (first iteration of a part: `state` is not read) -- are refreshed in place.  out[..,3],
 * jac[..,9] (nullable), refreshed[..] (nullable, int32 flags).  floor_[batch] = the mean-radiance floor of the parity scale. */
static void lazy_driver(const real* a, const real* r, const real* m, const real* n, const real* light, real* state, real* out, real* jac,
                        int32_t* refreshed, long P, int batch, int spp, const View* v, const real* floor_, real tol, int force) {
    Rules q;
    rules_init(&q, spp);
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < P * batch; ++idx) {
        long b = idx / P, p = idx % P;
        real wo[3], nh[3], A[9], jj[9], st_new[LAZY_NSTATE];
        view_of(v, v->wo ? idx : p, wo);
        unit_normal(n + idx * 3, nh);
        const real* coef = light + b * MATPBR_NSH * 3;
        diffuse_coef(wo, nh, coef, &q, A);
        real* st = state + idx * LAZY_NSTATE;
        const real rc = clampr(r[idx], R(0.07), R(1));
        int need = force ? 1 : lazy_eval_pixel(a + idx * 3, rc, m[idx], A, st, out + idx * 3, jj);
        if (need) {
            lazy_refresh_pixel(wo, nh, a + idx * 3, rc, m[idx], coef, &q, A, floor_[b], tol, force ? NULL : st, st_new, out + idx * 3);
            memcpy(st, st_new, sizeof(st_new));
            lazy_eval_pixel(a + idx * 3, rc, m[idx], A, st, out + idx * 3, jj);
        }
        if (jac) memcpy(jac + idx * 9, jj, sizeof(jj));
        if (refreshed) refreshed[idx] = need;
    }
}
void oracle_lazy_fwd(const real* a, const real* r, const real* m, const real* n, const real* light, real* state, real* out, real* jac,
                     int32_t* refreshed, int H, int W, int batch, int spp, real fov_x_deg, const real* floor_, real tol, int force) {
    View v = {H, W, 0, 0, W, fov_x_deg, NULL};
    lazy_driver(a, r, m, n, light, state, out, jac, refreshed, (long)H * W, batch, spp, &v, floor_, tol, force);
}
void oracle_lazy_fwd_lanes(const real* a, const real* r, const real* m, const real* n, const real* wo, const real* light, real* state,
                           real* out, real* jac, int32_t* refreshed, long N, int spp, real floor_, real tol, int force) {
    View v = {0, 0, 0, 0, 1, R(0), wo};
    lazy_driver(a, r, m, n, light, state, out, jac, refreshed, N, 1, spp, &v, &floor_, tol, force);
}

/* Per-pixel radiance transfer of the production estimator: R[c] = sum_k light[k][c] T[k][c]; T[B,H,W,25,3]. */
void oracle_shade_transfer(const real* a, const real* r, const real* m, const real* n, real* T, int H, int W, int batch, int spp,
                           real fov_x_deg) {
    const long P = (long)H * W;
    real zero[MATPBR_NSH * 3];
    memset(zero, 0, sizeof(zero));
    Rules q;
    rules_init(&q, spp);
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < P * batch; ++idx) {
        long p = idx % P;
        real wo[3], nh[3], ga[3], gr, gm, gn[3];
        oracle_view_dir((int)(p / W), (int)(p % W), H, W, fov_x_deg, wo);
        unit_normal(n + idx * 3, nh);
        real* Tp = T + idx * MATPBR_NSH * 3;
        memset(Tp, 0, sizeof(real) * MATPBR_NSH * 3);
        const real one[3] = {R(1), R(1), R(1)};
        split_pixel_bwd(wo, nh, a + idx * 3, r[idx], m[idx], zero, &q, one, ga, &gr, &gm, gn, Tp);
    }
}

/* a9 (scene prep): per-pixel geometric normal of the depth heightfield.  The reference shades with the
 * face normal of the triangulated depth mesh (inverse_img_w_mi.py:751-758, myutils/mesh_recon.py:41-74);
 * the per-pixel restatement is the normalised cross product of central differences of the back-projected
 * positions (one-sided at the border), oriented towards the camera. */
void oracle_normals_from_depth(const real* depth, real* out_n, int H, int W, int batch, real fov_x_deg) {
    const long P = (long)H * W;
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < P * batch; ++idx) {
        long b = idx / P, p = idx % P;
        int i = (int)(p / W), j = (int)(p % W);
        const real* d = depth + b * P;
        int j0 = j > 0 ? j - 1 : j, j1 = j < W - 1 ? j + 1 : j;
        int i0 = i > 0 ? i - 1 : i, i1 = i < H - 1 ? i + 1 : i;
        real pl[3], pr[3], pu[3], pd[3], c[3];
        oracle_pixel_to_world(i, j0, d[(long)i * W + j0], H, W, fov_x_deg, pl);
        oracle_pixel_to_world(i, j1, d[(long)i * W + j1], H, W, fov_x_deg, pr);
        oracle_pixel_to_world(i0, j, d[(long)i0 * W + j], H, W, fov_x_deg, pu);
        oracle_pixel_to_world(i1, j, d[(long)i1 * W + j], H, W, fov_x_deg, pd);
        real dx[3] = {pr[0] - pl[0], pr[1] - pl[1], pr[2] - pl[2]};
        real dy[3] = {pd[0] - pu[0], pd[1] - pu[1], pd[2] - pu[2]};
        /* +x is image-right, image-down is -y: dx cross (-dy) ... orient afterwards */
        real nn[3] = {dx[1] * dy[2] - dx[2] * dy[1], dx[2] * dy[0] - dx[0] * dy[2], dx[0] * dy[1] - dx[1] * dy[0]};
        real l = sqrt(dot3(nn, nn));
        oracle_pixel_to_world(i, j, d[(long)i * W + j], H, W, fov_x_deg, c);
        real s = (dot3(nn, c) > R(0)) ? -R(1) : R(1); /* face the camera at the origin */
        if (l > R(0)) for (int k = 0; k < 3; ++k) out_n[idx * 3 + k] = s * nn[k] / l;
        else { out_n[idx * 3] = R(0); out_n[idx * 3 + 1] = R(0); out_n[idx * 3 + 2] = R(1); }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Batch wrappers for the tests (N lanes, AoS [N,3] vectors).
 * ---------------------------------------------------------------------------------------------- */
void oracle_eval_brdf_batch(long N, const real* wi, const real* wo, const real* n, const real* a, const real* r,
                            const real* m, real* f, real* pdf) {
    for (long k = 0; k < N; ++k) oracle_eval_brdf(wi + 3 * k, wo + 3 * k, n + 3 * k, a + 3 * k, r[k], m[k], f + 3 * k, pdf + k);
}
void oracle_eval_brdf_grad_batch(long N, const real* wi, const real* wo, const real* n, const real* a, const real* r,
                                 const real* m, const real* g, real* d_a, real* d_r, real* d_m, real* d_n) {
    for (long k = 0; k < N; ++k)
        oracle_eval_brdf_grad(wi + 3 * k, wo + 3 * k, n + 3 * k, a + 3 * k, r[k], m[k], g + 3 * k, d_a + 3 * k, d_r + k, d_m + k, d_n + 3 * k);
}
void oracle_sample_brdf_batch(long N, const real* sample1, const real* sample2 /*[N,2]*/, const real* wo, const real* n,
                              const real* a, const real* r, const real* m, real* wi, real* pdf, real* weight) {
    for (long k = 0; k < N; ++k)
        oracle_sample_brdf(sample1[k], sample2[2 * k], sample2[2 * k + 1], wo + 3 * k, n + 3 * k, a + 3 * k, r[k], m[k],
                           wi + 3 * k, pdf + k, weight + 3 * k);
}
void oracle_sh_basis_batch(long N, const real* theta, const real* phi, real* Y /*[N,25]*/) {
    for (long k = 0; k < N; ++k) oracle_sh_basis_angles(theta[k], phi[k], Y + MATPBR_NSH * k);
}
void oracle_sh_basis_dir_batch(long N, const real* w, real* Y) {
    for (long k = 0; k < N; ++k) oracle_sh_basis_dir(w + 3 * k, Y + MATPBR_NSH * k);
}
void oracle_sample_table(int spp, real* u /*[spp/2,2]*/) {
    for (int i = 0; i < spp / 2; ++i) oracle_sample_point(spp, i, u + 2 * i, u + 2 * i + 1);
}
int oracle_rule_max(void) { return ORACLE_MAX_RULE; }
int oracle_sizeof_real(void) { return (int)sizeof(real); }
