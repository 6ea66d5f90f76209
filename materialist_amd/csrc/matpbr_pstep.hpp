// matpbr_pstep.hpp -- the step of hot loop B (`--model_name none`, inverse_img_w_mi.py:371-432) on FOLDED per-pixel models, as a persistent
// streaming kernel (gfx950, wave64, fp32).
//
// lazy_step_kernel (matpbr_lazy.hpp) reads, per pixel and iteration, the three maps, the 80-byte generic model, the target, the anchors and the
// Adam moments: 172.6 B/pixel in an 'rm' part against the 108 B/pixel of the canonical forward + backward pair.  Two things are wrong with that:
//   * bytes: a part moves SOME of the maps (--opt_order 'rm a', :343-357); what it leaves alone is a constant that folds into the model
//     (kFoldXY: the albedo, 64 B/pixel of model and no albedo read; kFoldGH: roughness and metallic, 24 B/pixel of model and neither map read).
//     'rm': r 4 + m 4 read, 8 written, model 68 (X0 / Y0 in 24 bits each, half-precision slopes of the detached derivatives: round 6), target 12, anchors 8,
//           Adam moments 32 = 136 B/pixel (was 172; 148 while the next render was stored);
//     'a' : a 12 read, 12 written, model 24, target 12, anchors 12, Adam moments 48 = 120 B/pixel (was 160);
//   * shape: 4096 workgroups of 512 pixels each fold the iteration's statistics before their first load.  Here a workgroup takes up to four
//     512-pixel blocks (at most 1024 workgroups: all resident at four per CU), requests its first two tiles BEFORE it folds the statistics, and
//     streams tile after tile from two statically named register sets (the loads of the tile after next are in flight while a tile is computed).
//     The few pixels that leave their model's interval go to the image's walk queue (one atomic per wave with entries) and are re-sampled by
//     lazy_pwalk_kernel, a launch of one wave per eight entries behind this one.
//   * launches: the loss statistics of iteration t + 1 need nothing but the render of iteration t + 1, which is formed HERE: the kernel leaves per block
//     the sums from which the next launch's heads have them (loss_acc, matpbr_lazy.hpp; two sets of records by iteration parity) -- no statistics
//     launch after the first iteration of a part, and no stored render (matpbr_brdf_phase_resolve / fold_resolve_kernel form what the caller reads).
// Thread t of a workgroup owns the pixels t and t + 256 of each of its blocks, as in lazy_step_kernel: per-block sums (render, regularisers)
// are formed in the same order whatever the batch size and the number of blocks per workgroup (batch = stand-alone, bit for bit).
// The arithmetic of a pixel is that of lazy_step_pixel with the folded expressions (matpbr_lazy.hpp, "folded models"); parts that move the
// albedo together with another map, and callers that ask for a gradient the folded form does not have, stay on lazy_step_kernel.
#pragma once
#include "matpbr_lazy.hpp"

namespace matpbr {

__device__ __forceinline__ void stu(void* base, unsigned off, uint32_t v) { *(uint32_t*)((char*)base + off) = v; }

// X2_c from the words that hold it: `lohi` (kFxLoHi: interval bytes | half X2_0) and `q` (kFxQ: half2 (X2_1, X2_2))
__device__ __forceinline__ float xy_x2(uint32_t lohi, uint32_t q, int c) { return c == 0 ? h2_hi(lohi) : (c == 1 ? h2_lo(q) : h2_hi(q)); }
// X_c(dr), Y_c(dr) of a kFoldXY model from its stored words (s = half2 (X1, Y1))
__device__ __forceinline__ void xy_eval(float X0, float Y0, uint32_t s, float x2, float dr, float& X, float& Y) {
    X = fmaf(fmaf(x2, dr, h2_lo(s)), dr, X0);
    Y = fmaf(fmaf(-x2, dr, h2_hi(s)), dr, Y0);
}

// =================================================================================================
// generic planes -> folded planes, at the start of a part (and the render of the part's first iteration in the folded expression)
// =================================================================================================
struct LazyFoldArgs {
    uint32_t* walk_cnt;               // [B][2][kWalkShards] the walk queue's counters (cleared here, at the start of a part)
    const float *a, *r, *m;           // the part's start parameters (raw: clamped here as every render clamps them)
    const uint32_t* plane[kLzPlanes];
    uint32_t* fplane[kFxPlanes];
    float* out;                       // [B,H,W,3] the render of these parameters
    float* block_sums;                // [B][nblk]: sum of the rgb of each 512-pixel block
    const float* stats;               // nullable: skip images whose EarlyStopping has fired
};
template <int MODE>
__global__ __launch_bounds__(kBlock) void lazy_fold_kernel(const LazyFoldArgs q, int P) {
    __shared__ float s_sum[4];
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x < 2 * kWalkShards) q.walk_cnt[b * 2 * kWalkShards + threadIdx.x] = 0u;
    if (q.stats && img_stopped(q.stats, b)) return;
    float tot = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = blockIdx.x * kLazyBlockPixels + h * kBlock + (int)threadIdx.x;
        if (p >= P) continue;
        const unsigned i = (unsigned)(b * P + p), o1 = i * 4u, o3 = i * 12u, of = fx_off(i);
        const F3 av = ld3(q.a, o3);
        const float a[3] = {fminf(fmaxf(av.x, 0.0f), 1.0f), fminf(fmaxf(av.y, 0.0f), 1.0f), fminf(fmaxf(av.z, 0.0f), 1.0f)};
        const float r = fminf(fmaxf(ldf(q.r, o1), 0.07f), 1.0f), m = fminf(fmaxf(ldf(q.m, o1), 0.0f), 1.0f), omm = 1.0f - m;
        const float rref = as_f(ldu(q.plane[kLzRref], o1));
        const float dr = r - rref;
        float rgb[3], x2h[3], jx1[3] = {0.0f, 0.0f, 0.0f}, jy1[3] = {0.0f, 0.0f, 0.0f}, fx0[3] = {0.0f, 0.0f, 0.0f}, fy0[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float Pv = as_f(ldu(q.plane[kLzP + c], o1)), SDv = as_f(ldu(q.plane[kLzSD + c], o1)), S1v = as_f(ldu(q.plane[kLzS1 + c], o1));
            const uint32_t pk = ldu(q.plane[kLzPk + c], o1), sk = ldu(q.plane[kLzSk + c], o1), dk = ldu(q.plane[kLzDk + c], o1);
            if (MODE == kFoldXY) {
                FoldXY f;
                const uint32_t ek = ldu(q.plane[kLzEk + c], o1);
                fold_xy(a[c], Pv, SDv, S1v, h2_lo(pk), h2_hi(pk), h2_lo(sk), h2_hi(sk), h2_lo(dk), h2_hi(dk), h2_lo(ek), h2_hi(ek), mref_code(m), f);
                const uint32_t s = pack_h2(f.X1, f.Y1);
                x2h[c] = f.X2; jx1[c] = f.JX1; jy1[c] = f.JY1; fx0[c] = f.X0; fy0[c] = f.Y0;
                stu(q.fplane[kFxS + c], of, s);
                stu(q.fplane[kFxJ + c], of, pack_h2(f.JX0, f.JY0));
                float X, Y;
                xy_eval(f.X0, f.Y0, s, (float)(_Float16)f.X2, dr, X, Y);      // the stored (half) words, as the step kernel reads them
                rgb[c] = fmaf(m, Y, X);
            } else {
                const float Pc = fmaf(fmaf(h2_hi(pk), dr, h2_lo(pk)), dr, Pv);
                const float SD = fmaf(h2_lo(sk), dr, SDv), S1 = fmaf(h2_hi(sk), dr, S1v);
                const float G = fmaf(m, SD, omm * Pc), Hc = fmaf(omm * 0.04f, SD, S1);
                stu(q.fplane[kFgG + c], of, as_u(G));
                stu(q.fplane[kFgH + c], of, as_u(Hc));
                rgb[c] = fmaf(a[c], G, Hc);
            }
            tot += rgb[c];
        }
        if (MODE == kFoldXY) {
            uint32_t xw[5];
            xy_pack(fx0, fy0, mref_code(m), xw);
#pragma unroll
            for (int k = 0; k < 5; ++k) stu(q.fplane[kFxXY + k], of, xw[k]);
            stu(q.fplane[kFxRref], of, as_u(rref));
            const uint32_t lh = ldu(q.plane[kLzLoHi], o1);
            stu(q.fplane[kFxLoHi], of, pack_lohi_x2(h2_lo(lh), h2_hi(lh), x2h[0]));
            stu(q.fplane[kFxQ], of, pack_h2(x2h[1], x2h[2]));
#pragma unroll
            for (int c = 0; c < 3; ++c) stu(q.fplane[kFxE + c], of, pack_h2(jx1[c], jy1[c]));
        }
        st3(q.out, o3, rgb[0], rgb[1], rgb[2]);
    }
    tot = wave_sum_to_lane63(tot);
    if ((threadIdx.x & 63) == 63) s_sum[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) q.block_sums[(long)b * gridDim.x + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}

// =================================================================================================
// the persistent step
// =================================================================================================
// what one pixel of a tile loads (one statically named set per tile parity: no register copies between the request and the use)
struct PxXY {
    float r, m, rref;
    uint32_t lohi;
    uint32_t xy[5];                   // X0_c, Y0_c and m_ref as the planes carry them (xy_pack)
    uint32_t s[3], j[3], e[3], q;
    F3 gt;
    float r0, m0, mr, vr, mm, vm;
};
struct PxGH {
    F3 a, gt, a0, ma, va;
    float G[3], H[3];
};
struct PStepFlags { bool part_r, part_m, slopes, att; };

// the parameters: the only loads whose address depends on the image's state row (MATPBR_FLAG_ROTATE_BEST: which buffer holds the current values)
__device__ __forceinline__ void pstep_load_params(PxXY& x, const StepPtrs& sp, unsigned i) {
    x.r = ldf(sp.r, i * 4u); x.m = ldf(sp.m, i * 4u);
}
__device__ __forceinline__ void pstep_load_fixed(PxXY& x, const LazyStepArgs& qs, unsigned i, const PStepFlags f) {
    const JacBwdArgs& q = qs.j;
    const unsigned o1 = i * 4u, o3 = i * 12u, of = fx_off(i);
    x.gt = ld3(q.gt_srgb, o3);
#pragma unroll
    for (int k = 0; k < 5; ++k) x.xy[k] = ldu(qs.fplane[kFxXY + k], of);
    x.rref = 0.0f; x.lohi = 0u; x.q = 0u;
#pragma unroll
    for (int c = 0; c < 3; ++c) x.s[c] = x.j[c] = x.e[c] = 0u;
    if (f.slopes) {           // a part that leaves the roughness alone never moves away from r_ref (uniform branch)
        x.rref = as_f(ldu(qs.fplane[kFxRref], of)); x.lohi = ldu(qs.fplane[kFxLoHi], of);
        x.q = ldu(qs.fplane[kFxQ], of);
#pragma unroll
        for (int c = 0; c < 3; ++c) x.s[c] = ldu(qs.fplane[kFxS + c], of);
        if (f.att) {
#pragma unroll
            for (int c = 0; c < 3; ++c) x.j[c] = x.s[c];       // the models' own slopes are the derivative: no first-order correction on top (e = 0)
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) { x.j[c] = ldu(qs.fplane[kFxJ + c], of); x.e[c] = ldu(qs.fplane[kFxE + c], of); }
        }
    }
    x.r0 = x.m0 = x.mr = x.vr = x.mm = x.vm = 0.0f;
    if (f.part_r) { x.r0 = ldf(q.r0, o1); if (q.am[1]) { x.mr = ldf(q.am[1], o1); x.vr = ldf(q.av[1], o1); } }
    if (f.part_m) { x.m0 = ldf(q.m0, o1); if (q.am[2]) { x.mm = ldf(q.am[2], o1); x.vm = ldf(q.av[2], o1); } }
}
__device__ __forceinline__ void pstep_load(PxXY& x, const LazyStepArgs& qs, const StepPtrs& sp, unsigned i, const PStepFlags f) {
    pstep_load_params(x, sp, i);
    pstep_load_fixed(x, qs, i, f);
}
__device__ __forceinline__ void pstep_load_params(PxGH& x, const StepPtrs& sp, unsigned i) { x.a = ld3(sp.a, i * 12u); }
__device__ __forceinline__ void pstep_load_fixed(PxGH& x, const LazyStepArgs& qs, unsigned i, const PStepFlags) {
    const JacBwdArgs& q = qs.j;
    const unsigned o3 = i * 12u, of = fx_off(i);
    x.gt = ld3(q.gt_srgb, o3);
    x.a0 = ld3(q.a0, o3);
#pragma unroll
    for (int c = 0; c < 3; ++c) { x.G[c] = as_f(ldu(qs.fplane[kFgG + c], of)); x.H[c] = as_f(ldu(qs.fplane[kFgH + c], of)); }
    x.ma = F3{0.0f, 0.0f, 0.0f}; x.va = F3{0.0f, 0.0f, 0.0f};
    if (q.am[0]) { x.ma = ld3(q.am[0], o3); x.va = ld3(q.av[0], o3); }
}
__device__ __forceinline__ void pstep_load(PxGH& x, const LazyStepArgs& qs, const StepPtrs& sp, unsigned i, const PStepFlags f) {
    pstep_load_params(x, sp, i);
    pstep_load_fixed(x, qs, i, f);
}

// torch.optim.Adam on one element (adam_update of matpbr_shade.hpp with the moments in registers)
__device__ __forceinline__ float adam_apply(float p, float gi, float m_old, float v_old, const JacBwdArgs& q, float& mi, float& vi) {
    mi = fmaf(q.b1, m_old, (1.0f - q.b1) * gi);
    vi = fmaf(q.b2, v_old, (1.0f - q.b2) * gi * gi);
    return p - q.lr_over_bc1 * mi / fmaf(fsqrt(vi), q.inv_sqrt_bc2, q.eps);
}
// d loss / d pred of 3 (l1/mse) mse + l1 on xs = max(pred ratio, eps)^(1/2.2)  (:388-418), as lazy_step_pixel forms it
__device__ __forceinline__ float loss_go(float prc, float gt, float ratio, float sr, float inv_n3, float& xs) {
    const float x = prc * ratio;
    const float xc = fmaxf(x, kLossEps);
    xs = pow_inv_gamma(xc);
    const float d = xs - gt;
    const float dxs = x > kLossEps ? xs * rcp(xc) * (1.0f / 2.2f) : 0.0f;
    return ratio * dxs * fmaf(6.0f * sr, d, fsign(d)) * inv_n3;
}

// one pixel of a kFoldXY part; returns whether its new roughness has left its model's interval
__device__ __forceinline__ bool pstep_pixel(const PxXY& x, const LazyStepArgs& qs, const StepPtrs& sp, unsigned i, const PStepFlags f, float ratio,
                                            float sr, bool improved, float& tot, float (&reg)[3], float (&acc)[5]) {
    const JacBwdArgs& q = qs.j;
    const unsigned o1 = i * 4u, o3 = i * 12u;
    if (qs.rotate) improved = false;      // no snapshot stores: the buffer just read IS the snapshot
    const float r = fminf(fmaxf(x.r, 0.07f), 1.0f), m = fminf(fmaxf(x.m, 0.0f), 1.0f);
    const float dr = r - x.rref;
    const float gt[3] = {x.gt.x, x.gt.y, x.gt.z};
    // (m_ref rides in xy_pack's fourth word: kFxJ holds JA0 = JX0 + m_ref JY0; with the models' own slopes as the derivative there is none)
    float X0[3], Y0[3];
    xy_unpack(x.xy, X0, Y0);
    const float dmr = m - (f.att ? 0.0f : mref_of(x.xy[3]));
    float drr = 0.0f, dm = 0.0f, xs_keep[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // the render of iteration t is not read back: it IS the model at the current parameters (the expression that wrote pred -- the step
        // before, the walk at dr = 0, or lazy_fold_kernel -- on the same operands: the same bits)
        const float x2 = xy_x2(x.lohi, x.q, c);
        float X, Y;
        xy_eval(X0[c], Y0[c], x.s[c], x2, dr, X, Y);
        const float go = loss_go(fmaf(m, Y, X), gt[c], ratio, sr, q.inv_n3, xs_keep[c]);
        // d out / d r to first order in dr: the curvature of the diffuse lobe (2 X2, exact) and the slopes of the detached specular derivatives
        const float jx1 = h2_lo(x.e[c]), jy1 = h2_hi(x.e[c]);
        // = JX + m JY with JX = JX0 + (2 X2 + JX1) dr, JY = JY0 + (JY1 - 2 X2) dr, JX0 = JA0 - m_ref JY0
        const float J0 = fmaf(dmr, h2_hi(x.j[c]), h2_lo(x.j[c])), J1 = fmaf(m, fmaf(-2.0f, x2, jy1), fmaf(2.0f, x2, jx1));
        drr = fmaf(go, fmaf(J1, dr, J0), drr);
        dm = fmaf(go, Y, dm);
    }
    if (improved && q.best_img) st3(q.best_img, o3, xs_keep[0], xs_keep[1], xs_keep[2]);
    float gr = drr + (f.part_r ? q.scale_delta * q.inv_n1 * fsign(r - x.r0) : 0.0f);
    float gm = dm + (f.part_m ? q.scale_delta * q.inv_n1 * fsign(m - x.m0) : 0.0f);
    gr = (x.r >= 0.07f && x.r <= 1.0f) ? gr : 0.0f;
    gm = (x.m >= 0.0f && x.m <= 1.0f) ? gm : 0.0f;
    if (q.d_r) stf(q.d_r, o1, gr);
    if (q.d_m) stf(q.d_m, o1, gm);
    if (improved && q.best_r) stf(q.best_r, o1, r);
    if (improved && q.best_m) stf(q.best_m, o1, m);
    float nr = x.r, nm = x.m;
    if (f.part_r && q.am[1]) {
        float mi, vi;
        nr = adam_apply(x.r, gr, x.mr, x.vr, q, mi, vi);
        stf(q.am[1], o1, mi); stf(q.av[1], o1, vi); stf(sp.pr, o1, nr);
    }
    if (f.part_m && q.am[2]) {
        float mi, vi;
        nm = adam_apply(x.m, gm, x.mm, x.vm, q, mi, vi);
        stf(q.am[2], o1, mi); stf(q.av[2], o1, vi); stf(sp.pm, o1, nm);
    }
    // ---- forward of iteration t+1 from the same model
    const float r1 = fminf(fmaxf(nr, 0.07f), 1.0f), m1 = fminf(fmaxf(nm, 0.0f), 1.0f), dr1 = r1 - x.rref;
    if (f.part_r) reg[1] += fabsf(r1 - x.r0);
    if (f.part_m) reg[2] += fabsf(m1 - x.m0);
    // (both codes 255: a constant model -- a pixel without geometry, ops.background_into_lazy_state -- that no roughness leaves; a walked interval is
    // at most kLzRhoMax, code 254)
    const bool need = f.slopes && !(dr1 >= -iv_unpack(x.lohi & 0xffu) && dr1 <= iv_unpack((x.lohi >> 8) & 0xffu)) && (x.lohi & 0xffffu) != 0xffffu;
    if (!need) {
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float X, Y;
            xy_eval(X0[c], Y0[c], x.s[c], xy_x2(x.lohi, x.q, c), dr1, X, Y);
            rgb[c] = fmaf(m1, Y, X);
            tot += rgb[c];
            loss_acc(rgb[c], gt[c], ratio, acc);               // its share of the next iteration's statistics
        }
        if (!qs.no_pred) st3(sp.pred_next, o3, rgb[0], rgb[1], rgb[2]);
    }
    return need;
}
// one pixel of a kFoldGH part (never leaves its model: nothing of it depends on the roughness)
__device__ __forceinline__ bool pstep_pixel(const PxGH& x, const LazyStepArgs& qs, const StepPtrs& sp, unsigned i, const PStepFlags, float ratio,
                                            float sr, bool improved, float& tot, float (&reg)[3], float (&acc)[5]) {
    const JacBwdArgs& q = qs.j;
    const unsigned o3 = i * 12u;
    if (qs.rotate) improved = false;
    const float ra[3] = {x.a.x, x.a.y, x.a.z}, gt[3] = {x.gt.x, x.gt.y, x.gt.z}, a0[3] = {x.a0.x, x.a0.y, x.a0.z};
    const float ma[3] = {x.ma.x, x.ma.y, x.ma.z}, va[3] = {x.va.x, x.va.y, x.va.z};
    float na[3], mi[3], vi[3], xs_keep[3], gs[3], ac[3], rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ac[c] = fminf(fmaxf(ra[c], 0.0f), 1.0f);
        const float go = loss_go(fmaf(ac[c], x.G[c], x.H[c]), gt[c], ratio, sr, q.inv_n3, xs_keep[c]);
        float gsum = go * x.G[c] + q.scale_delta * q.inv_n3 * fsign(ac[c] - a0[c]);                                                  // :398,418
        gsum = (ra[c] >= 0.0f && ra[c] <= 1.0f) ? gsum : 0.0f;                                                                         // clamp backward
        gs[c] = gsum;
        na[c] = ra[c]; mi[c] = ma[c]; vi[c] = va[c];
        if (q.am[0]) na[c] = adam_apply(ra[c], gsum, ma[c], va[c], q, mi[c], vi[c]);
        const float a1 = fminf(fmaxf(na[c], 0.0f), 1.0f);
        reg[0] += fabsf(a1 - a0[c]);
        rgb[c] = fmaf(a1, x.G[c], x.H[c]);
        tot += rgb[c];
        loss_acc(rgb[c], gt[c], ratio, acc);
    }
    if (q.d_a) st3(q.d_a, o3, gs[0], gs[1], gs[2]);
    if (improved && q.best_a) st3(q.best_a, o3, ac[0], ac[1], ac[2]);
    if (improved && q.best_img) st3(q.best_img, o3, xs_keep[0], xs_keep[1], xs_keep[2]);
    if (q.am[0]) {
        st3(q.am[0], o3, mi[0], mi[1], mi[2]);
        st3(q.av[0], o3, vi[0], vi[1], vi[2]);
        st3(sp.pa, o3, na[0], na[1], na[2]);
    }
    if (!qs.no_pred) st3(sp.pred_next, o3, rgb[0], rgb[1], rgb[2]);
    return false;
}

template <int MODE> struct PStepPx;
template <> struct PStepPx<kFoldXY> { typedef PxXY type; };
template <> struct PStepPx<kFoldGH> { typedef PxGH type; };

#define PS_NOTE(k, v) do { } while (0)
constexpr int kPstepListCap = kMaxTilesPerWg * 64;     // entries of a wave's list: every pixel it owns in the workgroup's tiles
constexpr int kPstepMaxBlocks = kMaxTilesPerWg / 2;

template <int MODE>
__global__ __launch_bounds__(kBlock, 4) void lazy_pstep_kernel(const LazyStepArgs qs, const float* __restrict__ light, const Geom g, const RuleTable tab) {
    typedef typename PStepPx<MODE>::type Px;
    __shared__ float s_state[kStateStride];
    __shared__ float s_fold[4][6];
    __shared__ float s_bsum[kPstepMaxBlocks][4];
    __shared__ float s_breg[kPstepMaxBlocks][4][3];
    __shared__ float s_bacc[kPstepMaxBlocks][4][5];
    __shared__ float s_fold9[4][9];
    __shared__ long long s_wk[6];
    __shared__ uint16_t s_wcnt[kPstepMaxBlocks][4];            // listed pixels per (block, wave)
    __shared__ uint16_t s_list[MODE == kFoldXY ? 4 : 1][MODE == kFoldXY ? kPstepListCap : 1];
    const JacBwdArgs& q = qs.j;
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const int nblk_img = lazy_fwd_blocks(P);
    PS_NOTE(12, __builtin_amdgcn_s_memrealtime());
    // this workgroup's blocks: blockIdx.x, blockIdx.x + gridDim.x, ... (>= 1 by the launch geometry).  Interleaved, not consecutive: the
    // workgroups that run side by side stream neighbouring blocks, and the pixels that leave their intervals together (a neighbourhood of
    // the image) are spread over many workgroups' lists
    const int nb = (nblk_img - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int ntile = 2 * nb;
    auto tile_px0 = [&](int t) -> int { return ((int)blockIdx.x + (t >> 1) * (int)gridDim.x) * kLazyBlockPixels + (t & 1) * kTile; };
    // ---- the first two tiles are requested before anything else: their latency runs under the fold of the statistics.  Models, target,
    // anchors and moments first -- nothing about their addresses depends on the image's state row, whose load is one more round trip at the
    // head of every workgroup (and the whole of a one-image launch is a handful of round trips)
    auto pix = [&](int t) -> int { const int p = tile_px0(t) + (int)threadIdx.x; return p < P ? p : P - 1; };
    PStepFlags f;
    f.part_r = (q.part_mask & MATPBR_PART_R) != 0;
    f.part_m = (q.part_mask & MATPBR_PART_M) != 0;
    f.slopes = f.part_r || q.d_r != nullptr;
    f.att = qs.attached != 0;
    const float* old = qs.state_old + b * kStateStride;
    const float stopped_old = old[kStStopped], sel_old = old[kStSel];     // (scalar loads, asked for first: they arrive while the ~90 vector loads below issue)
    Px A, B;
    pstep_load_fixed(A, qs, (unsigned)(b * P + pix(0)), f);
    pstep_load_fixed(B, qs, (unsigned)(b * P + pix(1)), f);
    if (stopped_old > 0.5f) {                              // EarlyStopping fired in an earlier iteration (uniform): nothing to do
        if (blockIdx.x == 0 && threadIdx.x < kStateStride) {
            float v = old[threadIdx.x];
            if (threadIdx.x == kStStopped) v = 2.0f;
            if (threadIdx.x == kStImproved) v = 0.0f;
            qs.state_new[b * kStateStride + threadIdx.x] = v;
            if (threadIdx.x < kStatsStride) qs.stats_out[b * kStatsStride + threadIdx.x] = v;
        }
        return;
    }
    // where this image's iteration reads its parameters and writes the new ones: MATPBR_FLAG_ROTATE_BEST keeps them in two buffers each and
    // the OLD state row says which holds the current values (the new selector is known after the commit below: the writes wait for it)
    StepPtrs sp{q.a, q.r, q.m, q.pa, q.pr, q.pm, qs.pred_next};
    if (qs.rotate) {
        const bool rd1 = __builtin_amdgcn_readfirstlane((int)(sel_old > 0.5f)) != 0;
        if (qs.alt_a) sp.a = rd1 ? qs.alt_a : q.pa;
        if (qs.alt_r) sp.r = rd1 ? qs.alt_r : q.pr;
        if (qs.alt_m) sp.m = rd1 ? qs.alt_m : q.pm;
    }
    pstep_load_params(A, sp, (unsigned)(b * P + pix(0)));
    pstep_load_params(B, sp, (unsigned)(b * P + pix(1)));
    __builtin_amdgcn_sched_barrier(0);
    // ---- the iteration's statistics: every workgroup folds the rows of partial sums of its image (fixed order: the same bits everywhere),
    // forms the scalars from the OLD SaveBest / EarlyStopping state; workgroup 0 of the image writes the NEW state and the caller's row
    float ratio, sr;
    bool improved;
    {
        const float* rows = qs.fold_part + (long)b * step_part_stride(qs.fold_rows);
        float v[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        float w9[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (qs.acc_mode) {
            // round 6: what the step before left per block for THIS iteration -- S, (A, Bq, Cq, L, Mq), the three regulariser sums -- in block order
            // (thread i: blocks i, i + 256, ...; DPP tree; the four waves in order: the same bits in every workgroup, whatever the batch size)
            for (int i = threadIdx.x; i < nblk_img; i += kBlock) {
                const float* rec = qs.rec_in + ((long)b * nblk_img + i) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) w9[k] += rec[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float w = wave_sum_to_lane63(w9[k]);
                if ((threadIdx.x & 63) == 63) s_fold9[threadIdx.x >> 6][k] = w;
            }
        } else {
            for (int i = threadIdx.x; i < qs.fold_rows; i += kBlock) {
#pragma unroll
                for (int k = 0; k < 5; ++k) v[k] += rows[(long)i * 5 + k];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float w = wave_sum_to_lane63(v[k]);
                if ((threadIdx.x & 63) == 63) s_fold[threadIdx.x >> 6][k] = w;
            }
        }
        float st[kStatsStride];
        float sp_total = qs.acc_mode ? 0.0f : rows[(long)qs.fold_rows * 5];
        float bratio = -1.0f;
        long long wk[6] = {0, 0, 0, 0, 0, 0};
        if (qs.acc_mode && qs.walk_acc != nullptr && threadIdx.x >= 64 && threadIdx.x < 70) {
            // the walked pixels' shares (the walk of the iteration before: the other parity), one thread per number over the shards: integer sums
            const long long* wa = qs.walk_acc + ((long)b * 2 + (qs.walk_par ^ 1)) * kWalkShards * 6 + (threadIdx.x - 64);
            long long t6 = 0;
#pragma unroll 8
            for (int sh = 0; sh < kWalkShards; ++sh) t6 += wa[sh * 6];
            s_wk[threadIdx.x - 64] = t6;
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < kStatsStride; ++i) st[i] = old[i];
            bratio = old[kStBestRatio];
        }
        __syncthreads();
        if (threadIdx.x == 0 && qs.acc_mode && qs.walk_acc != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; ++k) wk[k] = s_wk[k];
        }
        if (threadIdx.x == 0) {
            float t[5];
            if (qs.acc_mode) {
                float u[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) u[k] = (s_fold9[0][k] + s_fold9[1][k]) + (s_fold9[2][k] + s_fold9[3][k]);
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] += (float)((double)wk[k] * (1.0 / kWalkFix));
                sp_total = u[0];
                const float ratio_new = st[kStGtSum] / sp_total, ratio_old = st[kStRatio];
                const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(ratio_new / ratio_old) * (1.0f / 2.2f)) - 1.0f;
                t[0] = fmaf(e, fmaf(e, u[3], 2.0f * u[2]), u[1]);      // sum (x - gt)^2 = A + 2 e Bq + e^2 Cq
                t[1] = fmaf(e, u[5], u[4]);                            // sum |x - gt|   = L + e Mq
                t[2] = u[6]; t[3] = u[7]; t[4] = u[8];
            } else {
#pragma unroll
                for (int k = 0; k < 5; ++k) t[k] = (s_fold[0][k] + s_fold[1][k]) + (s_fold[2][k] + s_fold[3][k]);
                if (qs.reg_from_part) { t[2] = rows[(long)qs.fold_rows * 5 + 1]; t[3] = rows[(long)qs.fold_rows * 5 + 2]; t[4] = rows[(long)qs.fold_rows * 5 + 3]; }
            }
            st[kStRatio] = st[kStGtSum] / sp_total;
            const float mse = t[0] * q.inv_n3, l1 = t[1] * q.inv_n3;
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
    }
    if (qs.rotate) {                                           // uniform per image: scalar selects of the base pointers
        const bool wr1 = __builtin_amdgcn_readfirstlane((int)(s_state[kStSel] > 0.5f)) != 0;
        if (qs.alt_a) sp.pa = wr1 ? qs.alt_a : q.pa;
        if (qs.alt_r) sp.pr = wr1 ? qs.alt_r : q.pr;
        if (qs.alt_m) sp.pm = wr1 ? qs.alt_m : q.pm;
        sp.pred_next = qs.pred_buf[wr1 ? 1 : 0];
    }
    // ---- the tiles: set A holds the even ones, set B the odd ones; a set is requested again as soon as its tile is done
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int cntw = 0;                                              // entries of this wave's list (uniform)
    for (int t = 0; t < ntile; t += 2) {
        float tot = 0.0f, reg[3] = {0.0f, 0.0f, 0.0f}, acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        bool need0 = false, need1 = false;
        const int p0 = tile_px0(t) + (int)threadIdx.x, p1 = p0 + kTile;
        if (p0 < P) need0 = pstep_pixel(A, qs, sp, (unsigned)(b * P + p0), f, ratio, sr, improved, tot, reg, acc);
        if (t + 2 < ntile) pstep_load(A, qs, sp, (unsigned)(b * P + pix(t + 2)), f);
        __builtin_amdgcn_sched_barrier(0);
        if (p1 < P) need1 = pstep_pixel(B, qs, sp, (unsigned)(b * P + p1), f, ratio, sr, improved, tot, reg, acc);
        if (t + 3 < ntile) pstep_load(B, qs, sp, (unsigned)(b * P + pix(t + 3)), f);
        __builtin_amdgcn_sched_barrier(0);
        // the block's sums: DPP tree per wave here, the four waves in order at the end (fixed order)
        const int bl = t >> 1;
        const float w = wave_sum_to_lane63(tot);
        if (lane == 63) s_bsum[bl][wave] = w;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float wr = wave_sum_to_lane63(reg[k]);
            if (lane == 63) s_breg[bl][wave][k] = wr;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float wa = wave_sum_to_lane63(acc[k]);
            if (lane == 63) s_bacc[bl][wave][k] = wa;
        }
        if (MODE == kFoldXY) {                                 // listed pixels, in a fixed order: tile, then lane
            const unsigned long long b0 = __ballot(need0), b1 = __ballot(need1);
            const int n0 = __popcll(b0);
            if (need0) s_list[wave][cntw + __popcll(b0 & below)] = (uint16_t)(t * kTile + (int)threadIdx.x);
            if (need1) s_list[wave][cntw + n0 + __popcll(b1 & below)] = (uint16_t)((t + 1) * kTile + (int)threadIdx.x);
            cntw += n0 + __popcll(b1);
            if (lane == 0) s_wcnt[bl][wave] = (uint16_t)(n0 + __popcll(b1));
        }
    }
    if (MODE == kFoldXY) {
        // ---- the pixels this wave listed go to the image's WALK QUEUE, before the workgroup's last barrier: room for all of them is reserved
        // with one atomic (shard: this wave's number among the image's, modulo kWalkShards -- 2048 waves on one word would queue for 20 us),
        // whose round trip runs under the other waves' last tiles.  Where a pixel lands in the queue does not matter: its new model depends on
        // the pixel alone, and the walk sums are order-free.
        if (blockIdx.x == 0 && threadIdx.x < kWalkShards)          // the next iteration's counters (their last reader has long finished)
            qs.walk_cnt[(b * 2 + (qs.walk_par ^ 1)) * kWalkShards + threadIdx.x] = 0u;
        // ... and the sums THIS iteration's walk adds to (parity walk_par; the next step's heads read them): their last readers were the heads of the
        // step before this one, and this launch's heads read the other parity
        if (qs.walk_acc != nullptr && blockIdx.x == 0 && threadIdx.x < 6 * kWalkShards) qs.walk_acc[((long)b * 2 + qs.walk_par) * kWalkShards * 6 + threadIdx.x] = 0;
        if (cntw > 0) {                                            // (uniform per wave)
            const int shard = ((int)blockIdx.x * 4 + wave) % kWalkShards;
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(qs.walk_cnt + (b * 2 + qs.walk_par) * kWalkShards + shard, (uint32_t)cntw);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            uint32_t* qd = qs.walk_queue + ((long)b * kWalkShards + shard) * walk_shard_cap(P) + base;
            for (int e = lane; e < cntw; e += 64) {
                const int id = (int)s_list[wave][e];
                qd[e] = (uint32_t)(tile_px0(id >> 8) + (id & (kTile - 1)));
            }
        }
        PS_NOTE(15, cntw);
    }
    __syncthreads();
    // ---- per-block results: the sum of the render of the streamed pixels (the four waves in order) and the regulariser sums; the listed
    // pixels of a block, compacted in a fixed order (wave, tile, lane: whatever the batch size and the blocks per workgroup), go to the block's
    // list in lazy_state -- the work list of lazy_pwalk_kernel, and what matpbr_lazy_state_unpack reads
    if (MODE == kFoldXY) {
        // this wave's entries are ordered by tile, hence by block: entry e of the wave sits in block bl at position
        // (entries of earlier waves in bl) + (e - entries of this wave in earlier blocks)
        int start = 0;
        for (int bl = 0; bl < nb; ++bl) {
            const int mine = (int)s_wcnt[bl][wave];
            int before = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) before += w < wave ? (int)s_wcnt[bl][w] : 0;
            uint16_t* list = qs.lists + ((long)b * nblk_img + blockIdx.x + bl * gridDim.x) * kLazyBlockPixels;
            for (int e = lane; e < mine; e += 64) list[before + e] = (uint16_t)((int)s_list[wave][start + e] & (kLazyBlockPixels - 1));
            start += mine;
        }
    }
    if ((int)threadIdx.x < nb) {
        const int bl = threadIdx.x;
        qs.block_sums[(long)b * qs.n_sums + blockIdx.x + bl * gridDim.x] = (s_bsum[bl][0] + s_bsum[bl][1]) + (s_bsum[bl][2] + s_bsum[bl][3]);
        if (MODE == kFoldXY) qs.walk_fix[(long)b * nblk_img + blockIdx.x + bl * gridDim.x] = 0;      // lazy_pwalk_kernel adds to it
        if (qs.counts)
            qs.counts[(long)b * nblk_img + blockIdx.x + bl * gridDim.x] =
                MODE == kFoldXY ? (uint32_t)(((int)s_wcnt[bl][0] + (int)s_wcnt[bl][1]) + ((int)s_wcnt[bl][2] + (int)s_wcnt[bl][3])) : 0u;
    }
    // the block's record for the NEXT step's heads: S, (A, Bq, Cq, L, Mq), the three regulariser sums.  Two sets of records, by the parity of the
    // iteration: this launch's heads read the other set -- a workgroup that starts late must not find what an early one has already finished
    if (qs.rec_out && (int)threadIdx.x < 9 * nb) {
        const int bl = threadIdx.x / 9, k = threadIdx.x - 9 * bl;
        float v;
        if (k == 0) v = (s_bsum[bl][0] + s_bsum[bl][1]) + (s_bsum[bl][2] + s_bsum[bl][3]);
        else if (k < 6) v = (s_bacc[bl][0][k - 1] + s_bacc[bl][1][k - 1]) + (s_bacc[bl][2][k - 1] + s_bacc[bl][3][k - 1]);
        else v = (s_breg[bl][0][k - 6] + s_breg[bl][1][k - 6]) + (s_breg[bl][2][k - 6] + s_breg[bl][3][k - 6]);
        qs.rec_out[((long)b * nblk_img + blockIdx.x + bl * gridDim.x) * 9 + k] = v;
    }
    if (qs.reg_sums && (int)threadIdx.x < 3 * nb) {
        const int bl = threadIdx.x / 3, k = threadIdx.x - 3 * bl;
        qs.reg_sums[((long)b * nblk_img + blockIdx.x + bl * gridDim.x) * 3 + k] = (s_breg[bl][0][k] + s_breg[bl][1][k]) + (s_breg[bl][2][k] + s_breg[bl][3][k]);
    }
    PS_NOTE(13, __builtin_amdgcn_s_memrealtime());
}

// =================================================================================================
// matpbr_brdf_phase_resolve of a folded phase
// =================================================================================================
// The folded steps store no render (LazyStepArgs::no_pred): what the caller reads is formed here, when it asks.  `out`: the render of the CURRENT
// parameters -- the step kernel's expression on the models' words, the bits the next step will judge.  best_img (MATPBR_FLAG_ROTATE_BEST, whose
// steps store no snapshot either): max(render of SaveBest's maps x their exposure ratio, eps)^(1/2.2), the render being the model where it is
// exact in the best values (kFoldGH: linear in the albedo; a part that leaves the roughness alone: linear in the metallic; a pixel without
// geometry: a constant) and the renderer's own sum (shade_kernel, launched before this kernel into best_lin) where the roughness has moved --
// the models have moved on since the best iteration, the maps have not.  One thread per pixel.
struct FoldResolveArgs {
    const float *a, *r, *m;            // the current parameters (raw)
    const float *best_a, *best_m;      // SaveBest's maps (clamped); read only where named above
    const uint32_t* fplane[kFxPlanes];
    float* out;                        // [B,H,W,3]
    const float* best_lin;             // [B,H,W,3] kFoldXY with slopes: the exact render of SaveBest's maps
    float* best_img;                   // null: no best render asked for
    const float* state;                // [B][kStateStride] of the last step
    int slopes;
};
template <int MODE>
__global__ __launch_bounds__(kBlock) void fold_resolve_kernel(const FoldResolveArgs q, int P) {
    const int b = blockIdx.y;
    const int p = (int)blockIdx.x * kBlock + (int)threadIdx.x;
    if (p >= P) return;
    const unsigned i = (unsigned)(b * P + p), o1 = i * 4u, o3 = i * 12u, of = fx_off(i);
    const float bratio = q.best_img ? q.state[b * kStateStride + kStBestRatio] : -1.0f;
    float rgb[3], best[3] = {0.0f, 0.0f, 0.0f};
    if (MODE == kFoldXY) {
        const float r = fminf(fmaxf(ldf(q.r, o1), 0.07f), 1.0f), m = fminf(fmaxf(ldf(q.m, o1), 0.0f), 1.0f);
        uint32_t xy[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) xy[k] = ldu(q.fplane[kFxXY + k], of);
        float X0[3], Y0[3];
        xy_unpack(xy, X0, Y0);
        const uint32_t lohi_w = ldu(q.fplane[kFxLoHi], of);
        const bool constant = (lohi_w & 0xffffu) == 0xffffu;
        float rref = 0.0f;
        uint32_t lohi = 0u, qq = 0u, sw[3] = {0u, 0u, 0u};
        if (q.slopes) {                                         // (as pstep_load_fixed)
            rref = as_f(ldu(q.fplane[kFxRref], of)); lohi = lohi_w; qq = ldu(q.fplane[kFxQ], of);
#pragma unroll
            for (int c = 0; c < 3; ++c) sw[c] = ldu(q.fplane[kFxS + c], of);
        }
        const float dr = r - rref;
        const float mb = bratio >= 0.0f && q.best_m ? fminf(fmaxf(ldf(q.best_m, o1), 0.0f), 1.0f) : m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float X, Y;
            xy_eval(X0[c], Y0[c], sw[c], xy_x2(lohi, qq, c), dr, X, Y);
            rgb[c] = fmaf(m, Y, X);
            if (bratio >= 0.0f) best[c] = constant ? rgb[c] : (q.slopes ? ldf(q.best_lin, o3 + 4u * c) : fmaf(mb, Y, X));
        }
    } else {
        const F3 av = ld3(q.a, o3);
        const float a[3] = {fminf(fmaxf(av.x, 0.0f), 1.0f), fminf(fmaxf(av.y, 0.0f), 1.0f), fminf(fmaxf(av.z, 0.0f), 1.0f)};
        float ab[3] = {a[0], a[1], a[2]};
        if (bratio >= 0.0f && q.best_a) {
            const F3 bv = ld3(q.best_a, o3);
            ab[0] = fminf(fmaxf(bv.x, 0.0f), 1.0f); ab[1] = fminf(fmaxf(bv.y, 0.0f), 1.0f); ab[2] = fminf(fmaxf(bv.z, 0.0f), 1.0f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float G = as_f(ldu(q.fplane[kFgG + c], of)), Hc = as_f(ldu(q.fplane[kFgH + c], of));
            rgb[c] = fmaf(a[c], G, Hc);
            best[c] = fmaf(ab[c], G, Hc);
        }
    }
    st3(q.out, o3, rgb[0], rgb[1], rgb[2]);
    if (bratio >= 0.0f)
        st3(q.best_img, o3, pow_inv_gamma(fmaxf(best[0] * bratio, kLossEps)), pow_inv_gamma(fmaxf(best[1] * bratio, kLossEps)),
            pow_inv_gamma(fmaxf(best[2] * bratio, kLossEps)));
}

// =================================================================================================
// the pixels lazy_pstep_kernel<kFoldXY> listed, re-sampled
// =================================================================================================
// One wave (a 64-thread workgroup) per eight entries of the image's walk queue (wave w: shard w % kWalkShards, entries 8 (w / kWalkShards) ..):
// eight pixels, eight lanes each (resample_walk_pixel).  The walked pixels' models are rebuilt (generic AND folded planes), their render
// written, and the sum of it added to its block's walk sum in fixed point (integer atomics: order-free, hence reproducible and the same
// for every batch size).  A wave's chain is (counter, entries) -> pixel -> walk -> stores -- no prefix over an image's lists, no search
// (lazy_resample_kernel: 12 k of a wave's 28 k cycles) -- and the waves are full but for one per shard: ~100 waves per 512 x 512 image and
// iteration walk, the others leave after one round trip.  What was tried on the way: one wave per 512-pixel block (3 600 eleven-thousand-
// cycle walks per iteration at 8 x 512^2 for two pixels each: VALU-bound, 40 us); a fixed number of waves per group of four blocks (the
// longest of 1024 lists has ~46 entries: a second pass nearly every iteration with four waves, 31 us; with eight, 8192 workgroups whose
// dispatch alone is 20 us); waves packed four to a workgroup (a 240-register wave needs half a SIMD: a workgroup is placed only where all
// four SIMDs have room, 40 us); chunks of eight filled per workgroup of the step kernel (mostly two or three pixels per chunk: twice the
// waves, and at one image more chunks than waves); the walk inside the step kernel (128 registers: 25 k cycles per walk, 16 us of tail).
__global__ __launch_bounds__(64, 2) void lazy_pwalk_kernel(const LazyStepArgs qs, const float* __restrict__ light, const Geom g, const RuleTable tab) {
    __shared__ float s_light[kNL + 1];
    __shared__ float4 s_ring[kMaxRings];
    __shared__ float2 s_saz[kMaxRings * kMaxAz];
    const JacBwdArgs& q = qs.j;
    // workgroup -> (image, wave of the image), the image running fastest: the waves that have entries (the first ones of every image) are
    // then the first the dispatcher starts (image by image, the last image's would wait for 1800 empty workgroups: 5 us).  The images rotate
    // against the workgroup index from one group of B workgroups to the next: consecutive workgroups go to the XCDs in turn, and with
    // b = blockIdx.x % 8 image b's walkers would all sit on XCD b (measured: 66 to 177 walkers per XCD at 8 x 512^2 for images that walk
    // unequal numbers of pixels; rotated: 111 to 144)
    const int B = qs.batch, wv = (int)blockIdx.x / B, b = ((int)blockIdx.x + wv) % B, nwv = (int)gridDim.x / B;
    const int P = g.H * g.W;
    const int nblk = lazy_fwd_blocks(P);
    const int lane = threadIdx.x;
    const int shard = wv % kWalkShards, k0 = wv / kWalkShards, kstep = nwv / kWalkShards;
    // one round trip for everything that depends on nothing: the stop flag, the list's length, this wave's first entries, the state, the tables
    const float stopped = qs.state_old[b * kStateStride + kStStopped];
    const int n = (int)qs.walk_cnt[(b * 2 + qs.walk_par) * kWalkShards + shard];
    const uint32_t* queue = qs.walk_queue + ((long)b * kWalkShards + shard) * walk_shard_cap(P);
    uint32_t pix = queue[8 * k0 + (lane >> 3)];                             // (inside the list's storage whatever n is)
    const float* st = qs.state_new + b * kStateStride;                     // the state the step kernel has just committed
    const float ratio = st[kStRatio], gt_sum = st[kStGtSum], sel_f = st[kStSel];
    const float lt0 = light[(long)b * kNL + lane], lt1 = lane + 64 < kNL ? light[(long)b * kNL + lane + 64] : 0.0f;
    const float2 saz_v = (&tab.saz[0][0])[lane];
    const float4 ring_v = tab.sring[lane < kMaxRings ? lane : 0];
    if (stopped > 0.5f || 8 * k0 >= n) return;                             // the step kernel skipped the image / nothing for this wave (uniform)
    s_light[lane] = lt0 * kShNorm[lane / 3];
    if (lane + 64 < kNL) s_light[lane + 64] = lt1 * kShNorm[(lane + 64) / 3];
    s_saz[lane] = saz_v;
    if (lane < kMaxRings) s_ring[lane] = ring_v;
    StepPtrs sp{q.pa, q.pr, q.pm, q.pa, q.pr, q.pm, qs.pred_next};          // the parameters as WRITTEN by the step (a map the part does not move: as read)
    if (qs.rotate) {
        const bool wr1 = __builtin_amdgcn_readfirstlane((int)(sel_f > 0.5f)) != 0;
        if (qs.alt_a) sp.a = wr1 ? qs.alt_a : q.pa;
        if (qs.alt_r) sp.r = wr1 ? qs.alt_r : q.pr;
        if (qs.alt_m) sp.m = wr1 ? qs.alt_m : q.pm;
        sp.pred_next = qs.pred_buf[wr1 ? 1 : 0];
    }
    __syncthreads();
    const float floor_ = 0.5f * gt_sum / (3.0f * (float)P) / (ratio > 0.0f ? ratio : 1.0f);
    const float tol_k = qs.tol * kLzTolK, tol_s = qs.tol * kLzTolS;
    const long BPl = (long)B * P;
    const int sub = lane & 7;
    for (int k = k0; 8 * k < n; k += kstep) {                              // (one pass unless a list is longer than 8 gridDim.x / kWalkShards)
        if (k != k0) pix = queue[8 * k + (lane >> 3)];
        const bool ok = 8 * k + (lane >> 3) < n;
        const int first = __shfl((int)pix, 0);                             // entry 8 k exists; (unconditionally: a shuffle inside the select below
        const int p = ok ? (int)pix : first;                               //  would read lane 0 while it is masked off)
        float rs = 0.0f, rgb3[3] = {0.0f, 0.0f, 0.0f}, a5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        resample_walk_pixel<true>(qs, sp, s_light, s_ring, s_saz, g, tab, b, P, BPl, p, ok, sub, floor_, tol_k, tol_s, rs, rgb3);
        asm volatile("" ::: "memory");                          // (the target's three words are asked for HERE: hoisted above the walk they cost it its second wave per SIMD)
        if (qs.walk_acc != nullptr && ok && sub == 0) {
            const unsigned o3w = (unsigned)(b * P + p) * 12u;
#pragma unroll
            for (int c = 0; c < 3; ++c) loss_acc(rgb3[c], ldf(q.gt_srgb, o3w + 4 * c), ratio, a5);
        }
        if (ok && sub == 0)
            atomicAdd((unsigned long long*)(qs.walk_fix + (long)b * nblk + p / kLazyBlockPixels), (unsigned long long)(long long)__double2ll_rn((double)rs * kWalkFix));
        // the walked pixels' shares of the next iteration's statistics: each pixel's six numbers in fixed point, then INTEGER sums (a wave's eight
        // pixels first, one atomic per number and wave, on the SHARD's six words: on six words per image 130 waves queued for 15 us): order-free, hence
        // the same bits in every run and for every batch size
        if (qs.walk_acc != nullptr) {
            const bool mine = ok && sub == 0;
            long long v6[6];
            v6[0] = mine ? __double2ll_rn((double)rs * kWalkFix) : 0;
#pragma unroll
            for (int k2 = 0; k2 < 5; ++k2) v6[1 + k2] = mine ? __double2ll_rn((double)a5[k2] * kWalkFix) : 0;
#pragma unroll
            for (int k2 = 0; k2 < 6; ++k2) {
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) v6[k2] += __shfl_xor(v6[k2], off);      // lanes 0, 8, ..., 56 hold the pixels
                if (lane == 0) atomicAdd((unsigned long long*)(qs.walk_acc + (((long)b * 2 + qs.walk_par) * kWalkShards + shard) * 6 + k2), (unsigned long long)v6[k2]);
            }
        }
    }
}

}  // namespace matpbr
