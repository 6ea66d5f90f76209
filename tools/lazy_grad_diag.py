"""Where do the tail errors of the lazy step's d loss / d r come from?  Jacobian-only comparison (d loss / d pred formed on the lazy loop's own
render on both sides, as tests/test_gpu_lazy.py) at a few iterations of an 'rm' part, folded against generic planes, with the worst pixels'
model state printed (dr against the interval, the size of the first-order term, whether the e-cap bit).  usage: python tools/lazy_grad_diag.py [iterations...]
env: DIAG_IMAGE (synthetic scene, default 0), DIAG_TOL (FusedBrdfPhase lazy_tol, default 1.0), DIAG_FOLD_ONLY (skip the generic step), DIAG_DUMP."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from materialist_amd import loop, ops, render, synthetic

dev = torch.device("cuda:0")
H = W = 512
spp = 64
sc = synthetic.make_scene(int(os.environ.get('DIAG_IMAGE', '0')), H, W)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
scene._set("emitter.data", t(sc.light))
with torch.no_grad():
    gt = render.render_w_brdf(scene, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, spp)
init = [t(x) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
checks = [int(x) for x in (sys.argv[1:] or [50, 200, 400, 800, 1200])]
for fold in ((True,) if os.environ.get('DIAG_FOLD_ONLY') else (True, False)):
    ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp, lazy=True, keep_grads=True, fold=fold, lazy_tol=float(os.environ.get("DIAG_TOL", "1.0")))
    exact, jac = torch.empty_like(gt), ops.plane9(gt)
    g_ref = {k: torch.empty_like(v) for k, v in ph.g.items()}
    for it in range(max(checks) + 1):
        if it in checks:
            p_at = [ph.p[k].clone() for k in ("albedo", "roughness", "metallic")]
            pred_at = ph.pred.clone()
            st, _ = ops.lazy_state_unpack(ph.lazy_state, ph.p["albedo"])
        ph.step()
        if it in checks:
            ops.shade_fwd(*p_at, ph.n, ph.light, spp, clamp_params=True, out=exact, dcache=ph.dcache, jac=jac)
            ops.brdf_loss_bwd_jac(*p_at, jac, pred_at, ph.gt_srgb, ph.stats, ph.orig["albedo"], ph.orig["roughness"], ph.orig["metallic"], 0.1,
                                  g_ref["albedo"], g_ref["roughness"], g_ref["metallic"], optimize_part="rm")
            g, gr = ph.g["roughness"].reshape(-1), g_ref["roughness"].reshape(-1)
            e = (g - gr).abs() / torch.maximum(gr.abs(), gr.abs().mean())
            s = st.reshape(-1, st.shape[-1])
            dr = p_at[1].reshape(-1).clamp(0.07, 1) - s[:, 0]
            lo, hi = s[:, 1], s[:, 2]
            frac_int = torch.where(dr >= 0, dr / hi.clamp_min(1e-9), -dr / lo.clamp_min(1e-9))
            dS, eS = s[:, 16:22], s[:, 22:28]
            first = (eS.abs() * dr.abs()[:, None]).amax(1) / dS.abs().amax(1).clamp_min(1e-9)       # size of the first-order term relative to the derivative
            print(f"fold={fold} it={it}: L2 {float((g-gr).norm()/gr.norm()):.2e}  p99.9 {float(torch.quantile(e[::2].float(), 0.999)):.2e}  max {float(e.max()):.2e}  "
                  f"n(e>1e-3) {int((e > 1e-3).sum())}  n(e>5e-3) {int((e > 5e-3).sum())}  mean |dr| {float(dr.abs().mean()):.2e}  mean first-order/deriv {float(first.mean()):.3f}")
            # the models' own d out_c / d r (generic planes as they stand, half-precision words widened) against the exact one, per channel
            a_, m_ = p_at[0].reshape(-1, 3).clamp(0, 1), p_at[2].reshape(-1).clamp(0, 1)
            rr = p_at[1].reshape(-1).clamp(0.07, 1)
            dc = ph.dcache.reshape(9, -1)
            Jm = torch.stack([a_[:, c] * (1 - m_) * (dc[3 + c] + 2 * rr * dc[6 + c]) + (0.04 * (1 - m_) + m_ * a_[:, c]) * (s[:, 16 + c] + s[:, 22 + c] * dr) + (s[:, 19 + c] + s[:, 25 + c] * dr) for c in range(3)])
            Je = jac[6:9].reshape(3, -1)
            eJ = ((Jm - Je).abs() / torch.maximum(Je.abs(), Je.abs().mean())).amax(0)
            print(f"   models' own d out/d r per channel vs exact: p99.9 {float(torch.quantile(eJ[::2].float(), 0.999)):.2e} max {float(eJ.max()):.2e} n(>1e-3) {int((eJ > 1e-3).sum())};  "
                  f"of the {int((e > 2e-3).sum())} pixels with gradient error > 2e-3: {int(((e > 2e-3) & (eJ > 1e-3)).sum())} have a model error > 1e-3 (the others: cancellation between channels / terms)")
            J = jac[6:9].reshape(3, -1).abs()
            _, refd = ops.lazy_state_unpack(ph.lazy_state, ph.p["albedo"])
            print(f"   mean |d out/d r| {float(J.mean()):.3e}  median {float(J.median()):.3e}  parity floor 0.5 mean(gt) {0.5 * float(gt.mean()):.3e}  re-sampled this iteration {float(refd.float().mean()):.5f}")
            idx = torch.argsort(e, descending=True)[:4]
            for i in idx.tolist():
                print(f"   px {i}: eJ {float(eJ[i]):.2e} J exact {Je[:, i].tolist()} model {Jm[:, i].tolist()} e {float(e[i]):.2e} r {float(p_at[1].reshape(-1)[i]):.4f} dr {float(dr[i]):+.2e} lo {float(lo[i]):.2e} hi {float(hi[i]):.2e} rho {float(s[i,3]):.2e} "
                      f"frac {float(frac_int[i]):.2f} first/deriv {float(first[i]):.3f} g {float(g[i]):+.3e} gref {float(gr[i]):+.3e} gmean {float(gr.abs().mean()):.3e}")
            if os.environ.get("DIAG_DUMP"):      # the worst pixels' inputs and model state, for a CPU session with the oracle
                ii = idx.cpu().numpy()
                np.savez(os.environ["DIAG_DUMP"], idx=ii, state=s[idx].cpu().numpy(), a=a_[idx].cpu().numpy(), r=rr[idx].cpu().numpy(), m=m_[idx].cpu().numpy(),
                         n=ph.n.reshape(-1, 3)[idx].cpu().numpy(), light=ph.light.cpu().numpy(), J_exact=Je[:, idx].cpu().numpy(), J_model=Jm[:, idx].cpu().numpy(),
                         dcache=dc[:, idx].cpu().numpy(), floor=0.5 * float(gt.mean()))
            # error against position in the interval and against the size of the first-order term
            for lo_f, hi_f in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.01)):
                sel = (frac_int >= lo_f) & (frac_int < hi_f)
                if bool(sel.any()):
                    print(f"   interval position {lo_f:.2f}-{hi_f:.2f}: {int(sel.sum())} px, mean e {float(e[sel].mean()):.2e}, max e {float(e[sel].max()):.2e}")
