#!/usr/bin/env python3
"""Golden vectors for the ATTACHED sampling gradient of the live reference (a5 of SURVEY.md section 8a).

`MatDiffBSDF.sample_brdf` (myutils/mi_plugin.py:1296-1341) is torch-differentiable under the shim of gen_golden.py: the sampled
direction depends on the roughness through mi_specular_sampler (:217-253, alpha is not detached) and the MC weight
f cos / (pdf + 1e-6) keeps D in its pdf (:1335-1341).  This script records, for the 1024 lanes of sample_brdf.npz, the reference's
own autograd derivatives d weight_c / d r (c = r, g, b) and d wi / d r.  Inputs + outputs only; run in the build container:

    python tests/golden/gen_sample_brdf_grad.py        # writes tests/golden/sample_brdf_grad.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (the shim lives there)


def main():
    G._install_shim()
    import myutils.mi_plugin as P

    z = np.load(os.path.join(HERE, "sample_brdf.npz"))
    t = lambda k: torch.from_numpy(z[k])
    Ns = z["r"].shape[0]
    s_n, s_wo, s_a, s_m, s1, s2 = t("n"), t("wo"), t("a"), t("m"), t("sample1"), t("sample2")
    s_r = t("r").clone().requires_grad_(True)
    self = types.SimpleNamespace(use_mesh_normal=True)
    self.a = types.SimpleNamespace(array=s_a.T.reshape(-1), shape=(1, Ns, 3))
    self.r = types.SimpleNamespace(array=s_r, shape=(1, Ns, 1))
    self.m = types.SimpleNamespace(array=s_m, shape=(1, Ns, 1))
    self.n = types.SimpleNamespace(array=s_n.T.reshape(-1), shape=(1, Ns, 3))
    self.eval_brdf = lambda *args: P.MatDiffBSDF.eval_brdf(self, *args)
    screen = G._vec(torch.arange(Ns, dtype=torch.float64) + 0.5, torch.zeros(Ns))
    # The masked assignment `wi[mask] = ...` of :1331-1333 goes through the shim's lazily materialised Vector3f(0.0), which autograd
    # cannot trace; the two samplers are therefore called directly (the reference's own functions, all lanes each) and selected per
    # lane with the mask of :1331, followed by the reference's eval_brdf and the weight of :1336-1339 written out below.
    mask = (s1 > 0.5)
    wi_d = P.mi_diffuse_sampler(G.V.wrap(s2), G.V.wrap(s_n)).as_subclass(torch.Tensor)
    wi_s = P.mi_specular_sampler(G.V.wrap(s2), s_r, G.V.wrap(s_wo), G.V.wrap(s_n)).as_subclass(torch.Tensor)
    wi = torch.where(mask.unsqueeze(0), wi_d, wi_s)
    brdf, pdf = self.eval_brdf(G.V.wrap(wi), G.V.wrap(s_wo), G.V.wrap(s_n), None, screen)
    brdf, pdf = brdf.as_subclass(torch.Tensor), pdf.as_subclass(torch.Tensor)
    w = torch.where(pdf > 1e-6, brdf / (pdf + 1e-6), torch.zeros_like(brdf))      # :1336-1338
    np.testing.assert_allclose(G._np(wi), z["wi"], rtol=1e-12, atol=1e-14)        # same lanes / values as sample_brdf.npz
    np.testing.assert_allclose(G._np(w), z["weight"], rtol=1e-12, atol=1e-14)
    dw = np.stack([G._np(torch.autograd.grad(w[c].sum(), s_r, retain_graph=True)[0]) for c in range(3)])
    dwi = np.stack([G._np(torch.autograd.grad(wi[c].sum(), s_r, retain_graph=True)[0]) for c in range(3)])
    dpdf = G._np(torch.autograd.grad(pdf.sum(), s_r)[0])
    np.savez(os.path.join(HERE, "sample_brdf_grad.npz"), dweight_dr=dw, dwi_dr=dwi, dpdf_dr=dpdf)
    spec = z["sample1"] <= 0.5
    print("lanes", Ns, "specular", int(spec.sum()), "|dwi/dr| specular mean", float(np.abs(dwi[:, spec]).mean()),
          "diffuse max", float(np.abs(dwi[:, ~spec]).max()))


if __name__ == "__main__":
    main()
