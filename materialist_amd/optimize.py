"""`optimize_envmap_ARMN` (inverse_img_w_mi.py:106-599) on the HIP render, `--model_name none` mode: the alternating
env / BRDF optimisation of one image (or a batch of independent images) driven by `schedule.run_schedule`.

Differences from the reference, all forced by scope:
  * the envmap MLP (`PosMLP(output_type='envmap')`, :117-124) is f2/next; until it exists the light is optimised directly as
    16x32 texels through a softplus (the MLP's own output activation; zero-initialised parameters give ln 2 everywhere,
    exactly what the zero-initialised last layer of the MLP produces at the first epoch);
  * no frame dumps / mp4 / file outputs here (f1); the function returns tensors and the decision trace.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import loop as _loop
from . import loss as _loss
from . import render as _render
from .schedule import TraceEvent, run_schedule

ROUGHNESS_SHIFT, METALLIC_SHIFT = 0.7, 0.05   # :183-184


def optimize_envmap_ARMN(scene: _render.Scene, mat: Dict[str, torch.Tensor], optimize_order: Sequence[str] = ("arm",), spp: int = 64,
                         opt_env_from: int = 0, opt_src: str = "arm", scale_delta: float = 0.1, num_epochs: int = 5000,
                         sync_every: int = 25, env_size=(16, 32), log=None, frames=None, results_dir: Optional[str] = None,
                         shading_normal: Optional[torch.Tensor] = None, model_name: str = "none", use_mask: bool = False,
                         digests: Optional[list] = None) -> Dict[str, object]:
    """mat: albedo [H,W,3], roughness [H,W,1], metallic [H,W,1], normal [H,W,3], gt_image [H,W,3] (optionally gt_envmap).
    Returns the best maps / envmap / render, the final PSNR and the schedule trace.  `frames` (pipeline.FrameWriter) and
    `results_dir` switch on the reference's file outputs: a frame at every host poll (the reference: every 10 epochs,
    :257,438,559) and best_results/ after each phase (:302-303,465,590)."""
    dev = mat["gt_image"].device
    gt = mat["gt_image"].contiguous()
    mat = dict(mat)

    def stage_digest(stage: str, *tensors) -> None:
        # `digests` (a list the caller hands in): (stage, SHA-256 of the stage's tensors) at the schedule's boundaries -- what a run that came out
        # different is compared on to find the FIRST stage that differs (tools/pipeline_hashes.py, tests/golden/indoor2_digests.json).  A stage
        # boundary already synchronises with the host (the pollers have returned), so the copies cost nothing the loops would notice
        if digests is not None:
            digests.append((stage, _loss.tensors_digest(*tensors)))

    stage_digest("inputs", gt, mat["albedo"], mat["roughness"], mat["metallic"], scene.shading_normal())
    if "r" not in opt_src:                                                         # :185-188
        mat["roughness"] = torch.full_like(mat["roughness"], ROUGHNESS_SHIFT)
    if "m" not in opt_src:
        mat["metallic"] = torch.full_like(mat["metallic"], METALLIC_SHIFT)
    params = _render.traverse(scene)                                               # :216-220
    params["shape.bsdf.a"], params["shape.bsdf.r"], params["shape.bsdf.m"] = mat["albedo"], mat["roughness"], mat["metallic"]
    if not scene.use_mesh_normal:                                                  # 'n' in opt_order: shade with the predicted normal map (:335-340)
        mat["normal"] = torch.nn.functional.normalize(mat["normal"], p=2, dim=-1)  # :193
        params["shape.bsdf.n"] = mat["normal"]

    # regulariser anchors albedo_ori / roughness_ori / metallic_ori / normal_ori: captured ONCE, before the loops (:189-201), and
    # used by every part of every loop (:398-409) -- not the previous part's best maps
    originals = {k: mat[k].detach().clone() for k in ("albedo", "roughness", "metallic")}
    if not scene.use_mesh_normal:
        originals["normal"] = mat["normal"].detach().clone()

    mask = mat.get("mask") if use_mask else None                                   # --use_mask (:379-381,509-511,702-711)
    if use_mask and mask is None:
        raise ValueError("use_mask needs mat['mask'] ([H,W] bool)")
    saver = _loop.DeviceSaveBest()

    def keep_best(key: str, new: torch.Tensor, improved: torch.Tensor) -> None:
        """SaveBest per image: images of a batch whose best loss did not improve keep their earlier snapshot."""
        old = saver.best.get(key)
        if old is None or old.shape != new.shape or gt.ndim != 4:
            saver.best[key] = new.clone()
        else:
            sel = improved.to(new.device).reshape((-1,) + (1,) * (new.ndim - 1))
            saver.best[key] = torch.where(sel, new, old)

    state = {"final_envmap": None, "last_mse": None, "best_brdf_weights": None}
    if model_name == "pos_mlp":                                                     # :114-124,159-172,179-207
        from . import posmlp

        if gt.ndim != 3:
            raise NotImplementedError("pos_mlp mode optimises one image per call")
        env_net = posmlp.envmap_net().to(dev)
        start_envmap = torch.ones(env_size[0] * env_size[1], 3, device=dev)
        env_params = list(env_net.parameters())
        env_head = lambda: env_net(start_envmap).reshape(tuple(env_size) + (3,))
        armn = not scene.use_mesh_normal                                            # output_type (:159-172,203-206)
        brdf_net = posmlp.brdf_net("armn" if armn else "arm").to(dev)
        start_arm = torch.cat([mat["albedo"].reshape(-1, 3), mat["roughness"].reshape(-1, 1), mat["metallic"].reshape(-1, 1)], dim=-1)
        start_arm = torch.cat([start_arm, mat["normal"].reshape(-1, 3)], dim=-1) if armn else start_arm.clamp(0, 1)
    elif model_name == "none":
        # one light per image: a batch of independent images ([B,H,W,3] target) optimises B envmaps side by side
        lead = (gt.shape[0],) if gt.ndim == 4 else ()
        env_raw = torch.zeros(lead + tuple(env_size) + (3,), dtype=torch.float32, device=dev, requires_grad=True)
        env_params = [env_raw]
        env_head = lambda: torch.nn.functional.softplus(env_raw)
    else:
        raise ValueError("model_name should be 'none' or 'pos_mlp'")
    import time

    t_start = time.perf_counter()
    say = (lambda msg: log(f"[{time.perf_counter() - t_start:7.2f} s] {msg}")) if log is not None else (lambda *_: None)

    # ------------------------------------------------------------------ hot loop A (:236-254), device-resident
    background = scene.bg_mask is not None                                          # mesh_mask.png: pixels that see the environment directly

    def env_phase_runner_background(loop_num: int, lr_of, patience: int, min_delta: float, max_epochs: int):
        """Hot loop A on the operator face (scenes with pixels that see the environment directly)."""
        opt = torch.optim.Adam(env_params, lr=lr_of(0))
        ph = _loop.EnvHeadPhase(scene, gt, env_head, opt, spp=spp, saver=_loop.DeviceSaveBest())
        if saver.best_loss is not None:
            ph.saver.best_loss = saver.best_loss.clone().reshape(())
        es = _loop.EarlyStopping(patience, min_delta) if patience > 0 else None
        done, stop, mse = 0, "num_epochs", float("nan")
        while done < max_epochs:
            _loop.set_lr(opt, lr_of(done))
            mse = float(ph.step())
            done += 1
            if frames is not None and done % sync_every == 0 and frames.due("env"):
                frames.env_frame(loop_num, done - 1, gt, ph.pred, env_head().detach())
            if es is not None:
                es(mse)
                if es.early_stop:
                    stop = "early_stop"
                    break
        new_best = ph.saver.best_loss.reshape(-1)
        prev = saver.best_loss if saver.best_loss is not None else torch.full_like(new_best, float("inf"))
        if bool((new_best < prev).any()):
            saver.best_loss = torch.minimum(new_best, prev)
            saver.best.update(albedo=mat["albedo"].detach().clone(), roughness=mat["roughness"].detach().clone(),
                              metallic=mat["metallic"].detach().clone(), envmap=ph.saver.best["envmap"].clone(),
                              rendered_img=ph.saver.best["rendered_img"].clone())
        elif "envmap" not in saver.best:
            saver.best["envmap"] = ph.saver.best["envmap"].clone()
        state["last_mse"] = mse
        return done - 1, stop, mse

    def env_phase_runner(loop_num: int, lr_of, patience: int, min_delta: float, max_epochs: int):
        if background and gt.ndim != 3 and model_name == "pos_mlp":
            return env_phase_runner_background(loop_num, lr_of, patience, min_delta, max_epochs)
        graph = max_epochs > 8 and gt.is_cuda
        if model_name == "pos_mlp":
            # the reference's parameterisation (envmap_net, :117-124,238-239), every launch of the iteration on the C ABI
            from .envhead import EnvMlpPhase

            ph = EnvMlpPhase(scene, gt, env_net, start_envmap, spp=spp, lr=lr_of(0), patience=patience, min_delta=min_delta,
                             best_mse=saver.best_loss, history_len=max_epochs, use_graph=graph, env_size=env_size)
            set_lr, head_now = ph.set_lr, (lambda: ph.head())
        elif gt.ndim == 3 and gt.is_cuda and tuple(env_size)[0] * tuple(env_size)[1] <= 1024:
            # the texels themselves through a softplus: head, its backward and Adam on the C ABI too (seven kernels per iteration)
            from .envhead import EnvTexelPhase

            ph = EnvTexelPhase(scene, gt, env_raw, spp=spp, lr=lr_of(0), patience=patience, min_delta=min_delta, best_mse=saver.best_loss,
                               history_len=max_epochs, use_graph=graph)
            set_lr, head_now = ph.set_lr, (lambda: ph.head())
        else:
            opt = _loop.capturable_adam(env_params, lr_of(0)) if graph else torch.optim.Adam(env_params, lr=lr_of(0))   # fresh Adam per loop (:225-229)
            ph = _loop.FusedEnvPhase(scene, gt, env_head, opt, spp=spp, patience=patience,
                                     min_delta=min_delta, best_mse=saver.best_loss, history_len=max_epochs, use_graph=graph)
            set_lr, head_now = (lambda lr: _loop.set_lr(opt, lr)), (lambda: env_head().detach())
        done, stop, lr_now = 0, "num_epochs", lr_of(0)
        while done < max_epochs:
            k = min(sync_every, max_epochs - done)
            while k > 0:
                if lr_of(done) != lr_now:
                    lr_now = lr_of(done)
                    set_lr(lr_now)
                run = 1
                while run < k and lr_of(done + run) == lr_now:               # the iterations up to the next change of the learning rate
                    run += 1
                if run > 1 and hasattr(ph, "step_many"):
                    ph.step_many(run)                                        # one graph of `run` unrolled iterations
                else:
                    run = 1
                    ph.step()
                done, k = done + run, k - run
            info = ph.poll()
            if frames is not None and gt.ndim == 3 and frames.due("env"):
                frames.env_frame(loop_num, done - 1, gt, ph.pred, head_now())
            if bool(info["stopped"].all()):
                stop = "early_stop"
                break
        info = ph.poll()
        if hasattr(ph, "sync_params"):
            ph.sync_params()                                                         # env_raw as an optimiser over it would have left it
        iters = int(info["iters"].max())
        prev = saver.best_loss if saver.best_loss is not None else torch.full_like(info["best_mse"].to(dev), float("inf"))
        imp = info["best_mse"].to(dev) < prev
        if bool(imp.any()):                                                          # SaveBest.update on improvement (:247)
            saver.best_loss = torch.minimum(info["best_mse"].to(dev), prev)
            for k_, v in (("albedo", mat["albedo"].detach()), ("roughness", mat["roughness"].detach()), ("metallic", mat["metallic"].detach()),
                          ("envmap", ph.best_env), ("rendered_img", ph.best_img)):
                keep_best(k_, v, imp)
        elif "envmap" not in saver.best:
            saver.best["envmap"] = ph.best_env.clone()
        state["last_mse"] = float(ph.history()[iters - 1].max()) if iters > 0 else float("nan")
        return iters - 1, stop, state["last_mse"]

    def _save_results() -> None:
        if results_dir is not None and gt.ndim == 3:
            from .pipeline import save_results

            nrm = mat.get("normal", shading_normal if shading_normal is not None else scene.shading_normal())
            save_results(results_dir, saver.best, nrm)

    def on_env_phase_end(loop_num: int, save: bool) -> None:
        state["final_envmap"] = saver.best["envmap"].detach().clone()              # :296
        if frames is not None and gt.ndim == 3 and "rendered_img" in saver.best:
            frames.env_frame(loop_num, 9999, gt, saver.best["rendered_img"] if saver.best["rendered_img"].shape == gt.shape else gt,
                             state["final_envmap"], final=True)                     # opt_env_img.png (:298)
        if save:
            _save_results()                                                         # :302-303
        stage_digest(f"loop {loop_num} env", state["final_envmap"])
        say(f"loop {loop_num}: env phase done, mse {state['last_mse']:.5f}")

    def on_brdf_phase_begin(loop_num: int, which: str) -> None:                    # :317-342
        if which == "gt_or_ones":
            env = mat["gt_envmap"] if "gt_envmap" in mat else torch.ones(tuple(env_size) + (3,), device=dev)
        else:
            env = state["final_envmap"]
        params["emitter.data"] = env.detach()
        state["envmap4render"] = env.detach()

    # ------------------------------------------------------------------ hot loop B (:347-468), device-resident
    def brdf_part_runner_mlp(loop_num: int, part: str, patience: int, min_delta: float, n_epochs: int):
        from .armhead import ArmMlpPhase

        why = ArmMlpPhase.why_not(scene, gt, brdf_net, part, mask)
        if why is not None:           # not silently: the composition is several times slower than the launch-by-launch phase
            say(f"loop {loop_num}: part {part!r} (pos_mlp) runs the autograd composition, not the launch-by-launch phase: {why}")
        ph = _loop.pos_mlp_brdf_phase(scene, gt, brdf_net, start_arm, {k: mat[k] for k in ("albedo", "roughness", "metallic")},
                                      optimize_part=part, spp=spp, scale_delta=scale_delta, patience=patience, min_delta=min_delta,
                                      best_mse=saver.best_loss, history_len=n_epochs, mask=mask)
        stop, it = "num_epochs", 0
        for it in range(n_epochs):
            if ph.step_and_check():                                                 # per-epoch host check, as the reference (:550)
                stop = "early_stop"
                break
            if frames is not None and it % 10 == 0 and frames.due("mat"):
                frames.mat_frame(loop_num, part, it, gt, _loss.linear_to_srgb((ph.pred * ph.stats[0, 0]).clamp_min(1e-8)),
                                 {k: ph.best[k] for k in ("albedo", "roughness", "metallic")},
                                 shading_normal if shading_normal is not None else scene.shading_normal())
        best = ph.stats[:, ph.ops.STAT_BEST].clone()
        prev = saver.best_loss if saver.best_loss is not None else torch.full_like(best, float("inf"))
        if bool((best < prev).any()):
            saver.best_loss = torch.minimum(best, prev)
            for k_ in ("albedo", "roughness", "metallic"):
                saver.best[k_] = ph.best[k_].clone()
            saver.best["rendered_img"] = ph.best_img.clone()
            saver.best["envmap"] = state["envmap4render"].clone()
            state["best_brdf_weights"] = ph.best_weights
        if state["best_brdf_weights"] is not None:
            brdf_net.load_state_dict(state["best_brdf_weights"])                    # :586-587: reloaded after every part
        # the device-side EarlyStopping is polled every few iterations: the iterations enqueued between the stop and the poll were no-ops,
        # so the epoch and the learning rate reported are those of the last iteration that really ran
        if hasattr(ph, "iterations_run"):
            it = max(ph.iterations_run - 1, 0)
            lr_end = ph.lr_at(it) if hasattr(ph, "lr_at") else ph.opt.param_groups[0]["lr"]
        else:
            lr_end = ph.opt.param_groups[0]["lr"]
        say(f"loop {loop_num}: part {part!r} (pos_mlp) ran {it + 1} iterations ({stop}), best mse {float(best.min()):.5f}")
        return it, lr_end, stop

    def brdf_part_runner_normal(loop_num: int, part: str, patience: int, min_delta: float, n_epochs: int):
        """Parts that optimise the normal map (output_type 'armn', use_mesh_normal False; :335-340,378-379,406-409), and `--use_mask` on
        predicted normals: the autograd render with the torch-composed loss (BrdfPhase) and the reference's per-epoch host EarlyStopping."""
        say(f"loop {loop_num}: part {part!r} runs the autograd composition on the operator face (a mask under predicted normals, or 'n' alone under "
            "the geometric normals): several times slower than the fused phases")
        ph = _loop.BrdfPhase(scene, gt, mat["albedo"], mat["roughness"], mat["metallic"], None if scene.use_mesh_normal else mat["normal"],
                             optimize_part=part, spp=spp, scale_delta=scale_delta, saver=_loop.DeviceSaveBest(), mask=mask,
                             originals=originals)
        if saver.best_loss is not None:
            ph.saver.best_loss = saver.best_loss.clone().reshape(())
        es = _loop.EarlyStopping(patience, min_delta)
        stop, it = "num_epochs", 0
        for it in range(n_epochs):
            es(float(ph.step()))
            if es.early_stop:
                stop = "early_stop"
                break
        new_best = ph.saver.best_loss.reshape(-1)
        prev = saver.best_loss if saver.best_loss is not None else torch.full_like(new_best, float("inf"))
        if bool((new_best < prev).any()) and "albedo" in ph.saver.best:
            saver.best_loss = torch.minimum(new_best, prev)
            for k_ in ("albedo", "roughness", "metallic", "rendered_img"):
                saver.best[k_] = ph.saver.best[k_].clone()
            if "normal" in ph.saver.best:
                saver.best["normal"] = ph.saver.best["normal"].clone()
                mat["normal"] = saver.best["normal"]
            saver.best["envmap"] = state["envmap4render"].clone()
        say(f"loop {loop_num}: part {part!r} (with normals) ran {it + 1} iterations ({stop})")
        return it, ph.opt.param_groups[0]["lr"], stop

    def brdf_part_runner_mlp_normal(loop_num: int, part: str, patience: int, min_delta: float, n_epochs: int):
        """pos_mlp with output_type 'armn' (:165-172,493-506): the net predicts the normal map as well."""
        from .armhead import ArmMlpPhase

        fixed_keys = ("albedo", "roughness", "metallic") + (() if scene.use_mesh_normal else ("normal",))
        ph = _loop.PosMlpNormalPhase(scene, gt, brdf_net, start_arm, {k: mat[k] for k in fixed_keys},
                                     optimize_part=part, spp=spp, scale_delta=scale_delta, saver=_loop.DeviceSaveBest(), mask=mask)
        if ph.engine is not None:
            say(f"loop {loop_num}: part {part!r} (pos_mlp, armn) runs launch by launch on the C ABI (PosMlpNormalPhase with armhead.MlpEngine: render, losses, "
                "the network's layer products and AdamW; no autograd)")
        else:
            say(f"loop {loop_num}: part {part!r} (pos_mlp) runs PosMlpNormalPhase with the network under autograd (render, losses and layer products on the C "
                f"ABI), not a launch-by-launch phase: {ArmMlpPhase.why_not(scene, gt, brdf_net, part, mask)}")
        if saver.best_loss is not None:
            ph.saver.best_loss = saver.best_loss.clone().reshape(())
        es = _loop.EarlyStopping(patience, min_delta)
        stop, it = "num_epochs", 0
        for it in range(n_epochs):
            es(float(ph.step()))                                                    # per-epoch host check, as the reference (:550)
            if es.early_stop:
                stop = "early_stop"
                break
        new_best = ph.saver.best_loss.reshape(-1)
        prev = saver.best_loss if saver.best_loss is not None else torch.full_like(new_best, float("inf"))
        if bool((new_best < prev).any()) and "albedo" in ph.saver.best:
            saver.best_loss = torch.minimum(new_best, prev)
            for k_ in ("albedo", "roughness", "metallic", "rendered_img"):
                saver.best[k_] = ph.saver.best[k_].clone()
            if "normal" in ph.saver.best:
                saver.best["normal"] = ph.saver.best["normal"].clone()
                mat["normal"] = saver.best["normal"]
            saver.best["envmap"] = state["envmap4render"].clone()
            state["best_brdf_weights"] = {k: v.clone() for k, v in ph.best_weights.items()}
        if state["best_brdf_weights"] is not None:
            brdf_net.load_state_dict(state["best_brdf_weights"])                    # :586-587: reloaded after every part
        say(f"loop {loop_num}: part {part!r} (pos_mlp, armn) ran {it + 1} iterations ({stop})")
        return it, ph.opt.param_groups[0]["lr"], stop

    def brdf_part_runner(loop_num: int, part: str, patience: int, min_delta: float, n_epochs: int):
        if model_name == "pos_mlp":
            from .armhead import ArmMlpPhase

            # predicted normals, or pixels without geometry on an image the launch-by-launch phase does not take: the operator face
            if not scene.use_mesh_normal or (background and not ArmMlpPhase.supported(scene, gt, brdf_net, part, mask)):
                return brdf_part_runner_mlp_normal(loop_num, part, patience, min_delta, n_epochs)
        if model_name == "pos_mlp":
            return brdf_part_runner_mlp(loop_num, part, patience, min_delta, n_epochs)
        # Under a FIXED predicted normal map (use_mesh_normal False, no 'n' in the part) the fused phases shade with it as they do with the
        # geometric normals; a part that MOVES the normal map runs NormalBrdfPhase (launch by launch on the C ABI, device-side SaveBest /
        # EarlyStopping); under the geometric normals an 'n' in the part optimises nothing (:356,376); what is left -- masks with predicted
        # normals, a part that is 'n' alone under the geometric normals -- is the autograd composition's
        eff = part.replace("n", "") if scene.use_mesh_normal else part
        moves_n = "n" in eff
        if not eff or (mask is not None and not scene.use_mesh_normal) or (moves_n and not gt.is_cuda):
            return brdf_part_runner_normal(loop_num, part, patience, min_delta, n_epochs)
        phase_kw = dict(optimize_part=eff, spp=spp, scale_delta=scale_delta, patience=patience, min_delta=min_delta,
                        best_mse=saver.best_loss if saver.best_loss is not None else None, history_len=n_epochs, originals=originals)
        if moves_n:
            ph = _loop.NormalBrdfPhase(scene, gt, mat["albedo"], mat["roughness"], mat["metallic"], mat["normal"], **phase_kw)
        elif mask is not None and gt.ndim == 4:    # a batch under --use_mask: its images alone, each on a stream of its own
            ph = _loop.MaskedBatchPhase(scene, gt, mat["albedo"], mat["roughness"], mat["metallic"], mask, **phase_kw)
        elif mask is not None:    # --use_mask: launch by launch (two image-wide means per iteration), same device-side SaveBest / EarlyStopping
            ph = _loop.MaskedBrdfPhase(scene, gt, mat["albedo"], mat["roughness"], mat["metallic"], mask, **phase_kw)
        elif gt.ndim == 4 and gt.shape[0] >= 8 and gt.shape[0] % 2 == 0 and gt.is_cuda:
            # a shard of images: two groups stepping on streams of their own (the same results, bit for bit; one group's walk and statistics
            # launches run under the other's streaming step)
            ph = _loop.PipelinedBrdfPhase(scene, gt, mat["albedo"], mat["roughness"], mat["metallic"], groups=2, **phase_kw)
        else:
            ph = _loop.FusedBrdfPhase(scene, gt, mat["albedo"], mat["roughness"], mat["metallic"], **phase_kw)
        done, stop = 0, "num_epochs"
        while done < n_epochs:
            k = min(sync_every, n_epochs - done)
            ph.run(k)
            done += k
            info = ph.poll()
            if frames is not None and gt.ndim == 3 and frames.due("mat"):
                shown = ph.pred                                # lazy loop: the render of the current parameters (the next iteration's)
                frames.mat_frame(loop_num, part, done - 1, gt, _loss.linear_to_srgb((shown * (gt.mean() / shown.mean())).clamp_min(1e-8)),   # its own exposure ratio (:388)
                                 ph.current_maps(), ph.current_maps()["normal"] if moves_n else
                                 (shading_normal if shading_normal is not None else scene.shading_normal()))
            if bool(info["stopped"].all()):
                stop = "early_stop"
                break
        info = ph.poll()
        iters = int(info["iters"].max())
        improved = info["best_mse"].to(dev) < (saver.best_loss if saver.best_loss is not None else float("inf"))
        if bool(improved.any()):                                                    # SaveBest is global across phases (F11)
            saver.best_loss = torch.minimum(info["best_mse"].to(dev).reshape(saver.best_loss.shape), saver.best_loss) \
                if saver.best_loss is not None else info["best_mse"].to(dev)
            for k_, v in (("albedo", ph.best["albedo"]), ("roughness", ph.best["roughness"]), ("metallic", ph.best["metallic"]),
                          ("rendered_img", ph.best_img)):
                keep_best(k_, v, improved)
            env4 = state["envmap4render"]
            if gt.ndim == 4 and env4.ndim == 3:
                env4 = env4.unsqueeze(0).expand((gt.shape[0],) + tuple(env4.shape))
            keep_best("envmap", env4.contiguous(), improved)
            if moves_n:                                                          # SaveBest keeps the normal map it rendered with (:421-422)
                keep_best("normal", ph.best["normal"], improved)
                mat["normal"] = saver.best["normal"]
            elif not scene.use_mesh_normal and "normal" in mat:
                keep_best("normal", mat["normal"].detach(), improved)
        say(f"loop {loop_num}: part {part!r}{' (normal map, on the device)' if moves_n else ''} ran {iters} iterations ({stop}), best mse {float(info['best_mse'].min()):.5f}")
        return iters - 1, ph.lr_at(max(iters - 1, 0)), stop

    def on_brdf_part_end(loop_num: int, part: str) -> None:                        # :460-463: every map comes back from the saver
        for key in ("albedo", "roughness", "metallic"):
            mat[key] = saver.best[key].detach().clone()
        params["shape.bsdf.a"], params["shape.bsdf.r"], params["shape.bsdf.m"] = mat["albedo"], mat["roughness"], mat["metallic"]
        if not scene.use_mesh_normal and "normal" in saver.best:
            mat["normal"] = saver.best["normal"].detach().clone()
            params["shape.bsdf.n"] = mat["normal"]
        _save_results()                                                             # :465,590
        stage_digest(f"loop {loop_num} brdf {part}", mat["albedo"], mat["roughness"], mat["metallic"])

    trace: List[TraceEvent] = run_schedule(list(optimize_order), None, None, opt_src=opt_src, opt_env_from=opt_env_from,
                                           num_epochs=num_epochs, on_env_phase_end=on_env_phase_end,
                                           on_brdf_phase_begin=on_brdf_phase_begin, on_brdf_part_end=on_brdf_part_end,
                                           brdf_part_runner=brdf_part_runner, env_phase_runner=env_phase_runner)
    with torch.no_grad():
        params["emitter.data"] = saver.best["envmap"]
        final = _render.render_w_brdf(scene, saver.best["albedo"], saver.best["roughness"], saver.best["metallic"],
                                      None if scene.use_mesh_normal else mat["normal"], spp)
        red = (-3, -2, -1) if gt.ndim == 4 else None                                # per image for a batch
        ratio = gt.mean(dim=red, keepdim=True) / final.mean(dim=red, keepdim=True) if red else gt.mean() / final.mean()
        psnr_each = _loss.psnr(final * ratio, gt).reshape(-1)
    stage_digest("final", saver.best["albedo"], saver.best["roughness"], saver.best["metallic"], saver.best["envmap"], final)
    return {"albedo": saver.best["albedo"], "roughness": saver.best["roughness"], "metallic": saver.best["metallic"], "normal": mat.get("normal"),
            "envmap": saver.best["envmap"], "rendered_img": saver.best["rendered_img"], "final_render": final,
            "psnr": float(psnr_each.mean()), "psnr_per_image": psnr_each.tolist(), "best_loss": float(saver.best_loss.min()),
            "best_loss_per_image": saver.best_loss.reshape(-1).tolist(), "trace": trace}
