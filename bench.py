#!/usr/bin/env python3
"""Headline benchmark: optimisation iterations/s of the PBR shading hot path at 512x512.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one optimisation iteration (epoch) of hot loop B of BASELINE config 2 (`--model_name pos_mlp --opt_order 'rm a'`,
inverse_img_w_mi.py:470-590): material maps from the residual PosMLP (f32-MFMA sine-layer kernels of libmatpbr.so) -> shade_fwd -> gamma-2.2 MSE/L1
loss with mean-ratio scaling -> shade_bwd -> AdamW, everything downstream of the maps in libmatpbr.so, inputs resident in
HBM.  `--mode fused` times the same loop in `--model_name none` mode (maps optimised directly, whole iteration in
libmatpbr.so); both rates are reported in every run (`modes`).  Images are independent, so ranks never communicate inside
the timed region (weak scaling: `images_per_gpu` per rank).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_FWD = 44             # read a 12 + r 4 + m 4 + n 12, write rgb 12
BYTES_BWD_ARM = 64         # read a,r,m,n 32 + d_rgb 12, write d_a 12 + d_r 4 + d_m 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel; the reference renders with spp=64 (inverse_img_w_mi.py:625)")
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--mode", choices=["fused", "torch", "pos_mlp"], default="pos_mlp",
                    help="fused: --model_name none, whole iteration in libmatpbr.so; torch: same step composed from torch ops; "
                         "pos_mlp: the reference's default mode (maps from the residual PosMLP, its sine layers + render/loss/backward in libmatpbr.so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-relight", action="store_true", help="skip the 2048x2048 relighting measurement (1.3 GB transfer buffer)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(size, spp, target_s):
    """Time the fp32 OpenMP build of the CPU oracle (a port of the reference BRDF path, not Mitsuba) on a bounded
    crop of the same synthetic workload: shade fwd + bwd (materials) of an n x n crop; it/s scaled by pixel count."""
    import numpy as np

    from materialist_amd import synthetic
    from oracle.oracle import Oracle

    o = Oracle(np.float32)
    cores = len(os.sched_getaffinity(0))
    sc = synthetic.make_scene(0, size, size)
    n_full = Oracle(np.float64).normals_from_depth(sc.depth.astype(np.float64)).astype(np.float32)

    def run(n):
        sl = (slice(0, n), slice(0, n))
        a, r, m, nn = sc.albedo[sl], sc.roughness[sl], sc.metallic[sl], n_full[sl]
        t0 = time.perf_counter()
        out = o.shade_fwd(a, r, m, nn, sc.light, spp)
        o.shade_bwd(a, r, m, nn, sc.light, np.ones_like(out), spp, want_n=False, want_light=False)
        return time.perf_counter() - t0

    t_probe = run(32)
    per_px = t_probe / (32 * 32)
    n = int(min(size, max(32, (target_s / per_px) ** 0.5)))
    n -= n % 8
    reps, t = 0, 0.0
    while t < target_s and reps < 64:      # small crops finish early on many-core hosts: repeat up to the time budget
        t += run(n)
        reps += 1
    its = reps * (n * n) / (size * size) / t
    return {"value": its, "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"oracle f32+OpenMP shade fwd+bwd(arm), {reps} x ({n}x{n} crop of the {size}x{size} spp={spp} image), {t:.1f}s, scaled by pixels; "
                      "CPU restatement of the reference BRDF path (not Mitsuba)"}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from materialist_amd import loop, ops, render, synthetic
    from materialist_amd.dist import shard_range

    H = W = args.size
    B = args.images_per_gpu
    lo, hi = shard_range(world * B, world, rank)     # contiguous shard of independent images
    scenes = [synthetic.make_scene(i, H, W) for i in range(lo, hi)]
    t = lambda xs: torch.from_numpy(np.stack(xs) if B > 1 else xs[0]).to(dev)
    depth = t([s.depth for s in scenes])
    gt_a, gt_r, gt_m = t([s.albedo for s in scenes]), t([s.roughness for s in scenes]), t([s.metallic for s in scenes])
    light = t([s.light for s in scenes])
    scene = render.load_estimated_mesh(depth, use_mesh_normal=True)
    scene._set("emitter.data", light)                # BRDF phase renders under the current best light (:317-334)
    with torch.no_grad():
        gt_image = render.render_w_brdf(scene, gt_a, gt_r, gt_m, None, args.spp)
    init = (t([s.init_albedo for s in scenes]), t([s.init_roughness for s in scenes]), t([s.init_metallic for s in scenes]))
    if args.mode == "pos_mlp" and B > 1:
        args.mode = "fused"                      # the MLP modes optimise one image per process; a batch runs the none-mode loop

    def make_phase(mode):
        if mode == "fused":
            return loop.FusedBrdfPhase(scene, gt_image, *init, optimize_part="rm", spp=args.spp)
        if mode == "pos_mlp":
            from materialist_amd import posmlp

            net = posmlp.brdf_net("arm").to(dev)
            start_arm = torch.cat([init[0].reshape(-1, 3), init[1].reshape(-1, 1), init[2].reshape(-1, 1)], -1).clamp(0, 1)
            ph = loop.PosMlpBrdfPhase(scene, gt_image, net, start_arm, {"albedo": init[0], "roughness": init[1], "metallic": init[2]},
                                      optimize_part="rm", spp=args.spp)
            ph.current_maps = lambda: (lambda m: {"albedo": m["albedo"].detach().clamp(0, 1), "roughness": m["roughness"].detach().clamp(0.07, 1),
                                                  "metallic": m["metallic"].detach().clamp(0, 1)})(ph.maps_from_net()[0])
            return ph
        if mode == "env":
            from materialist_amd import posmlp

            s_env = render.load_estimated_mesh(depth, use_mesh_normal=True)
            pr = render.traverse(s_env)
            pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = gt_a, gt_r, gt_m
            enet = posmlp.envmap_net().to(dev)
            ones = torch.ones(512, 3, device=dev)
            return loop.FusedEnvPhase(s_env, gt_image, lambda: enet(ones).reshape(16, 32, 3), loop.capturable_adam(enet.parameters(), 1e-3),
                                      spp=args.spp, use_graph=True)     # whole iteration replayed from a hipGraph after 3 eager ones
        return loop.BrdfPhase(scene, gt_image, *init, None, optimize_part="rm", spp=args.spp)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(ph, warmup, steps):
        for _ in range(warmup):
            ph.step()
        fence()
        t0 = time.perf_counter()
        with ops.KernelTimer() as kt:
            for _ in range(steps):
                ph.step()
            fence()
            el_ = time.perf_counter() - t0
            ks = kt.summary()
        el = torch.tensor([el_], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), ks

    phase = make_phase(args.mode)
    psnr0 = float(loop._loss.psnr(render.render_w_brdf(scene, *[phase.current_maps()[k].detach() for k in ("albedo", "roughness", "metallic")], None, args.spp), gt_image).mean())
    elapsed, ksum = timed(phase, args.warmup, args.steps)
    total_units = args.steps * B * world
    value = total_units / elapsed

    m = phase.current_maps()
    with torch.no_grad():
        final = render.render_w_brdf(scene, m["albedo"].detach(), m["roughness"].detach(), m["metallic"].detach(), None, args.spp)
    psnr1 = float(loop._loss.psnr(final, gt_image).mean())

    # the other loops of the same pipeline, short runs, for the record (same fences; whole-job rates)
    modes = {args.mode: {"it_per_s": value, "ms_per_step": elapsed / args.steps * 1e3}}
    for extra in ("fused", "pos_mlp", "env"):
        if extra == args.mode or (extra != "fused" and B > 1) or args.mode == "torch":
            continue
        e_el, _ = timed(make_phase(extra), 5, 40)
        modes[extra] = {"it_per_s": 40 * B * world / e_el, "ms_per_step": e_el / 40 * 1e3}
    mode_names = {"fused": "hot loop B, --model_name none (whole iteration in libmatpbr.so)",
                  "pos_mlp": "hot loop B, --model_name pos_mlp (PosMLP sine layers, render, loss, backward in libmatpbr.so; autograd glue in torch)",
                  "env": "hot loop A, envmap PosMLP head + matpbr_env_phase_step", "torch": "hot loop B composed from torch ops"}
    modes = {k: dict(v, what=mode_names[k]) for k, v in modes.items()}

    # kernel durations for the roofline: 20 back-to-back launches between two HIP events on the launch stream (per-launch
    # event pairs inside the loop also time the inter-launch gap; they are reported as *_inloop_ms for reference)
    def back_to_back(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    mm = {k: v.detach().contiguous() for k, v in phase.current_maps().items() if v is not None}
    nrm, lgt = scene.shading_normal(), (scene.light if scene.light.ndim == 3 or B == 1 else scene.light.unsqueeze(0).expand(B, -1, -1).contiguous())
    d_probe = torch.randn_like(gt_image)
    ms_f = back_to_back(lambda: ops.shade_fwd(mm["albedo"], mm["roughness"], mm["metallic"], nrm, lgt, args.spp))
    ms_b = back_to_back(lambda: ops.shade_bwd(mm["albedo"], mm["roughness"], mm["metallic"], nrm, lgt, d_probe, args.spp, want_mat=True))

    # the PosMLP side of the pos_mlp iteration: nine [H*W,256]x[256,256] products per iteration on the exact-f32 MFMA, in the
    # hand-written kernels of libmatpbr.so (epilogues included); the BLAS product of the same shape is timed beside them
    gemm = None
    if B == 1:
        Mg = H * W
        xg, gg = torch.randn(Mg, 256, device=dev), torch.randn(Mg, 256, device=dev)
        wg, bg = torch.randn(256, 256, device=dev) / 16, torch.randn(256, device=dev)
        sg, cg, gp = (torch.empty(Mg, 256, device=dev) for _ in range(3))
        dbg = torch.empty(256, device=dev)
        flop = 2.0 * Mg * 256 * 256
        ms_fwd = back_to_back(lambda: ops.mlp_layer_fwd(xg, wg, bg, sg, cg, 256))
        ms_din = back_to_back(lambda: ops.mlp_layer_bwd_input(gg, wg, cg, gp, 256, 256, dbg))
        ms_dw = back_to_back(lambda: ops.mlp_layer_bwd_weight(gg, xg, 256, 256))
        ms_g = back_to_back(lambda: torch.mm(xg, wg))
        tf = lambda ms: flop / (ms * 1e-3) / 1e12
        gemm = {"bound": "mfma", "kernel": "mlp_gemm_nt_pipe<sincos> / <mul cos> / mlp_wgrad_tn: [H*W,256]x[256,256] on v_mfma_f32_32x32x2_f32",
                "achieved": 3 * flop / ((ms_fwd + ms_din + ms_dw) * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                "forward_sincos": {"avg_launch_ms": ms_fwd, "achieved": tf(ms_fwd)},
                "bwd_input_mulcos_colsum": {"avg_launch_ms": ms_din, "achieved": tf(ms_din)},
                "bwd_weight": {"avg_launch_ms": ms_dw, "achieved": tf(ms_dw)},
                "blas_product_same_shape": {"avg_launch_ms": ms_g, "achieved": tf(ms_g), "kernel": "hipBLASLt f32 (PyTorch-ROCm), no epilogue"}}
        gemm["frac"] = gemm["achieved"] / gemm["peak"]
        del xg, gg, wg, sg, cg, gp

    # BASELINE configs[4]: forward-only relighting, 2048x2048, 360 lights, through the precomputed transfer (HBM-bound kernel)
    relight = None
    if B == 1 and not args.no_relight:
        RS = 2048
        scr = synthetic.make_scene(lo, RS, RS)
        tr = lambda x: torch.from_numpy(x).to(dev)
        n_r = ops.normals_from_depth(tr(scr.depth))
        T_r = ops.shade_transfer(tr(scr.albedo), tr(scr.roughness), tr(scr.metallic), n_r, args.spp)
        L_r = torch.randn(360, 25, 3, device=dev) * 0.1
        L_r[:, 0] += 3.5
        out_r = torch.empty(8, RS, RS, 3, device=dev)
        ops.relight(T_r, L_r[:8].contiguous(), RS, RS, out_r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f0 in range(0, 360, 8):
            ops.relight(T_r, L_r[f0:f0 + 8], RS, RS, out_r)
        e1.record()
        torch.cuda.synchronize()
        ms_r = e0.elapsed_time(e1)
        bytes_r = 45 * (300 + 8 * 12) * RS * RS       # per 8-light pass: transfer read once (300 B/px) + 8 rgb writes
        relight = {"bound": "hbm", "kernel": "relight_kernel (2048x2048, 8 lights per pass)", "achieved": bytes_r / (ms_r * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                   "unit": "GB/s", "frames_per_s": 360 / (ms_r * 1e-3), "avg_launch_ms": ms_r / 45, "bytes_per_pixel_per_pass": 396}
        relight["frac"] = relight["achieved"] / relight["peak"]
        del T_r, out_r, L_r

    if rank == 0:
        px = H * W * B
        ms_f_in = ksum.get("shade_fwd", (0, None))[1]      # None (JSON null) when the mode issues no stand-alone launch of that kernel
        ms_b_in = ksum.get("shade_bwd", (0, None))[1]
        ach_b = BYTES_BWD_ARM * px / (ms_b * 1e-3) / 1e9
        ach_f = BYTES_FWD * px / (ms_f * 1e-3) / 1e9
        ach_fb = (BYTES_FWD + BYTES_BWD_ARM) * px / ((ms_f + ms_b) * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"shade_bwd_{H}x{W}_b{B}_spp{args.spp}")
            except Exception:
                traffic = None
        # VALU issue model: lane-instructions per pixel-sample counted from the ISA
        out = {
            "metric": "opt_iterations_per_sec_512x512", "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2-synthetic (BASELINE configs[1]): {H}x{W}, one epoch of hot loop B, part 'rm' of --opt_order 'rm a', "
                                   f"{'--model_name pos_mlp' if args.mode == 'pos_mlp' else '--model_name none'} "
                                   f"(maps -> shade_fwd -> gamma-2.2 MSE/L1 loss -> shade_bwd -> Adam(W)), spp={args.spp}, geometric normals, SH25 light",
                       "mode": args.mode, "height": H, "width": W, "spp": args.spp, "images_per_gpu": B, "light": "SH25"},
            "modes": modes,
            "roofline": {"bound": "hbm", "kernel": "shade_bwd_kernel<mat>", "achieved": ach_b, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach_b / (HBM_PEAK / 1e9), "traffic": traffic,
                         "bytes_per_pixel": BYTES_BWD_ARM, "avg_launch_ms": ms_b, "avg_launch_inloop_ms": ms_b_in,
                         "note": "canonical unfused shade_bwd<mat> on the same maps; the kernel is VALU-issue-bound at spp=64 (DESIGN.md section 4)",
                         "shade_fwd": {"achieved": ach_f, "frac": ach_f / (HBM_PEAK / 1e9), "avg_launch_ms": ms_f, "avg_launch_inloop_ms": ms_f_in,
                                       "bytes_per_pixel": BYTES_FWD},
                         "fwd+bwd": {"achieved": ach_fb, "frac": ach_fb / (HBM_PEAK / 1e9), "bytes_per_pixel": BYTES_FWD + BYTES_BWD_ARM},
                         "posmlp_gemm": gemm, "relight": relight},
            "psnr_db": {"initial_guess": psnr0, "after_timed_steps": psnr1, "vs": "own HIP render of the synthetic ground truth (Mitsuba cannot run, SURVEY F3)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H, args.spp, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
