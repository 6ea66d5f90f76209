#!/usr/bin/env python3
"""Headline benchmark: optimisation iterations/s of the PBR shading hot path at 512x512.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment, this process only LAUNCHES the ranks (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`, one rank per GPU over RCCL) and exits with their code; it never
touches the GPU itself.  Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

One "step" = one optimisation iteration (epoch) of hot loop B of BASELINE configs[1] (`--model_name pos_mlp --opt_order 'rm a'`,
inverse_img_w_mi.py:470-590): material maps from the residual PosMLP (f32-MFMA sine-layer kernels of libmatpbr.so) -> render
(GGX-lobe samples; diffuse-lobe coefficients cached for the phase) -> gamma-2.2 MSE/L1 loss with mean-ratio scaling ->
streaming backward -> AdamW, everything downstream of the maps in libmatpbr.so, inputs resident in HBM.  `modes` reports the
other loops of the pipeline measured in the same run: `fused` (`--model_name none`, whole iteration in libmatpbr.so), `fused_b8`
(BASELINE configs[2]'s per-GPU shard: 8 images x 512x512 in the kernels' batch dimension), `env` (hot loop A on the radiance
transfer).  Images are independent, so ranks never communicate inside the timed region (weak scaling: `images_per_gpu` per rank).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_FWD = 44             # read a 12 + r 4 + m 4 + n 12, write rgb 12
BYTES_BWD_ARM = 64         # read a,r,m,n 32 + d_rgb 12, write d_a 12 + d_r 4 + d_m 4
BYTES_ENV = 312            # hot loop A on the transfer: read T 300 + gt 12 per pixel (pred not written)

LINE_LIMIT = 4096          # the driver parses the LAST stdout line; round 5's 20 KB line did not parse
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "world_size", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_CONFIG_KEYS = ("workload", "mode", "height", "width", "spp", "images_per_gpu", "light", "parallelism")
_ROOF_KEYS = ("bound", "kernel", "workload", "achieved", "peak", "unit", "frac", "avg_launch_ms", "bytes_per_pixel", "traffic", "own_traffic_frac",
              "grad_rel_l2", "grad_p999_pixel", "grad_worst_pixel", "iteration_frac")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _short(v, n=120):
    if isinstance(v, str):
        return v if len(v) <= n else v[: n - 3] + "..."
    if isinstance(v, float):
        return float(f"{v:.6g}")
    return v


def compact_line(out: dict) -> str:
    """The ONE line the driver parses: the contract's keys, numbers and short names only (<= LINE_LIMIT bytes).  Everything else of `out`
    (modes, per-kernel legs, notes, host_enqueue ...) goes to bench_detail.json (write_detail)."""
    line = {k: _short(out[k]) for k in _TOP_KEYS if k in out}
    line["config"] = {k: _short(out["config"][k]) for k in _CONFIG_KEYS if k in out.get("config", {})}
    if out.get("roofline"):
        line["roofline"] = {k: _short(out["roofline"][k], 100) for k in _ROOF_KEYS if k in out["roofline"]}
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = {k: _short(out["cpu_baseline"][k], 160) for k in _CPU_KEYS if k in out["cpu_baseline"]}
    if out.get("modes"):
        line["modes_it_per_s"] = {k: _short(float(v["it_per_s"])) for k, v in out["modes"].items()}
    if out.get("psnr_db"):
        line["psnr_db"] = {k: _short(v) for k, v in out["psnr_db"].items() if not isinstance(v, str)}
    line["detail"] = "bench_detail.json"
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                      # never print a line the driver cannot parse: drop the optional blocks first
        for k in ("modes_it_per_s", "psnr_db"):
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def write_detail(out: dict) -> None:
    """Everything the run measured, for people: bench_detail.json beside bench.py (and under gpurun_out/ when that exists, so that it travels back)."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel; the reference renders with spp=64 (inverse_img_w_mi.py:625)")
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--mode", choices=["fused", "fused_one_phase", "fused_arm", "fused_a", "torch", "pos_mlp"], default=None,
                    help="default: pos_mlp (the reference's default mode: maps from the residual PosMLP) for one image per GPU, fused "
                         "(--model_name none, whole iteration in libmatpbr.so) for a batch; torch: the none-mode step composed from torch ops")
    ap.add_argument("--mlp-products", type=int, choices=[0, 6, 9], default=None,
                    help="partial products per f32 product of the 256-wide PosMLP layers: 6 (default) / 9 = split-operand kernels on the bf16 "
                         "matrix pipe (f32-accurate: three bf16 pieces per operand, f32 accumulate), 0 = the exact-f32 MFMA kernels")
    ap.add_argument("--mlp-lds-dma", type=int, choices=[0, 1, 2, 3], default=None,
                    help="main loop of the split-operand layer kernels: 2 (default) operands by LDS-DMA, two 256-thread workgroups per CU; "
                         "1 LDS-DMA, one 512-thread workgroup per CU; 0 register-staged (same bits; A/B switch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-relight", action="store_true", help="skip the 2048x2048 relighting measurement (1.3 GB transfer buffer)")
    ap.add_argument("--no-extras", action="store_true", help="headline mode only (no other loops, no kernel roofline legs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even at world size 1 and run the barrier / all_gather of the timing protocol: "
                         "what the N-rank runs do, provable on one GPU (tests/test_gpu_configs.py)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="exercise the launcher and the timing protocol with a CPU stand-in step over gloo (tests/test_bench_launcher.py)")
    return ap.parse_args(argv)


def launch_ranks(n: int, argv) -> int:
    """Start one rank per GPU under torch.distributed.run as a CHILD process (never exec: this process may not touch the GPU, and
    a process that did must not be replaced) and return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def cpu_baseline(size, spp, target_s):
    """The CPU side of the comparison, on a bounded crop of the same synthetic workload, scaled by pixel count.
      value        the SAME iteration as the headline (--model_name pos_mlp, part 'rm'): residual PosMLP forward + backward (torch CPU,
                   fp32, the module of materialist_amd/posmlp.py = the reference's mymodels/mlps.py network) + the reference-literal
                   spp-sample estimator (sample_brdf's 50/50 lobe choice with the mixture pdf, mi_plugin.py:1296-1341, oracle kind=1)
                   forward + material backward, fp32 + OpenMP;
      shade_only   round 2's leg: shade forward + backward alone with the production estimator (what the HIP kernels compute).
    Neither is Mitsuba (it cannot run here): kind = "port"."""
    import numpy as np
    import torch

    from materialist_amd import posmlp, synthetic
    from oracle.oracle import Oracle

    o = Oracle(np.float32)
    cores = len(os.sched_getaffinity(0))
    sc = synthetic.make_scene(0, size, size)
    n_full = Oracle(np.float64).normals_from_depth(sc.depth.astype(np.float64)).astype(np.float32)

    def shade(n, kind):
        sl = (slice(0, n), slice(0, n))
        a, r, m, nn = sc.albedo[sl], sc.roughness[sl], sc.metallic[sl], n_full[sl]
        t0 = time.perf_counter()
        out = o.shade_fwd(a, r, m, nn, sc.light, spp, kind=kind)
        o.shade_bwd(a, r, m, nn, sc.light, np.ones_like(out), spp, want_n=False, want_light=False, kind=kind)
        return time.perf_counter() - t0

    def timed(fn, budget):
        t_probe = fn(32)
        n = int(min(size, max(32, (budget / (t_probe / 1024.0)) ** 0.5)))
        n -= n % 8
        reps, t = 0, 0.0
        while t < budget and reps < 64:      # small crops finish early on many-core hosts: repeat up to the time budget
            t += fn(n)
            reps += 1
        return reps * (n * n) / (size * size) / t, n, reps, t

    its_shade, n_s, reps_s, t_s = timed(lambda n: shade(n, 0), 0.3 * target_s)
    its_lit, n_l, reps_l, t_l = timed(lambda n: shade(n, 1), 0.3 * target_s)
    threads = min(cores, 32)                 # torch's CPU GEMMs stop scaling (and start thrashing) far below 256 threads on these shapes
    torch.set_num_threads(threads)
    net = posmlp.brdf_net("arm")

    def mlp(n):
        start = torch.rand(n * n, 5)
        t0 = time.perf_counter()
        net.zero_grad(set_to_none=True)
        y = net(start)
        y.sum().backward()
        return time.perf_counter() - t0

    mlp(64)                                  # first call: thread pool / oneDNN set-up
    its_mlp, n_m, reps_m, t_m = timed(mlp, 0.4 * target_s)
    its = 1.0 / (1.0 / its_mlp + 1.0 / its_lit)
    return {"value": its, "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"the headline's iteration on the CPU, scaled by pixels from crops of the {size}x{size} image: PosMLP 'arm' forward + backward "
                      f"(torch CPU fp32, {threads} threads; {reps_m} x {n_m}x{n_m} points, {t_m:.1f}s -> {its_mlp:.3g} it/s) + reference-literal spp={spp} "
                      f"estimator (oracle kind=1, f32 + OpenMP on {cores} cores) forward + material backward ({reps_l} x {n_l}x{n_l}, {t_l:.1f}s -> "
                      f"{its_lit:.3g} it/s); not Mitsuba",
            "shade_only": {"value": its_shade, "unit": "it/s",
                           "sample": f"oracle f32+OpenMP shade fwd+bwd(arm) with the production estimator, {reps_s} x ({n_s}x{n_s} crop), {t_s:.1f}s, "
                                     "scaled by pixels (round 2's cpu_baseline)"}}


class _Protocol:
    """The timing contract: W untimed steps, then EXACTLY K steps between barrier + synchronize fences, MAX over ranks."""

    def __init__(self, dist, world, device, sync, collectives=None):
        self.dist, self.world, self.device, self.sync = dist, world, device, sync
        self.collectives = world > 1 if collectives is None else bool(collectives)     # --force-dist: also at world size 1

    def fence(self):
        self.sync()
        if self.collectives:
            self.dist.barrier()
        self.sync()

    def timed(self, step, warmup, steps):
        import torch

        for _ in range(warmup):
            step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        mine = time.perf_counter() - t0
        el = torch.tensor([mine], dtype=torch.float64, device=self.device)
        per_rank = [mine]
        if self.collectives:
            bufs = [torch.zeros_like(el) for _ in range(self.world)]
            self.dist.all_gather(bufs, el)
            per_rank = [float(b.item()) for b in bufs]
        return max(per_rank), per_rank


def selftest_cpu(args):
    """The launcher + protocol on CPU (gloo): a numpy stand-in for the step.  Not a benchmark."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x = np.random.default_rng(rank).random((64, 64))
    proto = _Protocol(dist, world, torch.device("cpu"), lambda: None)
    elapsed, per_rank = proto.timed(lambda: x @ x, args.warmup, args.steps)
    if rank == 0:
        print(json.dumps({"metric": "selftest_cpu_steps_per_sec", "value": args.steps * world / elapsed, "unit": "it/s", "n_gpus": world,
                          "world_size": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
                          "ranks": [{"rank": r, "it_per_s": args.steps / t} for r, t in enumerate(per_rank)]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))          # before anything touches the GPU in this process
    if args.selftest_cpu:
        return selftest_cpu(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from materialist_amd.dist import pin_rank_cores

    pinned = pin_rank_cores()          # the ranks of a node on disjoint core sets (before any thread pool starts)
    # the CPU baseline runs BEFORE the GPU legs (rank 0, N = 1 only): the GPU work then sits at the end of the run
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args.size, args.spp, args.cpu_seconds)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from materialist_amd import loop, ops, render, synthetic
    from materialist_amd.dist import shard_range

    from materialist_amd import posmlp as _posmlp

    if args.mlp_products is not None:
        _posmlp._PosMlpHipFn.PRODUCTS = args.mlp_products
    if args.mlp_lds_dma is not None:
        from materialist_amd import _lib as _l
        _l.load().matpbr_mlp_set_lds_dma(args.mlp_lds_dma)
    H = W = args.size
    proto = _Protocol(dist, world, dev, torch.cuda.synchronize, collectives=use_dist)
    mode = args.mode or ("pos_mlp" if args.images_per_gpu == 1 else "fused")
    if mode == "pos_mlp" and args.images_per_gpu > 1:
        raise SystemExit("--mode pos_mlp optimises one image per process (use --mode fused for --images-per-gpu > 1)")

    class Workload:
        """B independent synthetic images of this rank's shard, resident in HBM, and the phases that iterate on them."""

        def __init__(self, B, size=None):
            self.B = B
            lo, hi = shard_range(world * B, world, rank)     # contiguous shard of independent images
            self.lo = lo
            scenes = [synthetic.make_scene(i, size or H, size or W) for i in range(lo, hi)]
            t = lambda xs: torch.from_numpy(np.stack(xs) if B > 1 else xs[0]).to(dev)
            self.depth = t([s.depth for s in scenes])
            self.gt = tuple(t([getattr(s, k) for s in scenes]) for k in ("albedo", "roughness", "metallic"))
            self.light = t([s.light for s in scenes])
            self.scene = render.load_estimated_mesh(self.depth, use_mesh_normal=True)
            self.scene._set("emitter.data", self.light)      # BRDF phase renders under the current best light (:317-334)
            with torch.no_grad():
                self.gt_image = render.render_w_brdf(self.scene, *self.gt, None, args.spp)
            self.init = tuple(t([getattr(s, k) for s in scenes]) for k in ("init_albedo", "init_roughness", "init_metallic"))

        def phase(self, mode):
            if mode == "fused_arm":                         # all three maps at once (BASELINE configs[0]'s --opt_order arm): the generic step
                cls = loop.PipelinedBrdfPhase if self.B >= 8 and self.B % 2 == 0 else loop.FusedBrdfPhase
                return cls(self.scene, self.gt_image, *self.init, optimize_part="arm", spp=args.spp, **({"groups": 2} if cls is loop.PipelinedBrdfPhase else {}))
            if mode in ("fused", "fused_a") and self.B >= 8 and self.B % 2 == 0:
                # a shard of images, as optimize.optimize_envmap_arm runs it: two groups of images stepping on streams of their own (one group's
                # walk and statistics launches under the other's streaming step; the same results, bit for bit)
                return loop.PipelinedBrdfPhase(self.scene, self.gt_image, *self.init, groups=2, optimize_part="rm" if mode == "fused" else "a", spp=args.spp)
            if mode in ("fused", "fused_one_phase"):        # fused_one_phase: the whole shard as ONE phase (the roofline's kernel timing)
                return loop.FusedBrdfPhase(self.scene, self.gt_image, *self.init, optimize_part="rm", spp=args.spp)
            if mode == "fused_exact":                       # every pixel's 20 GGX samples walked in every iteration (round 2's loop)
                return loop.FusedBrdfPhase(self.scene, self.gt_image, *self.init, optimize_part="rm", spp=args.spp, lazy=False)
            if mode == "fused_a":                           # the second part of --opt_order 'rm a': roughness fixed, specular sums reused
                return loop.FusedBrdfPhase(self.scene, self.gt_image, *self.init, optimize_part="a", spp=args.spp)
            if mode == "pos_mlp":
                from materialist_amd import posmlp

                net = posmlp.brdf_net("arm").to(dev)
                init = self.init
                start_arm = torch.cat([init[0].reshape(-1, 3), init[1].reshape(-1, 1), init[2].reshape(-1, 1)], -1).clamp(0, 1)
                # EarlyStopping armed as in the pipeline (it never fires here: the point is to time the loop WITH its polling of the device flag)
                ph = loop.pos_mlp_brdf_phase(self.scene, self.gt_image, net, start_arm, {"albedo": init[0], "roughness": init[1], "metallic": init[2]},
                                             optimize_part="rm", spp=args.spp, patience=10 ** 6, min_delta=0.001)
                ph.bench_step = ph.step_and_check    # what optimize.brdf_part_runner_mlp calls per epoch (:550)
                if not hasattr(ph, "current_maps"):
                    ph.current_maps = lambda: (lambda m: {"albedo": m["albedo"].detach().clamp(0, 1), "roughness": m["roughness"].detach().clamp(0.07, 1),
                                                          "metallic": m["metallic"].detach().clamp(0, 1)})(ph.maps_from_net()[0])
                return ph
            if mode == "pos_mlp_n":
                # --model_name pos_mlp with 'n' in --opt_order: the eight-output 'armn' network (inverse_img_w_mi.py:167-172,493-506) predicts the normal
                # map too; loop.PosMlpNormalPhase with armhead.MlpEngine (launch by launch on the C ABI since round 6)
                from materialist_amd import posmlp

                s_n = render.load_estimated_mesh(self.depth, use_mesh_normal=False)
                s_n._set("emitter.data", self.light)
                geo = self.scene.shading_normal()
                init = self.init
                start_armn = torch.cat([init[0].reshape(-1, 3), init[1].reshape(-1, 1), init[2].reshape(-1, 1), geo.reshape(-1, 3)], -1).clamp(-1, 1).contiguous()
                ph = loop.PosMlpNormalPhase(s_n, self.gt_image, posmlp.brdf_net("armn").to(dev), start_armn,
                                            {"albedo": init[0], "roughness": init[1], "metallic": init[2], "normal": geo}, optimize_part="armn", spp=args.spp,
                                            saver=loop.DeviceSaveBest())
                ph.bench_step = ph._step_device                 # (ph.step() returns the MSE as a host float: the reference's per-epoch poll; not timed here)
                ph.current_maps = lambda: {k: v for k, v in zip(("albedo", "roughness", "metallic"), init)}
                return ph
            if mode == "fused_n":
                # a part of --opt_order that moves the normal map ('n' under use_mesh_normal False): loop.NormalBrdfPhase
                s_n = render.load_estimated_mesh(self.depth, use_mesh_normal=False)
                s_n._set("emitter.data", self.light)
                geo = self.scene.shading_normal()
                gen = torch.Generator(device="cpu").manual_seed(1)
                n0 = torch.nn.functional.normalize(geo + 0.1 * torch.randn(geo.shape, generator=gen).to(dev), dim=-1).contiguous()
                return loop.NormalBrdfPhase(s_n, self.gt_image, *self.init, n0, optimize_part="n", spp=args.spp)
            if mode == "env":
                from materialist_amd import posmlp
                from materialist_amd.envhead import EnvMlpPhase

                s_env = render.load_estimated_mesh(self.depth, use_mesh_normal=True)
                pr = render.traverse(s_env)
                pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = self.gt
                enet = posmlp.envmap_net().to(dev)
                return EnvMlpPhase(s_env, self.gt_image, enet, torch.ones(512, 3, device=dev), spp=args.spp, lr=1e-3, use_graph=True)
            if mode == "env_texels":
                from materialist_amd.envhead import EnvTexelPhase

                s_env = render.load_estimated_mesh(self.depth, use_mesh_normal=True)
                pr = render.traverse(s_env)
                pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = self.gt
                raw = torch.zeros(16, 32, 3, device=dev, requires_grad=True)
                return EnvTexelPhase(s_env, self.gt_image, raw, spp=args.spp, lr=1e-3, use_graph=True)
            return loop.BrdfPhase(self.scene, self.gt_image, *self.init, None, optimize_part="rm", spp=args.spp)

    wl = Workload(args.images_per_gpu)
    B = wl.B
    stepper = lambda p: getattr(p, "bench_step", p.step)
    # the same protocol on a process whose GPU has done nothing yet (no device warm-up): what the driver's first process on an idle box
    # would see without the throw-away phase below -- reported, never the headline
    c_el, _ = proto.timed(stepper(wl.phase(mode)), args.warmup, min(args.steps, 20))
    cold = min(args.steps, 20) * B * world / c_el
    # Device warm-up, outside the protocol's W + K steps: a throw-away phase of the same mode runs for ~0.5 s.  The first GPU work of a fresh
    # process on an idle MI355X runs below its sustained clocks for some tens of milliseconds (measured: --steps 20 --warmup 5 as the box's
    # first process 256-284 it/s, the same command right after it 407-420); W = 5 iterations (12 ms) do not cover that.  Reported in the line.
    # The phase that is timed is BUILT first (its 1.8 GB of buffers, its diffuse cache) and the warm-up runs right before its W + K steps:
    # built after the warm-up (rounds 2-4), the allocations left the GPU idle for ~0.1 s and the protocol's first iterations ran on clocks that
    # had dropped again (--steps 20 --warmup 5 read 3 % below --steps 300 on the same box).
    phase = wl.phase(mode)
    psnr0 = float(loop._loss.psnr(render.render_w_brdf(wl.scene, *[phase.current_maps()[k].detach() for k in ("albedo", "roughness", "metallic")], None, args.spp), wl.gt_image).mean())
    t_w = time.perf_counter()
    warm = wl.phase(mode)
    wstep = getattr(warm, "bench_step", warm.step)
    n_warm = 0
    while time.perf_counter() - t_w < 0.5 and n_warm < 400:
        for _ in range(10):
            wstep()
        torch.cuda.synchronize()
        n_warm += 10
    device_warmup = {"seconds": round(time.perf_counter() - t_w, 3), "iterations_of_a_throwaway_phase": n_warm}
    elapsed, per_rank = proto.timed(stepper(phase), args.warmup, args.steps)
    del warm, wstep
    value = args.steps * B * world / elapsed
    m = phase.current_maps()
    with torch.no_grad():
        final = render.render_w_brdf(wl.scene, m["albedo"].detach(), m["roughness"].detach(), m["metallic"].detach(), None, args.spp)
    psnr1 = float(loop._loss.psnr(final, wl.gt_image).mean())

    # the other loops of the same pipeline, for the record (same fences; whole-job rates over all ranks)
    modes = {mode: {"it_per_s": value, "ms_per_step": elapsed / args.steps * 1e3, "images_per_gpu": B}}
    wl8 = None
    if not args.no_extras and mode != "torch":
        for extra, steps in (("fused", 2000), ("fused_a", 2000), ("fused_exact", 500), ("torch", 300), ("pos_mlp", 100), ("env", 500), ("env_texels", 1000), ("fused_n", 300), ("pos_mlp_n", 100)):
            if extra == mode or (not extra.startswith("fused") and B > 1):
                continue
            ph_x = wl.phase(extra)
            if hasattr(ph_x, "step_many"):            # hot loop A as optimize.env_phase_runner drives it: 10 iterations per poll (pipeline.py sync_every)
                for _ in range(4):
                    ph_x.step()
                e_el, _ = proto.timed(lambda: (ph_x.step_many(10), ph_x.poll()), 1, steps // 10)
            else:
                e_el, _ = proto.timed(stepper(ph_x), 10, steps)
            modes[extra] = {"it_per_s": steps * B * world / e_el, "ms_per_step": e_el / steps * 1e3, "images_per_gpu": B}
        if "pos_mlp_n" in modes:
            # ... and as rounds 4-5 ran it: the network under autograd (posmlp._PosMlpHipFn: three bf16 pieces, layer by layer) with torch.optim.AdamW
            keep_e = loop.PosMlpNormalPhase.ENGINE
            loop.PosMlpNormalPhase.ENGINE = False
            try:
                e_el, _ = proto.timed(stepper(wl.phase("pos_mlp_n")), 5, 50)
            finally:
                loop.PosMlpNormalPhase.ENGINE = keep_e
            modes["pos_mlp_n_autograd"] = {"it_per_s": 50 * B * world / e_el, "ms_per_step": e_el / 50 * 1e3, "images_per_gpu": B}
        if mode == "pos_mlp" and _posmlp._PosMlpHipFn.PRODUCTS:
            # the same loop with the 256-wide layers on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32), for the record
            keep = _posmlp._PosMlpHipFn.PRODUCTS
            _posmlp._PosMlpHipFn.PRODUCTS = 0
            try:
                e_el, _ = proto.timed(stepper(wl.phase("pos_mlp")), 10, 100)
            finally:
                _posmlp._PosMlpHipFn.PRODUCTS = keep
            modes["pos_mlp_exact_f32"] = {"it_per_s": 100 * B * world / e_el, "ms_per_step": e_el / 100 * 1e3, "images_per_gpu": B}
            # ... and on three bf16 pieces per operand, six products, layer by layer (round 4's form of every product of the iteration)
            from materialist_amd.armhead import ArmMlpPhase as _Arm
            keep_sw = (_Arm.FWD_PRODUCTS, _Arm.FWD_CHAIN, _Arm.BWD_F16)
            _Arm.FWD_PRODUCTS, _Arm.FWD_CHAIN, _Arm.BWD_F16 = 0, False, False
            try:
                e_el, _ = proto.timed(stepper(wl.phase("pos_mlp")), 10, 100)
            finally:
                _Arm.FWD_PRODUCTS, _Arm.FWD_CHAIN, _Arm.BWD_F16 = keep_sw
            modes["pos_mlp_bf16x3"] = {"it_per_s": 100 * B * world / e_el, "ms_per_step": e_el / 100 * 1e3, "images_per_gpu": B}
        if B != 8 and H * W <= 512 * 512:
            wl8 = Workload(8)                 # BASELINE configs[2]: 64 images, 8 per GPU; this is one GPU's shard (every rank runs its own)
            ph8 = wl8.phase("fused")
            e_el, _ = proto.timed(ph8.step, 10, 300)
            modes["fused_b8"] = {"it_per_s": 300 * 8 * world / e_el, "ms_per_step": e_el / 300 * 1e3, "images_per_gpu": 8}
            e_el, _ = proto.timed(ph8.step, 0, 500)           # the SAME phase goes on: iterations 311-810 of the part
            modes["fused_b8_steady"] = {"it_per_s": 500 * 8 * world / e_el, "ms_per_step": e_el / 500 * 1e3, "images_per_gpu": 8}
            del ph8
            ph1 = wl8.phase("fused_one_phase")                # the same shard as ONE phase on one stream (round 3's form of fused_b8)
            e_el, _ = proto.timed(ph1.step, 10, 300)
            modes["fused_b8_one_phase"] = {"it_per_s": 300 * 8 * world / e_el, "ms_per_step": e_el / 300 * 1e3, "images_per_gpu": 8}
            del ph1
            e_el, _ = proto.timed(wl8.phase("fused_arm").step, 10, 300)
            modes["fused_b8_arm"] = {"it_per_s": 300 * 8 * world / e_el, "ms_per_step": e_el / 300 * 1e3, "images_per_gpu": 8}
            e_el, _ = proto.timed(wl8.phase("fused_a").step, 10, 300)
            modes["fused_b8_a"] = {"it_per_s": 300 * 8 * world / e_el, "ms_per_step": e_el / 300 * 1e3, "images_per_gpu": 8}
            e_el, _ = proto.timed(wl8.phase("fused_exact").step, 10, 100)
            modes["fused_b8_exact"] = {"it_per_s": 100 * 8 * world / e_el, "ms_per_step": e_el / 100 * 1e3, "images_per_gpu": 8}
    # 8-GPU readiness that one GPU can show: what an iteration costs the HOST (Python + ctypes + launch calls), next to what it costs the GPU.
    # Measured as the wall time per step() of the same phase classes on images so small that the GPU side is far shorter than the host side
    # (64 x 64 / 128 x 128: the launches and their arguments are the same); eight ranks on one node need that many microseconds of a core each
    host_enqueue = None
    if not args.no_extras and mode != "torch":
        host_enqueue = {}
        for name, hb, hsize, hmode in (("PipelinedBrdfPhase_b8_rm", 8, 64, "fused"), ("FusedBrdfPhase_b1_rm", 1, 64, "fused"), ("ArmMlpPhase_rm", 1, 128, "pos_mlp")):
            try:
                wl_s = Workload(hb, hsize)
                ph_s = wl_s.phase(hmode)
                for _ in range(20):
                    ph_s.step()
                torch.cuda.synchronize()
                t_h = time.perf_counter()
                for _ in range(200):
                    ph_s.step()
                host_us = (time.perf_counter() - t_h) / 200 * 1e6       # enqueue only: no synchronisation inside
                torch.cuda.synchronize()
                full = {"PipelinedBrdfPhase_b8_rm": "fused_b8", "FusedBrdfPhase_b1_rm": "fused", "ArmMlpPhase_rm": "pos_mlp"}[name]
                host_enqueue[name] = {"host_enqueue_us_per_iteration": host_us, "gpu_us_per_iteration_at_512": (modes[full]["ms_per_step"] * 1e3) if full in modes else None,
                                      "measured_on": f"{hb} x {hsize}x{hsize} (same launches; the GPU side is shorter than the host side there)"}
                del ph_s, wl_s
            except Exception as e:
                host_enqueue[name] = {"error": repr(e)}
        host_enqueue["cores"] = len(os.sched_getaffinity(0))
        host_enqueue["note"] = ("a rank's iteration is enqueued by ONE Python thread; with the ranks of a node pinned to disjoint core sets (bench.py launch_ranks, "
                                "run_batch.py) each needs one core that is not shared: where host_enqueue approaches the GPU time (the 8-image none-mode shard) a "
                                "shared or oversubscribed core is what would cost the 8-GPU scaling, not the fabric")
    mode_names = {"fused": "hot loop B, --model_name none, whole iteration in libmatpbr.so, THREE launches in a part that moves the roughness, two otherwise: the "
                           "partial sums of the loss statistics; the folded, persistent step (csrc/matpbr_pstep.hpp: at most 1024 workgroups stream the image "
                           "from two register sets, fold the statistics, commit SaveBest / EarlyStopping, run the backward pass, Adam and the next iteration's "
                           "render from per-pixel models into which the maps the part leaves alone are folded); the walk of the pixels that left their "
                           "model's interval, eight per wave from a queue (|render - exact sampling| <= 1e-3 on every pixel of every iteration, "
                           "tests/test_gpu_lazy.py)",
                  "fused_b8": "hot loop B, --model_name none, BASELINE configs[2]'s per-GPU shard (8 x 512x512), part 'rm', the first 310 iterations of the part: the "
                              "shard as two groups of four images stepping on streams of their own (loop.PipelinedBrdfPhase; an image's iteration does not "
                              "depend on the images beside it: the same results as one phase over the shard, bit for bit) -- one group's walk and statistics "
                              "launches, latency-bound, run under the other's streaming step (512 workgroups: MATPBR_FLAG_SHARE_GPU)",
                  "fused_one_phase": "--mode fused_one_phase (tools: traces and counter passes): the images of the shard as ONE FusedBrdfPhase on one stream",
                  "fused_b8_one_phase": "the shard as ONE FusedBrdfPhase on one stream (what fused_b8 was in rounds 2-3 and earlier in round 4)",
                  "fused_b8_arm": "the shard in a part that moves all three maps (BASELINE configs[0]'s --opt_order arm): nothing is constant to fold, the generic step "
                                  "(round 3's kernels: 80 B/pixel of model + the three maps), two groups of four images on streams of their own",
                  "fused_b8_steady": "fused_b8 further into the part (iterations 311-810 of the same phase): a part's first iterations re-sample ten times as many "
                                     "pixels as its steady state (0.26 % per iteration), and the reference's parts run for hundreds to thousands of iterations",
                  "fused_exact": "the same loop walking the 20 GGX samples of every pixel in every iteration (round 2: render+jac, statistics, streaming backward+Adam)",
                  "fused_b8_exact": "the 8-image shard with exact sampling in every iteration (round 2's fused_b8)",
                  "fused_a": "the same two launches in part 'a' of --opt_order 'rm a' (roughness fixed): no pixel ever leaves its model's interval, "
                             "nothing is re-sampled after the part's first render",
                  "fused_b8_a": "part 'a' on the 8-image shard, two groups of four images on streams of their own as fused_b8 (image-iterations/s)",
                  "pos_mlp": "hot loop B, --model_name pos_mlp: every launch of the iteration a kernel of libmatpbr.so (armhead.ArmMlpPhase): the network's forward pass "
                             "as ONE launch (transposed products chain the layers through registers: csrc/posmlp_chain.hip), render, loss, the backward products "
                             "(input gradients with cos and bias-gradient epilogues, weight gradients) on two f16 pieces under one exponent per 128-row tile, "
                             "AdamW on a flat buffer; no BLAS, no autograd",
                  "pos_mlp_bf16x3": "the pos_mlp loop as round 4 ran it: every 256-wide product on three bf16 pieces per operand and six matrix products, layer "
                                    "by layer (ArmMlpPhase with FWD_PRODUCTS 0, FWD_CHAIN False, BWD_F16 False); the default differs in how the f32 products are formed",
                  "pos_mlp_exact_f32": "the pos_mlp loop with --mlp-products 0: 256-wide layers on the exact-f32 MFMA kernels, autograd composition "
                                       "(loop.PosMlpBrdfPhase); the default differs from it only in how the f32 products are formed (same error vs fp64)",
                  "fused_n": "hot loop B, --model_name none, a part that moves the normal map ('n', use_mesh_normal False; loop.NormalBrdfPhase): render, loss "
                             "statistics with SaveBest / EarlyStopping on the device, d loss / d pred, the backward render into materials and normals "
                             "(both lobes' directions walked per pixel), regularisers + normalize backward + Adam -- nine launches of libmatpbr.so, no autograd",
                  "pos_mlp_n": "hot loop B, --model_name pos_mlp with 'n' in --opt_order (part 'armn'): the eight-output network launch by launch on the C ABI "
                               "(armhead.MlpEngine: thin-K first layer, 256-wide layers on two f16 pieces forward and backward, 8-column output layer, AdamW on "
                               "the flat buffer), render under the predicted normal map, loss statistics, material and normal gradients (loop.PosMlpNormalPhase)",
                  "pos_mlp_n_autograd": "pos_mlp_n as rounds 4-5 ran it (PosMlpNormalPhase.ENGINE False): the network under autograd on the bf16 x 3 layer kernels, "
                                        "torch.optim.AdamW",
                  "env_texels": "hot loop A of --model_name none (envhead.EnvTexelPhase): one pass over the radiance transfer, then ONE workgroup that folds its "
                                "partial sums, commits SaveBest / EarlyStopping, snapshots the best envmap, back-propagates through the SH projection and the "
                                "softplus and applies Adam (matpbr_env_texel_phase_step), then the next envmap's projection: three kernels from a hipGraph, seven in round 3",
                  "env": "hot loop A: envmap PosMLP (small-tile MFMA layers) + softplus / SH projection + one pass over the radiance transfer and one workgroup "
                         "behind it (fold, SaveBest / EarlyStopping, snapshot, projection backward) + backward chain + Adam: 14 kernels of libmatpbr.so per "
                         "iteration, ten iterations per hipGraph replay and poll",
                  "torch": "hot loop B as the reference's loop body runs it UNCHANGED on the operator face (loop.BrdfPhase: clamp, render_w_brdf with autograd, "
                           "torch losses, torch.optim.Adam; inverse_img_w_mi.py:371-432): render_w_brdf renders from the scene's cached per-pixel models"}
    modes = {k: dict(v, what=mode_names[k]) for k, v in modes.items()}

    # kernel durations for the roofline: back-to-back launches between two HIP events on the launch stream (torch's current
    # stream, the one the C ABI is handed)
    def back_to_back(fn, reps=50):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    roof = None
    if not args.no_extras:
        def kernel_times(w):
            """The BRDF fwd / bwd kernels on workload w: in-loop pair (cached diffuse lobe + jac, streaming loss backward) and the
            stand-alone operator pair (everything in-kernel)."""
            mm = [x.contiguous() for x in w.init]
            nrm = w.scene.shading_normal().contiguous()
            lgt = (w.scene.light if (w.scene.light.ndim == 3 or w.B == 1) else w.scene.light.unsqueeze(0).expand(w.B, -1, -1)).contiguous()
            dcache, jac = ops.diffuse_cache(nrm, lgt, args.spp), ops.plane9(mm[0])
            pred = torch.empty_like(w.gt_image)
            gt_srgb = loop._loss.linear_to_srgb(w.gt_image).contiguous()
            stats = ops.new_loss_stats(w.B, dev)
            g = [torch.empty_like(x) for x in mm]
            d_probe = torch.randn_like(w.gt_image)
            ops.shade_fwd(*mm, nrm, lgt, args.spp, clamp_params=True, out=pred, dcache=dcache, jac=jac)
            ops.brdf_loss_stats(pred, w.gt_image, gt_srgb, *mm, *mm, 0.1, stats, None, optimize_part="rm")
            t = {"fwd_loop": back_to_back(lambda: ops.shade_fwd(*mm, nrm, lgt, args.spp, clamp_params=True, out=pred, dcache=dcache, jac=jac)),
                 "bwd_loop": back_to_back(lambda: ops.brdf_loss_bwd_jac(*mm, jac, pred, gt_srgb, stats, *mm, 0.1, *g, optimize_part="rm")),
                 "fwd_op_uncached": back_to_back(lambda: ops.shade_fwd(*mm, nrm, lgt, args.spp, out=pred), 20),
                 "bwd_op_uncached": back_to_back(lambda: ops.shade_bwd(*mm, nrm, lgt, d_probe, args.spp, want_mat=True), 20),
                 "diffuse_cache": back_to_back(lambda: ops.diffuse_cache(nrm, lgt, args.spp, out=dcache), 20)}
            # the operator face as render.Scene runs it for a call that needs material gradients (render_w_brdf inside the reference's loop,
            # inverse_img_w_mi.py:384-386): forward from the scene's cached per-pixel models, backward from their half-precision jac planes.
            # Steady state of a loop: the models exist, the roughness has moved by an Adam-sized step since the call before
            lz = ops.lazy_state(mm[0])
            st_op = ops.new_loss_stats(w.B, dev)
            st_op[:, ops.STAT_RATIO] = 1.0
            o_op, j16 = ops.shade_fwd_lazy(*mm, nrm, lgt, args.spp, dcache, lz, force=True, stats=st_op, jac32=True)
            st_op[:, ops.STAT_GT_SUM] = o_op.reshape(w.B, -1).sum(dim=1)
            mm_op = [mm[0], (mm[1] + 3e-4).contiguous(), mm[2]]
            ops.shade_fwd_lazy(*mm_op, nrm, lgt, args.spp, dcache, lz, out=o_op, jac16=j16, stats=st_op, jac32=True)
            t["fwd_op"] = back_to_back(lambda: ops.shade_fwd_lazy(*mm_op, nrm, lgt, args.spp, dcache, lz, out=o_op, jac16=j16, stats=st_op, jac32=True))
            t["bwd_op"] = back_to_back(lambda: ops.shade_bwd_jac(*mm_op, j16, d_probe))
            del lz, o_op, j16
            s1 = torch.empty((3,) + tuple(jac.shape[1:]), dtype=torch.float32, device=dev)
            ops.shade_fwd(*mm, nrm, lgt, args.spp, clamp_params=True, out=pred, dcache=dcache, jac=jac, s1=s1)
            t["fwd_cached"] = back_to_back(lambda: ops.shade_fwd_cached(mm[0], mm[2], jac, s1, clamp_params=True, out=pred))
            t["bwd_loop_a"] = back_to_back(lambda: ops.brdf_loss_bwd_jac(*mm, jac, pred, gt_srgb, stats, *mm, 0.1, *g, optimize_part="a"))
            return t

        def entry(px, t_f, t_b, note):
            ach = lambda nbytes, ms: nbytes * px / (ms * 1e-3) / 1e9
            e = {"achieved": ach(BYTES_FWD + BYTES_BWD_ARM, t_f + t_b), "bytes_per_pixel": BYTES_FWD + BYTES_BWD_ARM, "avg_launch_ms": t_f + t_b,
                 "fwd": {"achieved": ach(BYTES_FWD, t_f), "frac": ach(BYTES_FWD, t_f) / (HBM_PEAK / 1e9), "avg_launch_ms": t_f, "bytes_per_pixel": BYTES_FWD},
                 "bwd": {"achieved": ach(BYTES_BWD_ARM, t_b), "frac": ach(BYTES_BWD_ARM, t_b) / (HBM_PEAK / 1e9), "avg_launch_ms": t_b,
                         "bytes_per_pixel": BYTES_BWD_ARM}, "note": note}
            e["frac"] = e["achieved"] / (HBM_PEAK / 1e9)
            return e

        def lazy_step_times(w):
            """The ONE launch that is the BRDF fwd+bwd pair of the lazy loop (backward of iteration t + Adam + render of iteration t+1), timed
            IN the loop (iterations 301-500 of a phase, so that the share of pixels re-sampled per launch is the loop's, not the start-up's)
            with HIP events ON it (the kernel's own begin and end: FusedBrdfPhase.step_timed), and the statistics launch back to back."""
            ph = w.phase("fused_one_phase")
            ph.run(300)
            ev = []
            for _ in range(200):
                ph.step_timed(ev)             # the real loop, HIP events on the launch in question (on the launch stream)
            torch.cuda.synchronize()
            t_step = sum(e[0].elapsed_time(e[1]) for e in ev) / len(ev)        # the step kernel's own begin -> end (hipExtLaunchKernelGGL events)
            t_res = sum(e[-1 if len(e) == 4 else 1].elapsed_time(e[2]) for e in ev) / len(ev)
            _, ref = ops.lazy_state_unpack(ph.lazy_state, ph.p["albedo"])
            # (a folded phase forms its statistics inside the step kernel from the second iteration of a part on: launch_stage(2) launches nothing there)
            t_stats = 0.0 if getattr(ph, "fold", False) else back_to_back(lambda: ph.launch_stage(2))
            return t_step, t_stats, float(ref.float().mean()), t_res

        def lazy_gradient_error(w):
            """What the timed kernel's gradients are worth: the loss gradients the lazy step forms from its per-pixel models at iteration
            400 of a part against the streaming backward pass on the EXACT jac planes of the same parameters (every sample of every pixel walked:
            tests/test_gpu_lazy.py measures the same every 50th iteration of 2000)."""
            ph = loop.FusedBrdfPhase(w.scene, w.gt_image, *w.init, optimize_part="rm", spp=args.spp, keep_grads=True)
            ph.run(400)
            p_at = [ph.p[k].clone() for k in ("albedo", "roughness", "metallic")]
            pred_at = ph.pred.clone()                                   # the lazy loop's own render of these parameters
            ph.step()
            exact, jac_e = torch.empty_like(w.gt_image), ops.plane9(w.gt_image)
            g_ref = {k: torch.empty_like(v) for k, v in ph.g.items()}
            ops.shade_fwd(*p_at, ph.n, ph.light, args.spp, clamp_params=True, out=exact, dcache=ph.dcache, jac=jac_e)
            res = {}
            for tag, pred_for_loss in (("derivatives_of_the_models", pred_at), ("whole_loss_gradient", exact)):
                ops.brdf_loss_bwd_jac(*p_at, jac_e, pred_for_loss, ph.gt_srgb, ph.stats, ph.orig["albedo"], ph.orig["roughness"], ph.orig["metallic"], 0.1,
                                      g_ref["albedo"], g_ref["roughness"], g_ref["metallic"], optimize_part="rm")
                res[tag] = {}
                for k in ("roughness", "metallic"):
                    e = ((ph.g[k] - g_ref[k]).abs() / torch.maximum(g_ref[k].abs(), g_ref[k].abs().mean())).reshape(-1)
                    res[tag][k] = {"rel_l2": float((ph.g[k] - g_ref[k]).norm() / g_ref[k].norm()), "worst_pixel": float(e.max()),
                                   "p999_pixel": float(torch.quantile(e[:: max(1, e.numel() // 2_000_000)].float(), 0.999)), "median_pixel": float(e.median()),
                                   "cosine": float((ph.g[k] * g_ref[k]).sum() / (ph.g[k].norm() * g_ref[k].norm()))}
            return res

        wr = wl8 if wl8 is not None else wl
        tk = kernel_times(wr)
        px = H * W * wr.B
        pmc = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath))
            except Exception:
                pmc = {}
        from materialist_amd.build import sources_digest
        if os.environ.get("MATPBR_LIB"):
            # a measurement build beside the product library: the counter passes were not collected on it
            pmc = {"stale": f"MATPBR_LIB={os.environ['MATPBR_LIB']}: the library that ran is not the one profiles/pmc_traffic.json was collected on"}
        elif pmc.get("csrc_sha16") != sources_digest():
            # the counter passes were collected on other kernels than the ones this run executes: no traffic figure rather than a stale one
            pmc = {"stale": f"profiles/pmc_traffic.json was collected on csrc {pmc.get('csrc_sha16')}, this build is {sources_digest()}: "
                            "re-run tools/pmc_passes_r06.sh + tools/pmc_to_traffic.py --write"}
        key = f"{H}x{W}_b{wr.B}_spp{args.spp}"
        t_step, t_stats, resampled, t_res = lazy_step_times(wr)
        ach = (BYTES_FWD + BYTES_BWD_ARM) * px / (t_step * 1e-3) / 1e9
        traffic = pmc.get(f"lazy_pstep_{key}")
        ge = lazy_gradient_error(wr)
        roof = {"bound": "hbm", "kernel": "lazy_pstep_kernel (BRDF bwd of iteration t + Adam + fwd of t+1, one launch)",
                "kernel_detail": "lazy_pstep_kernel<kFoldXY>: backward of iteration t (d loss/d pred, material gradients, regularisers, clamp gating, "
                                          "SaveBest by buffer rotation, Adam) + render of iteration t+1 from per-pixel local models in the roughness with the part's "
                                          "constant albedo folded in (68 B/pixel of model); the loss statistics of iteration t+1 (SaveBest / EarlyStopping / "
                                          "exposure ratio: no statistics launch, no stored render); the pixels that left their model's interval are queued and re-sampled "
                                          "(20 GGX samples) by the small launch behind it (resample_launch_ms)",
                "achieved": ach, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / (HBM_PEAK / 1e9), "avg_launch_ms": t_step,
                "timing": "mean over iterations 301-500 of a phase of the kernel's own begin -> end, two HIP events handed to the launch on the launch "
                          "stream (hipExtLaunchKernelGGL via matpbr_brdf_phase_stages_timed); events RECORDED around the launch (rounds 2-3) add the "
                          "stream's dispatch latency on both sides, 2-5 us, and disagreed with the rocprofv3 trace by that much",
                "bytes_per_pixel": BYTES_FWD + BYTES_BWD_ARM,
                "workload": f"{wr.B} x {H}x{W} (BASELINE configs[2] per-GPU shard)" if wr.B == 8 else f"{wr.B} x {H}x{W}",
                "resampled_fraction": resampled, "stats_launches_ms": t_stats, "resample_launch_ms": t_res,
                "gradient_error_of_this_kernel": dict(ge, note="d loss / d roughness, d loss / d metallic at iteration 401 of the part, every pixel, "
                                                      "against the backward pass on exact sampling of the same parameters (pixel errors relative to max(|g|, mean|g|)).  "
                                                      "derivatives_of_the_models: d loss / d pred formed on the lazy loop's own render for both sides -- the error of d out / d r "
                                                      "(first order in r - r_ref since round 5; radius and kink allowance of the intervals also on the derivative since round 6) and d out / d m alone; whole_loss_gradient: on the exact render -- adds the pixels "
                                                      "whose sign(pred - gt) of the L1 term differs between two renders that agree to 1e-3 (converged pixels; grows as the part "
                                                      "converges).  The RENDER of the same kernel is within 1e-3 of exact sampling on every pixel (tests/test_gpu_lazy.py)"),
                "grad_rel_l2": ge["derivatives_of_the_models"]["roughness"]["rel_l2"], "grad_worst_pixel": ge["derivatives_of_the_models"]["roughness"]["worst_pixel"],
                "grad_p999_pixel": ge["derivatives_of_the_models"]["roughness"]["p999_pixel"],
                "traffic": traffic, "traffic_source": pmc.get("source_r04") if traffic else pmc.get("stale"),
                "own_traffic_frac": (traffic / (t_step * 1e-3) / HBM_PEAK) if traffic else None,
                "note": "algorithmic bytes = SURVEY 8d's 44 (forward) + 64 (backward, arm) B/pixel for the pair this launch performs; `traffic` = the bytes it "
                        "really moves per launch (PMC; by construction 136 B/pixel in an 'rm' part: r, m read and written 16, the folded models 68, target 12, "
                        "anchors 8, Adam moments 32 -- the next render is no longer stored (round 6: its loss statistics are formed where it is formed, "
                        "stats_launches_ms is 0 from the second iteration of a part on; matpbr_brdf_phase_resolve evaluates the models when the caller reads pred)), "
                        "own_traffic_frac = traffic / duration / peak: how close the launch is to the HBM limit on its OWN bytes"}
        ex = entry(px, tk["fwd_loop"], tk["bwd_loop"],
                   "round 2's pair (FusedBrdfPhase(lazy=False)): shade_kernel<jac> walks the 20 GGX samples of every pixel (cached diffuse lobe) + "
                   "jac_bwd_kernel<fused> (streaming)")
        valu = pmc.get("shade_kernel_jac_valu_active_frac")
        ex["fwd"].update({"bound": "valu", "valu_frac": valu,
                          "valu_note": "SQ_ACTIVE_INST_VALU x 4 / (SIMDs x GRBM_GUI_ACTIVE / 8), profiles/r02_pmc_b8_{sq,grbm}.csv: the share of the launch the "
                                       "vector ALUs are issuing; the canonical HBM fraction beside it is reported for continuity only"})
        tb = pmc.get("detail_b8", {}).get("matpbr::jac_bwd_kernel<true>") if wr.B == 8 else pmc.get("detail_b1", {}).get("matpbr::jac_bwd_kernel<true>")
        ex["bwd"].update({"bound": "hbm", "own_traffic_frac": (tb / (tk["bwd_loop"] * 1e-3) / HBM_PEAK) if tb else None})
        roof.update({"exact_sampling_pair": ex,
                     "operator_face": entry(px, tk["fwd_op"], tk["bwd_op"],
                                            "render_w_brdf inside a loop that asks for material gradients (render.Scene with its per-(light, normals) cache): "
                                            "matpbr_shade_fwd_lazy (fp32 jac planes) from the scene's per-pixel models + matpbr_shade_bwd_jac, both streaming"),
                     "operator_face_uncached": entry(px, tk["fwd_op_uncached"], tk["bwd_op_uncached"],
                                                     "stand-alone matpbr_shade_fwd / matpbr_shade_bwd<mat>: both lobes sampled in-kernel (18 + 20 directions): a call "
                                                     "whose light or normals take part in autograd, or a light seen for the first time"),
                     "fixed_roughness_parts": entry(px, tk["fwd_cached"], tk["bwd_loop_a"],
                                                    "the pair in the parts of --opt_order that leave the roughness alone ('a' of 'rm a', about half of "
                                                    "the reference schedule's BRDF iterations): shade_cached_kernel combines the specular sums kept from the "
                                                    "part's first render (bit-identical to walking the samples) + jac_bwd_kernel<fused>; both streaming"),
                     "diffuse_cache_ms": tk["diffuse_cache"]})
        if wr is not wl:
            t1, ts1, rs1, tr1 = lazy_step_times(wl)
            a1 = (BYTES_FWD + BYTES_BWD_ARM) * H * W * wl.B / (t1 * 1e-3) / 1e9
            roof["single_image"] = {"achieved": a1, "frac": a1 / (HBM_PEAK / 1e9), "avg_launch_ms": t1, "stats_launches_ms": ts1, "resample_launch_ms": tr1, "resampled_fraction": rs1,
                                    "note": f"lazy_pstep_kernel on {wl.B} x {H}x{W} (working set inside the 256 MB Infinity Cache: launch-latency territory)"}
        # hot loop A: one pass over the radiance transfer per iteration (HBM-bound by construction)
        ph_e = wl.phase("env") if B == 1 else None
        if ph_e is not None:
            lc = wl.light.contiguous()
            P_ = lambda t: None if t is None else __import__("ctypes").c_void_p(t.data_ptr())
            from materialist_amd import _lib
            lib = _lib.load()

            def env_call():
                lib.matpbr_env_phase_step(P_(ph_e.T), P_(lc), P_(ph_e.gt_srgb), None, P_(ph_e.d_light), P_(ph_e.stats), None, 0, 0, 0.0, P_(ph_e.ws_env),
                                          ph_e.ws_env.numel() * 4, H, W, 1, __import__("ctypes").c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            ms_e = back_to_back(env_call)
            roof["env_prt"] = {"bound": "hbm", "kernel": "env_prt_kernel + env_final_kernel (render, loss, d_light in one pass over T)",
                               "achieved": BYTES_ENV * H * W / (ms_e * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "avg_launch_ms": ms_e,
                               "bytes_per_pixel": BYTES_ENV}
            roof["env_prt"]["frac"] = roof["env_prt"]["achieved"] / roof["env_prt"]["peak"]
            del ph_e

    # the PosMLP side of the pos_mlp iteration: nine [H*W,256]x[256,256] products per iteration, f32-accurate, in the hand-written
    # kernels of libmatpbr.so (epilogues included): the split-operand kernels on the bf16 matrix pipe (what the iteration runs),
    # the exact-f32 MFMA kernels and the BLAS product of the same shape beside them
    gemm = None
    if B == 1 and not args.no_extras:
        from materialist_amd import posmlp as _pm

        Mg = H * W
        xg, gg = torch.randn(Mg, 256, device=dev), torch.randn(Mg, 256, device=dev)
        wg, bg = torch.randn(256, 256, device=dev) / 16, torch.randn(256, device=dev)
        sg, cg, gp = (torch.empty(Mg, 256, device=dev) for _ in range(3))
        dbg = torch.empty(256, device=dev)
        dwg = torch.empty(256, 256, device=dev)
        flop = 2.0 * Mg * 256 * 256
        tf = lambda ms: flop / (ms * 1e-3) / 1e12
        P = int(_pm._PosMlpHipFn.PRODUCTS) or 6
        wsp = ops.mlp_split_weights(wg, 256, 256)
        bx = None
        if Mg % 128 == 0:
            # as the iteration runs them: the sines carry the sign of their cosine (no cosine matrix is written or read)
            ms_f = back_to_back(lambda: ops.mlp_layer_fwd_bx(xg, wsp, bg, sg, None, 256, 256, P), 20)
            ms_i = back_to_back(lambda: ops.mlp_layer_bwd_input_bx(gg, wsp, sg, gp, 256, 256, dbg, P, packed=True), 20)
            ms_w = back_to_back(lambda: ops.mlp_layer_bwd_weight_bx(gg, xg, 256, 256, P, out=dwg), 20)
            bx = {"products": P, "forward_sincos": {"avg_launch_ms": ms_f, "achieved": tf(ms_f)},
                  "bwd_input_mulcos_colsum": {"avg_launch_ms": ms_i, "achieved": tf(ms_i)},
                  "bwd_weight": {"avg_launch_ms": ms_w, "achieved": tf(ms_w)}, "achieved": 3 * flop / ((ms_f + ms_i + ms_w) * 1e-3) / 1e12}
        ms_fwd = back_to_back(lambda: ops.mlp_layer_fwd(xg, wg, bg, sg, cg, 256), 20)
        ms_din = back_to_back(lambda: ops.mlp_layer_bwd_input(gg, wg, cg, gp, 256, 256, dbg), 20)
        ms_dw = back_to_back(lambda: ops.mlp_layer_bwd_weight(gg, xg, 256, 256), 20)
        ms_g = back_to_back(lambda: torch.mm(xg, wg), 20)
        f32k = {"kernel": "mlp_gemm_nt_wide<sincos> / <mul cos> / mlp_wgrad_tn on v_mfma_f32_32x32x2_f32",
                "achieved": 3 * flop / ((ms_fwd + ms_din + ms_dw) * 1e-3) / 1e12,
                "forward_sincos": {"avg_launch_ms": ms_fwd, "achieved": tf(ms_fwd)},
                "bwd_input_mulcos_colsum": {"avg_launch_ms": ms_din, "achieved": tf(ms_din)},
                "bwd_weight": {"avg_launch_ms": ms_dw, "achieved": tf(ms_dw)}}
        if bx:
            # what the iteration runs since round 5: two f16 pieces per operand, three products (half the matrix work of the bf16 form) -- the layers are
            # then bound by their HBM traffic, and that is what they are priced against; the share of the f16 matrix pipe is reported beside it
            B4 = Mg * 256 * 4.0
            f16 = None
            try:
                wsp3, wsp3t = ops.mlp_split_weights(wg, 256, 256, f16=True), ops.mlp_split_weights(wg, 256, 256, transposed=True, f16=True)
                xs_ = torch.sin(xg)                                   # the rows of a forward layer are sines
                ops.mlp_layer_fwd_bx(xs_, wsp3, bg, sg, None, 256, 256, 3)
                g_small = gg * 1e-6                                   # loss gradients of a mean over 512 x 512 pixels
                tmx = g_small.abs().view(Mg // 128, -1).amax(1).contiguous().view(torch.int32)
                tmo = ops.mlp_tile_max(Mg, dev)
                ms3_f = back_to_back(lambda: ops.mlp_layer_fwd_bx(xs_, wsp3, bg, sg, None, 256, 256, 3), 20)
                ms3_i = back_to_back(lambda: ops.mlp_layer_bwd_input_blk(g_small, tmx, wsp3t, sg, gp, 256, 256, dbg, tmo), 20)
                ms3_w = back_to_back(lambda: ops.mlp_layer_bwd_weight_blk(g_small, tmx, sg, 256, 256, out=dwg), 20)
                leg = lambda ms, nb: {"avg_launch_ms": ms, "bytes": nb, "achieved": nb / (ms * 1e-3) / 1e9, "frac": nb / (ms * 1e-3) / HBM_PEAK,
                                      "f16_matrix_pipe_frac": 3 * flop / (ms * 1e-3) / 2.5e15}
                f16 = {"products": 3, "forward_sin": leg(ms3_f, 2 * B4), "bwd_input_mulcos_colsum": leg(ms3_i, 3 * B4), "bwd_weight": leg(ms3_w, 2 * B4),
                       "bytes_note": "algorithmic: forward reads the rows and writes the sign-carrying sines (2 x M x 256 x 4 B); the input gradient reads g and the "
                                     "sines below and writes g' (3 x); the weight gradient reads g and the rows (2 x)"}
                ph_c = wl.phase("pos_mlp")
                if getattr(ph_c, "chain", False):
                    ms_c = back_to_back(lambda: ph_c.forward(), 20)
                    nb_c = 4 * B4 + Mg * (16 + 8 + 5) * 4.0
                    f16["forward_chain"] = dict(leg(ms_c, nb_c), what="matpbr_mlp_chain_fwd (+ its image preparation): the whole network forward in one launch; writes "
                                                "the four layers' sines once, reads none of them (layer by layer: 91 us first layer + three 256-wide layers)",
                                                f16_matrix_pipe_frac=3 * 3 * flop / (ms_c * 1e-3) / 2.5e15)
                del ph_c
            except Exception as e:                                     # (a library without the round-5 entry points)
                f16 = {"error": repr(e)}
            tot_ms = (f16["forward_sin"]["avg_launch_ms"] + f16["bwd_input_mulcos_colsum"]["avg_launch_ms"] + f16["bwd_weight"]["avg_launch_ms"]) if "forward_sin" in f16 else None
            gemm = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
                    "achieved": (7 * B4 / (tot_ms * 1e-3) / 1e9) if tot_ms else 0.0,
                    "kernel": "the three 256-wide products of a layer as the pos_mlp iteration runs them: mlp_nt_gx<sin, 3> / mlp_nt_gx<mul cos, 3> / mlp_wgrad_hx "
                              "([H*W,256]x[256,256], two f16 pieces per operand, three products, f32 accumulate; epilogues included), back to back",
                    "two_piece_f16": f16,
                    "three_piece_bf16": dict(bx, note=f"round 4's form ({P} bf16 products per f32 product): TFLOP/s in f32-equivalent terms (2 M N K per product); "
                                                      f"its matrix-pipe share is {P} x that / 2500"),
                    "exact_f32": f32k,
                    "stand_alone_vs_in_loop": "these legs run each kernel 20 times back to back (the chip holds a lower clock under sustained matrix work than inside "
                                              "the iteration, where the same launches are 5-15 % shorter: profiles/r05_pos_mlp_sequence.txt)",
                    "blas_product_same_shape": {"avg_launch_ms": ms_g, "achieved": tf(ms_g), "kernel": "hipBLASLt f32 (PyTorch-ROCm), no epilogue"}}
        else:
            gemm = {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3, "peak_note": "dense f32 MFMA peak (exact-f32 kernels)", "kernel": f32k["kernel"],
                    "achieved": f32k["achieved"], "exact_f32": f32k,
                    "blas_product_same_shape": {"avg_launch_ms": ms_g, "achieved": tf(ms_g), "kernel": "hipBLASLt f32 (PyTorch-ROCm), no epilogue"}}
        gemm["frac"] = gemm["achieved"] / gemm["peak"]
        del xg, gg, wg, sg, cg, gp, dwg

    # BASELINE configs[4]: forward-only relighting, 2048x2048, 360 lights, through the precomputed transfer (HBM-bound kernel)
    relight = None
    if B == 1 and not args.no_relight and not args.no_extras:
        RS = 2048
        scr = synthetic.make_scene(wl.lo, RS, RS)
        tr = lambda x: torch.from_numpy(x).to(dev)
        n_r = ops.normals_from_depth(tr(scr.depth))
        T_r = ops.shade_transfer(tr(scr.albedo), tr(scr.roughness), tr(scr.metallic), n_r, args.spp)
        L_r = torch.randn(360, 25, 3, device=dev) * 0.1
        L_r[:, 0] += 3.5
        FPP = 24                                       # lights per pass (matpbr_relight's chunk): 360 frames = 15 passes
        out_r = torch.empty(FPP, RS, RS, 3, device=dev)
        ops.relight(T_r, L_r[:FPP].contiguous(), RS, RS, out_r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n_pass = 0
        for f0 in range(0, 360, FPP):
            ops.relight(T_r, L_r[f0:f0 + FPP], RS, RS, out_r[: min(FPP, 360 - f0)])
            n_pass += 1
        e1.record()
        torch.cuda.synchronize()
        ms_r = e0.elapsed_time(e1)
        bytes_r = (n_pass * 300 + 360 * 12) * RS * RS  # the transfer read once per pass (300 B/px) + one rgb write per frame
        relight = {"bound": "hbm", "kernel": f"relight_kernel (2048x2048, {FPP} lights per pass; 8 per pass in rounds 2-3: 26.6 k frames/s)",
                   "achieved": bytes_r / (ms_r * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                   "unit": "GB/s", "frames_per_s": 360 / (ms_r * 1e-3), "avg_launch_ms": ms_r / n_pass, "bytes_per_pixel_per_pass": 300 + FPP * 12}
        relight["frac"] = relight["achieved"] / relight["peak"]
        del T_r, out_r, L_r

    if rank == 0:
        out = {
            "metric": "opt_iterations_per_sec_512x512", "value": value, "unit": "it/s", "n_gpus": world,
            "device": (lambda pr: {"name": pr.name, "compute_units": pr.multi_processor_count, "memory_gb": round(pr.total_memory / 2 ** 30, 1),
                                   "gcn_arch": getattr(pr, "gcnArchName", None)})(torch.cuda.get_device_properties(dev)),
            "library": os.environ.get("MATPBR_LIB") or "materialist_amd/libmatpbr.so", "csrc_sha16": __import__("materialist_amd.build", fromlist=["sources_digest"]).sources_digest(),
            "world_size": dist.get_world_size() if use_dist else 1, "collective_backend": dist.get_backend() if use_dist else None,
            "steps": args.steps, "warmup": args.warmup, "device_warmup": device_warmup, "cold_first_process_it_per_s": cold, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] synthetic: {B}x{H}x{W}/GPU, --model_name {'pos_mlp' if mode == 'pos_mlp' else 'none'} --opt_order 'rm a' part rm, spp {args.spp}",
                       "workload_detail": f"C2-synthetic (BASELINE configs[1]): {H}x{W}, one epoch of hot loop B, part 'rm' of --opt_order 'rm a', "
                                   f"{'--model_name pos_mlp' if mode == 'pos_mlp' else '--model_name none'} "
                                   f"(maps -> render -> gamma-2.2 MSE/L1 loss -> backward -> Adam(W)), spp={args.spp}, geometric normals, SH25 light",
                       "parallelism": f"dp{world} (independent images, no data-path collective)",
                       "mode": mode, "mode_requested": args.mode, "height": H, "width": W, "spp": args.spp, "images_per_gpu": B, "light": "SH25",
                       "phase": type(phase).__name__ + (" (two groups of images on streams of their own: the same results as one phase, bit for bit)"
                                                        if type(phase).__name__ == "PipelinedBrdfPhase" else ""),
                       "mlp_products": int(_posmlp._PosMlpHipFn.PRODUCTS),
                       "mlp_arithmetic": ("f32 results.  Each f32 operand of a 256-wide product is carried as TWO round-to-nearest f16 pieces (a rounded piece leaves a "
                                          "signed remainder: two pieces reach 2^-24 of the operand, where bf16 needs three) and three products p1 q1 + p1 q2 + p2 q1 are formed "
                                          "on v_mfma_f32_32x32x16_f16 with f32 accumulation; the weights are cut as 256 w, the loss gradients of the backward products under "
                                          "one power-of-two exponent per 128-row tile; the first layer (K = 15) on v_mfma_f32_32x32x2_f32.  Against fp64 the error of every "
                                          "product is that of the exact-f32 MFMA kernels and of round 4's three-bf16-piece form (the f32 accumulation dominates all three): "
                                          "tests/test_gpu_parity.py::test_two_piece_f16_forward_layers_are_f32_accurate, ::test_block_scaled_f16_backward_products, "
                                          "::test_forward_chain_is_the_layer_by_layer_forward; modes.pos_mlp_bf16x3 and modes.pos_mlp_exact_f32 run the other two forms"
                                          if _posmlp._PosMlpHipFn.PRODUCTS else "exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32)")},
            "ranks": [{"rank": r, "it_per_s": args.steps * B / t} for r, t in enumerate(per_rank)],
            "modes": modes,
            "psnr_db": {"initial_guess": psnr0, "after_timed_steps": psnr1, "vs": "own HIP render of the synthetic ground truth (Mitsuba cannot run, SURVEY F3)"},
        }
        if pinned is not None:
            out["rank_core_pinning"] = pinned
        if host_enqueue is not None:
            out["host_enqueue"] = host_enqueue
        if roof is not None:
            roof["posmlp_gemm"], roof["relight"] = gemm, relight
            out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
        write_detail(out)
        print(compact_line(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
