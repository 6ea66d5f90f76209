"""`bench.py --gpus N` really launches N ranks: the parent only spawns `torch.distributed.run` (it never touches the GPU) and the
ranks run the timing protocol (barrier + synchronize fences, MAX over ranks, per-rank rates gathered on rank 0).  Exercised here
with the CPU stand-in step over gloo; on the GPU box the same launcher starts one rank per MI355X over RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launcher_starts_two_ranks():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-cpu", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                       # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["it_per_s"] > 0 for r in out["ranks"])
    assert out["steps"] == 5 and out["warmup"] == 1


def test_bench_single_process_does_not_launch():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-cpu", "--steps", "3", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and len(out["ranks"]) == 1
