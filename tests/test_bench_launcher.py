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


def test_bench_line_is_compact_and_carries_the_contract_keys():
    """The driver parses the LAST stdout line: round 5's 20 KB line came back `parsed: null`.  The line is assembled by bench.compact_line from the
    full record (which goes to bench_detail.json); it must stay under 4 KB whatever the record holds, and keep the contract's keys."""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))           # a real (20 KB) record of the same shape
    full["roofline"]["note"] = "x" * 50_000                                            # prose of any length never reaches the line
    full["config"]["workload"] = "w" * 5_000
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT == 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(("workload", "mode", "height", "width", "spp", "images_per_gpu")) <= set(line["config"]) and len(line["config"]["workload"]) <= 120
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "bytes_per_pixel")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["value"] == float(f"{full['value']:.6g}") and line["roofline"]["frac"] == float(f"{full['roofline']['frac']:.6g}")
    # a record with hundreds of modes still yields a parseable line (the optional blocks are dropped first)
    full["modes"] = {f"mode_{i}": {"it_per_s": 1.0 + i} for i in range(600)}
    assert len(bench.compact_line(full)) < 4096
