#!/bin/bash
# end-of-round measurements in one gpurun call: GPU tests, traces and counter passes, then the default bench line (which reads the
# per-launch traffic the counter passes of THIS build produced); everything lands in gpurun_out/ and is copied to profiles/ by hand
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r03_gpu_tests.txt
bash tools/trace_pos_mlp.sh r03 > /dev/null 2>&1
bash tools/pmc_passes_r03.sh > gpurun_out/r03_pmc.log 2>&1
python tools/pmc_to_traffic.py gpurun_out --write > /dev/null && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -2 gpurun_out/r03_gpu_tests.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["traffic"], r["own_traffic_frac"], d["cpu_baseline"]["value"])
print({k: round(v["it_per_s"]) for k, v in d["modes"].items()})
PY
