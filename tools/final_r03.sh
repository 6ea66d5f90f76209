#!/bin/bash
# end-of-round measurements: GPU tests, the default bench line, the traces and counter passes that profiles/ keeps
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r03_gpu_tests.txt
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
bash tools/trace_pos_mlp.sh r03 > /dev/null 2>&1
bash tools/pmc_passes_r03.sh > gpurun_out/r03_pmc.log 2>&1
tail -2 gpurun_out/r03_gpu_tests.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["cpu_baseline"]["value"])
print({k: round(v["it_per_s"]) for k, v in d["modes"].items()})
PY
