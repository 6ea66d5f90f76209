#!/bin/bash
# end-of-round measurements in one gpurun call: GPU tests, traces and counter passes, then the default bench line (which reads the
# per-launch traffic the counter passes of THIS build produced); everything lands in gpurun_out/ and is copied to profiles/ by hand
cd "$GRAFT_REPO_ROOT" || exit 1
MATPBR_TOLERANCE_REPORT=gpurun_out/r06_tolerances.tsv timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06_gpu_tests.txt
timeout 400 bash tools/trace_pos_mlp.sh r06 > /dev/null 2>&1
bash tools/pmc_passes_r06.sh > gpurun_out/r06_pmc.log 2>&1
PMC_ROUND=r06 python tools/pmc_to_traffic.py gpurun_out --write > /dev/null && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
for k in texels envmlp normal; do timeout 200 bash tools/env_trace.sh $k > gpurun_out/r06_iteration_$k.txt 2>&1; done
timeout 200 bash tools/op_face_trace.sh 8 > gpurun_out/r06_trace_operator_face_b8.csv 2>&1
timeout 600 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err; cp bench_detail.json gpurun_out/r06_bench.json
for s in indoor2:none indoor2:pos_mlp jinjya:none; do
  timeout 300 python tools/real_image.py --sample ${s%%:*} --model_name ${s##*:} --out /tmp/real_image > /dev/null 2>&1
done
cp /tmp/real_image/real_image_*.json gpurun_out/ 2>/dev/null
tail -2 gpurun_out/r06_gpu_tests.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["traffic"], r["own_traffic_frac"], d["cpu_baseline"]["value"])
# the perf gate (thresholds live here, not in pytest: the boxes of the pool are not all alike)
assert d["value"] >= 600, d["value"]
assert r["frac"] >= 0.45, r["frac"]
print({k: round(v["it_per_s"]) for k, v in d["modes"].items()})
PY
