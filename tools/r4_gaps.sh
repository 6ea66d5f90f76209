#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_g -o t -- python3 bench.py --images-per-gpu ${1:-1} --mode fused --no-extras --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/probe_g.json 2> gpurun_out/probe_g.err
python - <<'PY'
import csv, glob
path = glob.glob("gpurun_out/tr_g/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
k = [i for i, r in enumerate(rows) if "lazy_pstep" in r["Kernel_Name"]]
i0 = k[len(k) // 2]
for r0, r1 in zip(rows[i0 - 1:i0 + 9], rows[i0:i0 + 10]):
    print("%-28s dur %6.1f us   gap before %5.1f us  grid %s wg %s" % (r1["Kernel_Name"][:28], (int(r1["End_Timestamp"]) - int(r1["Start_Timestamp"])) / 1e3,
          (int(r1["Start_Timestamp"]) - int(r0["End_Timestamp"])) / 1e3, r1.get("Grid_Size_X", "?"), r1.get("Workgroup_Size_X", "?")))
PY
rm -rf gpurun_out/tr_g
