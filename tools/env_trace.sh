#!/bin/bash
# kernel trace of hot loop A (tools/env_profile.py), or of a part that moves the normal map: per-kernel durations of one iteration and the gaps between them.
#   usage: tools/env_trace.sh [mlp|texels|envmlp|normal]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_e -o t -- python3 tools/env_profile.py 300 graph ${1:-texels} 2>&1 | tail -2
ENV_TRACE_KIND=${1:-texels} python - <<'PY'
import csv, glob
path = glob.glob("gpurun_out/tr_e/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
import os
anchor = "shade_kernel" if os.environ.get("ENV_TRACE_KIND") == "normal" else "env_prt"
k = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = k[len(k) // 2]
n = k[len(k) // 2 + 1] - i0
for r0, r1 in zip(rows[i0 - 1:i0 + n], rows[i0:i0 + n + 1]):
    print("%-36s dur %6.1f us   gap before %5.1f us  grid %s wg %s" % (r1["Kernel_Name"][:36], (int(r1["End_Timestamp"]) - int(r1["Start_Timestamp"])) / 1e3,
          (int(r1["Start_Timestamp"]) - int(r0["End_Timestamp"])) / 1e3, r1.get("Grid_Size_X", "?"), r1.get("Workgroup_Size_X", "?")))
PY
rm -rf gpurun_out/tr_e
