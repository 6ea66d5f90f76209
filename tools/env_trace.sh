#!/bin/bash
# kernel trace of hot loop A (tools/env_profile.py): per-kernel durations and the gaps between them.  usage: tools/env_trace.sh [mlp|texels]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_e -o t -- python3 tools/env_profile.py 300 graph ${1:-texels} 2>&1 | tail -2
python - <<'PY'
import csv, glob
path = glob.glob("gpurun_out/tr_e/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
k = [i for i, r in enumerate(rows) if "env_prt" in r["Kernel_Name"] or "env_texel_iter" in r["Kernel_Name"]]
i0 = k[len(k) // 2]
n = k[len(k) // 2 + 1] - i0
for r0, r1 in zip(rows[i0 - 1:i0 + n], rows[i0:i0 + n + 1]):
    print("%-36s dur %6.1f us   gap before %5.1f us  grid %s wg %s" % (r1["Kernel_Name"][:36], (int(r1["End_Timestamp"]) - int(r1["Start_Timestamp"])) / 1e3,
          (int(r1["Start_Timestamp"]) - int(r0["End_Timestamp"])) / 1e3, r1.get("Grid_Size_X", "?"), r1.get("Workgroup_Size_X", "?")))
PY
rm -rf gpurun_out/tr_e
