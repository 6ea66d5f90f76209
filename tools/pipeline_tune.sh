#!/bin/bash
# tools/pipeline_ab.py on a library built with -DMATPBR_EXP_TUNE, once per cap of the step kernel's workgroups (does a step that leaves room on the
# CUs let another group's walk and statistics run under it?)   usage: tools/pipeline_tune.sh 1024 768 512
cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<PY
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_EXP_TUNE", "-o", "materialist_amd/libmatpbr.so", *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
PY
for cap in "$@"; do
  echo "#### MATPBR_PSTEP_WGS=$cap"
  MATPBR_PSTEP_WGS=$cap timeout 300 python tools/pipeline_ab.py rm 2>&1 | grep "^part"
done
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
