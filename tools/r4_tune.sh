#!/bin/bash
# usage: r4_tune.sh <env assignments...> : a library built with -DMATPBR_EXP_TUNE on the box, the none-mode loop traced once per assignment
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<PY
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_EXP_TUNE", "-o", "materialist_amd/libmatpbr.so", *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
PY
for kv in "$@"; do
  echo "#### $kv"
  env $kv bash tools/r4_probe2.sh
done
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
