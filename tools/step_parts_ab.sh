#!/bin/bash
# what the step kernel's time is made of: the same loop with parts compiled out (wrong results, valid durations)
cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
for def in MATPBR_EXP_NORESAMPLE_RT MATPBR_EXP_NORESAMPLE; do
python - <<PY
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-D$def", "-o", "/tmp/lib_$def.so", *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True)
PY
done
for round in 1 2; do
  for v in keep MATPBR_EXP_NORESAMPLE_RT MATPBR_EXP_NORESAMPLE; do
    cp /tmp/lib_$v.so materialist_amd/libmatpbr.so
    python bench.py --no-cpu-baseline --no-relight --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k: round(v['it_per_s']) for k, v in d['modes'].items() if k.startswith('fused') and 'exact' not in k}, round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
