#!/bin/bash
# round-4 baseline probe: per-launch durations of the none-mode loop (8 and 1 images), then the same with the statistics fold compiled out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
trace() {  # $1 images, $2 tag
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$2 -o t -- python3 bench.py --images-per-gpu $1 --mode fused --no-extras --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/probe_$2.json 2> gpurun_out/probe_$2.err
  echo "== $2"; python tools/step_durations.py gpurun_out/tr_$2 lazy_step_kernel 200 0 | head -2
  python tools/step_durations.py gpurun_out/tr_$2 lazy_resample 200 0 | head -1
  python tools/step_durations.py gpurun_out/tr_$2 loss_sums2 200 0 | head -1
  python -c "import json;d=json.loads(open('gpurun_out/probe_$2.json').read().strip().splitlines()[-1]);print('it/s',round(d['value']),'ms/step',d['ms_per_step'])"
  rm -rf gpurun_out/tr_$2
}
trace 8 b8; trace 1 b1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<PY
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_EXP_NOFOLD", "-o", "materialist_amd/libmatpbr.so", *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True)
PY
trace 8 b8_nofold; trace 1 b1_nofold
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
