#!/bin/bash
# per-launch durations of the none-mode loop (8 and 1 images; parts 'rm' and 'a') on the working-tree library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
trace() {  # $1 images, $2 tag
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$2 -o t -- python3 bench.py --images-per-gpu $1 --mode fused --no-extras --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/probe_$2.json 2> gpurun_out/probe_$2.err
  echo "== $2"
  for k in lazy_pstep lazy_pwalk lazy_step_kernel lazy_resample loss_sums2; do python tools/step_durations.py gpurun_out/tr_$2 $k 200 0 2>/dev/null | head -1 | sed "s/^/$k: /"; done
  python -c "import json;d=json.loads(open('gpurun_out/probe_$2.json').read().strip().splitlines()[-1]);print('it/s',round(d['value']),'ms/step',d['ms_per_step'])"
  rm -rf gpurun_out/tr_$2
}
trace 8 b8; trace 1 b1
