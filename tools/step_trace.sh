#!/bin/bash
# per-launch durations of the step kernel in the 8-image none-mode loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_step -o t -- python3 bench.py --images-per-gpu ${1:-8} --mode fused --no-extras --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/step_trace.json 2> gpurun_out/step_trace.err
python tools/step_durations.py gpurun_out/tr_step lazy_step_kernel 200 80
python tools/step_durations.py gpurun_out/tr_step lazy_resample 200 40
rm -rf gpurun_out/tr_step
