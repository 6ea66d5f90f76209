#!/bin/bash
# per-kernel durations of the 256-wide layer kernels at 512 x 512 (tools/layer_prof.py), under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lp -o t -- python3 tools/layer_prof.py > /dev/null 2> gpurun_out/lp.err
python tools/summarize_rocprof.py gpurun_out/lp | grep -E "mlp_nt|wgrad_bx" | cut -c1-150
rm -rf gpurun_out/lp
