#!/bin/bash
# per-kernel durations of the 256-wide layer kernels at 512 x 512 (tools/layer_prof.py), under rocprofv3 --kernel-trace;
# first with the LDS-DMA main loop (default), then with the register-staged one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for GL in 2 1 0 2 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lp -o t -- python3 tools/layer_prof.py $GL > /dev/null 2> gpurun_out/lp.err
  echo "lds_dma=$GL"; python tools/summarize_rocprof.py gpurun_out/lp | grep -E "mlp_nt|wgrad_bx" | cut -c1-150
  rm -rf gpurun_out/lp
done
