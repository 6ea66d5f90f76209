#!/bin/bash
# kernel trace of the reference loop body unchanged on the operator face (bench.py --mode torch: loop.BrdfPhase -> render.render_w_brdf with the scene's
# per-(light, normals) cache): per-kernel durations.   usage: tools/op_face_trace.sh [images per GPU]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_o -o t -- python3 bench.py --images-per-gpu ${1:-8} --mode torch --no-extras --no-cpu-baseline --steps 60 --warmup 10 > gpurun_out/op_face.json 2> gpurun_out/op_face.err
python tools/summarize_rocprof.py gpurun_out/tr_o | head -24 | cut -c1-150
rm -rf gpurun_out/tr_o
