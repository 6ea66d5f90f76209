#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B=materialist_amd/_build
python tools/walk_count.py 2>/dev/null | tail -1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -c _base/materialist_amd/csrc/matpbr_kernels.hip -o /tmp/mk_base.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libmatpbr_base.so /tmp/mk_base.o $B/posmlp_kernels.o $B/posmlp_chain.o $B/mesh_host.o || exit 1
MATPBR_LIB=/tmp/libmatpbr_base.so python tools/walk_count.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MATPBR_LIB=/tmp/libmatpbr_base.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_ab -o t -- python3 bench.py --images-per-gpu 8 --mode fused_one_phase --no-extras --no-cpu-baseline --steps 1000 --warmup 300 > /dev/null 2>&1
echo "== base (round 5 kernels)"; python tools/summarize_rocprof.py gpurun_out/tr_ab | sed -n 3,5p | cut -c1-100; rm -rf gpurun_out/tr_ab
