#!/bin/bash
# measurement builds of matpbr_kernels.hip BESIDE the product library: what each round-6 addition to the walk costs.  usage (on the GPU box): bash tools/ab_walk.sh "<-D flags>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B=materialist_amd/_build
run() {  # label, library
  MATPBR_LIB=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_ab -o t -- python3 bench.py --images-per-gpu 8 --mode fused_one_phase --no-extras --no-cpu-baseline --steps 1000 --warmup 300 > /dev/null 2>&1
  echo "== $1"; python tools/summarize_rocprof.py gpurun_out/tr_ab | sed -n 3,5p | cut -c1-100; rm -rf gpurun_out/tr_ab
}
run product materialist_amd/libmatpbr.so
i=0
for FL in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc $FL -c materialist_amd/csrc/matpbr_kernels.hip -o /tmp/mk_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libmatpbr_k$i.so /tmp/mk_v$i.o $B/posmlp_kernels.o $B/posmlp_chain.o $B/mesh_host.o || exit 1
  run "$FL" /tmp/libmatpbr_k$i.so
done
