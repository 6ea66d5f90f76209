#!/bin/bash
# as tools/ab.sh for measurement builds of matpbr_kernels.hip (the render / none-mode loop kernels).  usage: bash tools/ab_k.sh "<command>" "<-D flags 1>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
B=materialist_amd/_build
CMD=$1; shift
echo "== product"; bash -c "$CMD" 2>&1 | tail -2
i=0
for FL in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc $FL -c materialist_amd/csrc/matpbr_kernels.hip -o /tmp/mk_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libmatpbr_k$i.so /tmp/mk_v$i.o $B/posmlp_kernels.o $B/posmlp_chain.o $B/mesh_host.o || exit 1
  echo "== $FL"; MATPBR_LIB=/tmp/libmatpbr_k$i.so bash -c "$CMD" 2>&1 | tail -2
done
