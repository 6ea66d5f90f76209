#!/bin/bash
# measurement builds of the forward chain BESIDE the product library (MATPBR_LIB; the product .so is never touched):
#   usage: bash tools/chain_ab.sh "<-D flags of variant 1>" "<-D flags of variant 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
B=materialist_amd/_build
echo "== product"; python tools/chain_time.py 2>&1 | grep chain=True
i=0
for FL in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize $FL -c materialist_amd/csrc/posmlp_chain.hip -o /tmp/chain_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libmatpbr_v$i.so $B/matpbr_kernels.o $B/posmlp_kernels.o /tmp/chain_v$i.o $B/mesh_host.o || exit 1
  echo "== $FL"; MATPBR_LIB=/tmp/libmatpbr_v$i.so python tools/chain_time.py 2>&1 | grep chain=True
done
