#!/bin/bash
# measurement builds of posmlp_kernels.hip BESIDE the product library (MATPBR_LIB).  usage: bash tools/ab.sh <timing script> "<-D flags 1>" ...
# (the product library is what build.py built from the tree; "== product" runs it first)
cd "$GRAFT_REPO_ROOT" || exit 1
B=materialist_amd/_build
T=$1; shift
echo "== product"; if [[ $T == *.sh ]]; then bash $T 2>&1 | tail -1; else python $T 2>&1 | tail -1; fi
i=0
for FL in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc $FL -c materialist_amd/csrc/posmlp_kernels.hip -o /tmp/pk_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libmatpbr_w$i.so $B/matpbr_kernels.o /tmp/pk_v$i.o $B/posmlp_chain.o $B/mesh_host.o || exit 1
  echo "== $FL"; if [[ $T == *.sh ]]; then MATPBR_LIB=/tmp/libmatpbr_w$i.so bash $T 2>&1 | tail -1; else MATPBR_LIB=/tmp/libmatpbr_w$i.so python $T 2>&1 | tail -1; fi
done
