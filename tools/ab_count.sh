#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B=materialist_amd/_build
i=0
for FL in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc $FL -c materialist_amd/csrc/matpbr_kernels.hip -o /tmp/mk_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /tmp/libmatpbr_k$i.so /tmp/mk_v$i.o $B/posmlp_kernels.o $B/posmlp_chain.o $B/mesh_host.o || exit 1
  echo "== $FL"; MATPBR_LIB=/tmp/libmatpbr_k$i.so python tools/walk_count.py 2>/dev/null | tail -1 | cut -c1-140
done
