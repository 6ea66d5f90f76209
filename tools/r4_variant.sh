#!/bin/bash
# usage: r4_variant.sh "<-D flags>" : builds the library with extra flags on the box, traces the none-mode loop, restores the library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
for flags in "$@"; do
python - <<PY
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, *"$flags".split(), "-o", "materialist_amd/libmatpbr.so", *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
PY
echo "#### variant: $flags"
bash tools/r4_probe2.sh
done
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
