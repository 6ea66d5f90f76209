#!/bin/bash
# builds a stamped library beside the product one and prints the stamps (the product library is restored afterwards)
cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<'PY'
import subprocess
from materialist_amd import build as b
import os
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_BX_STAMPS", "-o", b.LIB_PATH, *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True)
PY
if [ "$1" = gx ]; then python tools/gx_stamps.py fwd; python tools/gx_stamps.py bwd; else python tools/bx_stamps.py fwd; python tools/bx_stamps.py bwd; fi
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
