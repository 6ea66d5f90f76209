#!/bin/bash
# the headline loop alone, twice (python bench.py --no-extras --no-cpu-baseline): value, ms per iteration.  MATPBR_LIB selects the library (tools/ab.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), end='   ')"; done; echo
