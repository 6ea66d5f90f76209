#!/bin/bash
# A/B on one box: the library built from the working tree against materialist_amd/libmatpbr_base.so (built from another revision)
cp materialist_amd/libmatpbr.so /tmp/lib_new.so
for round in 1 2; do
  for v in new base; do
    if [ $v = base ]; then cp materialist_amd/libmatpbr_base.so materialist_amd/libmatpbr.so; else cp /tmp/lib_new.so materialist_amd/libmatpbr.so; fi
    echo "== $v (round $round)"
    bash tools/layer_prof.sh
    python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s', d['value'], 'ms', d['ms_per_step'])"
  done
done
cp /tmp/lib_new.so materialist_amd/libmatpbr.so
