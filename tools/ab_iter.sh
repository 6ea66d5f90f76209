#!/bin/bash
# A/B of the headline iteration only: working-tree library against materialist_amd/libmatpbr_base.so, alternating, on one box
cp materialist_amd/libmatpbr.so /tmp/lib_new.so
for round in 1 2 3 4; do
  for v in new base; do
    if [ $v = base ]; then cp materialist_amd/libmatpbr_base.so materialist_amd/libmatpbr.so; else cp /tmp/lib_new.so materialist_amd/libmatpbr.so; fi
    python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')"
  done
done
cp /tmp/lib_new.so materialist_amd/libmatpbr.so
