#!/bin/bash
# A/B of the none-mode loop rates: working-tree library against materialist_amd/libmatpbr_base.so, alternating, on one box
cp materialist_amd/libmatpbr.so /tmp/lib_new.so
for round in 1 2; do
  for v in new base; do
    if [ $v = base ]; then cp materialist_amd/libmatpbr_base.so materialist_amd/libmatpbr.so; else cp /tmp/lib_new.so materialist_amd/libmatpbr.so; fi
    python bench.py --no-cpu-baseline --no-relight --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k: round(v['it_per_s']) for k, v in d['modes'].items() if k.startswith('fused')}, round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
cp /tmp/lib_new.so materialist_amd/libmatpbr.so
