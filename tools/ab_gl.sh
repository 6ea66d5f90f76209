#!/bin/bash
# A/B of the headline iteration on one box: LDS-DMA main loop of the layer kernels (1) against the register-staged one (0), alternating
for round in 1 2 3; do
  for v in ${MODES:-2 1 0}; do
    python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 --mlp-lds-dma $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds_dma=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')"
  done
done
