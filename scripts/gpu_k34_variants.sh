#!/bin/bash
# development: one C4 unit (10 Mbp x 10 Mbp) through the fused kernel with parts switched off
for c in 1 2; do for d in 0 2 1 3; do
  echo "== MIMEO_K34_CFG=$c MIMEO_K34_DEBUG=$d"
  MIMEO_K34_CFG=$c MIMEO_K34_DEBUG=$d timeout -k 10 300 python scripts/dev_unit.py 1e7 1000 2>&1 | tail -1 | sed 's/.*ms_index/ms_index/'
done; done
