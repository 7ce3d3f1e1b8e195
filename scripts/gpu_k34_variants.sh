#!/bin/bash
# development: one C4 unit (10 Mbp x 10 Mbp) through the fused kernel in its workgroup shapes
for c in 2 3 4 1; do
  echo "== MIMEO_K34_CFG=$c"
  MIMEO_K34_CFG=$c timeout -k 10 300 python scripts/dev_unit.py 1e7 1000 2>&1 | tail -1 | sed 's/.*ms_index/ms_index/'
  MIMEO_K34_CFG=$c timeout -k 10 300 python scripts/dev_unit.py 5e6 50 2>&1 | tail -1 | sed 's/.*ms_index/ms_index/'
done
