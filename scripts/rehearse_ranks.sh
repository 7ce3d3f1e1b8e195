#!/bin/bash
# Rehearsal of the N > 1 code path on ONE GPU: N ranks (gloo) share device 0 and run the row-mode bench at test size
# (c4small: 6 scaffolds x 1 Mbp).  Not a scaling number — it shows that N processes issue their rows concurrently,
# gather the same records as one rank, and what host time a unit costs when N processes drive the device.
N=${1:-6}
STEPS=${2:-1}
PORT=$((20000 + RANDOM % 20000))
export MIMEO_DIST_BACKEND=gloo MIMEO_FORCE_DEVICE=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --workload c4small --steps $STEPS --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stage_ms_per_step_rank0']
gpu = s['ms_index'] + s['ms_scan'] + s['ms_extend'] + s['ms_chain'] + s['ms_gapped']
print(json.dumps({'ranks': d['n_gpus'], 'ms_per_step': d['ms_per_step'], 'engine_call_ms_rank0': s['ms_total'], 'device_stage_ms_rank0': round(gpu, 2),
                  'units_per_step_rank0': d['config']['pair_strands_rank0'], 'result': d['result'], 'value': d['value']}))"
