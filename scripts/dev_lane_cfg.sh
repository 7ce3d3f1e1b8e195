# development aid: bench under several pipeline switches (bash scripts/dev_lane_cfg.sh "CFG1" "CFG2" ...)
run() { python bench.py --workload $1 --steps $2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('  ', sys.argv[1], round(l['ms_per_step'],1), l['stage_ms_per_step_rank0']['ms_extend'], round(l['roofline']['avg_launch_ms'],4))" $1; }
for cfg in "$@"; do
  echo "$cfg"; env $cfg bash -c "$(declare -f run); run c4 3; run c2 5"
done
