#!/bin/bash
# Re-measures single rows of BASELINE.md's table after a change that only concerns them (same commands as
# scripts/baseline_table.sh):   scripts/rerun_table_rows.sh r03 4 4indoor
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=$1; shift
run() { local cfg=$1; shift; timeout 1500 python3 bench.py "$@" 2> "/tmp/table_cfg$cfg.err" | tail -1 > "gpurun_out/${TAG}_table_cfg$cfg.json"; python3 -c "import json; d=json.load(open('gpurun_out/${TAG}_table_cfg$cfg.json')); print('cfg$cfg', d['value'], d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'], (d.get('parity_check') or {}).get('ok'))" || tail -5 "/tmp/table_cfg$cfg.err"; }
for row in "$@"; do
  case $row in
    1) run 1 --maps 2 --points 10000 --steps 10 --warmup 2 ;;
    2) run 2 --maps 4 --points 200000 --steps 5 --warmup 1 ;;
    3) run 3 ;;
    5) run 5 --maps 64 --points 50000 --steps 2 --warmup 1 ;;
    4) run 4 --maps 8 --points 2000000 --descriptor SHOT --steps 2 --warmup 1 ;;
    4indoor) run 4indoor --maps 8 --points 2000000 --descriptor SHOT --window 30 --resolution 0.05 --steps 1 --warmup 1 ;;
    2lattice) run 2lattice --maps 4 --points 200000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 3 --warmup 1 ;;
  esac
done
