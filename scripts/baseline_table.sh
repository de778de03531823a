#!/bin/bash
# BASELINE.md's results table: the five BASELINE.json configurations on one GPU, each with the CPU baselines
# B1 (one thread) and B2 (all cores) timed in the same run on the same box.  Run through gpurun:
#   scripts/baseline_table.sh r02        -> gpurun_out/r02_table_cfg{1..5}.json
# scripts/fill_baseline_table.py r02 then writes the table into BASELINE.md and copies the lines to profiles/.
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out
run() {   # run <cfg> <bench args...>       (ONLY="5 1": just those rows)
  local cfg=$1; shift
  if [ -n "${ONLY:-}" ] && [[ " $ONLY " != *" $cfg "* ]]; then return; fi
  timeout 1500 python3 bench.py "$@" 2> "/tmp/table_cfg$cfg.err" | tail -1 > "gpurun_out/${TAG}_table_cfg$cfg.json"
  python3 -c "import json; d=json.load(open('gpurun_out/${TAG}_table_cfg$cfg.json')); print('cfg$cfg', d['value'], (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline_all_cores') or {}).get('value'), (d.get('parity_check') or {}).get('ok'), d.get('icp_iterations_histogram'), d.get('gt_error', {}).get('recovered_within_0.5'))" \
    || { echo "cfg$cfg failed:"; tail -5 "/tmp/table_cfg$cfg.err"; }
}
# (steps of 4 and 14 ms: enough of them that one late host thread does not show -- five steps of row 2 gave 358 ... 435)
run 1 --maps 2 --points 10000 --steps 30 --warmup 3
run 2 --maps 4 --points 200000 --steps 15 --warmup 3
run 3
run 5 --maps 64 --points 50000 --steps 5 --warmup 1
run 4 --maps 8 --points 2000000 --descriptor SHOT --steps 2 --warmup 1
# configs[3] as SURVEY 8d words it: dense indoor -- 30 m windows, resolution 0.05 (the radii keep the reference's defaults)
run 4indoor --maps 8 --points 2000000 --descriptor SHOT --window 30 --resolution 0.05 --steps 1 --warmup 1
# the 'lattice' scene family with enough SAC-IA hypotheses for the algorithm to find the basin: ICP iterates, gt_error is small
run 2lattice --maps 4 --points 200000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 3 --warmup 1
# the headline size on the 'lattice' family with 20 000 hypotheses: poses worth refining, so "Mpoints/s (ICP)" is measured on an ICP
# that iterates (CPU baselines + parity_check on a sample of two maps and ONE pair: a pair with 20 000 hypotheses is ~ 50 s of one core)
# (four warm-up steps: a step moves 1.26 GB of hypothesis errors per pair through the memory pools, which keep growing through the
# first three -- with two warm-up steps the row scattered between 131 and 188 map-pairs/s, with four it is 186 - 187)
run 3lattice --maps 16 --points 500000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 3 --warmup 4
