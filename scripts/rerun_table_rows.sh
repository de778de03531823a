cd $GRAFT_REPO_ROOT
run() { local cfg=$1; shift; timeout 1500 python3 bench.py "$@" 2> "/tmp/table_cfg$cfg.err" | tail -1 > "gpurun_out/r03_table_cfg$cfg.json"; python3 -c "import json; d=json.load(open('gpurun_out/r03_table_cfg$cfg.json')); print('cfg$cfg', d['value'], d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'], json.dumps(d.get('parity_check'))[:900])" || tail -5 "/tmp/table_cfg$cfg.err"; }
run 2lattice --maps 4 --points 200000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 3 --warmup 1
run 3
run 4indoor --maps 8 --points 2000000 --descriptor SHOT --window 30 --resolution 0.05 --steps 1 --warmup 1
