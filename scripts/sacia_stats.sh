cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() { name=$1; shift; MM3D_SACIA_STATS=1 MM3D_BENCH_NO_ISOLATED=1 python3 bench.py "$@" --no-cpu-baseline --no-pcie --no-pair-stage 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['config']['workload'], '->', {k: v for k, v in d['sacia_certified'].items() if k != 'what'}, 'crc', d.get('pair_transforms_crc32'))"; }
run cfg1 --maps 2 --points 10000 --steps 3 --warmup 1
run cfg2 --maps 4 --points 200000 --steps 2 --warmup 1
run cfg3 --steps 2 --warmup 1
run cfg5 --maps 64 --points 50000 --steps 1 --warmup 1
run cfg2lattice --maps 4 --points 200000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 1 --warmup 1
run cfg3lattice --maps 16 --points 500000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 1 --warmup 1
run cfg4 --maps 8 --points 2000000 --descriptor SHOT --steps 1 --warmup 1
run cfg4indoor --maps 8 --points 2000000 --descriptor SHOT --window 30 --resolution 0.05 --steps 1 --warmup 1
