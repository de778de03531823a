run() { python3 bench.py --no-cpu-baseline --no-pcie "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['pair_transforms_crc32'])" "$1"; }
for r in 1 2; do
run cfg5_new --maps 64 --points 50000 --steps 3 --warmup 1
MM3D_PAIR_SHARE=2 run cfg5_old --maps 64 --points 50000 --steps 3 --warmup 1
run cfg2_new --maps 4 --points 200000 --steps 5 --warmup 1
MM3D_PAIR_SHARE=2 run cfg2_old --maps 4 --points 200000 --steps 5 --warmup 1
run lat2_new --maps 4 --points 200000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 3 --warmup 1
MM3D_PAIR_SHARE=2 run lat2_old --maps 4 --points 200000 --scenes lattice --overlap-step 0.25 --sac-iterations 20000 --steps 3 --warmup 1
run cfg4_new --maps 8 --points 2000000 --descriptor SHOT --steps 2 --warmup 1
MM3D_PAIR_SHARE=2 run cfg4_old --maps 8 --points 2000000 --descriptor SHOT --steps 2 --warmup 1
done
