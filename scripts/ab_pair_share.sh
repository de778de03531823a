# How many of the pairs that can start right now a worker takes as one batch (capi.cpp::claim_pairs): take = min(cap, avail / (share * S)).
run() { python3 bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_seconds_last_step']['t_features'], d['pair_transforms_crc32'])" "$1"; }
for r in 1 2; do
MM3D_PAIR_SHARE=2 run share2
MM3D_PAIR_SHARE=0.25 run share0.25
MM3D_PAIR_SHARE=0.125 run share0.125
MM3D_PAIR_SHARE=0.0625 run share0.0625
MM3D_PAIR_SHARE=0.0625 MM3D_PAIR_BATCH=32 run share0.0625_cap32
MM3D_PAIR_SHARE=0.125 MM3D_PAIR_BATCH=8 run share0.125_cap8
done
