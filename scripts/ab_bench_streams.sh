run() { python3 bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], d['value'], d['ms_per_step'], d['stage_seconds_last_step']['t_features'], d['host_cpu']['cores_busy'])" "$@"; }
run --streams 16
run --streams 20
run --streams 24
run --streams 32
run --streams 16
MM3D_FEATURE_WORKERS=16 run --streams 24
MM3D_FEATURE_WORKERS=12 run --streams 24
GPU_MAX_HW_QUEUES=8 run --streams 16
GPU_MAX_HW_QUEUES=6 run --streams 24
