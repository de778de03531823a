# Feature workers / streams of the library call on the headline (capi.cpp::estimate_maps_streams), after the pair batches grew.
run() { python3 bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_seconds_last_step']['t_features'], d['host_cpu']['cores_busy'])" "$1"; }
for r in 1 2; do
run default
MM3D_FEATURE_WORKERS=4 run fw4
MM3D_FEATURE_WORKERS=8 run fw8
MM3D_FEATURE_WORKERS=10 run fw10
MM3D_FEATURE_WORKERS=16 run fw16
run s12 --streams 12
run s20 --streams 20
run s24 --streams 24
MM3D_FEATURE_WORKERS=8 run s24_fw8 --streams 24
done
