run() { python3 bench.py --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_seconds_last_step'], d['host_cpu'])" "$1"; }
run base
run base2
MM3D_SIFT_NO_IDENTITY=1 run no_identity
MM3D_SIFT_NO_FUSED_NORMALS=1 run no_fused
MM3D_SIFT_NO_IDENTITY=1 MM3D_SIFT_HIL_FACTOR=0 run no_identity_old_items
MM3D_FEATURE_WORKERS=16 run fw16
MM3D_FEATURE_WORKERS=8 run fw8
MM3D_FEATURE_WORKERS=6 run fw6
