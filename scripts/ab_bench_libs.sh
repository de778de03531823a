#!/bin/bash
# A/B of library builds on the headline bench, interleaved so that drift between runs shows: scripts/ab_bench_libs.sh "<suffix> <suffix> ..." [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in $(seq 1 ${2:-2}); do
  for v in $1; do
    [ "$v" = base ] && lib=$R/map-merge_amd/libmm3d.so || lib=$R/map-merge_amd/libmm3d_$v.so
    MM3D_LIB=$lib python3 bench.py --no-cpu-baseline --no-pcie --steps ${STEPS:-3} --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_seconds_last_step']['t_features'], d['pair_transforms_crc32'])" "$v"
  done
done
