#!/bin/bash
# A/B of library builds on ANY bench configuration, interleaved: scripts/ab_bench_args.sh "<suffix> ..." <rounds> <bench.py arguments ...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
libs=$1; rounds=$2; shift 2
for round in $(seq 1 $rounds); do
  for v in $libs; do
    [ "$v" = base ] && lib=$R/map-merge_amd/libmm3d.so || lib=$R/map-merge_amd/libmm3d_$v.so
    MM3D_LIB=$lib python3 bench.py --no-cpu-baseline --no-pcie "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_seconds_last_step']; print(sys.argv[1], d['value'], d['ms_per_step'], {k: round(v,4) for k,v in s.items()}, d['pair_transforms_crc32'])" "$v"
  done
done
