#!/bin/bash
# A/B of environment knobs on one build, interleaved: scripts/ab_env.sh <rounds> "<NAME=VAL ...>|<NAME=VAL ...>|..." <bench.py arguments ...>
# (an empty variant "" is the default build)
rounds=$1; IFS='|' read -ra variants <<< "$2"; shift 2
for round in $(seq 1 $rounds); do
  for v in "${variants[@]}"; do
    env $v python3 bench.py --no-cpu-baseline --no-pcie "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_seconds_last_step']; print(repr(sys.argv[1]), d['value'], d['ms_per_step'], {k: round(v,4) for k,v in s.items()}, d['pair_transforms_crc32'], d.get('host_cpu',{}).get('stream_waits_per_step'))" "$v"
  done
done
