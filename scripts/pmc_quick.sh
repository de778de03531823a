#!/bin/bash
# SQ counters of the kernels whose names match $1 on two headline maps and one pair (scripts/pmc_driver.py), old and new library side by
# side: scripts/pmc_quick.sh <kernel regex> [lib suffix ...]   (run on the GPU box; prints one line per library and kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "${@:-base}"; do
  [ "$v" = base ] && lib=$R/map-merge_amd/libmm3d.so || lib=$R/map-merge_amd/libmm3d_$v.so
  rm -rf /tmp/pmcq_$v
  MM3D_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES \
      --output-format csv -d /tmp/pmcq_$v -- python3 $R/scripts/pmc_driver.py pair > /tmp/pmcq_$v.log 2>&1
  python3 $R/scripts/pmc_summary.py /tmp/pmcq_$v/*/*counter_collection.csv | python3 -c "
import sys, csv, re
rows = list(csv.DictReader(sys.stdin))
for r in rows:
    if re.search(sys.argv[1], r['kernel']):
        print(sys.argv[2], r['kernel'], 'dispatches', r['dispatches'], ' '.join(f'{k}={r[k]}' for k in r if k.startswith('SQ_')))
" "$pat" "$v"
done
