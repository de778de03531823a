#!/bin/bash
# SQ counters of ONE workload (two passes: rocprofv3 --pmc with --kernel-trace only), on the GPU box through gpurun:
#   scripts/profile_counters.sh <tag> <name> [pmc_driver.py options ...]      -> gpurun_out/<tag>_sq_<name>.csv
# scripts/assemble_counters.py <tag> <name> [the same options] then writes profiles/<tag>_counters_<name>.json (here, with the
# library of this tree: do not rebuild in between).  bench.py uses a counters file only for a run of the same workload.
set -u
TAG=$1; NAME=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out
fail() { echo "profile_counters: $*" >&2; exit 1; }
run() {   # run <dir> <counters...> : up to two tries, fresh output directory each
  local dir=$1; shift
  for try in 1 2; do
    rm -rf "$dir"
    if rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$dir" -- python3 scripts/pmc_driver.py pair "${ARGS[@]}" > "$dir.log" 2>&1 \
       && ls "$dir"/*/*counter_collection.csv > /dev/null 2>&1; then return 0; fi
    echo "profile_counters: try $try failed ($dir)" >&2
  done
  return 1
}
ARGS=("$@")
run /tmp/cnt_sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS || fail "pass A"
run /tmp/cnt_sqB SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM || fail "pass B"
python3 scripts/pmc_summary.py /tmp/cnt_sqA/*/*counter_collection.csv /tmp/cnt_sqB/*/*counter_collection.csv > "gpurun_out/${TAG}_sq_${NAME}.csv" || fail "pmc_summary"
[ -s "gpurun_out/${TAG}_sq_${NAME}.csv" ] || fail "empty result"
head -5 "gpurun_out/${TAG}_sq_${NAME}.csv" | cut -c1-200
