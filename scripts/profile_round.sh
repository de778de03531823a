#!/bin/bash
# Collects one round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r02a
# writes gpurun_out/<tag>_{kernel_stats.csv,hip_event_table.csv,concurrency.txt,pmc_FETCH_SIZE.csv,
# pmc_WRITE_SIZE.csv,pmc_sq_counters.csv,bench_line.json,hip_event_table_1stream.csv};
# scripts/assemble_profiles.py <tag> then turns them into profiles/<tag>_*.
# rocprofv3 + 16 host threads crashes now and then inside the profiler's copy interception, hence the
# retries; a step that still fails, or leaves no (or an empty) result file, ends the script with a
# non-zero status instead of letting stale files pass for evidence.
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out
rm -f gpurun_out/${TAG}_*

fail() { echo "profile_round: $*" >&2; exit 1; }
need() { [ -s "$1" ] || fail "missing or empty: $1"; }
# with_retries <tries> <dir> -- rocprofv3 ...: fresh child and fresh output directory per try
with_retries() {
  local tries=$1 dir=$2; shift 3
  for try in $(seq 1 "$tries"); do
    rm -rf "$dir"
    if "$@" > "$dir.log" 2>&1 && ls "$dir"/*/*.csv > /dev/null 2>&1; then return 0; fi
    echo "profile_round: try $try of $tries failed: $*" >&2
  done
  return 1
}

# (nothing behind the timed steps in this run: trace_concurrency.py looks at the last 60 % of the traced window)
export MM3D_BENCH_NO_ISOLATED=1
with_retries 3 /tmp/prof_kt -- rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 bench.py --steps 5 --warmup 3 --no-pcie --no-pair-stage \
    --no-cpu-baseline --kernel-table "$R/gpurun_out/${TAG}_hip_event_table.csv" || fail "kernel trace"
cp /tmp/prof_kt/*/*kernel_stats.csv "gpurun_out/${TAG}_kernel_stats.csv" || fail "no kernel_stats.csv"
need "gpurun_out/${TAG}_kernel_stats.csv"; need "gpurun_out/${TAG}_hip_event_table.csv"
python3 scripts/trace_concurrency.py /tmp/prof_kt/*/*kernel_trace.csv > "gpurun_out/${TAG}_concurrency.txt" || fail "trace_concurrency"
unset MM3D_BENCH_NO_ISOLATED
for cnt in FETCH_SIZE WRITE_SIZE; do
  with_retries 2 /tmp/prof_pmc -- rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d /tmp/prof_pmc -- python3 bench.py --steps 1 --warmup 1 \
      --streams 1 --no-cpu-baseline || fail "pmc $cnt"
  python3 scripts/pmc_summary.py /tmp/prof_pmc/*/*counter_collection.csv > "gpurun_out/${TAG}_pmc_${cnt}.csv" || fail "pmc_summary $cnt"
  need "gpurun_out/${TAG}_pmc_${cnt}.csv"
done
with_retries 2 /tmp/prof_sqA -- rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/prof_sqA -- python3 scripts/pmc_driver.py pair || fail "pmc SQ pass A"
with_retries 2 /tmp/prof_sqB -- rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS \
    SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/prof_sqB -- python3 scripts/pmc_driver.py pair \
    || fail "pmc SQ pass B"
python3 scripts/pmc_summary.py /tmp/prof_sqA/*/*counter_collection.csv /tmp/prof_sqB/*/*counter_collection.csv \
    > "gpurun_out/${TAG}_pmc_sq_counters.csv" || fail "pmc_summary SQ"
need "gpurun_out/${TAG}_pmc_sq_counters.csv"
python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --kernel-table "$R/gpurun_out/${TAG}_hip_event_table_1stream.csv" \
    > /tmp/one_stream.log 2>&1 || fail "one-stream bench"
need "gpurun_out/${TAG}_hip_event_table_1stream.csv"
python3 bench.py 2> /tmp/bench.err | tail -1 > "gpurun_out/${TAG}_bench_line.json"
python3 -c "import json,sys; json.load(open('gpurun_out/${TAG}_bench_line.json'))" || fail "bench line is not JSON (see /tmp/bench.err)"
cut -c1-260 "gpurun_out/${TAG}_bench_line.json"
cat "gpurun_out/${TAG}_concurrency.txt"
