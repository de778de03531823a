#!/bin/bash
# Collects one round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r01d
# writes gpurun_out/<tag>_{kernel_stats.csv,hip_event_table.csv,concurrency.txt,pmc_FETCH_SIZE.csv,
# pmc_WRITE_SIZE.csv,pmc_sq_counters.csv,bench_line.json}; scripts/assemble_profiles.py <tag> then turns
# them into profiles/<tag>_*.  rocprofv3 + 16 host threads crashes now and then inside the profiler's
# copy interception, hence the retries.
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out
for try in 1 2 3; do
  rm -rf /tmp/prof_kt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
      --kernel-table "$R/gpurun_out/${TAG}_hip_event_table.csv" > /tmp/prof_kt.log 2>&1 && break
done
cp /tmp/prof_kt/*/*kernel_stats.csv "gpurun_out/${TAG}_kernel_stats.csv"
python3 scripts/trace_concurrency.py /tmp/prof_kt/*/*kernel_trace.csv > "gpurun_out/${TAG}_concurrency.txt"
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d /tmp/prof_pmc -- python3 bench.py --steps 1 --warmup 1 --streams 1 \
      --no-cpu-baseline > /tmp/prof_pmc.log 2>&1
  python3 scripts/pmc_summary.py /tmp/prof_pmc/*/*counter_collection.csv > "gpurun_out/${TAG}_pmc_${cnt}.csv"
done
rm -rf /tmp/prof_sqA /tmp/prof_sqB
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
    --output-format csv -d /tmp/prof_sqA -- python3 scripts/pmc_driver.py pair > /tmp/prof_sqA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
    --output-format csv -d /tmp/prof_sqB -- python3 scripts/pmc_driver.py pair > /tmp/prof_sqB.log 2>&1
python3 scripts/pmc_summary.py /tmp/prof_sqA/*/*counter_collection.csv /tmp/prof_sqB/*/*counter_collection.csv > "gpurun_out/${TAG}_pmc_sq_counters.csv"
python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --kernel-table "$R/gpurun_out/${TAG}_hip_event_table_1stream.csv" > /tmp/one_stream.log 2>&1
python3 bench.py 2>&1 | tail -1 > "gpurun_out/${TAG}_bench_line.json"
cut -c1-260 "gpurun_out/${TAG}_bench_line.json"
cat "gpurun_out/${TAG}_concurrency.txt"
