#!/bin/bash
# SQ counters of the feature kernels of one map (scripts/pmc_normals.py), two passes: gpurun -- scripts/pmc_sq.sh <tag>
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out
rm -rf /tmp/pmcA /tmp/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --output-format csv -d /tmp/pmcA -- python3 scripts/pmc_normals.py > /tmp/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
  --output-format csv -d /tmp/pmcB -- python3 scripts/pmc_normals.py > /tmp/pmcB.log 2>&1
python3 scripts/pmc_summary.py /tmp/pmcA/*/*counter_collection.csv /tmp/pmcB/*/*counter_collection.csv > gpurun_out/${TAG}_pmc_sq.csv
head -12 gpurun_out/${TAG}_pmc_sq.csv
