#!/bin/bash
# A/B runs of instrumentation / tuning builds: scripts/ab_variants.sh "<lib suffixes>" -- one-map and one-pair kernel tables per build
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in $1; do
  lib=$R/map-merge_amd/libmm3d$v.so
  echo "=== $lib"
  MM3D_LIB=$lib TOPN=9 timeout 300 python3 scripts/one_map_latency.py 2>&1 | grep -E "features|sift_|spfh|normals_radius |fpfh_weight"
  MM3D_LIB=$lib timeout 300 python3 scripts/one_pair_latency.py 500000 2>&1 | grep -E "kernels|icp_corr|score_nn|sacia_err|sacia_select|seq_sum"
done
