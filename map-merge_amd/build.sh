#!/bin/bash
# Builds libmm3d.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# -ffp-contract=off: float arithmetic must round exactly like the CPU path it is checked against;
# kernels that may fuse say fmaf() explicitly.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-parameter"
BUILD=${BUILD_DIR:-build}
OUT=${OUT:-libmm3d.so}
mkdir -p $BUILD
SRCS="libm_debug.hip grid.hip filters.hip normals.hip sift.hip harris.hip fpfh.hip pfh.hip rsd.hip shot.hip sc3d.hip desc_knn.hip registration.hip nn.hip runtime.cpp linalg.cpp host_pipeline.cpp devices.cpp capi.cpp"
OBJS=""
pids=()
for s in $SRCS; do
  o=$BUILD/${s%.*}.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "csrc/$s" -nt "$o" ] || [ -n "$(find csrc ../include -name '*.h*' -newer "$o" 2>/dev/null)" ]; then
    ( $HIPCC $FLAGS -x hip -c "csrc/$s" -o "$o" ${EXTRA:-} ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# (librccl -- the one collective of the path, the all-gather of the pair records between the devices of one process -- is bound
# by csrc/devices.cpp on first use of mm3d_create_devices, not at load time: 570 MB that a one-GPU process never maps)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS -ldl
echo "built $(pwd)/$OUT"
