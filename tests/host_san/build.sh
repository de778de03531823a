#!/bin/bash
# Builds the library's host code -- runtime.cpp, capi.cpp, host_pipeline.cpp, linalg.cpp, devices.cpp, unchanged -- with a sanitizer against
# the fake HIP runtime and the fake device layer of this directory, into tests/host_san/_build/san_<kind> (git-ignored):
#   tests/host_san/build.sh thread | address
# clang's host pass of the HIP language (--cuda-host-only): the headers' __device__ helpers are parsed, never emitted.
set -euo pipefail
KIND=${1:-thread}
cd "$(dirname "$0")"
CSRC=../../map-merge_amd/csrc
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang++}
[ "$KIND" = thread ] && SAN="-fsanitize=thread" || SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
FLAGS="-x hip --cuda-host-only -nogpulib -std=c++17 -O1 -g -fno-omit-frame-pointer -ffp-contract=off -I/opt/rocm/include -I$CSRC -Wno-unused-function -Wno-option-ignored -Wno-unused-command-line-argument $SAN"
mkdir -p _build
objs=""
pids=()
for f in $CSRC/runtime.cpp $CSRC/capi.cpp $CSRC/host_pipeline.cpp $CSRC/linalg.cpp $CSRC/devices.cpp fake_hip.cpp fake_rccl.cpp fake_device.cpp san_main.cpp; do
  o=_build/$(basename "${f%.*}")_$KIND.o
  objs="$objs $o"
  ( $CLANG $FLAGS -c "$f" -o "$o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
# -rdynamic: devices.cpp binds RCCL with dlsym(RTLD_DEFAULT, ...) first -- the fake of fake_rccl.cpp must be visible to it
$CLANG $SAN -rdynamic -o _build/san_$KIND $objs -lpthread -ldl
echo "built $(pwd)/_build/san_$KIND"
