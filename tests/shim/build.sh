#!/bin/bash
# TEST infrastructure: compiles the reference-side binding (include/map_merge_3d_shim.hpp) as one
# translation unit against the reference's OWN public headers where they lie (R/include, read-only,
# nothing is copied) plus the stand-ins of tests/shim/mock, and links it against libmm3d.so with
# --no-undefined.  Output: tests/shim/_build/shim_check (git-ignored; travels to the GPU box).
set -euo pipefail
cd "$(dirname "$0")"
REF=${MM3D_REFERENCE:-/root/reference/map_merge_3d}
[ -d "$REF/include/map_merge_3d" ] || { echo "reference headers absent: shim_check not rebuilt"; exit 0; }
mkdir -p _build
g++ -std=c++14 -O1 -Wall -Wextra -Werror -Wno-unused-parameter -I"$REF/include" -Imock -I../../include shim_check.cpp \
    -o _build/shim_check -L../../map-merge_amd -lmm3d -Wl,-rpath,'$ORIGIN/../../../map-merge_amd' -Wl,--no-undefined
echo "built $(pwd)/_build/shim_check"
