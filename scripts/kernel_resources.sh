#!/bin/bash
# VGPRs / SGPRs / LDS / scratch of every kernel of one csrc file as the gfx950 code object records them (runs without a GPU):
#   scripts/kernel_resources.sh sift.hip [extra hipcc flags]
set -e
f=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function --cuda-device-only -c "$R/map-merge_amd/csrc/$f" -o $tmp/dev.o "$@"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'args': 
        if cur.get('name'): rows.append(cur)
        cur = {}
    if k in ('name', 'vgpr_count', 'sgpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size', 'vgpr_spill_count', 'agpr_count', 'max_flat_workgroup_size'): cur[k] = v
if cur.get('name'): rows.append(cur)
for r in rows:
    try: name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    except Exception: name = r['name']
    name = re.sub(r'\(.*', '', name)[:110]
    print(f\"{name:112s} vgpr {r.get('vgpr_count','?'):>4} agpr {r.get('agpr_count','0'):>3} sgpr {r.get('sgpr_count','?'):>4} lds {r.get('group_segment_fixed_size','?'):>7} scratch {r.get('private_segment_fixed_size','?'):>5} spill {r.get('vgpr_spill_count','?'):>3} wg {r.get('max_flat_workgroup_size','?')}\")
"
rm -rf $tmp
