#!/bin/bash
# Effective shader clock of the feature / pair kernels of scripts/pmc_driver.py: GRBM_GUI_ACTIVE (cycles the GPU was busy, at the
# shader clock) over each dispatch's duration from the kernel trace.  gpurun -- scripts/pmc_clock.sh <tag>
TAG=${1:-clk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out
rm -rf /tmp/pmcC
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcC -- python3 scripts/pmc_driver.py pair > /tmp/pmcC.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections, re
cc = glob.glob('/tmp/pmcC/*/*counter_collection.csv')[0]
kt = glob.glob('/tmp/pmcC/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d, name = dur.get(r['Dispatch_Id'], (0, r['Kernel_Name']))
    m = re.search(r'mm3d::(\w+)', name)
    k = m.group(1) if m else name[:30]
    if 'SnbCfg<8, 3584' in name: k += '_dense'
    if 'SnbCfg<8, 1792' in name: k += '_oct0'
    a = acc[k]; a[0] += float(r['Counter_Value']); a[1] += d; a[2] += 1
out = open('gpurun_out/%s_effective_clock.txt' % sys.argv[1], 'w')
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    line = "%-28s dispatches %3d  %9.1f us  GRBM_GUI_ACTIVE %.3e  -> %.2f GHz (if the counter is summed over 8 XCDs: %.2f)" % (k, n, ns / 1e3, cyc, cyc / max(ns, 1), cyc / max(ns, 1) / 8)
    print(line); out.write(line + "\n")
PY
