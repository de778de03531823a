#!/usr/bin/env python3
"""gpurun_out/<tag>_sq_<name>.csv (scripts/profile_counters.sh on the GPU box) -> profiles/<tag>_sq_<name>.csv and
profiles/<tag>_counters_<name>.json: the workload the counters belong to (bench.workload_signature of the same options), per
kernel the VALU wave-instructions per dispatch and the SQ ratios, and the machine-code hash of each kernel in THIS tree's
libmm3d.so (which must be the library the GPU run used).
usage: assemble_counters.py <tag> <name> [--points N --descriptor D --method M --scenes S --window W --resolution R --sac-iterations H]"""
import argparse
import csv
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("name")
ap.add_argument("--points", type=int, default=500000)
ap.add_argument("--descriptor", default="FPFH")
ap.add_argument("--method", default="SAC_IA")
ap.add_argument("--scenes", default="independent")
ap.add_argument("--window", type=float, default=0.0)
ap.add_argument("--resolution", type=float, default=0.0)
ap.add_argument("--sac-iterations", type=int, default=0)
ap.add_argument("--overlap-step", type=float, default=0.5)
a = ap.parse_args()
src = os.path.join(root, "gpurun_out", f"{a.tag}_sq_{a.name}.csv")
dst_csv = os.path.join(root, "profiles", f"{a.tag}_sq_{a.name}.csv")
shutil.copy(src, dst_csv)
per = {}
for row in csv.DictReader(open(src)):
    d = float(row.get("dispatches") or 0)
    if d <= 0 or row.get("SQ_INSTS_VALU") in (None, "", "nan"):
        continue
    name = bench.KERNEL_OF_SYMBOL.get(row["kernel"], row["kernel"])

    def ratio(x, y):
        try:
            return round(float(row[x]) / float(row[y]), 4) if float(row[y]) > 0 else None
        except (KeyError, ValueError, TypeError):
            return None
    per[name] = {"dispatches": int(d), "valu_wave_instructions_per_dispatch": float(row["SQ_INSTS_VALU"]) / d,
                 "stall": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "lds_conflict": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                 "valu_active_per_wave_cycle": ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")}
doc = {"_note": "SQ counters of scripts/pmc_driver.py pair on this workload (maps 0 and 1 + pair (0, 1), one stream), per dispatch; "
                "bench.py uses them for a run of the same workload only, and only for kernels whose machine code is unchanged",
       "workload": bench.workload_signature(a), "source": "profiles/" + os.path.basename(dst_csv), "kernels": per,
       "source_sha256": {k: bench.kernel_source_hash(k) for k in per if bench.kernel_source_hash(k)}}
out = os.path.join(root, "profiles", f"{a.tag}_counters_{a.name}.json")
json.dump(doc, open(out, "w"), indent=1)
print(out, len(per), "kernels")
