#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel: usage pmc_summary.py file.csv [file2.csv ...]"""
import csv, sys, collections, re

def short(name):
    m = re.search(r"mm3d::(\w+)(?:<(\d+)(?:, *\d+)*>)?", name)
    if m:
        # the ICP / score search is one template (k_nn_wave<MODE, SPLIT>, MODE 0 / 1): keep the two modes apart
        # and the SIFT scale-space kernel's repair configuration (largest tile, one block per overflow item: a handful of
        # tiny launches) apart from the launches that work whole octaves, so that "per dispatch" means per octave
        if m.group(1) == "k_sift_dog_lds" and re.search(r"SnbCfg<8, *3584", name):
            return "k_sift_dog_lds_dense"
        # (round 6) the later octaves' configuration now only serves the certified path's exact launches on the handful of
        # marked points (single-query items): apart from the first octave's launch, which is what "sift_dog" means in the step
        if m.group(1) == "k_sift_dog_lds" and re.search(r"SnbCfg<8, *2816", name):
            return "k_sift_dog_lds_exact"
        return m.group(1) + ("<%s>" % m.group(2) if m.group(1) == "k_nn_wave" and m.group(2) else "")
    m = re.search(r"(\w+)<", name)
    return (m.group(1) if m else name)[:40]

acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add((path, row["Dispatch_Id"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel,dispatches," + ",".join(names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("SQ_ACTIVE_INST_ANY", 0))):
    print(k + "," + str(len(disp[k]) // max(len(sys.argv) - 1, 1)) + "," + ",".join("%.4g" % acc[k].get(c, float("nan")) for c in names))
