#!/usr/bin/env python3
"""gpurun_out/<tag>_* (written by scripts/profile_round.sh on the GPU box) -> profiles/<tag>_*.
usage: assemble_profiles.py <tag>"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = lambda n: os.path.join(root, "gpurun_out", f"{tag}_{n}")
dst = lambda n: os.path.join(root, "profiles", f"{tag}_{n}")
f = {r["kernel"]: (int(r["dispatches"]), float(r["FETCH_SIZE"])) for r in csv.DictReader(open(src("pmc_FETCH_SIZE.csv")))}
w = {r["kernel"]: (int(r["dispatches"]), float(r["WRITE_SIZE"])) for r in csv.DictReader(open(src("pmc_WRITE_SIZE.csv")))}
sys.path.insert(0, root)
import bench  # noqa: E402  (KERNEL_SYMBOLS / kernel_source_hash: what makes a committed counter stale -- the machine code in THIS tree's libmm3d.so, which must be the library the GPU run used: do not rebuild between the run and this script)
names = {"k_sift_dog_lds": "sift_dog", "k_sift_dog_lds_exact": "sift_dog_exact", "k_sift_dog_fast": "sift_dog_fast", "k_sift_reject": "sift_reject", "k_sift_extrema_one": "sift_extrema_one", "k_normals_lds": "normals_radius", "k_sift_dog": "sift_dog_big", "k_spfh": "spfh", "k_sacia_err": "sacia_err", "k_nn_wave": ["icp_corr_reduce", "score_nn_reduce"],
         "k_nn_wave<0>": "icp_corr_reduce", "k_nn_wave<1>": "score_nn_reduce",
         "k_sacia_exact": "sacia_seq_sum", "k_sacia_select": "sacia_select", "k_knn_mfma": "desc_knn_mfma", "k_knn_rerank": "desc_knn_rerank",
         "k_fpfh_weight": "fpfh_weight", "k_fpfh_mark": "fpfh_mark", "k_normals": "normals_radius_big",
         "k_radius_count": "radius_outlier_count", "k_voxel_centroid": "voxel_centroid", "trampoline_kernel": "rocprim_radix_sort_pairs"}
out, rows = {}, []
for k, (d, fs) in f.items():
    ws = w.get(k, (d, 0.0))[1]
    b = (2 * fs + ws) * 1024 / d
    rows.append((k, d, fs / d, ws / d, b))
    n = names.get(k)
    if n:
        for nn in (n if isinstance(n, list) else [n]):
            out[nn] = round(b)
json.dump({"_note": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / dispatches (MI355X_MICROARCH.md: gfx950 FETCH_SIZE "
                    "reads half of a wide coalesced stream; separate --pmc passes; Infinity-Cache hits are counted).  bench.py --steps 1 "
                    "--warmup 1 --streams 1 under rocprofv3 --pmc.  rocprim_radix_sort_pairs is the average over ALL rocPRIM kernels (sort and scan passes).",
           # the workload these per-launch figures (and the SQ counters of the same tag) were counted on: bench.py uses them only
           # for a run of the same workload (bench.COUNTERS_WORKLOAD_DEFAULT = what scripts/profile_round.sh runs)
           "workload": bench.COUNTERS_WORKLOAD_DEFAULT,
           "bytes_per_launch": out,
           "source_sha256": {k: bench.kernel_source_hash(k) for k in out if bench.kernel_source_hash(k)}}, open(dst("traffic.json"), "w"), indent=1)
with open(dst("pmc_hbm_traffic.csv"), "w") as fo:
    fo.write("kernel,dispatches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
    for r in sorted(rows, key=lambda r: -r[4] * r[1]):
        fo.write("%s,%d,%.1f,%.1f,%.0f\n" % r)
for n in ("kernel_stats.csv", "hip_event_table.csv", "pmc_sq_counters.csv", "concurrency.txt", "bench_line.json"):
    shutil.copy(src(n), dst(n))
if os.path.exists(src("hip_event_table_1stream.csv")):
    shutil.copy(src("hip_event_table_1stream.csv"), dst("hip_event_table_1stream.csv"))
print("profiles/%s_* written" % tag)
