#!/usr/bin/env python3
"""gpurun_out/<tag>_table_cfg{1..5}.json (scripts/baseline_table.sh) -> the Results table of BASELINE.md and
profiles/<tag>_table_cfgN.json.   usage: fill_baseline_table.py <tag>"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for cfg in (1, 2, 3, 4, 5, "4indoor", "2lattice", "3lattice"):
    src = os.path.join(root, "gpurun_out", f"{tag}_table_cfg{cfg}.json")
    try:
        d = json.load(open(src))
    except Exception:
        rows.append(f"| {cfg} | not run | | | | | | | | | |")
        continue
    shutil.copy(src, os.path.join(root, "profiles", f"{tag}_table_cfg{cfg}.json"))
    kb = d.get("kernel_bounds_isolated", {})
    nb = kb.get("normals_radius", {})
    # (the wide rows' selector runs on split-bf16 MFMA under its own name: its fraction is of the bf16 peak)
    knn = kb.get("desc_knn_mfma_bf16") or kb.get("desc_knn_mfma") or {}
    knn_note = " of the bf16 peak (split-bf16 selector)" if "desc_knn_mfma_bf16" in kb else ""
    par = d.get("parity_check") or {}
    m = d.get("mpoints_per_s", {})
    b1 = d.get("cpu_baseline") or {"value": float("nan")}
    b2 = d.get("cpu_baseline_all_cores") or {"value": float("nan"), "cores": 0}
    gt = d.get("gt_error") or {}
    rows.append("| %s | %s | %.4g | %.4g (%d) | **%.4g** (%.1f ms/step) | n/a (driver) | %s | %s | %s | %s | %s | %s |" % (
        cfg, d["config"]["workload"], b1["value"], b2["value"], b2["cores"], d["value"], d["ms_per_step"],
        m.get("normals") if m.get("normals") is not None else ("fused into SIFT's first octave" if m.get("normals_fused") else None),
        ("%.2f %%" % (100 * nb["hbm_frac"])) if "hbm_frac" in nb else
        (("%.2f %% (stand-alone launch)" % (100 * m["normals_alone"]["hbm_frac"])) if (m.get("normals_alone") or {}).get("hbm_frac") is not None else "—"),
        (("%.1f %%" % (100 * knn["mfma_frac"])) + knn_note) if "mfma_frac" in knn else "—", m.get("icp"),
        ("ok: T %.1e%s, conf %.1e" % (par["pair_transform_frobenius"],
                                      (" (vs double-sum ICP %.0e)" % par["pair_transform_frobenius_vs_double_sums"]) if "pair_transform_frobenius_vs_double_sums" in par else "",
                                      par["confidence_rel_err"])) if par.get("ok") else ("FAILED" if par else "—"),
        (("%d of %d pairs within 0.5 (median %.2f)" % (gt["recovered_within_0.5"], gt["pairs_with_overlap_ge_0.3"], gt["median_frobenius"])) if gt.get("pairs_with_overlap_ge_0.3") else "—")
        + "; ICP iterations " + json.dumps(d.get("icp_iterations_histogram", {})).replace('"', "")))
head = ("| config | workload | B1 pairs/s (1 core) | B2 pairs/s (cores) | GPU×1 pairs/s | GPU×8 | normals Mpts/s | normals % HBM | dist-matrix % MFMA | ICP Mpts/s "
        "| parity_check (device vs oracle, maps 0, 1, pair (0,1)) | ground truth (pairs with >= 30 % overlap) |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
table = head + "\n".join(rows) + "\n"
p = os.path.join(root, "BASELINE.md")
s = open(p).read()
a, b = "<!-- results:begin -->", "<!-- results:end -->"
if a not in s:
    i = s.index("### Results")
    j = s.index("\n| config | B1 pairs/s")
    k = s.index("\n\n", j + 1) if "\n\n" in s[j + 1:] else len(s)
    s = s[:i] + "### Results\n\n" + a + "\n" + b + "\n" + s[k:]
i, j = s.index(a) + len(a), s.index(b)
note = ("\nMeasured by `scripts/baseline_table.sh %s` (one `python bench.py ...` per configuration on one MI355X box; B1 / B2 = the CPU oracle on "
        "that box's host, one thread / its physical cores, on the sample the line's `cpu_baseline.sample` names).  Lines: `profiles/%s_table_cfgN.json`.  "
        "GPU×8 is the driver's to measure (SCALE_rNN.json).\n\n" % (tag, tag))
s = s[:i] + note + table + s[j:]
open(p, "w").write(s)
print(table)
