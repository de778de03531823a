"""Prices a certified SAC-IA pick BEFORE it is built: SampleConsensusInitialAlignment keeps the hypothesis with the lowest error
sum (`if (i_iter == 0 || error < lowest_error)`, ia_ransac.hpp; R/src/matching.cpp:142-194) and hands on its TRANSFORM -- the
sum itself leaves the stage nowhere.  Until round 6 the device reproduced every hypothesis' float chain (k_seq_sum: 15.7 k
dependent additions per hypothesis on the headline).  With S_h = the same terms summed in any order in double, the CPU path's float sum lies in
S_h (1 +- n u (1 + 1e-3)) (n - 1 sequential roundings of partial sums that never exceed the total: all terms are >= 0); a
hypothesis whose interval lies wholly above the lowest upper end cannot be the minimum.  How many hypotheses per pair are left?
EVIDENCE TOOL (it runs the oracle, CPU only):   python3 scripts/sacia_price.py [maps] [points] [hypotheses]"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

ge.load()
po = ge.load_oracle()
n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
H = int(sys.argv[3]) if len(sys.argv) > 3 else 500
U = 2.0 ** -24
host = bench.make_workload(max(n_maps, 2), n_pts)
po.set_threads(os.cpu_count() or 1)
P = po.params_default()
maps = []
for mi in range(n_maps):
    d = po.downsample(host[mi], P.resolution)
    f = po.remove_outliers(d, P.descriptor_radius, P.outliers_min_neighbours)
    nrm = po.normals(f, P.normal_radius)
    kp_raw, _ = po.keypoints_sift(f, P.resolution, 3, 3, P.keypoint_threshold)
    kp, desc = po.descriptors_fpfh(f, nrm, kp_raw, P.descriptor_radius)
    maps.append((kp, desc))
    print(f"map {mi}: {len(f)} points, {len(kp)} keypoints", flush=True)
po.srand(1)
tot = []
for i, j in itertools.combinations(range(n_maps), 2):
    (skp, sd), (tkp, td) = maps[i], maps[j]
    T, bi, be, ef, ed = po.sac_ia_errors(skp, sd, tkp, td, P.inlier_threshold, P.max_correspondence_distance, H)
    n = len(skp)
    delta = n * U * 1.001 + 1e-12
    lo, hi = ed * (1 - delta), ed * (1 + delta)
    assert np.all(ef.astype(np.float64) >= lo) and np.all(ef.astype(np.float64) <= hi), "the bound does not hold"
    worst = float(np.max(np.abs(ef.astype(np.float64) - ed) / (ed * delta)))
    cand = np.flatnonzero(lo <= hi.min())
    assert bi in cand
    srt = np.sort(ed)
    print(f"pair ({i}, {j}): n = {n}, best error {be:.2f} = {be / n:.4f} n at hypothesis {bi}; second best + {(srt[1] - srt[0]) / srt[0] * 100:.3f} %; "
          f"interval half-width {delta * 100:.3f} %; candidates left {len(cand)} of {H}; float sums within {worst:.2f} of the bound", flush=True)
    tot.append(len(cand))
print(f"{len(tot)} pairs: candidates per pair min {min(tot)}, median {int(np.median(tot))}, max {max(tot)}; pairs decided without any chain: "
      f"{sum(1 for c in tot if c == 1)}")
