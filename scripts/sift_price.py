"""Prices the certified SIFT decision BEFORE it is built: on headline maps, per octave, how many points
the approximate scale space with a rigorous error bound decides, and how many need the exact sorted-list DoG.
EVIDENCE TOOL (it runs the oracle, CPU only -- build container or GPU box):
    python3 scripts/sift_price.py [maps] [points] [slack]          (default 2 x 500000, slack 1.0)

detectKeypoints(SIFT) returns only WHICH points of an octave cloud are keypoints (R/src/features.cpp:45-62: xyz copied, the
scale dropped).  The decision of a (point, scale) is |val| >= min_contrast and <= 75 comparisons with the DoG values of the
25 nearest neighbours.  With val* = the real value (here: the double evaluation of the oracle hook) and B a bound on
|float result of the CPU path - val*| + |device approximation - val*|:
    contrast:  |val*| + B <  min_contrast        -> certified no       |val*| - B >= min_contrast -> certified live
    minimum :  a neighbour value certainly below  -> certified no       no neighbour value possibly below/equal -> certified yes
(maximum mirrored).  What stays open needs the exact DoG of the point and of the neighbours whose comparison is open.

The bound ("SIFT: certified decisions"): a response R = N / D, N = sum I_j w_j, D = sum w_j over the n
neighbours inside 3 sigma, all terms >= 0.  CPU path: w_j within (4.5 + 2) u of the real weight (quotient rounding times
|x| <= 4.5; expf within 1 ulp), the product 1 u, the n - 1 sequential additions gamma_(n-1), the division 1 u:
|R_float - R| <= R (2 n + 14) u (1 + 1e-3).  Device: weights within 16 u (v_exp_f32 2 ulp, two roundings of an argument of
magnitude <= 6.5), float sums of at most ceil(n / 8) + 8 terms per partial, partials added in double: R (2 (n / 8 + 8) + 34) u.
DoG = difference of two responses (+ 255 u for the subtraction)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

ge.load()
po = ge.load_oracle()
n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
slack = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
U = 2.0 ** -24
MINC = 5.0
host = bench.make_workload(max(n_maps, 2), n_pts)
po.set_threads(os.cpu_count() or 1)
P = po.params_default()

for mi in range(n_maps):
    d = po.downsample(host[mi], P.resolution)
    f = po.remove_outliers(d, P.descriptor_radius, P.outliers_min_neighbours)
    for octv in range(3):
        r = po.sift_octave_debug(f, P.resolution, octv)
        if r is None:
            break
        cloud, dog, resp, cnt, knn = r
        n = len(cloud)
        nn = cnt.astype(np.float64)
        b_cpu = resp * (2 * nn + 14) * U * 1.001
        b_dev = resp * (2 * (np.ceil(nn / 8) + 8) + 34) * U * 1.001
        b_resp = (b_cpu + b_dev) * slack
        val = resp[:, 1:] - resp[:, :-1]                      # [n, 5] real DoG
        B = b_resp[:, 1:] + b_resp[:, :-1] + 255 * U          # [n, 5]
        # the bound against the CPU path's floats (must hold: it is what the certificate rests on)
        err = np.abs(dog.astype(np.float64) - val)
        b_cpu_dog = b_cpu[:, 1:] + b_cpu[:, :-1] + 255 * U
        worst = float(np.max(err / b_cpu_dog))
        lo, hi = val - B, val + B
        a = np.abs(val[:, 1:4])
        c_no = a + B[:, 1:4] < MINC
        c_live = a - B[:, 1:4] >= MINC
        c_open = ~(c_no | c_live)
        cand = ~c_no                                           # (point, scale) pairs that go on to the comparisons
        # neighbour values: [n, 25, 5]
        ok = knn >= 0
        kq = np.where(ok, knn, 0)
        self_mask = kq == np.arange(n)[:, None]
        need_exact = np.zeros(n, dtype=bool)
        open_ps = 0
        kp_yes = 0
        decided_no_by_cmp = 0
        for s in range(3):
            rows = np.nonzero(cand[:, s])[0]
            if len(rows) == 0:
                continue
            nb = kq[rows]                                      # [m, 25]
            okr = ok[rows]
            selfr = self_mask[rows]
            v_lo, v_hi = lo[rows, s + 1][:, None], hi[rows, s + 1][:, None]
            clear_min = np.zeros(nb.shape, dtype=bool)
            poss_min = np.zeros(nb.shape, dtype=bool)
            clear_max = np.zeros(nb.shape, dtype=bool)
            poss_max = np.zeros(nb.shape, dtype=bool)
            for t in (s, s + 1, s + 2):
                q_lo, q_hi = lo[nb, t], hi[nb, t]
                skip = selfr if t == s + 1 else np.zeros_like(selfr)
                m = okr & ~skip
                clear_min |= m & (q_hi < v_lo)
                poss_min |= m & (q_lo <= v_hi)
                clear_max |= m & (q_lo > v_hi)
                poss_max |= m & (q_hi >= v_lo)
            min_no, max_no = clear_min.any(1), clear_max.any(1)
            min_yes, max_yes = ~poss_min.any(1), ~poss_max.any(1)
            live = c_live[rows, s]
            decided = (min_no & max_no) | (live & (min_yes | max_yes)) | (live & (min_no | min_yes) & (max_no | max_yes))
            # (contrast open: only "both no" decides)
            decided &= ((min_no & max_no) | live)
            und = ~decided
            open_ps += int(und.sum())
            kp_yes += int((live & (min_yes | max_yes)).sum())
            decided_no_by_cmp += int((min_no & max_no).sum())
            ur = rows[und]
            need_exact[ur] = True
            # the neighbours whose comparison is open (possible, not clear) on a side that is still open
            om = (poss_min & ~clear_min)[und] & ~min_no[und][:, None]
            ox = (poss_max & ~clear_max)[und] & ~max_no[und][:, None]
            need_exact[np.unique(nb[und][om | ox])] = True
        print(f"map {mi} octave {octv}: n {n:7d}  neighbours inside 3 sigma of the six scales {np.round(cnt.mean(0)).astype(int).tolist()}  "
              f"median B of DoG 1..3 {np.median(B[:, 1:4], 0).round(5).tolist()}  CPU floats within {worst:.3f} of their bound", flush=True)
        print(f"    contrast (point, scale): certified no {c_no.mean() * 100:.2f} %  live {c_live.mean() * 100:.2f} %  open {c_open.sum()} "
              f"({c_open.mean() * 100:.3f} %); points with no candidate scale {(~cand.any(1)).mean() * 100:.1f} % (need no 25-NN)")
        print(f"    comparisons: {int(cand.sum())} (point, scale) candidates: no by two clear violators {decided_no_by_cmp}, certified keypoints {kp_yes}, "
              f"open {open_ps}; points needing the exact DoG (open + their open neighbours) {int(need_exact.sum())} = "
              f"{need_exact.mean() * 100:.3f} % of the octave", flush=True)
