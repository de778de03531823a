"""Exposure census for the five weakest rows of DESIGN.md section 4's audit list -- the part of the pin that needs no PCL.
EVIDENCE TOOL (runs the CPU oracle; build container or GPU box):
    python3 scripts/audit_exposure.py [config ...]        (default: 0 1 2 3 4 -> profiles/r06_audit_exposure.txt via stdout)

The oracle restates PCL 1.8.1 from memory.  For five recalled details nobody can say here whether the recollection is right;
what can be MEASURED is how much of each BASELINE workload's output would move if it were wrong:
  row 1   radiusSearch's order among neighbours at exactly the same float d2 (oracle: ascending index)
          -> neighbourhoods that hold such a tie (normals, descriptor and SIFT scale-space searches); no tie, no exposure
  row 2   the order inside a voxel after VoxelGrid's std::sort (oracle: input order)
          -> voxels whose float centroid changes when libstdc++'s own std::sort orders them (oracle/audit_sort.cpp)
  row 10  FLANN's distance functor (oracle: L2_Simple, sequential; the alternative: L2, four-way unrolled)
          -> descriptor rows whose 10 nearest neighbours (SAC-IA's feature neighbours) or their order change
  row 11  boost::uniform_int's mapping in RANSAC (oracle: mt() >> 1)
          -> pairs that run RANSAC at all (the SAC_IA configurations never draw from it)
  row 13  Eigen::umeyama's float summation order (oracle: sequential)
          -> ||T_a - T_b||_F of one ICP iteration's transform over sequential / pairwise / eight-lane sums, against the
             tolerance the parity tests state for the pair
R/src/features.cpp:19-40, 171-176; R/src/matching.cpp:50-75, 119-124, 204-220."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

ge.load()
po = ge.load_oracle()
L = po.lib()
import subprocess  # noqa: E402
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "audit"])
A = C.CDLL(os.path.join(ROOT, "oracle", "libaudit_sort.so"))
po.set_threads(os.cpu_count() or 1)

CONFIGS = {   # BASELINE.json configs[i]: (maps, points, window, resolution, descriptor, method, maps censused, scenes)
    0: dict(maps=2, points=10000, window=0.0, resolution=0.0, descriptor="FPFH", method="SAC_IA", take=2),
    1: dict(maps=4, points=200000, window=0.0, resolution=0.0, descriptor="FPFH", method="SAC_IA", take=2),
    2: dict(maps=16, points=500000, window=0.0, resolution=0.0, descriptor="FPFH", method="SAC_IA", take=2),
    3: dict(maps=8, points=2000000, window=30.0, resolution=0.05, descriptor="SHOT", method="MATCHING", take=1),
    4: dict(maps=64, points=50000, window=0.0, resolution=0.0, descriptor="FPFH", method="SAC_IA", take=4),
}


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def ties(pts, radius):
    out = (C.c_longlong * 4)()
    L.mo_audit_radius_ties(p(pts), len(pts), C.c_double(radius), out)
    return list(out)


def census(ci):
    cfg = CONFIGS[ci]
    t0 = time.time()
    host, _, _ = bench.make_workload_gt(cfg["maps"], cfg["points"], window=cfg["window"])
    P = po.params_default()
    if cfg["resolution"] > 0:
        P.resolution = cfg["resolution"]
    n_pairs = cfg["maps"] * (cfg["maps"] - 1) // 2
    print(f"== configs[{ci}]: {cfg['maps']} maps x {cfg['points']} raw points, {cfg['descriptor']} + {cfg['method']}"
          + (f", {cfg['window']:g} m windows, resolution {P.resolution:g}" if cfg["window"] else "") + f"; censused: maps 0..{cfg['take'] - 1}, pair (0, 1)", flush=True)
    feats = []
    for mi in range(cfg["take"]):
        raw = host[mi]
        # row 2
        o = (C.c_longlong * 7)()
        A.ma_voxel_sort(p(raw), len(raw), C.c_double(P.resolution), o)
        print(f"  map {mi} row 2 (voxel order): {o[0]} voxels of {o[6]} points; {o[1]} hold >= 3 points, {o[2]} hold >= 17; std::sort leaves {o[3]} of them in another "
              f"order than the input's; centroid xyz bits differ in {o[4]} voxels ({100.0 * o[4] / max(o[0], 1):.3f} %), packed rgba in {o[5]}", flush=True)
        d = po.downsample(raw, P.resolution)
        f = po.remove_outliers(d, P.descriptor_radius, P.outliers_min_neighbours)
        # row 1
        for what, r in (("normals", P.normal_radius), ("descriptor", P.descriptor_radius)):
            t = ties(f, r)
            print(f"  map {mi} row 1 (tie order), {what} search r = {r:g}: {t[1]} of {t[0]} neighbourhoods hold an exact d2 tie ({100.0 * t[1] / max(t[0], 1):.3f} %), "
                  f"{t[2]} tied pairs among {t[3]} neighbours", flush=True)
        for octv in range(3):
            r = po.sift_octave_debug(f, P.resolution, octv) if len(f) < 1500000 else None
            if r is None:
                continue
            oc = r[0]
            scale = np.float32(P.resolution) * np.float32(2 ** octv)
            rad = 3.0 * float(scale) * 2.0 ** (4.0 / 3.0)
            t = ties(oc, rad)
            print(f"  map {mi} row 1 (tie order), SIFT octave {octv} search r = {rad:.3f}: {t[1]} of {t[0]} neighbourhoods hold a tie ({100.0 * t[1] / max(t[0], 1):.3f} %), "
                  f"{t[2]} tied pairs among {t[3]} neighbours", flush=True)
        if mi < 2 and cfg["descriptor"] == "FPFH":
            n = po.normals(f, P.normal_radius)
            k_raw, _ = po.keypoints_sift(f, P.resolution, 3, 3, P.keypoint_threshold)
            k, e = po.descriptors_fpfh(f, n, k_raw, P.descriptor_radius)
            feats.append((f, k, e))
    if len(feats) == 2:
        (f0, k0, e0), (f1, k1, e1) = feats
        # row 10
        for a, b, name in ((e0, e1, "0 -> 1"), (e1, e0, "1 -> 0")):
            k = 10
            i1, d1 = po.desc_knn(a, b, k)
            i2 = np.empty((len(a), k), dtype=np.int32)
            d2 = np.empty((len(a), k), dtype=np.float32)
            L.mo_audit_desc_knn_unrolled(p(np.ascontiguousarray(a)), len(a), p(np.ascontiguousarray(b)), len(b), a.shape[1], k, p(i2), p(d2))
            order = int((i1 != i2).any(axis=1).sum())
            sets = int((np.sort(i1, 1) != np.sort(i2, 1)).any(axis=1).sum())
            bits = int((d1.view(np.uint32) != d2.view(np.uint32)).any(axis=1).sum())
            print(f"  row 10 (FLANN functor), descriptors {name}: of {len(a)} rows x 10 neighbours, {order} rows change a neighbour or the order "
                  f"({sets} change the SET), {bits} rows change a distance's bits -- under four-way-unrolled accumulation", flush=True)
        # row 13
        po.srand(1)
        Tg, _, _ = po.sac_ia(k0, e0, k1, e1, P.inlier_threshold, P.max_correspondence_distance, P.max_iterations)
        g = np.ascontiguousarray(Tg.T.reshape(16).astype(np.float32))
        so, do = np.empty((len(f0), 3), dtype=np.float32), np.empty((len(f0), 3), dtype=np.float32)
        nc = L.mo_audit_icp_correspondences(p(f0), len(f0), p(f1), len(f1), p(g), C.c_double(P.max_correspondence_distance), p(so), p(do))
        if nc >= 3:
            Ts = []
            for order in range(3):
                T = np.zeros(16, dtype=np.float32)
                L.mo_audit_umeyama_order(p(so), p(do), nc, order, p(T))
                Ts.append(T.astype(np.float64))
            spread = max(np.linalg.norm(Ts[a] - Ts[b]) for a in range(3) for b in range(a + 1, 3))
            tol = po.transform_tolerance(len(f0))
            print(f"  row 13 (umeyama's sums), pair (0, 1) first ICP iteration, {nc} correspondences: ||T_seq - T_pairwise||_F = {np.linalg.norm(Ts[0] - Ts[1]):.3e}, "
                  f"||T_seq - T_8lane||_F = {np.linalg.norm(Ts[0] - Ts[2]):.3e}, spread {spread:.3e}; the parity tests' oracle clause for this pair: {tol:.3e}", flush=True)
    # row 11
    ransac_pairs = n_pairs if cfg["method"] == "MATCHING" else 0
    print(f"  row 11 (uniform_int mapping): {ransac_pairs} of {n_pairs} pairs run RANSAC ({cfg['method']}"
          + ("; every draw of every such pair depends on the mapping: 3 per hypothesis, up to 1000 hypotheses" if ransac_pairs else ": SAC-IA draws from libc rand(), pinned against this image's glibc") + ")")
    print(f"  ({time.time() - t0:.0f} s)", flush=True)


if os.environ.get("MM3D_AUDIT_CENSUS", "1") != "0":       # (MM3D_AUDIT_CENSUS=0: only the downstream part below)
    for ci in ([int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]):
        census(ci)


def downstream_of_the_voxel_order(ci):
    """Row 2 followed downstream: the oracle's feature chain on the VoxelGrid output in input order (the oracle's, the device's)
    against the same chain on the output in libstdc++'s std::sort order (ma_downsample_stdsort).  The two clouds have the same
    voxels in the same order, so point i corresponds to point i."""
    cfg = CONFIGS[ci]
    if cfg["descriptor"] != "FPFH":
        return
    host, _, _ = bench.make_workload_gt(cfg["maps"], cfg["points"], window=cfg["window"])
    P = po.params_default()
    A.ma_downsample_stdsort.restype = C.c_int
    for mi in range(min(cfg["take"], 2)):
        raw = host[mi]
        d0 = po.downsample(raw, P.resolution)
        d1 = np.empty(len(raw), dtype=raw.dtype)
        n1 = A.ma_downsample_stdsort(p(raw), len(raw), C.c_double(P.resolution), p(d1))
        d1 = d1[:n1].copy()
        assert n1 == len(d0)
        xyz_diff = int((np.stack([d0["x"], d0["y"], d0["z"]], 1).view(np.uint32) != np.stack([d1["x"], d1["y"], d1["z"]], 1).view(np.uint32)).any(1).sum())

        def keep_mask(full, kept):                 # the filter preserves order: which points of `full` are in `kept`
            m = np.zeros(len(full), dtype=bool)
            fb, kb = full.view(np.uint8).reshape(len(full), -1), kept.view(np.uint8).reshape(len(kept), -1)
            j = 0
            for i in range(len(full)):
                if j < len(kept) and np.array_equal(fb[i], kb[j]):
                    m[i] = True
                    j += 1
            return m
        f0 = po.remove_outliers(d0, P.descriptor_radius, P.outliers_min_neighbours)
        f1 = po.remove_outliers(d1, P.descriptor_radius, P.outliers_min_neighbours)
        m0, m1 = keep_mask(d0, f0), keep_mask(d1, f1)
        flips = int((m0 != m1).sum())
        k0, _ = po.keypoints_sift(f0, P.resolution, 3, 3, P.keypoint_threshold)
        k1, _ = po.keypoints_sift(f1, P.resolution, 3, 3, P.keypoint_threshold)

        def key(k):                                # keypoints are octave-cloud points: compare them to a tenth of a millimetre
            return set(map(tuple, np.round(np.stack([k["x"], k["y"], k["z"]], 1).astype(np.float64) * 1e4).astype(np.int64)))
        s0, s1 = key(k0), key(k1)
        only0, only1 = len(s0 - s1), len(s1 - s0)
        b0 = set(map(bytes, np.stack([k0["x"], k0["y"], k0["z"]], 1).astype(np.float32)))
        b1 = set(map(bytes, np.stack([k1["x"], k1["y"], k1["z"]], 1).astype(np.float32)))
        bits_moved = len(b0 - b1)
        # FPFH rows of the keypoints the two runs share to 0.1 mm, in the first run's order
        n0, n1 = po.normals(f0, P.normal_radius), po.normals(f1, P.normal_radius)
        kk0, e0 = po.descriptors_fpfh(f0, n0, k0, P.descriptor_radius)
        kk1, e1 = po.descriptors_fpfh(f1, n1, k1, P.descriptor_radius)
        pos1 = {tuple(v): i for i, v in enumerate(np.round(np.stack([kk1["x"], kk1["y"], kk1["z"]], 1).astype(np.float64) * 1e4).astype(np.int64))}
        rows_bits = rows_1e3 = shared = 0
        for i, v in enumerate(np.round(np.stack([kk0["x"], kk0["y"], kk0["z"]], 1).astype(np.float64) * 1e4).astype(np.int64)):
            j = pos1.get(tuple(v))
            if j is None:
                continue
            shared += 1
            if e0[i].tobytes() != e1[j].tobytes():
                rows_bits += 1
                rows_1e3 += int(np.max(np.abs(e0[i] - e1[j])) > 1e-3)
        print(f"  map {mi} row 2 followed downstream (std::sort's order inside the voxels instead of input order): {xyz_diff} of {len(d0)} voxel centroids differ in "
              f"their bits; the outlier filter then keeps {int(m0.sum())} / {int(m1.sum())} points ({flips} decisions flip); SIFT finds {len(k0)} / {len(k1)} keypoints, "
              f"{only0} only with input order and {only1} only with std::sort's order ({100.0 * (only0 + only1) / max(len(k0), 1):.2f} % of the keypoints move by more than 0.1 mm or "
              f"appear / disappear; {bits_moved} keypoints have other coordinate bits); of the {shared} FPFH rows of shared keypoints {rows_bits} differ in some bit, "
              f"{rows_1e3} by more than 1e-3 in a bin (a bin is a percentage: 0 .. 100)", flush=True)


if os.environ.get("MM3D_AUDIT_DOWNSTREAM", "1") != "0":
    print("== row 2, downstream")
    for ci in ([int(a) for a in sys.argv[1:]] or [0, 1, 2, 4]):
        downstream_of_the_voxel_order(ci)
