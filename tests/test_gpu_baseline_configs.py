"""The BASELINE.json configurations at their full sizes through the reference's entry point
(estimateMapsTransforms, R/src/map_merging.cpp:188-275) on the MI355X.

What is checked at these sizes, where the CPU oracle cannot run a whole job in test time:
  * the library's stream scheduler never changes a bit: 1 stream == 16 streams for every pair record
    (transform, confidence, ICP trace, match / inlier counts), every configuration;
  * 16 x 500 000 (the headline configuration): the oracle runs maps 0 and 1 and pair (0, 1) (about a
    minute of CPU) and every stage is compared with the device's;
  * ground truth: the generator knows every map's pose.  Where the reference's algorithm itself finds
    the basin on this data (colour-aware descriptors + reciprocal matching + RANSAC) the recovered
    pair transforms are held against the ground truth with a stated bound.  FPFH + SAC_IA -- the
    configuration BASELINE.json quotes -- does NOT find it at these sizes, on the device or on the
    CPU path alike: FPFH sees geometry only and most SIFT keypoints of these scenes lie on the
    textured, geometrically bland ground, and SAC-IA tries 500 hypotheses, each from 3 random picks among
    10 nearest descriptors out of ~16 000.  That is the reference's behaviour, so the test there is
    parity with the oracle, not registration success.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FPFH, PFHRGB, SHOT = 2, 1, 4
MATCHING, SAC_IA = 0, 1

# name -> (maps, raw points per map, descriptor, method, scene / parameter overrides): BASELINE.json configs[1..4];
# configs[3] twice: on the generator's default 120 m outdoor windows and as BASELINE.json words it, "dense indoor"
# (30 m windows at resolution 0.05 -- nearly every neighbourhood overflows the LDS tiles there)
INDOOR = dict(window=30.0, resolution=0.05)
CONFIGS = {
    "4x200k_FPFH": (4, 200000, FPFH, SAC_IA, {}),
    "16x500k_FPFH_ICP": (16, 500000, FPFH, SAC_IA, {}),
    "8x2M_SIFT_SHOT": (8, 2000000, SHOT, SAC_IA, {}),
    "8x2M_SIFT_SHOT_dense_indoor": (8, 2000000, SHOT, SAC_IA, INDOOR),
    "64x50k_swarm": (64, 50000, FPFH, SAC_IA, {}),
}


@pytest.fixture(scope="module")
def workload(synth):
    cache = {}

    def get(n_maps, n_points, window=None):
        key = (n_maps, n_points, window)
        if key not in cache:
            cache.clear()                                   # one configuration resident at a time (8 x 2M = 256 MB)
            cache[key] = synth.cached_maps(n_maps, n_points, **({"window": window} if window else {}))
        return cache[key]

    return get


def xyz(a):
    return np.stack([a["x"], a["y"], a["z"]], axis=1)


def records_equal(a, b):
    return (len(a) == len(b)
            and np.array_equal(a["transform"].view(np.uint32), b["transform"].view(np.uint32))
            and np.array_equal(a["confidence"].view(np.uint64), b["confidence"].view(np.uint64))
            and all(np.array_equal(a[f], b[f]) for f in ("source_idx", "target_idx", "icp_iterations", "n_correspondences",
                                                          "n_inliers", "icp_correspondences")))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_configuration_runs_and_streams_do_not_change_a_bit(ctx, mm, workload, name):
    n_maps, n_points, desc, method, over = CONFIGS[name]
    raws, _, _ = workload(n_maps, n_points, over.get("window"))
    params = mm.MapMergingParams(descriptor_type=desc, estimation_method=method, refine_transform=1)
    if "resolution" in over:
        params.resolution = over["resolution"]
    runs = []
    for streams in (16, 1):
        ctx.setStreams(streams)
        ctx.srand(1)                                        # the reference's process starts at glibc seed 1
        runs.append(ctx.estimateMapsTransforms(raws, params, return_pairs=True))
    ctx.setStreams(1)
    (T16, p16), (T1, p1) = runs
    assert len(p16) == n_maps * (n_maps - 1) // 2           # every map has keypoints: every pair is estimated
    assert records_equal(p16, p1)
    assert len(T16) == len(T1) == n_maps and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(T16, T1))
    assert all(np.isfinite(t).all() and np.any(t) for t in T16)      # every map reached through the spanning tree
    it = p16["icp_iterations"]
    assert (it >= 1).all() and (it <= params.max_iterations).all()
    assert (p16["icp_correspondences"] >= 3).all()          # ICP stops below 3 correspondences (then iterations = 0)
    assert (p16["confidence"] > 0).all() and np.isfinite(p16["confidence"]).all()
    if method == SAC_IA:
        assert not p16["n_correspondences"].any() and not p16["n_inliers"].any()


def test_16x500k_stage_by_stage_against_the_oracle(ctx, po, mm, workload):
    """The headline configuration: maps 0 and 1 and pair (0, 1) on the CPU oracle, stage by stage."""
    raws, _, _ = workload(16, 500000)
    params = mm.MapMergingParams(descriptor_type=FPFH, estimation_method=SAC_IA, refine_transform=1)
    ctx.setStreams(16)
    ctx.srand(1)
    _, pairs = ctx.estimateMapsTransforms(raws, params, return_pairs=True)
    ctx.setStreams(1)
    assert pairs[0]["source_idx"] == 0 and pairs[0]["target_idx"] == 1
    ref, dev = [], []
    for i in (0, 1):
        d = po.downsample(raws[i], params.resolution)
        f = po.remove_outliers(d, params.descriptor_radius, params.outliers_min_neighbours)
        n = po.normals(f, params.normal_radius)
        kp_raw, _ = po.keypoints_sift(f, params.resolution, 3, 3, params.keypoint_threshold)
        kp, desc = po.descriptors_fpfh(f, n, kp_raw, params.descriptor_radius)
        ref.append(dict(filt=f, nrm=n, kp=kp, desc=desc))
        m = ctx.mapFeatures(ctx.cloud(raws[i]), params)
        dev.append(dict(filt=m.points.numpy(), kp=m.keypoints.numpy(), desc=m.descriptors.numpy()))
        m.free()
    for r, g in zip(ref, dev):
        # voxel grid + outlier filter: bit-exact
        assert np.array_equal(g["filt"].view(np.uint32), r["filt"].view(np.uint32))
        # keypoints and descriptors: the same arrays (sums run in the CPU path's neighbour order, libm restated)
        assert np.array_equal(xyz(g["kp"]).view(np.uint32), xyz(r["kp"]).view(np.uint32)), (len(g["kp"]), len(r["kp"]))
        assert np.array_equal(g["desc"].view(np.uint32), r["desc"].view(np.uint32))
    # pair (0, 1) is the first pair: the generator is in its initial state on both sides
    po.srand(1)
    T0, _, _ = po.sac_ia(ref[0]["kp"], ref[0]["desc"], ref[1]["kp"], ref[1]["desc"], params.inlier_threshold,
                         params.max_correspondence_distance, params.max_iterations)
    T_ref, it_ref = po.icp(ref[0]["filt"], ref[1]["filt"], T0, params.max_correspondence_distance, params.inlier_threshold,
                           params.max_iterations, params.transform_epsilon)
    score = po.transform_score(ref[0]["filt"], ref[1]["filt"], T_ref, params.max_correspondence_distance)
    got = pairs[0]["transform"].reshape(4, 4).T
    # ICP reduces in double on the device and in float on the CPU path: the stated tolerance (both clauses), confidence 1e-4 relative
    T_ex, it_ex = po.icp_double_sums(ref[0]["filt"], ref[1]["filt"], T0, params.max_correspondence_distance, params.max_iterations,
                                     params.transform_epsilon)
    check_pair_tolerance(po, got, T_ref, T_ex, pairs[0]["icp_iterations"], it_ref, it_ex, len(ref[0]["filt"]), what="headline pair (0, 1)")
    assert pairs[0]["confidence"] == pytest.approx(1.0 / score, rel=1e-4)
    # the reference's default method on the same maps: cross-match and inlier counts exact
    # (R/src/registration_visualisation.cpp:129-130), transform bit-equal
    corr = po.find_correspondences(ref[0]["desc"], ref[1]["desc"], int(params.matching_k))
    T_r, inl_r, _, _ = po.ransac(ref[0]["kp"], ref[1]["kp"], corr, params.inlier_threshold)
    d0, d1 = ctx.descriptors(dev[0]["desc"]), ctx.descriptors(dev[1]["desc"])
    got_corr = ctx.findFeatureCorrespondences(d0, d1, int(params.matching_k))
    assert len(got_corr) == len(corr) and np.array_equal(got_corr["index_match"], corr["index_match"])
    T_g, inl_g = ctx.estimateTransformFromCorrespondences(ctx.cloud(dev[0]["kp"]), ctx.cloud(dev[1]["kp"]), got_corr,
                                                          params.inlier_threshold)
    assert len(inl_g) == len(inl_r) and np.array_equal(T_g.view(np.uint32), T_r.view(np.uint32))


def oracle_threads():
    """Threads for the oracle's OpenMP loops: the container's CPU quota (threads beyond it only get the group throttled)."""
    import os
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def check_pair_tolerance(po, T_dev, T_oracle, T_exact, it_dev, it_oracle, it_exact, n_src, what=""):
    """THE stated pair-transform tolerance (oracle/pyoracle.py): both clauses."""
    fro_exact = float(np.linalg.norm(np.asarray(T_dev, np.float64) - np.asarray(T_exact, np.float64)))
    fro_oracle = float(np.linalg.norm(np.asarray(T_dev, np.float64) - np.asarray(T_oracle, np.float64)))
    assert int(it_dev) == int(it_exact) == int(it_oracle), (what, it_dev, it_oracle, it_exact)
    assert fro_exact <= po.TOL_T_EXACT, (what, fro_exact)
    cpu_noise = float(np.linalg.norm(np.asarray(T_oracle, np.float64) - np.asarray(T_exact, np.float64)))
    assert cpu_noise <= po.transform_tolerance(n_src), (what, cpu_noise, n_src)            # the CPU path's own noise is what the slope states
    assert fro_oracle <= po.transform_tolerance(n_src, cpu_noise), (what, fro_oracle, cpu_noise, n_src)
    return fro_oracle, fro_exact


def test_16x500k_all_120_pairs_within_the_stated_tolerance(ctx, po, mm, workload):
    """The WHOLE headline job against the oracle's restated estimateMapsTransforms (one rand() stream over all pairs,
    R/src/map_merging.cpp:256-269) and against its exact-arithmetic yardstick: every one of the 120 pair transforms
    within the stated tolerance, every ICP iteration count equal, confidences and global transforms within theirs.
    About two minutes of oracle on the box's cores."""
    import os
    import time
    raws, _, _ = workload(16, 500000)
    params = mm.MapMergingParams(descriptor_type=FPFH, estimation_method=SAC_IA, refine_transform=1)
    ctx.setStreams(16)
    ctx.srand(1)
    T_dev, dev = ctx.estimateMapsTransforms(raws, params, return_pairs=True)
    n_filtered = []
    for r in raws:
        m = ctx.mapFeatures(ctx.cloud(r), params)
        n_filtered.append(len(m.points))
        m.free()
    ctx.setStreams(1)
    op = po.params_default(); op.descriptor_type = FPFH; op.estimation_method = SAC_IA; op.refine_transform = 1
    po.set_threads(oracle_threads())
    po.set_exact_yardstick(True)
    po.srand(1)
    t0 = time.perf_counter()
    try:
        T_ref, ref = po.estimate_maps_transforms(raws, op)
        tr = po.last_run_traces()
        T_ex, it_ex, corr_ex = po.last_run_exact()
    finally:
        po.set_exact_yardstick(False)
        po.set_threads(1)
    t_cpu = time.perf_counter() - t0
    assert len(dev) == len(ref) == len(tr) == len(T_ex) == 120
    fo, fe = [], []
    for k in range(120):
        assert int(dev[k]["source_idx"]) == int(ref[k]["source_idx"]) and int(dev[k]["target_idx"]) == int(ref[k]["target_idx"])
        a, b = check_pair_tolerance(po, dev[k]["transform"], ref[k]["transform"], T_ex[k], dev[k]["icp_iterations"], tr[k]["icp_iterations"],
                                    it_ex[k], n_filtered[int(dev[k]["source_idx"])], what=f"pair {k}")
        fo.append(a); fe.append(b)
    conf = np.array([abs(float(a["confidence"]) / float(b["confidence"]) - 1.0) for a, b in zip(dev, ref)])
    assert conf.max() <= 1e-3, conf.max()
    g = np.array([np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) for a, b in zip(T_dev, T_ref)])
    assert g.max() <= 2e-3, g.max()
    # integer observable of the last ICP iteration: against exact arithmetic the counts agree; against the CPU path's
    # float-transformed cloud points at the max_correspondence_distance border flip (reported, not asserted)
    corr_eq_exact = int(sum(int(a["icp_correspondences"]) == int(c) for a, c in zip(dev, corr_ex)))
    corr_eq_oracle = int(sum(int(a["icp_correspondences"]) == int(t["icp_correspondences"]) for a, t in zip(dev, tr)))
    fo, fe = np.array(fo), np.array(fe)
    report = (f"120 pairs of 16 x 500000: ||T_dev - T_oracle||_F max {fo.max():.3e} median {np.median(fo):.3e} "
              f"(tolerance {po.transform_tolerance(max(n_filtered)):.2e} at {max(n_filtered)} source points); "
              f"||T_dev - T_exact||_F max {fe.max():.3e} median {np.median(fe):.3e} (tolerance {po.TOL_T_EXACT:g}); "
              f"ICP iteration counts equal 120 of 120 (oracle and exact); last-iteration correspondence counts equal: "
              f"{corr_eq_exact} of 120 vs exact arithmetic, {corr_eq_oracle} of 120 vs the CPU path; confidence rel max {conf.max():.3e}; "
              f"global transforms max {g.max():.3e}; oracle {t_cpu:.0f} s on {oracle_threads()} threads")
    print(report)
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "all_pairs_tolerance.txt"), "w").write(report + "\n")
    assert corr_eq_exact >= 110, corr_eq_exact               # (measured 114: the device forms T p in float, the yardstick in double)


def test_8x2M_dense_indoor_stage_by_stage_against_the_oracle(ctx, po, mm, workload):
    """configs[3] as BASELINE.json words it (dense indoor: 30 m windows, resolution 0.05; SIFT + SHOT-1344): maps 0 and 1
    and pair (0, 1) on the CPU oracle, stage by stage.  Nearly every neighbourhood of these clouds overflows the LDS tiles,
    so this is the test of the dense-spot paths at size (R/src/features.cpp:45-62, dispatch_descriptors.h:46)."""
    raws, _, _ = workload(8, 2000000, INDOOR["window"])
    params = mm.MapMergingParams(descriptor_type=SHOT, estimation_method=SAC_IA, refine_transform=1)
    params.resolution = INDOOR["resolution"]
    ctx.setStreams(16)
    ctx.srand(1)
    _, pairs = ctx.estimateMapsTransforms(raws, params, return_pairs=True)
    ctx.setStreams(1)
    assert pairs[0]["source_idx"] == 0 and pairs[0]["target_idx"] == 1
    po.set_threads(oracle_threads())
    try:
        ref = []
        for i in (0, 1):
            d = po.downsample(raws[i], params.resolution)
            f = po.remove_outliers(d, params.descriptor_radius, params.outliers_min_neighbours)
            n = po.normals(f, params.normal_radius)
            kp_raw, _ = po.keypoints_sift(f, params.resolution, 3, 3, params.keypoint_threshold)
            kp, desc = po.descriptors_shot(f, n, kp_raw, params.descriptor_radius)
            ref.append(dict(filt=f, kp=kp, desc=desc))
            m = ctx.mapFeatures(ctx.cloud(raws[i]), params)
            g = dict(filt=m.points.numpy(), kp=m.keypoints.numpy(), desc=m.descriptors.numpy())
            m.free()
            assert len(f) > 1000000                         # dense: more than a million points survive the filters
            assert np.array_equal(g["filt"].view(np.uint32), f.view(np.uint32))
            assert np.array_equal(xyz(g["kp"]).view(np.uint32), xyz(kp).view(np.uint32)), (len(g["kp"]), len(kp))
            same = (g["desc"].view(np.uint32) == desc.view(np.uint32)).all(axis=1)
            assert same.mean() >= 0.99 and np.abs(g["desc"] - desc).max() <= 2e-6, (same.mean(), np.abs(g["desc"] - desc).max())
        po.srand(1)
        T0, _, _ = po.sac_ia(ref[0]["kp"], ref[0]["desc"], ref[1]["kp"], ref[1]["desc"], params.inlier_threshold,
                             params.max_correspondence_distance, params.max_iterations)
        T_ref, it_ref = po.icp(ref[0]["filt"], ref[1]["filt"], T0, params.max_correspondence_distance, params.inlier_threshold,
                               params.max_iterations, params.transform_epsilon)
        T_ex, it_ex = po.icp_double_sums(ref[0]["filt"], ref[1]["filt"], T0, params.max_correspondence_distance, params.max_iterations,
                                         params.transform_epsilon)
        score = po.transform_score(ref[0]["filt"], ref[1]["filt"], T_ref, params.max_correspondence_distance)
    finally:
        po.set_threads(1)
    got = pairs[0]["transform"].reshape(4, 4).T
    check_pair_tolerance(po, got, T_ref, T_ex, pairs[0]["icp_iterations"], it_ref, it_ex, len(ref[0]["filt"]), what="dense indoor pair (0, 1)")
    assert pairs[0]["confidence"] == pytest.approx(1.0 / score, rel=1e-3)


def test_ground_truth_is_recovered_where_the_reference_algorithm_finds_it(ctx, mm, synth, workload):
    """4 x 200 000 with PFHRGB + MATCHING (+ RANSAC + ICP): every pair overlaps (>= 36 % of a window) and the
    algorithm finds the basin, so the device's answers are held against the generator's poses.
    Bounds: pair transform within 0.5 (Frobenius) of the ground truth -- ICP stops at the reference's loose
    transform_epsilon = 1e-2, a few centimetres to decimetres short -- median <= 0.1, global poses <= 0.2."""
    n_maps, n_points = 4, 200000
    raws, Tg, _ = workload(n_maps, n_points)
    params = mm.MapMergingParams(descriptor_type=PFHRGB, estimation_method=MATCHING, refine_transform=1)
    ctx.setStreams(16)
    ctx.srand(1)
    T, pairs = ctx.estimateMapsTransforms(raws, params, return_pairs=True)
    ctx.setStreams(1)
    errs = []
    for p in pairs:
        i, j = int(p["source_idx"]), int(p["target_idx"])
        assert synth.window_overlap(n_maps, n_points, i, j) >= 0.3
        errs.append(np.linalg.norm(p["transform"].reshape(4, 4).T - synth.relative_gt(Tg[i], Tg[j])))
        assert p["n_inliers"] >= 3 and p["n_correspondences"] >= p["n_inliers"]
    assert max(errs) <= 0.5 and np.median(errs) <= 0.1, errs
    ref = [k for k, t in enumerate(T) if np.array_equal(t, np.eye(4, dtype=np.float32))]
    assert len(ref) == 1
    for k in range(n_maps):
        # T[k] takes map k into the reference map's frame
        assert np.linalg.norm(T[k] - synth.relative_gt(Tg[k], Tg[ref[0]])) <= 0.2, k


def test_config0_two_clouds_of_10000_points_against_the_oracle(ctx, po, mm, synth):
    """BASELINE.json configs[0], literally: 2 overlapping synthetic clouds x 10 000 points, FPFH + SAC-IA + ICP -- the
    case the reference's registration_visualisation tool runs on the CPU.  The oracle runs the whole job; the device's
    features are its bits, the pair transform within 1e-3 (Frobenius), confidence 1e-3 relative, ICP trace exact."""
    raws, _, _ = synth.cached_maps(2, 10000)
    assert all(len(r) == 10000 for r in raws)
    params = mm.MapMergingParams(descriptor_type=FPFH, estimation_method=SAC_IA, refine_transform=1)
    op = po.params_default(); op.descriptor_type = FPFH; op.estimation_method = SAC_IA; op.refine_transform = 1
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([r.view(po.POINT) for r in raws], op)
    tr = po.last_run_traces()[0]
    T, pairs = ctx.estimateMapsTransforms(raws, params, return_pairs=True)
    assert len(pairs) == len(ref_pairs) == 1 and len(T) == len(ref_T) == 2
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= 1e-3
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)
    assert pairs[0]["icp_iterations"] == tr["icp_iterations"] and pairs[0]["icp_correspondences"] == tr["icp_correspondences"]
    for g, r in zip(T, ref_T):
        assert np.linalg.norm(g - r) <= 2e-3
    for r in raws:                                          # stage by stage on both clouds: the oracle's bits
        f = po.remove_outliers(po.downsample(r.view(po.POINT), op.resolution), op.descriptor_radius, op.outliers_min_neighbours)
        n = po.normals(f, op.normal_radius)
        kp, _ = po.keypoints_sift(f, op.resolution, 3, 3, op.keypoint_threshold)
        kp, d = po.descriptors_fpfh(f, n, kp, op.descriptor_radius)
        m = ctx.mapFeatures(ctx.cloud(r), params)
        assert np.array_equal(m.points.numpy().view(np.uint32).reshape(-1), f.view(np.uint32).reshape(-1))
        assert np.array_equal(xyz(m.keypoints.numpy()).view(np.uint32), xyz(kp).view(np.uint32))
        assert np.array_equal(m.descriptors.numpy().view(np.uint32), d.view(np.uint32))
        m.free()


def test_lattice_scenes_fpfh_sac_ia_recovers_the_ground_truth(ctx, po, mm, synth):
    """4 x 200 000 points of the 'lattice' scene family (overlapping maps share their surface samples, synth.lattice_map),
    windows three quarters of a side apart in the loop (every pair shares >= 2/3 of a window), FPFH + SAC-IA + ICP with
    20 000 hypotheses instead of the reference's default 500.  With the default the algorithm does not find the basin on
    either scene family, on the device or on the CPU oracle alike: a hypothesis is three random picks among the ten
    nearest descriptors of three random keypoints, i.e. at best (overlap x 1/10)^3 ~ 3e-4 per draw (scratch notes in
    DESIGN.md section 6 give the oracle's numbers).  Here five of the six pairs come out right on the oracle (errors
    0.04 .. 0.66 Frobenius, the sixth has the lowest confidence by a factor of two); the device must do the same."""
    n_maps, n_points, step = 4, 200000, 0.25
    raws, Tg, _ = synth.cached_maps(n_maps, n_points, family="lattice", overlap_step=step)
    params = mm.MapMergingParams(descriptor_type=FPFH, estimation_method=SAC_IA, refine_transform=1, max_iterations=20000)
    ctx.setStreams(8)
    ctx.srand(1)
    T, pairs = ctx.estimateMapsTransforms(raws, params, return_pairs=True)
    ctx.setStreams(1)
    assert len(pairs) == 6
    errs, conf = [], []
    for p in pairs:
        i, j = int(p["source_idx"]), int(p["target_idx"])
        assert synth.window_overlap(n_maps, n_points, i, j, overlap_step=step) >= 0.6
        errs.append(float(np.linalg.norm(p["transform"].reshape(4, 4).T - synth.relative_gt(Tg[i], Tg[j]))))
        conf.append(float(p["confidence"]))
    good = [e <= 1.0 for e in errs]
    assert sum(good) >= 5, (errs, conf)
    # ICP iterates on these scenes (the SAC-IA poses are decimetres off, not metres)
    assert (pairs["icp_iterations"] >= 1).all()
    # the pair that failed, if any, is the one the pose graph trusts least
    if sum(good) < 6:
        assert np.argmin(conf) == good.index(False), (errs, conf)
    # and the features are still the oracle's bits on this scene family.  (They were not when it was first run: map 1
    # has pairs of points whose Darboux source / target choice -- acos(|angle1|) > acos(|angle2|) in double -- is decided
    # by a cosine rounded above 1 or by two angles below 2^-28, where "acos is decreasing" is not the whole truth;
    # device_util.hpp::acos_abs_greater.)
    po.set_threads(16)
    r = raws[1].view(po.POINT)
    f = po.remove_outliers(po.downsample(r, 0.1), 0.8, 50)
    n = po.normals(f, 0.6)
    kp, _ = po.keypoints_sift(f, 0.1, 3, 3, 5.0)
    kp, d = po.descriptors_fpfh(f, n, kp, 0.8)
    po.set_threads(1)
    m = ctx.mapFeatures(ctx.cloud(raws[1]), params)
    assert np.array_equal(xyz(m.keypoints.numpy()).view(np.uint32), xyz(kp).view(np.uint32))
    assert np.array_equal(m.descriptors.numpy().view(np.uint32), d.view(np.uint32))
    m.free()
