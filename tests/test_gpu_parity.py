"""GPU parity: every stage of the HIP path against the CPU oracle on identical seeded inputs.

All calls go through the C ABI (libmm3d.so via map_merge_amd).  Integer/index results must be
bit-exact; floating-point stages carry their tolerance in the test.  Run with `-m gpu` on the
MI355X box.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_RAW = 12000
RES, R_DESC, R_NRM, MIN_NB = 0.1, 0.8, 0.6, 50


def xyz(a):
    return np.stack([a["x"], a["y"], a["z"]], axis=1)


@pytest.fixture(scope="module")
def scene(po, synth):
    """Two overlapping synthetic maps and every oracle stage on them."""
    world, maps = synth.synth_maps(2, N_RAW, overlap_step=0.35)
    out = []
    for x, c, T in maps:
        raw = synth.pack_points(x, c)
        down = po.downsample(raw, RES)
        filt = po.remove_outliers(down, R_DESC, MIN_NB)
        nrm = po.normals(filt, R_NRM)
        kp_raw, scales = po.keypoints_sift(filt, RES, 3, 3, 5.0)
        kp, desc = po.descriptors_fpfh(filt, nrm, kp_raw, R_DESC)
        out.append(dict(raw=raw, down=down, filt=filt, nrm=nrm, kp_raw=kp_raw, kp=kp, desc=desc, T=T))
    return out


def test_downsample_bit_exact(ctx, scene):
    for m in scene:
        got = ctx.downSample(ctx.cloud(m["raw"]), RES).numpy()
        assert got.shape == m["down"].shape
        assert np.array_equal(got.view(np.uint32), m["down"].view(np.uint32))


def test_downsample_of_a_very_dense_spot(ctx, po, mm):
    """Thousands of raw points inside one voxel (a bin of the counting sort overflows and the stable radix sort takes
    over): the centroid is still summed in input order, bit for bit."""
    rng = np.random.default_rng(5)
    a = np.zeros(9000, dtype=mm.POINT)
    a["x"][:6000] = 3.0 + rng.uniform(0, 0.09, 6000); a["y"][:6000] = 1.0 + rng.uniform(0, 0.09, 6000); a["z"][:6000] = rng.uniform(0, 0.09, 6000)
    a["x"][6000:] = rng.uniform(-20, 20, 3000); a["y"][6000:] = rng.uniform(-20, 20, 3000); a["z"][6000:] = rng.uniform(0, 3, 3000)
    a["rgba"] = 0xFF000000 | rng.integers(0, 1 << 24, 9000).astype(np.uint32)
    got = ctx.downSample(ctx.cloud(a), 0.1).numpy()
    ref = po.downsample(a, 0.1)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_downsample_edge_cases(ctx, po, mm):
    empty = np.empty(0, dtype=mm.POINT)
    assert len(ctx.downSample(ctx.cloud(empty), 0.1)) == 0
    one = np.zeros(1, dtype=mm.POINT); one["x"] = 1.5; one["rgba"] = 0xFF102030
    got = ctx.downSample(ctx.cloud(one), 0.1).numpy()
    assert np.array_equal(got, po.downsample(one, 0.1))
    # leaf too small for int32 voxel indices: VoxelGrid returns the input unchanged
    far = np.zeros(3, dtype=mm.POINT); far["x"] = [0, 1000, 2000]; far["y"] = [0, 1500, 3000]; far["z"] = [0, 500, 900]
    got = ctx.downSample(ctx.cloud(far), 0.001).numpy()
    assert np.array_equal(got, far) and np.array_equal(po.downsample(far, 0.001), far)
    # non-finite points are skipped
    nf = np.zeros(4, dtype=mm.POINT); nf["x"] = [0.01, np.nan, 0.02, np.inf]; nf["rgba"] = 0xFF646464
    assert np.array_equal(ctx.downSample(ctx.cloud(nf), 0.1).numpy().view(np.uint32),
                          po.downsample(nf, 0.1).view(np.uint32))


def test_what_a_voxel_filtered_cloud_promises_its_grids(ctx, po, mm):
    """downSample's output carries the leaf of its voxel grid (every centroid within two leaves of a member of its voxel), and
    removeOutliers' subset inherits it: a grid build on such a cloud need not ask the device whether a cell outgrew the
    counting sort (csrc/types.hpp: voxel_leaf).  A caller's raw cloud promises nothing, the int32-overflow pass-through promises
    nothing -- and neither does a filter whose float sums carried a centroid away: 4 000 points per voxel 300 km from the origin
    (ulp 0.03 m, partial sums of 1e9 with an ulp of 64 - 128) put centroids metres from their voxels, the CPU path's bits all
    the same, and the filters after it (their grid asks the device again) still agree with the oracle bit for bit."""
    import ctypes as C
    leaf_of = mm.lib().mm3d_debug_cloud_voxel_leaf
    leaf_of.restype = C.c_float
    leaf_of.argtypes = [C.c_void_p]
    rng = np.random.default_rng(9)
    a = np.zeros(60000, dtype=mm.POINT)
    a["x"], a["y"], a["z"] = rng.uniform(-6, 6, 60000), rng.uniform(-6, 6, 60000), rng.uniform(0, 0.3, 60000)
    a["rgba"] = 0xFF000000 | rng.integers(0, 1 << 24, 60000).astype(np.uint32)
    raw = ctx.cloud(a)
    assert leaf_of(raw._h) == 0.0
    down = ctx.downSample(raw, 0.1)
    assert leaf_of(down._h) == np.float32(0.1)
    filt = ctx.removeOutliers(down, 0.8, 50)
    assert leaf_of(filt._h) == np.float32(0.1) and 0 < len(filt) <= len(down)
    ref = po.remove_outliers(po.downsample(a, 0.1), 0.8, 50)
    assert np.array_equal(filt.numpy().view(np.uint32), ref.view(np.uint32))
    far = np.zeros(3, dtype=mm.POINT); far["x"] = [0, 1000, 2000]; far["y"] = [0, 1500, 3000]; far["z"] = [0, 500, 900]
    assert leaf_of(ctx.downSample(ctx.cloud(far), 0.001)._h) == 0.0
    # centroids carried away by their own float sums
    n_vox, per = 24, 4000
    b = np.zeros(n_vox * per, dtype=mm.POINT)
    base = np.float32(3.0e5)
    cells = rng.integers(0, 40, (n_vox, 3)).astype(np.float32) * np.float32(0.8)
    p = np.repeat(cells, per, axis=0) + rng.uniform(0, 0.09, (n_vox * per, 3)).astype(np.float32)
    order = rng.permutation(len(p))
    b["x"], b["y"], b["z"] = (base + p[order, 0]).astype(np.float32), (base + p[order, 1]).astype(np.float32), p[order, 2]
    b["rgba"] = 0xFF000000 | rng.integers(0, 1 << 24, len(b)).astype(np.uint32)
    ref = po.downsample(b, 0.1)
    got = ctx.downSample(ctx.cloud(b), 0.1)
    assert np.array_equal(got.numpy().view(np.uint32), ref.view(np.uint32))
    # (some centroid of the oracle's lies farther than two leaves from every input point of its voxel's neighbourhood)
    d = np.abs(ref["x"][:, None].astype(np.float64) - b["x"][None, ::97].astype(np.float64)).min(axis=1)
    assert d.max() > 0.2, d.max()
    assert leaf_of(got._h) == 0.0
    ref2 = po.remove_outliers(ref, 0.8, 0)
    got2 = ctx.removeOutliers(got, 0.8, 0)
    assert leaf_of(got2._h) == 0.0 and np.array_equal(got2.numpy().view(np.uint32), ref2.view(np.uint32))


def test_remove_outliers_exact(ctx, scene):
    for m in scene:
        got = ctx.removeOutliers(ctx.cloud(m["down"]), R_DESC, MIN_NB).numpy()
        assert np.array_equal(got.view(np.uint32), m["filt"].view(np.uint32))


def test_remove_outliers_thresholds(ctx, po, scene):
    d = scene[0]["down"][:3000]
    for radius, k in [(0.3, 5), (0.5, 20), (0.8, 200), (0.2, 0)]:
        got = ctx.removeOutliers(ctx.cloud(d), radius, k).numpy()
        assert np.array_equal(got.view(np.uint32), po.remove_outliers(d, radius, k).view(np.uint32)), (radius, k)


def test_normals(ctx, scene):
    """computeSurfaceNormals: the float raw moments are accumulated in radiusSearch's (distance, index) order and
    the eigen solve uses the restated glibc atan2f / cosf / sinf, so normals and curvature are the oracle's bits."""
    for m in scene:
        got = ctx.computeSurfaceNormals(ctx.cloud(m["filt"]), R_NRM).numpy()
        ref = m["nrm"]
        assert not np.isnan(ref["nx"]).any()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # fewer than 3 neighbours -> NaN; non-finite input -> NaN
    few = scene[0]["filt"][:4].copy()
    few["x"] = [0.0, 0.1, 50.0, np.nan]; few["y"] = 0.0; few["z"] = 0.0
    got = ctx.computeSurfaceNormals(ctx.cloud(few), R_NRM).numpy()
    assert np.isnan(got["nx"]).all() and np.isnan(got["curvature"]).all()


def test_neighbour_lists_longer_than_a_rank_batch(ctx, po, scene):
    """Neighbourhoods of several hundred points (more than the 128 / 256 entries one rank batch of sorted_nb.hpp holds):
    the lists are then written bucket by bucket and ranked in bucket-aligned batches.  Same bits as the oracle for
    normals (raw-moment chains), FPFH rows (weighting chains) and SIFT keypoints (Gaussian chains)."""
    m = scene[0]
    big_r = 1.9                                           # ~ 1000 neighbours on this scene
    cloud = ctx.cloud(m["filt"])
    got = ctx.computeSurfaceNormals(cloud, big_r).numpy()
    ref = po.normals(m["filt"], big_r)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    kp = m["kp_raw"][:300].copy()
    kp_ref, ref_d = po.descriptors_fpfh(m["filt"], m["nrm"], kp, big_r)
    k = ctx.cloud(kp)
    got_d = ctx.computeLocalDescriptors(cloud, ctx.normals(m["nrm"]), k, 2, big_r).numpy()
    assert got_d.shape == ref_d.shape and np.array_equal(got_d.view(np.uint32), ref_d.view(np.uint32))
    # SIFT with a coarse base scale: 3 sigma of the widest scale is ~ 2.3 m
    got_k = ctx.detectKeypoints(cloud, None, 0, 2.0, R_NRM, 0.3).numpy()
    ref_k, _ = po.keypoints_sift(m["filt"], 0.3, 3, 3, 2.0)
    assert len(got_k) == len(ref_k) and np.array_equal(xyz(got_k).view(np.uint32), xyz(ref_k).view(np.uint32))


def test_sift_keypoints(ctx, scene):
    """detectKeypoints(SIFT): the Gaussian sums run in radiusSearch's order with glibc's expf restated, so the
    keypoints are the oracle's, in the oracle's (octave, index, scale) order."""
    for m in scene:
        got = ctx.detectKeypoints(ctx.cloud(m["filt"]), None, 0, 5.0, R_NRM, RES).numpy()
        ref = m["kp_raw"]
        assert len(got) == len(ref) > 100
        assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))
        assert (got["rgba"] == 0).all()


def test_sift_blob_known_answer(ctx, po):
    """The blob-detector property through the device path (tests/test_oracle_cpu.py::test_sift_known_answers has the CPU side):
    a flat lattice with ONE Gaussian intensity blob gives one keypoint, on the blob; a uniform lattice and a blob under the
    contrast threshold give none.  And the device agrees with the oracle on each."""
    gx, gy = np.meshgrid(np.arange(-40, 41) * 0.1, np.arange(-40, 41) * 0.1)
    pts = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1).astype(np.float32)
    for amplitude, sigma, expect in ((150.0, 0.25, 1), (150.0, 0.4, 1), (-55.0, 0.4, 1), (4.0, 0.4, 0), (0.0, 0.4, 0)):
        c = np.zeros(len(pts), dtype=po.POINT)
        c["x"], c["y"], c["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
        g = amplitude * np.exp(-((pts[:, 0] - 0.33) ** 2 + (pts[:, 1] + 0.21) ** 2) / (2 * sigma * sigma))
        lum = np.clip(60.0 + g, 0, 255).astype(np.uint32)
        c["rgba"] = (0xFF << 24) | (lum << 16) | (lum << 8) | lum
        got = ctx.detectKeypoints(ctx.cloud(c), None, 0, 5.0, R_NRM, 0.1).numpy()
        ref, _ = po.keypoints_sift(c, 0.1, 3, 3, 5.0)
        assert len(got) == len(ref) == expect
        assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))
        if expect:
            assert np.hypot(got["x"] - 0.33, got["y"] + 0.21).max() < 0.6 * sigma


def test_sift_ties_across_scales_are_not_extrema(ctx, po):
    """Equality at a point's own scale, STRICT comparisons against the adjacent scales (sift_keypoint.hpp,
    findScaleSpaceExtrema): a uniform grey-128 lattice with the contrast threshold at zero makes every DoG value 0.0f and
    every comparison a tie (tests/test_oracle_cpu.py has the arithmetic) -- no keypoint, on the device as on the oracle;
    and a scene with ordinary texture still gives the oracle's keypoints with the threshold at zero."""
    gx, gy = np.meshgrid(np.arange(-20, 21) * 0.1, np.arange(-20, 21) * 0.1)
    pts = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1).astype(np.float32)
    c = np.zeros(len(pts), dtype=po.POINT)
    c["x"], c["y"], c["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    c["rgba"] = 0xFF808080
    assert len(po.keypoints_sift(c, 0.1, 3, 3, 0.0)[0]) == 0
    assert len(ctx.detectKeypoints(ctx.cloud(c), None, 0, 0.0, R_NRM, 0.1)) == 0
    # a step edge in grey levels that are powers of two: many exact ties among the responses away from the edge
    c["rgba"] = np.where(c["x"] < 0.0, 0xFF404040, 0xFF808080).astype(np.uint32)
    ref, _ = po.keypoints_sift(c, 0.1, 3, 3, 0.0)
    got = ctx.detectKeypoints(ctx.cloud(c), None, 0, 0.0, R_NRM, 0.1).numpy()
    assert len(got) == len(ref)
    assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))


def test_sift_keypoints_where_the_25_nearest_reach_beyond_the_scale_space_ball(ctx, po, scene):
    """The extremum test reads a point's 25 nearest neighbours from the scale-space kernel's sorted list when the
    3 sigma_max ball holds that many, and searches for them otherwise.  A cloud thinned to a fifth (most balls hold
    fewer than 25 points in the first octave, more in the later ones) takes both routes; the keypoints must be the
    oracle's either way."""
    rng = np.random.default_rng(5)
    filt = scene[0]["filt"]
    thin = filt[np.sort(rng.choice(len(filt), len(filt) // 5, replace=False))].copy()
    ref, _ = po.keypoints_sift(thin, RES, 3, 3, 1.0)
    got = ctx.detectKeypoints(ctx.cloud(thin), None, 0, 1.0, R_NRM, RES).numpy()
    assert len(got) == len(ref) > 20
    assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))


def test_sift_keypoints_with_a_dense_spot(ctx, po, scene):
    """A solid block of points (every 0.1 m voxel of a 1.5 m cube occupied: ~1 800 neighbours within 3 sigma_max)
    inside an ordinary scene: its work items overflow the first octave's LDS tile, go through the large
    configuration and, where that does not hold them either, the global-memory lists; the extremum test is taken
    again after each.  Same keypoints as the oracle."""
    rng = np.random.default_rng(9)
    filt = scene[0]["filt"]
    c0 = xyz(filt).mean(axis=0)
    blob = np.zeros(9000, dtype=filt.dtype)
    p = rng.uniform(-0.75, 0.75, (9000, 3)).astype(np.float32) + c0.astype(np.float32)
    blob["x"], blob["y"], blob["z"] = p[:, 0], p[:, 1], p[:, 2]
    blob["rgba"] = rng.integers(0, 1 << 24, 9000).astype(np.uint32)
    cloud = np.concatenate([filt, blob])
    ref, _ = po.keypoints_sift(cloud, RES, 3, 3, 5.0)
    got = ctx.detectKeypoints(ctx.cloud(cloud), None, 0, 5.0, R_NRM, RES).numpy()
    assert len(got) == len(ref) > 100
    assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))


def test_fpfh(ctx, scene):
    for m in scene:
        pts, nrm = ctx.cloud(m["filt"]), ctx.normals(m["nrm"])
        kp = ctx.cloud(m["kp_raw"])
        desc = ctx.computeLocalDescriptors(pts, nrm, kp, 2, R_DESC)
        got, ref = desc.numpy(), m["desc"]
        assert got.shape == ref.shape
        assert np.array_equal(kp.numpy().view(np.uint32), m["kp"].view(np.uint32))   # same pruning
        # each 11-bin block sums to 100
        assert np.allclose(got.reshape(len(got), 3, 11).sum(axis=2), 100.0, atol=1e-2)
        # bins counted in integers with glibc's atan2f restated, weighting in radiusSearch's order: the oracle's bits
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_fpfh_pairs_shared_inside_a_block_and_their_ties(ctx, po):
    """The SPFH kernel evaluates a pair of support points of one 256-point block once and votes into both histograms
    (computePairFeatures is symmetric under the swap, tests/test_oracle_cpu.py), EXCEPT when the pair's two angles tie: then
    the two calls differ and both are evaluated.  Three surfaces against the oracle's bits: a lattice plane with one normal for
    every point (angle1 == angle2 on every pair: all ties), the same with every other normal flipped (|angle1| == |angle2|),
    and a noisy sheet with duplicated points (f4 == 0 pairs) and some unnormalised / zero normals (|angle| > 1: no switch
    either way)."""
    rng = np.random.default_rng(5)
    def cloud(xyz):
        c = np.zeros(len(xyz), dtype=po.POINT)
        c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        c["rgba"] = 0xFF808080
        return c
    def normals(n):
        out = np.zeros(len(n), dtype=po.NORMAL)
        out["nx"], out["ny"], out["nz"] = n[:, 0], n[:, 1], n[:, 2]
        return out
    gx, gy = np.meshgrid(np.arange(60, dtype=np.float32) * 0.1, np.arange(50, dtype=np.float32) * 0.1)
    plane = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 1.5, np.float32)], axis=1)
    up = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(plane), 1))
    flip = up.copy(); flip[::2] *= -1.0
    sheet = plane + rng.normal(0, 0.02, plane.shape).astype(np.float32)
    sheet[100:130] = sheet[200:230]                                   # coincident points
    nsh = up + rng.normal(0, 0.2, up.shape).astype(np.float32)
    nsh /= np.linalg.norm(nsh, axis=1, keepdims=True)
    nsh[300:340] *= 2.5                                               # |angle| can exceed 1
    nsh[400:420] = 0.0
    for xyz, nrm in ((plane, up), (plane, flip), (sheet, nsh.astype(np.float32))):
        pts, nn = cloud(xyz), normals(nrm)
        kp = pts[rng.choice(len(pts), 120, replace=False)].copy()
        kp_ref, desc_ref = po.descriptors_fpfh(pts, nn, kp, 0.45)
        k = ctx.cloud(kp)
        desc = ctx.computeLocalDescriptors(ctx.cloud(pts), ctx.normals(nn), k, 2, 0.45)
        got = desc.numpy()
        assert got.shape == desc_ref.shape and len(got) > 100
        assert np.array_equal(k.numpy().view(np.uint32), kp_ref.view(np.uint32))
        assert np.array_equal(got.view(np.uint32), desc_ref.view(np.uint32))


def test_fpfh_prunes_isolated_keypoints(ctx, po, scene, mm):
    m = scene[0]
    kp = m["kp_raw"][:50].copy()
    kp["x"][7] += 500.0          # no surface point within the radius -> NaN descriptor -> pruned
    kp["z"][20] -= 300.0
    kp_ref, desc_ref = po.descriptors_fpfh(m["filt"], m["nrm"], kp, R_DESC)
    k = ctx.cloud(kp)
    desc = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), k, 2, R_DESC)
    assert len(desc) == len(desc_ref) == 48
    assert np.array_equal(k.numpy().view(np.uint32), kp_ref.view(np.uint32))


def test_correspondences_exact(ctx, scene):
    a, b = scene
    da, db = ctx.descriptors(a["desc"]), ctx.descriptors(b["desc"])
    for k in (1, 5, 10):
        got = ctx.findFeatureCorrespondences(da, db, k)
        import __graft_entry__ as ge
        ref = ge.load_oracle().find_correspondences(a["desc"], b["desc"], k)
        assert np.array_equal(got["index_query"], ref["index_query"])
        assert np.array_equal(got["index_match"], ref["index_match"])
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32))


def test_desc_knn_mfma_path_exact(ctx, po, mm, scene):
    """Large enough for the matrix-core candidate stage; results must still be the exact FLANN-order
    k-NN (indices and distance bits), and almost no row may need the exact fallback."""
    rng = np.random.default_rng(7)
    base = np.concatenate([scene[0]["desc"], scene[1]["desc"]])
    # FPFH-like rows: resample real descriptors with small perturbations, plus exact duplicates (ties)
    A = base[rng.integers(0, len(base), 1500)] + rng.normal(0, 0.3, (1500, 33)).astype(np.float32)
    B = base[rng.integers(0, len(base), 2100)] + rng.normal(0, 0.3, (2100, 33)).astype(np.float32)
    B[100:140] = B[200:240]                      # duplicated targets: ties must go to the lower index
    A[:20] = B[300:320]                          # zero distances
    A, B = A.astype(np.float32), B.astype(np.float32)
    L = mm.lib()
    L.mm3d_debug_knn_fallback_rows.restype = L.mm3d_debug_knn_rows.restype = __import__("ctypes").c_longlong
    L.mm3d_set_debug(ctx._h, 1)
    da, db = ctx.descriptors(A), ctx.descriptors(B)
    for k in (1, 5, 10, 16):
        got = ctx.findFeatureCorrespondences(da, db, k)
        ref = po.find_correspondences(A, B, k)
        assert np.array_equal(got["index_query"], ref["index_query"]), k
        assert np.array_equal(got["index_match"], ref["index_match"]), k
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32)), k
    rows, fb = L.mm3d_debug_knn_rows(ctx._h), L.mm3d_debug_knn_fallback_rows(ctx._h)
    L.mm3d_set_debug(ctx._h, 0)
    assert rows > 0 and fb <= 0.05 * rows, (rows, fb)   # the certificate holds for nearly every row


def test_desc_knn_filter_path_exact(ctx, po, mm, scene):
    """Short rows against many targets take the threshold filter (thresholds from a sample of the target tiles,
    candidates = every target under its query's threshold): the same exact FLANN-order k-NN, the SAC-IA shape
    (few queries, targets split over parts) and the matching shape (both directions), ties and zero distances
    included; almost no row may need the exact fallback."""
    rng = np.random.default_rng(17)
    base = np.concatenate([scene[0]["desc"], scene[1]["desc"]])
    L = mm.lib()
    L.mm3d_debug_knn_fallback_rows.restype = L.mm3d_debug_knn_rows.restype = __import__("ctypes").c_longlong
    for na, nb in ((700, 9000), (5000, 6000)):
        A = (base[rng.integers(0, len(base), na)] + rng.normal(0, 0.3, (na, 33))).astype(np.float32)
        B = (base[rng.integers(0, len(base), nb)] + rng.normal(0, 0.3, (nb, 33))).astype(np.float32)
        B[100:140] = B[200:240]
        A[:20] = B[300:320]
        L.mm3d_set_debug(ctx._h, 1)
        r0, f0 = L.mm3d_debug_knn_rows(ctx._h), L.mm3d_debug_knn_fallback_rows(ctx._h)
        da, db = ctx.descriptors(A), ctx.descriptors(B)
        for k in (1, 5, 10):
            got = ctx.findFeatureCorrespondences(da, db, k)
            ref = po.find_correspondences(A, B, k)
            assert np.array_equal(got["index_query"], ref["index_query"]), (na, nb, k)
            assert np.array_equal(got["index_match"], ref["index_match"]), (na, nb, k)
            assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32)), (na, nb, k)
        rows, fb = L.mm3d_debug_knn_rows(ctx._h) - r0, L.mm3d_debug_knn_fallback_rows(ctx._h) - f0
        L.mm3d_set_debug(ctx._h, 0)
        assert rows > 0 and fb <= 0.05 * rows, (na, nb, rows, fb)


def test_desc_knn_filter_candidate_overflow_is_searched_exactly(ctx, po, mm, scene):
    """Thousands of identical target rows put more candidates under a query's threshold than its buffer holds (512):
    the row loses its certificate and is searched exactly inside the re-rank kernel.  Indices (ties to the lower
    index) and distance bits as the oracle's."""
    rng = np.random.default_rng(23)
    base = np.concatenate([scene[0]["desc"], scene[1]["desc"]])
    nb = 4000
    B = (base[rng.integers(0, len(base), nb)] + rng.normal(0, 0.3, (nb, 33))).astype(np.float32)
    B[500:2500] = B[3000]                          # 2 000 copies of one row
    A = (base[rng.integers(0, len(base), 300)] + rng.normal(0, 0.3, (300, 33))).astype(np.float32)
    A[:40] = B[3000] + rng.normal(0, 1e-3, (40, 33)).astype(np.float32)   # queries whose neighbours ARE the copies
    A[40:50] = B[3000]                             # and exact hits: 2 001 targets at distance zero
    L = mm.lib()
    L.mm3d_debug_knn_fallback_rows.restype = L.mm3d_debug_knn_rows.restype = __import__("ctypes").c_longlong
    L.mm3d_set_debug(ctx._h, 1)
    f0 = L.mm3d_debug_knn_fallback_rows(ctx._h)
    da, db = ctx.descriptors(A), ctx.descriptors(B)
    for k in (1, 10):
        got = ctx.findFeatureCorrespondences(da, db, k)
        ref = po.find_correspondences(A, B, k)
        assert np.array_equal(got["index_query"], ref["index_query"]), k
        assert np.array_equal(got["index_match"], ref["index_match"]), k
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32)), k
    fb = L.mm3d_debug_knn_fallback_rows(ctx._h) - f0
    L.mm3d_set_debug(ctx._h, 0)
    assert fb >= 40, fb                            # the rows among the copies did take the exact search


def test_matching_k_is_any_positive_number(ctx, po, mm, scene):
    """matching_k is an arbitrary size_t in the reference (R/src/map_merging.cpp:43-47 -> FLANN nearestKSearch):
    beyond the 16 neighbours the register kernels keep, the plain exact kernel takes over; more neighbours than
    rows exist behave like "all of them".  Indices and distance bits as the oracle's, ties by index."""
    rng = np.random.default_rng(11)
    base = np.concatenate([scene[0]["desc"], scene[1]["desc"]])
    A = (base[rng.integers(0, len(base), 300)] + rng.normal(0, 0.3, (300, 33))).astype(np.float32)
    B = (base[rng.integers(0, len(base), 260)] + rng.normal(0, 0.3, (260, 33))).astype(np.float32)
    B[10:30] = B[40:60]
    A[:8] = B[100:108]
    da, db = ctx.descriptors(A), ctx.descriptors(B)
    for k in (17, 40, 260, 5000):
        got = ctx.findFeatureCorrespondences(da, db, k)
        ref = po.find_correspondences(A, B, min(k, 300))
        assert np.array_equal(got["index_query"], ref["index_query"]), k
        assert np.array_equal(got["index_match"], ref["index_match"]), k
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32)), k
    # wide rows through the same kernel
    W = rng.normal(0, 1, (90, 125)).astype(np.float32)
    V = rng.normal(0, 1, (70, 125)).astype(np.float32)
    got = ctx.findFeatureCorrespondences(ctx.descriptors(W, mm.Descriptor.PFH), ctx.descriptors(V, mm.Descriptor.PFH), 24)
    ref = po.find_correspondences(W, V, 24)
    assert np.array_equal(got["index_match"], ref["index_match"]) and np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32))


def test_ransac_exact(ctx, po, scene):
    a, b = scene
    corr = po.find_correspondences(a["desc"], b["desc"], 5)
    T_ref, inl_ref, iters, best = po.ransac(a["kp"], b["kp"], corr, 0.5)
    T, inl = ctx.estimateTransformFromCorrespondences(ctx.cloud(a["kp"]), ctx.cloud(b["kp"]), corr, 0.5)
    assert len(inl) == len(inl_ref)                       # inlier count exact
    assert np.array_equal(inl["index_query"], inl_ref["index_query"])
    assert np.array_equal(T.view(np.uint32), T_ref.view(np.uint32))
    # self-registration: a rigid copy must be recovered with every correspondence an inlier
    R = np.array([[0.8, -0.6, 0], [0.6, 0.8, 0], [0, 0, 1.0]])
    moved = a["kp"].copy()
    p = xyz(a["kp"]).astype(np.float64) @ R.T + np.array([1.0, -2.0, 0.5])
    moved["x"], moved["y"], moved["z"] = p[:, 0], p[:, 1], p[:, 2]
    ident = np.zeros(len(moved), dtype=corr.dtype)
    ident["index_query"] = ident["index_match"] = np.arange(len(moved))
    T2, inl2 = ctx.estimateTransformFromCorrespondences(ctx.cloud(a["kp"]), ctx.cloud(moved), ident, 0.5)
    T2r, inl2r, _, _ = po.ransac(a["kp"], moved, ident, 0.5)
    assert len(inl2) == len(inl2r) == len(moved)
    assert np.array_equal(T2.view(np.uint32), T2r.view(np.uint32))
    assert np.allclose(T2[:3, :3], R, atol=1e-4)
    # too few correspondences -> zero matrix, no inliers (R/src/matching.cpp:128-133)
    T3, inl3 = ctx.estimateTransformFromCorrespondences(ctx.cloud(a["kp"]), ctx.cloud(b["kp"]), corr[:2], 0.5)
    assert not T3.any() and len(inl3) == 0


def test_sac_ia_exact(ctx, po, scene):
    a, b = scene
    po.srand(1)
    T_ref, best_it, best_err = po.sac_ia(a["kp"], a["desc"], b["kp"], b["desc"], 0.5, 1.0, 500)
    ctx.srand(1)
    T = ctx.estimateTransformFromDescriptorsSets(ctx.cloud(a["kp"]), ctx.descriptors(a["desc"]), ctx.cloud(b["kp"]),
                                                 ctx.descriptors(b["desc"]), 0.5, 1.0, 500)
    assert np.array_equal(T.view(np.uint32), T_ref.view(np.uint32))   # same best hypothesis, same floats
    # the generator state advanced identically: a second run continues the stream on both sides
    T_ref2, _, _ = po.sac_ia(a["kp"], a["desc"], b["kp"], b["desc"], 0.5, 1.0, 50)
    T2 = ctx.estimateTransformFromDescriptorsSets(ctx.cloud(a["kp"]), ctx.descriptors(a["desc"]), ctx.cloud(b["kp"]),
                                                  ctx.descriptors(b["desc"]), 0.5, 1.0, 50)
    assert np.array_equal(T2.view(np.uint32), T_ref2.view(np.uint32))


def test_sac_ia_more_hypotheses_than_a_grid_dimension(ctx, po, scene):
    """max_iterations is any positive int in the reference (matching.cpp:142-194 hands it to setMaximumIterations);
    70 000 hypotheses do not fit gridDim.y of one launch, so the error kernel runs in slices.  A subset of the
    keypoints keeps the oracle's 70 000 x K nearest-keypoint lookups to seconds."""
    a, b = scene
    ka, da, kb, db = a["kp"][:160], a["desc"][:160], b["kp"][:200], b["desc"][:200]
    po.srand(3)
    T_ref, _, _ = po.sac_ia(ka, da, kb, db, 0.5, 1.0, 70000)
    ctx.srand(3)
    T = ctx.estimateTransformFromDescriptorsSets(ctx.cloud(ka), ctx.descriptors(da), ctx.cloud(kb), ctx.descriptors(db), 0.5, 1.0, 70000)
    assert np.array_equal(T.view(np.uint32), T_ref.view(np.uint32))


def test_sac_ia_certified_pick_in_all_its_cases(ctx, po, mm, scene):
    """SAC-IA's winner is certified from the hypotheses' error sums in double (csrc/registration.hip::k_sacia_select); the CPU
    path's float chain runs only for what the intervals leave open.  The cases: one candidate (the usual one: no chain at all);
    nothing in range (every term of every hypothesis is 1.0f, every sum is n exactly: the FIRST hypothesis, no chain);
    few keypoints, so that many hypotheses draw the same triples and tie exactly (the chains decide, the first minimum wins);
    and a range so small that most hypotheses are all-ones and a few are not.  Same transform bits as the oracle in each."""
    a, b = scene

    def both(ka, da, kb, db, msd, corr, H, seed):
        po.srand(seed)
        T_ref, best_it, _ = po.sac_ia(ka, da, kb, db, msd, corr, H)
        ctx.srand(seed)
        mm.sacia_stats(reset=True, collect=1)
        T = ctx.estimateTransformFromDescriptorsSets(ctx.cloud(ka), ctx.descriptors(da), ctx.cloud(kb), ctx.descriptors(db), msd, corr, H)
        st = mm.sacia_stats(collect=0)
        assert np.array_equal(T.view(np.uint32), T_ref.view(np.uint32)), (st, best_it)
        return st, best_it

    st, _ = both(a["kp"], a["desc"], b["kp"], b["desc"], 0.5, 1.0, 500, 1)
    print("usual:", st)
    assert st[0] == 1 and st[1] == 1 and st[2] == 1 and st[3] == 0
    # (a hypothesis carries its three samples onto their partners, so it is a range below the rounding of that fit -- d2 <=
    # 1e-14 -- that makes every term 1.0f)
    st, it = both(a["kp"], a["desc"], b["kp"], b["desc"], 0.5, 1e-14, 300, 2)
    print("nothing in range:", st, it)
    assert st[1] == 1 and st[2] == 300 and st[3] == 0 and it == 0
    ka, da, kb, db = a["kp"][:5], a["desc"][:5], b["kp"][:4], b["desc"][:4]
    st, _ = both(ka, da, kb, db, 0.05, 1.0, 400, 3)
    print("ties:", st)
    assert st[2] > 1
    st, _ = both(a["kp"][:300], a["desc"][:300], b["kp"][:300], b["desc"][:300], 0.5, 0.01, 500, 4)
    print("small range:", st)
    # three source keypoints and one target: every hypothesis carries the same triple, in one of six orders, onto the same point
    # -- sums that differ by an ulp or not at all: far more candidates than the chain kernel has blocks per pair (it works them
    # off in turns)
    st, _ = both(a["kp"][:3], a["desc"][:3], b["kp"][:1], b["desc"][:1], 0.01, 1.0, 300, 5)
    print("one triple:", st)
    assert st[2] > 8


def _small_rot(ax, ay, az, t):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T


def test_transform_score(ctx, po, scene, synth):
    a, b = scene
    gt = synth.relative_gt(a["T"], b["T"]).astype(np.float32)
    ca, cb = ctx.cloud(a["filt"]), ctx.cloud(b["filt"])
    for T in (gt, np.eye(4, dtype=np.float32), (gt @ _small_rot(0.02, -0.01, 0.05, [0.2, 0.1, 0.0])).astype(np.float32)):
        ref = po.transform_score(a["filt"], b["filt"], T, 1.0)
        got = ctx.transformScore(ca, cb, T, 1.0)
        assert got == pytest.approx(ref, rel=1e-6), (got, ref)   # double sum of identical float d2
    # nothing in range -> DBL_MAX; zero transform is scored like any other matrix
    far = np.eye(4, dtype=np.float32); far[0, 3] = 1e4
    assert ctx.transformScore(ca, cb, far, 1.0) == po.transform_score(a["filt"], b["filt"], far, 1.0) == np.finfo(np.float64).max
    z = np.zeros((4, 4), dtype=np.float32)
    assert ctx.transformScore(ca, cb, z, 1.0) == pytest.approx(po.transform_score(a["filt"], b["filt"], z, 1.0), rel=1e-6)


def test_nearest_neighbour_search_over_ranges(ctx, po, scene, synth):
    """The ICP / score search's cells are a quarter of its range, so the range sets how many points a cell holds and how many
    passes a lane needs (round 5: per-lane boxes, skipped shells, dropped corners, lanes that provably have nothing in range):
    ranges from a twentieth of the point spacing's scale to the whole scene, poses from aligned to far off."""
    a, b = scene
    gt = synth.relative_gt(a["T"], b["T"]).astype(np.float32)
    ca, cb = ctx.cloud(a["filt"]), ctx.cloud(b["filt"])
    poses = (gt, (gt @ _small_rot(0.05, -0.03, 0.2, [0.6, -0.4, 0.2])).astype(np.float32),
             (gt @ _small_rot(0.0, 0.0, 0.0, [3.0, 2.0, 0.5])).astype(np.float32))
    for max_distance in (0.0025, 0.04, 0.25, 4.0, 100.0):          # compared with the SQUARED distance: ranges 0.05 m .. 10 m
        for T in poses:
            ref = po.transform_score(a["filt"], b["filt"], T, max_distance)
            got = ctx.transformScore(ca, cb, T, max_distance)
            if ref == np.finfo(np.float64).max:
                assert got == ref, (max_distance, got)
            else:
                assert got == pytest.approx(ref, rel=1e-6), (max_distance, got, ref)
    for max_corr in (0.1, 2.0):
        guess = (gt @ _small_rot(0.02, -0.01, 0.04, [0.2, -0.1, 0.05])).astype(np.float32)
        T_ref, it_ref = po.icp(a["filt"], b["filt"], guess, max_corr, 0.5, 30, 1e-6)
        T = ctx.estimateTransformICP(ca, cb, guess, max_corr, 0.5, 30, 1e-6)
        assert np.linalg.norm(T - T_ref) <= 1e-3, (max_corr, np.linalg.norm(T - T_ref))
        assert ctx.last_icp_iterations == it_ref, (max_corr, ctx.last_icp_iterations, it_ref)


def test_icp(ctx, po, scene, synth):
    a, b = scene
    gt = synth.relative_gt(a["T"], b["T"])
    ca, cb = ctx.cloud(a["filt"]), ctx.cloud(b["filt"])
    for k, pert in enumerate([_small_rot(0.03, -0.02, 0.06, [0.3, -0.2, 0.1]), _small_rot(-0.01, 0.02, -0.1, [-0.4, 0.3, -0.05]),
                              np.eye(4)]):
        guess = (gt @ pert).astype(np.float32)
        for eps, iters in ((1e-2, 500), (1e-6, 30)):
            T_ref, it_ref = po.icp(a["filt"], b["filt"], guess, 1.0, 0.5, iters, eps)
            T = ctx.estimateTransformICP(ca, cb, guess, 1.0, 0.5, iters, eps)
            # tolerance: Frobenius 1e-3 (rotation part 2e-4): the device reduces in double, the oracle in
            # float like Eigen, and applies the accumulated transform instead of re-transforming the cloud
            assert np.linalg.norm(T - T_ref) <= 1e-3, (k, eps, np.linalg.norm(T - T_ref))
            assert np.linalg.norm(T[:3, :3] - T_ref[:3, :3]) <= 2e-4
            assert ctx.last_icp_iterations == it_ref, (k, eps, ctx.last_icp_iterations, it_ref)
    # zero initial guess stays zero (no guard before ICP in the reference, matching.cpp:250)
    z = np.zeros((4, 4), dtype=np.float32)
    assert not ctx.estimateTransformICP(ca, cb, z, 1.0, 0.5, 20, 1e-2).any()
    assert not po.icp(a["filt"], b["filt"], z, 1.0, 0.5, 20, 1e-2)[0].any()


def test_estimate_maps_transforms_end_to_end(ctx, po, mm, scene):
    a, b = scene
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    op = po.params_default(); op.descriptor_type = 2; op.estimation_method = 1
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert len(T) == len(ref_T) == 2 and len(pairs) == len(ref_pairs) == 1
    pt = pairs[0]["transform"].reshape(4, 4).T
    rt = ref_pairs[0]["transform"].reshape(4, 4).T
    # every stage up to SAC-IA's winning hypothesis is bit-equal to the CPU path; ICP reduces in double on the
    # device and in float on the CPU: Frobenius 1e-3 on the pair transform, 1e-3 relative on the confidence,
    # the integer observables (ICP iterations and correspondences) exact
    tr = po.last_run_traces()[0]
    assert np.linalg.norm(pt - rt) <= 1e-3, np.linalg.norm(pt - rt)
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)
    assert pairs[0]["icp_iterations"] == tr["icp_iterations"] and pairs[0]["icp_correspondences"] == tr["icp_correspondences"]
    for g, r in zip(T, ref_T):
        assert np.linalg.norm(g - r) <= 2e-3
    for m, ref in ((a, a["kp"]), (b, b["kp"])):
        f = ctx.mapFeatures(ctx.cloud(m["raw"]), params)
        assert np.array_equal(xyz(f.keypoints.numpy()).view(np.uint32), xyz(ref).view(np.uint32))
        assert np.array_equal(f.descriptors.numpy().view(np.uint32), m["desc"].view(np.uint32))
        f.free()
    # MATCHING + RANSAC path: cross-match and inlier counts exact (R/src/registration_visualisation.cpp:129-130)
    params.estimation_method = 0; op.estimation_method = 0
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    tr = po.last_run_traces()[0]
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert pairs[0]["n_correspondences"] == tr["n_correspondences"] > 0 and pairs[0]["n_inliers"] == tr["n_inliers"] > 0
    assert pairs[0]["icp_iterations"] == tr["icp_iterations"] and pairs[0]["icp_correspondences"] == tr["icp_correspondences"]
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= 1e-3
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)


def test_reference_gtests(ctx, mm):
    """R/test/test_map_merging.cpp:9-40 through the ABI."""
    P = mm.MapMergingParams()
    assert ctx.estimateMapsTransforms([], P) == []                                  # estimateMapsTransforms.empty
    r = ctx.estimateMapsTransforms([np.empty(0, dtype=mm.POINT)], P)                # estimateMapsTransforms.one
    assert len(r) == 1 and np.array_equal(r[0], np.eye(4, dtype=np.float32))
    assert ctx.composeMaps([], [], 0.0) is None                                     # composeMaps.empty
    with pytest.raises(Exception):                                                  # composeMaps.wrongSizes
        ctx.composeMaps([ctx.cloud(np.empty(0, dtype=mm.POINT))], [], 0.0)
    res = ctx.composeMaps([ctx.cloud(np.empty(0, dtype=mm.POINT))], [np.eye(4)], 0.0)   # composeMaps.one
    assert res is not None and len(res) == 0


def test_compose_maps(ctx, po, scene):
    a, b = scene
    T = [np.eye(4, dtype=np.float32), _small_rot(0.0, 0.0, 0.3, [1, 2, 0]).astype(np.float32)]
    ref = po.compose_maps([a["filt"], b["filt"]], T, 0.05)
    got = ctx.composeMaps([ctx.cloud(a["filt"]), ctx.cloud(b["filt"])], T, 0.05).numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # zero transforms are skipped
    T[1] = np.zeros((4, 4), dtype=np.float32)
    ref = po.compose_maps([a["filt"], b["filt"]], T, 0.05)
    got = ctx.composeMaps([ctx.cloud(a["filt"]), ctx.cloud(b["filt"])], T, 0.05).numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_pairs_on_several_contexts_are_bit_identical(ctx, mm, scene):
    """bench.py runs several contexts (stream + host thread) per GPU over shared, prepared maps:
    every pair must come out with the same bits as on one context, whatever the interleaving."""
    from concurrent.futures import ThreadPoolExecutor
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    raws = [m["raw"] for m in scene] + [scene[0]["raw"][::2].copy()]
    order = [(0, 1), (0, 2), (1, 2), (1, 0), (2, 0), (2, 1)]

    def features(c, i):
        m = c.mapFeatures(c.cloud(raws[i]), params)
        c.mapPrepare(m, params)
        return m

    # one context, sequential
    ctx.srand(1)
    maps = [features(ctx, i) for i in range(3)]
    ref = [ctx.pairEstimate(maps[i], maps[j], params).copy() for i, j in order]
    # three contexts: features in parallel, then every context replays all pairs and executes its own
    S = 3
    ctxs = [mm.Context(0) for _ in range(S)]
    try:
        with ThreadPoolExecutor(S) as tp:
            maps2 = list(tp.map(lambda s: features(ctxs[s], s), range(S)))
            got = [None] * len(order)

            def work(s):
                c = ctxs[s]
                c.srand(1)
                for p, (i, j) in enumerate(order):
                    r = c.pairEstimate(maps2[i], maps2[j], params, execute=(p % S == s))
                    if p % S == s:
                        got[p] = r.copy()
                c.synchronize()

            for _ in range(3):      # a few rounds: different interleavings
                list(tp.map(work, range(S)))
                for r, g in zip(ref, got):
                    assert np.array_equal(r["transform"].view(np.uint32), g["transform"].view(np.uint32))
                    assert r["confidence"] == g["confidence"] and r["icp_iterations"] == g["icp_iterations"]
        for m in maps2:
            m.free()
    finally:
        for c in ctxs:
            c.close()
    for m in maps:
        m.free()


def test_pfh(ctx, po, scene):
    """The reference's default descriptor (PFHSignature125): every pair of a keypoint's neighbours."""
    for m in scene:
        kp_ref, ref = po.descriptors_pfh(m["filt"], m["nrm"], m["kp_raw"], R_DESC)
        kp = ctx.cloud(m["kp_raw"])
        desc = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), kp, 0, R_DESC)
        got = desc.numpy()
        assert got.shape == ref.shape and got.shape[1] == 125
        assert np.array_equal(kp.numpy().view(np.uint32), kp_ref.view(np.uint32))       # same pruning
        assert np.allclose(got.sum(axis=1), 100.0, atol=2e-2)
        # bins are counted in integers (glibc's atan2f restated) and the float chain is replayed: the oracle's bits
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # isolated keypoints are pruned like the reference does
    kp = scene[0]["kp_raw"][:40].copy()
    kp["x"][3] += 400.0
    kp_ref, ref = po.descriptors_pfh(scene[0]["filt"], scene[0]["nrm"], kp, R_DESC)
    k = ctx.cloud(kp)
    desc = ctx.computeLocalDescriptors(ctx.cloud(scene[0]["filt"]), ctx.normals(scene[0]["nrm"]), k, 0, R_DESC)
    assert len(desc) == len(ref) == 39 and np.array_equal(k.numpy().view(np.uint32), kp_ref.view(np.uint32))


def test_desc_knn_dim125_exact(ctx, po, mm, scene):
    """k-NN over PFH rows (125 wide, 64 MFMA steps): exact FLANN-order results, few fallbacks."""
    rng = np.random.default_rng(11)
    _, base = po.descriptors_pfh(scene[0]["filt"], scene[0]["nrm"], scene[0]["kp_raw"], R_DESC)
    A = base[rng.integers(0, len(base), 900)] + rng.normal(0, 0.2, (900, 125)).astype(np.float32)
    B = base[rng.integers(0, len(base), 1300)] + rng.normal(0, 0.2, (1300, 125)).astype(np.float32)
    B[50:70] = B[150:170]
    A[:10] = B[300:310]
    A, B = A.astype(np.float32), B.astype(np.float32)
    L = mm.lib()
    L.mm3d_debug_knn_fallback_rows.restype = L.mm3d_debug_knn_rows.restype = __import__("ctypes").c_longlong
    L.mm3d_set_debug(ctx._h, 1)
    r0, f0 = L.mm3d_debug_knn_rows(ctx._h), L.mm3d_debug_knn_fallback_rows(ctx._h)
    da, db = ctx.descriptors(A, 0), ctx.descriptors(B, 0)
    for k in (1, 5, 10):
        got = ctx.findFeatureCorrespondences(da, db, k)
        ref = po.find_correspondences(A, B, k)
        assert np.array_equal(got["index_query"], ref["index_query"]), k
        assert np.array_equal(got["index_match"], ref["index_match"]), k
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32)), k
    rows, fb = L.mm3d_debug_knn_rows(ctx._h) - r0, L.mm3d_debug_knn_fallback_rows(ctx._h) - f0
    L.mm3d_set_debug(ctx._h, 0)
    assert rows > 0 and fb <= 0.05 * rows, (rows, fb)


def test_default_configuration_end_to_end(ctx, po, mm, scene):
    """The reference's DEFAULT parameters: PFH descriptors + reciprocal matching + RANSAC + ICP."""
    a, b = scene
    params = mm.MapMergingParams()                      # descriptor PFH, estimation MATCHING
    op = po.params_default()
    assert params.descriptor_type == 0 and params.estimation_method == 0 and op.descriptor_type == 0
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert len(T) == len(ref_T) == 2 and len(pairs) == len(ref_pairs) == 1
    # every stage before ICP is bit-equal to the CPU path: counts exact, Frobenius 1e-3, confidence 1e-3
    tr = po.last_run_traces()[0]
    assert pairs[0]["n_correspondences"] == tr["n_correspondences"] > 0 and pairs[0]["n_inliers"] == tr["n_inliers"] > 0
    assert pairs[0]["icp_iterations"] == tr["icp_iterations"] and pairs[0]["icp_correspondences"] == tr["icp_correspondences"]
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= 1e-3
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)


def test_bench_ranks_and_streams_give_the_same_bits():
    """bench.py end to end on a small job: 1 rank x 1 stream, 1 rank x 4 streams and 2 ranks x 2 streams
    (the ranks share this box's single GPU through gloo: MM3D_BENCH_BACKEND / MM3D_BENCH_DEVICE) must
    print the same pair-transform CRC -- the exchange, rebuild, replay and gather paths are exact."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--maps", "4", "--points", "40000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-cache"]

    def run(cmd, env=None):
        e = dict(os.environ)
        e.update(env or {})
        out = subprocess.run(cmd, cwd=root, env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    a = run([sys.executable, "bench.py", "--streams", "1"] + common)
    b = run([sys.executable, "bench.py", "--streams", "4"] + common)
    c = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29533", "bench.py", "--gpus", "2", "--streams", "2"] + common,
            env={"MM3D_BENCH_BACKEND": "gloo", "MM3D_BENCH_DEVICE": "0"})
    # the N > 1 driver (mm3d_shard_*) on one rank, and on three ranks (uneven ownership: maps 0..3 -> ranks 0 1 2 2)
    d = run([sys.executable, "bench.py", "--streams", "4", "--engine", "shard"] + common)
    e = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
             "--master-port", "29534", "bench.py", "--gpus", "3", "--streams", "2"] + common,
            env={"MM3D_BENCH_BACKEND": "gloo", "MM3D_BENCH_DEVICE": "0"})
    crcs = [x["pair_transforms_crc32"] for x in (a, b, c, d, e)]
    assert len(set(crcs)) == 1, crcs
    assert a["maps_estimated"] == b["maps_estimated"] == c["maps_estimated"] == d["maps_estimated"] == e["maps_estimated"]
    assert c["n_gpus"] == 2 and e["n_gpus"] == 3


def test_round4_shortcuts_do_not_change_a_bit():
    """The round-4 shortcuts of the whole-map path -- SIFT's first octave on the input cloud itself when its voxelisation is
    the identity, the normals riding on that octave's sorted lists, the grid shared with the descriptors, the Hilbert blocks
    scaled with the octave's leaf, the sixteen-wave configuration -- each switched off (or on) by its environment knob in a
    fresh process: every combination prints the same pair-transform CRC as the default."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, "bench.py", "--maps", "3", "--points", "120000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pcie",
              "--streams", "3"]

    def run(env):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run(common, cwd=root, env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    base = run({})
    variants = [{"MM3D_SIFT_NO_IDENTITY": "1"}, {"MM3D_SIFT_NO_FUSED_NORMALS": "1"}, {"MM3D_SIFT_NO_SHARED_GRID": "1"},
                {"MM3D_SIFT_HIL_FACTOR": "0"}, {"MM3D_SIFT_LARGE16": "1"},
                {"MM3D_SIFT_NO_IDENTITY": "1", "MM3D_SIFT_HIL_FACTOR": "0", "MM3D_SIFT_NO_SHARED_GRID": "1"}]
    for v in variants:
        r = run(v)
        assert r["pair_transforms_crc32"] == base["pair_transforms_crc32"] and r["maps_estimated"] == base["maps_estimated"], (v, r["pair_transforms_crc32"])
    # and the profile shows what the default does: no launch of the normals' own in the step (the figure is then the stand-alone
    # launch's, from the untimed isolated pass: mpoints_per_s.normals_alone)
    assert base["mpoints_per_s"]["normals_fused"] and base["mpoints_per_s"]["normals"] == base["mpoints_per_s"]["normals_alone"]["mpoints_per_s"]
    unfused = run({"MM3D_SIFT_NO_FUSED_NORMALS": "1"})["mpoints_per_s"]
    assert unfused["normals"] is not None and unfused["normals_fused"] is None


def test_pfh_neighbourhoods_beyond_lds(ctx, po, scene):
    """More than 1024 neighbours per keypoint: the neighbour list no longer fits the block's LDS and the
    keypoint goes through the global-scratch pass of the same kernel."""
    m = scene[0]
    kp = m["kp_raw"][:6].copy()
    radius = 2.2                                          # ~1500-3000 neighbours on this scene
    kp_ref, ref = po.descriptors_pfh(m["filt"], m["nrm"], kp, radius)
    k = ctx.cloud(kp)
    got = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), k, 0, radius).numpy()
    assert got.shape == ref.shape == (6, 125)
    assert np.allclose(got.sum(axis=1), 100.0, atol=5e-2)
    assert np.abs(got - ref).max() <= 2e-2, np.abs(got - ref).max()


def test_shot(ctx, po, scene):
    """SHOT1344 (SHOTColorEstimation, dispatch_descriptors.h:46): frames and rows against the oracle.
    The kernel keeps the oracle's (distance, index) neighbour order in every float sum, so rows are
    bit-equal unless a double acos / atan2 differs in its last bit between the two libms AND that moves
    a float rounding: the test asks for >= 99 % bit-equal rows and 2e-6 on the rest."""
    for m in scene:
        kp_ref, ref = po.descriptors_shot(m["filt"], m["nrm"], m["kp_raw"], R_DESC)
        raw_ref, rf_ref = po.shot_raw(m["filt"], m["nrm"], m["kp_raw"], R_DESC)
        kp = ctx.cloud(m["kp_raw"])
        desc = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), kp, 4, R_DESC)
        got = desc.numpy()
        assert got.shape == ref.shape and got.shape[1] == 1344 and len(ref) > 100
        assert np.array_equal(kp.numpy().view(np.uint32), kp_ref.view(np.uint32))       # same pruning
        rf = desc.frames()
        keep = np.isfinite(raw_ref).all(1)
        assert np.array_equal(rf.view(np.uint32), rf_ref[keep].view(np.uint32))         # frames: bit-exact
        assert np.allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-5)
        same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
        assert same.mean() >= 0.99, same.mean()
        assert np.abs(got - ref).max() <= 2e-6, np.abs(got - ref).max()
    # isolated keypoints are pruned like the reference does
    kp = scene[0]["kp_raw"][:40].copy()
    kp["x"][3] += 400.0
    kp_ref, ref = po.descriptors_shot(scene[0]["filt"], scene[0]["nrm"], kp, R_DESC)
    k = ctx.cloud(kp)
    desc = ctx.computeLocalDescriptors(ctx.cloud(scene[0]["filt"]), ctx.normals(scene[0]["nrm"]), k, 4, R_DESC)
    assert len(desc) == len(ref) == 39 and np.array_equal(k.numpy().view(np.uint32), kp_ref.view(np.uint32))
    assert np.array_equal(desc.frames().shape, (39, 9))


def test_shot_neighbourhoods_beyond_lds(ctx, po, scene):
    """More than 1024 neighbours: the sort keys move to global scratch, same kernel, same bits."""
    m = scene[0]
    kp = m["kp_raw"][:6].copy()
    radius = 2.2
    kp_ref, ref = po.descriptors_shot(m["filt"], m["nrm"], kp, radius)
    k = ctx.cloud(kp)
    got = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), k, 4, radius).numpy()
    assert got.shape == ref.shape == (6, 1344)
    assert np.abs(got - ref).max() <= 2e-6, np.abs(got - ref).max()


def test_desc_knn_dim1344_exact(ctx, po, mm, scene):
    """k-NN over SHOT rows (1344 wide: the contraction dimension is streamed, 676 MFMA steps): exact
    FLANN-order results incl. duplicates, few fallbacks; and the small-problem brute-force path."""
    rng = np.random.default_rng(12)
    _, base = po.descriptors_shot(scene[0]["filt"], scene[0]["nrm"], scene[0]["kp_raw"], R_DESC)

    def rows(n, noise):
        X = base[rng.integers(0, len(base), n)] + np.abs(rng.normal(0, noise, (n, 1344))).astype(np.float32)
        return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)

    A, B = rows(700, 0.004), rows(1500, 0.004)
    B[50:70] = B[150:170]
    A[:10] = B[300:310]
    L = mm.lib()
    L.mm3d_debug_knn_fallback_rows.restype = L.mm3d_debug_knn_rows.restype = __import__("ctypes").c_longlong
    L.mm3d_set_debug(ctx._h, 1)
    r0, f0 = L.mm3d_debug_knn_rows(ctx._h), L.mm3d_debug_knn_fallback_rows(ctx._h)
    da, db = ctx.descriptors(A, 4), ctx.descriptors(B, 4)
    for k in (1, 5, 10):
        got = ctx.findFeatureCorrespondences(da, db, k)
        ref = po.find_correspondences(A, B, k)
        assert np.array_equal(got["index_query"], ref["index_query"]), k
        assert np.array_equal(got["index_match"], ref["index_match"]), k
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32)), k
    rows_, fb = L.mm3d_debug_knn_rows(ctx._h) - r0, L.mm3d_debug_knn_fallback_rows(ctx._h) - f0
    L.mm3d_set_debug(ctx._h, 0)
    assert rows_ > 0 and fb <= 0.05 * rows_, (rows_, fb)
    # small sets take the brute-force kernels for every row
    sa, sb = ctx.descriptors(A[:37], 4), ctx.descriptors(B[:90], 4)
    got = ctx.findFeatureCorrespondences(sa, sb, 5)
    ref = po.find_correspondences(A[:37], B[:90], 5)
    assert np.array_equal(got["index_query"], ref["index_query"]) and np.array_equal(got["index_match"], ref["index_match"])
    assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32))


def test_shot_configuration_end_to_end(ctx, po, mm, scene, synth):
    """descriptor_type = SHOT through estimateMapsTransforms, both estimation methods (north_star:
    "computeLocalDescriptors FPFH/SHOT").  The device rows are 1 ulp away from the oracle's on ~1 % of
    the keypoints, so as on the MATCHING path above the comparison is "same basin" (Frobenius 0.15,
    confidence within 20 %), plus the recovered relative pose against the ground truth."""
    a, b = scene
    for method in (1, 0):
        params = mm.MapMergingParams(descriptor_type=4, estimation_method=method)
        op = po.params_default(); op.descriptor_type = 4; op.estimation_method = method
        po.srand(1); ctx.srand(1)
        ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
        T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
        assert len(T) == len(ref_T) == 2 and len(pairs) == len(ref_pairs) == 1
        assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= 1e-3, method
        assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)
        assert pairs[0]["n_correspondences"] == len(po.last_pair_trace()["correspondences"]) if method == 0 and "correspondences" in (po.last_pair_trace() or {}) else True
        # and the right basin: the generator's ground truth (ICP stops at transform_epsilon = 1e-2 on
        # this sparse 12 k-point scene, a few decimetres short -- the CPU path stops at the same place)
        gt = synth.relative_gt(a["T"], b["T"])
        est = pairs[0]["transform"].reshape(4, 4).T
        ref = ref_pairs[0]["transform"].reshape(4, 4).T
        assert np.linalg.norm(est - gt) <= 0.75, (method, np.linalg.norm(est - gt))
        assert abs(np.linalg.norm(est - gt) - np.linalg.norm(ref - gt)) <= 0.15


def test_pfhrgb(ctx, po, mm, scene):
    """PFHRGBSignature250 (dispatch_descriptors.h:39): rows against the oracle, k-NN over 250-wide rows, and the
    descriptor through estimateMapsTransforms."""
    for m in scene:
        kp_ref, ref = po.descriptors_pfhrgb(m["filt"], m["nrm"], m["kp_raw"], R_DESC)
        kp = ctx.cloud(m["kp_raw"])
        got = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), kp, 1, R_DESC).numpy()
        assert got.shape == ref.shape and got.shape[1] == 250
        assert np.array_equal(kp.numpy().view(np.uint32), kp_ref.view(np.uint32))
        assert np.allclose(got[:, :125].sum(axis=1), 200.0, atol=5e-2) and np.allclose(got[:, 125:].sum(axis=1), 200.0, atol=5e-2)
        # colour bins are integer arithmetic, geometry bins use glibc's atan2f restated: the oracle's bits
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # k-NN on these rows (256-wide padded contraction on the streaming kernels)
    rng = np.random.default_rng(13)
    _, base = po.descriptors_pfhrgb(scene[0]["filt"], scene[0]["nrm"], scene[0]["kp_raw"], R_DESC)
    A = (base[rng.integers(0, len(base), 800)] + rng.normal(0, 0.3, (800, 250))).astype(np.float32)
    B = (base[rng.integers(0, len(base), 1200)] + rng.normal(0, 0.3, (1200, 250))).astype(np.float32)
    B[50:70] = B[150:170]
    A[:10] = B[300:310]
    da, db = ctx.descriptors(A, 1), ctx.descriptors(B, 1)
    for k in (1, 5):
        got = ctx.findFeatureCorrespondences(da, db, k)
        ref = po.find_correspondences(A, B, k)
        assert np.array_equal(got["index_query"], ref["index_query"]) and np.array_equal(got["index_match"], ref["index_match"])
        assert np.array_equal(got["distance"].view(np.uint32), ref["distance"].view(np.uint32))
    # end to end: bit-equal up to ICP, then Frobenius 1e-3 / confidence 1e-3
    a, b = scene
    params = mm.MapMergingParams(descriptor_type=1, estimation_method=1)
    op = po.params_default(); op.descriptor_type = 1; op.estimation_method = 1
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert len(pairs) == len(ref_pairs) == 1
    # (the two maps lie 25 m apart: the CPU path's ICP sums are sequential float sums over coordinates of that
    # size, the device's run in double -- 1e-3 per 10 m of translation)
    tol = 1e-3 * max(1.0, float(np.linalg.norm(ref_pairs[0]["transform"][12:15])) / 10.0)
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= tol
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)


def test_harris_keypoints(ctx, po, mm, scene):
    """detectKeypoints(HARRIS) (features.cpp:64-83): the responses (their sums run in radiusSearch's order, like the
    oracle's), the set of corners and the refined positions are the oracle's bit for bit."""
    for m in scene:
        pts, nrm = ctx.cloud(m["filt"]), ctx.normals(m["nrm"])
        kp_ref, idx_ref, resp_ref = po.keypoints_harris(m["filt"], m["nrm"], 0.002, R_NRM)
        resp = ctx.harrisResponse(pts, nrm, R_NRM)
        assert np.array_equal(resp.view(np.uint32), resp_ref.view(np.uint32)), np.abs(resp - resp_ref).max()
        kp = ctx.detectKeypoints(pts, nrm, 1, 0.002, R_NRM, RES).numpy()
        assert len(kp_ref) >= 10
        assert np.array_equal(xyz(kp).view(np.uint32), xyz(kp_ref).view(np.uint32))
        assert (kp["rgba"] == 0).all()
    # through the pipeline: keypoint_type = HARRIS with FPFH + SAC-IA on the two maps
    a, b = scene
    params = mm.MapMergingParams(keypoint_type=1, keypoint_threshold=0.0005, descriptor_type=2, estimation_method=1)
    op = po.params_default(); op.keypoint_type = 1; op.keypoint_threshold = 0.0005; op.descriptor_type = 2; op.estimation_method = 1
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert len(pairs) == len(ref_pairs) == 1
    # corners, descriptors and the SAC-IA winner are the oracle's bits; ICP and the score reduce in double on the device
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= 1e-3
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)
    f = ctx.mapFeatures(ctx.cloud(a["raw"]), params)
    nk = len(f.keypoints)
    f.free()
    assert 5 <= nk <= 200


def test_golden_features_fixture_on_device(ctx, po, mm, synth):
    """The device against the committed vectors of tests/golden/features_6k.npz (made by make_golden.py from
    the oracle): Harris corners, PFH / PFHRGB / SHOT rows, frames."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features_6k.npz"))
    world, maps = synth.synth_maps(2, int(g["n_raw"]), overlap_step=float(g["overlap_step"]))
    raw = synth.pack_points(maps[0][0], maps[0][1])
    filt = ctx.removeOutliers(ctx.downSample(ctx.cloud(raw), 0.1), 0.8, 50)
    nrm_host = po.normals(filt.numpy(), 0.6)            # descriptors are compared on identical normals
    nrm = ctx.normals(nrm_host)
    kp48 = po.keypoints_sift(filt.numpy(), 0.1, 3, 3, 5.0)[0][:48].copy()
    hk = ctx.detectKeypoints(filt, nrm, 1, 0.002, 0.6, 0.1).numpy()
    a = {tuple(r) for r in xyz(hk).view(np.uint32).tolist()}
    b = {tuple(r) for r in g["harris_kp"].view(np.uint32).tolist()}
    assert len(a ^ b) <= 1
    assert np.abs(ctx.harrisResponse(filt, nrm, 0.6) - g["harris_response"]).max() <= 2e-6
    for name, dt, tol in (("pfh", 0, 0.5), ("pfhrgb", 1, 0.5), ("shot", 4, 2e-6)):
        k = ctx.cloud(kp48)
        d = ctx.computeLocalDescriptors(filt, nrm, k, dt, 0.8)
        assert np.array_equal(xyz(k.numpy()).view(np.uint32), g[name + "_kp"].view(np.uint32)), name
        assert np.abs(d.numpy() - g[name]).max() <= tol, (name, np.abs(d.numpy() - g[name]).max())
        if name == "shot":
            assert np.array_equal(d.frames().view(np.uint32), g["shot_rf"].view(np.uint32))
            assert (d.numpy().view(np.uint32) == g[name].view(np.uint32)).all(axis=1).mean() >= 0.95


def test_rsd(ctx, po, mm, scene):
    """PrincipalRadiiRSD (dispatch_descriptors.h:43): rows against the oracle (minima / maxima of double angles:
    bit-equal unless the two libms' acos differ in the last bit and that moves a float rounding), 2-wide k-NN
    with its heavy ties, and the descriptor through estimateMapsTransforms."""
    for m in scene:
        kp_ref, ref = po.descriptors_rsd(m["filt"], m["nrm"], m["kp_raw"], R_DESC)
        kp = ctx.cloud(m["kp_raw"])
        d = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), kp, 3, R_DESC)
        got = d.numpy()
        assert got.shape == ref.shape and got.shape[1] == 2 and len(kp) == len(kp_ref)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(axis=1).mean() >= 0.98
        assert np.abs(got - ref).max() <= 1e-6
        # k-NN on identical inputs: exact, ties to the lower index
        for k in (1, 5):
            a = ctx.findFeatureCorrespondences(d, d, k)
            b = po.find_correspondences(got, got, k)
            assert np.array_equal(a["index_match"], b["index_match"]) and np.array_equal(a["distance"].view(np.uint32), b["distance"].view(np.uint32))
    a, b = scene
    params = mm.MapMergingParams(descriptor_type=3, estimation_method=1)
    op = po.params_default(); op.descriptor_type = 3; op.estimation_method = 1
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert len(pairs) == len(ref_pairs) == 1
    # (the two maps lie 25 m apart: the CPU path's ICP sums are sequential float sums over coordinates of that
    # size, the device's run in double -- 1e-3 per 10 m of translation)
    tol = 1e-3 * max(1.0, float(np.linalg.norm(ref_pairs[0]["transform"][12:15])) / 10.0)
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= tol
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)


def test_sc3d(ctx, po, mm, scene):
    """ShapeContext1980 (dispatch_descriptors.h:47): the random frame directions replay boost::mt19937(12345) in
    keypoint order, bins take their votes in the oracle's neighbour order: rows are bit-equal unless an atan2f /
    acosf ulp moves a neighbour across a bin edge (one vote of that row moves)."""
    for m in scene:
        kp_ref, ref = po.descriptors_sc3d(m["filt"], m["nrm"], m["kp_raw"], R_DESC)
        kp = ctx.cloud(m["kp_raw"])
        got = ctx.computeLocalDescriptors(ctx.cloud(m["filt"]), ctx.normals(m["nrm"]), kp, 5, R_DESC).numpy()
        assert got.shape == ref.shape and got.shape[1] == 1980 and len(ref) > 100
        assert np.array_equal(kp.numpy().view(np.uint32), kp_ref.view(np.uint32))
        same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
        assert same.mean() >= 0.9, same.mean()
        # a moved vote changes two bins of the row by that vote; the row's mass moves by at most the volume-factor ratio
        assert np.abs(got.sum(1) - ref.sum(1)).max() <= 0.05 * ref.sum(1).max()
        assert (np.abs(got - ref) > 1e-5).sum(axis=1).max() <= 8
    # pruning: a keypoint without neighbours takes no random draws and goes
    kp = scene[0]["kp_raw"][:40].copy()
    kp["x"][3] += 400.0
    kp_ref, ref = po.descriptors_sc3d(scene[0]["filt"], scene[0]["nrm"], kp, R_DESC)
    k = ctx.cloud(kp)
    d = ctx.computeLocalDescriptors(ctx.cloud(scene[0]["filt"]), ctx.normals(scene[0]["nrm"]), k, 5, R_DESC)
    assert len(d) == len(ref) == 39 and np.array_equal(k.numpy().view(np.uint32), kp_ref.view(np.uint32))
    assert (d.numpy().view(np.uint32) == ref.view(np.uint32)).all(axis=1).mean() >= 0.9
    # k-NN over 1980-wide rows and the whole pipeline
    got = d.numpy()
    a = ctx.findFeatureCorrespondences(d, d, 3)
    b = po.find_correspondences(got, got, 3)
    assert np.array_equal(a["index_match"], b["index_match"]) and np.array_equal(a["distance"].view(np.uint32), b["distance"].view(np.uint32))
    a, b = scene
    params = mm.MapMergingParams(descriptor_type=5, estimation_method=1)
    op = po.params_default(); op.descriptor_type = 5; op.estimation_method = 1
    po.srand(1); ctx.srand(1)
    ref_T, ref_pairs = po.estimate_maps_transforms([a["raw"], b["raw"]], op)
    T, pairs = ctx.estimateMapsTransforms([a["raw"], b["raw"]], params, return_pairs=True)
    assert len(pairs) == len(ref_pairs) == 1
    # (the two maps lie 25 m apart: the CPU path's ICP sums are sequential float sums over coordinates of that
    # size, the device's run in double -- 1e-3 per 10 m of translation)
    tol = 1e-3 * max(1.0, float(np.linalg.norm(ref_pairs[0]["transform"][12:15])) / 10.0)
    assert np.linalg.norm(pairs[0]["transform"] - ref_pairs[0]["transform"]) <= tol
    assert pairs[0]["confidence"] == pytest.approx(ref_pairs[0]["confidence"], rel=1e-3)



def test_desc_knn_dim352_shape_exact(ctx, po, mm):
    """pcl::SHOT352's width (BASELINE.json configs[3] names it; the reference binds SHOT1344, so the width has no descriptor
    type and goes through mm3d_debug_desc_knn): the wide path -- split-bf16 MFMA selector, exact re-rank, certificate -- against
    the oracle's FLANN-order k-NN, indices and distance bits, with duplicate rows and few fallbacks."""
    import ctypes as C
    rng = np.random.default_rng(21)
    centres = np.abs(rng.normal(0, 1, (40, 352))).astype(np.float32)

    def rows(n):
        X = centres[rng.integers(0, 40, n)] + np.abs(rng.normal(0, 0.05, (n, 352))).astype(np.float32)
        return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)

    A, B = rows(900), rows(2100)
    B[50:70] = B[150:170]
    A[:10] = B[300:310]
    L = mm.lib()
    L.mm3d_debug_knn_fallback_rows.restype = L.mm3d_debug_knn_rows.restype = C.c_longlong
    L.mm3d_set_debug(ctx._h, 1)
    r0, f0 = L.mm3d_debug_knn_rows(ctx._h), L.mm3d_debug_knn_fallback_rows(ctx._h)
    for k in (1, 5, 10):
        gi, gd = np.empty((len(A), k), dtype=np.int32), np.empty((len(A), k), dtype=np.float32)
        ctx._ck(L.mm3d_debug_desc_knn(ctx._h, A.ctypes.data_as(C.c_void_p), C.c_size_t(len(A)), B.ctypes.data_as(C.c_void_p), C.c_size_t(len(B)), 352, k,
                                      gi.ctypes.data_as(C.c_void_p), gd.ctypes.data_as(C.c_void_p)))
        ri, rd = po.desc_knn(A, B, k)
        assert np.array_equal(gi, ri), k
        assert np.array_equal(gd.view(np.uint32), rd.view(np.uint32)), k
    rows_, fb = L.mm3d_debug_knn_rows(ctx._h) - r0, L.mm3d_debug_knn_fallback_rows(ctx._h) - f0
    L.mm3d_set_debug(ctx._h, 0)
    assert rows_ > 0 and fb <= 0.05 * rows_, (rows_, fb)


def test_float_chain_replay_is_exact(ctx, mm):
    """PFH bins are counted in integers and the float chain "0 + incr + incr + ..." is replayed once per bin, binade by
    binade (device_util.hpp::float_chain_sum); the skip must give the bits of the plain loop."""
    import ctypes as C
    rng = np.random.default_rng(21)
    pairs = rng.integers(1, 60000, 1500)
    incr = (np.float32(100.0) / pairs.astype(np.float32)).astype(np.float32)
    hits = (rng.random(1500) * pairs).astype(np.uint32)
    extra_i = np.array([1.5, 0.75, 3.0, 0.1, 1.0, 2.5, 1e-3, 7.0, 0.3333333, 1e-8, 0.0, 5e-3], np.float32)
    extra_h = np.array([70000, 65536, 9, 100000, 1 << 20, 12345, 200000, 8, 7, 300000, 50, 16777216 // 64], np.uint32)
    incr, hits = np.concatenate([incr, extra_i]), np.concatenate([hits, extra_h])
    out = np.empty(len(incr), np.float32)
    ctx._ck(mm.lib().mm3d_debug_float_chain(ctx._h, incr.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), len(incr),
                                            out.ctypes.data_as(C.c_void_p)))
    ref = np.zeros(len(incr), np.float32)
    left = hits.astype(np.int64).copy()
    while (left > 0).any():                      # the plain loop, vectorised over the cases
        m = left > 0
        ref[m] = (ref[m] + incr[m]).astype(np.float32)
        left[m] -= 1
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
