"""Degenerate inputs through the whole path: nothing may crash or hang, and the outcome is the one the
reference's semantics give (no keypoints -> the pair is skipped, map_merging.cpp:246-254)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(mm, xyz, grey=128):
    a = np.zeros(len(xyz), dtype=mm.POINT)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    a["rgba"] = 0xFF000000 | (grey << 16) | (grey << 8) | grey
    return a


@pytest.fixture(scope="module")
def textured(mm):
    rng = np.random.default_rng(3)
    c = _cloud(mm, np.concatenate([rng.uniform(0, 8, (30000, 2)), 0.2 * rng.standard_normal((30000, 1))], 1).astype(np.float32))
    c["rgba"] = 0xFF000000 | (rng.integers(0, 255, 30000).astype(np.uint32) * 0x010101)
    return c


def test_degenerate_clouds(ctx, mm, textured):
    rng = np.random.default_rng(4)
    cases = {
        "empty": _cloud(mm, np.zeros((0, 3), np.float32)),
        "all_nan": _cloud(mm, np.full((100, 3), np.nan, np.float32)),
        "identical": _cloud(mm, np.tile(np.array([[1, 2, 3]], np.float32), (500, 1))),
        "tiny": _cloud(mm, rng.uniform(0, 1, (10, 3)).astype(np.float32)),
        "line": _cloud(mm, np.stack([np.linspace(0, 30, 3000), np.zeros(3000), np.zeros(3000)], 1).astype(np.float32)),
        "flat_untextured": _cloud(mm, np.concatenate([rng.uniform(0, 10, (20000, 2)), np.zeros((20000, 1))], 1).astype(np.float32)),
    }
    P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    for name, c in cases.items():
        m = ctx.mapFeatures(ctx.cloud(c), P)
        assert len(m.keypoints) == 0 and len(m.descriptors) == 0, name      # nothing SIFT could fire on
        m.free()
        T, pairs = ctx.estimateMapsTransforms([c, textured], P, return_pairs=True)
        assert len(pairs) == 0, name                                        # pair skipped: a map without keypoints
        assert len(T) == 2 and not np.any(T[0]) and not np.any(T[1]), name  # documented: zero matrices, not UB


def test_non_finite_points_are_ignored(ctx, po, mm, textured):
    dirty = textured.copy()
    dirty["x"][::7] = np.nan
    dirty["z"][::11] = np.inf
    clean = dirty[np.isfinite(dirty["x"]) & np.isfinite(dirty["z"])]
    P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    a = ctx.mapFeatures(ctx.cloud(dirty), P)
    b = ctx.mapFeatures(ctx.cloud(clean), P)
    # the voxel grid skips non-finite points, so both clouds give the same filtered cloud and keypoints
    assert np.array_equal(a.points.numpy().view(np.uint32), b.points.numpy().view(np.uint32))
    assert np.array_equal(a.keypoints.numpy().view(np.uint32), b.keypoints.numpy().view(np.uint32))
    assert len(a.keypoints) > 50
    a.free(); b.free()


def test_degenerate_inputs_for_the_other_feature_types(ctx, mm, textured):
    """HARRIS keypoints and the PFH / PFHRGB / SHOT descriptors on inputs that leave them nothing to do."""
    rng = np.random.default_rng(5)
    flat = _cloud(mm, np.concatenate([rng.uniform(0, 10, (20000, 2)), np.zeros((20000, 1))], 1).astype(np.float32))
    nan = _cloud(mm, np.full((100, 3), np.nan, np.float32))
    empty = _cloud(mm, np.zeros((0, 3), np.float32))
    for kt, thr in ((1, 0.001), (0, 5.0)):
        for dt in (0, 1, 3, 4, 5):
            P = mm.MapMergingParams(keypoint_type=kt, keypoint_threshold=thr, descriptor_type=dt, estimation_method=1)
            for name, c in (("flat", flat), ("nan", nan), ("empty", empty)):
                m = ctx.mapFeatures(ctx.cloud(c), P)
                assert len(m.keypoints) == len(m.descriptors), (kt, dt, name)
                if name != "flat" or kt == 0:
                    assert len(m.keypoints) == 0, (kt, dt, name)
                m.free()
            T, pairs = ctx.estimateMapsTransforms([nan, textured], P, return_pairs=True)
            assert len(pairs) == 0 and len(T) == 2
    # descriptors for keypoints far away from the surface: pruned (PFH, SHOT: NaN rows) or all-zero rows (PFHRGB)
    pts = ctx.cloud(textured)
    nrm = ctx.computeSurfaceNormals(pts, 0.6)
    far = _cloud(mm, np.array([[100, 100, 100], [200, 0, 0]], np.float32))
    for dt, n_left in ((0, 0), (4, 0), (1, 2)):
        k = ctx.cloud(far)
        d = ctx.computeLocalDescriptors(pts, nrm, k, dt, 0.8)
        assert len(d) == len(k) == n_left, dt
        if n_left:
            assert not d.numpy().any()
    # no keypoints at all
    k = ctx.cloud(far[:0])
    for dt in (0, 1, 4):
        assert len(ctx.computeLocalDescriptors(pts, nrm, k, dt, 0.8)) == 0
    # RSD keeps (0, 0) rows for them, SC3D prunes them; a value outside the enum is answered, not crashed on
    k = ctx.cloud(far)
    assert len(ctx.computeLocalDescriptors(pts, nrm, k, 3, 0.8)) == 2
    k = ctx.cloud(far)
    assert len(ctx.computeLocalDescriptors(pts, nrm, k, 5, 0.8)) == 0 and len(k) == 0
    with pytest.raises(Exception):
        ctx.computeLocalDescriptors(pts, nrm, ctx.cloud(far), 9, 0.8)


def test_a_column_of_thousands_of_points(ctx, po, mm, textured):
    """A thin vertical structure: thousands of points in one 0.25 m column of the query order (the counting sort of the
    Hilbert order gives up there and the stable radix sort takes over).  ICP, score and normals as the oracle's."""
    rng = np.random.default_rng(3)
    pole = np.zeros(5000, dtype=mm.POINT)
    pole["x"] = 1.0 + rng.normal(0, 0.01, 5000)
    pole["y"] = 2.0 + rng.normal(0, 0.01, 5000)
    pole["z"] = rng.uniform(0, 40, 5000)
    pole["rgba"] = 0xFF000000 | rng.integers(0, 1 << 24, 5000).astype(np.uint32)
    src = np.concatenate([pole, textured[:3000]])
    tgt = src.copy()
    tgt["x"] += 0.05
    T0 = np.eye(4, dtype=np.float32)
    s_ref = po.transform_score(src, tgt, T0, 1.0)
    s_got = ctx.transformScore(ctx.cloud(src), ctx.cloud(tgt), T0, 1.0)
    assert s_got == pytest.approx(s_ref, rel=1e-6)
    T_ref, it_ref = po.icp(src, tgt, T0, 1.0, 0.5, 50, 1e-2)
    T_got = ctx.estimateTransformICP(ctx.cloud(src), ctx.cloud(tgt), T0, 1.0, 0.5, 50, 1e-2)
    assert np.linalg.norm(T_got - T_ref) < 1e-3 and ctx.last_icp_iterations == it_ref
    n_ref = po.normals(src, 0.3)
    n_got = ctx.computeSurfaceNormals(ctx.cloud(src), 0.3).numpy()
    assert np.array_equal(n_got.view(np.uint32), n_ref.view(np.uint32))


def test_a_long_column_inside_a_large_cloud(ctx, po, mm):
    """More than 16.6 k points AND a column of more than 1024 of them: attempt 0 of the Hilbert work-item scan reads
    keys the counting sort never wrote (it gave up); it must stay inert instead of scattering a head per point past
    the items' bound n / 64 + 16386.  Run twice so that the second pass works on recycled pool memory."""
    rng = np.random.default_rng(11)
    n_ground, n_pole = 40000, 3000
    src = np.zeros(n_ground + n_pole, dtype=mm.POINT)
    src["x"][:n_ground] = rng.uniform(-20, 20, n_ground)
    src["y"][:n_ground] = rng.uniform(-20, 20, n_ground)
    src["z"][:n_ground] = rng.normal(0, 0.02, n_ground)
    src["x"][n_ground:] = 3.1 + rng.normal(0, 0.01, n_pole)
    src["y"][n_ground:] = -4.2 + rng.normal(0, 0.01, n_pole)
    src["z"][n_ground:] = rng.uniform(0, 30, n_pole)
    src["rgba"] = 0xFF000000 | rng.integers(0, 1 << 24, len(src)).astype(np.uint32)
    tgt = src.copy()
    tgt["y"] += 0.04
    T0 = np.eye(4, dtype=np.float32)
    s_ref = po.transform_score(src, tgt, T0, 1.0)
    for _ in range(2):
        s_got = ctx.transformScore(ctx.cloud(src), ctx.cloud(tgt), T0, 1.0)
        assert s_got == pytest.approx(s_ref, rel=1e-6)
    T_ref, it_ref = po.icp(src, tgt, T0, 1.0, 0.5, 50, 1e-2)
    T_got = ctx.estimateTransformICP(ctx.cloud(src), ctx.cloud(tgt), T0, 1.0, 0.5, 50, 1e-2)
    assert np.linalg.norm(T_got - T_ref) < 1e-3 and ctx.last_icp_iterations == it_ref
