"""The certified SIFT decision (csrc/sift_cert.hpp): detectKeypoints(SIFT) returns an index set (R/src/features.cpp:45-62),
so the later octaves decide it from an UNSORTED scale space with a rigorous error bound and give only the points that
leaves open the CPU path's exact float sums.  Here: the pieces the certificate rests on, against the CPU oracle.
Run with `-m gpu` on the MI355X box; all calls go through the C ABI."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RES, R_DESC, R_NRM, MIN_NB = 0.1, 0.8, 0.6, 50


def xyz(a):
    return np.stack([a["x"], a["y"], a["z"]], axis=1)


@pytest.fixture(autouse=True)
def every_octave_certified(mm):
    """The product certifies octaves of at least 15 000 points (smaller ones are cheaper on the sorted lists); these tests want
    the certified path on their small scenes too."""
    mm.lib().mm3d_debug_sift_cert_min(0)
    yield
    mm.lib().mm3d_debug_sift_cert_min(-1)


@pytest.fixture(scope="module")
def big_scene(po, synth):
    """One synthetic map large enough for three populated octaves (about 50 k filtered points)."""
    world, maps = synth.synth_maps(2, 70000, overlap_step=0.35)
    x, c, _ = maps[0]
    raw = synth.pack_points(x, c)
    return po.remove_outliers(po.downsample(raw, RES), R_DESC, MIN_NB)


def test_v_exp_f32_is_within_two_ulp_on_the_weights_range(ctx, mm):
    """The bound budgets 2 ulp for v_exp_f32.  Every float argument the pass can form lies in [-6.5, 0]: all 106 M floats of
    [-6.6, -1e-3] and every 97th of (-1e-3, 0] against numpy's double exp2."""
    lo, hi = np.float32(-1e-3).view(np.uint32), np.float32(-6.6).view(np.uint32)
    worst = 0.0
    chunk = 1 << 24
    starts = list(range(int(lo), int(hi) + 1, chunk))
    for a in starts:
        bits = np.arange(a, min(a + chunk, int(hi) + 1), dtype=np.uint32)
        x = bits.view(np.float32)
        out = np.empty_like(x)
        ctx._ck(mm.lib().mm3d_debug_libm(ctx._h, 5, x.ctypes.data_as(C.c_void_p), None, len(x), out.ctypes.data_as(C.c_void_p)))
        true = np.exp2(x.astype(np.float64))
        ulp = np.spacing(true.astype(np.float32)).astype(np.float64)
        worst = max(worst, float(np.max(np.abs(out.astype(np.float64) - true) / ulp)))
    bits = np.arange(0x80000000, int(lo), 97, dtype=np.uint32)
    x = bits.view(np.float32)
    out = np.empty_like(x)
    ctx._ck(mm.lib().mm3d_debug_libm(ctx._h, 5, x.ctypes.data_as(C.c_void_p), None, len(x), out.ctypes.data_as(C.c_void_p)))
    true = np.exp2(x.astype(np.float64))
    worst = max(worst, float(np.max(np.abs(out.astype(np.float64) - true) / np.spacing(true.astype(np.float32)).astype(np.float64))))
    print(f"v_exp_f32: worst error {worst:.3f} ulp")
    assert worst <= 2.0


def test_the_bound_holds_on_every_point(ctx, po, big_scene):
    """|the CPU path's float DoG - val*| <= B for every point and DoG column of every octave, the octave clouds being the
    oracle's; and the bound is tight enough to decide with (median B far below the contrast threshold of 5)."""
    cloud = ctx.cloud(big_scene)
    for octv in range(3):
        r = po.sift_octave_debug(big_scene, RES, octv)
        got = ctx.siftCertOctave(cloud, RES, octv)
        assert r is not None and got is not None
        oc, dog, resp, cnt, knn = r
        val, bound = got
        assert val.shape == dog.shape
        err = np.abs(dog.astype(np.float64) - val.astype(np.float64))
        assert np.isfinite(bound).all() and (bound > 0).all()
        worst = float(np.max(err / bound))
        print(f"octave {octv}: n {len(oc)}, worst |float DoG - val*| / B = {worst:.3f}, median B {np.median(bound):.5f}")
        assert worst <= 1.0
        assert np.median(bound) < 0.05
        # and against the real value (double evaluation): the device's own share of the bound
        real = resp[:, 1:] - resp[:, :-1]
        assert float(np.max(np.abs(real - val.astype(np.float64)) / bound)) <= 1.0


def test_certified_octaves_give_the_oracles_keypoints_and_say_what_they_did(ctx, po, mm, big_scene):
    """Without normals to fuse into it the first octave is certified too (three octaves); with them (the reference's call: the
    descriptors need the normals) it keeps its sorted lists and the later two are certified."""
    mm.sift_cert_stats(reset=True)
    m = ctx.mapFeatures(ctx.cloud(big_scene), mm.MapMergingParams(descriptor_type=mm.Descriptor.FPFH, estimation_method=mm.EstimationMethod.SAC_IA))
    assert mm.sift_cert_stats()[0] == 2 and len(m.keypoints) > 500      # the map's feature chain fuses the normals into the first octave
    m.free()
    mm.sift_cert_stats(reset=True)
    ref, _ = po.keypoints_sift(big_scene, RES, 3, 3, 5.0)
    got = ctx.detectKeypoints(ctx.cloud(big_scene), None, 0, 5.0, R_NRM, RES).numpy()
    assert len(got) == len(ref) > 500
    assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))
    st = mm.sift_cert_stats()
    print("certified SIFT statistics:", st)
    assert st[0] == 3                      # all three octaves were decided on the certified path (no normals asked for: nothing is fused into the first)
    assert st[4] == 0 and st[5] == 0 and st[6] == 0
    assert 0 < st[2] < 0.02 * st[1]        # a small share of their points needed the exact sums


def test_ties_everywhere_send_every_point_to_the_exact_path(ctx, po, mm):
    """The power-of-two step-edge lattice of test_sift_ties_across_scales_are_not_extrema with the contrast threshold at zero:
    exact ties among the responses, so no interval comparison can be certain -- every candidate is left open, takes the exact
    path, and the keypoints are still the oracle's."""
    gx, gy = np.meshgrid(np.arange(-30, 31) * 0.1, np.arange(-30, 31) * 0.1)
    pts = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1).astype(np.float32)
    c = np.zeros(len(pts), dtype=po.POINT)
    c["x"], c["y"], c["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    for rgba in (np.full(len(c), 0xFF808080, dtype=np.uint32), np.where(c["x"] < 0.0, 0xFF404040, 0xFF808080).astype(np.uint32)):
        c["rgba"] = rgba
        mm.sift_cert_stats(reset=True)
        ref, _ = po.keypoints_sift(c, 0.1, 3, 3, 0.0)
        got = ctx.detectKeypoints(ctx.cloud(c), None, 0, 0.0, R_NRM, 0.1).numpy()
        st = mm.sift_cert_stats()
        print("ties:", st, len(ref))
        assert len(got) == len(ref)
        assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))
        assert st[0] == 3 and st[4] == 0 and st[5] == 0 and st[6] == 0
        assert st[2] > 0.5 * st[1]         # most points of the octaves needed the exact sums


def test_the_hard_scenes_of_the_parity_suite_on_the_certified_path(ctx, po, big_scene):
    """tests/test_gpu_parity.py holds these scenes to the oracle on the sorted-list path (their octaves are under the 15 000
    points from which the product certifies); here the same scenes with every octave certified: a cloud thinned to a fifth (most
    3 sigma_max balls of the first octave hold fewer than 25 points: the search radius is unknown and the ring-growing interval
    search serves them), a solid block of points inside an ordinary scene (boxes far larger than the tile: the unsorted pass
    streams them), and the single-blob lattice (one keypoint, on the blob)."""
    rng = np.random.default_rng(5)
    thin = big_scene[np.sort(rng.choice(len(big_scene), len(big_scene) // 5, replace=False))].copy()
    ref, _ = po.keypoints_sift(thin, RES, 3, 3, 1.0)
    got = ctx.detectKeypoints(ctx.cloud(thin), None, 0, 1.0, R_NRM, RES).numpy()
    assert len(got) == len(ref) > 20 and np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))
    c0 = xyz(big_scene).mean(axis=0)
    blob = np.zeros(9000, dtype=big_scene.dtype)
    p = rng.uniform(-0.75, 0.75, (9000, 3)).astype(np.float32) + c0.astype(np.float32)
    blob["x"], blob["y"], blob["z"] = p[:, 0], p[:, 1], p[:, 2]
    blob["rgba"] = rng.integers(0, 1 << 24, 9000).astype(np.uint32)
    cloud = np.concatenate([big_scene, blob])
    ref, _ = po.keypoints_sift(cloud, RES, 3, 3, 5.0)
    got = ctx.detectKeypoints(ctx.cloud(cloud), None, 0, 5.0, R_NRM, RES).numpy()
    assert len(got) == len(ref) > 100 and np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))
    gx, gy = np.meshgrid(np.arange(-40, 41) * 0.1, np.arange(-40, 41) * 0.1)
    pts = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1).astype(np.float32)
    for amplitude, sigma, expect in ((150.0, 0.25, 1), (-55.0, 0.4, 1), (4.0, 0.4, 0)):
        c = np.zeros(len(pts), dtype=po.POINT)
        c["x"], c["y"], c["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
        g = amplitude * np.exp(-((pts[:, 0] - 0.33) ** 2 + (pts[:, 1] + 0.21) ** 2) / (2 * sigma * sigma))
        lum = np.clip(60.0 + g, 0, 255).astype(np.uint32)
        c["rgba"] = (0xFF << 24) | (lum << 16) | (lum << 8) | lum
        got = ctx.detectKeypoints(ctx.cloud(c), None, 0, 5.0, R_NRM, 0.1).numpy()
        ref, _ = po.keypoints_sift(c, 0.1, 3, 3, 5.0)
        assert len(got) == len(ref) == expect and np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32))


def random_scene(po, seed):
    """Eight seeded scenes of different character -- textured sheets, clumps with empty space between them, sparse scatter, a
    jittered lattice with duplicated points, other resolutions and thresholds: (downsampled cloud, resolution, threshold)."""
    rng = np.random.default_rng(100 + seed)
    kind = seed % 4
    n = int(rng.integers(6000, 40000))
    if kind == 0:        # an undulating textured sheet
        xy = rng.uniform(-8, 8, (n, 2))
        z = 0.3 * np.sin(xy[:, 0]) * np.cos(0.7 * xy[:, 1]) + rng.normal(0, 0.01, n)
        p = np.column_stack([xy, z])
    elif kind == 1:      # clumps of very different density, empty space between them
        c = rng.uniform(-10, 10, (12, 3)) * np.array([1, 1, 0.2])
        p = c[rng.integers(0, 12, n)] + rng.normal(0, 1, (n, 3)) * rng.uniform(0.1, 1.2, (n, 1))
    elif kind == 2:      # sparse scatter in a slab
        p = rng.uniform(-15, 15, (n, 3)) * np.array([1, 1, 0.05])
    else:                # a jittered lattice, a tenth of the points duplicated
        g = np.stack(np.meshgrid(np.arange(-60, 60), np.arange(-60, 60)), -1).reshape(-1, 2)[:n] * 0.11
        p = np.column_stack([g, np.zeros(len(g))]) + rng.normal(0, 0.004, (len(g), 3))
        p = np.concatenate([p, p[rng.integers(0, len(p), len(p) // 10)]])
    c = np.zeros(len(p), dtype=po.POINT)
    c["x"], c["y"], c["z"] = p[:, 0].astype(np.float32), p[:, 1].astype(np.float32), p[:, 2].astype(np.float32)
    tex = 128 + 90 * np.sin(1.7 * p[:, 0] + 0.3 * seed) * np.cos(2.3 * p[:, 1]) + rng.normal(0, 12, len(p))
    lum = np.clip(tex, 0, 255).astype(np.uint32)
    c["rgba"] = (0xFF << 24) | (lum << 16) | (np.clip(lum + rng.integers(-20, 20, len(p)), 0, 255).astype(np.uint32) << 8) | lum
    res = (0.1, 0.1, 0.2, 0.05)[seed % 4] if seed < 4 else 0.1
    thr = (5.0, 1.0, 0.2, 12.0)[(seed // 2) % 4]
    return po.downsample(c, res), res, thr


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_on_the_certified_path(ctx, po, seed):
    """The eight scenes through the certified octaves against the oracle's keypoints, bit for bit."""
    cloud, res, thr = random_scene(po, seed)
    ref, _ = po.keypoints_sift(cloud, res, 3, 3, thr)
    got = ctx.detectKeypoints(ctx.cloud(cloud), None, 0, thr, R_NRM, res).numpy()
    assert len(got) == len(ref), (seed, len(got), len(ref))
    assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32)), seed


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_on_the_sorted_lists(ctx, po, mm, seed):
    """The same scenes with every octave on the sorted neighbour lists (what the product does below 15 000 points): the points
    whose 3 sigma_max ball holds fewer than 25 neighbours -- most of the clumps' outskirts and of the sparse slab -- are decided
    by k_sift_extrema_one on the exact values, in the same pass as the others."""
    cloud, res, thr = random_scene(po, seed)
    mm.lib().mm3d_debug_sift_cert_min(1 << 30)
    mm.sift_cert_stats(reset=True)
    ref, _ = po.keypoints_sift(cloud, res, 3, 3, thr)
    got = ctx.detectKeypoints(ctx.cloud(cloud), None, 0, thr, R_NRM, res).numpy()
    assert mm.sift_cert_stats()[0] == 0          # no octave took the certified path
    assert len(got) == len(ref), (seed, len(got), len(ref))
    assert np.array_equal(xyz(got).view(np.uint32), xyz(ref).view(np.uint32)), seed
