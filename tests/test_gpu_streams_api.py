"""mm3d_set_streams: the reference's own entry point, estimateMapsTransforms, run over several HIP streams
inside the library.  Results must be bit-identical to the one-stream run, the generator must end in the
same state, and errors raised on a helper stream must come back as a status."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clouds(synth):
    _, maps = synth.synth_maps(5, 30000, overlap_step=0.4)
    return [synth.pack_points(x, c) for x, c, _ in maps]


@pytest.mark.parametrize("method", [1, 0])
def test_streams_do_not_change_the_bits(mm, clouds, method):
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=method)
    results = []
    for n_streams in (1, 3, 8):
        c = mm.Context(0)
        try:
            c.setStreams(n_streams)
            assert mm.lib().mm3d_get_streams(c._h) == n_streams
            c.srand(1)
            T, pairs = c.estimateMapsTransforms(clouds, params, return_pairs=True)
            # the generator must be where the sequential loop leaves it: a second call continues the stream
            T2, pairs2 = c.estimateMapsTransforms(clouds[:2], params, return_pairs=True)
            results.append((np.stack(T), pairs.copy(), pairs2.copy()))
        finally:
            c.close()
    for T, pairs, pairs2 in results[1:]:
        assert np.array_equal(T.view(np.uint32), results[0][0].view(np.uint32))
        assert np.array_equal(pairs.view(np.uint8), results[0][1].view(np.uint8))
        assert np.array_equal(pairs2.view(np.uint8), results[0][2].view(np.uint8))
    assert len(results[0][1]) == 10


def test_streams_with_degenerate_maps(mm, clouds):
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    empty = np.empty(0, dtype=mm.POINT)
    c = mm.Context(0)
    try:
        c.setStreams(4)
        T, pairs = c.estimateMapsTransforms([clouds[0], empty, clouds[1]], params, return_pairs=True)
        assert len(T) == 3 and len(pairs) == 1 and pairs[0]["source_idx"] == 0 and pairs[0]["target_idx"] == 2
        one = mm.Context(0)
        try:
            T1, pairs1 = one.estimateMapsTransforms([clouds[0], empty, clouds[1]], params, return_pairs=True)
        finally:
            one.close()
        assert np.array_equal(np.stack(T).view(np.uint32), np.stack(T1).view(np.uint32))
        assert np.array_equal(pairs.view(np.uint8), pairs1.view(np.uint8))
        # an invalid configuration fails on every worker: the status comes back, nothing hangs
        bad = mm.MapMergingParams(descriptor_type=9, estimation_method=1)      # not a Descriptor value
        with pytest.raises(Exception):
            c.estimateMapsTransforms([clouds[0], clouds[1]], bad)
        # and the context is still usable afterwards
        T, pairs = c.estimateMapsTransforms([clouds[0], clouds[1]], params, return_pairs=True)
        assert len(pairs) == 1
    finally:
        c.close()


def test_late_map_without_keypoints(mm, clouds):
    """The scheduler positions the rand() stream for a pair before the targets of earlier pairs exist, assuming they will
    have keypoints.  A big untextured cloud (slow to process, no SIFT keypoint) in the middle of the list breaks that
    assumption for whoever got ahead of it; the call must then still return the sequential loop's result, and leave
    the generator where that loop leaves it."""
    rng = np.random.default_rng(7)
    flat = np.zeros(400000, dtype=mm.POINT)
    flat["x"], flat["y"] = rng.uniform(0, 40, 400000), rng.uniform(0, 40, 400000)
    flat["rgba"] = 0xFF808080
    order = [clouds[0], flat, clouds[1], clouds[2], clouds[3]]
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    results = []
    for n_streams in (1, 8, 8, 8):
        c = mm.Context(0)
        try:
            c.setStreams(n_streams)
            c.srand(1)
            T, pairs = c.estimateMapsTransforms(order, params, return_pairs=True)
            T2, pairs2 = c.estimateMapsTransforms(clouds[:2], params, return_pairs=True)
            results.append((np.stack(T), pairs.copy(), pairs2.copy()))
        finally:
            c.close()
    assert len(results[0][1]) == 6 and not (results[0][1]["source_idx"] == 1).any() and not (results[0][1]["target_idx"] == 1).any()
    for T, pairs, pairs2 in results[1:]:
        assert np.array_equal(T.view(np.uint32), results[0][0].view(np.uint32))
        assert np.array_equal(pairs.view(np.uint8), results[0][1].view(np.uint8))
        assert np.array_equal(pairs2.view(np.uint8), results[0][2].view(np.uint8))


@pytest.mark.parametrize("kp_type,thr,desc,method", [(0, 5.0, 0, 0), (0, 5.0, 1, 1), (0, 5.0, 3, 1), (0, 5.0, 4, 1), (0, 5.0, 4, 0),
                                                     (0, 5.0, 5, 1), (1, 0.0005, 2, 1)])
def test_every_feature_type_under_the_stream_scheduler(mm, clouds, kp_type, thr, desc, method):
    """Keypoint / descriptor / method combinations through the multi-stream entry point: same bits as one stream."""
    params = mm.MapMergingParams(keypoint_type=kp_type, keypoint_threshold=thr, descriptor_type=desc, estimation_method=method)
    out = []
    for n_streams in (1, 6):
        c = mm.Context(0)
        try:
            c.setStreams(n_streams)
            c.srand(1)
            T, pairs = c.estimateMapsTransforms(clouds[:4], params, return_pairs=True)
            out.append((np.stack(T), pairs.copy()))
        finally:
            c.close()
    assert len(out[0][1]) == 6
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1].view(np.uint8), out[1][1].view(np.uint8))


def _run_sharded(mm, clouds, params, world, streams):
    """The N > 1 driver (mm3d_shard_*) with `world` ranks played by `world` contexts of this one GPU; the exchange
    is done with host buffers in place of the all-gather.  Returns the merged pair records."""
    from map_merge_amd import sharding
    n = len(clouds)
    ctxs, shards = [], []
    try:
        for r in range(world):
            c = mm.Context(0)
            c.setStreams(streams)
            c.srand(1)
            ctxs.append(c)
            shards.append(c.shardBegin(clouds, params, r, world))
        npts, nkp = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        for sh in shards:
            a, b = sh.bundleSizes()
            npts += a
            nkp += b
        bundles = {}
        for i in range(n):
            o = sharding.map_owner(i, world)
            buf = np.zeros(max(shards[o].bundleBytes(int(npts[i]), int(nkp[i])), 16), dtype=np.uint8)
            shards[o].pack(i, buf.ctypes.data)
            bundles[i] = buf
        for r, sh in enumerate(shards):
            for i in range(n):
                if sharding.map_owner(i, world) != r:
                    sh.unpack(i, bundles[i].ctypes.data, int(npts[i]), int(nkp[i]))
        merged, counts = None, []
        for r, sh in enumerate(shards):
            rec, mine = sh.pairs()
            counts.append(int(mine.sum()))
            if merged is None:
                merged = rec.copy()
            assert np.array_equal(rec["source_idx"], merged["source_idx"]) and np.array_equal(rec["target_idx"], merged["target_idx"])
            owners = np.array([sharding.pair_owner(int(a), int(b), world) for a, b in zip(rec["source_idx"], rec["target_idx"])])
            assert np.array_equal(mine, owners == r)
            merged[mine] = rec[mine]
        return merged, counts
    finally:
        for sh in shards:
            sh.end()
        for c in ctxs:
            c.close()


def test_sharded_driver_equals_the_one_call_path(mm, clouds):
    """mm3d_shard_* with 1, 2 and 3 ranks (owner-computes targets, packed bundles, rand() states replayed per rank)
    gives the records of ONE mm3d_estimate_maps_transforms call bit for bit -- also with a map that has no keypoint
    at all in the middle of the list (its pairs do not exist, the generator must not move for them)."""
    rng = np.random.default_rng(7)
    flat = np.zeros(60000, dtype=mm.POINT)
    flat["x"], flat["y"] = rng.uniform(0, 30, 60000), rng.uniform(0, 30, 60000)
    flat["rgba"] = 0xFF808080
    order = [clouds[0], clouds[1], flat, clouds[2], clouds[3]]
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
    c = mm.Context(0)
    try:
        c.setStreams(4)
        c.srand(1)
        T_ref, ref = c.estimateMapsTransforms(order, params, return_pairs=True)
    finally:
        c.close()
    assert len(ref) == 6
    for world in (1, 2, 3):
        merged, counts = _run_sharded(mm, order, params, world, 3)
        assert np.array_equal(merged.view(np.uint8), ref.view(np.uint8)), world
        assert sum(counts) == len(ref)
        T = mm.globalTransforms(merged, params.confidence_threshold, len(order))
        assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(T, T_ref))


@pytest.mark.parametrize("method", [1, 0])
def test_pairs_in_batches_equal_the_sequential_loop(mm, synth, method):
    """Many small maps on few streams: every worker extracts features first (n >= 2 S) and the pairs then run in
    batches of up to 16 that share a target -- one descriptor search, one SAC-IA scoring, one ICP / score launch per
    step for the whole batch.  The one-stream call is the reference's sequential loop (no batches); every pair
    record and the generator's end state must be the same bits.  One map has no keypoints (too few points), so
    batches with gaps and the rand() bookkeeping around dead pairs are covered too."""
    _, maps = synth.synth_maps(12, 9000, overlap_step=0.25)
    clouds = [synth.pack_points(x, c) for x, c, _ in maps]
    # maps of different sizes in one batch: different item counts per search job, different keypoint counts per
    # scoring job, different numbers of sampled rows in the merged descriptor search
    rng = np.random.default_rng(11)
    for i, keep in ((1, 3500), (4, 5000), (8, 2500)):
        clouds[i] = clouds[i][np.sort(rng.choice(len(clouds[i]), keep, replace=False))]
    clouds[5] = clouds[5][:40]                                  # survives the filters with no keypoint at all
    params = mm.MapMergingParams(descriptor_type=2, estimation_method=method)
    results = []
    for n_streams in (1, 2, 5):
        c = mm.Context(0)
        try:
            c.setStreams(n_streams)
            c.srand(7)
            T, pairs = c.estimateMapsTransforms(clouds, params, return_pairs=True)
            T2, pairs2 = c.estimateMapsTransforms(clouds[:3], params, return_pairs=True)
            results.append((np.stack(T), pairs.copy(), pairs2.copy()))
        finally:
            c.close()
    assert len(results[0][1]) >= 40                             # 55 pairs without the dead map's
    for T, pairs, pairs2 in results[1:]:
        assert np.array_equal(pairs.view(np.uint8), results[0][1].view(np.uint8))
        assert np.array_equal(T.view(np.uint32), results[0][0].view(np.uint32))
        assert np.array_equal(pairs2.view(np.uint8), results[0][2].view(np.uint8))


def test_device_error_flags_come_back_from_every_path(mm, clouds):
    """A point with more than 16384 neighbours inside the normal radius is more than the sorted-neighbour scratch
    holds: the kernel raises a device-side flag.  The standalone call looks at it at once; inside
    estimateMapsTransforms the map is private to its worker and the flag is only looked at at the next real wait
    (Context::check_later) -- either way the call must come back with MM3D_EUNSUPPORTED (-4), nothing may hang, and
    the context must work afterwards."""
    rng = np.random.default_rng(5)
    blob = np.zeros(20000, dtype=mm.POINT)
    xyz = rng.uniform(0.0, 0.5, size=(20000, 3)).astype(np.float32)
    blob["x"], blob["y"], blob["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    blob["rgba"] = rng.integers(0, 1 << 24, size=20000, dtype=np.uint32)
    dense = mm.MapMergingParams(descriptor_type=2, estimation_method=1, resolution=0.01, descriptor_radius=0.08,
                                outliers_min_neighbours=1, normal_radius=0.6)
    c = mm.Context(0)
    try:
        with pytest.raises(mm.Mm3dError) as e1:
            c.computeSurfaceNormals(c.cloud(blob), 0.6)
        assert e1.value.status == -4
        for n_streams in (1, 2):
            c.setStreams(n_streams)
            with pytest.raises(mm.Mm3dError) as e2:
                c.estimateMapsTransforms([blob, blob.copy(), clouds[0]], dense)
            assert e2.value.status == -4
        # the context (and its helper) go on working, and give what a fresh context gives
        params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
        c.srand(1)
        T, pairs = c.estimateMapsTransforms(clouds[:3], params, return_pairs=True)
        fresh = mm.Context(0)
        try:
            fresh.setStreams(2)
            T1, pairs1 = fresh.estimateMapsTransforms(clouds[:3], params, return_pairs=True)
        finally:
            fresh.close()
        assert np.array_equal(pairs.view(np.uint8), pairs1.view(np.uint8))
        assert np.array_equal(np.stack(T).view(np.uint32), np.stack(T1).view(np.uint32))
    finally:
        c.close()
