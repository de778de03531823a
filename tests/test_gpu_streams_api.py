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
