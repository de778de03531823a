"""mm3d_create_devices: ONE process, a list of GPUs, behind the reference's own entry point (the reference's caller is a
single process: R/src/map_merge_node.cpp:133-153).  estimateMapsTransforms then shards its two loops over the devices inside
the library -- features by owner, bundles pulled GPU to GPU, pairs by target owner -- and gathers the 104-byte pair records
through ONE RCCL all-gather.  The bits must be those of a plain one-device context.

A one-GPU box can only list its device once (RCCL refuses a device twice in a communicator), so here the RCCL path runs with a
world of one, and the multi-device driver itself -- threads, ownership, peer copies, record packing -- runs on the list
[0, 0] through the documented test hook (MM3D_DEVICES_ALLOW_DUPLICATES=1: same driver, records through host memory).  Where
the box has more GPUs the full list runs for real."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clouds(synth):
    _, maps = synth.synth_maps(5, 30000, overlap_step=0.4)
    return [synth.pack_points(x, c) for x, c, _ in maps]


@pytest.fixture(scope="module")
def reference(mm, clouds):
    """(params, T, pairs, pairs of a follow-up call) per method from a plain one-device, one-stream context."""
    out = {}
    for method in (1, 0):
        params = mm.MapMergingParams(descriptor_type=2, estimation_method=method)
        c = mm.Context(0)
        try:
            c.srand(1)
            T, pairs = c.estimateMapsTransforms(clouds, params, return_pairs=True)
            T2, pairs2 = c.estimateMapsTransforms(clouds[:2], params, return_pairs=True)
            out[method] = (params, np.stack(T), pairs.copy(), pairs2.copy())
        finally:
            c.close()
    return out


def _run(mm, devices, streams, clouds, params):
    c = mm.Context(devices=devices)
    try:
        assert c.devices == list(devices)
        c.setStreams(streams)
        c.srand(1)
        T, pairs = c.estimateMapsTransforms(clouds, params, return_pairs=True)
        secs = c.lastRunDeviceSeconds()
        # the generator is where the sequential loop leaves it: a second call continues the stream
        T2, pairs2 = c.estimateMapsTransforms(clouds[:2], params, return_pairs=True)
        return np.stack(T), pairs.copy(), pairs2.copy(), c.uses_rccl, secs
    finally:
        c.close()


def _same(got, ref):
    assert np.array_equal(got[0].view(np.uint32), ref[1].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint8), ref[2].view(np.uint8))
    assert np.array_equal(got[2].view(np.uint8), ref[3].view(np.uint8))


@pytest.mark.parametrize("method", [1, 0])
def test_device_list_of_one_goes_through_rccl_and_gives_the_same_bits(mm, clouds, reference, method):
    ref = reference[method]
    got = _run(mm, [0], 4, clouds, ref[0])
    assert got[3], "a device list without duplicates gathers its pair records with ncclAllGather"
    _same(got, ref)
    assert len(got[1]) == 10 and got[4][2] > 0.0          # the gather ran and was timed


@pytest.mark.parametrize("method,streams", [(1, 1), (1, 4), (0, 3)])
def test_two_ranks_on_one_gpu_through_the_test_hook(mm, clouds, reference, method, streams):
    ref = reference[method]
    with pytest.raises(Exception):
        mm.Context(devices=[0, 0])                         # a device twice: refused without the hook
    os.environ["MM3D_DEVICES_ALLOW_DUPLICATES"] = "1"
    try:
        got = _run(mm, [0, 0], streams, clouds, ref[0])
        got3 = _run(mm, [0, 0, 0], streams, clouds, ref[0])
    finally:
        del os.environ["MM3D_DEVICES_ALLOW_DUPLICATES"]
    assert not got[3]                                      # no communicator exists for such a list
    _same(got, ref)
    _same(got3, ref)


def test_every_gpu_of_the_box(mm, clouds, reference):
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU here: the full list is covered by the driver's multi-GPU box")
    ref = reference[1]
    got = _run(mm, list(range(n)), 4, clouds, ref[0])
    assert got[3]
    _same(got, ref)


def test_bad_lists_and_failures(mm, clouds):
    for bad in ([], [99], [0, 99], [-1]):
        with pytest.raises(Exception):
            mm.Context(devices=bad)
    c = mm.Context(devices=[0])
    try:
        c.setStreams(3)
        params = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
        with pytest.raises(Exception):
            c.estimateMapsTransforms(clouds[:3], mm.MapMergingParams(descriptor_type=9, estimation_method=1))
        T, pairs = c.estimateMapsTransforms(clouds[:3], params, return_pairs=True)   # still usable
        assert len(pairs) == 3
        # degenerate inputs of the reference's gtests (R/test/test_map_merging.cpp:9-21) through a device-list context
        assert c.estimateMapsTransforms([], params) == []
        one = c.estimateMapsTransforms([np.empty(0, dtype=mm.POINT)], params)
        assert len(one) == 1 and np.array_equal(one[0], np.eye(4, dtype=np.float32))
        # every other entry point works on the list's first device
        d = c.downSample(c.cloud(clouds[0]), 0.1)
        assert len(d) > 1000
    finally:
        c.close()


def test_a_process_that_loads_a_second_rccl_afterwards_exits_cleanly():
    """libmm3d.so binds RCCL on first use (dlopen, RTLD_LOCAL).  torch ships its own librccl.so: a process that imports torch
    AFTER a device-list context existed used to abort at exit ("double free or corruption") while RCCL was opened RTLD_GLOBAL --
    and `pytest -m gpu` with it, after all tests had passed.  (The other order binds torch's copy: one RCCL in the process.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for order in ("after", "before"):
        code = ("import sys; sys.path.insert(0, %r)\n"
                "%s"
                "import __graft_entry__ as ge\nmm = ge.load()\nc = mm.Context(devices=[0]); assert c.uses_rccl; c.close()\n"
                "%s"
                "print('done', flush=True)\n") % (root, "import torch\n" if order == "before" else "",
                                                  "import torch; torch.cuda.device_count()\n" if order == "after" else "")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "done" in r.stdout, (order, r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_a_box_without_rccl_gets_an_error_not_a_crash():
    """mm3d_create_devices on a machine whose librccl cannot be loaded must return MM3D_EDEVICE with the loader's reason
    (round 5 built the message from two dlerror() calls -- the second returns NULL -- and the process died in strlen).
    MM3D_RCCL_LIB names the library to load; a file that does not exist is the "no RCCL here" case.  In a child process: the
    binding is made once per process."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import __graft_entry__ as ge\nmm = ge.load()\n"
            "try:\n    mm.Context(devices=[0])\n    print('created')\n"
            "except mm.Mm3dError as e:\n    print('refused:', e)\n") % root
    env = dict(os.environ, MM3D_RCCL_LIB="/nonexistent/librccl_missing.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "refused:" in r.stdout and "librccl could not be loaded" in r.stdout and "librccl_missing" in r.stdout, r.stdout[-800:]
    # and a one-device context made by mm3d_create never asks for the library at all
    code2 = ("import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as ge\nmm = ge.load()\nc = mm.Context(0); c.close(); print('plain ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "plain ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
