"""The reference's C++ API through include/map_merge_3d_shim.hpp on the MI355X: every free function of
features.h / matching.h / map_merging.h called the way R/src/map_merging.cpp:212-269 and
map_merge_tool.cpp:37-49 call them, and every result held bit for bit against direct calls of the C ABI
(which the other -m gpu tests hold against the oracle).  The binary is tests/shim/_build/shim_check,
built by __graft_entry__.build() where the reference's headers exist (tests/shim/build.sh)."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "shim", "_build", "shim_check")


class Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.d, self.o)[0]
        self.o += 8
        return v

    def arr(self, dtype, n):
        a = np.frombuffer(self.d, dtype=dtype, count=n, offset=self.o)
        self.o += a.nbytes
        return a


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8).ravel()


def test_shim_results_equal_the_c_abi(tmp_path, mm, synth):
    if not os.path.exists(EXE):
        pytest.skip("shim_check was not built (no reference headers at build time)")
    _, maps = synth.synth_maps(3, 12000)
    raws = [synth.pack_points(x, c) for x, c, _ in maps]
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<Q", len(raws)))
        for r in raws:
            f.write(struct.pack("<Q", len(r)))
            f.write(r.tobytes())
    r = subprocess.run([EXE, "gpu", inp, outp], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "shim_check gpu: ok" in r.stdout, r.stdout + r.stderr
    rd = Reader(open(outp, "rb").read())

    p = mm.MapMergingParams(descriptor_type=mm.Descriptor.FPFH, estimation_method=mm.EstimationMethod.MATCHING)
    ctx = mm.Context(0)
    ctx.setStreams(16)
    pts, kps, desc = [], [], []
    for i in range(2):
        d = ctx.downSample(ctx.cloud(raws[i]), p.resolution)
        f = ctx.removeOutliers(d, p.descriptor_radius, p.outliers_min_neighbours)
        n = ctx.computeSurfaceNormals(f, p.normal_radius)
        k = ctx.detectKeypoints(f, n, p.keypoint_type, p.keypoint_threshold, p.normal_radius, p.resolution)
        ds = ctx.computeLocalDescriptors(f, n, k, p.descriptor_type, p.descriptor_radius)
        assert np.array_equal(bits(rd.arr(mm.POINT, rd.u64())), bits(d.numpy()))
        assert np.array_equal(bits(rd.arr(mm.POINT, rd.u64())), bits(f.numpy()))
        assert np.array_equal(bits(rd.arr(mm.NORMAL, rd.u64())), bits(n.numpy()))
        assert np.array_equal(bits(rd.arr(mm.POINT, rd.u64())), bits(k.numpy()))          # pruned in place
        rows, step = rd.u64(), rd.u64()
        assert rows == len(ds) and step == 4 * 33
        assert np.array_equal(bits(rd.arr(np.float32, rows * 33)), bits(ds.numpy()))
        pts.append(f), kps.append(k), desc.append(ds)
    corr = ctx.findFeatureCorrespondences(desc[0], desc[1], int(p.matching_k))
    assert np.array_equal(bits(rd.arr(mm.CORR, rd.u64())), bits(corr))
    T_r, inl = ctx.estimateTransformFromCorrespondences(kps[0], kps[1], corr, p.inlier_threshold)
    assert np.array_equal(bits(rd.arr(np.float32, 16)), bits(T_r.T))                       # file: column-major
    assert rd.u64() == len(inl)
    T_i = ctx.estimateTransformICP(pts[0], pts[1], T_r, p.max_correspondence_distance, p.inlier_threshold, p.max_iterations,
                                   p.transform_epsilon)
    assert np.array_equal(bits(rd.arr(np.float32, 16)), bits(T_i.T))
    T_e = ctx.estimateTransform(pts[0], kps[0], desc[0], pts[1], kps[1], desc[1], p.estimation_method, True, p.inlier_threshold,
                                p.max_correspondence_distance, p.max_iterations, int(p.matching_k), p.transform_epsilon)
    assert np.array_equal(bits(rd.arr(np.float32, 16)), bits(T_e.T))
    score = ctx.transformScore(pts[0], pts[1], T_e, p.max_correspondence_distance)
    assert rd.arr(np.float64, 1)[0] == score
    T_s = ctx.estimateTransformFromDescriptorsSets(kps[0], desc[0], kps[1], desc[1], p.inlier_threshold,
                                                   p.max_correspondence_distance, p.max_iterations)
    assert np.array_equal(bits(rd.arr(np.float32, 16)), bits(T_s.T))
    Ts = ctx.estimateMapsTransforms(raws, p)
    assert rd.u64() == len(Ts)
    for T in Ts:
        assert np.array_equal(bits(rd.arr(np.float32, 16)), bits(np.asarray(T).T))
    merged = ctx.composeMaps([ctx.cloud(r) for r in raws[:len(Ts)]], Ts, p.output_resolution)
    assert np.array_equal(bits(rd.arr(mm.POINT, rd.u64())), bits(merged.numpy()))
    assert rd.o == len(rd.d)
    ctx.close()


def test_shim_on_a_device_list_writes_the_same_bytes(tmp_path, synth):
    """MM3D_DEVICES makes the shim's estimation context a device-list one (mm3d_create_devices: estimateMapsTransforms sharded
    inside the library, pair records through the RCCL all-gather) -- the way the ROS node gets several GPUs without a source
    change.  Every result of the shim's check program must be byte for byte what the plain context writes:
    with the one GPU of this box as the list [0], and as "all"."""
    if not os.path.exists(EXE):
        pytest.skip("shim_check was not built (no reference headers at build time)")
    _, maps = synth.synth_maps(3, 12000)
    raws = [synth.pack_points(x, c) for x, c, _ in maps]
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<Q", len(raws)))
        for r in raws:
            f.write(struct.pack("<Q", len(r)))
            f.write(r.tobytes())
    outs = []
    for tag, env in (("plain", {}), ("list0", {"MM3D_DEVICES": "0"}), ("all", {"MM3D_DEVICES": "all"})):
        outp = str(tmp_path / f"out_{tag}.bin")
        e = dict(os.environ)
        e.pop("MM3D_DEVICES", None)
        e.update(env)
        r = subprocess.run([EXE, "gpu", inp, outp], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0 and "shim_check gpu: ok" in r.stdout, (tag, r.stdout[-2000:] + r.stderr[-2000:])
        if env:
            assert "RCCL version" in r.stdout + r.stderr, "a device-list context creates RCCL communicators (its banner)"
        outs.append(open(outp, "rb").read())
    assert outs[0] == outs[1] == outs[2]
