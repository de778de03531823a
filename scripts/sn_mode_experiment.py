"""Experiment: k_sift_dog with parts of its rank step switched off (library built with -DMM3D_SN_MODE, MM3D_LIB)."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as ge
mm = ge.load()
import bench
host = bench.make_workload(16, 500000)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
d = ctx.downSample(ctx.cloud(host[0]), P.resolution)
f = ctx.removeOutliers(d, P.descriptor_radius, P.outliers_min_neighbours)
n = ctx.computeSurfaceNormals(f, P.normal_radius)
for mode in sys.argv[1:]:
    os.environ["MM3D_SN_MODE"] = mode
    for rep in range(2):
        ctx.profile_reset(); ctx.profile(True)
        try:
            k = ctx.detectKeypoints(f, n, P.keypoint_type, P.keypoint_threshold, P.normal_radius, P.resolution)
        except Exception as e:
            pass
        ctx.synchronize(); ctx.profile(False)
    e = ctx.profile_entries()
    print("mode", mode, {k: round(v["ms"] * 1e3, 1) for k, v in e.items() if k.startswith("sift_dog")}, "us for 3 octaves")
