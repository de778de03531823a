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
k = ctx.detectKeypoints(f, n, P.keypoint_type, P.keypoint_threshold, P.normal_radius, P.resolution)
ctx.synchronize()
