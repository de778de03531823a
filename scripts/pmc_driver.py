import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as ge
mm = ge.load()
import bench
host = bench.make_workload(2, 500000)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
maps = [ctx.mapFeatures(ctx.cloud(host[i]), P) for i in (0, 1)]
if len(sys.argv) > 1 and sys.argv[1] == 'pair':
    ctx.srand(1)
    r = ctx.pairEstimate(maps[0], maps[1], P)
ctx.synchronize()
