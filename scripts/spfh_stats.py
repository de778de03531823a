"""What the block-cooperative SPFH kernel does on a headline map.  Needs a library built with -DMM3D_SPFH_STATS (MM3D_LIB selects it):
blocks, staged candidates, in-radius hits, pooled hits (pairs binned), second votes (pairs shared by two points of a block), ties, the share of
the pooled hits whose certified bins were refused (they take the exact pair features), and -- with -DMM3D_SPFH_VERIFY as well, where EVERY
pair is also evaluated exactly -- the number of certified bins that differ from the exact ones (must be 0).
    scripts/spfh_stats.py [points] [maps]"""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as ge
mm = ge.load()
import bench
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
host = bench.make_workload(16, PTS)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
L = mm.lib()
out = (C.c_ulonglong * 16)()
L.mm3d_debug_spfh_stats(out, 1)
N_MAPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for i in range(N_MAPS):
    m = ctx.mapFeatures(ctx.cloud(host[i]), P)
    m.free()
ctx.synchronize()
L.mm3d_debug_spfh_stats(out, 1)
v = list(out)
print("waves with a support point", v[0], "candidates tested per wave", v[1] / max(v[0], 1), "live points", v[6])
print("in-radius hits", v[2], "= per live point", v[2] / max(v[6], 1))
print("pair features evaluated", v[3], "= %.3f of the hits; second votes %d (%.3f of the hits), ties %d" % (v[3] / max(v[2], 1), v[4], v[4] / max(v[2], 1), v[5]))
print("certified bins refused for %d pooled hits = %.5f of them (exact pair features instead); certified bins that differ from the exact ones: %d%s"
      % (v[7], v[7] / max(v[3], 1), v[8], "" if v[8] == 0 else "   <-- WRONG"))
