"""Phase breakdown of the sorted-neighbour-list kernels (sorted_nb.hpp: k_normals, k_sift_dog, k_fpfh_weight).
Needs a library built with the counters on:
    BUILD_DIR=build_snstats OUT=libmm3d_snstats.so EXTRA=-DMM3D_SN_STATS map-merge_amd/build.sh
and MM3D_LIB=.../libmm3d_snstats.so in the environment.  Prints 100 MHz ticks per phase summed over waves."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as ge
mm = ge.load()
import bench, numpy as np
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
host = bench.make_workload(64 if PTS == 50000 else 16, PTS)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
L = mm.lib()
names = ["groups", "pass1", "prefix", "pass2", "rank", "chains", "queries", "entries", "staged"]
out = (C.c_ulonglong * 16)()


def show(tag, fn):
    fn(out, 0)
    v = list(out)
    tot = sum(v[1:6])
    print(tag, {n: v[i] for i, n in enumerate(names)}, "| shares:",
          {n: round(v[i] / max(tot, 1), 3) for i, n in enumerate(names) if 1 <= i <= 5},
          "| entries/query", round(v[7] / max(v[6], 1), 1), "staged/group", round(v[8] / max(v[0], 1), 1),
          "ticks/group", round(tot / max(v[0], 1), 1), "| staging: header ticks", v[9], "tile ticks", v[10], "tiles", v[11], "rows", v[12])
    fn(out, 1)


raw = ctx.cloud(host[0])
d = ctx.downSample(raw, P.resolution)
f = ctx.removeOutliers(d, P.descriptor_radius, P.outliers_min_neighbours)
for fn in (L.mm3d_debug_sn_stats, L.mm3d_debug_sn_stats_sift):
    fn(out, 1)
n = ctx.computeSurfaceNormals(f, P.normal_radius)
ctx.synchronize()
show("normals", L.mm3d_debug_sn_stats)
k = ctx.detectKeypoints(f, n, P.keypoint_type, P.keypoint_threshold, P.normal_radius, P.resolution)
ctx.synchronize()
show("sift_dog (3 octaves)", L.mm3d_debug_sn_stats_sift)
if hasattr(L, "mm3d_debug_sn_stats_fpfh"):
    L.mm3d_debug_sn_stats_fpfh(out, 1)
    ds = ctx.computeLocalDescriptors(f, n, k, P.descriptor_type, P.descriptor_radius)
    ctx.synchronize()
    show("fpfh_weight", L.mm3d_debug_sn_stats_fpfh)
