"""Phase breakdown of the LDS sorted-neighbour kernels (snb_lds.hpp: k_normals_lds, k_sift_dog_lds).
Needs a library built with the counters on:
    BUILD_DIR=build_snbstats OUT=libmm3d_snbstats.so EXTRA=-DMM3D_SNB_STATS map-merge_amd/build.sh
and MM3D_LIB=.../libmm3d_snbstats.so in the environment.  Prints shader-clock ticks per phase summed over waves."""
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
names = ["items", "claim+box", "stage", "stage_barrier", "A", "B", "C", "D", "E", "consume", "end_barrier", "queries", "hits", "staged", "rounds", "total"]
out = (C.c_ulonglong * 32)()


def show(tag, fn):
    fn(out, 0)
    v = list(out)
    tot = max(v[15], 1)
    ph = {n: v[i] for i, n in enumerate(names)}
    print(tag, "items", v[0], "queries", v[11], "hits/query", round(v[12] / max(v[11], 1), 1), "staged/item", round(v[13] / max(v[0], 1), 1),
          "rounds/item", round(v[14] / max(v[0], 1), 2))
    print("   share of wave time:", {n: round(v[i] / tot, 3) for i, n in enumerate(names) if 1 <= i <= 10})
    print("   ticks per query: ", {n: round(v[i] / max(v[11], 1)) for i, n in enumerate(names) if 4 <= i <= 9},
          "| per item (per wave):", {n: round(v[i] / max(v[0], 1)) for i, n in enumerate(names) if i in (1, 2, 3, 10)})
    fn(out, 1)


raw = ctx.cloud(host[0])
d = ctx.downSample(raw, P.resolution)
f = ctx.removeOutliers(d, P.descriptor_radius, P.outliers_min_neighbours)
L.mm3d_debug_snb_stats_normals(out, 1)
L.mm3d_debug_snb_stats_sift(out, 1)
n = ctx.computeSurfaceNormals(f, P.normal_radius)
ctx.synchronize()
show("normals", L.mm3d_debug_snb_stats_normals)
k = ctx.detectKeypoints(f, n, P.keypoint_type, P.keypoint_threshold, P.normal_radius, P.resolution)
ctx.synchronize()
show("sift_dog (3 octaves)", L.mm3d_debug_snb_stats_sift)
