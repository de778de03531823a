"""Per-map kernel times of the feature chain, one map after the other on one stream (do the maps differ?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
import bench, torch
mm = ge.load()
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
host, _, _ = bench.make_workload_gt(16, PTS)
dev = torch.device("cuda", 0)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
names = ("sift_dog", "sift_dog_oct0", "sift_dog_oct1", "sift_dog_oct2", "normals_radius", "spfh", "sift_extrema_one")
for rep in range(2):
    for i in range(N):
        raw_t = torch.from_numpy(host[i].view(np.uint8).reshape(-1, 16)).to(dev)
        ctx.profile_reset(); ctx.profile(True)
        raw = ctx.cloud_from_ptr(raw_t.data_ptr(), len(host[i]))
        m = ctx.mapFeatures(raw, P)
        ctx.synchronize()
        e = ctx.profile_entries()
        if rep == 1:
            print(i, "pts", len(m.points.numpy()), "kp", len(m.keypoints.numpy()), {k: round(e[k]["ms"], 3) for k in names if k in e},
                  "total", round(sum(v["ms"] for v in e.values()), 2))
        raw.free(); m.free()
