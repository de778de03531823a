"""Latency of ONE pair estimate alone on the GPU (what bounds a job of many small maps): wall time of
mm3d_pair_estimate against the sum of its kernels' HIP-event times.  argv: points per map (default 50000)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
import bench, torch
mm = ge.load()
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
host = bench.make_workload(64 if PTS == 50000 else 16, PTS)
dev = torch.device("cuda", 0)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
maps = []
for i in range(4):
    t = torch.from_numpy(host[i].view(np.uint8).reshape(-1, 16)).to(dev)
    raw = ctx.cloud_from_ptr(t.data_ptr(), len(host[i]))
    m = ctx.mapFeatures(raw, P); ctx.mapPrepare(m, P); raw.free()
    maps.append(m)
ctx.synchronize()
pairs = [(0, 1), (0, 2), (1, 2), (2, 3), (0, 3), (1, 3)]
for prof in (False, False, True):
    ctx.profile_reset(); ctx.profile(prof)
    ctx.srand(1)
    t0 = time.perf_counter()
    for a, b in pairs:
        r = ctx.pairEstimate(maps[a], maps[b], P)
    ctx.synchronize()
    t1 = time.perf_counter()
    print(f"profile={prof}: {1e3 * (t1 - t0) / len(pairs):.3f} ms wall per pair")
e = ctx.profile_entries()
tot = sum(v["ms"] for v in e.values()); n = sum(v["launches"] for v in e.values())
print(f"kernels {tot / len(pairs):.3f} ms per pair in {n / len(pairs):.1f} launches")
for k, v in sorted(e.items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print(f"   {k:26s} {v['launches'] / len(pairs):6.1f}  {v['ms'] / len(pairs):.4f} ms")
