"""Latency of ONE map's feature chain alone on the GPU (what bounds a rank that owns few maps): wall time of
mm3d_map_features + mm3d_map_prepare against the sum of its kernels' HIP-event times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
import bench, torch
mm = ge.load()
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
host = bench.make_workload(16, PTS)
dev = torch.device("cuda", 0)
raw_t = torch.from_numpy(host[0].view(np.uint8).reshape(-1, 16)).to(dev)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
import ctypes as C
L = mm.lib()
w0, w1 = (C.c_longlong * 2)(), (C.c_longlong * 2)()
for rep in range(4):
    ctx.profile_reset(); ctx.profile(rep == 3)
    L.mm3d_debug_waits(ctx._h, w0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    raw = ctx.cloud_from_ptr(raw_t.data_ptr(), len(host[0]))
    m = ctx.mapFeatures(raw, P)
    t1 = time.perf_counter()
    ctx.mapPrepare(m, P)
    ctx.synchronize()
    t2 = time.perf_counter()
    L.mm3d_debug_waits(ctx._h, w1)
    if rep == 2:     # (the last run without kernel events)
        print(f"no events: features {1e3 * (t1 - t0):.2f} ms + prepare {1e3 * (t2 - t1):.2f} ms wall, {w1[0] - w0[0]} host waits taking {1e-6 * (w1[1] - w0[1]):.2f} ms")
    raw.free(); m.free()
e = ctx.profile_entries()
tot = sum(v["ms"] for v in e.values()); n = sum(v["launches"] for v in e.values())
print(f"features {1e3 * (t1 - t0):.2f} ms + prepare {1e3 * (t2 - t1):.2f} ms wall; kernels {tot:.2f} ms in {n} launches")
for k, v in sorted(e.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("TOPN", "14"))]:
    print(f"   {k:26s} {v['launches']:4d}  {v['ms']:.3f} ms")
