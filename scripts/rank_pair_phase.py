"""Where the pair phase of ONE rank of an 8-rank run spends its wall time (emulated on one GPU like scripts/emulate_rank.py):
wall time of mm3d_shard_pairs against the sum of its kernels' HIP-event times, per kernel.  usage: rank_pair_phase.py [world] [rank]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
import bench, torch
mm = ge.load()
from map_merge_amd import sharding
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_maps, n_pts = 16, 500000
host, _, _ = bench.make_workload_gt(n_maps, n_pts)
dev = torch.device("cuda", 0)
dev_raw = [torch.from_numpy(h.view(np.uint8).reshape(-1, 16)).to(dev) for h in host]
views = [(dev_raw[i].data_ptr(), len(host[i])) for i in range(n_maps)]
params = mm.MapMergingParams(descriptor_type=mm.Descriptor.FPFH, estimation_method=mm.EstimationMethod.SAC_IA, refine_transform=1)
ctx = mm.Context(0)
ctx.setStreams(int(os.environ.get("MM3D_STREAMS", "16")))
sh = ctx.shardBegin(views, params, 0, 1)
npts, nkp = sh.bundleSizes()
bundles = []
for i in range(n_maps):
    b = torch.zeros(max(sh.bundleBytes(int(npts[i]), int(nkp[i])), 16), dtype=torch.uint8, device=dev)
    sh.pack(i, b.data_ptr()); bundles.append(b)
sh.end()
for prof in (False, False, True):
    ctx.srand(1)
    s = ctx.shardBegin(views, params, rank, world)
    s.unpackMany([(i, bundles[i].data_ptr(), int(npts[i]), int(nkp[i])) for i in range(n_maps) if sharding.map_owner(i, world) != rank])
    ctx.synchronize()
    ctx.profile_reset(); ctx.profile(prof)
    t0 = time.perf_counter()
    rec, mine = s.pairs()
    t1 = time.perf_counter()
    ctx.profile(False)
    s.end()
    print(f"profile={prof}: pairs phase {1e3 * (t1 - t0):.2f} ms wall for {int(mine.sum())} pairs")
e = ctx.profile_entries()
tot = sum(v["ms"] for v in e.values()); n = sum(v["launches"] for v in e.values())
print(f"kernels {tot:.2f} ms in {n} launches")
for k, v in sorted(e.items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print(f"   {k:26s} {v['launches']:5d}  {v['ms']:.3f} ms")
