"""The whole job against the CPU oracle, every pair: mm3d_estimate_maps_transforms on the device and the oracle's
restated estimateMapsTransforms (one rand() stream over all pairs, like the reference) on the same clouds with the
same seed.  Per pair: Frobenius distance of the two pair transforms, relative difference of the confidences, ICP
iteration counts and last-iteration correspondence counts; then the global transforms.  TEST / EVIDENCE TOOL (it runs
the oracle: minutes of CPU on all cores), run on the GPU box:
    python3 scripts/check_all_pairs.py [maps] [points]        (default 16 x 500000: the headline workload)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

mm = ge.load()
po = ge.load_oracle()
n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
host = bench.make_workload(n_maps, n_pts)
ctx = mm.Context(0)
ctx.setStreams(16)
P = mm.MapMergingParams(descriptor_type=mm.Descriptor.FPFH, estimation_method=mm.EstimationMethod.SAC_IA, refine_transform=1)
op = po.params_default()
op.descriptor_type = 2
op.estimation_method = 1
op.refine_transform = 1
ctx.srand(1)
t0 = time.perf_counter()
T_dev, pairs_dev = ctx.estimateMapsTransforms(host, P, return_pairs=True)
t_dev = time.perf_counter() - t0
po.set_threads(os.cpu_count() or 1)
po.srand(1)
t0 = time.perf_counter()
T_ref, pairs_ref = po.estimate_maps_transforms(host, op)
t_cpu = time.perf_counter() - t0
tr = po.last_run_traces()
assert len(pairs_dev) == len(pairs_ref) == len(tr), (len(pairs_dev), len(pairs_ref), len(tr))
fro = np.array([np.linalg.norm(np.asarray(a["transform"], dtype=np.float64) - np.asarray(b["transform"], dtype=np.float64))
                for a, b in zip(pairs_dev, pairs_ref)])
conf = np.array([abs(float(a["confidence"]) / float(b["confidence"]) - 1.0) if float(b["confidence"]) != 0.0 else abs(float(a["confidence"]))
                 for a, b in zip(pairs_dev, pairs_ref)])
it_eq = sum(int(a["icp_iterations"]) == int(t["icp_iterations"]) for a, t in zip(pairs_dev, tr))
corr_eq = sum(int(a["icp_correspondences"]) == int(t["icp_correspondences"]) for a, t in zip(pairs_dev, tr))
ids_eq = sum(int(a["source_idx"]) == int(b["source_idx"]) and int(a["target_idx"]) == int(b["target_idx"]) for a, b in zip(pairs_dev, pairs_ref))
print(f"{len(fro)} pairs: pair transform Frobenius distance device vs oracle: max {fro.max():.3e}, median {np.median(fro):.3e}, "
      f"{int((fro <= 1e-3).sum())} within 1e-3")
print(f"confidence relative difference: max {conf.max():.3e}, median {np.median(conf):.3e}")
print(f"ICP iteration counts equal: {it_eq} of {len(tr)}; last-iteration correspondence counts equal: {corr_eq} of {len(tr)}; "
      f"(source, target) order equal: {ids_eq} of {len(tr)}")
g = np.array([np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) for a, b in zip(T_dev, T_ref)])
print(f"{len(g)} global transforms: Frobenius distance max {g.max():.3e}")
print(f"device {t_dev:.2f} s (first call, with warm-up), oracle {t_cpu:.1f} s on {os.cpu_count()} threads")
worst = np.argsort(-fro)[:5]
for i in worst:
    print(f"   pair {i}: ({int(pairs_dev[i]['source_idx'])}, {int(pairs_dev[i]['target_idx'])}) Frobenius {fro[i]:.3e}, ICP iterations "
          f"{int(pairs_dev[i]['icp_iterations'])} / {int(tr[i]['icp_iterations'])}")
# (the CPU path sums ICP's moments in float over all source points, the device in double: DESIGN.md section 4 has the
# yardstick; a pair that iterates twice collects that noise twice)
sys.exit(0 if (fro <= 2e-3).all() and it_eq == len(tr) else 1)
