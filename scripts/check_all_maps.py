"""Parity of EVERY map of a bench workload, not only the two `bench.py`'s parity_check looks at: the device's
filtered points, keypoints and descriptor rows (mm3d_map_features through the C ABI) against the CPU oracle on all
host cores, array for array, bit for bit.  TEST / EVIDENCE TOOL (it runs the oracle), run on the GPU box:
    python3 scripts/check_all_maps.py [maps] [points]        (default 16 x 500000: the headline workload)
Prints one line per map and a summary; exit status 1 on any difference."""
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
P = mm.MapMergingParams(descriptor_type=mm.Descriptor.FPFH, estimation_method=mm.EstimationMethod.SAC_IA, refine_transform=1)
po.set_threads(os.cpu_count() or 1)
bad = 0
t_dev = t_cpu = 0.0
for i in range(n_maps):
    t0 = time.perf_counter()
    m = ctx.mapFeatures(ctx.cloud(host[i]), P)
    pts, kp, desc = m.points.numpy(), m.keypoints.numpy(), m.descriptors.numpy()
    t_dev += time.perf_counter() - t0
    t0 = time.perf_counter()
    d = po.downsample(host[i], P.resolution)
    f = po.remove_outliers(d, P.descriptor_radius, P.outliers_min_neighbours)
    n = po.normals(f, P.normal_radius)
    k_raw, _ = po.keypoints_sift(f, P.resolution, 3, 3, P.keypoint_threshold)
    k, e = po.descriptors_fpfh(f, n, k_raw, P.descriptor_radius)
    t_cpu += time.perf_counter() - t0
    same = [pts.tobytes() == f.tobytes(), kp.tobytes() == k.tobytes(), desc.shape == e.shape and desc.tobytes() == e.tobytes()]
    print(f"map {i:2d}: {len(pts):7d} filtered points {'==' if same[0] else '!='}  {len(kp):6d} keypoints {'==' if same[1] else '!='}  "
          f"{desc.shape[0]:6d} x {desc.shape[1]} descriptor floats {'==' if same[2] else '!='}", flush=True)
    bad += 0 if all(same) else 1
    m.free()
print(f"{n_maps - bad} of {n_maps} maps bit-equal to the oracle (points, keypoints, descriptors); device {t_dev:.1f} s incl. downloads, "
      f"oracle {t_cpu:.1f} s on {os.cpu_count()} threads")
sys.exit(1 if bad else 0)
