"""The ONE-process N-device driver (mm3d_create_devices) on one GPU: a list that names device 0 N times (the test hook
MM3D_DEVICES_ALLOW_DUPLICATES=1: every "device" is the same GPU, the pair records are gathered through host memory) runs the
whole multi-device schedule -- a thread and a stream set per device, features by owner, peer copies, pairs by target owner --
with the host costs of an N-GPU run and the GPU time of all N devices on one.  What it shows: the HOST side of the schedule
(round 6: one shared rand() table and per-map readiness against round 5's three lock-step stages with a private replay of
every pair on every device), the bits (CRC of the pair transforms: the one-device run's), and the library's own timeline
(MM3D_DEVICES_DEBUG=1).  The wall time is NOT a prediction for N GPUs.
    python3 scripts/emulate_devices.py [maps] [points] [N] [streams per device]      (default 64 50000 8 2: configs[4])"""
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if os.environ.get("MM3D_EMULATE_CHILD"):
    import __graft_entry__ as ge
    import bench
    import torch
    mm = ge.load()
    n_maps, n_pts, N, S = (int(a) for a in sys.argv[1:5])
    host, _, _ = bench.make_workload_gt(n_maps, n_pts)
    dev = torch.device("cuda", 0)
    dev_raw = [torch.from_numpy(h.view(np.uint8).reshape(-1, 16)).to(dev) for h in host]
    params = mm.MapMergingParams(descriptor_type=mm.Descriptor.FPFH, estimation_method=mm.EstimationMethod.SAC_IA, refine_transform=1)
    ctx = mm.Context(devices=[0] * N) if N > 1 else mm.Context(0)
    ctx.setStreams(S if N > 1 else S * 8)
    views = [(dev_raw[i].data_ptr(), len(host[i])) for i in range(n_maps)]
    best = None
    for rep in range(4):
        ctx.srand(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T, pairs = ctx.estimateMapsTransforms(views, params, return_pairs=True)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    crc = zlib.crc32(np.ascontiguousarray(pairs["transform"]).tobytes())
    print(f"RESULT wall {1e3 * best:.2f} ms per call (best of 4), {len(pairs)} pairs, pair_transforms_crc32 {crc}", flush=True)
    sys.exit(0)

n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
S = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for label, env in (("one device (reference bits)", {"N": 1}), ("pipelined (round 6)", {}), ("staged (round 5)", {"MM3D_DEVICES_STAGED": "1"})):
    e = dict(os.environ, MM3D_EMULATE_CHILD="1", MM3D_DEVICES_ALLOW_DUPLICATES="1", MM3D_DEVICES_DEBUG="1")
    n = env.pop("N", N)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n_maps), str(n_pts), str(n), str(S)], env=e, capture_output=True, text=True, timeout=1500)
    print(f"== {label}: {n_maps} x {n_pts}, {n} device(s) x {S if n > 1 else S * 8} streams")
    lines = [l for l in r.stderr.splitlines() if l.startswith("mm3d devices")]
    for l in lines[-(n if "staged" in label else 1):]:
        print("   " + l)
    print("   " + (r.stdout.strip().splitlines() or ["(no result)"])[-1])
    if r.returncode != 0:
        print(r.stderr[-2000:])
