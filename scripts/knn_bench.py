"""Times the descriptor k-NN kernels alone (HIP events of the library's profile table).
usage: python3 scripts/knn_bench.py [dim ...]   -- dims among 33 125 1344; sizes: 15.7k x 15.7k and 1.4k x 15.7k"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402

mm = ge.load()
ctx = mm.Context(0)
TYPE = {33: 2, 125: 0, 1344: 4}
dims = [int(a) for a in sys.argv[1:]] or [33, 125, 1344]
rng = np.random.default_rng(0)
for dim in dims:
    for na, nb in ((15700, 15700), (1400, 15700)):
        centres = np.abs(rng.normal(0, 1, (64, dim))).astype(np.float32)

        def rows(n):
            X = centres[rng.integers(0, 64, n)] + np.abs(rng.normal(0, 0.3, (n, dim))).astype(np.float32)
            return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)

        da, db = ctx.descriptors(rows(na), TYPE[dim]), ctx.descriptors(rows(nb), TYPE[dim])
        ctx.findFeatureCorrespondences(da, db, 5)          # warm-up (both directions of the reciprocal match)
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            ctx.findFeatureCorrespondences(da, db, 5)
        ctx.synchronize()
        ent = ctx.profile_entries()
        ctx.profile(False)
        tot = sum(e["ms"] for e in ent.values())
        print(f"dim {dim}  {na} x {nb} (+ reverse): {tot / 3:.3f} ms per reciprocal match")
        for name, e in sorted(ent.items(), key=lambda kv: -kv[1]["ms"]):
            ms, n, b = e["ms"], e["launches"], e["bytes"]
            if ms / tot < 0.02:
                continue
            extra = f"  {b / ms * 1e3 / 1e12:.1f} TF/s" if name == "desc_knn_mfma" else ""
            print(f"   {name:22s} {n:4d} launches  {ms / n * 1e3:9.1f} us avg{extra}")
        da.free(); db.free()
