"""Times the descriptor k-NN kernels alone (HIP events of the library's profile table).
usage: python3 scripts/knn_bench.py [dim ...]   -- dims among 33 125 1344 352; sizes: 15.7k x 15.7k and 1.4k x 15.7k (1344 / 352 also
6.4k x 6.4k, the keypoint count of configs[3]).  352 = pcl::SHOT352's shape (SURVEY 8d's "MFMA-width study": the reference binds
SHOT1344, so the width has no descriptor type; the rows go through mm3d_debug_desc_knn, one direction per call).
MM3D_KNN_WIDE_BF16=0 times the f32 selector of the wide rows instead of the split-bf16 one."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402

mm = ge.load()
ctx = mm.Context(0)
TYPE = {33: 2, 125: 0, 1344: 4}
dims = [int(a) for a in sys.argv[1:]] or [33, 125, 1344, 352]
rng = np.random.default_rng(0)
for dim in dims:
    for na, nb in ((15700, 15700), (1400, 15700)) + (((6400, 6400),) if dim in (1344, 352) else ()):
        centres = np.abs(rng.normal(0, 1, (64, dim))).astype(np.float32)

        def rows(n):
            X = centres[rng.integers(0, 64, n)] + np.abs(rng.normal(0, 0.3, (n, dim))).astype(np.float32)
            return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)

        if dim in TYPE:
            da, db = ctx.descriptors(rows(na), TYPE[dim]), ctx.descriptors(rows(nb), TYPE[dim])
            run = lambda: ctx.findFeatureCorrespondences(da, db, 5)          # noqa: E731  (both directions of the reciprocal match)
        else:
            ra, rb = rows(na), rows(nb)
            oi, od = np.empty((na, 5), dtype=np.int32), np.empty((na, 5), dtype=np.float32)
            oi2, od2 = np.empty((nb, 5), dtype=np.int32), np.empty((nb, 5), dtype=np.float32)
            p = lambda x: x.ctypes.data_as(C.c_void_p)                       # noqa: E731

            def run():
                ctx._ck(mm.lib().mm3d_debug_desc_knn(ctx._h, p(ra), na, p(rb), nb, dim, 5, p(oi), p(od)))
                ctx._ck(mm.lib().mm3d_debug_desc_knn(ctx._h, p(rb), nb, p(ra), na, dim, 5, p(oi2), p(od2)))
        run()                                                                # warm-up
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            run()
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
            if name == "desc_knn_mfma_bf16":        # the profile carries the bf16 flops the selector executes (3 products, padded K)
                kp, kp16 = (dim + 9) // 8 * 8, (dim + 6 + 15) // 16 * 16
                extra = f"  {b / ms * 1e3 / 1e12:.0f} TF/s of bf16 MFMA executed = {b / 3 * kp / kp16 / ms * 1e3 / 1e12:.1f} TF/s f32-equivalent"
            print(f"   {name:22s} {n:4d} launches  {ms / n * 1e3:9.1f} us avg{extra}")
        if dim in TYPE:
            da.free(); db.free()
