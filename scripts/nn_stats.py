"""Per-wave timing of the ICP / score search kernel (k_nn_wave).  Needs a library built with the kernel's
counters on:  hipcc ... -c csrc/nn.hip -DMM3D_NN_STATS=2  (=1 also counts passes / rows / staged points, which
perturbs the timing), linked in place of build/nn.o.  Prints the histogram of wave durations and the share of the
three phases (row headers, staging, candidate scan)."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as ge
mm = ge.load()   # MM3D_LIB selects the instrumentation build
import bench, numpy as np
PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
host = bench.make_workload(64 if PTS == 50000 else 16, PTS)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
PAIRS = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(0, 1)]
need = sorted({i for pr in PAIRS for i in pr})
maps = {i: ctx.mapFeatures(ctx.cloud(host[i]), P) for i in need}
L = mm.lib()
out = (C.c_ulonglong * 64)()
for (a, b) in PAIRS:
    ctx.srand(1)
    L.mm3d_debug_nn_stats(out, 1)
    r = ctx.pairEstimate(maps[a], maps[b], P)
    ctx.synchronize()
    L.mm3d_debug_nn_stats(out, 1)
    v = list(out)
    print(f"pair ({a}, {b}): icp iters", r["icp_iterations"])
    print("  waves", v[0], "passes", v[1], "row chunks", v[2], "staged", v[3], "active lanes@pass", v[4], "rows", v[5])
    print("  max wave ticks(100MHz)", v[6], "= us", v[6] / 100.0, " mean us", v[7] / max(v[0], 1) / 100.0)
    print("  hist log2(ticks):", {b: v[8 + b] for b in range(24) if v[8 + b]})
    print("  phase ticks: headers", v[32], "staging", v[33], "scan", v[34], " total wave ticks", v[7])
    print("  per ring size E: passes", v[40:48], " ticks", v[48:56])
    print("  staged candidates dropped by the corner filter (no active lane can still use them):", v[38], "of", v[3])
    print("  lanes by the ring they ask for before the first pass (1..7, >=8):", v[56:64])
