"""The small workload the SQ-counter passes run under rocprofv3 --pmc: the feature chains of maps 0 and 1 of a bench workload on
ONE stream and (with `pair`) the estimate of pair (0, 1) -- every kernel of the path once or twice, so that "per dispatch" means
per map / per pair.
    python3 scripts/pmc_driver.py [pair] [--points N] [--descriptor FPFH|PFH|SHOT] [--method SAC_IA|MATCHING] [--scenes S]
                                  [--window W] [--resolution R] [--sac-iterations H]
The options are bench.py's and mean the same (bench.workload_signature: the counters are used for a bench run of the same
workload only); the default is the headline workload."""
import argparse
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402
mm = ge.load()
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="?", default="maps")
ap.add_argument("--points", type=int, default=500000)
ap.add_argument("--descriptor", default="FPFH")
ap.add_argument("--method", default="SAC_IA")
ap.add_argument("--scenes", default="independent")
ap.add_argument("--window", type=float, default=0.0)
ap.add_argument("--resolution", type=float, default=0.0)
ap.add_argument("--sac-iterations", type=int, default=0)
ap.add_argument("--overlap-step", type=float, default=0.5)
args = ap.parse_args()
host, _, _ = bench.make_workload_gt(2, args.points, scenes=args.scenes, overlap_step=args.overlap_step, window=args.window)
ctx = mm.Context(0)
P = mm.MapMergingParams(descriptor_type=mm.Descriptor[args.descriptor], estimation_method=mm.EstimationMethod[args.method], refine_transform=1)
if args.sac_iterations > 0:
    P.max_iterations = args.sac_iterations
if args.resolution > 0:
    P.resolution = args.resolution
maps = [ctx.mapFeatures(ctx.cloud(host[i]), P) for i in (0, 1)]
for m in maps:
    ctx.mapPrepare(m, P)
if args.what == 'pair':
    ctx.srand(1)
    r = ctx.pairEstimate(maps[0], maps[1], P)
ctx.synchronize()
