"""Repeated estimateMapsTransforms calls on one context (the ROS node calls it every estimation tick): device and host
memory must level off."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as ge
mm = ge.load()
import torch, numpy as np
from map_merge_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
_, maps = synth.synth_maps(N, 60000 if N <= 8 else 12000, overlap_step=0.4 if N <= 8 else 0.25)
clouds = [synth.pack_points(x, c) for x, c, _ in maps]
ctx = mm.Context(0); ctx.setStreams(8)
P = mm.MapMergingParams(descriptor_type=2, estimation_method=1)
for it in range(120):
    ctx.srand(1)
    T, pairs = ctx.estimateMapsTransforms(clouds, P, return_pairs=True)
    if it % 20 == 0 or it == 119:
        free, total = torch.cuda.mem_get_info()
        rss = int(open("/proc/self/statm").read().split()[1]) * 4096 >> 20
        print(it, "device used MB", (total - free) >> 20, "host RSS MB", rss, flush=True)
