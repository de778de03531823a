#!/usr/bin/env python3
"""bench.py -- map-pairs/sec of the registration hot path on MI355X (BASELINE.json metric).

One "step" = one full estimateMapsTransforms over the workload: per-map features (voxel grid ->
outlier filter -> normals -> SIFT keypoints -> FPFH), every map pair (SAC-IA -> ICP -> score) and
the pose graph, with the raw clouds already resident in HBM when the timed region starts.

  python bench.py --gpus 1 --steps 2 --warmup 1                       # N = 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        # N > 1, one rank per GPU

N > 1: strong scaling of the SAME job, driven from inside the library (mm3d_shard_*).  A rank computes
the features of the maps it owns, ONE all-gather (RCCL) hands every rank all feature bundles, a rank
estimates the pairs whose TARGET it owns (so it builds target-side search structures for 16 / N maps
only), the pair records are all-gathered (RCCL) and every rank solves the pose graph.  Every rank
replays the rand() draws of all pairs, so the stream matches the reference's single global one.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the engine's own stream) and, at N = 1, `cpu_baseline` (the CPU oracle, single
thread like the reference, on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import sys
import threading
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA (v_mfma_f32_32x32x2_f32)
MFMA_KERNELS = {"desc_knn_mfma"}   # kernels whose profile "bytes" field carries FLOPs (csrc/desc_knn.hip)
VALU_PEAK_WINSTR_S = 256 * 4 * 2.4e9 / 4   # wave-instructions per second: 256 CUs x 4 SIMDs, 4 cycles per wave64 instruction, 2.4 GHz
# profile name (MM3D_LAUNCH) of the kernels whose C++ symbol differs from it (scripts/pmc_summary.py prints symbols)
KERNEL_OF_SYMBOL = {"k_sift_dog": "sift_dog", "k_sift_extrema": "sift_extrema", "k_spfh": "spfh", "k_normals": "normals_radius",
                    "k_nn_wave<0>": "icp_corr_reduce", "k_nn_wave<1>": "score_nn_reduce", "k_sacia_err": "sacia_err", "k_sacia_seq_sum": "sacia_seq_sum",
                    "k_fpfh_weight": "fpfh_weight", "k_knn_mfma": "desc_knn_mfma", "k_knn_rerank": "desc_knn_rerank",
                    "k_radius_outlier_count": "radius_outlier_count"}


def cgroup_cpu_limit():
    """CPUs this process may use at a time (cgroup v2 cpu.max quota / period), None if unlimited or unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        return None


def cgroup_throttle():
    """Microseconds this cgroup's threads have spent throttled by the CPU quota so far (0 if unknown)."""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1])
    except Exception:
        pass
    return 0


def make_workload(n_maps, n_points, cache=True):
    from map_merge_amd import synth
    path = f"/tmp/mm3d_bench_{n_maps}x{n_points}.npy"
    if cache and os.path.exists(path):
        try:
            arr = np.load(path)
            if arr.shape[0] == n_maps:
                return [arr[i] for i in range(n_maps)]
        except Exception:
            pass                                           # unreadable cache: regenerate
    _, maps = synth.synth_maps(n_maps, n_points)
    packed = [synth.pack_points(x, c) for x, c, _ in maps]
    if cache:
        try:                                               # several ranks may get here at once: write aside, rename atomically
            tmp = f"{path}.{os.getpid()}.tmp.npy"
            np.save(tmp, np.stack(packed))
            os.replace(tmp, path)
        except Exception:
            pass
    return packed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--maps", type=int, default=16)
    ap.add_argument("--points", type=int, default=500000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--streams", type=int, default=16, help="contexts (HIP stream + host thread) per GPU")
    ap.add_argument("--engine", choices=["library", "shard", "python"], default="library",
                    help="one GPU only: 'library' = one mm3d_estimate_maps_transforms call, streams inside libmm3d; "
                         "'shard' = the N > 1 driver (mm3d_shard_*: what several ranks always use) on one rank; "
                         "'python' = the shardable pieces driven from Python threads (diagnostic)")
    ap.add_argument("--feature-streams", type=int, default=6,
                    help="one GPU only: pipeline the stages, this many streams extract features (0 = two barriered stages)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: do not bracket kernels with HIP events in the timed region (no roofline then)")
    ap.add_argument("--descriptor", choices=["FPFH", "PFH", "SHOT"], default="FPFH")
    ap.add_argument("--method", choices=["SAC_IA", "MATCHING"], default="SAC_IA")
    ap.add_argument("--host-input", choices=["none", "pcl32"], default="none",
                    help="diagnostic (one GPU, library engine): hand the clouds over as HOST arrays of pcl::PointXYZRGB records "
                         "(stride 32, rgba at 16) like the reference's callers do, so that the step includes the upload; "
                         "never the reported configuration (inputs are HBM-resident for `value`)")
    ap.add_argument("--kernel-table", default=None, help="write rank 0's full per-kernel HIP-event table (CSV) here")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("MM3D_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("MM3D_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (test knobs: MM3D_BENCH_BACKEND=gloo MM3D_BENCH_DEVICE=0 run several ranks on ONE GPU to exercise
        # the exchange / gather code where only a single-GPU box is available)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world:
        if rank == 0:
            print(f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    mm = ge.load()
    from map_merge_amd import sharding
    ctx = mm.Context(local_rank)
    # BASELINE.json's configuration is FPFH + SAC_IA; the other combinations (PFH is the reference's default
    # descriptor, MATCHING its default method) can be timed with the flags
    desc_type = mm.Descriptor[args.descriptor]
    desc_dim = {"FPFH": 33, "PFH": 125, "SHOT": 1344}[args.descriptor]
    params = mm.MapMergingParams(descriptor_type=desc_type, estimation_method=mm.EstimationMethod[args.method],
                                 refine_transform=1)

    # ---- synthetic workload, resident in HBM before timing ---------------------------------
    n_maps, n_pts = args.maps, args.points
    host = make_workload(n_maps, n_pts, cache=not args.no_cache)
    dev_raw = [torch.from_numpy(h.view(np.uint8).reshape(-1, 16)).to(dev) for h in host]
    torch.cuda.synchronize()
    pairs_idx = [(i, j) for i in range(n_maps - 1) for j in range(i + 1, n_maps)]
    host_pcl = []
    if args.host_input == "pcl32":                        # pcl::PointXYZRGB as it lies in memory: x y z 1 | rgba pad pad pad
        PCL = np.dtype({"names": ["x", "y", "z", "w", "rgba"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u4"], "offsets": [0, 4, 8, 12, 16],
                        "itemsize": 32})
        for h in host:
            a = np.zeros(len(h), dtype=PCL)
            a["x"], a["y"], a["z"], a["w"], a["rgba"] = h["x"], h["y"], h["z"], 1.0, h["rgba"]
            host_pcl.append(a)

    stats = {}

    # Within a rank the maps and pairs are dealt once more over S contexts (one HIP stream, one host
    # thread each): a pair is a chain of dependent launches with a few host round trips, so one
    # stream leaves SIMDs idle that another stream's kernels can use.
    # A stream's host thread no longer spins while it waits (stream_wait in csrc/grid.hip polls and naps: ~0.15 of
    # a core per stream), so every rank keeps its --streams whatever the container's CPU quota is; a thread that
    # spun (MM3D_WAIT=spin) held a core, and eight ranks x 16 streams on a 16-CPU quota got every rank throttled.
    S = max(1, args.streams)
    if os.environ.get("MM3D_WAIT") == "spin":
        quota = cgroup_cpu_limit()
        cpus = quota if quota is not None else float(os.cpu_count() or 16)
        S = max(1, min(S, max(2, int(cpus // max(world, 1)) - (1 if world > 1 else 0))))
    ctxs = [ctx] + [mm.Context(local_rank) for _ in range(S - 1)]
    tpool = ThreadPoolExecutor(S) if S > 1 else None

    def run_streams(fn):
        if tpool is None:
            fn(0)
        else:
            list(tpool.map(fn, range(S)))                  # re-raises a worker's exception

    def step_sharded():
        """N > 1 (and `--engine shard` on one GPU): the driver is the library's (mm3d_shard_*): this rank's maps on the
        context's streams, ONE all-gather of the packed feature bundles (RCCL), the pairs whose target this rank
        owns, one all-gather of the pair records (RCCL), the pose graph on every rank."""
        t0 = time.perf_counter()
        ctx.srand(1)                                       # the reference's process starts at glibc seed 1
        views = [(dev_raw[i].data_ptr(), len(host[i])) for i in range(n_maps)]
        sh = ctx.shardBegin(views, params, rank, world)
        t1 = time.perf_counter()
        npts, nkp = sharding.exchange_bundles(sh, world, rank, dist if world > 1 else None, dev if backend == "nccl" else None)
        t2 = time.perf_counter()
        mine, is_mine = sh.pairs()
        t3 = time.perf_counter()
        owners = [sharding.pair_owner(int(r["source_idx"]), int(r["target_idx"]), world) for r in mine]
        mine = sharding.gather_pair_records(mine, owners, world, rank, dist if world > 1 else None, dev if backend == "nccl" else None)
        T = mm.globalTransforms(mine, params.confidence_threshold, n_maps)
        t4 = time.perf_counter()
        sh.end()
        stats.update(dict(n_pairs=len(mine), t_features=t1 - t0, t_exchange=t2 - t1, t_pairs=t3 - t2, t_gather_graph=t4 - t3,
                          pts_filtered=[int(v) for v in npts], keypoints=[int(v) for v in nkp],
                          icp_iters=[int(x) for x in mine["icp_iterations"]], pairs_here=int(is_mine.sum()),
                          n_estimated=int(sum(1 for t in T if np.any(t))),
                          crc=zlib.crc32(np.ascontiguousarray(mine["transform"]).tobytes()) & 0xffffffff))
        return T

    def step_pipelined():
        """One GPU: no exchange separates the stages, so a pair starts as soon as its two maps exist.  Only
        `--feature-streams` of the streams extract features (maps finish earlier that way), the others -- and
        the feature streams once the maps are handed out -- claim pairs in order and wait for their maps."""
        t0 = time.perf_counter()
        maps, kn = [None] * n_maps, [0] * n_maps
        ready = [threading.Event() for _ in range(n_maps)]
        lock = threading.Lock()
        next_map, next_pair = iter(range(n_maps)), iter(range(len(pairs_idx)))
        recs = np.zeros(len(pairs_idx), dtype=mm.PAIR)
        is_live = np.zeros(len(pairs_idx), dtype=bool)
        t_last_map = [t0]
        F = max(1, min(S, args.feature_streams))

        def live(q):
            i, j = pairs_idx[q]
            for e in (ready[i], ready[j]):
                if not e.wait(timeout=300):
                    raise RuntimeError("a map never became ready")
            return kn[i] > 0 and kn[j] > 0

        def worker(s):
            c = ctxs[s]
            c.srand(1)                                     # the reference's process starts at glibc seed 1
            try:
                while s < F:
                    with lock:
                        i = next(next_map, None)
                    if i is None:
                        break
                    raw = c.cloud_from_ptr(dev_raw[i].data_ptr(), len(host[i]))
                    m = c.mapFeatures(raw, params)
                    raw.free()
                    c.mapPrepare(m, params)
                    maps[i], kn[i] = m, len(m.keypoints)
                    with lock:
                        t_last_map[0] = max(t_last_map[0], time.perf_counter())
                    ready[i].set()
            except BaseException:
                for e in ready:                            # do not leave the other streams waiting
                    e.set()
                raise
            pos = 0
            while True:
                with lock:
                    p = next(next_pair, None)
                if p is None:
                    break
                for q in range(pos, p):                    # replay the draws of the pairs other streams execute
                    if live(q):
                        c.pairEstimate(maps[pairs_idx[q][0]], maps[pairs_idx[q][1]], params, execute=False)
                if live(p):
                    recs[p] = c.pairEstimate(maps[pairs_idx[p][0]], maps[pairs_idx[p][1]], params, execute=True)
                    is_live[p] = True
                pos = p + 1
            c.synchronize()

        run_streams(worker)
        t3 = time.perf_counter()
        recs["source_idx"] = [i for i, _ in pairs_idx]
        recs["target_idx"] = [j for _, j in pairs_idx]
        mine = recs[is_live]
        T = mm.globalTransforms(mine, params.confidence_threshold, n_maps)
        t4 = time.perf_counter()
        stats.update(dict(n_pairs=len(mine), t_features=t_last_map[0] - t0, t_exchange=0.0, t_pairs=t3 - t_last_map[0],
                          t_gather_graph=t4 - t3, pts_filtered=[len(m.points) for m in maps], keypoints=kn,
                          icp_iters=[int(x) for x in mine["icp_iterations"]],
                          n_estimated=int(sum(1 for t in T if np.any(t))),
                          crc=zlib.crc32(np.ascontiguousarray(mine["transform"]).tobytes()) & 0xffffffff))
        for m in maps:
            m.free()
        return T

    def step_library():
        """One GPU, default: the whole job is ONE call of the reference's own entry point on HBM-resident
        clouds; the library deals the two loops to its streams itself (mm3d_set_streams: C++ threads and
        helper contexts inside libmm3d, the same pipelined scheme as step_pipelined without the Python
        in between)."""
        ctx.srand(1)                                       # the reference's process starts at glibc seed 1
        if args.host_input == "pcl32":
            views = [(host_pcl[i].ctypes.data, len(host[i]), 32, 16) for i in range(n_maps)]
        else:
            views = [(dev_raw[i].data_ptr(), len(host[i])) for i in range(n_maps)]
        T, mine = ctx.estimateMapsTransforms(views, params, return_pairs=True)
        L = mm.lib()
        f_s, tot_s = C.c_double(), C.c_double()
        L.mm3d_last_run_stage_seconds(ctx._h, C.byref(f_s), C.byref(tot_s))
        pts, kps = (C.c_size_t * n_maps)(), (C.c_size_t * n_maps)()
        L.mm3d_last_run_map_sizes.restype = C.c_size_t
        L.mm3d_last_run_map_sizes(ctx._h, pts, kps, C.c_size_t(n_maps))
        stats.update(dict(n_pairs=len(mine), t_features=f_s.value, t_exchange=0.0, t_pairs=tot_s.value - f_s.value,
                          t_gather_graph=0.0, pts_filtered=list(pts), keypoints=list(kps),
                          icp_iters=[int(x) for x in mine["icp_iterations"]],
                          n_estimated=int(sum(1 for t in T if np.any(t))),
                          crc=zlib.crc32(np.ascontiguousarray(mine["transform"]).tobytes()) & 0xffffffff))
        return T

    if world > 1 or args.engine in ("library", "shard"):
        for c in ctxs[1:]:
            c.close()
        ctxs = [ctx]
        ctx.setStreams(S)
        step = step_library if (world == 1 and args.engine == "library") else step_sharded
    else:
        step = step_pipelined

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for c in ctxs:
        c.profile_reset()
        c.profile(not args.no_kernel_events)
    barrier()
    cpu0, thr0 = time.process_time(), cgroup_throttle()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu = {"cores_busy": round((time.process_time() - cpu0) / max(elapsed, 1e-9), 2), "cgroup_cpu_limit": cgroup_cpu_limit(),
                "cgroup_throttled_ms_per_step": round((cgroup_throttle() - thr0) / 1e3 / max(args.steps, 1), 2)}
    for c in ctxs:
        c.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {}
    for c in ctxs:                                         # per-kernel HIP-event times, summed over the rank's streams
        for k, v in c.profile_entries().items():
            e = prof.setdefault(k, {"ms": 0.0, "launches": 0, "bytes": 0.0})
            for f in ("ms", "launches", "bytes"):
                e[f] += v[f]

    # untimed: the same kernels alone on the GPU (one stream, maps 0 and 1 and their pair).  In the
    # timed region several streams share the CUs, so a kernel's HIP-event duration there includes the
    # time it spent sharing; the isolated figure is what speaks about the kernel itself.
    iso, gpu_sample = {}, None
    if rank == 0:
        ctx.profile_reset()
        ctx.profile(True)
        ctx.srand(1)
        two = []
        for i in (0, 1):
            raw = ctx.cloud_from_ptr(dev_raw[i].data_ptr(), len(host[i]))
            two.append(ctx.mapFeatures(raw, params))
            raw.free()
        rec01 = ctx.pairEstimate(two[0], two[1], params)
        ctx.synchronize()
        ctx.profile(False)
        iso = ctx.profile_entries()
        # what the device computed for maps 0 and 1 and pair (0, 1): bench's parity_check holds it against the oracle
        gpu_sample = {"maps": [dict(points=m.points.numpy(), keypoints=m.keypoints.numpy(), descriptors=m.descriptors.numpy())
                               for m in two], "pair": rec01}
        for m in two:
            m.free()
    # HBM-side traffic per launch from the most recent committed PMC passes (scripts/profile_round.sh)
    pmc_traffic, pmc_source = {}, None
    try:
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        with open(latest) as f:
            pmc_traffic = json.load(f)["bytes_per_launch"]
        pmc_source = "profiles/" + os.path.basename(latest).replace("traffic.json", "pmc_hbm_traffic.csv")
    except Exception:
        pass

    # VALU instructions per launch from the most recent committed SQ-counter pass (same maps 0 and 1, one stream)
    valu_insts, valu_source = {}, None
    try:
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_counters.csv")))[-1]
        import csv
        with open(latest) as f:
            for row in csv.DictReader(f):
                d = float(row.get("dispatches") or 0)
                if d > 0 and row.get("SQ_INSTS_VALU") not in (None, "", "nan"):
                    valu_insts[KERNEL_OF_SYMBOL.get(row["kernel"], row["kernel"])] = float(row["SQ_INSTS_VALU"]) / d
        valu_source = "profiles/" + os.path.basename(latest)
    except Exception:
        pass

    def bound_of(name, launch_ms, work):
        """Which ceiling the kernel is nearest to, from what can be known here: algorithmic bytes (or flops) per launch
        against HBM (or MFMA) peak, and -- where the committed SQ counters cover the kernel -- VALU wave-instructions per
        launch against the issue peak (256 CUs x 4 SIMDs, one wave-instruction per 4 cycles at 2.4 GHz)."""
        out = {}
        if launch_ms <= 0:
            return out
        if name in MFMA_KERNELS:
            out["mfma_frac"] = round(work / (launch_ms * 1e-3) / (MFMA_F32_PEAK_TFLOPS * 1e12), 5)
        else:
            out["hbm_frac"] = round(work / (launch_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 5)
        if name in valu_insts:
            out["valu_frac"] = round(valu_insts[name] / (launch_ms * 1e-3) / VALU_PEAK_WINSTR_S, 4)
        out["nearest"] = max(((v, k[:-5]) for k, v in out.items()), default=(0, None))[1]
        return out

    if rank == 0:
        n_pairs = stats["n_pairs"]
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        value = n_pairs * args.steps / elapsed
        # dominant kernel of this rank, by device time
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else (None, None)
        roofline = None
        if dom[0]:
            k = dom[1]
            avg_ms = k["ms"] / max(k["launches"], 1)
            work_per_launch = k["bytes"] / max(k["launches"], 1)   # bytes, or FLOPs for the MFMA kernel
            rate = work_per_launch / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
            if dom[0] in MFMA_KERNELS:
                roofline = {"kernel": dom[0], "bound": "mfma", "achieved": round(rate / 1e12, 4), "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(rate / 1e12 / MFMA_F32_PEAK_TFLOPS, 6), "traffic": None,
                            "avg_launch_us": round(avg_ms * 1e3, 3), "launches_per_step": k["launches"] / max(args.steps, 1),
                            "algorithmic_flops_per_launch": round(work_per_launch, 1)}
            else:
                roofline = {"kernel": dom[0], "bound": "hbm", "achieved": round(rate / 1e9, 3), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(rate / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                            "avg_launch_us": round(avg_ms * 1e3, 3), "launches_per_step": k["launches"] / max(args.steps, 1),
                            "algorithmic_bytes_per_launch": round(work_per_launch, 1)}
            # HBM-side bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 per
            # the gfx950 note + WRITE_SIZE); null when that kernel was not in the counted run
            roofline["traffic"] = pmc_traffic.get(dom[0])
            roofline["traffic_source"] = pmc_source if dom[0] in pmc_traffic else None
            if dom[0] in iso and iso[dom[0]]["launches"]:
                iso_ms = iso[dom[0]]["ms"] / iso[dom[0]]["launches"]
                iso_work = iso[dom[0]]["bytes"] / iso[dom[0]]["launches"]
                peak = MFMA_F32_PEAK_TFLOPS * 1e12 if dom[0] in MFMA_KERNELS else HBM_PEAK_GBS * 1e9
                roofline["isolated_avg_launch_us"] = round(iso_ms * 1e3, 3)
                roofline["isolated_frac"] = round(iso_work / (iso_ms * 1e-3) / peak, 6) if iso_ms > 0 else None
            iso_us = roofline.get("isolated_avg_launch_us")
            if roofline["traffic"] and iso_us:
                # what the memory side actually moved per launch (PMC) against the same peak: far above the algorithmic
                # bytes where a kernel keeps scratch lists in global memory (sift_dog, normals_radius)
                roofline["traffic_gbs"] = round(roofline["traffic"] / (iso_us * 1e-6) / 1e9, 1)
                roofline["traffic_frac"] = round(roofline["traffic"] / (iso_us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
            if dom[0] in valu_insts and iso_us:
                v = valu_insts[dom[0]] / (iso_us * 1e-6)
                roofline["valu"] = {"wave_instructions_per_launch": round(valu_insts[dom[0]]), "achieved": round(v / 1e9, 2),
                                    "peak": round(VALU_PEAK_WINSTR_S / 1e9, 1), "unit": "G wave-instr/s", "frac": round(v / VALU_PEAK_WINSTR_S, 4),
                                    "source": valu_source, "timed": "isolated_avg_launch_us"}
            roofline["note"] = ("timed region runs %d streams per GPU, so avg_launch_us includes time shared with other kernels; "
                                "neighbourhood kernels (sift_dog, spfh, sacia_err, *_nn_reduce) are f32-VALU-bound on "
                                "in-radius pair work, not HBM-bound: see DESIGN.md section 6" % S)
        ranked = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
        top = ranked[:8]
        if args.kernel_table:
            with open(args.kernel_table, "w") as f:
                f.write("kernel,launches_per_step,ms_per_step,avg_launch_us,algorithmic_bytes_or_flops_per_launch\n")
                for k, v in ranked:
                    f.write("%s,%.1f,%.3f,%.2f,%.0f\n" % (k, v["launches"] / max(args.steps, 1), v["ms"] / max(args.steps, 1),
                                                         1e3 * v["ms"] / max(v["launches"], 1), v["bytes"] / max(v["launches"], 1)))
                f.write("TOTAL,%.1f,%.3f,,\n" % (sum(v["launches"] for _, v in ranked) / max(args.steps, 1),
                                                 sum(v["ms"] for _, v in ranked) / max(args.steps, 1)))
        npts_f = stats["pts_filtered"]
        icp_pts = sum(npts_f[i] * it for (i, j), it in zip([(i, j) for (i, j) in pairs_idx], stats["icp_iters"]))
        out = {
            "metric": "map-pairs/sec (normals+%s+%s+ICP, end to end incl. per-map features)"
                      % (args.descriptor, "SAC-IA" if args.method == "SAC_IA" else "matching+RANSAC"),
            "value": round(value, 4), "unit": "map-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_maps} maps x {n_pts} raw pts, {args.descriptor} + {args.method} + ICP refine, {n_pairs} pairs"
                                   + (" [DIAGNOSTIC: host pcl::PointXYZRGB input, upload inside the step]" if args.host_input != "none" else ""),
                       "parallelism": (f"one mm3d_estimate_maps_transforms call, {S} streams inside the library"
                                       if world == 1 and args.engine == "library" else
                                       f"mm3d_shard_*: maps by owner, pairs by target owner over {world} GPU(s) x {S} streams inside the library"
                                       if step is step_sharded else
                                       f"maps and pairs dealt over {S} streams (Python threads)"),
                       "points_after_filter_mean": int(np.mean(npts_f)), "keypoints_mean": int(np.mean(stats["keypoints"]))},
            "pair_stage_pairs_per_s": round(n_pairs / max(stats["t_pairs"], 1e-9), 3),
            "mpoints_per_s": {
                "normals": round(sum(npts_f) / 1e6 / max(prof.get("normals_radius", {}).get("ms", 0) / 1e3 / max(args.steps, 1), 1e-9), 2)
                if "normals_radius" in prof else None,
                # FPFH: support points per second of the SPFH kernel (|S| = its algorithmic bytes / 156 B) and
                # keypoints per second of the weighting kernel
                "fpfh_spfh": round(prof["spfh"]["bytes"] / 156.0 / 1e6 / max(prof["spfh"]["ms"] / 1e3, 1e-9), 2) if "spfh" in prof else None,
                "fpfh_weight": round(sum(stats["keypoints"]) / 1e6 / max(prof["fpfh_weight"]["ms"] / 1e3 / max(args.steps, 1), 1e-9), 3)
                if "fpfh_weight" in prof else None,
                "icp": round(icp_pts / 1e6 / max(prof.get("icp_corr_reduce", {}).get("ms", 0) / 1e3 / max(args.steps, 1), 1e-9), 2)
                if "icp_corr_reduce" in prof else None,
            },
            # per kernel, alone on the GPU (the untimed one-stream pass): launch time and the ceiling it is nearest to
            "kernel_bounds_isolated": {k: dict(avg_launch_us=round(1e3 * v["ms"] / max(v["launches"], 1), 1),
                                               **bound_of(k, v["ms"] / max(v["launches"], 1), v["bytes"] / max(v["launches"], 1)))
                                       for k, v in [kv for n_, kv in enumerate(sorted(iso.items(), key=lambda kv: -kv[1]["ms"]))
                                                    if n_ < 10 or kv[0] in ("desc_knn_mfma", "normals_radius", "icp_corr_reduce", "score_nn_reduce")]},
            "stage_seconds_last_step": {k: round(stats[k], 4) for k in ("t_features", "t_exchange", "t_pairs", "t_gather_graph")},
            "top_kernels_ms_per_step": {k: round(v["ms"] / max(args.steps, 1), 3) for k, v in top},
            "maps_estimated": stats["n_estimated"],
            "icp_iterations_histogram": {str(k): int(v) for k, v in zip(*np.unique(stats["icp_iters"], return_counts=True))},
            "host_cpu": host_cpu,                           # the box's CPU quota bounds how many host threads can wait at once
            "pair_transforms_crc32": stats["crc"],          # same job, same bits: independent of --gpus / --streams
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            b1, b2, parity = cpu_baseline(host, n_maps, n_pairs, gpu_sample, True, args.descriptor, args.method, params)
            out["cpu_baseline"] = b1
            out["cpu_baseline_all_cores"] = b2
            out["parity_check"] = parity
        print(json.dumps(out))
    if tpool is not None:
        tpool.shutdown()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(host, n_maps, n_pairs, gpu_sample, check, descriptor="FPFH", method="SAC_IA", params=None):
    """The CPU oracle (kind "port": our restatement of the reference's PCL path) on a bounded sample of the
    workload: the features of ONE map and ONE pair (maps 0 and 1), extrapolated to the whole job as
    n_maps * t_map + n_pairs * t_pair.
      B1 `cpu_baseline`: one thread, like the reference's hot path (one run: the sample is ~30 s of CPU).
      B2 `cpu_baseline_all_cores`: the same code with its loops over points on every host core (OpenMP;
          results identical, tests/test_oracle_cpu.py), median of three runs.
    The oracle's results for that sample are then held against what the device computed for the same maps
    and pair in this very run (`parity_check`)."""
    import statistics
    po = ge.load_oracle()
    p = params if params is not None else po.params_default()   # the same parameter values the device ran with
    describe = {"FPFH": po.descriptors_fpfh, "PFH": po.descriptors_pfh, "SHOT": po.descriptors_shot}[descriptor]
    # B2's threads: the physical cores (SMT siblings do not help these loops), at most 64 -- and not more than the
    # container's CPU quota lets run at once (the GPU boxes show 256 CPUs and grant 16: threads beyond the quota
    # only get the whole group throttled)
    cores = max(1, min(64, (os.cpu_count() or 2) // 2))
    quota = cgroup_cpu_limit()
    if quota is not None:
        cores = max(1, min(cores, int(quota)))
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "?")
    except Exception:
        model = "?"

    def features(cloud):
        d = po.downsample(cloud, p.resolution)
        f = po.remove_outliers(d, p.descriptor_radius, p.outliers_min_neighbours)
        n = po.normals(f, p.normal_radius)
        kp, _ = po.keypoints_sift(f, p.resolution, 3, 3, p.keypoint_threshold)
        kp, desc = describe(f, n, kp, p.descriptor_radius)
        return f, kp, desc

    def pair(f0, k0, d0, f1, k1, d1):
        po.srand(1)
        if method == "SAC_IA":
            T, _, _ = po.sac_ia(k0, d0, k1, d1, p.inlier_threshold, p.max_correspondence_distance, p.max_iterations)
        else:
            corr = po.find_correspondences(d0, d1, int(p.matching_k))
            T, _, _, _ = po.ransac(k0, k1, corr, p.inlier_threshold)
        T, it = po.icp(f0, f1, T, p.max_correspondence_distance, p.inlier_threshold, p.max_iterations, p.transform_epsilon)
        score = po.transform_score(f0, f1, T, p.max_correspondence_distance)
        return T, it, score

    # B2 first (fast): all cores, median of 3; it also provides map 1's features for B1's pair
    po.set_threads(cores)
    tm, tp = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        f0, k0, d0 = features(host[0])
        tm.append(time.perf_counter() - t0)
    f1, k1, d1 = features(host[1])
    for _ in range(3):
        t0 = time.perf_counter()
        T, it, score = pair(f0, k0, d0, f1, k1, d1)
        tp.append(time.perf_counter() - t0)
    t_map2, t_pair2 = statistics.median(tm), statistics.median(tp)
    job2 = n_maps * t_map2 + n_pairs * t_pair2
    b2 = {"value": round(n_pairs / job2, 6), "unit": "map-pairs/s", "cores": cores, "kind": "port", "cpu": model,
          "sample": f"median of 3: 1 of {n_maps} maps' features ({t_map2:.2f} s) + 1 of {n_pairs} pairs ({t_pair2:.2f} s) on {cores} OpenMP "
                    f"threads, extrapolated to the job as {n_maps}*t_map + {n_pairs}*t_pair = {job2:.0f} s"}
    # B1: one thread, one run
    po.set_threads(1)
    t0 = time.perf_counter()
    g0, h0, e0 = features(host[0])
    t_map = time.perf_counter() - t0
    t0 = time.perf_counter()
    T1, it1, score1 = pair(g0, h0, e0, f1, k1, d1)
    t_pair = time.perf_counter() - t0
    job = n_maps * t_map + n_pairs * t_pair
    b1 = {"value": round(n_pairs / job, 6), "unit": "map-pairs/s", "cores": 1, "kind": "port", "cpu": model,
          "sample": f"1 of {n_maps} maps' features ({t_map:.1f} s) + 1 of {n_pairs} pairs ({t_pair:.1f} s), "
                    f"extrapolated to the job as {n_maps}*t_map + {n_pairs}*t_pair = {job:.0f} s"}
    threads_agree = (g0.tobytes() == f0.tobytes() and h0.tobytes() == k0.tobytes() and e0.tobytes() == d0.tobytes()
                     and T1.tobytes() == T.tobytes() and it1 == it and score1 == score)
    parity = None
    if check and gpu_sample is not None:
        def same(a, b):
            return a.shape == b.shape and a.tobytes() == b.tobytes()
        xyz = lambda a: np.stack([a["x"], a["y"], a["z"]], axis=1)           # noqa: E731
        g = gpu_sample["maps"]
        rec = gpu_sample["pair"]
        T_dev = np.asarray(rec["transform"], dtype=np.float32).reshape(4, 4).T
        fro = float(np.linalg.norm(T_dev - T))
        conf_rel = abs(float(rec["confidence"]) * score - 1.0)
        # ICP's Umeyama sums run in double on the device and as sequential float sums over all source points on the CPU
        # path, whose own rounding noise grows with the number of points: 1e-3 up to 500 k points, proportional beyond
        t_tol = 1e-3 * max(1.0, len(f0) / 5e5)
        parity = {
            "sample": "maps 0 and 1 and pair (0, 1) of the timed workload: device (this run) vs CPU oracle",
            "filtered_points_bit_equal": bool(same(g[0]["points"], f0) and same(g[1]["points"], f1)),
            "keypoints_bit_equal": bool(same(xyz(g[0]["keypoints"]), xyz(k0)) and same(xyz(g[1]["keypoints"]), xyz(k1))),
            "descriptors_bit_equal": bool(same(g[0]["descriptors"], d0) and same(g[1]["descriptors"], d1)),
            "n_points": [int(len(f0)), int(len(f1))], "n_keypoints": [int(len(k0)), int(len(k1))],
            "pair_transform_frobenius": round(fro, 9), "pair_transform_tolerance": t_tol,
            "confidence_rel_err": round(conf_rel, 9), "confidence_tolerance": 1e-4,
            "icp_iterations": [int(rec["icp_iterations"]), int(it)],
            "oracle_threads_agree": bool(threads_agree),
        }
        parity["ok"] = bool(parity["filtered_points_bit_equal"] and parity["keypoints_bit_equal"] and parity["descriptors_bit_equal"]
                            and fro <= t_tol and conf_rel <= 1e-4 and int(rec["icp_iterations"]) == int(it) and threads_agree)
    return b1, b2, parity


if __name__ == "__main__":
    main()
