"""Sharding of the map / pair loops of estimateMapsTransforms over one process per GPU.

The reference runs both loops sequentially in one thread (R/src/map_merging.cpp:212-242 per map,
:256-269 per pair).  Maps and pairs are independent units.  The driver itself lives in the library
(mm3d_shard_*, include/mm3d.h): a rank extracts the features of the maps it owns and estimates the pairs
whose TARGET it owns; this module is the torch.distributed plumbing around it -- (C2) ONE all-gather of
the maps' feature bundles, packed per rank, and (C1) one all-gather of the fixed-size pair records (RCCL
over xGMI on the GPU box, gloo in the CPU tests) -- before every rank solves the (tiny, host-side) pose graph.
"""
from __future__ import annotations

import numpy as np


def map_owner(i: int, world: int) -> int:
    """0 1 .. w-1 w-1 .. 1 0 0 1 ..  (== mm3d_shard_map_owner): target j has j pairs, and j and its mirror
    image share a rank, so the pair counts per rank come out even (16 maps on 8 ranks: 15 pairs each)."""
    if world <= 1:
        return 0
    j = i % (2 * world)
    return j if j < world else 2 * world - 1 - j


def pair_owner(i: int, j: int, world: int) -> int:
    """The rank that owns the pair's target map (it holds the target-side search structures)."""
    return map_owner(j, world)


def live_pairs(n_maps: int, keypoint_counts) -> list:
    """Pairs (i < j) whose maps both have keypoints, in the reference's order (map_merging.cpp:246-254)."""
    return [(i, j) for i in range(n_maps - 1) for j in range(i + 1, n_maps)
            if keypoint_counts[i] > 0 and keypoint_counts[j] > 0]


def all_gather_bytes(buf, world: int, dist):
    """One all-gather of equally sized byte tensors (device tensors over RCCL, host tensors over gloo)."""
    import torch
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return out


def gather_pair_records(records: np.ndarray, owners, world: int, rank: int, dist=None, device=None) -> np.ndarray:
    """All-gather the pair records; slot q is taken from rank owners[q].  `records` is the local array (all
    slots present, only the owned ones meaningful).  Returns the merged array."""
    if world == 1:
        return records
    import torch
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(len(records), -1)
    buf = torch.from_numpy(raw.copy())
    if device is not None:
        buf = buf.to(device)
    allb = all_gather_bytes(buf, world, dist)
    merged = records.copy()
    owners = np.asarray(owners)
    for r in range(world):
        if r == rank:
            continue
        other = allb[r].cpu().numpy().reshape(-1).view(records.dtype)
        sel = owners == r
        merged[sel] = other[sel]
    return merged


def exchange_bundles(shard, world: int, rank: int, dist, device):
    """C2: every map's feature bundle from its owner to all ranks with ONE all-gather.  Each rank packs the
    bundles of its own maps back to back (mm3d_shard_pack) into a buffer padded to the largest rank's total;
    after the all-gather the other maps are handed to the library (mm3d_shard_unpack).  Returns (points,
    keypoints) per map."""
    import torch
    n = shard.n
    npts, nkp = shard.bundleSizes()
    if world == 1:
        return npts, nkp
    sizes = torch.from_numpy(np.stack([npts, nkp]).astype(np.int64))
    if device is not None:
        sizes = sizes.to(device)
    dist.all_reduce(sizes)                                 # every column has exactly one non-zero contributor
    npts, nkp = (sizes.cpu().numpy()[k].astype(np.uint64) for k in (0, 1))
    nbytes = [shard.bundleBytes(int(npts[i]), int(nkp[i])) for i in range(n)]
    offset, total = [0] * n, [0] * world
    for i in range(n):                                     # a map's place inside its owner's buffer: index order, 256-byte aligned
        o = map_owner(i, world)
        offset[i] = total[o]
        total[o] += (nbytes[i] + 255) // 256 * 256
    width = max(max(total), 256)
    # torch.empty, not zeros: a fill kernel would run on torch's current stream, unordered with the library's own
    # stream that shard.pack copies on; the padding between bundles is never read (unpack uses offset[i] and the sizes)
    buf = torch.empty(width, dtype=torch.uint8, device=device if device is not None else "cpu")
    for i in range(n):
        if map_owner(i, world) == rank and nbytes[i]:
            shard.pack(i, buf.data_ptr() + offset[i])
    parts = all_gather_bytes(buf, world, dist)
    if device is not None:
        torch.cuda.synchronize()
    shard.unpackMany([(i, parts[map_owner(i, world)].data_ptr() + offset[i], int(npts[i]), int(nkp[i]))
                      for i in range(n) if map_owner(i, world) != rank])
    return npts, nkp
