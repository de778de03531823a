"""Sharding of the map/pair loops of estimateMapsTransforms over one process per GPU.

The reference runs both loops sequentially in one thread (R/src/map_merging.cpp:212-242 per map,
:256-269 per pair).  Maps and pairs are independent units, so they are dealt round-robin; the only
exchange steps are (C2) a broadcast of each map's feature bundle from its owner and (C1) one
all-gather of the fixed-size pair records (RCCL over xGMI on the GPU box, gloo in the CPU tests)
before every rank solves the (tiny, host-side) pose graph.
"""
from __future__ import annotations

import numpy as np


def map_owner(i: int, world: int) -> int:
    return i % world


def pair_owner(p: int, world: int) -> int:
    return p % world


def live_pairs(n_maps: int, keypoint_counts) -> list:
    """Pairs (i < j) whose maps both have keypoints, in the reference's order (map_merging.cpp:246-254)."""
    return [(i, j) for i in range(n_maps - 1) for j in range(i + 1, n_maps)
            if keypoint_counts[i] > 0 and keypoint_counts[j] > 0]


def gather_pair_records(records: np.ndarray, world: int, rank: int, dist=None, device=None) -> np.ndarray:
    """All-gather the pair records; slot p is taken from rank pair_owner(p).  `records` is the local
    array (all slots present, only the owned ones meaningful).  Returns the merged array."""
    if world == 1:
        return records
    import torch
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(len(records), -1)
    buf = torch.from_numpy(raw.copy())
    if device is not None:
        buf = buf.to(device)
    allb = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(allb, buf)
    merged = records.copy()
    owners = np.arange(len(records)) % world
    for r in range(world):
        if r == rank:
            continue
        other = allb[r].cpu().numpy().reshape(-1).view(records.dtype)
        sel = owners == r
        merged[sel] = other[sel]
    return merged


def broadcast_sizes(sizes, owner: int, dist, device=None):
    import torch
    t = torch.tensor(list(sizes), dtype=torch.int64)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, owner)
    return [int(v) for v in t.cpu()]
