#!/usr/bin/env python3
"""Regenerates tests/golden/pair_6k.npz: inputs and the CPU oracle's outputs for one small pair.

The reference (C++ on PCL) cannot be built or imported in this image, so these vectors come from
oracle/ (the restatement), not from the reference itself: they pin the oracle against drift and
against the host CPU it runs on, they do not pin it to PCL (oracle/mm3d_oracle.h: "parity unpinned").
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

po = ge.load_oracle()
ge.load()
from map_merge_amd import synth  # noqa: E402

N_RAW, STEP = 6000, 0.35
world, maps = synth.synth_maps(2, N_RAW, overlap_step=STEP)
out = {"n_raw": N_RAW, "overlap_step": STEP}
raws = []
for i, (x, c, T) in enumerate(maps):
    raw = synth.pack_points(x, c)
    raws.append(raw)
    filt = po.remove_outliers(po.downsample(raw, 0.1), 0.8, 50)
    nrm = po.normals(filt, 0.6)
    kp, _ = po.keypoints_sift(filt, 0.1, 3, 3, 5.0)
    kp, desc = po.descriptors_fpfh(filt, nrm, kp, 0.8)
    out[f"raw{i}"] = raw.view(np.uint32).reshape(-1, 4)
    out[f"filt{i}"] = filt.view(np.uint32).reshape(-1, 4)
    out[f"nrm{i}"] = np.stack([nrm["nx"], nrm["ny"], nrm["nz"], nrm["curvature"]], 1)
    out[f"kp{i}"] = np.stack([kp["x"], kp["y"], kp["z"]], 1)
    out[f"desc{i}"] = desc
    out[f"T_gt{i}"] = T
p = po.params_default()
p.descriptor_type, p.estimation_method = 2, 1
po.srand(1)
T, pairs = po.estimate_maps_transforms(raws, p)
out["T_global"] = np.stack(T)
out["pair_transform"] = pairs["transform"]
out["pair_confidence"] = pairs["confidence"]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_6k.npz"), **out)
print("wrote pair_6k.npz:", {k: getattr(v, "shape", v) for k, v in out.items()})

# features_6k.npz: the other keypoint / descriptor types on map 0 of the same pair (first 48 keypoints)
filt = po.remove_outliers(po.downsample(raws[0], 0.1), 0.8, 50)
nrm = po.normals(filt, 0.6)
kp_sift, _ = po.keypoints_sift(filt, 0.1, 3, 3, 5.0)
kp48 = kp_sift[:48].copy()
feat = {"n_raw": N_RAW, "overlap_step": STEP}
hk, hidx, hresp = po.keypoints_harris(filt, nrm, 0.002, 0.6)
feat["harris_kp"] = np.stack([hk["x"], hk["y"], hk["z"]], 1)
feat["harris_idx"] = hidx
feat["harris_response"] = hresp
for name, fn in (("pfh", po.descriptors_pfh), ("pfhrgb", po.descriptors_pfhrgb), ("shot", po.descriptors_shot)):
    k, d = fn(filt, nrm, kp48, 0.8)
    feat[name + "_kp"] = np.stack([k["x"], k["y"], k["z"]], 1)
    feat[name] = d
_, feat["shot_rf"] = po.shot_raw(filt, nrm, kp48, 0.8)
for m, dt in ((1, 4), (0, 1)):   # SHOT + SAC_IA, PFHRGB + MATCHING through the whole pipeline
    p = po.params_default()
    p.descriptor_type, p.estimation_method = dt, m
    po.srand(1)
    T, pairs = po.estimate_maps_transforms(raws, p)
    feat[f"pair_transform_d{dt}_m{m}"] = pairs["transform"]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "features_6k.npz"), **feat)
print("wrote features_6k.npz:", {k: getattr(v, "shape", v) for k, v in feat.items()})
