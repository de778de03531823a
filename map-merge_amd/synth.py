"""Seeded synthetic worlds and overlapping maps.

The reference ships no data (no .pcd in /root/reference); its harnesses
(R/src/map_merge_tool.cpp:22-38, R/src/registration_visualisation.cpp:34-47) load
user-provided PCD files.  This module generates the stand-in inputs used by the tests and by
bench.py: textured planar / box / cylinder scenes with known ground-truth SE(3) per map.

A map is returned in the layout the C ABI takes (include/mm3d.h): float32 xyz [N,3] and
uint8 rgb [N,3]; `pack_points` makes the 16-byte x,y,z,rgba records (rgba = 0xFFRRGGBB).
"""
from __future__ import annotations

import numpy as np

__all__ = ["World", "synth_world", "synth_map", "synth_maps", "lattice_map", "lattice_maps", "cached_maps", "window_overlap",
           "pack_points", "relative_gt", "POINT_DTYPE"]

POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])


class World:
    """Rectangular patches (o, u, v) and vertical cylinders, each with a base colour."""

    def __init__(self, extent, patches, cylinders, seed):
        self.extent = float(extent)
        self.patches = patches          # dict of arrays: o[P,3], u[P,3], v[P,3], col[P,3]
        self.cylinders = cylinders      # dict: c[C,3] (base centre), r[C], h[C], col[C,3]
        self.seed = seed


def _box_patches(c, size, yaw):
    """Five faces (no bottom) of a yaw-rotated box with base centre c and size (sx,sy,sz)."""
    sx, sy, sz = size
    cy, sn = np.cos(yaw), np.sin(yaw)
    ex = np.array([cy, sn, 0.0]) * sx
    ey = np.array([-sn, cy, 0.0]) * sy
    ez = np.array([0.0, 0.0, sz])
    o = c - 0.5 * ex - 0.5 * ey
    return [
        (o, ex, ez), (o + ey, ex, ez),          # two sides along x
        (o, ey, ez), (o + ex, ey, ez),          # two sides along y
        (o + ez, ex, ey),                        # top
    ]


def terrain_height(xy, seed):
    """Rolling ground: smooth height field (metres) so that flat regions still carry geometry."""
    p = np.concatenate([xy, np.zeros((len(xy), 1))], axis=1)
    return 0.45 * _value_noise(p, 3.1, seed + 11) + 0.18 * _value_noise(p, 1.3, seed + 12)


def synth_world(seed: int = 1234, extent: float = 120.0, object_density: float = 0.08) -> World:
    """Ground plane plus ~object_density objects per square metre inside [-extent/2, extent/2]^2."""
    rng = np.random.Generator(np.random.PCG64(seed))
    h = extent / 2.0
    plist = [(np.array([-h, -h, 0.0]), np.array([extent, 0, 0.0]), np.array([0, extent, 0.0]))]
    pcol = [np.array([120.0, 120.0, 120.0])]
    cyl_c, cyl_r, cyl_h, cyl_col = [], [], [], []
    n_obj = max(4, int(round(object_density * extent * extent)))
    for _ in range(n_obj):
        kind = rng.integers(0, 3)
        c = np.array([rng.uniform(-h, h), rng.uniform(-h, h), 0.0])
        col = rng.uniform(60.0, 200.0, size=3)
        if kind == 0:      # box
            size = (rng.uniform(1.0, 4.0), rng.uniform(1.0, 4.0), rng.uniform(0.8, 3.5))
            for p in _box_patches(c, size, rng.uniform(0, np.pi)):
                plist.append(p); pcol.append(col)
        elif kind == 1:    # wall (thin, tall)
            size = (rng.uniform(3.0, 9.0), 0.2, rng.uniform(2.0, 4.0))
            for p in _box_patches(c, size, rng.uniform(0, np.pi)):
                plist.append(p); pcol.append(col)
        else:              # cylinder
            cyl_c.append(c); cyl_r.append(rng.uniform(0.4, 1.5)); cyl_h.append(rng.uniform(1.0, 4.0))
            cyl_col.append(col)
    patches = {
        "o": np.array([p[0] for p in plist]), "u": np.array([p[1] for p in plist]),
        "v": np.array([p[2] for p in plist]), "col": np.array(pcol),
    }
    cylinders = {
        "c": np.array(cyl_c).reshape(-1, 3), "r": np.array(cyl_r), "h": np.array(cyl_h),
        "col": np.array(cyl_col).reshape(-1, 3),
    }
    return World(extent, patches, cylinders, seed)


def _value_noise(p, cell, seed):
    """Trilinear value noise in [-1, 1] with the given lattice spacing (hash-based, no tables)."""
    q = p / cell
    i0 = np.floor(q).astype(np.int64)
    f = q - i0
    f = f * f * (3.0 - 2.0 * f)
    out = np.zeros(len(p))
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                ix, iy, iz = i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz
                hsh = (ix * 73856093) ^ (iy * 19349663) ^ (iz * 83492791) ^ (seed * 2654435761)
                hsh = (hsh ^ (hsh >> 13)) * 1274126177
                hsh = hsh ^ (hsh >> 16)
                val = ((hsh & 0xFFFF).astype(np.float64) / 32767.5) - 1.0
                w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * \
                    (f[:, 2] if dz else 1 - f[:, 2])
                out += w * val
    return out


def _texture(pw, base_col, seed):
    """RGB texture: base colour modulated by multi-octave value noise (intensity std ~40/255)."""
    n = 70.0 * _value_noise(pw, 0.9, seed) + 45.0 * _value_noise(pw, 0.45, seed + 1) + \
        25.0 * _value_noise(pw, 0.22, seed + 2)
    tint = 20.0 * np.stack([_value_noise(pw, 1.7, seed + 3), _value_noise(pw, 1.7, seed + 4),
                            _value_noise(pw, 1.7, seed + 5)], axis=1)
    rgb = base_col + n[:, None] + tint
    return np.clip(np.rint(rgb), 0, 255).astype(np.uint8)


def _rot(yaw, pitch, roll):
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def window_for(n_points: int) -> float:
    """Window side so that raw density stays ~constant (60 m at 500k points)."""
    return 60.0 * float(np.sqrt(n_points / 500000.0))


def synth_map(world: World, index: int, n_points: int, n_maps: int = 16, seed: int | None = None,
              window: float | None = None, noise: float = 0.01, overlap_step: float = 0.5):
    """Sample one map.  Returns (xyz float32 [N,3], rgb uint8 [N,3], T_gt float64 4x4) where T_gt
    maps WORLD coordinates to this map's frame."""
    seed = 1000 + index if seed is None else seed
    rng = np.random.Generator(np.random.PCG64(seed))
    w = window_for(n_points) if window is None else float(window)
    # window centres walk a loop so consecutive maps overlap ~50 % and the loop closes
    loop_r = overlap_step * w * max(n_maps, 2) / (2.0 * np.pi)
    max_r = max(world.extent / 2.0 - w / 2.0 - 1.0, 0.0)
    loop_r = min(loop_r, max_r)
    ang = 2.0 * np.pi * index / max(n_maps, 1)
    centre = np.array([loop_r * np.cos(ang), loop_r * np.sin(ang), 0.0])
    lo, hi = centre[:2] - w / 2.0, centre[:2] + w / 2.0

    P, C = world.patches, world.cylinders
    # area of each primitive (whole primitive; points outside the window are rejected)
    pa = np.linalg.norm(np.cross(P["u"], P["v"]), axis=1)
    # clip the ground plane analytically (it is huge): replace by the window rectangle
    o = P["o"].copy(); u = P["u"].copy(); v = P["v"].copy()
    o[0] = np.array([lo[0], lo[1], 0.0]); u[0] = np.array([w, 0, 0.0]); v[0] = np.array([0, w, 0.0])
    pa[0] = w * w
    # drop primitives whose bounding box misses the window
    corners = np.stack([o, o + u, o + v, o + u + v], axis=1)[:, :, :2]
    keep = (corners.max(axis=1) >= lo).all(axis=1) & (corners.min(axis=1) <= hi).all(axis=1)
    pa = np.where(keep, pa, 0.0)
    if len(C["r"]):
        ca = 2.0 * np.pi * C["r"] * C["h"]
        ckeep = ((C["c"][:, :2] + C["r"][:, None]) >= lo).all(axis=1) & \
                ((C["c"][:, :2] - C["r"][:, None]) <= hi).all(axis=1)
        ca = np.where(ckeep, ca, 0.0)
    else:
        ca = np.zeros(0)
    areas = np.concatenate([pa, ca])
    prob = areas / areas.sum()
    pts, cols = [], []
    need = n_points
    while need > 0:
        m = int(need * 1.3) + 64
        prim = rng.choice(len(areas), size=m, p=prob)
        a, b = rng.random(m), rng.random(m)
        pw = np.empty((m, 3)); bc = np.empty((m, 3))
        isp = prim < len(pa)
        ip = prim[isp]
        pw[isp] = o[ip] + a[isp, None] * u[ip] + b[isp, None] * v[ip]
        isg = isp & (prim == 0)                      # ground patch: add the rolling terrain
        pw[isg, 2] = terrain_height(pw[isg, :2], world.seed)
        bc[isp] = P["col"][ip]
        ic = prim[~isp] - len(pa)
        th = 2.0 * np.pi * a[~isp]
        pw[~isp] = C["c"][ic] + np.stack([C["r"][ic] * np.cos(th), C["r"][ic] * np.sin(th),
                                           C["h"][ic] * b[~isp]], axis=1)
        bc[~isp] = C["col"][ic]
        inside = (pw[:, :2] >= lo).all(axis=1) & (pw[:, :2] <= hi).all(axis=1)
        pw, bc = pw[inside][:need], bc[inside][:need]
        pts.append(pw); cols.append(_texture(pw, bc, world.seed))
        need -= len(pw)
    pw = np.concatenate(pts); rgb = np.concatenate(cols)
    pw = pw + rng.normal(0.0, noise, size=pw.shape)
    # ground-truth pose: map = R (world - centre) + t
    yaw = rng.uniform(-np.pi, np.pi)
    pitch, roll = np.deg2rad(rng.uniform(-5, 5, size=2))
    R = _rot(yaw, float(pitch), float(roll))
    t = rng.uniform(-10.0, 10.0, size=3) * np.array([1.0, 1.0, 0.1])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t - R @ centre
    pm = pw @ R.T + T[:3, 3]
    perm = rng.permutation(len(pm))
    return pm[perm].astype(np.float32), rgb[perm], T


def synth_maps(n_maps: int, n_points: int, world_seed: int = 1234, **kw):
    """World sized for the loop of n_maps windows, plus the n_maps maps and their T_gt."""
    w = kw.get("window") or window_for(n_points)
    loop_r = kw.get("overlap_step", 0.5) * w * max(n_maps, 2) / (2.0 * np.pi)
    extent = max(2.0 * (loop_r + w / 2.0 + 2.0), w + 4.0)
    world = synth_world(world_seed, extent=extent)
    maps = [synth_map(world, i, n_points, n_maps=n_maps, **kw) for i in range(n_maps)]
    return world, maps


def cached_maps(n_maps: int, n_points: int, cache_dir: str | None = "/tmp", family: str = "independent", **kw):
    """synth_maps (family 'independent') or lattice_maps (family 'lattice') as packed records plus the ground-truth
    poses, kept in `cache_dir` between runs (generating 16 x 500 000 points takes ~20 s of CPU).
    Returns (list of POINT_DTYPE arrays, list of 4x4 T_gt, window)."""
    import os
    w = kw.get("window") or window_for(n_points)
    tag = "_".join([family] * (family != "independent") + [f"{k}{v}" for k, v in sorted(kw.items())])
    path = os.path.join(cache_dir, f"mm3d_synth_{n_maps}x{n_points}{('_' + tag) if tag else ''}.npz") if cache_dir else None
    if path and os.path.exists(path):
        try:
            z = np.load(path)
            if int(z["n_maps"]) == n_maps:
                return [z[f"pts{i}"] for i in range(n_maps)], [z["T"][i] for i in range(n_maps)], float(w)
        except Exception:
            pass                                           # unreadable cache: regenerate
    _, maps = (lattice_maps if family == "lattice" else synth_maps)(n_maps, n_points, **kw)
    packed = [pack_points(x, c) for x, c, _ in maps]
    Ts = [T for _, _, T in maps]
    if path:
        try:                                               # several ranks may get here at once: write aside, rename atomically
            tmp = f"{path}.{os.getpid()}.tmp.npz"
            np.savez(tmp, n_maps=n_maps, T=np.stack(Ts), **{f"pts{i}": a for i, a in enumerate(packed)})
            os.replace(tmp, path)
        except Exception:
            pass
    return packed, Ts, float(w)


def window_overlap(n_maps: int, n_points: int, i: int, j: int, overlap_step: float = 0.5, window: float | None = None) -> float:
    """Fraction of map i's window that map j's window covers (both are axis-aligned squares of the same side
    whose centres walk the loop of synth_map); 0 = disjoint."""
    w = window_for(n_points) if window is None else float(window)
    loop_r = overlap_step * w * max(n_maps, 2) / (2.0 * np.pi)
    a = [2.0 * np.pi * k / max(n_maps, 1) for k in (i, j)]
    d = np.abs(np.array([loop_r * (np.cos(a[0]) - np.cos(a[1])), loop_r * (np.sin(a[0]) - np.sin(a[1]))]))
    return float(max(0.0, w - d[0]) * max(0.0, w - d[1]) / (w * w))


def pack_points(xyz: np.ndarray, rgb: np.ndarray) -> np.ndarray:
    """[N] records x,y,z,rgba (PCL PointXYZRGB payload: rgba = a<<24 | r<<16 | g<<8 | b, a=255)."""
    out = np.empty(len(xyz), dtype=POINT_DTYPE)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    r, g, b = (rgb[:, k].astype(np.uint32) for k in range(3))
    out["rgba"] = (np.uint32(255) << 24) | (r << 16) | (g << 8) | b
    return out


def relative_gt(T_i: np.ndarray, T_j: np.ndarray) -> np.ndarray:
    """Ground-truth transform taking map-i coordinates to map-j coordinates."""
    return T_j @ np.linalg.inv(T_i)


# ---- a second scene family: the same world seen through a world-anchored sampling lattice ------------------
# synth_map draws every map's points independently, so two maps never hold the same surface sample: SIFT keypoints
# (intensity extrema of the sampled cloud) are barely repeatable between them, and FPFH + SAC-IA -- 500 hypotheses,
# each three random picks among the 10 nearest descriptors -- does not find the basin (tests/test_gpu_baseline_configs.py).
# Here the WORLD is sampled once, on a jittered lattice over every primitive (spacing `pitch`, jitter and sensor noise
# hashed from the lattice node), and a map is the part of that sample inside its window, moved into the map's frame:
# overlapping maps share their surface samples the way two passes of a survey-grade scanner over a static scene nearly
# do.  Nothing about the registration problem is given away (poses, order and noise are the generator's secret), but
# keypoints and descriptors repeat, which is what the reference's pipeline needs in order to converge.
def _hash01(ix, iy, salt):
    h = (ix.astype(np.int64) * 73856093) ^ (iy.astype(np.int64) * 19349663) ^ (int(salt) * 83492791)
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    return (h & 0xFFFFFF).astype(np.float64) / float(0x1000000)


def _lattice_patch(o, u, v, pitch, salt, lo, hi):
    """Jittered lattice over the parallelogram o + a u + b v, clipped to the window [lo, hi] (xy)."""
    lu, lv = np.linalg.norm(u), np.linalg.norm(v)
    nu, nv = max(int(lu / pitch), 1), max(int(lv / pitch), 1)
    # only the lattice rows / columns that can reach the window (the ground patch is huge)
    eu, ev = u / lu, v / lv
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    iu, iv = iu.ravel(), iv.ravel()
    a = (iu + 0.15 + 0.7 * _hash01(iu, iv, salt)) * pitch
    b = (iv + 0.15 + 0.7 * _hash01(iu, iv, salt + 1)) * pitch
    p = o + a[:, None] * eu + b[:, None] * ev
    keep = (p[:, 0] >= lo[0]) & (p[:, 0] <= hi[0]) & (p[:, 1] >= lo[1]) & (p[:, 1] <= hi[1])
    return p[keep], iu[keep], iv[keep]


def lattice_map(world: World, index: int, n_points: int, n_maps: int = 16, window: float | None = None, noise: float = 0.01,
                overlap_step: float = 0.5, pitch: float | None = None):
    """One map of the lattice scene family: (xyz float32 [N,3], rgb uint8 [N,3], T_gt) like synth_map.  N is close to
    n_points (the lattice inside the window, thinned by a hashed threshold to at most n_points)."""
    rng = np.random.Generator(np.random.PCG64(1000 + index))
    w = window_for(n_points) if window is None else float(window)
    loop_r = overlap_step * w * max(n_maps, 2) / (2.0 * np.pi)
    loop_r = min(loop_r, max(world.extent / 2.0 - w / 2.0 - 1.0, 0.0))
    ang = 2.0 * np.pi * index / max(n_maps, 1)
    centre = np.array([loop_r * np.cos(ang), loop_r * np.sin(ang), 0.0])
    lo, hi = centre[:2] - w / 2.0, centre[:2] + w / 2.0
    # pitch: the window's ground alone gives n_points / 1.15 samples, the objects the rest (thinned below if more)
    pitch = float(np.sqrt(w * w * 1.15 / n_points)) if pitch is None else float(pitch)
    P, C = world.patches, world.cylinders
    pts, cols, keys = [], [], []
    for k in range(len(P["o"])):
        o, u, v = P["o"][k], P["u"][k], P["v"][k]
        if k == 0:
            # the ground: lattice anchored at the world's corner, only the nodes near the window are generated
            i0, i1 = int((lo[0] - o[0]) / pitch) - 1, int((hi[0] - o[0]) / pitch) + 2
            j0, j1 = int((lo[1] - o[1]) / pitch) - 1, int((hi[1] - o[1]) / pitch) + 2
            iu, iv = np.meshgrid(np.arange(max(i0, 0), i1), np.arange(max(j0, 0), j1), indexing="ij")
            iu, iv = iu.ravel(), iv.ravel()
            x = o[0] + (iu + 0.15 + 0.7 * _hash01(iu, iv, 17)) * pitch
            y = o[1] + (iv + 0.15 + 0.7 * _hash01(iu, iv, 18)) * pitch
            keep = (x >= lo[0]) & (x <= hi[0]) & (y >= lo[1]) & (y <= hi[1])
            p = np.stack([x[keep], y[keep], np.zeros(keep.sum())], axis=1)
            p[:, 2] = terrain_height(p[:, :2], world.seed)
            iu, iv = iu[keep], iv[keep]
        else:
            c4 = np.stack([o, o + u, o + v, o + u + v])[:, :2]
            if (c4.max(axis=0) < lo).any() or (c4.min(axis=0) > hi).any():
                continue
            p, iu, iv = _lattice_patch(o, u, v, pitch, 100 + 2 * k, lo, hi)
        if len(p) == 0:
            continue
        pts.append(p); cols.append(np.broadcast_to(P["col"][k], (len(p), 3))); keys.append(_hash01(iu, iv, 7000 + k))
    for k in range(len(C["r"])):
        c, r, h = C["c"][k], C["r"][k], C["h"][k]
        if ((c[:2] + r) < lo).any() or ((c[:2] - r) > hi).any():
            continue
        nth, nh = max(int(2.0 * np.pi * r / pitch), 3), max(int(h / pitch), 1)
        it, ih = np.meshgrid(np.arange(nth), np.arange(nh), indexing="ij")
        it, ih = it.ravel(), ih.ravel()
        th = 2.0 * np.pi * (it + 0.15 + 0.7 * _hash01(it, ih, 300 + 2 * k)) / nth
        z = (ih + 0.15 + 0.7 * _hash01(it, ih, 301 + 2 * k)) * (h / nh)
        p = c + np.stack([r * np.cos(th), r * np.sin(th), z], axis=1)
        keep = (p[:, 0] >= lo[0]) & (p[:, 0] <= hi[0]) & (p[:, 1] >= lo[1]) & (p[:, 1] <= hi[1])
        if keep.any():
            pts.append(p[keep]); cols.append(np.broadcast_to(C["col"][k], (keep.sum(), 3))); keys.append(_hash01(it[keep], ih[keep], 9000 + k))
    pw = np.concatenate(pts); bc = np.concatenate(cols).astype(np.float64); key = np.concatenate(keys)
    if len(pw) > n_points:                                 # thin by the nodes' own hash: the same nodes go in every map
        thr = np.partition(key, n_points - 1)[n_points - 1]
        sel = key <= thr
        pw, bc, key = pw[sel][:n_points], bc[sel][:n_points], key[sel][:n_points]
    rgb = _texture(pw, bc, world.seed)
    # sensor noise hashed from the (quantised) world position: the same in every map that sees the sample
    q = np.floor(pw * 1000.0).astype(np.int64)
    nz = np.stack([_hash01(q[:, 0], q[:, 1] + 7 * q[:, 2], 31 + a) for a in range(3)], axis=1) - 0.5
    pw = pw + nz * (noise * np.sqrt(12.0))
    yaw = rng.uniform(-np.pi, np.pi)
    pitch_a, roll = np.deg2rad(rng.uniform(-5, 5, size=2))
    R = _rot(yaw, float(pitch_a), float(roll))
    t = rng.uniform(-10.0, 10.0, size=3) * np.array([1.0, 1.0, 0.1])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t - R @ centre
    pm = pw @ R.T + T[:3, 3]
    perm = rng.permutation(len(pm))
    return pm[perm].astype(np.float32), rgb[perm], T


def lattice_maps(n_maps: int, n_points: int, world_seed: int = 1234, object_density: float = 0.08, **kw):
    """The lattice scene family: world + maps + T_gt, like synth_maps."""
    w = kw.get("window") or window_for(n_points)
    loop_r = kw.get("overlap_step", 0.5) * w * max(n_maps, 2) / (2.0 * np.pi)
    extent = max(2.0 * (loop_r + w / 2.0 + 2.0), w + 4.0)
    world = synth_world(world_seed, extent=extent, object_density=object_density)
    return world, [lattice_map(world, i, n_points, n_maps=n_maps, **kw) for i in range(n_maps)]
