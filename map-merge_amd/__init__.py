"""map_merge_3d on MI355X: Python host mirror of the reference's C++ API over libmm3d.so.

The product is the C-ABI library (include/mm3d.h).  This module is the thin host side used by the
tests and bench.py; it mirrors the names and argument meaning of the reference's free functions
(R/include/map_merge_3d/features.h:34-98, matching.h:26-152, map_merging.h:28-101) so the parity
tests read like the reference's own harnesses.  There is NO CPU fallback: if libmm3d.so is missing
or no GPU is visible, calls raise.

The directory is named `map-merge_amd`; import it as `map_merge_amd` through `load()` in
__graft_entry__.py / tests/conftest.py (a hyphen is not importable).
"""
from __future__ import annotations

import ctypes as C
import enum
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM3D_LIB: an instrumentation build of the same library (scripts/nn_stats.py, scripts/sn_stats.py)
LIB_PATH = os.environ.get("MM3D_LIB") or os.path.join(_HERE, "libmm3d.so")

POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])
NORMAL = np.dtype([("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("curvature", "<f4")])
CORR = np.dtype([("index_query", "<i4"), ("index_match", "<i4"), ("distance", "<f4")])
PAIR = np.dtype([("source_idx", "<u8"), ("target_idx", "<u8"), ("transform", "<f4", (16,)),
                 ("confidence", "<f8"), ("icp_iterations", "<i4"), ("n_correspondences", "<i4"),
                 ("n_inliers", "<i4"), ("icp_correspondences", "<i4")])


class Mm3dError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"mm3d status {status}: {msg}")
        self.status = status


class Descriptor(enum.IntEnum):      # features.h:20-24
    PFH = 0
    PFHRGB = 1
    FPFH = 2
    RSD = 3
    SHOT = 4
    SC3D = 5


class Keypoint(enum.IntEnum):        # features.h:49
    SIFT = 0
    HARRIS = 1


class EstimationMethod(enum.IntEnum):  # matching.h:103
    MATCHING = 0
    SAC_IA = 1


class MapMergingParams(C.Structure):
    """R/include/map_merge_3d/map_merging.h:28-44, field for field."""
    _fields_ = [("resolution", C.c_double), ("descriptor_radius", C.c_double),
                ("outliers_min_neighbours", C.c_int), ("normal_radius", C.c_double),
                ("keypoint_type", C.c_int), ("keypoint_threshold", C.c_double),
                ("descriptor_type", C.c_int), ("estimation_method", C.c_int),
                ("refine_transform", C.c_int), ("inlier_threshold", C.c_double),
                ("max_correspondence_distance", C.c_double), ("max_iterations", C.c_int),
                ("matching_k", C.c_uint64), ("transform_epsilon", C.c_double),
                ("confidence_threshold", C.c_double), ("output_resolution", C.c_double)]

    def __init__(self, **kw):
        super().__init__()
        lib().mm3d_params_default(C.byref(self))
        for k, v in kw.items():
            setattr(self, k, int(v) if isinstance(v, enum.IntEnum) else v)

    @staticmethod
    def fromCommandLine(argv):
        """MapMergingParams::fromCommandLine (R/src/map_merging.cpp:10-54); argv[0] is the program."""
        p = MapMergingParams()
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        st = lib().mm3d_params_from_command_line(len(argv), arr, C.byref(p))
        if st != 0:
            # enums::from_string throws std::runtime_error on a bad value (enum.h:58-60)
            raise RuntimeError("from_string: invalid value for enum")
        return p

    def __str__(self):
        n = lib().mm3d_params_to_string(C.byref(self), None, 0)
        buf = C.create_string_buffer(n)
        lib().mm3d_params_to_string(C.byref(self), buf, n)
        return buf.value.decode()


class _View(C.Structure):
    _fields_ = [("points", C.c_void_p), ("n", C.c_size_t), ("stride", C.c_size_t), ("rgba_offset", C.c_size_t)]


_LIB = None


def build(force: bool = False) -> str:
    """Compile libmm3d.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call([os.path.join(_HERE, "build.sh")])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise Mm3dError(-2, f"{LIB_PATH} is missing: run map-merge_amd/build.sh (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        L.mm3d_last_error.restype = C.c_char_p
        for f in ("mm3d_cloud_size", "mm3d_normals_size", "mm3d_desc_size", "mm3d_params_to_string"):
            getattr(L, f).restype = C.c_size_t
        for f in ("mm3d_descriptor_name", "mm3d_descriptor_field_name", "mm3d_keypoint_name",
                  "mm3d_estimation_method_name"):
            getattr(L, f).restype = C.c_char_p
        for f in ("mm3d_map_points", "mm3d_map_keypoints", "mm3d_map_descriptors"):
            getattr(L, f).restype = C.c_void_p
        _LIB = L
    return _LIB


def _T(T):
    """4x4 (row-major numpy) -> column-major float[16] as the ABI takes it."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T.reshape(16))


def _Tout(a):
    return np.asarray(a, dtype=np.float32).reshape(4, 4).T.copy()


def sift_cert_stats(reset=False):
    """Process-wide counters of the certified SIFT decision (mm3d_debug_sift_cert_stats): octaves, points, exact-path points,
    open points, octaves sent back, bound violations, still-open points, unstaged items."""
    out = (C.c_longlong * 8)()
    lib().mm3d_debug_sift_cert_stats(out, 1 if reset else 0)
    return list(out)


def sacia_stats(reset=False, collect=-1):
    """Process-wide counters of SAC-IA's certified pick (mm3d_debug_sacia_stats): pairs scored, pairs decided without a float
    chain, candidate hypotheses the intervals left, chains run.  collect = 1 / 0 switches the collection on / off."""
    out = (C.c_longlong * 4)()
    lib().mm3d_debug_sacia_stats(out, 1 if reset else 0, int(collect))
    return list(out)


class Context:
    """One registration engine on one GPU (mm3d_ctx) -- or, with `devices`, on a list of GPUs of this one process
    (mm3d_create_devices): estimateMapsTransforms then shards over them inside the library and gathers the pair
    records through RCCL; every other call works on the first device of the list."""

    def __init__(self, device: int = 0, devices=None):
        self._h = C.c_void_p()
        if devices is not None:
            devices = [int(d) for d in devices]
            arr = (C.c_int * max(len(devices), 1))(*devices)
            st = lib().mm3d_create_devices(arr, len(devices), C.byref(self._h))
            if st != 0:
                why = (lib().mm3d_last_error(None) or b"").decode()
                raise Mm3dError(st, f"mm3d_create_devices({devices}) failed: {why or 'bad list, a device twice, no such device, or RCCL could not create its communicators'} "
                                    "(there is no CPU path)")
            device = devices[0]
        else:
            st = lib().mm3d_create(int(device), C.byref(self._h))
            if st != 0:
                raise Mm3dError(st, "mm3d_create failed: no usable MI355X/HIP device (there is no CPU path)")
        self.device = device

    @property
    def devices(self):
        return [lib().mm3d_device_at(self._h, i) for i in range(lib().mm3d_device_count(self._h))]

    @property
    def uses_rccl(self) -> bool:
        return bool(lib().mm3d_devices_use_rccl(self._h))

    def lastRunDeviceSeconds(self):
        """(exchange_s, pairs_s, gather_s) of the most recent estimateMapsTransforms on a device list."""
        a, b, g = C.c_double(), C.c_double(), C.c_double()
        self._ck(lib().mm3d_last_run_device_seconds(self._h, C.byref(a), C.byref(b), C.byref(g)))
        return a.value, b.value, g.value

    def close(self):
        if self._h:
            lib().mm3d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != 0:
            raise Mm3dError(st, (lib().mm3d_last_error(self._h) or b"").decode())

    def srand(self, seed: int):
        lib().mm3d_srand(self._h, C.c_uint(seed))

    def setStreams(self, n: int):
        """mm3d_set_streams: HIP streams estimateMapsTransforms deals its two loops to (results are
        bit-identical for every setting)."""
        self._ck(lib().mm3d_set_streams(self._h, int(n)))

    def synchronize(self):
        self._ck(lib().mm3d_synchronize(self._h))

    # ---- objects -------------------------------------------------------------------------
    def cloud(self, pts) -> "Cloud":
        """Host numpy records (POINT dtype) -> device cloud."""
        pts = np.ascontiguousarray(pts, dtype=POINT)
        h = C.c_void_p()
        self._ck(lib().mm3d_cloud_create(self._h, pts.ctypes.data_as(C.c_void_p), C.c_size_t(len(pts)),
                                         C.c_size_t(16), C.c_size_t(12), C.byref(h)))
        return Cloud(self, h)

    def cloud_from_ptr(self, ptr: int, n: int, stride: int = 16, rgba_offset: int = 12) -> "Cloud":
        """Host or device (HBM) address, e.g. a torch tensor's data_ptr()."""
        h = C.c_void_p()
        self._ck(lib().mm3d_cloud_create(self._h, C.c_void_p(ptr), C.c_size_t(n), C.c_size_t(stride),
                                         C.c_size_t(rgba_offset), C.byref(h)))
        return Cloud(self, h)

    def normals(self, nrm) -> "Normals":
        nrm = np.ascontiguousarray(nrm, dtype=NORMAL)
        h = C.c_void_p()
        self._ck(lib().mm3d_normals_create(self._h, nrm.ctypes.data_as(C.c_void_p), C.c_size_t(len(nrm)),
                                           C.c_size_t(16), C.byref(h)))
        return Normals(self, h)

    def descriptors(self, data, descriptor=Descriptor.FPFH) -> "Descriptors":
        data = np.ascontiguousarray(data, dtype=np.float32)
        h = C.c_void_p()
        self._ck(lib().mm3d_desc_create(self._h, data.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)),
                                        int(descriptor), C.byref(h)))
        return Descriptors(self, h)

    # ---- features.h ------------------------------------------------------------------------
    def downSample(self, cloud: "Cloud", resolution: float) -> "Cloud":
        h = C.c_void_p()
        self._ck(lib().mm3d_downsample(self._h, cloud._h, C.c_double(resolution), C.byref(h)))
        return Cloud(self, h)

    def removeOutliers(self, cloud: "Cloud", radius: float, min_neighbours: int) -> "Cloud":
        h = C.c_void_p()
        self._ck(lib().mm3d_remove_outliers(self._h, cloud._h, C.c_double(radius), int(min_neighbours), C.byref(h)))
        return Cloud(self, h)

    def computeSurfaceNormals(self, cloud: "Cloud", radius: float) -> "Normals":
        h = C.c_void_p()
        self._ck(lib().mm3d_compute_normals(self._h, cloud._h, C.c_double(radius), C.byref(h)))
        return Normals(self, h)

    def detectKeypoints(self, points: "Cloud", normals, type, threshold: float, radius: float,
                        resolution: float) -> "Cloud":
        h = C.c_void_p()
        self._ck(lib().mm3d_detect_keypoints(self._h, points._h, normals._h if normals is not None else None,
                                             int(type), C.c_double(threshold), C.c_double(radius),
                                             C.c_double(resolution), C.byref(h)))
        return Cloud(self, h)

    def siftCertOctave(self, points: "Cloud", min_scale: float, octave: int):
        """Test hook (mm3d_debug_sift_cert_octave): the certified SIFT pass of one octave -- (val[n, 5], bound[n, 5]) with
        bound >= |the CPU path's float DoG - val|, or None when the octave does not exist."""
        n = C.c_size_t(0)
        self._ck(lib().mm3d_debug_sift_cert_octave(self._h, points._h, C.c_double(min_scale), int(octave), None, None, C.c_size_t(0), C.byref(n)))
        if n.value == 0:
            return None
        val = np.empty((n.value, 5), dtype=np.float32)
        bound = np.empty((n.value, 5), dtype=np.float32)
        self._ck(lib().mm3d_debug_sift_cert_octave(self._h, points._h, C.c_double(min_scale), int(octave), val.ctypes.data_as(C.c_void_p),
                                                   bound.ctypes.data_as(C.c_void_p), C.c_size_t(n.value), C.byref(n)))
        return val, bound

    def harrisResponse(self, points: "Cloud", normals: "Normals", radius: float) -> np.ndarray:
        """HarrisKeypoint3D::responseHarris of every point (what detectKeypoints(HARRIS) thresholds)."""
        out = np.zeros(max(len(points), 1), dtype=np.float32)
        self._ck(lib().mm3d_harris_response(self._h, points._h, normals._h, C.c_double(radius), out.ctypes.data_as(C.c_void_p)))
        return out[:len(points)]

    def computeLocalDescriptors(self, points: "Cloud", normals: "Normals", keypoints: "Cloud", descriptor,
                                feature_radius: float) -> "Descriptors":
        """Prunes `keypoints` in place like the reference (features.h:72-74)."""
        h = C.c_void_p()
        self._ck(lib().mm3d_compute_descriptors(self._h, points._h, normals._h, keypoints._h, int(descriptor),
                                                C.c_double(feature_radius), C.byref(h)))
        return Descriptors(self, h)

    # ---- matching.h ------------------------------------------------------------------------
    def findFeatureCorrespondences(self, source: "Descriptors", target: "Descriptors", k: int = 5):
        n = C.c_size_t()
        self._ck(lib().mm3d_find_correspondences(self._h, source._h, target._h, C.c_size_t(k), None, C.c_size_t(0),
                                                 C.byref(n)))
        out = np.empty(max(n.value, 1), dtype=CORR)
        self._ck(lib().mm3d_find_correspondences(self._h, source._h, target._h, C.c_size_t(k),
                                                 out.ctypes.data_as(C.c_void_p), C.c_size_t(len(out)), C.byref(n)))
        return out[:n.value].copy()

    def estimateTransformFromCorrespondences(self, source_keypoints, target_keypoints, correspondences,
                                             inlier_threshold: float):
        corr = np.ascontiguousarray(correspondences, dtype=CORR)
        T = np.zeros(16, dtype=np.float32)
        inl = np.empty(max(len(corr), 1), dtype=CORR)
        n = C.c_size_t()
        self._ck(lib().mm3d_estimate_transform_from_correspondences(
            self._h, source_keypoints._h, target_keypoints._h, corr.ctypes.data_as(C.c_void_p), C.c_size_t(len(corr)),
            C.c_double(inlier_threshold), T.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p),
            C.c_size_t(len(inl)), C.byref(n)))
        return _Tout(T), inl[:n.value].copy()

    def estimateTransformFromDescriptorsSets(self, source_keypoints, source_descriptors, target_keypoints,
                                             target_descriptors, min_sample_distance, max_correspondence_distance,
                                             max_iterations):
        T = np.zeros(16, dtype=np.float32)
        self._ck(lib().mm3d_estimate_transform_from_descriptors(
            self._h, source_keypoints._h, source_descriptors._h, target_keypoints._h, target_descriptors._h,
            C.c_double(min_sample_distance), C.c_double(max_correspondence_distance), int(max_iterations),
            T.ctypes.data_as(C.c_void_p)))
        return _Tout(T)

    def estimateTransformICP(self, source_points, target_points, initial_guess, max_correspondence_distance,
                             outlier_rejection_threshold, max_iterations=100, transformation_epsilon=0.0):
        g = _T(initial_guess)
        T = np.zeros(16, dtype=np.float32)
        self._ck(lib().mm3d_estimate_transform_icp(
            self._h, source_points._h, target_points._h, g.ctypes.data_as(C.c_void_p),
            C.c_double(max_correspondence_distance), C.c_double(outlier_rejection_threshold), int(max_iterations),
            C.c_double(transformation_epsilon), T.ctypes.data_as(C.c_void_p)))
        self.last_icp_iterations = lib().mm3d_last_icp_iterations(self._h)
        return _Tout(T)

    def estimateTransform(self, source_points, source_keypoints, source_descriptors, target_points,
                          target_keypoints, target_descriptors, method, refine, inlier_threshold,
                          max_correspondence_distance, max_iterations, matching_k, transform_epsilon):
        T = np.zeros(16, dtype=np.float32)
        self._ck(lib().mm3d_estimate_transform(
            self._h, source_points._h, source_keypoints._h, source_descriptors._h, target_points._h,
            target_keypoints._h, target_descriptors._h, int(method), int(bool(refine)), C.c_double(inlier_threshold),
            C.c_double(max_correspondence_distance), int(max_iterations), C.c_size_t(matching_k),
            C.c_double(transform_epsilon), T.ctypes.data_as(C.c_void_p)))
        return _Tout(T)

    def transformScore(self, source_points, target_points, transform, max_distance) -> float:
        t = _T(transform)
        s = C.c_double()
        self._ck(lib().mm3d_transform_score(self._h, source_points._h, target_points._h, t.ctypes.data_as(C.c_void_p),
                                            C.c_double(max_distance), C.byref(s)))
        return s.value

    # ---- map_merging.h ---------------------------------------------------------------------
    def estimateMapsTransforms(self, clouds, params: MapMergingParams, return_pairs: bool = False):
        """clouds: list of host numpy POINT arrays (or (ptr, n) tuples for HBM-resident inputs)."""
        n = len(clouds)
        keep = []
        views = (_View * max(n, 1))()
        for i, c in enumerate(clouds):
            if isinstance(c, tuple):                      # (ptr, n) packed 16-byte records, or (ptr, n, stride, rgba_offset)
                views[i] = _View(C.c_void_p(c[0]), c[1], c[2] if len(c) > 2 else 16, c[3] if len(c) > 3 else 12)
            else:
                a = np.ascontiguousarray(c, dtype=POINT)
                keep.append(a)
                views[i] = _View(a.ctypes.data_as(C.c_void_p), len(a), 16, 12)
        out = np.zeros((max(n, 1), 16), dtype=np.float32)
        pairs = np.zeros(max(n * (n - 1) // 2, 1), dtype=PAIR)
        n_out, n_pairs = C.c_size_t(), C.c_size_t()
        self._ck(lib().mm3d_estimate_maps_transforms(self._h, views, C.c_size_t(n), C.byref(params),
                                                     out.ctypes.data_as(C.c_void_p), C.byref(n_out),
                                                     pairs.ctypes.data_as(C.c_void_p), C.byref(n_pairs)))
        res = [_Tout(out[i]) for i in range(n_out.value)]
        return (res, pairs[:n_pairs.value].copy()) if return_pairs else res

    def composeMaps(self, clouds, transforms, resolution: float):
        n = len(clouds)
        if n == 0:
            return None                                  # nullptr (map_merging.h:97)
        if n != len(transforms):
            # the reference throws (R/src/map_merging.cpp:285-288)
            raise RuntimeError("composeMaps: clouds and transforms size must be the same.")
        arr = (C.c_void_p * n)(*[c._h for c in clouds])
        tr = np.ascontiguousarray(np.stack([_T(t) for t in transforms]))
        h = C.c_void_p()
        self._ck(lib().mm3d_compose_maps(self._h, arr, C.c_size_t(n), tr.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                         C.c_double(resolution), C.byref(h)))
        return Cloud(self, h)

    # ---- shardable pieces ------------------------------------------------------------------
    def mapFeatures(self, raw: "Cloud", params: MapMergingParams) -> "Map":
        h = C.c_void_p()
        self._ck(lib().mm3d_map_features(self._h, raw._h, C.byref(params), C.byref(h)))
        return Map(self, h)

    def mapFromParts(self, points: "Cloud", keypoints: "Cloud", desc: "Descriptors") -> "Map":
        h = C.c_void_p()
        self._ck(lib().mm3d_map_from_parts(self._h, points._h, keypoints._h, desc._h, C.byref(h)))
        for o in (points, keypoints, desc):
            o._owned = False                             # the map owns them now
        return Map(self, h)

    def mapPrepare(self, m: "Map", params: MapMergingParams) -> None:
        """mm3d_map_prepare: build the map's search structures now, so that pair estimates only read it
        (and may then run on several contexts at once)."""
        self._ck(lib().mm3d_map_prepare(self._h, m._h, C.byref(params)))

    def pairEstimate(self, source: "Map", target: "Map", params: MapMergingParams, execute: bool = True):
        r = np.zeros(1, dtype=PAIR)
        self._ck(lib().mm3d_pair_estimate(self._h, source._h, target._h, C.byref(params), int(execute),
                                          r.ctypes.data_as(C.c_void_p)))
        return r[0]

    def pairsSkip(self, sources, targets, params: MapMergingParams):
        """Replays the rand() draws of pairs this context does not execute (one call for the whole run)."""
        n = len(sources)
        if n == 0:
            return
        a = (C.c_void_p * n)(*[m._h for m in sources])
        b = (C.c_void_p * n)(*[m._h for m in targets])
        self._ck(lib().mm3d_pairs_skip(self._h, a, b, C.c_size_t(n), C.byref(params)))

    def shardBegin(self, clouds, params: MapMergingParams, rank: int, world: int) -> "Shard":
        """mm3d_shard_begin: the per-cloud loop for the maps `rank` owns, on this context's streams.
        clouds: host POINT arrays or (device ptr, n) tuples, all n of them (only the owned ones are read)."""
        n = len(clouds)
        keep = []
        views = (_View * max(n, 1))()
        for i, c in enumerate(clouds):
            if isinstance(c, tuple):
                views[i] = _View(C.c_void_p(c[0]), c[1], 16, 12)
            else:
                a = np.ascontiguousarray(c, dtype=POINT)
                keep.append(a)
                views[i] = _View(a.ctypes.data_as(C.c_void_p), len(a), 16, 12)
        h = C.c_void_p()
        self._ck(lib().mm3d_shard_begin(self._h, views, C.c_size_t(n), C.byref(params), int(rank), int(world), C.byref(h)))
        return Shard(self, h, n, int(params.descriptor_type))

    # ---- measurement -----------------------------------------------------------------------
    def profile(self, on: bool):
        self._ck(lib().mm3d_profile_enable(self._h, int(on)))

    def profile_reset(self):
        lib().mm3d_profile_reset(self._h)

    def profile_entries(self):
        out = {}
        for i in range(lib().mm3d_profile_count(self._h)):
            name, ms, n, b = C.c_char_p(), C.c_double(), C.c_uint64(), C.c_double()
            lib().mm3d_profile_entry(self._h, i, C.byref(name), C.byref(ms), C.byref(n), C.byref(b))
            out[name.value.decode()] = {"ms": ms.value, "launches": n.value, "bytes": b.value}
        return out


class Shard:
    """One rank's part of estimateMapsTransforms on N processes (include/mm3d.h, mm3d_shard_*)."""

    def __init__(self, ctx: "Context", h, n: int, descriptor_type: int):
        self._ctx, self._h, self.n, self.descriptor_type = ctx, h, n, descriptor_type

    def bundleSizes(self):
        """(n_points[n], n_keypoints[n]) of the maps this rank owns, zero elsewhere."""
        a, b = np.zeros(self.n, dtype=np.uint64), np.zeros(self.n, dtype=np.uint64)
        self._ctx._ck(lib().mm3d_shard_bundle_sizes(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        return a, b

    def bundleBytes(self, n_points: int, n_keypoints: int) -> int:
        f = lib().mm3d_shard_bundle_bytes
        f.restype = C.c_size_t
        return int(f(C.c_uint64(int(n_points)), C.c_uint64(int(n_keypoints)), int(self.descriptor_type)))

    def pack(self, i: int, dst_ptr: int):
        self._ctx._ck(lib().mm3d_shard_pack(self._h, C.c_size_t(i), C.c_void_p(dst_ptr)))

    def unpack(self, i: int, src_ptr: int, n_points: int, n_keypoints: int):
        self._ctx._ck(lib().mm3d_shard_unpack(self._h, C.c_size_t(i), C.c_void_p(src_ptr), C.c_uint64(int(n_points)),
                                              C.c_uint64(int(n_keypoints))))

    def unpackMany(self, items):
        """items: (map index, source pointer, n_points, n_keypoints) of the maps other ranks own; on the context's streams."""
        n = len(items)
        if n == 0:
            return
        maps = (C.c_size_t * n)(*[int(i[0]) for i in items])
        srcs = (C.c_void_p * n)(*[int(i[1]) for i in items])
        a = (C.c_uint64 * n)(*[int(i[2]) for i in items])
        b = (C.c_uint64 * n)(*[int(i[3]) for i in items])
        self._ctx._ck(lib().mm3d_shard_unpack_many(self._h, C.c_size_t(n), maps, srcs, a, b))

    def pairs(self):
        """(records of every live pair in the reference's order, mine[q]): the pairs whose target this rank owns
        are estimated, the other slots carry only the pair's indices."""
        cap = max(self.n * (self.n - 1) // 2, 1)
        rec = np.zeros(cap, dtype=PAIR)
        mine = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t()
        self._ctx._ck(lib().mm3d_shard_pairs(self._h, rec.ctypes.data_as(C.c_void_p), mine.ctypes.data_as(C.c_void_p),
                                             C.c_size_t(cap), C.byref(n)))
        return rec[:n.value].copy(), mine[:n.value].astype(bool)

    def end(self):
        if self._h:
            lib().mm3d_shard_end(self._h)
            self._h = None


def shardMapOwner(i: int, world: int) -> int:
    return int(lib().mm3d_shard_map_owner(C.c_size_t(i), int(world)))


def globalTransforms(pairs, confidence_threshold: float, n_clouds: int):
    """computeGlobalTransforms (R/src/map_merging.cpp:153-186); host only."""
    pairs = np.ascontiguousarray(pairs, dtype=PAIR)
    out = np.zeros((max(n_clouds, 1), 16), dtype=np.float32)
    n_out = C.c_size_t()
    st = lib().mm3d_global_transforms(pairs.ctypes.data_as(C.c_void_p), C.c_size_t(len(pairs)),
                                      C.c_double(confidence_threshold), C.c_size_t(n_clouds),
                                      out.ctypes.data_as(C.c_void_p), C.byref(n_out))
    if st != 0:
        raise Mm3dError(st, "mm3d_global_transforms")
    return [_Tout(out[i]) for i in range(n_out.value)]


class _Obj:
    _free = None

    def __init__(self, ctx: Context, h, owned=True):
        self.ctx, self._h, self._owned = ctx, h, owned

    def free(self):
        if self._h and self._owned and self.ctx._h:
            getattr(lib(), self._free)(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Cloud(_Obj):
    _free = "mm3d_cloud_free"

    def __len__(self):
        return lib().mm3d_cloud_size(self._h)

    def numpy(self) -> np.ndarray:
        out = np.empty(len(self), dtype=POINT)
        self.ctx._ck(lib().mm3d_cloud_download(self.ctx._h, self._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(16),
                                               C.c_size_t(12)))
        return out


class Normals(_Obj):
    _free = "mm3d_normals_free"

    def __len__(self):
        return lib().mm3d_normals_size(self._h)

    def numpy(self) -> np.ndarray:
        out = np.empty(len(self), dtype=NORMAL)
        self.ctx._ck(lib().mm3d_normals_download(self.ctx._h, self._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(16)))
        return out


class Descriptors(_Obj):
    _free = "mm3d_desc_free"

    def __len__(self):
        return lib().mm3d_desc_size(self._h)

    @property
    def dim(self):
        return lib().mm3d_desc_dim(self._h)

    def numpy(self) -> np.ndarray:
        out = np.empty((len(self), self.dim), dtype=np.float32)
        self.ctx._ck(lib().mm3d_desc_download(self.ctx._h, self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def frames(self) -> np.ndarray:
        """SHOT only: the local reference frames [n, 9] (x, y, z axes), the rf field of pcl::SHOT1344."""
        out = np.empty((len(self), 9), dtype=np.float32)
        self.ctx._ck(lib().mm3d_desc_download_frames(self.ctx._h, self._h, out.ctypes.data_as(C.c_void_p)))
        return out


class Map(_Obj):
    _free = "mm3d_map_free"

    @property
    def points(self) -> Cloud:
        return Cloud(self.ctx, C.c_void_p(lib().mm3d_map_points(self._h)), owned=False)

    @property
    def keypoints(self) -> Cloud:
        return Cloud(self.ctx, C.c_void_p(lib().mm3d_map_keypoints(self._h)), owned=False)

    @property
    def descriptors(self) -> Descriptors:
        return Descriptors(self.ctx, C.c_void_p(lib().mm3d_map_descriptors(self._h)), owned=False)
