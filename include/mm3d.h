/*
 * mm3d.h -- C ABI of libmm3d.so, the MI355X (gfx950) registration engine that replaces the
 * hot path of map_merge_3d (static library `map_merging`, R/CMakeLists.txt:67-74).
 *
 * The reference has no FFI: its seam is the C++ free-function API in
 * R/include/map_merge_3d/{features,matching,map_merging}.h.  Every entry point below names the
 * reference declaration it replaces; include/map_merge_3d_shim.hpp shows the C++ forwarding
 * layer a maintainer links instead of the static library.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; nothing throws across the boundary;
 *     every function returns an mm3d_status (MM3D_OK == 0) and mm3d_last_error() gives text.
 *   - "points" are pcl::PointXYZRGB payloads: float x,y,z at byte 0 and uint32 rgba
 *     (0xAARRGGBB, PCL byte order b,g,r,a) at byte `rgba_offset`, `stride` bytes apart.
 *     pcl::PointXYZRGB itself is stride 32 / rgba_offset 16; packed records are 16 / 12.
 *     Source pointers may be host or device (HBM) addresses.
 *   - 4x4 transforms are column-major float[16] (Eigen::Matrix4f storage); the all-zero matrix
 *     is the reference's "could not be estimated" sentinel (matching.h:41-42, map_merging.h:81-83).
 *   - enums carry the reference's integer values (features.h:20-24,49; matching.h:103).
 *   - one estimation at a time per context (internally serialised); contexts are independent.
 */
#ifndef MM3D_H_
#define MM3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MM3D_OK = 0,
  MM3D_EINVAL = -1,        /* bad argument (also: unknown enum string, like enums::from_string) */
  MM3D_EDEVICE = -2,       /* HIP runtime / device failure */
  MM3D_ENOMEM = -3,
  MM3D_EUNSUPPORTED = -4,  /* a size outside what the kernels are built for: an ordered-sum neighbourhood of more than 16384 points,
                              an SPFH neighbourhood of more than 65535, nr_scales != 3 (INTEGRATION.md "Size limits") */
  MM3D_ECAPACITY = -5      /* caller buffer too small; required size is reported */
} mm3d_status;

/* R/include/map_merge_3d/features.h:20-24 (ENUM_CLASS(Descriptor, PFH, PFHRGB, FPFH, RSD, SHOT, SC3D)) */
typedef enum { MM3D_DESC_PFH = 0, MM3D_DESC_PFHRGB, MM3D_DESC_FPFH, MM3D_DESC_RSD, MM3D_DESC_SHOT, MM3D_DESC_SC3D } mm3d_descriptor;
/* R/include/map_merge_3d/features.h:49 */
typedef enum { MM3D_KP_SIFT = 0, MM3D_KP_HARRIS } mm3d_keypoint;
/* R/include/map_merge_3d/matching.h:103 */
typedef enum { MM3D_EST_MATCHING = 0, MM3D_EST_SAC_IA } mm3d_estimation_method;

/* enums::to_string / enums::from_string (R/include/map_merge_3d/enum.h:30-67) and the
 * PointCloud2 field-name table of R/src/dispatch_descriptors.h:38-48. */
const char *mm3d_descriptor_name(int d);              /* "PFH" ... or NULL */
int mm3d_descriptor_from_string(const char *s);       /* value or MM3D_EINVAL */
const char *mm3d_descriptor_field_name(int d);        /* "pfh","pfhrgb","fpfh","r_min","shot","shape_context" */
int mm3d_descriptor_dim(int d);                       /* 125, 250, 33, 2, 1344, 1980 */
const char *mm3d_keypoint_name(int k);
int mm3d_keypoint_from_string(const char *s);
const char *mm3d_estimation_method_name(int m);
int mm3d_estimation_method_from_string(const char *s);

/* MapMergingParams, field for field (R/include/map_merge_3d/map_merging.h:28-44). */
typedef struct {
  double resolution;
  double descriptor_radius;
  int outliers_min_neighbours;
  double normal_radius;
  int keypoint_type;
  double keypoint_threshold;
  int descriptor_type;
  int estimation_method;
  int refine_transform;
  double inlier_threshold;
  double max_correspondence_distance;
  int max_iterations;
  uint64_t matching_k;
  double transform_epsilon;
  double confidence_threshold;
  double output_resolution;
} mm3d_params;
/* the defaults of map_merging.h:28-44 (dependent defaults evaluated from resolution = 0.1) */
void mm3d_params_default(mm3d_params *p);
/* MapMergingParams::fromCommandLine (R/src/map_merging.cpp:10-54): "--name value", unknown
 * options ignored, matching_k applied only if > 0; bad enum string -> MM3D_EINVAL. */
int mm3d_params_from_command_line(int argc, const char *const *argv, mm3d_params *p);
/* operator<<(ostream, MapMergingParams) (R/src/map_merging.cpp:100-123); returns bytes needed. */
size_t mm3d_params_to_string(const mm3d_params *p, char *buf, size_t cap);

typedef struct { int32_t index_query, index_match; float distance; } mm3d_corr; /* pcl::Correspondence */

typedef struct mm3d_ctx mm3d_ctx;
typedef struct mm3d_cloud mm3d_cloud;      /* device-resident PointCloud<PointXYZRGB> */
typedef struct mm3d_normals mm3d_normals;  /* device-resident PointCloud<Normal> */
typedef struct mm3d_desc mm3d_desc;        /* device-resident descriptors (PCLPointCloud2 payload) */

/* ---- context -------------------------------------------------------------------------- */
int mm3d_create(int device, mm3d_ctx **out);
/* One process, several GPUs.  The reference's callers are single processes (MapMerge3d::transformsEstimation, a ROS timer
 * callback: R/src/map_merge_node.cpp:133-153; map_merge_tool's main: R/src/map_merge_tool.cpp:37-38), so the multi-GPU form
 * of the path sits behind the same entry point: a context made from a device list runs mm3d_estimate_maps_transforms
 * sharded over those devices INSIDE the library -- features by map owner, the other maps' bundles pulled GPU to GPU
 * (hipMemcpyPeerAsync over xGMI), pairs by target owner, ONE RCCL all-gather (ncclAllGather, communicators from
 * ncclCommInitAll) of the 104-byte pair records, pose graph on the host -- with the bits of one device.  Every other entry
 * point of such a context works on its first device.  mm3d_set_streams applies to every device of the list.
 * Creating the communicators takes seconds (once, here); librccl.so.1 is loaded by this call (a context made by mm3d_create
 * never needs it).  A device listed twice is MM3D_EINVAL (test hook:
 * MM3D_DEVICES_ALLOW_DUPLICATES=1 admits it, the records are then gathered through host memory instead of RCCL, which
 * refuses a device twice; mm3d_devices_use_rccl tells). */
int mm3d_create_devices(const int *devices, int n_devices, mm3d_ctx **out);
int mm3d_device_count(const mm3d_ctx *ctx);          /* 1 for a context made by mm3d_create */
int mm3d_device_at(const mm3d_ctx *ctx, int i);      /* the i-th device of the list, or MM3D_EINVAL */
int mm3d_devices_use_rccl(const mm3d_ctx *ctx);      /* 1: pair records travel through ncclAllGather; 0: plain context or the test hook */
void mm3d_destroy(mm3d_ctx *ctx);
const char *mm3d_last_error(const mm3d_ctx *ctx);   /* ctx == NULL: why this thread's last mm3d_create / mm3d_create_devices failed */
/* diagnostics of the most recent ICP run on this context (pcl::Registration::nr_iterations_, converged_) */
int mm3d_last_icp_iterations(const mm3d_ctx *ctx);
int mm3d_last_icp_converged(const mm3d_ctx *ctx);
/* debug counters (cost a host sync per call when on): descriptor k-NN rows that failed the MFMA
 * certificate and were redone by the exact kernel, out of all rows, since the last reset */
void mm3d_set_debug(mm3d_ctx *ctx, int on);
long long mm3d_debug_knn_fallback_rows(mm3d_ctx *ctx);
long long mm3d_debug_knn_rows(mm3d_ctx *ctx);
/* host waits for the context's stream (and its mm3d_set_streams workers') since the context was made: out[0] = their number,
 * out[1] = nanoseconds spent in them */
void mm3d_debug_waits(mm3d_ctx *ctx, long long out[2]);
/* test hook: out[i] = the float sum "0 + incr[i] + incr[i] + ..." (hits[i] additions) as the PFH kernels
 * replay it for a histogram bin (PFHEstimation: "histogram[h] += hist_incr" once per pair) */
int mm3d_debug_float_chain(mm3d_ctx *ctx, const float *incr, const unsigned *hits, int n, float *out);
/* test hook: out[i] = the device's restatement of glibc's expf (fn 0), atanf (1), sinf (2), cosf (3) of x[i] or
 * atan2f(y[i], x[i]) (4) -- csrc/libm_exact.hpp, the functions the CPU path's PCL calls through libm */
int mm3d_debug_libm(mm3d_ctx *ctx, int fn, const float *x, const float *y, int n, float *out);
/* (fn 5: the raw v_exp_f32, 2^x, of the certified SIFT pass.)
 * test hooks of the certified SIFT decision (csrc/sift_cert.hpp; the later octaves of detectKeypoints(SIFT),
 * R/src/features.cpp:45-62): the unsorted scale space of octave `octave` (0-based) on `points` -- val[5 i + s] and
 * bound[5 i + s] >= |the CPU path's float DoG - val| for point i of the octave's cloud -- *n_out = that cloud's size
 * (nothing is written when it exceeds capacity; 0: no such octave); and process-wide counters since the last reset:
 * out[0] octaves decided on the certified path, [1] their points, [2] points that took the exact sorted-list path,
 * [3] points left open by the first pass, [4] octaves sent back to the sorted-list path, [5] bound violations (must be 0),
 * [6] points still open after the second pass (must be 0), [7] work items the unsorted pass could not stage */
int mm3d_debug_sift_cert_octave(mm3d_ctx *ctx, const mm3d_cloud *points, double min_scale, int octave, float *val, float *bound,
                                size_t capacity, size_t *n_out);
void mm3d_debug_sift_cert_stats(long long out[8], int reset);
/* test hook: octaves of at least n points take the certified path (default 15 000, MM3D_SIFT_CERT_MIN; n < 0 restores it) */
void mm3d_debug_sift_cert_min(int n);
/* test hook: the leaf of the VoxelGrid whose centroids the cloud's points are known to be (downSample's output and
 * removeOutliers' subset of it, R/src/features.cpp:19-40; every centroid within two leaves of a member of its voxel), 0 when
 * nothing is known -- what lets a grid build on the cloud skip its "did a cell outgrow the counting sort" wait */
float mm3d_debug_cloud_voxel_leaf(const mm3d_cloud *cloud);
/* test / study hook of SAC-IA's certified pick (csrc/registration.hip::k_sacia_select; R/src/matching.cpp:142-194 ->
 * SampleConsensusInitialAlignment keeps the hypothesis of the lowest error sum): process-wide counters since the last reset --
 * out[0] pairs scored, [1] pairs whose winner the sums in double decided (no float chain was run), [2] candidate hypotheses
 * the intervals left, [3] float chains run.  collect = 1 / 0 switches the collection on / off (it costs a wait per batch;
 * MM3D_SACIA_STATS in the environment switches it on for the whole process), collect < 0 leaves it as it is */
void mm3d_debug_sacia_stats(long long out[4], int reset, int collect);
/* test / study hook: the descriptor k-NN of findFeatureCorrespondences (R/src/matching.cpp:50-75) on raw rows of width `dim`
 * -- the widths of the reference's descriptors (2, 33, 125, 250, 1344, 1980) and 352, pcl::SHOT352's shape, which the reference
 * does not bind (dispatch_descriptors.h:44-46 binds SHOT1344) but BASELINE.json configs[3] names: idx / d2 receive na x k
 * nearest target rows in FLANN's (distance, index) order, exact */
int mm3d_debug_desc_knn(mm3d_ctx *ctx, const float *a, size_t na, const float *b, size_t nb, int dim, int k, int *idx, float *d2);
/* SAC-IA draws from libc rand() in the reference (process-global, glibc seed 1).  The context
 * carries its own replay of that generator; mm3d_srand re-seeds it (srand semantics). */
void mm3d_srand(mm3d_ctx *ctx, unsigned seed);
/* Number of HIP streams (each with its own memory pool and, while a call runs, its own host thread)
 * that mm3d_estimate_maps_transforms deals the per-cloud and per-pair loops of
 * map_merging.cpp:212-242,256-269 to.  1 (the default) = the reference's sequential loops on one
 * stream.  One pair is a chain of dependent kernels that cannot fill an MI355X by itself; 16 streams
 * roughly double the throughput.  The results do not depend on the setting, bit for bit: every
 * stream replays the rand() draws of the pairs it does not run.  When many pairs are ready at once (many small
 * maps) a stream takes up to 16 of them that share their target and runs them as one batch (one descriptor
 * search, one launch per step for all of them); the results are the sequential loop's all the same.
 * A stream's host thread polls and naps while it waits for the device (about 0.2 of a core per busy stream; MM3D_WAIT=spin
 * makes it spin): far more streams than the process has CPUs (a container's quota counts) slow everything down, and the
 * device runs four kernels at a time anyway.  1 <= n <= 64. */
int mm3d_set_streams(mm3d_ctx *ctx, int n_streams);
int mm3d_get_streams(const mm3d_ctx *ctx);
/* diagnostics of the most recent mm3d_estimate_maps_transforms on this context: seconds from entry
 * until the last map's features existed and until the call returned; per input cloud, the number
 * of points after downSample + removeOutliers and of keypoints after descriptor pruning (returns
 * the number of clouds; at most `capacity` entries are written) */
int mm3d_last_run_stage_seconds(const mm3d_ctx *ctx, double *features_s, double *total_s);
/* the same for a device-list context: seconds from entry until the slowest device had pulled the other maps' bundles, until
 * the slowest device had finished its pairs, and the duration of the RCCL gather of the pair records alone */
int mm3d_last_run_device_seconds(const mm3d_ctx *ctx, double *exchange_s, double *pairs_s, double *gather_s);
size_t mm3d_last_run_map_sizes(const mm3d_ctx *ctx, size_t *points, size_t *keypoints, size_t capacity);

/* ---- cloud objects -------------------------------------------------------------------- */
int mm3d_cloud_create(mm3d_ctx *ctx, const void *points, size_t n, size_t stride, size_t rgba_offset,
                      mm3d_cloud **out);
size_t mm3d_cloud_size(const mm3d_cloud *c);
int mm3d_cloud_download(mm3d_ctx *ctx, const mm3d_cloud *c, void *dst, size_t stride, size_t rgba_offset);
void mm3d_cloud_free(mm3d_ctx *ctx, mm3d_cloud *c);
size_t mm3d_normals_size(const mm3d_normals *n);
/* 16-byte records nx,ny,nz,curvature (pcl::Normal payload) */
int mm3d_normals_download(mm3d_ctx *ctx, const mm3d_normals *n, void *dst, size_t stride);
int mm3d_normals_create(mm3d_ctx *ctx, const void *normals, size_t n, size_t stride, mm3d_normals **out);
void mm3d_normals_free(mm3d_ctx *ctx, mm3d_normals *n);
size_t mm3d_desc_size(const mm3d_desc *d);
int mm3d_desc_dim(const mm3d_desc *d);
int mm3d_desc_type(const mm3d_desc *d);
int mm3d_desc_download(mm3d_ctx *ctx, const mm3d_desc *d, float *dst /* size*dim */);
/* SHOT only: the local reference frames, 9 floats per row (x, y, z axes) = the "rf" field of
 * pcl::SHOT1344; MM3D_EINVAL for other descriptor types. */
int mm3d_desc_download_frames(mm3d_ctx *ctx, const mm3d_desc *d, float *dst /* size*9 */);
int mm3d_desc_create(mm3d_ctx *ctx, const float *data, size_t n, int descriptor_type, mm3d_desc **out);
void mm3d_desc_free(mm3d_ctx *ctx, mm3d_desc *d);

/* ---- features.h ----------------------------------------------------------------------- */
/* downSample (R/include/map_merge_3d/features.h:34, R/src/features.cpp:17-27) */
int mm3d_downsample(mm3d_ctx *ctx, const mm3d_cloud *in, double resolution, mm3d_cloud **out);
/* removeOutliers (features.h:45, features.cpp:31-43) */
int mm3d_remove_outliers(mm3d_ctx *ctx, const mm3d_cloud *in, double radius, int min_neighbours,
                         mm3d_cloud **out);
/* computeSurfaceNormals (features.h:97, features.cpp:168-179) */
int mm3d_compute_normals(mm3d_ctx *ctx, const mm3d_cloud *in, double radius, mm3d_normals **out);
/* detectKeypoints (features.h:65, features.cpp:85-96): SIFT (features.cpp:45-62; normals and radius are
 * not used) or HARRIS (features.cpp:64-83: HarrisKeypoint3D on the given normals, non-maximum suppression
 * and refinement on, threshold, radius).  An invalid enum is UB in the reference (falls off the
 * switch); here MM3D_EINVAL. */
int mm3d_detect_keypoints(mm3d_ctx *ctx, const mm3d_cloud *points, const mm3d_normals *normals,
                          int type, double threshold, double radius, double resolution,
                          mm3d_cloud **keypoints);
/* The Harris response of every point (HarrisKeypoint3D::responseHarris, what detectKeypoints(HARRIS)
 * thresholds and suppresses); dst receives mm3d_cloud_size(points) floats. */
int mm3d_harris_response(mm3d_ctx *ctx, const mm3d_cloud *points, const mm3d_normals *normals, double radius,
                         float *dst);
/* computeLocalDescriptors (features.h:83, features.cpp:99-166).  Like the reference it prunes
 * keypoints whose descriptor is not finite: *keypoints is replaced IN PLACE by the pruned cloud.
 * All six rows of the dispatch table are built: PFH (dispatch_descriptors.h:38, the reference's
 * default), PFHRGB (:39), FPFH (:40), RSD (:43), SHOT (:46, i.e. SHOTColorEstimation / SHOT1344:
 * 352 shape + 992 colour bins) and SC3D (:47); other values -> MM3D_EINVAL. */
int mm3d_compute_descriptors(mm3d_ctx *ctx, const mm3d_cloud *points, const mm3d_normals *normals,
                             mm3d_cloud *keypoints, int descriptor, double feature_radius,
                             mm3d_desc **out);

/* ---- matching.h ----------------------------------------------------------------------- */
/* findFeatureCorrespondences (matching.h:26, matching.cpp:31-108).  Two-call protocol: with
 * out == NULL only *n is written. */
int mm3d_find_correspondences(mm3d_ctx *ctx, const mm3d_desc *source, const mm3d_desc *target, size_t k,
                              mm3d_corr *out, size_t cap, size_t *n);
/* estimateTransformFromCorrespondences (matching.h:44, matching.cpp:110-140) */
int mm3d_estimate_transform_from_correspondences(mm3d_ctx *ctx, const mm3d_cloud *source_keypoints,
                                                 const mm3d_cloud *target_keypoints, const mm3d_corr *corr,
                                                 size_t n_corr, double inlier_threshold, float T[16],
                                                 mm3d_corr *inliers, size_t cap, size_t *n_inliers);
/* estimateTransformFromDescriptorsSets, SAC-IA (matching.h:68, matching.cpp:142-194) */
int mm3d_estimate_transform_from_descriptors(mm3d_ctx *ctx, const mm3d_cloud *source_keypoints,
                                             const mm3d_desc *source_descriptors,
                                             const mm3d_cloud *target_keypoints,
                                             const mm3d_desc *target_descriptors, double min_sample_distance,
                                             double max_correspondence_distance, int max_iterations,
                                             float T[16]);
/* estimateTransformICP (matching.h:94, matching.cpp:196-221) */
int mm3d_estimate_transform_icp(mm3d_ctx *ctx, const mm3d_cloud *source, const mm3d_cloud *target,
                                const float initial_guess[16], double max_correspondence_distance,
                                double outlier_rejection_threshold, int max_iterations,
                                double transformation_epsilon, float T[16]);
/* estimateTransform (matching.h:129, matching.cpp:223-257) */
int mm3d_estimate_transform(mm3d_ctx *ctx, const mm3d_cloud *source_points, const mm3d_cloud *source_keypoints,
                            const mm3d_desc *source_descriptors, const mm3d_cloud *target_points,
                            const mm3d_cloud *target_keypoints, const mm3d_desc *target_descriptors,
                            int method, int refine, double inlier_threshold,
                            double max_correspondence_distance, int max_iterations, size_t matching_k,
                            double transform_epsilon, float T[16]);
/* transformScore (matching.h:150, matching.cpp:259-268) */
int mm3d_transform_score(mm3d_ctx *ctx, const mm3d_cloud *source, const mm3d_cloud *target,
                         const float T[16], double max_distance, double *score);

/* ---- map_merging.h -------------------------------------------------------------------- */
typedef struct { const void *points; size_t n; size_t stride; size_t rgba_offset; } mm3d_cloud_view;
/* TransformEstimate (R/src/graph.h:24-36) plus the integer observables of the pair: what
 * registration_visualisation prints as "cross-matches count" / "inliers count"
 * (R/src/registration_visualisation.cpp:129-130; both 0 for SAC_IA, which has neither) and the ICP trace
 * (pcl::Registration::nr_iterations_ and the number of correspondences of its last iteration). */
typedef struct {
  uint64_t source_idx, target_idx;
  float transform[16];
  double confidence;
  int32_t icp_iterations;
  int32_t n_correspondences;   /* findFeatureCorrespondences(...)->size()          (MATCHING) */
  int32_t n_inliers;           /* inliers->size() of estimateTransformFromCorrespondences (MATCHING) */
  int32_t icp_correspondences; /* correspondences within max_correspondence_distance in the last ICP iteration */
} mm3d_pair_result;

/* estimateMapsTransforms (map_merging.h:85, map_merging.cpp:188-275).
 * out_T has room for n*16 floats; *n_out = 0 for no cloud, 1 (identity) for one cloud,
 * otherwise `max pair index + 1` like the reference (map_merging.cpp:168) -- or n with all-zero
 * matrices when no pair survives (the reference is UB there).  pairs (optional, capacity
 * n*(n-1)/2) receives the pairwise estimates in pair order. */
int mm3d_estimate_maps_transforms(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n,
                                  const mm3d_params *params, float *out_T, size_t *n_out,
                                  mm3d_pair_result *pairs, size_t *n_pairs);
/* composeMaps (map_merging.h:99, map_merging.cpp:277-305): *out = NULL for n == 0 (nullptr in
 * the reference); n != n_transforms -> MM3D_EINVAL (the reference throws). */
int mm3d_compose_maps(mm3d_ctx *ctx, const mm3d_cloud *const *clouds, size_t n, const float *transforms,
                      size_t n_transforms, double resolution, mm3d_cloud **out);

/* ---- the same path in shardable pieces (one process per GPU; see bench.py) -------------- */
typedef struct mm3d_map mm3d_map;   /* per-map bundle: filtered cloud + keypoints + descriptors */
/* the per-cloud loop body of map_merging.cpp:212-242 */
int mm3d_map_features(mm3d_ctx *ctx, const mm3d_cloud *raw, const mm3d_params *params, mm3d_map **out);
const mm3d_cloud *mm3d_map_points(const mm3d_map *m);
const mm3d_cloud *mm3d_map_keypoints(const mm3d_map *m);
const mm3d_desc *mm3d_map_descriptors(const mm3d_map *m);
int mm3d_map_from_parts(mm3d_ctx *ctx, mm3d_cloud *points, mm3d_cloud *keypoints, mm3d_desc *desc,
                        mm3d_map **out);            /* takes ownership (feature exchange between ranks) */
/* Builds every search structure that pair estimates with `params` read from this map (point and
 * keypoint grids with their distance transforms, the Hilbert-ordered query copy, the host copy of
 * the keypoints).  Optional -- they are otherwise built lazily by the first pair that needs them --
 * but after it mm3d_pair_estimate only READS the map, so one map may serve pairs running on
 * several contexts (streams) at once.  Call it on the context that created the map. */
int mm3d_map_prepare(mm3d_ctx *ctx, mm3d_map *m, const mm3d_params *params);
void mm3d_map_free(mm3d_ctx *ctx, mm3d_map *m);
/* the per-pair loop body of map_merging.cpp:256-269.  execute == 0 only advances the context's
 * rand() replay exactly as the pair would (ranks that do not own the pair stay in lock-step with
 * the reference's single global stream). */
int mm3d_pair_estimate(mm3d_ctx *ctx, const mm3d_map *source, const mm3d_map *target,
                       const mm3d_params *params, int execute, mm3d_pair_result *out);
/* The rand() draws of n consecutive pairs of that loop that this context does NOT execute (what
 * mm3d_pair_estimate(execute = 0) does for one pair, in one call and without touching the device): keeps the
 * context's generator where the reference's sequential loop would have it. */
int mm3d_pairs_skip(mm3d_ctx *ctx, const mm3d_map *const *sources, const mm3d_map *const *targets, size_t n,
                    const mm3d_params *params);
/* computeGlobalTransforms (map_merging.cpp:153-186 + graph.cpp); host only, needs no device. */
int mm3d_global_transforms(const mm3d_pair_result *pairs, size_t n_pairs, double confidence_threshold,
                           size_t n_clouds, float *out_T, size_t *n_out);

/* ---- the same job on N processes, one per GPU, driven from inside the library ------------------------
 * estimateMapsTransforms' two loops shard naturally: maps are independent in the per-cloud loop
 * (map_merging.cpp:212-242), pairs in the per-pair loop (:256-269).  The caller only moves bytes between
 * its ranks (bench.py: torch.distributed over RCCL): one all-gather of feature bundles, one of pair records.
 *   1. mm3d_shard_begin: the per-cloud loop for the maps this rank owns (mm3d_shard_map_owner), on the
 *      context's streams (mm3d_set_streams), including the target-side search structures of those maps;
 *   2. mm3d_shard_bundle_sizes / mm3d_shard_pack: an owned map's bundle -- a 256-byte header, filtered points
 *      (16-byte records), keypoints (16-byte records), descriptors (rows of float), and both clouds once more in
 *      the library's Hilbert query order with their work items (the source role of ICP / score / SAC-IA scoring
 *      reads them in that order: the owner has it, a receiver would have to sort) -- contiguously at `dst`
 *      (device or host; mm3d_shard_bundle_bytes(points, keypoints, descriptor) bytes: every part at the size the
 *      two counts allow, opaque to the caller and only valid between ranks of one library build);
 *      after the exchange mm3d_shard_unpack hands every other map's bundle over (copies and one wait, no kernel);
 *   3. mm3d_shard_pairs: every live pair in the reference's order with the pairs whose TARGET this rank owns
 *      estimated on the context's streams (mine[q] = 1), the others zero; the rand() stream of the
 *      reference's single sequential loop is replayed on every rank, so the union over the ranks equals the
 *      one-process result bit for bit;
 *   4. the merged records go to mm3d_global_transforms on every rank. */
typedef struct mm3d_shard mm3d_shard;
int mm3d_shard_map_owner(size_t map, int world);   /* 0 1 .. w-1 w-1 .. 1 0 0 1 ..: evens out the pairs per target owner */
int mm3d_shard_begin(mm3d_ctx *ctx, const mm3d_cloud_view *clouds, size_t n, const mm3d_params *params, int rank, int world,
                     mm3d_shard **out);
int mm3d_shard_bundle_sizes(const mm3d_shard *sh, uint64_t *n_points /* [n] */, uint64_t *n_keypoints /* [n] */);   /* 0 for maps not owned */
size_t mm3d_shard_bundle_bytes(uint64_t n_points, uint64_t n_keypoints, int descriptor_type);
int mm3d_shard_pack(mm3d_shard *sh, size_t map, void *dst);
int mm3d_shard_unpack(mm3d_shard *sh, size_t map, const void *src, uint64_t n_points, uint64_t n_keypoints);
/* the same for several maps at once, dealt to the context's streams */
int mm3d_shard_unpack_many(mm3d_shard *sh, size_t count, const size_t *maps, const void *const *srcs, const uint64_t *n_points,
                           const uint64_t *n_keypoints);
int mm3d_shard_pairs(mm3d_shard *sh, mm3d_pair_result *pairs, unsigned char *mine, size_t capacity, size_t *n_pairs);
void mm3d_shard_end(mm3d_shard *sh);

/* ---- measurement ---------------------------------------------------------------------- */
/* per-kernel HIP-event timing on the context's own stream (bench.py roofline leg) */
int mm3d_profile_enable(mm3d_ctx *ctx, int on);
void mm3d_profile_reset(mm3d_ctx *ctx);
/* number of distinct kernels recorded; names via mm3d_profile_entry */
int mm3d_profile_count(mm3d_ctx *ctx);
int mm3d_profile_entry(mm3d_ctx *ctx, int i, const char **name, double *total_ms, uint64_t *launches,
                       double *algorithmic_bytes);
int mm3d_synchronize(mm3d_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MM3D_H_ */
