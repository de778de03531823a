// types.hpp -- device-resident objects behind the opaque C handles, and the stage entry points
// shared between the translation units of libmm3d.
//
// HBM layout:
//   cloud points   float4 {x, y, z, rgba-bits}            16 B / point, reference order
//   normals        float4 {nx, ny, nz, curvature}         16 B / point, same order
//   descriptors    float  [n][dim] row-major              (FPFH: 132 B / keypoint)
//   grid           cell_start int32[dx*dy*dz + 1] (x fastest) + a cell-sorted copy of the points
//                  whose .w carries the ORIGINAL index, so a (y,z) row of cells is one contiguous
//                  span of candidates.
#pragma once

#include "common.hpp"

namespace mm3d { struct DeviceSet; }

// the opaque C handles are these structs
struct mm3d_ctx : mm3d::Context {
  // mm3d_set_streams: helper contexts (one HIP stream + one host thread each while a call is running)
  // that mm3d_estimate_maps_transforms deals maps and pairs to; owned by this context
  std::vector<mm3d_ctx *> helpers;
  // mm3d_create_devices: the root contexts of the OTHER devices of the list (this context is the first device's); each has
  // its own helpers.  mm3d_estimate_maps_transforms then shards its two loops over the devices inside the library
  // (capi.cpp::estimate_maps_devices) and gathers the pair records through RCCL (devices.cpp).  Owned by this context.
  std::vector<mm3d_ctx *> peers;
  mm3d::DeviceSet *device_set = nullptr;                 // non-null exactly for contexts made by mm3d_create_devices
  // diagnostics of the most recent mm3d_estimate_maps_transforms (mm3d_last_run_*)
  double last_features_s = 0.0, last_total_s = 0.0;      // when the last map was ready / when the call returned
  double last_exchange_s = 0.0, last_pairs_s = 0.0, last_gather_s = 0.0;   // several devices: bundles moved / pairs done (slowest device) / the RCCL gather alone
  std::vector<size_t> last_points, last_keypoints;       // per input cloud, after filtering / after pruning
};

namespace mm3d {

struct GridView {
  float minx, miny, minz;
  float inv;   // 1 / cell
  float cell;
  int dx, dy, dz;
  const int *cell_start;
  const float4 *pts;   // cell-sorted; .w = original index bits
  int n;
  const unsigned char *dt;   // optional: Chebyshev distance (in cells, capped) to the nearest occupied cell
  int dt_cap;                // values > dt_cap are stored as 255
  // optional merged neighbourhood lists: nb_pts[nb_start[c] .. nb_start[c+1]) = every point of the
  // (2 nb_R + 1)^3 block of cells around cell c, so a fixed-radius query reads ONE contiguous span
  const int *nb_start;
  // .w = distance of the entry from the centre of cell c, and the list ascends in it: a nearest-point scan
  // stops once .w minus the cell's half diagonal exceeds what it still looks for.  (A list too long to
  // sort in LDS keeps stencil order and .w = 0: it is scanned to the end.)
  const float4 *nb_pts;
  int nb_R;
};

struct Grid {
  float cell = 0.f;
  float mn[3] = {0, 0, 0};
  int dims[3] = {1, 1, 1};
  DevBuf<int> cell_start;
  DevBuf<float4> sorted;
  int n = 0;
  DevBuf<unsigned char> dt;   // built on demand by grid_ensure_dt
  int dt_cap = 0;
  DevBuf<int> nb_start;       // built on demand by grid_ensure_nblists
  DevBuf<float4> nb_pts;      // x, y, z, distance from the list's cell centre
  int nb_R = 0;
  std::mutex cache_mu;        // grid_ensure_dt / grid_ensure_nblists
  GridView view() const
  {
    GridView v;
    v.minx = mn[0]; v.miny = mn[1]; v.minz = mn[2];
    v.inv = 1.0f / cell; v.cell = cell;
    v.dx = dims[0]; v.dy = dims[1]; v.dz = dims[2];
    v.cell_start = cell_start.get(); v.pts = sorted.get(); v.n = n;
    v.dt = dt.get(); v.dt_cap = dt_cap;
    v.nb_start = nb_start.get(); v.nb_pts = nb_pts.get(); v.nb_R = nb_R;
    return v;
  }
};

}  // namespace mm3d

struct mm3d_cloud {
  mm3d::DevBuf<float4> pts;
  size_t n = 0;
  // lazily computed, under cache_mu: several contexts (streams) may ask the same cloud for a structure that
  // mm3d_map_prepare did not build; the first builds it (and drains its stream), the others wait
  std::recursive_mutex cache_mu;
  bool have_bbox = false;
  float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
  size_t n_finite = 0;
  std::map<int, std::unique_ptr<mm3d::Grid>> grids;   // key: cell size in units of 1e-4 m
  std::vector<float4> host;                            // host copy (keypoint clouds only)
  mm3d::DevBuf<float4> hil_pts;                        // finite points in Hilbert order, .w = original index
  mm3d::DevBuf<uint32_t> hil_keys;                     // their sort keys: (Hilbert index of the 0.25 m column << 10) | z cell
  mm3d::DevBuf<int2> wave_items;                       // {first point, count <= 64}: one compact patch per wave
  int n_wave_items = 0;
  // > 0: the points are centroids of distinct voxels of a VoxelGrid with this leaf (downsample made the cloud, or a filter kept
  // a subset of such a cloud), none farther than two leaves from a member of its voxel (checked on the device where the
  // centroids were formed).  A grid cell of side C then holds at most (floor(C / leaf) + 6)^3 of them, and cloud_grid need not
  // ask the device whether a cell outgrew its counting sort.  0: unknown (a caller's raw cloud, a copy from a peer).
  float voxel_leaf = 0.f;
  // the points were replaced (descriptor pruning): everything derived from them goes
  void reset_caches()
  {
    grids.clear(); host.clear(); have_bbox = false; n_finite = 0;
    hil_pts = mm3d::DevBuf<float4>(); hil_keys = mm3d::DevBuf<uint32_t>(); wave_items = mm3d::DevBuf<int2>(); n_wave_items = 0;
  }
};

struct mm3d_normals {
  mm3d::DevBuf<float4> nrm;
  size_t n = 0;
};

struct mm3d_desc {
  mm3d::DevBuf<float> data;
  size_t n = 0;
  int dim = 0;
  int type = 0;
  // target-side operands of the MFMA k-NN (column sums + centred, augmented, MFMA-ordered rows):
  // they depend on this set alone, so a map that is the target of 15 pairs prepares them once
  // (desc_knn_prepare_target, called from mm3d_map_prepare)
  mm3d::DevBuf<float> knn_colsum, knn_Bp;
  std::mutex cache_mu;        // desc_knn_prepare_target
  // the rows' Euclidean norms in ascending order and the row index of each (the exact fallback of the k-NN only
  // visits targets whose norm is within the current k-th distance of the query's: | |a| - |b| | <= |a - b|)
  mm3d::DevBuf<uint32_t> knn_nsort, knn_nperm;
  // SHOT only: the local reference frames (x, y, z axes, 9 floats per row) -- the "rf" field of
  // pcl::SHOT1344; not part of the point representation that matching reads
  mm3d::DevBuf<float> rf;
};

struct mm3d_map {
  mm3d_cloud *points = nullptr;
  mm3d_cloud *keypoints = nullptr;
  mm3d_desc *desc = nullptr;
};

namespace mm3d {

// grid.hip
void cloud_bbox(Context *c, mm3d_cloud *cl);
const Grid &cloud_grid(Context *c, const mm3d_cloud *cl, float cell);
// per-cell Chebyshev distance transform (capped at R cells), cached on the grid
void grid_ensure_dt(Context *c, const Grid &g, int R);
// per-cell merged candidate lists over the (2R+1)^3 block of cells, cached on the grid
void grid_ensure_nblists(Context *c, const Grid &g, int R);
// Hilbert-ordered copy of the finite points (.w = original index) + wave work items, cached on the cloud
// min_cell: lower bound of the Hilbert cell (a work item never straddles a block of 8 x 8 cells); callers whose cloud is
// sparser than 0.1 m (SIFT's later octaves) pass a larger one so that their items still hold 64 points
void cloud_hilbert(Context *c, const mm3d_cloud *cl, float min_cell = 0.25f);
mm3d_cloud *cloud_from_device(Context *c, DevBuf<float4> &&pts, size_t n);
mm3d_cloud *cloud_from_memory(Context *c, const void *src, size_t n, size_t stride, size_t rgba_off);
void cloud_download(Context *c, const mm3d_cloud *cl, void *dst, size_t stride, size_t rgba_off);
// wait = false: the copy is enqueued and the caller waits for the stream itself before the vector is read (by anybody)
const std::vector<float4> &cloud_host(Context *c, const mm3d_cloud *cl, bool wait = true);
// ordered compaction: keeps in[i] where flags[i] != 0, preserving order; returns kept count
size_t compact_points(Context *c, const float4 *in, const int *flags, size_t n, DevBuf<float4> &out, unsigned *box_host = nullptr);   // box_host: 7 words for cloud_set_bbox
void cloud_set_bbox(mm3d_cloud *cl, const unsigned box[7]);
void exclusive_scan_int(Context *c, const int *in, int *out, size_t n);
// one chained-scan launch's share of the context's scan state (grid.hip::scan_prepare; scan_fused.hpp)
constexpr int kScanItems = 16, kScanTile = 256 * kScanItems;
struct ScanLaunchState { unsigned long long *status; unsigned *ticket; unsigned ticket_base, epoch, tiles; };
ScanLaunchState scan_prepare(Context *c, size_t n);
void counting_sort_pairs_u32(Context *c, const uint32_t *keys, int n, uint64_t key_range, uint32_t *keys_out, uint32_t *idx_out,
                             int *too_long, bool no_invalid_keys = false);
void sort_pairs_u32(Context *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                    size_t n, int end_bit);

// filters.hip
mm3d_cloud *downsample(Context *c, const mm3d_cloud *in, double resolution);
bool downsample_is_identity(Context *c, const mm3d_cloud *in, double resolution);   // would downsample() return `in` bit for bit? (one small launch + a wait)
mm3d_cloud *remove_outliers(Context *c, const mm3d_cloud *in, double radius, int min_neighbours);
mm3d_cloud *transform_concat(Context *c, const mm3d_cloud *const *clouds, size_t n, const float *T);

// normals.hip
mm3d_normals *compute_normals(Context *c, const mm3d_cloud *in, double radius);

// sift.hip
mm3d_cloud *detect_keypoints_sift(Context *c, const mm3d_cloud *points, double min_scale, int nr_octaves, int nr_scales, double min_contrast,
                                  double normals_radius = 0.0, mm3d_normals **normals_out = nullptr, float grid_cell_hint = 0.0f);   // (+ the points' normals, fused when possible;
                                  // grid_cell_hint: cell of a grid the caller will build on `points` anyway)
void normals_of_items(Context *c, const mm3d_cloud *in, const Grid &g, double radius, const int *ov_items, const int *ov_count_dev, int n_overflow,
                      float4 *out);

// harris.hip
mm3d_cloud *detect_keypoints_harris(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, double threshold,
                                    double radius);
void harris_response(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, double radius, DevBuf<float> &out);

// fpfh.hip
mm3d_desc *compute_fpfh(Context *c, const mm3d_cloud *points, const mm3d_normals *normals,
                        mm3d_cloud *keypoints, double radius);

// pfh.hip
mm3d_desc *compute_pfh(Context *c, const mm3d_cloud *points, const mm3d_normals *normals,
                       mm3d_cloud *keypoints, double radius);

mm3d_desc *compute_pfhrgb(Context *c, const mm3d_cloud *points, const mm3d_normals *normals,
                          mm3d_cloud *keypoints, double radius);

void debug_libm(Context *c, int fn, const float *x_host, const float *y_host, int n, float *out_host);
void debug_float_chain(Context *c, const float *incr_host, const unsigned *hits_host, int n, float *out_host);
// sift.hip, test hooks of the certified SIFT decision (sift_cert.hpp): the unsorted scale space of ONE octave -- val* and the
// bound B per point and DoG column, [n][5] by the octave cloud's index -- and the process-wide statistics
size_t debug_sift_cert_octave(Context *c, const mm3d_cloud *points, double min_scale, int octave, float *val_host, float *bound_host, size_t capacity);
void debug_sift_cert_stats(long long *out, int reset);
void debug_sift_cert_min(int n);     // test hook: octaves of at least n points are certified (< 0: the default / MM3D_SIFT_CERT_MIN)

// rsd.hip
mm3d_desc *compute_rsd(Context *c, const mm3d_cloud *points, const mm3d_normals *normals,
                       mm3d_cloud *keypoints, double radius);

// sc3d.hip
mm3d_desc *compute_sc3d(Context *c, const mm3d_cloud *points, const mm3d_normals *normals,
                        mm3d_cloud *keypoints, double radius);

// shot.hip
mm3d_desc *compute_shot(Context *c, const mm3d_cloud *points, const mm3d_normals *normals,
                        mm3d_cloud *keypoints, double radius);

// desc_knn.hip
// k nearest rows of B for every row of A (squared L2, FLANN accumulation order); idx -1 padded
void desc_knn(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2);
// cache the target-side operands of B on the set itself (a no-op for small or already prepared sets)
void desc_knn_prepare_target(Context *c, const mm3d_desc *B);
// the same for a subset of A's rows given as a device index list; result row r belongs to rows[r]
struct KnnRows { const mm3d_desc *A; const int *rows_dev; int n_rows; };
void desc_knn_rows_multi(Context *c, const KnnRows *srcs, int n_srcs, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2);
void desc_knn_rows(Context *c, const mm3d_desc *A, const int *rows_dev, int n_rows, const mm3d_desc *B, int k,
                   DevBuf<int> &idx, DevBuf<float> &d2);

// registration.hip
struct IcpResult { float T[16]; int iterations; int converged; };
struct PairTail { float T[16]; int iterations; int converged; int n_corr; double score; };
// one pair of a batch of ICP + score tails (icp_score_batch): inputs, then the result
struct IcpScoreJob {
  const mm3d_cloud *src = nullptr, *tgt = nullptr;
  const float *guess_dev = nullptr;     // the guess on the device, or
  float guess_host[16] = {0};           // on the host
  PairTail out{};
  bool closed = false;
};
void icp_score_batch(Context *c, IcpScoreJob *jobs, int n_jobs, bool run_icp, double max_corr_dist, int max_iterations, double eps,
                     bool want_score, double score_max_distance);
struct PairCounts { int n_correspondences = 0, n_inliers = 0, icp_correspondences = 0; };
// ICP (optional) from a guess on the device (guess_dev != null) or on the host, then transformScore
// (optional) of the result, with one host synchronisation
PairTail icp_score(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float *guess_dev, const float guess_host[16],
                   bool run_icp, double max_corr_dist, int max_iterations, double eps, bool want_score,
                   double score_max_distance);
IcpResult icp(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float guess[16],
              double max_corr_dist, int max_iterations, double eps);
double transform_score(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float T[16],
                       double max_distance);
// hypothesis scoring
void ransac_count(Context *c, const float4 *src_kp, const float4 *tgt_kp, const int *idx_src,
                  const int *idx_tgt, int n_corr, const float *T_all /* H*16 device */, int H,
                  double thr2, int *counts /* device H */);
// SAC-IA scoring of a batch of pairs (models from the replayed samples, truncated errors, their ordered sums, the
// first minimum): everything is device memory of the caller's, asynchronous
struct SacPair {
  const mm3d_cloud *src_kp, *tgt_kp;
  const int *samp;        // H*3: sampled source keypoints
  const int *corr_ref;    // H*3: index into nn
  const int *nn;          // k-NN table of the sampled rows
  float *T_best;          // out: the winning model, 16 floats
};
void sacia_score_batch(Context *c, const SacPair *pairs, int n, int H, float corr_thresh);
void debug_sacia_stats(long long out[4], int reset, int collect);
// build (and cache on the clouds) every search structure pair estimates with these parameters read
void prepare_pair_search(Context *c, const mm3d_cloud *points, double max_corr_dist, double score_max_distance);
void prepare_sacia_target(Context *c, const mm3d_cloud *kp, float corr_thresh);

// host_pipeline.cpp
size_t find_correspondences(Context *c, const mm3d_desc *s, const mm3d_desc *t, size_t k, std::vector<mm3d_corr> &out);
size_t ransac_transform(Context *c, const mm3d_cloud *skp, const mm3d_cloud *tkp, const mm3d_corr *corr,
                        size_t n_corr, double inlier_threshold, float T[16], std::vector<mm3d_corr> &inliers);
// T_dev == nullptr: the winning transform is returned in T (host).  Otherwise, when the device path
// ran, *T_dev receives it (16 floats on the device), T is left at identity and true is returned.
bool sac_ia(Context *c, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tkp, const mm3d_desc *td,
            double min_sample_distance, double max_corr_dist, int max_iterations, float T[16], bool execute,
            DevBuf<float> *T_dev = nullptr);
void sac_ia_draws(GlibcRand &rnd, const std::vector<float4> &skp, int ns, float min_sample_distance, int H, int kk, int *samp,
                  int *pick);
// advances rnd by the rand() draws one estimateTransform(source -> any non-empty target) consumes (host only)
void pair_rand_replay(GlibcRand &rnd, int method, const std::vector<float4> &skp_host, double inlier_threshold, int max_iterations);
// estimateTransform + (optionally) transformScore of the result; returns the ICP iteration count
// estimateTransform up to its initial estimate (correspondences + RANSAC, or SAC-IA): T0 on the host, or dT0 on the device
struct PairFront {
  float T0[16];
  DevBuf<float> dT0;
  bool on_device = false;
  PairCounts counts;
  // SAC-IA between its steps (sac_ia_replay / sac_ia_knn / sac_ia_finish): the replayed samples (samp | corr_ref |
  // rows, device) and the k-NN table of the sampled rows (its own buffer, or a slice of a batch's)
  DevBuf<int> sac_idx, sac_nn;
  DevBuf<float> sac_nd;
  const int *sac_nn_ptr = nullptr;
  int sac_rows = 0;          // distinct sampled rows
  int sac_h = 0;             // hypotheses to score; 0: nothing to do (too few keypoints, or not executed)
};
// SAC-IA in steps, so that several pairs can share launches: replay = the rand() stream and the upload of the
// sampled rows; knn = the descriptor k-NN of the sampled rows of every pair with the SAME target (td) as one search;
// finish = sacia_score_batch over the prepared pairs
void sac_ia_replay(Context *c, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tkp, const mm3d_desc *td,
                   double min_sample_distance, int max_iterations, bool execute, PairFront &f);
struct SacPrepared { const mm3d_cloud *skp, *tkp; const mm3d_desc *sd, *td; PairFront *front; };
void sac_ia_knn(Context *c, SacPrepared *same_target, int n, DevBuf<int> &nn_owner, DevBuf<float> &nd_owner);
void sac_ia_finish(Context *c, SacPrepared *pairs, int n, double max_corr_dist);
void estimate_pair_front(Context *c, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tkp, const mm3d_desc *td, int method,
                         double inlier_threshold, double max_corr_dist, int max_iterations, size_t matching_k, bool execute, PairFront &f);
int estimate_pair(Context *c, const mm3d_cloud *sp, const mm3d_cloud *skp, const mm3d_desc *sd, const mm3d_cloud *tp,
                  const mm3d_cloud *tkp, const mm3d_desc *td, int method, int refine, double inlier_threshold,
                  double max_corr_dist, int max_iterations, size_t matching_k, double eps, float T[16], bool execute,
                  bool want_score, double score_max_distance, double *score, PairCounts *counts = nullptr);
int estimate_transform(Context *c, const mm3d_cloud *sp, const mm3d_cloud *skp, const mm3d_desc *sd,
                       const mm3d_cloud *tp, const mm3d_cloud *tkp, const mm3d_desc *td, int method, int refine,
                       double inlier_threshold, double max_corr_dist, int max_iterations, size_t matching_k,
                       double eps, float T[16], bool execute);
int global_transforms(const mm3d_pair_result *pairs, size_t n_pairs, double thr, size_t n_clouds, float *out,
                      size_t *n_out);

// devices.cpp (host only): one process, several GPUs
// The RCCL communicators of a device list (ncclCommInitAll) and what the devices exchange: the maps' bundles, pulled by the
// device that needs them with hipMemcpyPeerAsync over xGMI, and the pair records, all-gathered with ncclAllGather.
struct DeviceSet;
DeviceSet *device_set_create(const int *devices, int n);          // throws Error; distinct devices get a communicator each
void device_set_destroy(DeviceSet *ds);
bool device_set_has_comms(const DeviceSet *ds);                   // false only for the duplicate-device test hook
// a copy of `src` (which lives on src_device) on c's device: points, bounding box and the source-side search structures
// (Hilbert query copy, work items, their keys), and the host copy when `src` has one.  Asynchronous on c's stream.
mm3d_cloud *cloud_clone_from_peer(Context *c, const mm3d_cloud *src, int src_device);
mm3d_desc *desc_clone_from_peer(Context *c, const mm3d_desc *src, int src_device);
// rank r contributes send[r] (its records, `slots` of them, zero padded); out receives world * slots records, rank by rank,
// as they arrived on roots[0]'s device.  One ncclAllGather per rank inside one group; returns the seconds it took.
double gather_pair_records(DeviceSet *ds, const std::vector<mm3d_ctx *> &roots, const std::vector<std::vector<mm3d_pair_result>> &send,
                           size_t slots, std::vector<mm3d_pair_result> &out);

// linalg (host)
void umeyama_f32(const float *src, const float *dst, int n, float T[16]);
void umeyama_f64(const double *src, const double *dst, int n, double T[16]);
void mat4_mul(const float A[16], const float B[16], float out[16]);
void mat4_inverse(const float A[16], float out[16]);
void eigen33_values(const float cov[9], float evals[3]);

}  // namespace mm3d
