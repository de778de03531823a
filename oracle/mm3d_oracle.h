/*
 * mm3d_oracle.h -- CPU restatement of the map_merge_3d registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle_mm3d.so.  The product (libmm3d.so) never links or calls it.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in PCL 1.8.1 (+ FLANN
 * 1.9.1, Eigen 3.3.4, boost 1.65, glibc rand()), none of which is vendored in
 * /root/reference or installed in this image, and the reference's own tests
 * (R/test/test_map_merging.cpp:9-40) hold no numeric vectors for it.  This
 * file set restates the published PCL algorithms at the reference's call
 * sites; what CAN be pinned is pinned in tests/ (glibc rand() stream against
 * the libc in this image, mt19937 against numpy's legacy seeding, the five
 * degenerate-input gtests, analytic SE(3)/plane known answers).
 *
 * R/ = /root/reference/map_merge_3d/.  All matrices are column-major float[16]
 * (Eigen::Matrix4f layout): element (r,c) at m[c*4+r].
 */
#ifndef MM3D_ORACLE_H_
#define MM3D_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZRGB payload: x,y,z + rgba packed 0xAARRGGBB (PCL byte order b,g,r,a). */
typedef struct { float x, y, z; uint32_t rgba; } mo_point;
/* pcl::Normal payload. */
typedef struct { float nx, ny, nz, curvature; } mo_normal;
/* pcl::Correspondence */
typedef struct { int index_query, index_match; float distance; } mo_corr;

/* R/include/map_merge_3d/map_merging.h:28-44, same order, same defaults. */
typedef struct {
  double resolution;
  double descriptor_radius;
  int outliers_min_neighbours;
  double normal_radius;
  int keypoint_type;      /* 0 SIFT, 1 HARRIS   (features.h:49) */
  double keypoint_threshold;
  int descriptor_type;    /* 0 PFH 1 PFHRGB 2 FPFH 3 RSD 4 SHOT 5 SC3D (features.h:20) */
  int estimation_method;  /* 0 MATCHING, 1 SAC_IA (matching.h:103) */
  int refine_transform;
  double inlier_threshold;
  double max_correspondence_distance;
  int max_iterations;
  uint64_t matching_k;
  double transform_epsilon;
  double confidence_threshold;
  double output_resolution;
} mo_params;

void mo_params_default(mo_params *p);

/* ---------------- exact neighbour search (stands in for FLANN) ------------- */
/* Baseline B2 (SURVEY 8d "best-effort CPU"): the loops over points / keypoints / rows run on `n` OpenMP
 * threads; every order-sensitive float sum still runs sequentially, so the results do not depend on n.
 * The default is 1 thread: the reference's hot path is single-threaded (baseline B1). */
void mo_set_threads(int n);
int mo_get_threads(void);

typedef struct mo_grid mo_grid;
mo_grid *mo_grid_build(const mo_point *pts, int n, float cell);
void mo_grid_free(mo_grid *g);
/* all i with d2(q,p_i) < r2 (strict, FLANN RadiusResultSet), sorted by (d2,i).
 * idx/d2 are caller buffers of capacity cap; returns the full count (may be
 * > cap, then only the first cap in sorted order are NOT guaranteed: size it). */
int mo_radius_search(const mo_grid *g, float qx, float qy, float qz, float r2,
                     int *idx, float *d2, int cap);
/* k nearest, sorted by (d2,i); optional bound d2 <= max_d2 (pass INFINITY). */
int mo_knn_search(const mo_grid *g, float qx, float qy, float qz, int k,
                  float max_d2, int *idx, float *d2);

/* ---------------- features (R/src/features.cpp) --------------------------- */
/* downSample: pcl::VoxelGrid, R/src/features.cpp:17-27.  out has capacity n. */
int mo_downsample(const mo_point *in, int n, double resolution, mo_point *out);
/* removeOutliers: pcl::RadiusOutlierRemoval, R/src/features.cpp:31-43. */
int mo_remove_outliers(const mo_point *in, int n, double radius,
                       int min_neighbors, mo_point *out);
/* computeSurfaceNormals: pcl::NormalEstimation, R/src/features.cpp:168-179. */
void mo_normals(const mo_point *in, int n, double radius, mo_normal *out);
/* detectKeypoints(SIFT): R/src/features.cpp:45-62,85-96.  Returns count; the
 * keypoints (xyz, rgba = 0) are malloc'ed into *out (caller frees with
 * mo_free).  scales_out (optional, malloc'ed) receives the PointWithScale scale. */
int mo_keypoints_sift(const mo_point *in, int n, double min_scale,
                      int nr_octaves, int nr_scales_per_octave,
                      double min_contrast, mo_point **out, float **scales_out);
/* test / evidence hook (o_sift.c): one octave's scale space laid open -- float DoG, the responses in double, the counts
 * inside 3 sigma, the 25 nearest neighbours.  Everything malloc'ed (mo_free). */
int mo_sift_octave_debug(const mo_point *in, int n, double min_scale, int octave, int nr_scales_per_octave,
                         mo_point **cloud_out, int *n_out, float **dog_out, double **resp_out, int **cnt_out, int **knn_out);
/* exposure census of the audit list (o_audit.c; scripts/audit_exposure.py) */
void mo_audit_radius_ties(const mo_point *pts, int n, double radius, long long out[4]);
void mo_audit_desc_knn_unrolled(const float *a, int na, const float *b, int nb, int dim, int k, int *idx, float *d2);
void mo_audit_umeyama_order(const float *src, const float *dst, int n, int order, float T[16]);
int mo_audit_icp_correspondences(const mo_point *src, int ns, const mo_point *tgt, int nt, const float guess[16], double max_corr,
                                 float *src_out, float *dst_out);
void mo_umeyama_core_f32(const float sg[9], const float sm[3], const float dm[3], float one_over_n, float T[16]);
/* detectKeypoints(HARRIS): R/src/features.cpp:64-83 (o_harris.c).  Returns the count; keypoints (refined xyz,
 * rgba = 0) malloc'ed into *out; kept_idx (optional, malloc'ed) = the source indices; response_out
 * (optional, n floats) = the Harris response of every point. */
int mo_keypoints_harris(const mo_point *in, const mo_normal *normals, int n, double threshold, double radius,
                        mo_point **out, int **kept_idx, float *response_out);
void mo_harris_response(const mo_point *in, const mo_normal *normals, int n, double radius, float *response);
/* test hook (o_fpfh.c): pcl::computePairFeatures on n pairs, out[5 i ..] = {f1, f2, f3, f4, branch taken} */
void mo_pair_features(const mo_point *p1, const mo_normal *n1, const mo_point *p2, const mo_normal *n2, int n, float *out);
/* computeLocalDescriptors(FPFH): R/src/features.cpp:99-150 +
 * dispatch_descriptors.h:40.  keypoints are pruned IN PLACE (n_kp updated);
 * desc must hold n_kp*33 floats; returns the number of surviving keypoints. */
int mo_descriptors_fpfh(const mo_point *surface, const mo_normal *normals, int n,
                        mo_point *keypoints, int n_kp, double radius,
                        float *desc);
/* computeLocalDescriptors(PFH) -- the reference's default descriptor, dispatch_descriptors.h:38:
 * desc must hold n_kp*125 floats; keypoints pruned in place; returns the survivors. */
int mo_descriptors_pfh(const mo_point *surface, const mo_normal *normals, int n,
                       mo_point *keypoints, int n_kp, double radius, float *desc);
int mo_pfh_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
               int n_kp, double radius, float *desc /* n_kp x 125, NaN rows where no neighbour */);
/* computeLocalDescriptors(PFHRGB): dispatch_descriptors.h:39 = PFHRGBEstimation / PFHRGBSignature250.
 * desc must hold n_kp*250 floats; keypoints pruned in place; returns the survivors. */
int mo_descriptors_pfhrgb(const mo_point *surface, const mo_normal *normals, int n,
                          mo_point *keypoints, int n_kp, double radius, float *desc);
int mo_pfhrgb_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
                  int n_kp, double radius, float *desc /* n_kp x 250 */);
/* computeLocalDescriptors(RSD): dispatch_descriptors.h:43 = RSDEstimation / PrincipalRadiiRSD (r_min, r_max). */
int mo_descriptors_rsd(const mo_point *surface, const mo_normal *normals, int n,
                       mo_point *keypoints, int n_kp, double radius, float *desc /* n_kp x 2 */);
int mo_rsd_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
               int n_kp, double radius, float *desc /* n_kp x 2 */);
/* computeLocalDescriptors(SC3D): dispatch_descriptors.h:47 = ShapeContext3DEstimation / ShapeContext1980
 * (o_sc3d.c); consumes 3 draws of a boost::mt19937 seeded with 12345 per keypoint that has a neighbour. */
int mo_descriptors_sc3d(const mo_point *surface, const mo_normal *normals, int n,
                        mo_point *keypoints, int n_kp, double radius, float *desc /* n_kp x 1980 */);
int mo_sc3d_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
                int n_kp, double radius, float *desc /* n_kp x 1980 */);
void mo_sc3d_tables(double search_radius, float radii[16], float theta_div[12], float phi_div[13], float volume_lut[1980]);
/* computeLocalDescriptors(SHOT): dispatch_descriptors.h:46 = SHOTColorEstimation / SHOT1344 (o_shot.c).
 * desc must hold n_kp*1344 floats; keypoints pruned in place; returns the survivors. */
int mo_descriptors_shot(const mo_point *surface, const mo_normal *normals, int n,
                        mo_point *keypoints, int n_kp, double radius, float *desc);
/* un-pruned rows (NaN where PCL gives up) and the local reference frames (x, y, z axes; may be NULL) */
int mo_shot_raw(const mo_point *surface, const mo_normal *normals, int n, const mo_point *keypoints,
                int n_kp, double radius, float *desc /* n_kp x 1344 */, float *rf /* n_kp x 9 */);
/* RGB2CIELAB + the normalisation of computePointSHOT: lab = (L/100, a/120, b/120) */
void mo_shot_rgb2lab(unsigned char R, unsigned char G, unsigned char B, float lab[3]);
/* Raw (un-pruned) FPFH plus the SPFH support set, for stage-level parity. */
int mo_fpfh_raw(const mo_point *surface, const mo_normal *normals, int n,
                const mo_point *keypoints, int n_kp, double radius, float *desc,
                int *support_idx /* cap n, may be NULL */, float *spfh /* n*33 or NULL */);
void mo_free(void *p);

/* ---------------- matching (R/src/matching.cpp) --------------------------- */
/* exact k-NN in descriptor space, FLANN L2_Simple accumulation order. */
void mo_desc_knn(const float *a, int na, const float *b, int nb, int dim, int k,
                 int *idx /* na*k, -1 padded */, float *d2 /* na*k */);
/* findFeatureCorrespondences: R/src/matching.cpp:31-93.  out capacity ns. */
int mo_find_correspondences(const float *ds, int ns, const float *dt, int nt,
                            int dim, size_t k, mo_corr *out);
/* estimateTransformFromCorrespondences: R/src/matching.cpp:110-140.
 * inliers capacity n_corr.  Returns number of inliers (0 + zero matrix on failure).
 * iters_out / best_count_out (optional): RANSAC trace for parity checks. */
int mo_ransac(const mo_point *src_kp, int ns, const mo_point *tgt_kp, int nt,
              const mo_corr *corr, int n_corr, double inlier_threshold,
              float T[16], mo_corr *inliers, int *iters_out, int *best_count_out);
/* glibc rand() replay; state is process-global like libc's (mo_srand(1) at load). */
void mo_srand(unsigned seed);
int mo_rand(void);
/* estimateTransformFromDescriptorsSets (SAC-IA): R/src/matching.cpp:142-194.
 * best_iter_out / best_err_out optional. */
void mo_sac_ia(const mo_point *src_kp, const float *src_desc, int ns,
               const mo_point *tgt_kp, const float *tgt_desc, int nt, int dim,
               double min_sample_distance, double max_correspondence_distance,
               int max_iterations, float T[16], int *best_iter_out,
               float *best_err_out);
/* study hook (scripts/sacia_price.py): while `sink` is not NULL, mo_sac_ia also leaves the float error sum of hypothesis i
 * (computeErrorMetric, ia_ransac.hpp) in sink[i] for i < capacity and the same sum in double in sink64[i] */
void mo_sac_ia_error_sink(float *sink, double *sink64, int capacity);
/* estimateTransformICP: R/src/matching.cpp:196-221. iters_out optional. */
void mo_icp(const mo_point *src, int ns, const mo_point *tgt, int nt,
            const float guess[16], double max_correspondence_distance,
            double outlier_rejection_threshold, int max_iterations,
            double transformation_epsilon, float T[16], int *iters_out);
/* the same ICP with double sums over the original points (NOT the reference's arithmetic; see o_matching.c) */
void mo_icp_double_sums(const mo_point *src, int ns, const mo_point *tgt, int nt, const float guess[16],
                        double max_correspondence_distance, int max_iterations, double transformation_epsilon,
                        float T[16], int *iters_out);
/* estimateTransform: R/src/matching.cpp:223-257. */
void mo_estimate_transform(const mo_point *src, int ns, const mo_point *src_kp,
                           const float *src_desc, int nsk, const mo_point *tgt,
                           int nt, const mo_point *tgt_kp, const float *tgt_desc,
                           int ntk, int dim, int method, int refine,
                           double inlier_threshold,
                           double max_correspondence_distance, int max_iterations,
                           size_t matching_k, double transform_epsilon,
                           float T[16]);
/* the integer observables of the most recent mo_estimate_transform: cross-match and inlier counts
 * (R/src/registration_visualisation.cpp:129-130; MATCHING only) and the ICP trace */
typedef struct { int n_correspondences, n_inliers, icp_iterations, icp_correspondences; } mo_pair_trace;
void mo_last_pair_trace(mo_pair_trace *out);
void mo_last_pair_init(float T[16]);                  /* the initial estimate the last mo_estimate_transform gave ICP */
int mo_last_double_sums_correspondences(void);       /* last-iteration count of the last mo_icp_double_sums */
/* transformScore: R/src/matching.cpp:259-268. */
double mo_transform_score(const mo_point *src, int ns, const mo_point *tgt,
                          int nt, const float T[16], double max_distance);

/* the host libm over arrays: fn 0 expf, 1 atanf, 2 sinf, 3 cosf (of x), 4 atan2f(y, x)  (o_libm.c) */
void mo_libm_eval(int fn, const float *x, const float *y, int n, float *out);
/* small dense helpers exposed for tests */
void mo_umeyama_f32(const float *src, const float *dst, int n, float T[16]);
void mo_umeyama_f64(const double *src, const double *dst, int n, double T[16]);
void mo_mat4_inverse(const float A[16], float out[16]);
void mo_mat4_mul(const float A[16], const float B[16], float out[16]);
void mo_mt19937_seed(uint32_t seed);
uint32_t mo_mt19937_next(void);

/* ---------------- pose graph (R/src/graph.cpp, R/src/map_merging.cpp:137-186) */
typedef struct {
  size_t source_idx, target_idx;
  float transform[16];
  double confidence;
} mo_estimate;
/* returns number of nodes written to out (nodes_count = max index + 1), or 0
 * when there is no estimate (the reference is UB there; we return 0 nodes). */
int mo_global_transforms(const mo_estimate *pairs, int n_pairs,
                         double confidence_threshold, float *out /* nodes*16 */,
                         int out_cap_nodes);
int mo_largest_component(const mo_estimate *pairs, int n_pairs, double thr,
                         int *kept /* n_pairs flags */);
int mo_max_spanning_tree_centers(const mo_estimate *pairs, int n_pairs,
                                 size_t centers[2]);

/* ---------------- high level (R/src/map_merging.cpp) ----------------------- */
/* estimateMapsTransforms: R/src/map_merging.cpp:188-275.  Returns number of
 * transforms written (0, 1, or nodes_count).  pair_out (optional, capacity
 * n*(n-1)/2) receives the pairwise estimates in pair order. */
int mo_estimate_maps_transforms(const mo_point *const *clouds, const int *sizes,
                                int n_clouds, const mo_params *params,
                                float *out_T, mo_estimate *pair_out,
                                int *n_pairs_out);
int mo_last_run_traces(mo_pair_trace *out, int cap);
/* Exact-arithmetic yardstick of a whole job (NOT the reference's arithmetic): with the switch on,
 * mo_estimate_maps_transforms also runs mo_icp_double_sums from every pair's initial estimate; mo_last_run_exact copies
 * the per-pair results (16 floats, iteration count, last-iteration correspondence count) and returns how many exist. */
void mo_set_exact_yardstick(int on);
int mo_last_run_exact(float *T, int *iters, int *corr, int cap);
/* composeMaps: R/src/map_merging.cpp:277-305. returns -1 for empty input
 * (nullptr in the reference), -2 for size mismatch (the reference throws). */
int mo_compose_maps(const mo_point *const *clouds, const int *sizes, int n_clouds,
                    const float *transforms, int n_transforms, double resolution,
                    mo_point **out);

#ifdef __cplusplus
}
#endif
#endif
