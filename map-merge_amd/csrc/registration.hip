// registration.hip -- per-pair kernels: ICP correspondence+reduction, fitness score, RANSAC and
// SAC-IA hypothesis scoring (K10-K13 in SURVEY 2.2).
//
// estimateTransformICP  R/src/matching.cpp:196-221 -> pcl::IterativeClosestPoint (point-to-point,
//                       TransformationEstimationSVD/Umeyama, DefaultConvergenceCriteria)
// transformScore        R/src/matching.cpp:259-268 -> TransformationValidationEuclidean
// RANSAC scoring        SampleConsensusModelRegistration::countWithinDistance (matching.cpp:119-124)
// SAC-IA scoring        SampleConsensusInitialAlignment::computeErrorMetric (matching.cpp:159-173)
//
// The ICP iteration is two launches and no host round trip: icp_corr_reduce (one 1-NN per source
// point in the cached target grid, 17 double partial sums per block through wave shuffles) and
// icp_finalize (reduce the partials, Umeyama via a 3x3 Jacobi SVD, accumulate the transform and
// evaluate PCL's convergence tests on the device).  The host only polls a `done` word every few
// iterations.  Algorithmic traffic: 12 B per source point per iteration.
#include <cfloat>

#include "device_util.hpp"

namespace mm3d {

constexpr int kAcc = 17;   // sum p(3) | sum q(3) | sum q p^T (9, row = q) | sum d2 | count

struct IcpState {
  float T[16];      // cumulative transform applied to the original source points (starts at the guess)
  float Tinc[16];
  double prev_mse;
  double rot_thresh, trans_thresh;
  int iters, done, converged, n_corr, max_iter, pad;
};

// exact nearest neighbour within sqrt(max_d2): ring-by-ring walk of the target grid.
// Returns d2 (INFINITY if none) and the neighbour's coordinates.
__device__ __forceinline__ float nn_search(const GridView &g, float px, float py, float pz, float max_d2,
                                           float rmax, int max_ring, float4 &best_p)
{
  float best = INFINITY;
  int best_i = 0x7fffffff;
  const int cx = cell_floor(px, g.minx, g.inv), cy = cell_floor(py, g.miny, g.inv), cz = cell_floor(pz, g.minz, g.inv);
  for (int ring = 0; ring <= max_ring; ++ring) {
    if (ring >= 2) {
      // every unvisited cell is at least (ring-1) cells away from the query
      const float guard = (float)(ring - 1) * g.cell;
      if (guard > rmax) break;
      if (best <= guard * guard * 0.99999f) break;
    }
    const int z0 = cz - ring, z1 = cz + ring, y0 = cy - ring, y1 = cy + ring;
    const int xa = cx - ring, xb = cx + ring;
    if (xb < 0 || xa >= g.dx) continue;
    for (int z = z0 < 0 ? 0 : z0; z <= (z1 >= g.dz ? g.dz - 1 : z1); ++z) {
      const bool zs = (z == z0 || z == z1);
      for (int y = y0 < 0 ? 0 : y0; y <= (y1 >= g.dy ? g.dy - 1 : y1); ++y) {
        const bool shell = zs || y == y0 || y == y1;
        const int row = (z * g.dy + y) * g.dx;
        // shell rows: the whole x span; inner rows: only the two end cells
        const int npass = (shell || ring == 0) ? 1 : 2;
        for (int pass = 0; pass < npass; ++pass) {
          int lo, hi;
          if (npass == 1) { lo = xa; hi = xb; }
          else if (pass == 0) { lo = xa; hi = xa; }
          else { lo = xb; hi = xb; }
          lo = lo < 0 ? 0 : lo;
          hi = hi >= g.dx ? g.dx - 1 : hi;
          if (lo > hi) continue;
          const int b = g.cell_start[row + lo], e = g.cell_start[row + hi + 1];
          for (int j = b; j < e; ++j) {
            const float4 p = g.pts[j];
            const float d = dist2(px, py, pz, p.x, p.y, p.z);
            const int oi = __float_as_int(p.w);
            if (d < best || (d == best && oi < best_i)) { best = d; best_i = oi; best_p = p; }
          }
        }
      }
    }
  }
  return best <= max_d2 ? best : INFINITY;
}

// MODE 0: ICP (transform from the device state, accumulate Umeyama moments)
// MODE 1: transformScore (transform from Tc, accumulate sum d2 / count for d2 <= max_d2)
template <int MODE>
__global__ void __launch_bounds__(256)
k_nn_reduce(const float4 *__restrict__ src, int n, GridView g, const IcpState *__restrict__ st,
            const float *__restrict__ Tc, float max_d2, float rmax, int max_ring, double *__restrict__ partials)
{
  __shared__ float Ts[16];
  __shared__ double red[4][kAcc];
  if (MODE == 0 && st->done) return;
  if (threadIdx.x < 16) Ts[threadIdx.x] = (MODE == 0) ? st->T[threadIdx.x] : Tc[threadIdx.x];
  __syncthreads();
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int i = bid * blockDim.x + threadIdx.x;
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
  if (i < n) {
    const float4 s = src[i];
    const float3 p = xform(Ts, s.x, s.y, s.z);
    float4 q;
    const float d2 = nn_search(g, p.x, p.y, p.z, max_d2, rmax, max_ring, q);
    if (d2 <= max_d2) {   // false for INFINITY / NaN
      if (MODE == 0) {
        acc[0] = p.x; acc[1] = p.y; acc[2] = p.z;
        acc[3] = q.x; acc[4] = q.y; acc[5] = q.z;
        acc[6] = (double)q.x * p.x; acc[7] = (double)q.x * p.y; acc[8] = (double)q.x * p.z;
        acc[9] = (double)q.y * p.x; acc[10] = (double)q.y * p.y; acc[11] = (double)q.y * p.z;
        acc[12] = (double)q.z * p.x; acc[13] = (double)q.z * p.y; acc[14] = (double)q.z * p.z;
      }
      acc[15] = d2;
      acc[16] = 1.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = (MODE == 0 ? 0 : 15); k < kAcc; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    const int k = threadIdx.x;
    double v = 0.0;
    if (MODE == 0 || k >= 15) v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    partials[(size_t)bid * kAcc + k] = v;
  }
}

// ---- 3x3 SVD (one-sided Jacobi, double) and Umeyama on the device ------------------------------
__device__ void svd3(const double *A, double *U, double *S, double *V)
{
  double B[9];
  for (int i = 0; i < 9; ++i) { B[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int k = 0; k < 3; ++k) {
      const int p = (k == 2) ? 1 : 0, q = (k == 0) ? 1 : 2;
      double alpha = 0, beta = 0, gamma = 0;
      for (int i = 0; i < 3; ++i) {
        alpha += B[i * 3 + p] * B[i * 3 + p];
        beta += B[i * 3 + q] * B[i * 3 + q];
        gamma += B[i * 3 + p] * B[i * 3 + q];
      }
      if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
      rotated = 1;
      const double zeta = (beta - alpha) / (2.0 * gamma);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
      for (int i = 0; i < 3; ++i) {
        const double bp = B[i * 3 + p], bq = B[i * 3 + q];
        B[i * 3 + p] = c * bp - s * bq;
        B[i * 3 + q] = s * bp + c * bq;
        const double vp = V[i * 3 + p], vq = V[i * 3 + q];
        V[i * 3 + p] = c * vp - s * vq;
        V[i * 3 + q] = s * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  double nrm[3];
  for (int j = 0; j < 3; ++j) nrm[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
  int o0 = 0, o1 = 1, o2 = 2, t;
  if (nrm[o1] > nrm[o0]) { t = o0; o0 = o1; o1 = t; }
  if (nrm[o2] > nrm[o0]) { t = o0; o0 = o2; o2 = t; }
  if (nrm[o2] > nrm[o1]) { t = o1; o1 = o2; o2 = t; }
  const int ord[3] = {o0, o1, o2};
  double Vs[9];
  for (int j = 0; j < 3; ++j) {
    S[j] = nrm[ord[j]];
    for (int i = 0; i < 3; ++i) {
      Vs[i * 3 + j] = V[i * 3 + ord[j]];
      U[i * 3 + j] = (S[j] > 0.0) ? B[i * 3 + ord[j]] / S[j] : 0.0;
    }
  }
  for (int i = 0; i < 9; ++i) V[i] = Vs[i];
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= 0.0) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (S[1] <= tiny) {
    const double u0[3] = {U[0], U[3], U[6]};
    const int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = u0[m];
    const double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    U[1] = v[0] / n; U[4] = v[1] / n; U[7] = v[2] / n;
  }
  if (S[2] <= tiny) {
    const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1];
    U[5] = a[2] * b[0] - a[0] * b[2];
    U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

__device__ inline double det3(const double *M)
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// one block: reduce partials, Umeyama, accumulate, convergence (DefaultConvergenceCriteria)
__global__ void __launch_bounds__(256) k_icp_finalize(const double *__restrict__ partials, int nblocks, IcpState *st)
{
  __shared__ double red[4][kAcc];
  __shared__ double tot[kAcc];
  if (st->done) return;
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] += partials[(size_t)b * kAcc + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kAcc; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) tot[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  __syncthreads();
  if (threadIdx.x != 0) return;

  const double cnt = tot[16];
  st->n_corr = (int)cnt;
  if (cnt < 3.0) {   // min_number_correspondences_: "Not enough correspondences" -> not converged, stop
    st->converged = 0;
    st->done = 1;
    return;
  }
  const double inv = 1.0 / cnt;
  double mp[3] = {tot[0] * inv, tot[1] * inv, tot[2] * inv}, mq[3] = {tot[3] * inv, tot[4] * inv, tot[5] * inv};
  double sigma[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) sigma[r * 3 + c] = tot[6 + r * 3 + c] * inv - mq[r] * mp[c];
  double U[9], S[3], V[9];
  svd3(sigma, U, S, V);
  double Sd[3] = {1.0, 1.0, 1.0};
  if (det3(sigma) < 0) Sd[2] = -1.0;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(fabs(S[i]) <= fabs(S[0]) * 1e-5)) ++rank;
  if (rank == 2) {
    if (det3(U) * det3(V) > 0) { Sd[2] = 1.0; }
    else { Sd[2] = -1.0; }
  }
  float Ti[16];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += U[r * 3 + k] * Sd[k] * V[c * 3 + k];
      Ti[c * 4 + r] = (float)a;
    }
  }
  for (int r = 0; r < 3; ++r) {
    // t = dst_mean - R * src_mean (with the float R, like Eigen's float instantiation)
    double a = mq[r] - ((double)Ti[0 * 4 + r] * mp[0] + (double)Ti[1 * 4 + r] * mp[1] + (double)Ti[2 * 4 + r] * mp[2]);
    Ti[12 + r] = (float)a;
  }
  Ti[3] = Ti[7] = Ti[11] = 0.0f;
  Ti[15] = 1.0f;
  // final = Tinc * final
  float Tn[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float a = 0.0f;
      for (int k = 0; k < 4; ++k) a += Ti[k * 4 + r] * st->T[c * 4 + k];
      Tn[c * 4 + r] = a;
    }
  for (int i = 0; i < 16; ++i) { st->T[i] = Tn[i]; st->Tinc[i] = Ti[i]; }
  const int iters = ++st->iters;
  // DefaultConvergenceCriteria::hasConverged
  if (iters >= st->max_iter) { st->converged = 1; st->done = 1; return; }
  const double cos_angle = 0.5 * ((double)Ti[0] + (double)Ti[5] + (double)Ti[10] - 1.0);
  const double translation_sqr = (double)Ti[12] * Ti[12] + (double)Ti[13] * Ti[13] + (double)Ti[14] * Ti[14];
  if (cos_angle >= st->rot_thresh && translation_sqr <= st->trans_thresh) { st->converged = 1; st->done = 1; return; }
  const double mse = tot[15] * inv;
  if (fabs(mse - st->prev_mse) < 1e-12) { st->converged = 1; st->done = 1; return; }
  st->prev_mse = mse;
}

static const float4 *sorted_source(Context *c, const mm3d_cloud *src, int &n)
{
  // any cached grid gives a spatially coherent order; otherwise build the coarse one
  if (!src->grids.empty()) {
    const Grid &g = *src->grids.begin()->second;
    n = g.n;
    return g.sorted.get();
  }
  const Grid &g = cloud_grid(c, src, 0.5f);
  n = g.n;
  return g.sorted.get();
}

static float nn_cell_for(double radius)
{
  float cell = (float)(radius * 0.25);
  if (!(cell > 1e-3f)) cell = 0.25f;
  return cell;
}

IcpResult icp(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float guess[16],
              double max_corr_dist, int max_iterations, double eps)
{
  IcpResult res;
  memcpy(res.T, guess, sizeof(res.T));   // Identity * guess when nothing runs
  res.iterations = 0;
  res.converged = 0;
  if (src->n == 0 || tgt->n == 0) return res;
  const Grid &tg = cloud_grid(c, tgt, nn_cell_for(max_corr_dist));
  int ns = 0;
  const float4 *sp = sorted_source(c, src, ns);
  if (ns == 0 || tg.n == 0) return res;
  const double max_dist_sqr = max_corr_dist * max_corr_dist;
  // (double)d2 > max_dist_sqr rejects: accept d2 <= largest float not above max_dist_sqr
  float max_d2 = (float)max_dist_sqr;
  if ((double)max_d2 > max_dist_sqr) max_d2 = std::nextafterf(max_d2, -INFINITY);
  const float rmax = (float)(max_corr_dist * 1.0001 + 1e-5);
  const int max_ring = (int)std::ceil(rmax / tg.cell) + 1;

  IcpState h;
  memset(&h, 0, sizeof(h));
  memcpy(h.T, guess, sizeof(h.T));
  h.prev_mse = DBL_MAX;
  h.rot_thresh = 1.0 - eps;
  h.trans_thresh = eps;
  h.max_iter = max_iterations;
  IcpState *hp = (IcpState *)c->pin(sizeof(IcpState));
  *hp = h;
  DevBuf<IcpState> st(c, 1);
  MM3D_HIP(hipMemcpyAsync(st.get(), hp, sizeof(IcpState), hipMemcpyHostToDevice, c->stream));
  const unsigned nblocks = div_up(ns, 256);
  DevBuf<double> partials(c, (size_t)nblocks * kAcc);
  const GridView gv = tg.view();
  const int chunk = 4;
  for (;;) {
    for (int k = 0; k < chunk; ++k) {
      MM3D_LAUNCH(c, "icp_corr_reduce", ns * 12.0, k_nn_reduce<0>, dim3(nblocks), dim3(256), 0, sp, ns, gv, st.get(),
                  (const float *)nullptr, max_d2, rmax, max_ring, partials.get());
      MM3D_LAUNCH(c, "icp_finalize", nblocks * kAcc * 8.0, k_icp_finalize, dim3(1), dim3(256), 0, partials.get(),
                  (int)nblocks, st.get());
    }
    MM3D_HIP(hipMemcpyAsync(hp, st.get(), sizeof(IcpState), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (hp->done) break;
  }
  memcpy(res.T, hp->T, sizeof(res.T));
  res.iterations = hp->iters;
  res.converged = hp->converged;
  c->last_icp_iterations = res.iterations;
  c->last_icp_converged = res.converged;
  return res;
}

__global__ void __launch_bounds__(256) k_score_finalize(const double *__restrict__ partials, int nblocks, double *out)
{
  __shared__ double red[4][2];
  double s = 0.0, n = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    s += partials[(size_t)b * kAcc + 15];
    n += partials[(size_t)b * kAcc + 16];
  }
  s = wave_sum(s); n = wave_sum(n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s; red[wave][1] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    out[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}

double transform_score(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float T[16], double max_distance)
{
  if (src->n == 0 || tgt->n == 0) return DBL_MAX;
  // max_range_ is compared with the SQUARED distance (PCL quirk): search radius sqrt(max_distance)
  const double radius = std::sqrt(max_distance > 0 ? max_distance : 0.0);
  const Grid &tg = cloud_grid(c, tgt, nn_cell_for(radius));
  int ns = 0;
  const float4 *sp = sorted_source(c, src, ns);
  if (ns == 0 || tg.n == 0) return DBL_MAX;
  float max_d2 = (float)max_distance;
  if ((double)max_d2 > max_distance) max_d2 = std::nextafterf(max_d2, -INFINITY);
  const float rmax = (float)(radius * 1.0001 + 1e-5);
  const int max_ring = (int)std::ceil(rmax / tg.cell) + 1;
  const unsigned nblocks = div_up(ns, 256);
  DevBuf<double> partials(c, (size_t)nblocks * kAcc);
  DevBuf<float> dT(c, 16);
  DevBuf<double> out(c, 2);
  float *hT = (float *)c->pin(256);
  memcpy(hT, T, 64);
  MM3D_HIP(hipMemcpyAsync(dT.get(), hT, 64, hipMemcpyHostToDevice, c->stream));
  MM3D_LAUNCH(c, "score_nn_reduce", ns * 12.0 + tg.n * 12.0, k_nn_reduce<1>, dim3(nblocks), dim3(256), 0, sp, ns, tg.view(),
              (const IcpState *)nullptr, (const float *)dT.get(), max_d2, rmax, max_ring, partials.get());
  MM3D_LAUNCH(c, "score_finalize", 0, k_score_finalize, dim3(1), dim3(256), 0, partials.get(), (int)nblocks, out.get());
  double *ho = (double *)((char *)c->pin(256) + 128);
  MM3D_HIP(hipMemcpyAsync(ho, out.get(), 16, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  return ho[1] > 0.0 ? ho[0] / ho[1] : DBL_MAX;
}

// ---------------------------------------------------------------- RANSAC hypothesis scoring
// one wave per hypothesis; lanes stride over the correspondences; exact float predicate
__global__ void __launch_bounds__(256)
k_ransac_count(const float4 *__restrict__ skp, const float4 *__restrict__ tkp, const int *__restrict__ is,
               const int *__restrict__ it, int n_corr, const float *__restrict__ T_all, int H, float thr_le,
               int *__restrict__ counts)
{
  const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (h >= H) return;
  const int lane = threadIdx.x & 63;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = T_all[(size_t)h * 16 + k];
  int cnt = 0;
  for (int i = lane; i < n_corr; i += kWave) {
    const float4 s = skp[is[i]], t = tkp[it[i]];
    const float3 p = xform(T, s.x, s.y, s.z);
    const float dx = p.x - t.x, dy = p.y - t.y, dz = p.z - t.z;
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    cnt += (d <= thr_le) ? 1 : 0;
  }
  cnt = wave_sum(cnt);
  if (lane == 0) counts[h] = cnt;
}

void ransac_count(Context *c, const float4 *src_kp, const float4 *tgt_kp, const int *idx_src, const int *idx_tgt,
                  int n_corr, const float *T_all, int H, double thr2, int *counts)
{
  // (double)d < thr2  <=>  d <= largest float strictly below thr2
  float thr_le = (float)thr2;
  if ((double)thr_le >= thr2) thr_le = std::nextafterf(thr_le, -INFINITY);
  MM3D_LAUNCH(c, "ransac_count", (double)H * n_corr * 8.0, k_ransac_count, dim3(div_up(H, 4)), dim3(256), 0, src_kp, tgt_kp,
              idx_src, idx_tgt, n_corr, T_all, H, thr_le, counts);
}

// ---------------------------------------------------------------- SAC-IA hypothesis scoring
// E[i*H + h] = TruncatedError(d2 of (T_h * src_i) to its nearest target keypoint); h is the fast
// index so that the per-hypothesis sequential sum below is coalesced.
__global__ void __launch_bounds__(256)
k_sacia_err(const float4 *__restrict__ skp, int ns, GridView g, const float *__restrict__ T_all, int H, float thresh,
            float radius, float *__restrict__ E)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)ns * H) return;
  const int h = (int)(t / ns), i = (int)(t % ns);   // row h of E is contiguous over the keypoints
  const float *T = T_all + (size_t)h * 16;
  float Tl[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Tl[k] = T[k];
  const float4 s = skp[i];
  const float3 p = xform(Tl, s.x, s.y, s.z);
  float best = INFINITY;
  for_each_candidate(g, p.x, p.y, p.z, radius, [&](const float4 &q) {
    best = fminf(best, dist2(p.x, p.y, p.z, q.x, q.y, q.z));
    return true;
  });
  E[t] = (best <= thresh) ? best / thresh : 1.0f;
}

// error += e in source-keypoint order, float: the same chain the CPU path evaluates
__global__ void k_seq_sum(const float *__restrict__ E, int ns, int H, float *__restrict__ err)
{
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const float *row = E + (size_t)h * ns;
  float e = 0.0f;
  int i = 0;
  // the additions stay strictly sequential; only the loads are batched (8 in flight per thread)
  for (; i + 8 <= ns; i += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = row[i + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) e += v[k];
  }
  for (; i < ns; ++i) e += row[i];
  err[h] = e;
}

void sacia_errors(Context *c, const mm3d_cloud *src_kp, const mm3d_cloud *tgt_kp, const float *T_all, int H,
                  float corr_thresh, float *errors)
{
  const int ns = (int)src_kp->n;
  float radius = std::sqrt(corr_thresh > 0.f ? corr_thresh : 0.f);
  float cell = radius > 0.25f ? radius : 0.25f;
  const Grid &g = cloud_grid(c, tgt_kp, cell);
  DevBuf<float> E(c, (size_t)ns * H);
  const size_t total = (size_t)ns * H;
  MM3D_LAUNCH(c, "sacia_err", total * 4.0 + ns * 16.0, k_sacia_err, dim3(div_up(total, 256)), dim3(256), 0, src_kp->pts.get(), ns,
              g.view(), T_all, H, corr_thresh, radius, E.get());
  MM3D_LAUNCH(c, "sacia_seq_sum", total * 4.0, k_seq_sum, dim3(div_up(H, 64)), dim3(64), 0, E.get(), ns, H, errors);
  c->sync();
}

}  // namespace mm3d
