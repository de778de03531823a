// desc_knn.hip -- exact k nearest neighbours in descriptor space (K8/K9 in SURVEY 2.2).
//
// Replaces the FLANN kd-trees over descriptors that the reference builds per pair:
//   findFeatureCorrespondences   R/src/matching.cpp:50-75  (k = matching_k, both directions)
//   SAC-IA findSimilarFeatures   R/src/matching.cpp:159-173 (k_correspondences_ = 10)
// A kd-tree in 33+ dimensions degenerates to a linear scan; here it IS a linear scan, tiled.
//
// Three stages (desc_knn below), templates on the row width (33 floats = FPFH, 125 = PFH):
//   1. knn_prep: descriptors -> centred, augmented, MFMA-ordered operands.  With mu = the targets'
//      column mean, query a' = [a - mu, |a - mu|^2, 1, 0], target b' = [-2 (b - mu), 1, |b - mu|^2, 0]
//      (36 / 128 wide), so a'.b' = |a - b|^2 directly.  The target side is cached on the descriptor
//      set (desc_knn_prepare_target): a map is the target of 15 pairs.
//   2. knn_mfma: the one genuine dense contraction of the pipeline on the matrix cores,
//      v_mfma_f32_32x32x2_f32 (exact f32 FMA chains, 157 TF peak).  One wave owns 32 queries as the
//      COLUMNS of the product, so each lane sees 16 target rows of one query per tile and keeps a
//      register-resident sorted list of the 8 best approximate distances.
//   3. knn_rerank (one wave per query row): the candidates (2 lane halves x 4 target slices x parts
//      x 8) are re-ranked with FLANN's L2_Simple accumulation (diff*diff summed in dimension order,
//      no FMA) -- the ONLY distances that leave this file -- and certified: the k-th exact distance
//      must clear the smallest "worst kept approximate distance" of any full list by more than the
//      expansion's rounding bound.  Rows that fail go through the exact brute-force kernels
//      (k_knn_exact_range; wide rows: k_knn_exact_wide / k_knn_merge_parts).  The result is therefore the exact k-NN with ties to
//      the lower index, bit-identical to the CPU path.
#include "device_util.hpp"

namespace mm3d {

int snb_cu_count(int device);                    // grid.hip

constexpr int kMaxK = 16;
// The kernels are templates on the descriptor dimension kD: 33 (FPFH) and 125 (PFH) are instantiated.
// padded contraction length (dimension + the two augmentation columns, even): 36 / 128
// (wide rows -- PFHRGB's 250 and SHOT's 1344 floats -- pad to a multiple of 8 so the step loop unrolls by 4: 256 / 1352)
constexpr int knn_kp(int d) { return d > 128 ? (d + 9) / 8 * 8 : (d + 3) / 2 * 2; }
constexpr int kSlices = 4;       // waves per query tile, each scanning a quarter of the targets
constexpr int kLists = 2 * kSlices;
constexpr int kListLen = 8;      // per-lane candidate list of the MFMA stage (8 lists x 8 = 64 candidates per query and part)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- exact brute force (VALU)
// one thread per query row (registers), target rows staged through LDS in tiles of 64 and
// broadcast-read by every lane.  Used for small problems and for rows that fail the certificate.
template <int D>
__global__ void __launch_bounds__(128)
k_knn_exact(const float *__restrict__ A, int na, const float *__restrict__ B, int nb, int k,
            const int *__restrict__ rows /* optional subset of A rows */, const int *__restrict__ nrows_dev, int nrows_host,
            int *__restrict__ idx, float *__restrict__ d2out)
{
  constexpr int TB = 64;
  __shared__ float tile[TB][D + 1];
  const int nrows = nrows_dev ? *nrows_dev : nrows_host;
  if ((int)(blockIdx.x * blockDim.x) >= nrows) return;   // uniform per block
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

// per-part key lists of the wide brute-force kernel (k_knn_exact_wide), merged by k_knn_merge_parts
constexpr int kFbParts = 16;
constexpr unsigned long long kEmptyKey = 0x7f8000007fffffffull;   // (+inf, no index)

__global__ void k_knn_iota(int *__restrict__ rows, int n, int *__restrict__ count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rows[i] = i;
  if (i == 0) *count = n;
}

__global__ void k_knn_merge_parts(const unsigned long long *__restrict__ part_keys, int k, const int *__restrict__ rows,
                                  const int *__restrict__ nrows_dev, int *__restrict__ idx, float *__restrict__ d2out)
{
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= *nrows_dev) return;
  const int row = rows[slot];
  const unsigned long long *pk = part_keys + (size_t)slot * kFbParts * kMaxK;
  unsigned long long prev = 0ull;
  bool first = true;
  for (int o = 0; o < k; ++o) {
    unsigned long long best = kEmptyKey;
    for (int p = 0; p < kFbParts; ++p)
      for (int s = 0; s < k; ++s) {
        const unsigned long long key = pk[p * kMaxK + s];
        if ((first || key > prev) && key < best) best = key;
      }
    const float d = __uint_as_float((unsigned)(best >> 32));
    idx[(size_t)row * k + o] = d < INFINITY ? (int)(unsigned)(best & 0xffffffffull) : -1;
    d2out[(size_t)row * k + o] = d;
    prev = best;
    first = false;
  }
}

// ---------------------------------------------------------------- stage 1: operand preparation
// Xp[(tile * kSteps + s) * 64 + lane] = x'[tile*32 + (lane & 31)][2*s + (lane >> 5)]: one coalesced
// 256-byte wave load per MFMA step.
//
// column sums of the targets (33 floats) -> mu = sum / n.  Distances do not change when both sides
// are shifted by the same vector, but the rounding error of the |a|^2 + |b|^2 - 2ab expansion does:
// FPFH rows share a large common component, and centred rows make the certificate below tight.
template <int kD>
__global__ void k_knn_colsum(const float *__restrict__ X, int n, float *__restrict__ sum)
{
  static_assert(kD <= 128, "one 128-lane group per partial sum");
  const int d = threadIdx.x & 127;
  const int part = blockIdx.x * (blockDim.x >> 7) + (threadIdx.x >> 7), nparts = gridDim.x * (blockDim.x >> 7);
  if (d >= kD) return;
  float acc = 0.0f;
  for (int r = part; r < n; r += nparts) acc += X[(size_t)r * kD + d];
  atomicAdd(&sum[d], acc);
}

template <int kD>
__global__ void k_knn_prep(const float *__restrict__ X, int n, int ntiles, int is_target, const float *__restrict__ colsum,
                           float inv_nb, float *__restrict__ Xp)
{
  constexpr int kSteps = knn_kp(kD) / 2;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)ntiles * kSteps * 64) return;
  const int lane = (int)(e & 63);
  const int s = (int)((e >> 6) % kSteps);
  const int tile = (int)(e / (64 * kSteps));
  const int row = tile * 32 + (lane & 31);
  const int kk = 2 * s + (lane >> 5);
  float v = 0.0f;
  if (row < n) {
    const float *x = X + (size_t)row * kD;
    if (kk < kD) {
      const float xc = x[kk] - colsum[kk] * inv_nb;
      v = is_target ? -2.0f * xc : xc;
    } else if (kk == kD || kk == kD + 1) {
      const bool want_norm = is_target ? (kk == kD + 1) : (kk == kD);
      if (want_norm) {
        float nrm = 0.0f;
        for (int d = 0; d < kD; ++d) { const float xc = x[d] - colsum[d] * inv_nb; nrm = fmaf(xc, xc, nrm); }
        v = nrm;
      } else {
        v = 1.0f;
      }
    }
  } else if (is_target) {
    // padding targets: a'.b' = 1e30, never a candidate ahead of a real row
    v = (kk == kD + 1) ? 1e30f : 0.0f;
  }
  Xp[e] = v;
}

// ---------------------------------------------------------------- stage 2: MFMA distance tiles
// block = one tile of 32 queries; its 4 waves scan disjoint quarters of the target tiles.
template <int kD>
__global__ void __launch_bounds__(256)
k_knn_mfma(const float *__restrict__ Ap, int na, const float *__restrict__ Bp, int nb, int nb_tiles, int tile_step,
           float *__restrict__ cand_d, int *__restrict__ cand_i)
{
  constexpr int kSteps = knn_kp(kD) / 2;   // 32x32x2 MFMA steps: 18 (FPFH) / 64 (PFH)
  const int lane = threadIdx.x & 63;
  const int slice = threadIdx.x >> 6;
  const int tile_a = blockIdx.x;
  float af[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) af[s] = Ap[((size_t)tile_a * kSteps + s) * 64 + lane];
  float ld[kListLen];
  int li[kListLen];
#pragma unroll
  for (int s = 0; s < kListLen; ++s) { ld[s] = INFINITY; li[s] = -1; }
  // (part, slice) interleave the target tiles (tile c belongs to unit c % (parts * kSlices)): similar
  // descriptors sit at nearby indices (same keypoint at several scales, spatial neighbours), and
  // interleaving spreads a query's true neighbours evenly over the lists, which is what keeps short
  // lists certifiable.  gridDim.y = parts > 1 when there are few query tiles (SAC-IA only looks up
  // its sampled rows): the targets are then split over more blocks so the launch still fills the chip.
  // tile_step > 1: only every tile_step-th target tile is visited (the sample that sets the filter's thresholds)
  const int part = blockIdx.y, stride = kSlices * (int)gridDim.y * tile_step;
  const int c0 = (part * kSlices + slice) * tile_step, c1 = nb_tiles;
  float bf[kSteps], bn[kSteps];
  if (c0 < c1) {
#pragma unroll
    for (int s = 0; s < kSteps; ++s) bf[s] = Bp[((size_t)c0 * kSteps + s) * 64 + lane];
  }
  for (int c = c0; c < c1; c += stride) {
    // prefetch the next target tile while the matrix core works on this one
    const int cn = (c + stride < c1) ? c + stride : c;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) bn[s] = Bp[((size_t)cn * kSteps + s) * 64 + lane];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // rows of the product = targets (A operand), columns = queries (B operand)
#pragma unroll
    for (int s = 0; s < kSteps; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
    const int rbase = c * 32 + 4 * (lane >> 5);
    float tmin = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[r]);
    if (__any(tmin < ld[kListLen - 1]))   // wave-uniform: most late tiles improve no lane's list
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[r];
      if (v < ld[kListLen - 1]) {
        float cd = v;
        int ci = rbase + (r & 3) + 8 * (r >> 2);
        bool carrying = false;
#pragma unroll
        for (int s = 0; s < kListLen; ++s) {
          const bool sw = carrying || cd < ld[s];
          carrying = sw;
          const float td = ld[s];
          const int ti = li[s];
          ld[s] = sw ? cd : td; li[s] = sw ? ci : ti;
          cd = sw ? td : cd; ci = sw ? ti : ci;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kSteps; ++s) bf[s] = bn[s];
  }
  const int a = tile_a * 32 + (lane & 31);
  if (a >= na) return;
  const int n_lists = kLists * (int)gridDim.y;
  const int list = part * kLists + (lane >> 5) * kSlices + slice;
  float *od = cand_d + ((size_t)a * n_lists + list) * kListLen;
  int *oi = cand_i + ((size_t)a * n_lists + list) * kListLen;
#pragma unroll
  for (int s = 0; s < kListLen; ++s) {
    const bool real = li[s] >= 0 && li[s] < nb;
    od[s] = real ? ld[s] : INFINITY;
    oi[s] = real ? li[s] : -1;
  }
}

// ---------------------------------------------------------------- stage 2': threshold filter (short rows, many targets)
// Keeping sorted lists in the MFMA loop costs five times the matrix work (every tile finds SOME lane with a new
// entry, and a wave's lists never mature when the targets are split over many blocks).  Instead:
//   a. k_knn_mfma over every kSampleStep-th target tile (lists as above, a small job), and per query
//      theta = the (k + kThetaExtra)-th smallest approximate distance of that sample (k_knn_theta) -- an upper
//      bound of the k-th smallest over all targets, with room for the certificate;
//   b. k_knn_filter: the full product; a lane only compares its 16 values with its query's theta and appends
//      the few that pass (about (k + kThetaExtra) * kSampleStep per query) to the query's candidate buffer;
//   c. k_knn_rerank_filter: exact distances of those candidates; a target that is NOT a candidate has an
//      approximate distance >= theta, which is what the certificate needs (tau = theta).  A buffer that
//      overflows leaves its row to the exact fallback.
constexpr int kSampleStep = 8;
constexpr int kThetaExtra = 6;
constexpr int kFilterCap = 512;

__device__ __forceinline__ unsigned knn_ordered_bits(float v);
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v);
template <int kD>
__device__ __forceinline__ void knn_exact_range_row(const float (&x)[kD], float U, int row, const float *__restrict__ B, int nb, int k,
                                                    const uint32_t *__restrict__ nsort, const uint32_t *__restrict__ nperm, int lane,
                                                    int *__restrict__ idx, float *__restrict__ d2out);

// wave-wide minimum / sum through the DPP network (row shifts, then row broadcasts; the last lane holds the result)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned knn_dpp_keep(unsigned v)           // lanes without a source keep their own value
{
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ unsigned knn_wave_min_u32(unsigned v)
{
  v = min(v, knn_dpp_keep<0x111, 0xf>(v));
  v = min(v, knn_dpp_keep<0x112, 0xf>(v));
  v = min(v, knn_dpp_keep<0x114, 0xf>(v));
  v = min(v, knn_dpp_keep<0x118, 0xf>(v));
  v = min(v, knn_dpp_keep<0x142, 0xa>(v));
  v = min(v, knn_dpp_keep<0x143, 0xc>(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int knn_wave_sum_i32(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return __builtin_amdgcn_readlane(v, 63);
}

// one wave per query row: theta[a] = the kq-th smallest approximate distance among the row's sample candidates,
// counted with multiplicity (n_cand <= 64 * kThetaPerLane: a lane keeps its share in registers as order-preserving
// 32-bit keys and the wave draws distinct minima until kq values are covered)
constexpr int kThetaPerLane = 8;
__global__ void __launch_bounds__(256)
k_knn_theta(const float *__restrict__ cand_d, const int *__restrict__ cand_i, int na, int n_cand, int kq, float *__restrict__ theta)
{
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (a >= na) return;               // wave-uniform
  const float *cd_ = cand_d + (size_t)a * n_cand;
  const int *ci_ = cand_i + (size_t)a * n_cand;
  unsigned key[kThetaPerLane];       // 0xffffffff = none (an ordered key of a finite distance is never all ones)
#pragma unroll
  for (int u = 0; u < kThetaPerLane; ++u) {
    const int e = lane + u * kWave;
    key[u] = 0xffffffffu;
    if (e < n_cand && ci_[e] >= 0) key[u] = knn_ordered_bits(cd_[e]);
  }
  unsigned best = 0xffffffffu;
  int covered = 0;
  while (covered < kq) {
    unsigned cur = key[0];
#pragma unroll
    for (int u = 1; u < kThetaPerLane; ++u) cur = min(cur, key[u]);
    best = knn_wave_min_u32(cur);
    if (best == 0xffffffffu) break;                   // fewer than kq candidates: no bound
    int mine = 0;
#pragma unroll
    for (int u = 0; u < kThetaPerLane; ++u) {
      mine += key[u] == best ? 1 : 0;
      key[u] = key[u] == best ? 0xffffffffu : key[u];
    }
    covered += knn_wave_sum_i32(mine);
  }
  // back from the ordered key to the float
  const unsigned bits = (best & 0x80000000u) ? (best & 0x7fffffffu) : ~best;
  if (lane == 0) theta[a] = best == 0xffffffffu ? INFINITY : __uint_as_float(bits);
}

template <int kD>
__global__ void __launch_bounds__(256)
k_knn_filter(const float *__restrict__ Ap, int na, const float *__restrict__ Bp, int nb, int nb_tiles, const float *__restrict__ theta,
             int *__restrict__ cand_n /* [na], zeroed */, int *__restrict__ cand_i /* [na][kFilterCap] */)
{
  constexpr int kSteps = knn_kp(kD) / 2;
  const int lane = threadIdx.x & 63;
  const int slice = threadIdx.x >> 6;
  const int tile_a = blockIdx.x;
  float af[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) af[s] = Ap[((size_t)tile_a * kSteps + s) * 64 + lane];
  const int a = tile_a * 32 + (lane & 31);
  const float th = a < na ? theta[a] : -INFINITY;      // rows past the end take nothing
  int *cnt = cand_n + (a < na ? a : 0);
  int *out = cand_i + (size_t)(a < na ? a : 0) * kFilterCap;
  const int part = blockIdx.y, stride = kSlices * (int)gridDim.y;
  const int c0 = part * kSlices + slice, c1 = nb_tiles;
  float bf[kSteps], bn[kSteps];
  if (c0 < c1) {
#pragma unroll
    for (int s = 0; s < kSteps; ++s) bf[s] = Bp[((size_t)c0 * kSteps + s) * 64 + lane];
  }
  // A tile's passing values are claimed with ONE atomic per lane (their count) and written out an iteration later,
  // after the next tile's MFMA chain: the atomic's round trip is never waited for.
  unsigned pend_mask = 0u;
  int pend_pos = 0, pend_base = 0;
  auto flush = [&]() {
    unsigned m = pend_mask;
    int pos = pend_pos;
    while (m) {
      const int r = __ffs((int)m) - 1;
      m &= m - 1u;
      if (pos < kFilterCap) out[pos] = pend_base + (r & 3) + 8 * (r >> 2);
      ++pos;
    }
  };
  for (int c = c0; c < c1; c += stride) {
    const int cn = (c + stride < c1) ? c + stride : c;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) bn[s] = Bp[((size_t)cn * kSteps + s) * 64 + lane];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
    if (__any(pend_mask != 0u)) flush();
    const int rbase = c * 32 + 4 * (lane >> 5);
    unsigned mask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) mask |= (acc[r] < th && rbase + (r & 3) + 8 * (r >> 2) < nb) ? (1u << r) : 0u;
    pend_mask = mask;
    pend_base = rbase;
    if (mask) pend_pos = atomicAdd(cnt, (int)__popc(mask));
#pragma unroll
    for (int s = 0; s < kSteps; ++s) bf[s] = bn[s];
  }
  flush();
}

// One wave per query row: exact distances of the filter's candidates, the k best by (distance, index), and the
// certificate against theta (see k_knn_rerank for the bound).
template <int kD>
__global__ void __launch_bounds__(256)
k_knn_rerank_filter(const float *__restrict__ A, int na, const float *__restrict__ B, int nb, int k, const int *__restrict__ cand_n,
                    const int *__restrict__ cand_i, const float *__restrict__ theta, const float *__restrict__ colsum, float inv_nb,
                    const uint32_t *__restrict__ nsort, const uint32_t *__restrict__ nperm, int *__restrict__ idx, float *__restrict__ d2out,
                    int *__restrict__ fb_count)
{
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (a >= na) return;               // wave-uniform
  float x[kD];
  float na2 = 0.0f;
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    x[d] = A[(size_t)a * kD + d];
    const float xc = x[d] - colsum[d] * inv_nb;
    na2 = fmaf(xc, xc, na2);         // |a - mu|^2
  }
  float bd[kMaxK];
  int bi[kMaxK];
#pragma unroll
  for (int s = 0; s < kMaxK; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
  const int found = cand_n[a];
  const int n_cand = found < kFilterCap ? found : kFilterCap;
  const int *ci_ = cand_i + (size_t)a * kFilterCap;
  for (int e = lane; e < n_cand; e += kWave) {
    const int j = ci_[e];
    const float *b = B + (size_t)j * kD;
    float r = 0.0f;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float df = x[d] - b[d];
      r = __fadd_rn(r, __fmul_rn(df, df));
    }
    if (r < bd[kMaxK - 1] || (r == bd[kMaxK - 1] && j < bi[kMaxK - 1])) {
      float cd = r;
      int ci = j;
      bool carrying = false;
#pragma unroll
      for (int t = 0; t < kMaxK; ++t) {
        const bool sw = carrying || cd < bd[t] || (cd == bd[t] && ci < bi[t]);
        carrying = sw;
        const float td = bd[t];
        const int ti = bi[t];
        bd[t] = sw ? cd : td; bi[t] = sw ? ci : ti;
        cd = sw ? td : cd; ci = sw ? ti : ci;
      }
    }
  }
  float kth = INFINITY;
  for (int o = 0; o < k; ++o) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(bd[0]) << 32) | (unsigned)bi[0];
    const unsigned long long best = wave_min_u64(key);
    const float d = __uint_as_float((unsigned)(best >> 32));
    if (lane == 0) {
      idx[(size_t)a * k + o] = d < INFINITY ? (int)(unsigned)(best & 0xffffffffull) : -1;
      d2out[(size_t)a * k + o] = d;
    }
    if (o == k - 1) kth = d;
    if (key == best && bd[0] < INFINITY) {
#pragma unroll
      for (int s = 0; s + 1 < kMaxK; ++s) { bd[s] = bd[s + 1]; bi[s] = bi[s + 1]; }
      bd[kMaxK - 1] = INFINITY; bi[kMaxK - 1] = 0x7fffffff;
    }
  }
  // every target outside the candidates has approx >= theta: k_knn_rerank's certificate with tau = theta
  const float tau = theta[a];
  const float rho = (sqrtf(na2) + sqrtf(kth)) * 1.001f + 1e-3f;
  const float eps = (2e-5f * (na2 + rho * rho) + 1e-5f * kth) * ((float)knn_kp(kD) / 36.0f);
  const bool certified = found <= kFilterCap && (!(tau < INFINITY) ? found >= nb : (kth < tau - eps));
  // A row without a certificate (wave-uniform, rare) is searched exactly right here -- the targets whose norm lies within
  // sqrt(kth) of the row's own, k_knn_exact_range's argument -- instead of being listed for a launch of its own: that
  // launch would follow EVERY search and, with nothing to do, still queue for registers behind the other streams'
  // kernels (68 us of stream time per pair on the 16-stream bench).
  if (!certified) {
    if (lane == 0) atomicAdd(fb_count, 1);           // (statistics: mm3d_debug_knn_fallback_rows)
    knn_exact_range_row<kD>(x, kth, a, B, nb, k, nsort, nperm, lane, idx, d2out);
  }
}

// ---------------------------------------------------------------- stage 3: exact re-rank + certificate
// One WAVE per query row: the lanes split the row's candidates (n_lists x kListLen of them: 64 for a
// full table, several hundred when the targets were split over parts), each re-ranks its share with
// FLANN's accumulation order into a private sorted list, and the k winners are drawn by repeated
// wave-wide minimum over (distance bits, index) keys -- i.e. ties go to the lower index.
template <int kD>
__global__ void __launch_bounds__(256)
k_knn_rerank(const float *__restrict__ A, int na, const float *__restrict__ B, int nb, int k, int n_lists,
             const float *__restrict__ cand_d, const int *__restrict__ cand_i, const float *__restrict__ colsum, float inv_nb,
             int *__restrict__ idx, float *__restrict__ d2out, int *__restrict__ fb_rows, int *__restrict__ fb_count)
{
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (a >= na) return;               // wave-uniform
  float x[kD];
  float na2 = 0.0f;
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    x[d] = A[(size_t)a * kD + d];
    const float xc = x[d] - colsum[d] * inv_nb;
    na2 = fmaf(xc, xc, na2);         // |a - mu|^2
  }
  float bd[kMaxK];
  int bi[kMaxK];
#pragma unroll
  for (int s = 0; s < kMaxK; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
  float tau = INFINITY;
  const int n_cand = n_lists * kListLen;
  const float *cd_ = cand_d + (size_t)a * n_cand;
  const int *ci_ = cand_i + (size_t)a * n_cand;
  for (int e = lane; e < n_cand; e += kWave) {
    const int j = ci_[e];
    if (j < 0) continue;
    // a full list hides targets whose approximate distance is >= its worst (last) entry
    if ((e & (kListLen - 1)) == kListLen - 1) tau = fminf(tau, cd_[e]);
    const float *b = B + (size_t)j * kD;
    float r = 0.0f;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float df = x[d] - b[d];
      r = __fadd_rn(r, __fmul_rn(df, df));
    }
    if (r < bd[kMaxK - 1] || (r == bd[kMaxK - 1] && j < bi[kMaxK - 1])) {
      float cd = r;
      int ci = j;
      bool carrying = false;
#pragma unroll
      for (int t = 0; t < kMaxK; ++t) {
        const bool sw = carrying || cd < bd[t] || (cd == bd[t] && ci < bi[t]);
        carrying = sw;
        const float td = bd[t];
        const int ti = bi[t];
        bd[t] = sw ? cd : td; bi[t] = sw ? ci : ti;
        cd = sw ? td : cd; ci = sw ? ti : ci;
      }
    }
  }
  tau = wave_min_f(tau);
  // merge the 64 sorted lists (the lists partition the targets, so every index occurs once)
  float kth = INFINITY;
  for (int o = 0; o < k; ++o) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(bd[0]) << 32) | (unsigned)bi[0];
    unsigned long long best = key;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const unsigned long long other = __shfl_xor(best, s, kWave);
      best = other < best ? other : best;
    }
    const float d = __uint_as_float((unsigned)(best >> 32));
    if (lane == 0) {
      idx[(size_t)a * k + o] = d < INFINITY ? (int)(unsigned)(best & 0xffffffffull) : -1;
      d2out[(size_t)a * k + o] = d;
    }
    if (o == k - 1) kth = d;
    if (key == best && bd[0] < INFINITY) {
#pragma unroll
      for (int s = 0; s + 1 < kMaxK; ++s) { bd[s] = bd[s + 1]; bi[s] = bi[s + 1]; }
      bd[kMaxK - 1] = INFINITY; bi[kMaxK - 1] = 0x7fffffff;
    }
  }
  // Certificate.  A target b outside the candidate lists has approx(b) >= tau.  If |b - mu| > rho :=
  // |a - mu| + sqrt(kth) (plus slack) then |a - b|^2 > kth by the triangle inequality, so only
  // targets with |b - mu| <= rho matter, and for those the expansion's rounding error is bounded by
  // ~5e-6 (|a-mu| + |b-mu|)^2 + 2e-6 kth  (36-term f32 FMA chain on the centred operands, the two
  // norms, the centring itself, and the 33-term exact sum).  2e-5 (|a-mu|^2 + rho^2) + 1e-5 kth
  // doubles that bound.
  const float rho = (sqrtf(na2) + sqrtf(kth)) * 1.001f + 1e-3f;
  // (stated for the 36-term chain of FPFH; the bound grows linearly with the chain length)
  const float eps = (2e-5f * (na2 + rho * rho) + 1e-5f * kth) * ((float)knn_kp(kD) / 36.0f);
  const bool certified = !(tau < INFINITY) || (kth < tau - eps);
  if (!certified && lane == 0) fb_rows[atomicAdd(fb_count, 1)] = a;
}

// ---------------------------------------------------------------- exact fallback on a norm range
// |a - b| >= | |a| - |b| |: a target can only be among the k nearest of a if its norm lies within sqrt(U) of
// |a|, U = the k-th exact distance the re-rank already found (an upper bound of the true k-th distance).
// The target set keeps its norms sorted (knn_norm_order), so the candidates of a row are ONE contiguous
// range of that order: for rows that fail the certificate because hundreds of near-identical targets
// crowd their lists (PFH rows of flat ground) the range is those near-duplicates, not all targets.
// Norms are float fmaf chains (relative error <= (D + 2) 2^-24); the range is widened by 1e-4 (|a| + reach).
template <int kD>
__global__ void k_knn_norms(const float *__restrict__ X, int n, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.0f;
  for (int d = 0; d < kD; ++d) { const float v = X[(size_t)i * kD + d]; s = fmaf(v, v, s); }
  keys[i] = __float_as_uint(sqrtf(s));      // >= 0 (or NaN: sorts last, never in a range): the bits order like the value
  vals[i] = (uint32_t)i;
}

// one row, one wave: x = the row (in every lane's registers), U = an upper bound of its k-th distance
template <int kD>
__device__ __forceinline__ void knn_exact_range_row(const float (&x)[kD], float U, int row, const float *__restrict__ B, int nb, int k,
                                                    const uint32_t *__restrict__ nsort, const uint32_t *__restrict__ nperm, int lane,
                                                    int *__restrict__ idx, float *__restrict__ d2out)
{
  float s = 0.0f;
#pragma unroll
  for (int d = 0; d < kD; ++d) s = fmaf(x[d], x[d], s);
  const float na = sqrtf(s);
  int lo = 0, hi = nb;
  if (U < INFINITY && na == na) {
    const float reach = sqrtf(U);
    const float slack = 1e-4f * (na + reach) + 1e-6f;
    const float lo_v = na - reach - slack, hi_v = na + reach + slack;
    // first index with norm >= lo_v / first index with norm > hi_v (NaN norms sit at the end and compare false)
    int a = 0, b = nb;
    while (a < b) { const int m = (a + b) >> 1; if (__uint_as_float(nsort[m]) >= lo_v) b = m; else a = m + 1; }
    lo = a;
    a = lo; b = nb;
    while (a < b) { const int m = (a + b) >> 1; if (__uint_as_float(nsort[m]) > hi_v) b = m; else a = m + 1; }
    hi = a;
  }
  float bd[kMaxK];
  int bi[kMaxK];
#pragma unroll
  for (int t = 0; t < kMaxK; ++t) { bd[t] = INFINITY; bi[t] = 0x7fffffff; }
  for (int t = lo + lane; t < hi; t += kWave) {
    const int j = (int)nperm[t];
    const float *bp = B + (size_t)j * kD;
    float r = 0.0f;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float df = x[d] - bp[d];
      r = __fadd_rn(r, __fmul_rn(df, df));
    }
    if (r < bd[kMaxK - 1] || (r == bd[kMaxK - 1] && j < bi[kMaxK - 1])) {
      float cd = r;
      int ci = j;
      bool carrying = false;
#pragma unroll
      for (int t2 = 0; t2 < kMaxK; ++t2) {
        const bool sw = carrying || cd < bd[t2] || (cd == bd[t2] && ci < bi[t2]);
        carrying = sw;
        const float td = bd[t2];
        const int ti = bi[t2];
        bd[t2] = sw ? cd : td; bi[t2] = sw ? ci : ti;
        cd = sw ? td : cd; ci = sw ? ti : ci;
      }
    }
  }
  for (int o = 0; o < k; ++o) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(bd[0]) << 32) | (unsigned)bi[0];
    unsigned long long best = key;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      const unsigned long long other = __shfl_xor(best, sft, kWave);
      best = other < best ? other : best;
    }
    const float d = __uint_as_float((unsigned)(best >> 32));
    if (lane == 0) {
      idx[(size_t)row * k + o] = d < INFINITY ? (int)(unsigned)(best & 0xffffffffull) : -1;
      d2out[(size_t)row * k + o] = d;
    }
    if (key == best && bd[0] < INFINITY) {
#pragma unroll
      for (int t = 0; t + 1 < kMaxK; ++t) { bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; }
      bd[kMaxK - 1] = INFINITY; bi[kMaxK - 1] = 0x7fffffff;
    }
  }
}

template <int kD>
__global__ void __launch_bounds__(256)
k_knn_exact_range(const float *__restrict__ A, const float *__restrict__ B, int nb, int k, const int *__restrict__ rows,
                  const int *__restrict__ nrows_dev, const uint32_t *__restrict__ nsort, const uint32_t *__restrict__ nperm,
                  int *__restrict__ idx, float *__restrict__ d2out)
{
  const int nrows = *nrows_dev;
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
  for (int slot = wave0; slot < nrows; slot += nwaves) {          // wave-uniform
    const int row = rows[slot];
    float x[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) x[d] = A[(size_t)row * kD + d];
    const float U = d2out[(size_t)row * k + k - 1];
    knn_exact_range_row<kD>(x, U, row, B, nb, k, nsort, nperm, lane, idx, d2out);
  }
}

// ================================================================ wide rows (SHOT1344)
// The narrow kernels above keep a whole row in registers; a 1344-float row does not fit, so the wide
// path streams the contraction dimension instead.  Same three stages, same candidate format.

__global__ void k_knn_colsum_wide(const float *__restrict__ X, int n, int dim, float *__restrict__ sum)
{
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= dim) return;
  float acc = 0.0f;
  for (int r = blockIdx.y; r < n; r += gridDim.y) acc += X[(size_t)r * dim + d];
  atomicAdd(&sum[d], acc);
}

// block = kWideQT tiles of 32 queries x (its share of) the target tiles; each of the 4 waves owns a
// target tile at a time and keeps kWideQT accumulators, so one target operand load feeds kWideQT
// MFMAs; the query operands are shared by the 4 waves through L1.  Operands stream from global
// memory in MFMA order (one coalesced 256-byte wave load per step and tile), two groups of 4 steps
// in flight.
constexpr int kWideQT = 4;

template <int kD>
__global__ void __launch_bounds__(256)
k_knn_mfma_wide(const float *__restrict__ Ap, int na, int na_tiles, const float *__restrict__ Bp, int nb, int nb_tiles,
                float *__restrict__ cand_d, int *__restrict__ cand_i)
{
  constexpr int kSteps = knn_kp(kD) / 2;
  static_assert(kSteps % 4 == 0, "step loop is unrolled by 4");
  const int lane = threadIdx.x & 63;
  const int slice = threadIdx.x >> 6;
  const int tile_a0 = blockIdx.x * kWideQT;
  const float *ap[kWideQT];
#pragma unroll
  for (int q = 0; q < kWideQT; ++q) ap[q] = Ap + (size_t)min(tile_a0 + q, na_tiles - 1) * kSteps * 64 + lane;
  float ld[kWideQT][kListLen];
  int li[kWideQT][kListLen];
#pragma unroll
  for (int q = 0; q < kWideQT; ++q)
#pragma unroll
    for (int s = 0; s < kListLen; ++s) { ld[q][s] = INFINITY; li[q][s] = -1; }
  const int part = blockIdx.y, stride = kSlices * (int)gridDim.y;
  for (int c = part * kSlices + slice; c < nb_tiles; c += stride) {
    const float *bp = Bp + (size_t)c * kSteps * 64 + lane;
    f32x16 acc[kWideQT];
#pragma unroll
    for (int q = 0; q < kWideQT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    float bcur[4], acur[kWideQT][4], bnxt[4], anxt[kWideQT][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bcur[u] = bp[u * 64];
#pragma unroll
      for (int q = 0; q < kWideQT; ++q) acur[q][u] = ap[q][u * 64];
    }
    for (int s0 = 0; s0 < kSteps; s0 += 4) {
      const int sn = s0 + 4 < kSteps ? s0 + 4 : s0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bnxt[u] = bp[(size_t)(sn + u) * 64];
#pragma unroll
        for (int q = 0; q < kWideQT; ++q) anxt[q][u] = ap[q][(size_t)(sn + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < kWideQT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(bcur[u], acur[q][u], acc[q], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bcur[u] = bnxt[u];
#pragma unroll
        for (int q = 0; q < kWideQT; ++q) acur[q][u] = anxt[q][u];
      }
    }
    const int rbase = c * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < kWideQT; ++q) {
      float tmin = acc[q][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[q][r]);
      if (__any(tmin < ld[q][kListLen - 1]))
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[q][r];
          if (v < ld[q][kListLen - 1]) {
            float cd = v;
            int ci = rbase + (r & 3) + 8 * (r >> 2);
            bool carrying = false;
#pragma unroll
            for (int s = 0; s < kListLen; ++s) {
              const bool sw = carrying || cd < ld[q][s];
              carrying = sw;
              const float td = ld[q][s];
              const int ti = li[q][s];
              ld[q][s] = sw ? cd : td; li[q][s] = sw ? ci : ti;
              cd = sw ? td : cd; ci = sw ? ti : ci;
            }
          }
        }
    }
  }
  const int n_lists = kLists * (int)gridDim.y;
  const int list = part * kLists + (lane >> 5) * kSlices + slice;
#pragma unroll
  for (int q = 0; q < kWideQT; ++q) {
    const int a = (tile_a0 + q) * 32 + (lane & 31);
    if (a >= na) continue;
    float *od = cand_d + ((size_t)a * n_lists + list) * kListLen;
    int *oi = cand_i + ((size_t)a * n_lists + list) * kListLen;
#pragma unroll
    for (int s = 0; s < kListLen; ++s) {
      const bool real = li[q][s] >= 0 && li[q][s] < nb;
      od[s] = real ? ld[q][s] : INFINITY;
      oi[s] = real ? li[q][s] : -1;
    }
  }
}

// ---- the wide selector on split-bf16 MFMA (round 6) ------------------------------------------------------------------
// k_knn_mfma_wide only SELECTS candidates for the exact re-rank below, and it did so on the f32 matrix cores (157 TF/s
// peak) of a part whose bf16 cores do 2.5 PF/s.  A float splits into two bf16 halves, x = hi + lo + r with |r| <= 2^-18 |x|,
// and a product into hi hi + hi lo + lo hi (+ terms of 3 * 2^-18 |x y| at most): three v_mfma_f32_32x32x16_bf16 per 16
// contraction steps instead of eight v_mfma_f32_32x32x2f32, at a sixteenth of the time each -- and the same operand bytes
// (two bf16 per float).  The error of the approximation against the f32 expansion, <= 3 * 2^-18 |a - mu| |2 (b - mu)| <=
// 1.2e-5 (|a - mu|^2 + rho^2), is added (doubled and rounded up: 5e-5) to the certificate's epsilon (k_knn_rerank_wide:
// eps_add); the squared norms ride in three bf16 pieces each against exact ones, so they lose nothing.
// Operand layout: Xp[((tile * kChunks + chunk) * 64 + lane) * 2 + {0: hi, 1: lo}] = 8 bf16 (16 bytes): row tile * 32 +
// (lane & 31), contraction indices 16 chunk + 8 (lane >> 5) + 0..7 -- the A / B register layout of the instruction.
constexpr int knn_kp16(int d) { return (d + 6 + 15) / 16 * 16; }      // + three norm pieces + three ones
typedef __bf16 knn_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned knn_bf16_rn(float x)               // round to nearest even (finite inputs)
{
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float knn_bf16_f(unsigned h) { return __uint_as_float(h << 16); }

template <int kD>
__global__ void k_knn_rownorm_c(const float *__restrict__ X, int n, const float *__restrict__ colsum, float inv_nb, float *__restrict__ nrm)
{
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float acc = 0.0f;
  for (int d = lane; d < kD; d += kWave) { const float xc = X[(size_t)row * kD + d] - colsum[d] * inv_nb; acc = fmaf(xc, xc, acc); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if (lane == 0) nrm[row] = acc;
}

template <int kD>
__global__ void k_knn_prep_bf(const float *__restrict__ X, int n, int ntiles, int is_target, const float *__restrict__ colsum, float inv_nb,
                              const float *__restrict__ nrm, uint4 *__restrict__ Xp)
{
  constexpr int kChunks = knn_kp16(kD) / 16;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)ntiles * kChunks * 64) return;
  const int lane = (int)(e & 63);
  const int chunk = (int)((e >> 6) % kChunks);
  const int tile = (int)(e / (64 * kChunks));
  const int row = tile * 32 + (lane & 31);
  const int k0 = 16 * chunk + 8 * (lane >> 5);
  unsigned hi[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = k0 + j;
    float v = 0.0f;
    bool exact_piece = false;
    if (row < n) {
      if (kk < kD) {
        const float xc = X[(size_t)row * kD + kk] - colsum[kk] * inv_nb;
        v = is_target ? -2.0f * xc : xc;
      } else if (kk < kD + 6) {
        // query: [norm pieces x 3, 1, 1, 1]; target: [1, 1, 1, norm pieces x 3]
        const int slot = kk - kD;
        const bool norm_slot = is_target ? slot >= 3 : slot < 3;
        if (norm_slot) {
          const float nv = nrm[row];
          const float p0 = knn_bf16_f(knn_bf16_rn(nv)), p1 = knn_bf16_f(knn_bf16_rn(nv - p0)), p2 = knn_bf16_f(knn_bf16_rn((nv - p0) - p1));
          const int piece = slot % 3;
          v = piece == 0 ? p0 : (piece == 1 ? p1 : p2);
        } else {
          v = 1.0f;
        }
        exact_piece = true;
      }
    } else if (is_target) {
      v = (kk == kD + 3) ? 1e30f : 0.0f;            // padding targets: a'.b' = 1e30, never ahead of a real row
      exact_piece = true;
    }
    const unsigned h = knn_bf16_rn(v);
    hi[j] = h;
    lo[j] = exact_piece ? 0u : knn_bf16_rn(v - knn_bf16_f(h));
  }
  uint4 H, L;
  H.x = hi[0] | (hi[1] << 16); H.y = hi[2] | (hi[3] << 16); H.z = hi[4] | (hi[5] << 16); H.w = hi[6] | (hi[7] << 16);
  L.x = lo[0] | (lo[1] << 16); L.y = lo[2] | (lo[3] << 16); L.z = lo[4] | (lo[5] << 16); L.w = lo[6] | (lo[7] << 16);
  Xp[e * 2] = H;
  Xp[e * 2 + 1] = L;
}

union KnnBfReg { uint4 u; knn_bf16x8 v; };

#ifndef MM3D_KNN_BF_WPE
#define MM3D_KNN_BF_WPE 1
#endif
template <int kD>
__global__ void __launch_bounds__(256, MM3D_KNN_BF_WPE)
k_knn_mfma_wide_bf(const uint4 *__restrict__ Ap, int na, int na_tiles, const uint4 *__restrict__ Bp, int nb, int nb_tiles,
                   float *__restrict__ cand_d, int *__restrict__ cand_i)
{
  constexpr int kChunks = knn_kp16(kD) / 16;
  const int lane = threadIdx.x & 63;
  const int slice = threadIdx.x >> 6;
  const int tile_a0 = blockIdx.x * kWideQT;
  const uint4 *ap[kWideQT];
#pragma unroll
  for (int q = 0; q < kWideQT; ++q) ap[q] = Ap + ((size_t)min(tile_a0 + q, na_tiles - 1) * kChunks * 64 + lane) * 2;
  float ld[kWideQT][kListLen];
  int li[kWideQT][kListLen];
#pragma unroll
  for (int q = 0; q < kWideQT; ++q)
#pragma unroll
    for (int s = 0; s < kListLen; ++s) { ld[q][s] = INFINITY; li[q][s] = -1; }
  // TWO target tiles per wave and pass (round 6: the query operands of a chunk feed eight MFMA chains instead of four -- the
  // kernel was bound by the L1 traffic of its operands, 10 KB per wave and chunk for twelve instructions)
  const int part = blockIdx.y, stride = kSlices * (int)gridDim.y;
  for (int c0 = part * kSlices + slice; c0 < nb_tiles; c0 += 2 * stride) {
    const int c1 = c0 + stride;
    const bool two = c1 < nb_tiles;                       // wave-uniform
    const uint4 *bp0 = Bp + ((size_t)c0 * kChunks * 64 + lane) * 2;
    const uint4 *bp1 = Bp + ((size_t)(two ? c1 : c0) * kChunks * 64 + lane) * 2;
    f32x16 acc[2][kWideQT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < kWideQT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][q][r] = 0.0f;
    // two operand sets in turn (no register copies): set 1 - p of chunk ch + 1 is in flight while set p of chunk ch is multiplied
    KnnBfReg bh[2][2], bl[2][2], ah[2][kWideQT], al[2][kWideQT];
    auto load_set = [&](int p, int ch) {
      bh[p][0].u = bp0[(size_t)ch * 128]; bl[p][0].u = bp0[(size_t)ch * 128 + 1];
      bh[p][1].u = bp1[(size_t)ch * 128]; bl[p][1].u = bp1[(size_t)ch * 128 + 1];
#pragma unroll
      for (int q = 0; q < kWideQT; ++q) { ah[p][q].u = ap[q][(size_t)ch * 128]; al[p][q].u = ap[q][(size_t)ch * 128 + 1]; }
    };
    auto mul_set = [&](int p) {
      // targets as rows (first operand), queries as columns; the three products of an accumulator are eight instructions apart
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < kWideQT; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[p][t].v, ah[p][q].v, acc[t][q], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < kWideQT; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[p][t].v, al[p][q].v, acc[t][q], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < kWideQT; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[p][t].v, ah[p][q].v, acc[t][q], 0, 0, 0);
    };
    load_set(0, 0);
    int ch = 0;
    for (; ch + 2 <= kChunks; ch += 2) {
      load_set(1, ch + 1);
      mul_set(0);
      load_set(0, ch + 2 < kChunks ? ch + 2 : ch + 1);
      mul_set(1);
    }
    if (ch < kChunks) mul_set(0);                  // an odd number of chunks: the last one sits in set 0
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && !two) break;
      const int rbase = (t == 0 ? c0 : c1) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int q = 0; q < kWideQT; ++q) {
        float tmin = acc[t][q][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[t][q][r]);
        if (__any(tmin < ld[q][kListLen - 1]))
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[t][q][r];
            if (v < ld[q][kListLen - 1]) {
              float cd = v;
              int ci = rbase + (r & 3) + 8 * (r >> 2);
              bool carrying = false;
#pragma unroll
              for (int s2 = 0; s2 < kListLen; ++s2) {
                const bool sw = carrying || cd < ld[q][s2];
                carrying = sw;
                const float td = ld[q][s2];
                const int ti = li[q][s2];
                ld[q][s2] = sw ? cd : td; li[q][s2] = sw ? ci : ti;
                cd = sw ? td : cd; ci = sw ? ti : ci;
              }
            }
          }
      }
    }
  }
  const int n_lists = kLists * (int)gridDim.y;
  const int list = part * kLists + (lane >> 5) * kSlices + slice;
#pragma unroll
  for (int q = 0; q < kWideQT; ++q) {
    const int a = (tile_a0 + q) * 32 + (lane & 31);
    if (a >= na) continue;
    float *od = cand_d + ((size_t)a * n_lists + list) * kListLen;
    int *oi = cand_i + ((size_t)a * n_lists + list) * kListLen;
#pragma unroll
    for (int s = 0; s < kListLen; ++s) {
      const bool real = li[q][s] >= 0 && li[q][s] < nb;
      od[s] = real ? ld[q][s] : INFINITY;
      oi[s] = real ? li[q][s] : -1;
    }
  }
}

// FLANN's L2_Simple between the row staged in LDS (broadcast reads) and target row b (this lane's);
// rows are read 4 floats at a time when their length allows 16-byte alignment (1344), else 2 (250)
template <int kD>
__device__ __forceinline__ float knn_wide_dist(const float *__restrict__ x /* LDS */, const float *__restrict__ b)
{
  float r = 0.0f;
  if constexpr (kD % 4 == 0) {
    const float4 *b4 = (const float4 *)b;
    const float4 *x4 = (const float4 *)x;
#pragma unroll 4
    for (int d = 0; d < kD / 4; ++d) {
      const float4 bv = b4[d], xv = x4[d];
      float df = xv.x - bv.x; r = __fadd_rn(r, __fmul_rn(df, df));
      df = xv.y - bv.y; r = __fadd_rn(r, __fmul_rn(df, df));
      df = xv.z - bv.z; r = __fadd_rn(r, __fmul_rn(df, df));
      df = xv.w - bv.w; r = __fadd_rn(r, __fmul_rn(df, df));
    }
  } else if constexpr (kD % 2 == 0) {
    const float2 *b2 = (const float2 *)b;
    const float2 *x2 = (const float2 *)x;
#pragma unroll 4
    for (int d = 0; d < kD / 2; ++d) {
      const float2 bv = b2[d], xv = x2[d];
      float df = xv.x - bv.x; r = __fadd_rn(r, __fmul_rn(df, df));
      df = xv.y - bv.y; r = __fadd_rn(r, __fmul_rn(df, df));
    }
  } else {
#pragma unroll 5
    for (int d = 0; d < kD; ++d) {
      const float df = x[d] - b[d];
      r = __fadd_rn(r, __fmul_rn(df, df));
    }
  }
  return r;
}

// order-preserving map of a float's bits (approximate distances can come out slightly negative)
__device__ __forceinline__ unsigned knn_ordered_bits(float v)
{
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const unsigned long long o = __shfl_xor(v, s, kWave);
    v = o < v ? o : v;
  }
  return v;
}

// One wave per query row.  An exact distance costs a 1344-term chain per candidate, so only the
// candidates that can still be among the k nearest are re-ranked:
//   1. the k candidates with the smallest approximate distance are re-ranked first; the largest of
//      their exact distances, U, bounds the k-th exact distance from above;
//   2. a candidate b of the true k nearest has exact(b) <= U, hence |b - mu| <= rho(U) and
//      approx(b) <= U + eps(rho(U)): every candidate under that threshold is re-ranked, nobody else;
//   3. merge by (distance, index) and certify against the lists' worst kept entries as above.
template <int kD>
__global__ void __launch_bounds__(256)
k_knn_rerank_wide(const float *__restrict__ A, int na, const float *__restrict__ B, int nb, int k, int n_lists,
                  const float *__restrict__ cand_d, const int *__restrict__ cand_i, const float *__restrict__ colsum, float inv_nb,
                  int *__restrict__ idx, float *__restrict__ d2out, int *__restrict__ fb_rows, int *__restrict__ fb_count,
                  float eps_mul /* 1: the f32 selector */, float eps_add /* x (|a - mu|^2 + rho^2): the split-bf16 selector's own error */)
{
  constexpr int kMaxCand = 64 * 32;                 // parts <= 32
  __shared__ __attribute__((aligned(16))) float s_x[4][kD];
  __shared__ int s_sel[4][kMaxCand];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int a_raw = blockIdx.x * 4 + w;
  const bool live = a_raw < na;
  const int a = live ? a_raw : na - 1;
  float na2 = 0.0f;
  for (int d = lane; d < kD; d += kWave) {
    const float v = A[(size_t)a * kD + d];
    s_x[w][d] = v;
    const float xc = v - colsum[d] * inv_nb;
    na2 = fmaf(xc, xc, na2);
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) na2 += __shfl_xor(na2, s, kWave);
  __syncthreads();
  const int n_cand = n_lists * kListLen;            // a multiple of 64
  const float *cd_ = cand_d + (size_t)a * n_cand;
  const int *ci_ = cand_i + (size_t)a * n_cand;
  // 1. tau, and the k best approximate candidates (k rounds of wave-wide minimum over (approx, slot) keys)
  float tau = INFINITY;
  unsigned long long cur = ~0ull;
  for (int e = lane; e < n_cand; e += kWave) {
    if (ci_[e] < 0) continue;
    const float ad = cd_[e];
    if ((e & (kListLen - 1)) == kListLen - 1) tau = fminf(tau, ad);
    const unsigned long long key = ((unsigned long long)knn_ordered_bits(ad) << 32) | (unsigned)e;
    cur = key < cur ? key : cur;
  }
  tau = wave_min_f(tau);
  int my_first = -1;                                // lane o < k: slot of the o-th best approximate candidate
  for (int o = 0; o < k; ++o) {
    const unsigned long long best = wave_min_u64(cur);
    if (best == ~0ull) break;
    if (lane == o) my_first = (int)(best & 0xffffffffull);
    if (cur == best) {                              // the owner moves on to its next candidate
      unsigned long long nxt = ~0ull;
      for (int e = lane; e < n_cand; e += kWave) {
        if (ci_[e] < 0) continue;
        const unsigned long long key = ((unsigned long long)knn_ordered_bits(cd_[e]) << 32) | (unsigned)e;
        if (key > best && key < nxt) nxt = key;
      }
      cur = nxt;
    }
  }
  float first_d = -INFINITY;
  if (my_first >= 0) first_d = knn_wide_dist<kD>(s_x[w], B + (size_t)ci_[my_first] * kD);
  const int n_first = __popcll(ballot(my_first >= 0));
  const float U = n_first >= k ? wave_max_f(first_d) : INFINITY;
  // 2. every candidate that can still be among the k nearest
  const float rho_u = (sqrtf(na2) + sqrtf(U)) * 1.001f + 1e-3f;
  const float eps_u = (2e-5f * (na2 + rho_u * rho_u) + 1e-5f * U) * ((float)knn_kp(kD) / 36.0f) * eps_mul + eps_add * (na2 + rho_u * rho_u);
  const float thr = U + eps_u;                      // +inf when fewer than k candidates exist
  int n_sel = 0;
  for (int e0 = 0; e0 < n_cand; e0 += kWave) {
    const int e = e0 + lane;
    const bool take = ci_[e] >= 0 && !(cd_[e] > thr);
    const unsigned long long mask = ballot(take);
    if (take) s_sel[w][n_sel + __popcll(mask & ((1ull << lane) - 1ull))] = e;
    n_sel += __popcll(mask);
  }
  __syncthreads();
  float bd[kMaxK];
  int bi[kMaxK];
#pragma unroll
  for (int s = 0; s < kMaxK; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
  for (int t = lane; t < n_sel; t += kWave) {
    const int j = ci_[s_sel[w][t]];
    const float r = knn_wide_dist<kD>(s_x[w], B + (size_t)j * kD);
    if (r < bd[kMaxK - 1] || (r == bd[kMaxK - 1] && j < bi[kMaxK - 1])) {
      float cd = r;
      int ci = j;
      bool carrying = false;
#pragma unroll
      for (int t2 = 0; t2 < kMaxK; ++t2) {
        const bool sw = carrying || cd < bd[t2] || (cd == bd[t2] && ci < bi[t2]);
        carrying = sw;
        const float td = bd[t2];
        const int ti = bi[t2];
        bd[t2] = sw ? cd : td; bi[t2] = sw ? ci : ti;
        cd = sw ? td : cd; ci = sw ? ti : ci;
      }
    }
  }
  // 3. merge the lanes' lists, certificate
  float kth = INFINITY;
  for (int o = 0; o < k; ++o) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(bd[0]) << 32) | (unsigned)bi[0];
    const unsigned long long best = wave_min_u64(key);
    const float d = __uint_as_float((unsigned)(best >> 32));
    if (lane == 0 && live) {
      idx[(size_t)a * k + o] = d < INFINITY ? (int)(unsigned)(best & 0xffffffffull) : -1;
      d2out[(size_t)a * k + o] = d;
    }
    if (o == k - 1) kth = d;
    if (key == best && bd[0] < INFINITY) {
#pragma unroll
      for (int s = 0; s + 1 < kMaxK; ++s) { bd[s] = bd[s + 1]; bi[s] = bi[s + 1]; }
      bd[kMaxK - 1] = INFINITY; bi[kMaxK - 1] = 0x7fffffff;
    }
  }
  const float rho = (sqrtf(na2) + sqrtf(kth)) * 1.001f + 1e-3f;
  const float eps = (2e-5f * (na2 + rho * rho) + 1e-5f * kth) * ((float)knn_kp(kD) / 36.0f) * eps_mul + eps_add * (na2 + rho * rho);
  const bool certified = !(tau < INFINITY) || (kth < tau - eps);
  if (!certified && lane == 0 && live) fb_rows[atomicAdd(fb_count, 1)] = a;
}

// Exact brute force for the rows without a certificate (and for small problems): one wave owns
// kWideFbRows rows staged in LDS and one interleaved part of the target tiles; a lane takes one
// target per tile and runs the full chains for all the rows (each target float is loaded once per
// kWideFbRows rows), keeping a sorted top-16 per row in registers.  Output: the same per-part key
// lists that k_knn_merge_parts merges.
constexpr int kWideFbRows = 4;

template <int kD>
__global__ void __launch_bounds__(64)
k_knn_exact_wide(const float *__restrict__ A, const float *__restrict__ B, int nb, int k, const int *__restrict__ rows,
                 const int *__restrict__ nrows_dev, int nrows_host, unsigned long long *__restrict__ part_keys)
{
  __shared__ __attribute__((aligned(16))) float s_x[kWideFbRows][kD];
  const int nrows = nrows_dev ? *nrows_dev : nrows_host;
  const int lane = threadIdx.x, part = blockIdx.y;
  const int ntiles = (nb + 63) / 64;
  for (int g = blockIdx.x; g * kWideFbRows < nrows; g += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kWideFbRows; ++r) {
      const int slot = min(g * kWideFbRows + r, nrows - 1);
      const int row = rows ? rows[slot] : slot;
      for (int d = lane; d < kD; d += 64) s_x[r][d] = A[(size_t)row * kD + d];
    }
    __syncthreads();
    float bd[kWideFbRows][kMaxK];
    int bi[kWideFbRows][kMaxK];
#pragma unroll
    for (int r = 0; r < kWideFbRows; ++r)
#pragma unroll
      for (int s = 0; s < kMaxK; ++s) { bd[r][s] = INFINITY; bi[r][s] = 0x7fffffff; }
    for (int tl = part; tl < ntiles; tl += kFbParts) {
      const int j = tl * 64 + lane;
      if (j >= nb) continue;
      float acc[kWideFbRows];
#pragma unroll
      for (int r = 0; r < kWideFbRows; ++r) acc[r] = 0.0f;
      if constexpr (kD % 4 == 0) {
        const float4 *b4 = (const float4 *)(B + (size_t)j * kD);
#pragma unroll 2
        for (int d = 0; d < kD / 4; ++d) {
          const float4 bv = b4[d];
#pragma unroll
          for (int r = 0; r < kWideFbRows; ++r) {
            const float4 xv = ((const float4 *)s_x[r])[d];
            float df = xv.x - bv.x; acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
            df = xv.y - bv.y; acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
            df = xv.z - bv.z; acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
            df = xv.w - bv.w; acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
          }
        }
      } else {
        const float2 *b2 = (const float2 *)(B + (size_t)j * kD);
#pragma unroll 2
        for (int d = 0; d < kD / 2; ++d) {
          const float2 bv = b2[d];
#pragma unroll
          for (int r = 0; r < kWideFbRows; ++r) {
            const float2 xv = ((const float2 *)s_x[r])[d];
            float df = xv.x - bv.x; acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
            df = xv.y - bv.y; acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kWideFbRows; ++r) {
        if (acc[r] < bd[r][kMaxK - 1]) {     // a lane sees its targets in ascending index order: strict < keeps the lower index
          float cd = acc[r];
          int ci = j;
          bool carrying = false;
#pragma unroll
          for (int s = 0; s < kMaxK; ++s) {
            const bool sw = carrying || cd < bd[r][s];
            carrying = sw;
            const float td = bd[r][s];
            const int ti = bi[r][s];
            bd[r][s] = sw ? cd : td; bi[r][s] = sw ? ci : ti;
            cd = sw ? td : cd; ci = sw ? ti : ci;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kWideFbRows; ++r) {
      const int slot = g * kWideFbRows + r;
      for (int o = 0; o < k; ++o) {           // k rounds of wave-wide minimum over the lanes' list heads
        const unsigned long long key = ((unsigned long long)__float_as_uint(bd[r][0]) << 32) | (unsigned)bi[r][0];
        const unsigned long long best = wave_min_u64(key);
        if (lane == 0 && slot < nrows) part_keys[((size_t)slot * kFbParts + part) * kMaxK + o] = best;
        if (key == best && bd[r][0] < INFINITY) {
#pragma unroll
          for (int s = 0; s + 1 < kMaxK; ++s) { bd[r][s] = bd[r][s + 1]; bi[r][s] = bi[r][s + 1]; }
          bd[r][kMaxK - 1] = INFINITY; bi[r][kMaxK - 1] = 0x7fffffff;
        }
      }
    }
  }
}

// column sums and MFMA-ordered operands of a TARGET set
// wide rows (more than 128 floats): the selector runs on split-bf16 MFMA (k_knn_mfma_wide_bf); MM3D_KNN_WIDE_BF16=0 restores
// the f32 one (the A/B).  Read once: the cached target operands of a descriptor set are laid out for one of the two.
static bool knn_wide_bf16()
{
  static const bool v = [] { const char *e = getenv("MM3D_KNN_WIDE_BF16"); return !(e && !atoi(e)); }();
  return v;
}

template <int kD>
static void knn_target_operands(Context *c, const mm3d_desc *B, DevBuf<float> &colsum, DevBuf<float> &Bp)
{
  constexpr int kKP = knn_kp(kD), kSteps = kKP / 2;
  const int nb = (int)B->n, nb_tiles = (nb + 31) / 32;
  constexpr int kCols = kD > 128 ? kD : 128;
  colsum = DevBuf<float>(c, kCols);
  if constexpr (kD > 128) {
    if (knn_wide_bf16()) {
      constexpr int kChunks = knn_kp16(kD) / 16;
      MM3D_HIP(hipMemsetAsync(colsum.get(), 0, kCols * sizeof(float), c->stream));
      MM3D_LAUNCH(c, "desc_knn_prep", nb * kD * 4.0, k_knn_colsum_wide, dim3(div_up(kD, 256), 64), dim3(256), 0, (const float *)B->data.get(), nb,
                  kD, colsum.get());
      DevBuf<float> nrm(c, (size_t)nb);
      MM3D_LAUNCH(c, "desc_knn_prep", nb * kD * 4.0, (k_knn_rownorm_c<kD>), dim3(div_up(nb, 4)), dim3(256), 0, (const float *)B->data.get(), nb,
                  (const float *)colsum.get(), 1.0f / (float)nb, nrm.get());
      Bp = DevBuf<float>(c, (size_t)nb_tiles * kChunks * 64 * 8);          // 32 bytes per lane and chunk
      MM3D_LAUNCH(c, "desc_knn_prep", nb * (kD + kKP) * 4.0, (k_knn_prep_bf<kD>), dim3(div_up((size_t)nb_tiles * kChunks * 64, 256)), dim3(256), 0,
                  (const float *)B->data.get(), nb, nb_tiles, 1, (const float *)colsum.get(), 1.0f / (float)nb, (const float *)nrm.get(),
                  reinterpret_cast<uint4 *>(Bp.get()));
      c->settle();                                   // (nrm goes out of scope)
      return;
    }
  }
  Bp = DevBuf<float>(c, (size_t)nb_tiles * kSteps * 64);
  MM3D_HIP(hipMemsetAsync(colsum.get(), 0, kCols * sizeof(float), c->stream));
  if constexpr (kD <= 128)
    MM3D_LAUNCH(c, "desc_knn_prep", nb * kD * 4.0, (k_knn_colsum<kD>), dim3(64), dim3(256), 0, (const float *)B->data.get(), nb, colsum.get());
  else
    MM3D_LAUNCH(c, "desc_knn_prep", nb * kD * 4.0, k_knn_colsum_wide, dim3(div_up(kD, 256), 64), dim3(256), 0, (const float *)B->data.get(), nb,
                kD, colsum.get());
  MM3D_LAUNCH(c, "desc_knn_prep", nb * (kD + kKP) * 4.0, (k_knn_prep<kD>), dim3(div_up((size_t)nb_tiles * kSteps * 64, 256)), dim3(256), 0,
              (const float *)B->data.get(), nb, nb_tiles, 1, (const float *)colsum.get(), 1.0f / (float)nb, Bp.get());
}

// the target rows' norms, sorted, with their row indices (k_knn_exact_range)
template <int kD>
static void knn_norm_order(Context *c, const mm3d_desc *B, DevBuf<uint32_t> &nsort, DevBuf<uint32_t> &nperm)
{
  const int nb = (int)B->n;
  DevBuf<uint32_t> keys(c, nb), vals(c, nb);
  nsort = DevBuf<uint32_t>(c, nb);
  nperm = DevBuf<uint32_t>(c, nb);
  MM3D_LAUNCH(c, "desc_knn_prep", nb * kD * 4.0, (k_knn_norms<kD>), dim3(div_up(nb, 256)), dim3(256), 0, (const float *)B->data.get(), nb,
              keys.get(), vals.get());
  sort_pairs_u32(c, keys.get(), nsort.get(), vals.get(), nperm.get(), (size_t)nb, 32);
  c->settle();                                 // keys / vals go out of scope
}

void desc_knn_prepare_target(Context *c, const mm3d_desc *B_)
{
  auto *B = const_cast<mm3d_desc *>(B_);
  std::lock_guard<std::mutex> lk(B->cache_mu);
  if (B->n < 64 || B->knn_Bp.get()) return;
  if (B->dim == 33) { knn_target_operands<33>(c, B, B->knn_colsum, B->knn_Bp); knn_norm_order<33>(c, B, B->knn_nsort, B->knn_nperm); }
  else if (B->dim == 2) { knn_target_operands<2>(c, B, B->knn_colsum, B->knn_Bp); knn_norm_order<2>(c, B, B->knn_nsort, B->knn_nperm); }
  else if (B->dim == 125) { knn_target_operands<125>(c, B, B->knn_colsum, B->knn_Bp); knn_norm_order<125>(c, B, B->knn_nsort, B->knn_nperm); }
  else if (B->dim == 250) knn_target_operands<250>(c, B, B->knn_colsum, B->knn_Bp);
  else if (B->dim == 1344) knn_target_operands<1344>(c, B, B->knn_colsum, B->knn_Bp);
  else if (B->dim == 1980) knn_target_operands<1980>(c, B, B->knn_colsum, B->knn_Bp);
  // another context may read the operands the moment the lock is released (a pair estimate without mm3d_map_prepare)
  c->settle();
}

template <int kD>
static void desc_knn_impl(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  constexpr int kKP = knn_kp(kD), kSteps = kKP / 2;
  const int na = (int)A->n, nb = (int)B->n;
  idx = DevBuf<int>(c, (size_t)na * k);
  d2 = DevBuf<float>(c, (size_t)na * k);
  if (na == 0) return;
  const float *Ad = A->data.get(), *Bd = B->data.get();
  if ((double)na * nb < 65536.0 || nb < 64) {
    MM3D_LAUNCH(c, "desc_knn_exact", ((double)na + nb) * kD * 4.0, (k_knn_exact<kD>), dim3(div_up(na, 128)), dim3(128), 0, Ad, na,
                Bd, nb, k, (const int *)nullptr, (const int *)nullptr, na, idx.get(), d2.get());
    return;
  }
  const int na_tiles = (na + 31) / 32, nb_tiles = (nb + 31) / 32;
  DevBuf<float> Ap(c, (size_t)na_tiles * kSteps * 64);
  DevBuf<unsigned> meta(c, 4);   // [1] fallback count
  MM3D_HIP(hipMemsetAsync(meta.get(), 0, 16, c->stream));
  // target-side operands: cached on the descriptor set when it was prepared, else built here
  DevBuf<float> colsum_tmp, Bp_tmp;
  const float *colsum = B->knn_colsum.get(), *Bp = B->knn_Bp.get();
  if (!colsum || !Bp) {
    knn_target_operands<kD>(c, B, colsum_tmp, Bp_tmp);
    colsum = colsum_tmp.get();
    Bp = Bp_tmp.get();
  }
  const float inv_nb = 1.0f / (float)nb;
  MM3D_LAUNCH(c, "desc_knn_prep", na * (kD + kKP) * 4.0, (k_knn_prep<kD>), dim3(div_up((size_t)na_tiles * kSteps * 64, 256)), dim3(256), 0, Ad, na,
              na_tiles, 0, colsum, inv_nb, Ap.get());
  DevBuf<int> fb_rows(c, na);
  // the targets' norms in sorted order (rows without a certificate: exact search over the targets whose norm is within
  // the row's current k-th distance of its own)
  DevBuf<uint32_t> nsort_tmp, nperm_tmp;
  const uint32_t *nsort = B->knn_nsort.get(), *nperm = B->knn_nperm.get();
  if (!nsort || !nperm) {
    knn_norm_order<kD>(c, B, nsort_tmp, nperm_tmp);
    nsort = nsort_tmp.get();
    nperm = nperm_tmp.get();
  }
  bool listed = true;                 // uncertified rows are listed in fb_rows for k_knn_exact_range (the filter path searches them in place)
  // few query tiles (SAC-IA's sampled rows): split the targets over `parts` blocks per query tile so
  // that the launch still has >= 2 blocks per CU
  int parts = 1;
  while (parts < 16 && na_tiles * parts < 512 && nb_tiles / (kSlices * parts * 2) >= 4) parts *= 2;
  static const bool no_filter = getenv("MM3D_KNN_NO_FILTER") != nullptr;     // A/B: the list path for every shape
  if (kD < 64 && nb_tiles >= 64 && k + kThetaExtra <= 32 && !no_filter) {
   if constexpr (kD < 64) {
    // short rows against many targets: thresholds from a sample of the target tiles, then the filtered product
    const int ns_tiles = div_up(nb_tiles, kSampleStep);
    int sparts = 1;
    while (sparts < 4 && na_tiles * sparts < 512 && ns_tiles / (kSlices * sparts * 2) >= 2) sparts *= 2;
    const int s_lists = kLists * sparts;
    DevBuf<float> samp_d(c, (size_t)na * s_lists * kListLen);
    DevBuf<int> samp_i(c, (size_t)na * s_lists * kListLen);
    DevBuf<float> theta(c, (size_t)na);
    DevBuf<int> cand_n(c, (size_t)na);
    DevBuf<int> cand_i(c, (size_t)na * kFilterCap);
    MM3D_HIP(hipMemsetAsync(cand_n.get(), 0, (size_t)na * sizeof(int), c->stream));
    MM3D_LAUNCH(c, "desc_knn_sample", 2.0 * (double)na_tiles * 32 * (double)ns_tiles * 32 * kKP, (k_knn_mfma<kD>), dim3(na_tiles, sparts), dim3(256), 0,
                (const float *)Ap.get(), na, Bp, nb, nb_tiles, kSampleStep, samp_d.get(), samp_i.get());
    static_assert(kLists * 4 * kListLen <= 64 * kThetaPerLane, "k_knn_theta keeps a row's sample candidates in registers");
    MM3D_LAUNCH(c, "desc_knn_theta", na * (double)(s_lists * kListLen * 8), k_knn_theta, dim3(div_up(na, 4)), dim3(256), 0, (const float *)samp_d.get(),
                (const int *)samp_i.get(), na, s_lists * kListLen, k + kThetaExtra, theta.get());
    // roofline unit for this kernel is FLOPs (2 * na * nb * kKP per launch), reported as such by bench.py
    // (no lists to keep: the targets are split until the launch has eight waves per SIMD or a wave is down to four tiles)
    int fparts = 1;
    while (fparts < 64 && na_tiles * fparts < 2048 && nb_tiles / (kSlices * fparts * 2) >= 4) fparts *= 2;
    if (const char *e = getenv("MM3D_KNN_PARTS")) fparts = std::max(1, atoi(e));   // experiment
    MM3D_LAUNCH(c, "desc_knn_mfma", 2.0 * (double)na_tiles * 32 * (double)nb_tiles * 32 * kKP, (k_knn_filter<kD>), dim3(na_tiles, fparts), dim3(256), 0,
                (const float *)Ap.get(), na, Bp, nb, nb_tiles, (const float *)theta.get(), cand_n.get(), cand_i.get());
    MM3D_LAUNCH(c, "desc_knn_rerank", na * (double)((k + kThetaExtra) * kSampleStep * (kD * 4 + 4) + kD * 4), (k_knn_rerank_filter<kD>), dim3(div_up(na, 4)),
                dim3(256), 0, Ad, na, Bd, nb, k, (const int *)cand_n.get(), (const int *)cand_i.get(), (const float *)theta.get(), colsum, inv_nb,
                nsort, nperm, idx.get(), d2.get(), (int *)(meta.get() + 1));
    listed = false;
   }
  } else {
    // each part keeps its own 8 lists per query
    const int n_lists = kLists * parts;
    DevBuf<float> cand_d(c, (size_t)na * n_lists * kListLen);
    DevBuf<int> cand_i(c, (size_t)na * n_lists * kListLen);
    MM3D_LAUNCH(c, "desc_knn_mfma", 2.0 * (double)na_tiles * 32 * (double)nb_tiles * 32 * kKP, (k_knn_mfma<kD>), dim3(na_tiles, parts),
                dim3(256), 0, (const float *)Ap.get(), na, Bp, nb, nb_tiles, 1, cand_d.get(), cand_i.get());
    // short rows (RSD, FPFH): every candidate is re-ranked, a 33-term chain is cheaper than choosing; long rows (PFH):
    // the pruned re-rank of the wide path (k best approximate candidates first, then only what can still matter)
    if constexpr (kD >= 64)
      MM3D_LAUNCH(c, "desc_knn_rerank", na * (double)(n_lists * kListLen * 8 + 32 * kD * 4), (k_knn_rerank_wide<kD>), dim3(div_up(na, 4)), dim3(256), 0,
                  Ad, na, Bd, nb, k, n_lists, (const float *)cand_d.get(), (const int *)cand_i.get(), colsum, inv_nb,
                  idx.get(), d2.get(), fb_rows.get(), (int *)(meta.get() + 1), 1.0f, 0.0f);
    else
      MM3D_LAUNCH(c, "desc_knn_rerank", na * (double)(n_lists * kListLen * (kD * 4 + 8) + kD * 4), (k_knn_rerank<kD>), dim3(div_up(na, 4)), dim3(256), 0,
                  Ad, na, Bd, nb, k, n_lists, (const float *)cand_d.get(), (const int *)cand_i.get(), colsum, inv_nb,
                  idx.get(), d2.get(), fb_rows.get(), (int *)(meta.get() + 1));
  }
  // listed rows (the grid is sized for the worst case; waves beyond the device-side count exit at once)
  const int fb_blocks = div_up(na, 4) < 1024 ? div_up(na, 4) : 1024;
  if (listed)
    MM3D_LAUNCH(c, "desc_knn_fallback", 0.0, (k_knn_exact_range<kD>), dim3(fb_blocks), dim3(256), 0, Ad, Bd, nb, k,
                (const int *)fb_rows.get(), (const int *)(meta.get() + 1), nsort, nperm, idx.get(), d2.get());
  if (c->debug) {
    unsigned *h = (unsigned *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(h, meta.get(), 16, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->knn_fallback_rows += (long long)h[1];
    c->knn_rows += na;
  }
}

// the wide path: same stages as desc_knn_impl with the streaming kernels
template <int kD>
static void desc_knn_wide_impl(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  constexpr int kKP = knn_kp(kD), kSteps = kKP / 2;
  const int na = (int)A->n, nb = (int)B->n;
  idx = DevBuf<int>(c, (size_t)na * k);
  d2 = DevBuf<float>(c, (size_t)na * k);
  if (na == 0) return;
  const float *Ad = A->data.get(), *Bd = B->data.get();
  DevBuf<unsigned long long> part_keys(c, (size_t)na * kFbParts * kMaxK);
  DevBuf<unsigned> meta(c, 4);   // [1] fallback count
  DevBuf<int> fb_rows(c, na);
  const int fb_groups = div_up(na, kWideFbRows);
  if ((double)na * nb < 65536.0 || nb < 64) {
    // small problems: brute force for every row
    MM3D_LAUNCH(c, "iota", na * 4.0, k_knn_iota, dim3(div_up(na, 256)), dim3(256), 0, fb_rows.get(), na, (int *)(meta.get() + 1));
    MM3D_LAUNCH(c, "desc_knn_fallback", ((double)na + nb) * kD * 4.0, (k_knn_exact_wide<kD>), dim3(fb_groups < 64 ? fb_groups : 64, kFbParts),
                dim3(64), 0, Ad, Bd, nb, k, (const int *)fb_rows.get(), (const int *)nullptr, na, part_keys.get());
    MM3D_LAUNCH(c, "desc_knn_fallback", 0.0, k_knn_merge_parts, dim3(div_up(na, 64)), dim3(64), 0,
                (const unsigned long long *)part_keys.get(), k, (const int *)fb_rows.get(), (const int *)(meta.get() + 1), idx.get(),
                d2.get());
    return;
  }
  const int na_tiles = (na + 31) / 32, nb_tiles = (nb + 31) / 32;
  const bool bf = knn_wide_bf16();
  constexpr int kChunks = knn_kp16(kD) / 16;
  DevBuf<float> Ap(c, bf ? (size_t)na_tiles * kChunks * 64 * 8 : (size_t)na_tiles * kSteps * 64);
  MM3D_HIP(hipMemsetAsync(meta.get(), 0, 16, c->stream));
  DevBuf<float> colsum_tmp, Bp_tmp;
  const float *colsum = B->knn_colsum.get(), *Bp = B->knn_Bp.get();
  if (!colsum || !Bp) {
    knn_target_operands<kD>(c, B, colsum_tmp, Bp_tmp);
    colsum = colsum_tmp.get();
    Bp = Bp_tmp.get();
  }
  const float inv_nb = 1.0f / (float)nb;
  DevBuf<float> nrm_a;
  if (bf) {
    nrm_a = DevBuf<float>(c, (size_t)na);
    MM3D_LAUNCH(c, "desc_knn_prep", na * kD * 4.0, (k_knn_rownorm_c<kD>), dim3(div_up(na, 4)), dim3(256), 0, Ad, na, colsum, inv_nb, nrm_a.get());
    MM3D_LAUNCH(c, "desc_knn_prep", na * (kD + kKP) * 4.0, (k_knn_prep_bf<kD>), dim3(div_up((size_t)na_tiles * kChunks * 64, 256)), dim3(256), 0, Ad, na,
                na_tiles, 0, colsum, inv_nb, (const float *)nrm_a.get(), reinterpret_cast<uint4 *>(Ap.get()));
  } else {
    MM3D_LAUNCH(c, "desc_knn_prep", na * (kD + kKP) * 4.0, (k_knn_prep<kD>), dim3(div_up((size_t)na_tiles * kSteps * 64, 256)), dim3(256), 0, Ad, na,
                na_tiles, 0, colsum, inv_nb, Ap.get());
  }
  const int a_blocks = div_up(na_tiles, kWideQT);
  int parts = 1;
  if (bf) {
    // one resident block per CU (its registers): at most one round of blocks, and at least four target tiles per wave
    static const int cus = snb_cu_count(c->device);
    parts = std::max(1, std::min(std::min(32, cus / std::max(a_blocks, 1)), nb_tiles / (kSlices * 4)));     // (any number: a part is eight lists)
  } else {
    while (parts < 32 && a_blocks * parts < 512 && nb_tiles / (kSlices * parts * 2) >= 2) parts *= 2;
  }
  const int n_lists = kLists * parts;
  DevBuf<float> cand_d(c, (size_t)na * n_lists * kListLen);
  DevBuf<int> cand_i(c, (size_t)na * n_lists * kListLen);
  // (its own profile name, and the flops it EXECUTES: three bf16 products per padded contraction step -- against the bf16 MFMA
  // peak in bench.py; the f32-equivalent figure, 2 na nb kKP, is a third of it less the padding)
  if (bf)
    MM3D_LAUNCH(c, "desc_knn_mfma_bf16", 3.0 * 2.0 * (double)na_tiles * 32 * (double)nb_tiles * 32 * knn_kp16(kD), (k_knn_mfma_wide_bf<kD>), dim3(a_blocks, parts),
                dim3(256), 0, reinterpret_cast<const uint4 *>(Ap.get()), na, na_tiles, reinterpret_cast<const uint4 *>(Bp), nb, nb_tiles, cand_d.get(),
                cand_i.get());
  else
    MM3D_LAUNCH(c, "desc_knn_mfma", 2.0 * (double)na_tiles * 32 * (double)nb_tiles * 32 * kKP, (k_knn_mfma_wide<kD>), dim3(a_blocks, parts),
                dim3(256), 0, (const float *)Ap.get(), na, na_tiles, Bp, nb, nb_tiles, cand_d.get(), cand_i.get());
  // the split-bf16 selector: three accumulation chains instead of one (x 1.5), and its own approximation error (5e-5, see
  // k_knn_mfma_wide_bf)
  MM3D_LAUNCH(c, "desc_knn_rerank", na * (double)(n_lists * kListLen * 8 + 32 * kD * 4), (k_knn_rerank_wide<kD>), dim3(div_up(na, 4)), dim3(256), 0,
              Ad, na, Bd, nb, k, n_lists, (const float *)cand_d.get(), (const int *)cand_i.get(), colsum, inv_nb, idx.get(), d2.get(),
              fb_rows.get(), (int *)(meta.get() + 1), bf ? 1.5f : 1.0f, bf ? 5e-5f : 0.0f);
  MM3D_LAUNCH(c, "desc_knn_fallback", 0.0, (k_knn_exact_wide<kD>), dim3(fb_groups < 64 ? fb_groups : 64, kFbParts), dim3(64), 0, Ad, Bd, nb, k,
              (const int *)fb_rows.get(), (const int *)(meta.get() + 1), 0, part_keys.get());
  MM3D_LAUNCH(c, "desc_knn_fallback", 0.0, k_knn_merge_parts, dim3(div_up(na, 64)), dim3(64), 0,
              (const unsigned long long *)part_keys.get(), k, (const int *)fb_rows.get(), (const int *)(meta.get() + 1), idx.get(),
              d2.get());
  if (c->debug) {
    unsigned *h = (unsigned *)c->pin(64);
    MM3D_HIP(hipMemcpyAsync(h, meta.get(), 16, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->knn_fallback_rows += (long long)h[1];
    c->knn_rows += na;
  }
}

// ---------------------------------------------------------------- any k, any row width
// The reference takes matching_k from the command line / the parameter server as an arbitrary size_t
// (R/src/map_merging.cpp:43-47) and hands it to FLANN.  The kernels above keep k <= 16 candidates in
// registers; a larger k goes through this plain exact kernel: one wave per query row (rows dealt to a fixed
// number of waves), every target's distance in FLANN's accumulation order into the wave's scratch row, then
// the k smallest (distance, index) keys one after the other (each the minimum of the keys above the last).
constexpr int kAnyMaxDim = 2048;
__global__ void __launch_bounds__(256)
k_knn_any(const float *__restrict__ A, int na, const float *__restrict__ B, int nb, int dim, int k, float *__restrict__ scratch /* [waves][nb] */,
          int *__restrict__ idx, float *__restrict__ d2out)
{
  __shared__ float s_row[4][kAnyMaxDim];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_waves = gridDim.x * 4, w = blockIdx.x * 4 + wave;
  float *dist = scratch + (size_t)w * nb;
  float *x = s_row[wave];
  for (int a = w; a < na; a += n_waves) {
    wave_lds_fence();
    for (int d = lane; d < dim; d += kWave) x[d] = A[(size_t)a * dim + d];
    wave_lds_fence();
    for (int j = lane; j < nb; j += kWave) {
      const float *b = B + (size_t)j * dim;
      float r = 0.0f;
      for (int d = 0; d < dim; ++d) {
        const float df = x[d] - b[d];
        r = __fadd_rn(r, __fmul_rn(df, df));
      }
      dist[j] = r;
    }
    unsigned long long last = 0;
    bool first = true;
    for (int o = 0; o < k; ++o) {
      unsigned long long best = ~0ull;
      for (int j = lane; j < nb; j += kWave) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(dist[j]) << 32) | (unsigned)j;   // distances are >= 0
        if ((first || key > last) && key < best) best = key;
      }
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) {
        const unsigned long long other = __shfl_xor(best, sft, kWave);
        best = other < best ? other : best;
      }
      const bool found = best != ~0ull;
      if (lane == 0) {
        idx[(size_t)a * k + o] = found ? (int)(unsigned)(best & 0xffffffffull) : -1;      // fewer than k targets: the tail is empty
        d2out[(size_t)a * k + o] = found ? __uint_as_float((unsigned)(best >> 32)) : INFINITY;
      }
      last = best;
      first = false;
    }
  }
}

static void desc_knn_any(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  const int na = (int)A->n, nb = (int)B->n, dim = A->dim;
  if (dim > kAnyMaxDim) throw Error(MM3D_EUNSUPPORTED, "descriptor k-NN: rows wider than 2048 floats");
  idx = DevBuf<int>(c, (size_t)na * k);
  d2 = DevBuf<float>(c, (size_t)na * k);
  if (na == 0) return;
  const int blocks = (int)std::min<size_t>(256, div_up((size_t)na, (size_t)4));
  DevBuf<float> scratch(c, (size_t)blocks * 4 * (size_t)(nb > 0 ? nb : 1));
  MM3D_LAUNCH(c, "desc_knn_any", 2.0 * na * (double)nb * dim, k_knn_any, dim3(blocks), dim3(256), 0, (const float *)A->data.get(), na,
              (const float *)B->data.get(), nb, dim, k, scratch.get(), idx.get(), d2.get());
}

void desc_knn(Context *c, const mm3d_desc *A, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  MM3D_REQUIRE(A->dim == B->dim, "descriptor dimensions differ");
  MM3D_REQUIRE(k >= 1, "k must be positive");
  if (k > kMaxK) { desc_knn_any(c, A, B, k, idx, d2); return; }
  if (A->dim == 33) desc_knn_impl<33>(c, A, B, k, idx, d2);          // FPFHSignature33
  else if (A->dim == 2) desc_knn_impl<2>(c, A, B, k, idx, d2);       // PrincipalRadiiRSD (r_min, r_max)
  else if (A->dim == 125) desc_knn_impl<125>(c, A, B, k, idx, d2);   // PFHSignature125
  else if (A->dim == 250) desc_knn_wide_impl<250>(c, A, B, k, idx, d2);     // PFHRGBSignature250
  else if (A->dim == 1344) desc_knn_wide_impl<1344>(c, A, B, k, idx, d2);   // SHOT1344
  else if (A->dim == 352) desc_knn_wide_impl<352>(c, A, B, k, idx, d2);     // SHOT352's shape only (mm3d_debug_desc_knn: the reference binds SHOT1344)
  else if (A->dim == 1980) desc_knn_wide_impl<1980>(c, A, B, k, idx, d2);   // ShapeContext1980
  else throw Error(MM3D_EUNSUPPORTED, "descriptor k-NN is built for RSD (2), FPFH (33), PFH (125), PFHRGB (250), SHOT (1344) and SC3D (1980) rows");
}

__global__ void k_gather_desc_rows(const float *__restrict__ X, const int *__restrict__ rows, int n_rows, int dim,
                                   float *__restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n_rows * dim) return;
  out[e] = X[(size_t)rows[e / dim] * dim + e % dim];
}

// the same for up to kGatherMax sources in ONE launch (blockIdx.y = the source): a batch of pairs that share their target
// used to cost a gather launch per source -- with many small maps nearly 2 000 of the step's launches
constexpr int kGatherMax = 16;
struct GatherSrcs {
  const float *X[kGatherMax];
  const int *rows[kGatherMax];
  int n_rows[kGatherMax];
  int first_row[kGatherMax];                           // where the source's rows start in the output
};
__global__ void k_gather_desc_rows_multi(GatherSrcs G, int dim, float *__restrict__ out)
{
  const int s = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)G.n_rows[s] * dim) return;
  out[(size_t)G.first_row[s] * dim + e] = G.X[s][(size_t)G.rows[s][e / dim] * dim + e % dim];
}

// k-NN of a subset of A's rows (device index list); result row r belongs to rows[r]
void desc_knn_rows(Context *c, const mm3d_desc *A, const int *rows_dev, int n_rows, const mm3d_desc *B, int k, DevBuf<int> &idx,
                   DevBuf<float> &d2)
{
  mm3d_desc sub;
  sub.dim = A->dim;
  sub.type = A->type;
  sub.n = (size_t)n_rows;
  sub.data = DevBuf<float>(c, (size_t)n_rows * A->dim);
  if (n_rows > 0)
    MM3D_LAUNCH(c, "desc_knn_prep", n_rows * A->dim * 8.0, k_gather_desc_rows, dim3(div_up((size_t)n_rows * A->dim, 256)), dim3(256), 0,
                (const float *)A->data.get(), rows_dev, n_rows, A->dim, sub.data.get());
  desc_knn(c, &sub, B, k, idx, d2);
}

// The same for rows taken from several descriptor sets (the sampled rows of several sources against ONE target):
// one query matrix, one search; source i's rows start at the sum of the earlier sources' row counts.
void desc_knn_rows_multi(Context *c, const KnnRows *srcs, int n_srcs, const mm3d_desc *B, int k, DevBuf<int> &idx, DevBuf<float> &d2)
{
  MM3D_REQUIRE(n_srcs > 0, "desc_knn_rows_multi: no sources");
  mm3d_desc sub;
  sub.dim = srcs[0].A->dim;
  sub.type = srcs[0].A->type;
  size_t total = 0;
  for (int i = 0; i < n_srcs; ++i) {
    MM3D_REQUIRE(srcs[i].A->dim == sub.dim, "desc_knn_rows_multi: descriptor sizes differ");
    total += (size_t)srcs[i].n_rows;
  }
  sub.n = total;
  sub.data = DevBuf<float>(c, total * sub.dim);
  size_t off = 0;
  for (int i0 = 0; i0 < n_srcs; i0 += kGatherMax) {      // (a batch holds at most 16 pairs: one launch)
    GatherSrcs G;
    std::memset(&G, 0, sizeof(G));
    const int cnt = std::min(kGatherMax, n_srcs - i0);
    int max_rows = 0;
    size_t rows_here = 0;
    for (int i = 0; i < cnt; ++i) {
      const KnnRows &S = srcs[i0 + i];
      G.X[i] = (const float *)S.A->data.get();
      G.rows[i] = S.rows_dev;
      G.n_rows[i] = S.n_rows;
      G.first_row[i] = (int)off;
      off += (size_t)S.n_rows;
      rows_here += (size_t)S.n_rows;
      max_rows = std::max(max_rows, S.n_rows);
    }
    if (max_rows > 0)
      MM3D_LAUNCH(c, "desc_knn_prep", rows_here * sub.dim * 8.0, k_gather_desc_rows_multi, dim3(div_up((size_t)max_rows * sub.dim, 256), cnt), dim3(256), 0, G,
                  sub.dim, sub.data.get());
  }
  desc_knn(c, &sub, B, k, idx, d2);
}

}  // namespace mm3d
