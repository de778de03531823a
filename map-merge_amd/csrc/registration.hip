// registration.hip -- hypothesis scoring for RANSAC and SAC-IA (K10/K11 in SURVEY 2.2).
//
// RANSAC scoring   SampleConsensusModelRegistration::countWithinDistance behind
//                  CorrespondenceRejectorSampleConsensus (R/src/matching.cpp:119-124)
// SAC-IA scoring   SampleConsensusInitialAlignment::computeErrorMetric (R/src/matching.cpp:159-173)
//
// The host replays the reference's random sample stream (host_pipeline.cpp) and hands over the
// sampled indices of ALL hypotheses; these kernels build the models (SAC-IA) and score every
// hypothesis in one launch with exactly the float predicate / float summation order of the
// sequential CPU loop, so the host's replay of the accept logic picks the same hypothesis.
// (ICP and transformScore live in nn.hip.)
#include <cfloat>

#include "device_util.hpp"
#include "linalg_shared.hpp"

namespace mm3d {

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

// ---------------------------------------------------------------- SAC-IA hypothesis models
// TransformationEstimationSVD on the 3 sampled pairs of every hypothesis: one thread each, the same
// host+device source the CPU side uses (linalg_shared.hpp), so T is bit-identical to a host build.
// corr_ref[e] indexes the k-NN table of the sampled rows (row * k + the replayed random pick), so the
// correspondence itself never travels to the host.
// One pair's SAC-IA scoring inside a batched launch (blockIdx.y / .z picks the pair): thousands of small pairs
// would otherwise cost four latency-bound launches each.
struct SacJob {
  const float4 *skp, *tkp;       // keypoints in reference order (the models' samples)
  const int *samp, *corr_ref, *nn;
  float *T_all;                  // [H][16]
  const float4 *skp_q;           // the queries of the error kernel (any order; .w = keypoint index when `permuted`)
  int permuted, ns, ns_pad;
  GridView g;                    // target keypoint grid with merged 3x3x3 lists
  float *E;                      // [H][ns_pad]
  float *err;                    // [H]
  float *T_best;                 // [16]
};

__global__ void k_sacia_models(const SacJob *__restrict__ jobs, int H)
{
  const SacJob &J = jobs[blockIdx.y];
  const float4 *__restrict__ skp = J.skp, *__restrict__ tkp = J.tkp;
  const int *__restrict__ samp = J.samp, *__restrict__ corr_ref = J.corr_ref, *__restrict__ nn = J.nn;
  float *__restrict__ T_all = J.T_all;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  float s[9], d[9], T[16];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 p = skp[samp[h * 3 + i]], q = tkp[nn[corr_ref[h * 3 + i]]];
    s[i * 3] = p.x; s[i * 3 + 1] = p.y; s[i * 3 + 2] = p.z;
    d[i * 3] = q.x; d[i * 3 + 1] = q.y; d[i * 3 + 2] = q.z;
  }
  umeyama_f32_shared(s, d, 3, T);
#pragma unroll
  for (int i = 0; i < 16; ++i) T_all[(size_t)h * 16 + i] = T[i];
}

// ---------------------------------------------------------------- SAC-IA hypothesis scoring
// E[h][i] = TruncatedError(d2 of (T_h * src_i) to its nearest target keypoint); rows padded to a
// multiple of 4 floats so the summation kernel can stream them with 16-byte loads.
#ifndef MM3D_SAC_SUB
#define MM3D_SAC_SUB 1
#endif
constexpr int kSacSub = MM3D_SAC_SUB;
__global__ void __launch_bounds__(256)
k_sacia_err(const SacJob *__restrict__ jobs, int h_first, float thresh, float radius)
{
  const SacJob &J = jobs[blockIdx.z];
  const float4 *__restrict__ skp = J.skp_q;
  const int permuted = J.permuted, ns = J.ns, ns_pad = J.ns_pad;
  const GridView g = J.g;
  const float *__restrict__ T_all = J.T_all;
  float *__restrict__ E = J.E;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const int h = h_first + (int)blockIdx.y;            // uniform: the model sits in scalar registers
  const float *T = T_all + (size_t)h * 16;
  float Tl[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Tl[k] = T[k];
  const float4 s = skp[i];
  const float3 p = xform(Tl, s.x, s.y, s.z);
  float best = INFINITY;
  // 500 x K_s queries per pair against the same K_t targets: the stencil walk is taken out of the
  // query.  The target grid (cell = radius / kSacSub) carries, per cell, the merged list of the block
  // of cells the radius can reach (grid_ensure_nblists), ascending in distance from the cell centre.
  // A query scans ONE contiguous span and stops at the first entry whose centre distance, less the
  // query's own, exceeds min(best so far, radius): nothing later can be nearer.  An empty
  // span is the "nothing in range" answer most wrong hypotheses get.  The minimum itself is taken
  // over exact float distances, so the order of the scan does not show in the result.
  const int cx = cell_floor(p.x, g.minx, g.inv), cy = cell_floor(p.y, g.miny, g.inv), cz = cell_floor(p.z, g.minz, g.inv);
  const bool inside = cx >= 0 && cx < g.dx && cy >= 0 && cy < g.dy && cz >= 0 && cz < g.dz;
  if (inside) {
    const size_t c = ((size_t)cz * g.dy + cy) * g.dx + cx;
    const int b = g.nb_start[c], e = g.nb_start[c + 1];
    // |q - e| >= |e - centre| - |q - centre| for every entry e; 1e-3 cell covers the rounding of both terms
    const float ox = p.x - (g.minx + ((float)cx + 0.5f) * g.cell), oy = p.y - (g.miny + ((float)cy + 0.5f) * g.cell);
    const float oz = p.z - (g.minz + ((float)cz + 0.5f) * g.cell);
    const float slack = sqrtf(ox * ox + oy * oy + oz * oz) + 1e-3f * g.cell;
    float want = thresh;                              // squared distance still worth finding
    // four loads in flight per step; min is idempotent, so the tail just re-reads the last point
    for (int j = b; j < e; j += 4) {
      const int last = e - 1;
      const float4 q0 = g.nb_pts[j];
      const float4 q1 = g.nb_pts[min(j + 1, last)];
      const float4 q2 = g.nb_pts[min(j + 2, last)];
      const float4 q3 = g.nb_pts[min(j + 3, last)];
      const float d0 = dist2(p.x, p.y, p.z, q0.x, q0.y, q0.z), d1 = dist2(p.x, p.y, p.z, q1.x, q1.y, q1.z);
      const float d2 = dist2(p.x, p.y, p.z, q2.x, q2.y, q2.z), d3 = dist2(p.x, p.y, p.z, q3.x, q3.y, q3.z);
      best = fminf(best, fminf(fminf(d0, d1), fminf(d2, d3)));
      want = fminf(want, best);
      const float lb = q3.w - slack;
      if (lb > 0.0f && lb * lb > want) break;
    }
  } else {
    // outside the grid: clipped stencil walk
    for_each_candidate(g, p.x, p.y, p.z, radius, [&](const float4 &q) {
      best = fminf(best, dist2(p.x, p.y, p.z, q.x, q.y, q.z));
      return true;
    });
  }
  // the queries run in the source keypoints' Hilbert order (neighbouring lanes land in neighbouring target
  // cells: similar span lengths, shared cache lines); the summation order is the keypoint index order
  const int slot = permuted ? __float_as_int(s.w) : i;
  E[(size_t)h * ns_pad + slot] = (best <= thresh) ? best / thresh : 1.0f;
}

// error += e in source-keypoint order, float: the same chain the CPU path evaluates, bit for bit.
// A block owns kSumRows hypotheses; all 256 threads stream the rows' tiles into LDS with coalesced
// 16-byte loads (double buffered), the first kSumRows lanes walk their row out of LDS sequentially.
constexpr int kSumRows = 8;
constexpr int kSumTile = 1024;
constexpr int kSumPad = 36;
__global__ void __launch_bounds__(256) k_seq_sum(const SacJob *__restrict__ jobs, int H)
{
  const float *__restrict__ E = jobs[blockIdx.y].E;
  const int ns = jobs[blockIdx.y].ns, ns_pad = jobs[blockIdx.y].ns_pad;
  float *__restrict__ err = jobs[blockIdx.y].err;
  // row stride = tile + 36 floats: lane r's 16-byte reads land on banks 36 r mod 64 (all distinct), and
  // the zeroed tail lets the chain prefetch two groups past the end without a branch
  __shared__ __attribute__((aligned(16))) float buf[2][kSumRows][kSumTile + kSumPad];
  const int h0 = blockIdx.x * kSumRows;
  for (int e = threadIdx.x; e < 2 * kSumRows * kSumPad; e += blockDim.x)
    buf[e / (kSumRows * kSumPad)][(e / kSumPad) % kSumRows][kSumTile + e % kSumPad] = 0.0f;
  const int rows = min(kSumRows, H - h0);
  const int ntiles = (ns + kSumTile - 1) / kSumTile;
  auto stage = [&](int t, int b) {
    // kSumRows x kSumTile floats = 2048 float4, staged by waves 1..3 only (after the first tile): wave 0
    // carries the add chains and must not sit on a global-memory wait between tiles
    const int first = t == 0 ? 0 : kWave;
    for (int e = (int)threadIdx.x - first; e >= 0 && e < kSumRows * (kSumTile / 4); e += (int)blockDim.x - first) {
      const int r = e / (kSumTile / 4), c4 = e % (kSumTile / 4);
      const int i = t * kSumTile + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && i < ns_pad) v = *reinterpret_cast<const float4 *>(E + (size_t)(h0 + r) * ns_pad + i);
      // everything past the row's end becomes +0: e + 0 == e, so the chain below can run in whole
      // groups of 16 without a tail
      if (i + 1 >= ns) v.y = 0.f;
      if (i + 2 >= ns) v.z = 0.f;
      if (i + 3 >= ns) v.w = 0.f;
      if (i >= ns) v.x = 0.f;
      *reinterpret_cast<float4 *>(&buf[b][r][c4 * 4]) = v;
    }
  };
  float e = 0.0f;
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
    if ((int)threadIdx.x < rows) {
      const int cnt = min(kSumTile, ns - t * kSumTile);
      const float4 *row = reinterpret_cast<const float4 *>(buf[t & 1][threadIdx.x]);
      const int groups = (cnt + 15) >> 4;
      // The add chain is the critical path (one dependent v_add_f32 after another).  Two register
      // sets of 16 values leapfrog: while one is being added the other is in flight from LDS.  Groups
      // past the end are zeros (+0 leaves the sum unchanged).
      float4 a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
      float4 b0 = row[4], b1 = row[5], b2 = row[6], b3 = row[7];
#pragma unroll 1
      for (int gi = 0; gi < groups; gi += 2) {
        e += a0.x; e += a0.y; e += a0.z; e += a0.w;
        e += a1.x; e += a1.y; e += a1.z; e += a1.w;
        e += a2.x; e += a2.y; e += a2.z; e += a2.w;
        e += a3.x; e += a3.y; e += a3.z; e += a3.w;
        a0 = row[gi * 4 + 8]; a1 = row[gi * 4 + 9]; a2 = row[gi * 4 + 10]; a3 = row[gi * 4 + 11];
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch HERE: the scheduler otherwise sinks it to its first use
        e += b0.x; e += b0.y; e += b0.z; e += b0.w;
        e += b1.x; e += b1.y; e += b1.z; e += b1.w;
        e += b2.x; e += b2.y; e += b2.z; e += b2.w;
        e += b3.x; e += b3.y; e += b3.z; e += b3.w;
        b0 = row[gi * 4 + 12]; b1 = row[gi * 4 + 13]; b2 = row[gi * 4 + 14]; b3 = row[gi * 4 + 15];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < rows) err[h0 + threadIdx.x] = e;
}

// the target-side search structure of SAC-IA scoring (cached on the keypoint cloud)
static const Grid &sacia_target_grid(Context *c, const mm3d_cloud *tgt_kp, float corr_thresh)
{
  const float radius = std::sqrt(corr_thresh > 0.f ? corr_thresh : 0.f);
  // kSacSub cells a hair longer than the search radius: everything within the radius of a point of cell c
  // lies in c's (2 kSacSub + 1)^3 block (cloud_grid may only ever ENLARGE the cell, which keeps that true)
  const float want = radius * 1.001f / (float)kSacSub;
  const float cell = want > 0.125f ? want : 0.125f;
  const Grid &g = cloud_grid(c, tgt_kp, cell);
  grid_ensure_nblists(c, g, kSacSub);
  return g;
}

void prepare_sacia_target(Context *c, const mm3d_cloud *kp, float corr_thresh)
{
  if (kp->n) {
    (void)sacia_target_grid(c, kp, corr_thresh);
    cloud_hilbert(c, kp);                      // source role: query order of sacia_errors
  }
}

__global__ void k_sacia_pick(const SacJob *__restrict__ jobs, int H);

// Models, errors, error sums and the pick for a batch of pairs: four launches whatever the batch size.
// pairs[i].samp / corr_ref / nn and .T_best (16 floats) are device memory of the caller's; T_all, E and err are
// scratch of this call (pool buffers of this context: whoever gets them next is enqueued behind these kernels).
void sacia_score_batch(Context *c, const SacPair *pairs, int n, int H, float corr_thresh)
{
  if (n == 0 || H == 0) return;
  const float radius = std::sqrt(corr_thresh > 0.f ? corr_thresh : 0.f);
  std::vector<DevBuf<float>> bufs;
  bufs.reserve((size_t)n * 3);
  SacJob *hj = (SacJob *)c->pin(sizeof(SacJob) * (size_t)n);
  int max_ns = 0;
  double err_bytes = 0.0, sum_bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const SacPair &P = pairs[i];
    const int ns = (int)P.src_kp->n;
    const int ns_pad = (ns + 3) & ~3;
    const Grid &g = sacia_target_grid(c, P.tgt_kp, corr_thresh);
    cloud_hilbert(c, P.src_kp);                    // cached on the cloud (prepare_sacia_target)
    const bool permuted = P.src_kp->hil_pts.get() && P.src_kp->n_finite == P.src_kp->n;
    bufs.emplace_back(c, (size_t)H * 16);
    float *T_all = bufs.back().get();
    bufs.emplace_back(c, (size_t)ns_pad * H);
    float *E = bufs.back().get();
    bufs.emplace_back(c, (size_t)H);
    float *err = bufs.back().get();
    SacJob q;
    std::memset(&q, 0, sizeof(q));
    q.skp = (const float4 *)P.src_kp->pts.get();
    q.tkp = (const float4 *)P.tgt_kp->pts.get();
    q.samp = P.samp; q.corr_ref = P.corr_ref; q.nn = P.nn;
    q.T_all = T_all;
    q.skp_q = permuted ? (const float4 *)P.src_kp->hil_pts.get() : (const float4 *)P.src_kp->pts.get();
    q.permuted = permuted ? 1 : 0;
    q.ns = ns; q.ns_pad = ns_pad;
    q.g = g.view();
    q.E = E; q.err = err; q.T_best = P.T_best;
    hj[i] = q;
    max_ns = std::max(max_ns, ns);
    err_bytes += (double)ns * H * 4.0 + ns * 16.0;
    sum_bytes += (double)ns * H * 4.0;
  }
  DevBuf<SacJob> d_jobs(c, (size_t)n);
  MM3D_HIP(hipMemcpyAsync(d_jobs.get(), hj, sizeof(SacJob) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  const SacJob *dj = d_jobs.get();
  MM3D_LAUNCH(c, "sacia_models", n * H * 88.0, k_sacia_models, dim3(div_up(H, 64), n), dim3(64), 0, dj, H);
  for (int h0 = 0; h0 < H; h0 += 65535) {               // gridDim.y holds at most 65535 hypotheses
    const int hn = std::min(65535, H - h0);
    MM3D_LAUNCH(c, "sacia_err", err_bytes * hn / H, k_sacia_err, dim3(div_up(max_ns, 256), hn, n), dim3(256), 0, dj, h0, corr_thresh, radius);
  }
  MM3D_LAUNCH(c, "sacia_seq_sum", sum_bytes, k_seq_sum, dim3(div_up(H, kSumRows), n), dim3(256), 0, dj, H);
  // "if (i_iter == 0 || error < lowest_error)": the first minimum, picked on the device
  MM3D_LAUNCH(c, "sacia_pick", n * (H * 4.0 + 128.0), k_sacia_pick, dim3(n), dim3(64), 0, dj, H);
}

// "if (i == 0 || error < lowest_error) keep": the first minimum, by one wave.  A NaN never wins a
// '<', and a NaN at i == 0 is never beaten.
__global__ void __launch_bounds__(64) k_sacia_pick(const SacJob *__restrict__ jobs, int H)
{
  const float *__restrict__ err = jobs[blockIdx.x].err;
  const float *__restrict__ T_all = jobs[blockIdx.x].T_all;
  float *__restrict__ T_best = jobs[blockIdx.x].T_best;
  const int lane = threadIdx.x;
  const float e0 = err[0];
  unsigned long long best = ~0ull;
  if (e0 == e0) {
    for (int i = lane; i < H; i += kWave) {
      const float e = err[i];
      if (e == e) {
        // e >= 0 (sums of non-negative terms) or -0: order the bits as values, ties to the lower index
        const unsigned long long key = ((unsigned long long)f2ord(e) << 32) | (unsigned)i;
        best = key < best ? key : best;
      }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const unsigned long long other = __shfl_xor(best, s, kWave);
      best = other < best ? other : best;
    }
  }
  const int h = (e0 == e0) ? (int)(unsigned)(best & 0xffffffffull) : 0;
  if (lane < 16) T_best[lane] = T_all[(size_t)h * 16 + lane];
}


}  // namespace mm3d
