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
#include <atomic>
#include <cfloat>
#include <cstdlib>

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
  const float4 *skp_q;           // the queries of the error kernel (any order: the certificate's sums are order-free)
  int ns;
  GridView g;                    // target keypoint grid with merged 3x3x3 lists
  float *err;                    // [H]: the float chains' results of the hypotheses that needed one (k_sacia_exact)
  float *T_best;                 // [16]
  // the certified pick (below): per hypothesis the terms' sum in double and the number of terms that are not 1.0f (both
  // zeroed), its class (0 out, 1 the chain decides, 2 its float sum is known), the list of the class-1 hypotheses, and the
  // pair's verdict
  double *S;                     // [H]
  int *n_part;                   // [H]
  int *chain_list;               // [H]
  unsigned char *cls;            // [H]
  struct SacCtl *ctl;
};
struct SacCtl { int n_cand, decided, winner, n_chain; };

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
// One term of computeErrorMetric: TruncatedError(d2 of (T * s) to its nearest target keypoint).
#ifndef MM3D_SAC_SUB
#define MM3D_SAC_SUB 1
#endif
constexpr int kSacSub = MM3D_SAC_SUB;
__device__ __forceinline__ float sacia_term(const GridView &g, const float *Tl, const float4 s, float thresh, float radius)
{
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
  return (best <= thresh) ? best / thresh : 1.0f;
}

// Every (hypothesis, keypoint) term, summed per hypothesis in double and counted when it is not 1.0f: what the certified pick
// below decides from.  The queries run in the source keypoints' Hilbert order (neighbouring lanes land in neighbouring target
// cells: similar span lengths, shared cache lines).  Until round 6 the terms were WRITTEN, E[h][i], 31 MB per headline pair
// in scattered 4-byte stores, for the float chains to read back; the chains that still run recompute theirs (k_sacia_exact):
// the kernel alone 0.55 -> 0.42 ms per batch of three headline pairs, the headline + 3 %.
__global__ void __launch_bounds__(256)
k_sacia_err(const SacJob *__restrict__ jobs, int h_first, float thresh, float radius)
{
  const SacJob &J = jobs[blockIdx.z];
  const int ns = J.ns;
  if ((int)(blockIdx.x * blockDim.x) >= ns) return;   // (the grid is as wide as the batch's largest pair)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < ns;
  const int h = h_first + (int)blockIdx.y;            // uniform: the model sits in scalar registers
  const float *T = J.T_all + (size_t)h * 16;
  float Tl[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Tl[k] = T[k];
  const float e_term = valid ? sacia_term(J.g, Tl, J.skp_q[i], thresh, radius) : 0.0f;
  // for the certified pick: the terms' sum in double (any order) and how many terms are not exactly 1.0f
  __shared__ double s_sum[4];
  __shared__ int s_part[4];
  const double ws = wave_sum((double)e_term);
  const int part = __popcll(ballot(valid && e_term != 1.0f));
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = ws; s_part[threadIdx.x >> 6] = part; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&J.S[h], (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
    const int pt = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (pt) atomicAdd(&J.n_part[h], pt);
  }
}

// The certified pick.  SampleConsensusInitialAlignment keeps the hypothesis with the lowest error sum -- "if (i_iter == 0 ||
// error < lowest_error)", ia_ransac.hpp -- and hands on that hypothesis' TRANSFORM; the sum itself leaves the stage nowhere.
// The sum is a float chain in keypoint order (15.7 k dependent additions per hypothesis on the headline), and rounds 2 - 5
// reproduced every hypothesis' chain (k_seq_sum: 200 us per launch inside the step, all of it latency).  What has to be the CPU
// path's is the DECISION.  Every term lies in [0, 1], so the chain's partial sums never exceed the real sum R of the n float
// terms, and its result F satisfies |F - R| <= (n - 1) u R (1 + (n - 1) u), u = 2^-24; S, the same terms summed in double in
// whatever order the atomics arrive, is within n 2^-53 R of R.  So F lies in [S (1 - d), S (1 + d)], d = n u (1 + 1e-3) + 1e-12
// -- 0.09 % at 15.7 k keypoints -- and a hypothesis whose interval lies wholly above the lowest upper end is not a minimum.
// On the headline the second-best hypothesis of a pair is 0.7 - 8 % above the best (scripts/sacia_price.py,
// profiles/r06_sacia_price.txt): ONE candidate is left and no chain runs at all.  A hypothesis all of whose terms are 1.0f
// (no source keypoint lands within range of a target keypoint: maps that do not overlap) has F = n exactly.  What is left
// takes the chain (k_sacia_chain), and the first minimum among the candidates is the CPU path's pick: every minimum of F is a
// candidate, ties included.
__global__ void __launch_bounds__(256) k_sacia_select(const SacJob *__restrict__ jobs, int H)
{
  const SacJob &J = jobs[blockIdx.x];
  const double *__restrict__ S = J.S;
  const int *__restrict__ n_part = J.n_part;
  const double n = (double)J.ns;
  const double d = n * 5.9604644775390625e-8 * 1.001 + 1e-12;
  __shared__ double s_min[4];
  __shared__ int s_cnt[4], s_chain[4], s_first[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the lowest upper end (a sum that is not a number compares false everywhere: no candidate at all -> everything is one)
  double up = INFINITY;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const double hi = n_part[h] == 0 ? n : S[h] * (1.0 + d);
    up = hi < up ? hi : up;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(up, o, kWave); up = t < up ? t : up; }
  if (lane == 0) s_min[wave] = up;
  __syncthreads();
  up = fmin(fmin(s_min[0], s_min[1]), fmin(s_min[2], s_min[3]));
  int cnt = 0, chain = 0, first = 0x7fffffff;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const bool exact = n_part[h] == 0;
    const double lo = exact ? n : S[h] * (1.0 - d);
    const bool cand = lo <= up;
    J.cls[h] = cand ? (exact ? 2 : 1) : 0;
    if (exact) J.err[h] = (float)J.ns;                  // n ones add up to n without a rounding (n < 2^24)
    cnt += cand ? 1 : 0;
    chain += (cand && !exact) ? 1 : 0;
    if (cand) first = min(first, h);
    if (cand && !exact) J.chain_list[atomicAdd(&J.ctl->n_chain, 1)] = h;       // (ctl arrives zeroed; any order)
  }
  cnt = wave_sum(cnt); chain = wave_sum(chain); first = wave_min_int(first);
  if (lane == 0) { s_cnt[wave] = cnt; s_chain[wave] = chain; s_first[wave] = first; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    chain = s_chain[0] + s_chain[1] + s_chain[2] + s_chain[3];
    first = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
    SacCtl c;
    c.n_cand = cnt; c.n_chain = cnt == 0 ? H : chain;
    // one candidate: it is the minimum.  Candidates whose sums are all KNOWN and equal (every term of every one is 1.0f):
    // the first of them.  No candidate (a sum that is not a number): every hypothesis takes the chain.
    c.decided = (cnt == 1 || (cnt > 0 && chain == 0)) ? 1 : 0;
    c.winner = c.decided ? first : -1;
    *J.ctl = c;
  }
  __syncthreads();
  const SacCtl c = *J.ctl;
  if (c.n_cand == 0)
    for (int h = threadIdx.x; h < H; h += blockDim.x) { J.cls[h] = 1; J.chain_list[h] = h; }
  if (c.decided && threadIdx.x < 16) J.T_best[threadIdx.x] = J.T_all[(size_t)c.winner * 16 + threadIdx.x];
}

// error += e in source-keypoint order, float -- the chain the CPU path evaluates, bit for bit -- for the hypotheses the
// certificate left open (chain_list): a block per hypothesis computes the terms of 256 keypoints at a time, in INDEX order,
// into LDS, and one lane adds them up in that order.  ~250 us for a hypothesis of 15.7 k keypoints, a few of them in one pair
// in forty; the launch is there for every batch and nearly always finds nothing to do (1 KB of LDS: it queues for none).
constexpr int kSacExactBlocks = 8;       // blocks per pair; more open hypotheses than that are worked off in turns
__global__ void __launch_bounds__(256) k_sacia_exact(const SacJob *__restrict__ jobs, float thresh, float radius)
{
  const SacJob &J = jobs[blockIdx.y];
  if (J.ctl->decided) return;
  const int n_chain = J.ctl->n_chain, ns = J.ns;
  __shared__ __attribute__((aligned(16))) float s_e[256];
  for (int k = blockIdx.x; k < n_chain; k += gridDim.x) {
    const int h = J.chain_list[k];
    const float *T = J.T_all + (size_t)h * 16;
    float Tl[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Tl[q] = T[q];
    float e = 0.0f;
    for (int i0 = 0; i0 < ns; i0 += 256) {
      const int i = i0 + (int)threadIdx.x;
      // (+0 past the end: e + 0 == e)
      s_e[threadIdx.x] = i < ns ? sacia_term(J.g, Tl, J.skp[i], thresh, radius) : 0.0f;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float4 *row = reinterpret_cast<const float4 *>(s_e);
#pragma unroll 4
        for (int g4 = 0; g4 < 64; ++g4) {
          const float4 v = row[g4];
          e += v.x; e += v.y; e += v.z; e += v.w;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) J.err[h] = e;
  }
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

// process-wide statistics of the certified pick (mm3d_debug_sacia_stats; collected only while switched on: it costs a wait
// per batch): 0 pairs, 1 pairs decided without a chain, 2 candidates left by the intervals, 3 chains run
static std::atomic<long long> g_sacia_stats[4];
static std::atomic<int> g_sacia_collect{0};
void debug_sacia_stats(long long out[4], int reset, int collect)
{
  for (int i = 0; i < 4; ++i) { out[i] = g_sacia_stats[i].load(); if (reset) g_sacia_stats[i] = 0; }
  if (collect >= 0) g_sacia_collect = collect;
}

// Models, errors and their sums, the certified pick (select, the float chains of what it leaves open, pick) for a batch of pairs:
// five launches whatever the batch size.
// pairs[i].samp / corr_ref / nn and .T_best (16 floats) are device memory of the caller's; T_all, err and the certificate's words are
// scratch of this call (pool buffers of this context: whoever gets them next is enqueued behind these kernels).
void sacia_score_batch(Context *c, const SacPair *pairs, int n, int H, float corr_thresh)
{
  if (n == 0 || H == 0) return;
  const float radius = std::sqrt(corr_thresh > 0.f ? corr_thresh : 0.f);
  std::vector<DevBuf<float>> bufs;
  bufs.reserve((size_t)n * 3);
  // the certified pick's words, one fill for the batch: per pair S [H doubles] | n_part [H ints] | chain_list [H ints] |
  // ctl [16 B] | cls [H bytes]
  const size_t per_pair = (((size_t)H * 17 + sizeof(SacCtl)) + 15) & ~(size_t)15;
  DevBuf<unsigned char> cert(c, per_pair * (size_t)n);
  MM3D_HIP(hipMemsetAsync(cert.get(), 0, per_pair * (size_t)n, c->stream));
  SacJob *hj = (SacJob *)c->pin(sizeof(SacJob) * (size_t)n);
  int max_ns = 0;
  double err_bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const SacPair &P = pairs[i];
    const int ns = (int)P.src_kp->n;
    const Grid &g = sacia_target_grid(c, P.tgt_kp, corr_thresh);
    cloud_hilbert(c, P.src_kp);                    // cached on the cloud (prepare_sacia_target)
    const bool permuted = P.src_kp->hil_pts.get() && P.src_kp->n_finite == P.src_kp->n;
    bufs.emplace_back(c, (size_t)H * 16);
    float *T_all = bufs.back().get();
    bufs.emplace_back(c, (size_t)H);
    float *err = bufs.back().get();
    SacJob q;
    std::memset(&q, 0, sizeof(q));
    q.skp = (const float4 *)P.src_kp->pts.get();
    q.tkp = (const float4 *)P.tgt_kp->pts.get();
    q.samp = P.samp; q.corr_ref = P.corr_ref; q.nn = P.nn;
    q.T_all = T_all;
    q.skp_q = permuted ? (const float4 *)P.src_kp->hil_pts.get() : (const float4 *)P.src_kp->pts.get();
    q.ns = ns;
    q.g = g.view();
    q.err = err; q.T_best = P.T_best;
    unsigned char *cp = cert.get() + per_pair * (size_t)i;
    q.S = reinterpret_cast<double *>(cp);
    q.n_part = reinterpret_cast<int *>(cp + (size_t)H * 8);
    q.chain_list = reinterpret_cast<int *>(cp + (size_t)H * 12);
    q.ctl = reinterpret_cast<SacCtl *>(cp + (size_t)H * 16);
    q.cls = cp + (size_t)H * 16 + sizeof(SacCtl);
    hj[i] = q;
    max_ns = std::max(max_ns, ns);
    err_bytes += (double)ns * H * 4.0 + ns * 16.0;          // (SURVEY 8d's figure: 4 B per (hypothesis, keypoint))
  }
  DevBuf<SacJob> d_jobs(c, (size_t)n);
  MM3D_HIP(hipMemcpyAsync(d_jobs.get(), hj, sizeof(SacJob) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  const SacJob *dj = d_jobs.get();
  MM3D_LAUNCH(c, "sacia_models", n * H * 88.0, k_sacia_models, dim3(div_up(H, 64), n), dim3(64), 0, dj, H);
  for (int h0 = 0; h0 < H; h0 += 65535) {               // gridDim.y holds at most 65535 hypotheses
    const int hn = std::min(65535, H - h0);
    MM3D_LAUNCH(c, "sacia_err", err_bytes * hn / H, k_sacia_err, dim3(div_up(max_ns, 256), hn, n), dim3(256), 0, dj, h0, corr_thresh, radius);
  }
  // "if (i_iter == 0 || error < lowest_error)": the first minimum, picked on the device -- certified from the sums in
  // double where that decides it (nearly always: one candidate, no chain), from the CPU path's float chains of the
  // candidates where it does not
  MM3D_LAUNCH(c, "sacia_select", n * (H * 12.0 + 128.0), k_sacia_select, dim3(n), dim3(256), 0, dj, H);
  MM3D_LAUNCH(c, "sacia_seq_sum", 0.0, k_sacia_exact, dim3(kSacExactBlocks, n), dim3(256), 0, dj, corr_thresh, radius);
  MM3D_LAUNCH(c, "sacia_pick", n * 128.0, k_sacia_pick, dim3(n), dim3(64), 0, dj, H);
  static const bool env_collect = getenv("MM3D_SACIA_STATS") != nullptr;
  if (env_collect || g_sacia_collect.load()) {
    SacCtl *hc = (SacCtl *)c->pin(sizeof(SacCtl) * (size_t)n);
    for (int i = 0; i < n; ++i)
      MM3D_HIP(hipMemcpyAsync(hc + i, hj[i].ctl, sizeof(SacCtl), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    for (int i = 0; i < n; ++i) {
      g_sacia_stats[0] += 1; g_sacia_stats[1] += hc[i].decided ? 1 : 0;
      g_sacia_stats[2] += hc[i].n_cand; g_sacia_stats[3] += hc[i].decided ? 0 : (hc[i].n_cand ? hc[i].n_chain : H);
    }
  }
}

// "if (i == 0 || error < lowest_error) keep": the first minimum, by one wave.  A NaN never wins a
// '<', and a NaN at i == 0 is never beaten.
__global__ void __launch_bounds__(64) k_sacia_pick(const SacJob *__restrict__ jobs, int H)
{
  if (jobs[blockIdx.x].ctl->decided) return;          // (k_sacia_select has written the winner's model)
  const float *__restrict__ err = jobs[blockIdx.x].err;
  const unsigned char *__restrict__ cls = jobs[blockIdx.x].cls;
  const float *__restrict__ T_all = jobs[blockIdx.x].T_all;
  float *__restrict__ T_best = jobs[blockIdx.x].T_best;
  const int lane = threadIdx.x;
  // (hypothesis 0 is always a candidate or beaten by one; its sum is a number: every term is)
  const float e0 = cls[0] ? err[0] : 0.0f;
  unsigned long long best = ~0ull;
  if (e0 == e0) {
    for (int i = lane; i < H; i += kWave) {
      if (!cls[i]) continue;                            // certainly not a minimum
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
