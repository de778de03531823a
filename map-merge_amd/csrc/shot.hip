// shot.hip -- computeLocalDescriptors(SHOT) on gfx950.
//
// R/src/dispatch_descriptors.h:46 binds Descriptor::SHOT to pcl::SHOTColorEstimation<PointXYZRGB,
// Normal, SHOT1344> (shape 32 x 11 + colour 32 x 31 bins), configured by R/src/features.cpp:105-109;
// rows with a non-finite bin are pruned together with their keypoints (features.cpp:118-143).
// PCL 1.8.1 features/impl/shot_lrf.hpp (getLocalRF) and shot.hpp (computePointSHOT,
// createBinDistanceShape, interpolateDoubleChannel, RGB2CIELAB, normalizeHistogram).
//
// One WAVE per keypoint (64-thread blocks):
//   1. gather the radius neighbours as (squared distance bits, index) keys into LDS and sort them
//      (bitonic): the float sums below are order dependent and follow the (distance, index) order;
//   2. local reference frame: the weighted covariance is seven sequential double chains (six matrix
//      entries + the weight sum) run by seven lanes over batches staged in LDS, then a Jacobi
//      eigen-solver in double (same text as the CPU restatement) and PCL's sign disambiguation
//      (ballot counts; the 5-around-the-median tie rule included);
//   3. votes: each lane turns one neighbour into its five (volume, slot, value) votes per channel
//      (cosine / colour neighbour slot, radial shell, inclination, azimuth, own slot; double
//      arithmetic like PCL), then the wave walks the batch IN ORDER with lane = (channel, volume)
//      owning that volume's slots, so every bin sees its float additions in neighbour order;
//   4. L2 normalisation with the sequential double sum of PCL.
// Neighbourhoods beyond the LDS key capacity rerun with the keys in global scratch.
// Algorithmic bytes: 48 B per gathered neighbour (point, normal, Lab) x 3 passes + 5412 B per row;
// the kernel is bound by its sequential chains and double-precision transcendentals, not by HBM.
#include <cmath>

#include "device_util.hpp"

namespace mm3d {

constexpr int kShotDim = 1344;
constexpr int kShotShapeBins = 10, kShotColorBins = 30, kShotSectors = 32;
constexpr int kShotColorOffset = kShotSectors * (kShotShapeBins + 1);   // 352
constexpr int kShotCap = 1024;        // neighbour keys held in LDS (a power of two)
constexpr int kLutRgb = 256, kLutXyz = 4000;
constexpr unsigned short kNoVote = 0xffff;

// SHOTColorEstimation::RGB2CIELAB + the /100, /120, /120 of computePointSHOT.  lut = sRGB_LUT[256]
// then sXYZ_LUT[4000], built on the host with the libm PCL would use (see shot_luts).  The table
// index int(v * 4000) can reach 4000 in PCL (one past the end, UB); it is clamped to 3999.
__device__ __forceinline__ float4 shot_rgb2lab(unsigned rgba, const float *__restrict__ lut)
{
  const float fr = lut[(rgba >> 16) & 0xffu], fg = lut[(rgba >> 8) & 0xffu], fb = lut[rgba & 0xffu];
  const float x = fr * 0.412453f + fg * 0.357580f + fb * 0.180423f;
  const float y = fr * 0.212671f + fg * 0.715160f + fb * 0.072169f;
  const float z = fr * 0.019334f + fg * 0.119193f + fb * 0.950227f;
  float vx = x / 0.95047f, vy = y, vz = z / 1.08883f;
  int ix = (int)(vx * 4000), iy = (int)(vy * 4000), iz = (int)(vz * 4000);
  ix = ix > 3999 ? 3999 : ix; iy = iy > 3999 ? 3999 : iy; iz = iz > 3999 ? 3999 : iz;
  vx = lut[kLutRgb + ix]; vy = lut[kLutRgb + iy]; vz = lut[kLutRgb + iz];
  float L = 116.0f * vy - 16.0f;
  if (L > 100) L = 100.0f;
  float A = 500.0f * (vx - vy);
  if (A > 120) A = 120.0f; else if (A < -120) A = -120.0f;
  float B2 = 200.0f * (vy - vz);
  if (B2 > 120) B2 = 120.0f; else if (B2 < -120) B2 = -120.0f;
  return make_float4(L / 100.0f, A / 120.0f, B2 / 120.0f, 0.0f);
}

__global__ void k_shot_lab(const float4 *__restrict__ pts, int n, const float *__restrict__ lut, float4 *__restrict__ lab)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lab[i] = shot_rgb2lab(__float_as_uint(pts[i].w), lut);
}

// Eigen-decomposition of a symmetric 3x3 in double: cyclic Jacobi, eigenvalues ascending, vectors in
// the columns of V -- the same text as oracle/o_shot.c::sym_eig3 (both stand in for
// Eigen::SelfAdjointEigenSolver<Matrix3d>; the axes' signs are fixed by the disambiguation).
__device__ void shot_sym_eig3(double a[3][3], double w[3], double V[3][3])
{
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (!(off > 4.93e-32 * (dg + 2.0 * off))) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p][q];
        if (apq == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const int r = 3 - p - q;
        const double app = a[p][p], aqq = a[q][q], apr = a[p][r], aqr = a[q][r];
        a[p][p] = app - t * apq;
        a[q][q] = aqq + t * apq;
        a[p][q] = a[q][p] = 0.0;
        a[p][r] = a[r][p] = c * apr - s * aqr;
        a[q][r] = a[r][q] = s * apr + c * aqr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2 - i; ++j)
      if (w[j + 1] < w[j]) {
        double t = w[j]; w[j] = w[j + 1]; w[j + 1] = t;
#pragma unroll
        for (int k = 0; k < 3; ++k) { t = V[k][j]; V[k][j] = V[k][j + 1]; V[k][j + 1] = t; }
      }
}

struct ShotVotes {                 // one neighbour: [channel][vote] = (volume << 8 | slot), value
  unsigned short vs[2][5];
  float val[2][5];
};

__device__ __forceinline__ void shot_vote(ShotVotes &r, int v, int vol, int slot_s, int slot_c, float val)
{
  r.vs[0][v] = (unsigned short)((vol << 8) | slot_s);
  r.vs[1][v] = (unsigned short)((vol << 8) | slot_c);
  r.val[0][v] = val; r.val[1][v] = val;
}

// interpolateDoubleChannel for one neighbour (shot.hpp), votes instead of "shot[...] +=".
// "shot[i] -= (float)x" is recorded as the vote -(float)x: a - b and a + (-b) round identically.
__device__ void shot_neighbour_votes(float dx, float dy, float dz, float d2, float4 n, float4 lab, float4 lab_ref,
                                     const float *rf, double radius, ShotVotes &r)
{
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int v = 0; v < 5; ++v) { r.vs[c][v] = kNoVote; r.val[c][v] = 0.0f; }
  if (!isfinite(n.x) || !isfinite(n.y) || !isfinite(n.z)) return;          // createBinDistanceShape: NaN normal
  double cosine = (double)(n.x * rf[6] + n.y * rf[7] + n.z * rf[8]);
  if (cosine > 1.0) cosine = 1.0;
  if (cosine < -1.0) cosine = -1.0;
  double bin_shape = ((1.0 + cosine) * kShotShapeBins) / 2;
  double color_distance =
      (double)((fabsf(lab_ref.x - lab.x) + ((fabsf(lab_ref.y - lab.y) + fabsf(lab_ref.z - lab.z)) / 2)) / 3);
  if (color_distance > 1.0) color_distance = 1.0;
  if (color_distance < 0.0) color_distance = 0.0;
  double bin_color = color_distance * kShotColorBins;

  const double distance = (double)sqrtf(d2);
  if (fabs(distance - 0.0) < 1e-15) return;
  double x_ref = (double)(dx * rf[0] + dy * rf[1] + dz * rf[2]);
  double y_ref = (double)(dx * rf[3] + dy * rf[4] + dz * rf[5]);
  double z_ref = (double)(dx * rf[6] + dy * rf[7] + dz * rf[8]);
  if (fabs(y_ref) < 1e-30) y_ref = 0;
  if (fabs(x_ref) < 1e-30) x_ref = 0;
  if (fabs(z_ref) < 1e-30) z_ref = 0;
  const double radius3_4 = (radius * 3) / 4, radius1_4 = radius / 4, radius1_2 = radius / 2;
  const int bit4 = ((y_ref > 0) || ((y_ref == 0.0) && (x_ref < 0))) ? 1 : 0;
  const int bit3 = ((x_ref > 0) || ((x_ref == 0.0) && (y_ref > 0))) ? !bit4 : bit4;
  int desc_index = (bit4 << 3) + (bit3 << 2);
  desc_index = desc_index << 1;
  if ((x_ref * y_ref > 0) || (x_ref == 0.0)) desc_index += (fabs(x_ref) >= fabs(y_ref)) ? 0 : 4;
  else desc_index += (fabs(x_ref) > fabs(y_ref)) ? 4 : 0;
  desc_index += z_ref > 0 ? 1 : 0;
  desc_index += (distance > radius1_2) ? 2 : 0;

  const int step_shape = (int)floor(bin_shape + 0.5);
  const int step_color = (int)floor(bin_color + 0.5);
  bin_shape -= step_shape;
  bin_color -= step_color;
  double w_shape = 1 - fabs(bin_shape), w_color = 1 - fabs(bin_color);
  // vote 0: the neighbouring cosine / colour slot of the own volume
  r.vs[0][0] = (unsigned short)((desc_index << 8) |
                                (bin_shape > 0 ? (step_shape + 1) % kShotShapeBins : (step_shape - 1 + kShotShapeBins) % kShotShapeBins));
  r.val[0][0] = bin_shape > 0 ? (float)bin_shape : -(float)bin_shape;
  r.vs[1][0] = (unsigned short)((desc_index << 8) |
                                (bin_color > 0 ? (step_color + 1) % kShotColorBins : (step_color - 1 + kShotColorBins) % kShotColorBins));
  r.val[1][0] = bin_color > 0 ? (float)bin_color : -(float)bin_color;

  // vote 1: radial shells
  if (distance > radius1_2) {
    const double rd = (distance - radius3_4) / radius1_2;
    if (distance > radius3_4) { w_shape += 1 - rd; w_color += 1 - rd; }
    else {
      w_shape += 1 + rd; w_color += 1 + rd;
      shot_vote(r, 1, desc_index - 2, step_shape, step_color, -(float)rd);
    }
  } else {
    const double rd = (distance - radius1_4) / radius1_2;
    if (distance < radius1_4) { w_shape += 1 + rd; w_color += 1 + rd; }
    else {
      w_shape += 1 - rd; w_color += 1 - rd;
      shot_vote(r, 1, desc_index + 2, step_shape, step_color, (float)rd);
    }
  }
  // vote 2: inclination
  double inc_cos = z_ref / distance;
  if (inc_cos < -1.0) inc_cos = -1.0;
  if (inc_cos > 1.0) inc_cos = 1.0;
  const double inclination = acos(inc_cos);
  constexpr double kRad45 = 0.78539816339744830961566084581988, kRad90 = 1.5707963267948966192313216916398,
                   kRad135 = 2.3561944901923449288469825374596, kRadPi78 = 2.7488935718910690836548129603691;
  if (inclination > kRad90 || (fabs(inclination - kRad90) < 1e-30 && z_ref <= 0)) {
    const double id = (inclination - kRad135) / kRad90;
    if (inclination > kRad135) { w_shape += 1 - id; w_color += 1 - id; }
    else {
      w_shape += 1 + id; w_color += 1 + id;
      shot_vote(r, 2, desc_index + 1, step_shape, step_color, -(float)id);
    }
  } else {
    const double id = (inclination - kRad45) / kRad90;
    if (inclination < kRad45) { w_shape += 1 + id; w_color += 1 + id; }
    else {
      w_shape += 1 - id; w_color += 1 - id;
      shot_vote(r, 2, desc_index - 1, step_shape, step_color, (float)id);
    }
  }
  // vote 3: azimuth
  if (y_ref != 0.0 || x_ref != 0.0) {
    const double azimuth = atan2(y_ref, x_ref);
    const int sel = desc_index >> 2;
    double ad = (azimuth - (-kRadPi78 + kRad45 * sel)) / kRad45;
    ad = fmax(-0.5, fmin(ad, 0.5));
    if (ad > 0) {
      w_shape += 1 - ad; w_color += 1 - ad;
      shot_vote(r, 3, (desc_index + 4) % kShotSectors, step_shape, step_color, (float)ad);
    } else {
      w_shape += 1 + ad; w_color += 1 + ad;
      shot_vote(r, 3, (desc_index - 4 + kShotSectors) % kShotSectors, step_shape, step_color, -(float)ad);
    }
  }
  // vote 4: the own slot
  r.vs[0][4] = (unsigned short)((desc_index << 8) | step_shape); r.val[0][4] = (float)w_shape;
  r.vs[1][4] = (unsigned short)((desc_index << 8) | step_color); r.val[1][4] = (float)w_color;
}

// rows: optional list of keypoints to process (neighbourhoods that overflowed the LDS keys), their
// keys then live in `scratch` (cap entries per row, cap a power of two)
__global__ void __launch_bounds__(64)
k_shot(const float4 *__restrict__ kp, int nk, GridView g, const float4 *__restrict__ pts /* original order: xyz, rgba */,
       const float4 *__restrict__ nrm, const float4 *__restrict__ lab, const float *__restrict__ lut, float radius_f, double radius,
       float r2, const int *__restrict__ rows, unsigned long long *__restrict__ scratch, int cap, float *__restrict__ desc /* [nk][1344] */,
       float *__restrict__ rf_out /* [nk][9] */, int *__restrict__ valid,
       int *__restrict__ overflow /* [0] count, [1..] keypoint ids, [nk + 1] max count */)
{
  __shared__ unsigned long long s_keys[kShotCap];
  __shared__ float s_hist[kShotDim];
  __shared__ float4 s_v[64];
  __shared__ unsigned short s_vs[64][2][5];
  __shared__ float s_val[64][2][5];
  __shared__ int s_m;
  const int lane = threadIdx.x;
  const int k = rows ? rows[blockIdx.x] : (int)blockIdx.x;
  unsigned long long *keys = rows ? scratch + (size_t)blockIdx.x * cap : s_keys;
  const float qnan = __uint_as_float(0x7fc00000u);
  float *out = desc + (size_t)k * kShotDim;
  auto give_up = [&]() {   // computeFeature: NaN descriptor and NaN frame
    for (int b = lane; b < kShotDim; b += 64) out[b] = qnan;
    if (lane < 9) rf_out[(size_t)k * 9 + lane] = qnan;
    if (lane == 0) valid[k] = 0;
  };
  if (lane == 0) s_m = 0;
  __syncthreads();
  const float4 q = kp[k];
  if (!isfinite(q.x) || !isfinite(q.y) || !isfinite(q.z)) { give_up(); return; }

  // ---- 1. gather (any order) ...
  const float ri = radius_f * 1.0001f + 1e-4f;
  if (!(cell_floor(q.x + ri, g.minx, g.inv) < 0 || cell_floor(q.x - ri, g.minx, g.inv) > g.dx - 1)) {
    const int x0 = clampi(cell_floor(q.x - ri, g.minx, g.inv), 0, g.dx - 1), x1 = clampi(cell_floor(q.x + ri, g.minx, g.inv), 0, g.dx - 1);
    int y0 = cell_floor(q.y - ri, g.miny, g.inv), y1 = cell_floor(q.y + ri, g.miny, g.inv);
    int z0 = cell_floor(q.z - ri, g.minz, g.inv), z1 = cell_floor(q.z + ri, g.minz, g.inv);
    y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
    y1 = y1 > g.dy - 1 ? g.dy - 1 : y1; z1 = z1 > g.dz - 1 ? g.dz - 1 : z1;
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int row = (z * g.dy + y) * g.dx;
        const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
        for (int j = b + lane; j < e; j += 64) {
          const float4 p = g.pts[j];
          const float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
          if (d2 < r2) {
            const int slot = atomicAdd(&s_m, 1);
            if (slot < cap) keys[slot] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p.w);
          }
        }
      }
  }
  __syncthreads();
  const int m = s_m;
  if (m > cap) {
    if (lane == 0 && !rows) {
      const int o = atomicAdd(&overflow[0], 1);
      overflow[1 + o] = k;
      atomicMax(&overflow[nk + 1], m);
    }
    return;
  }
  if (m == 0) { give_up(); return; }
  // ---- ... and sort by (distance, index)
  int n2 = 1;
  while (n2 < m) n2 <<= 1;
  for (int i = m + lane; i < n2; i += 64) keys[i] = ~0ull;
  __syncthreads();
  for (int k2 = 2; k2 <= n2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (n2 >> 1); t += 64) {
        const int i = 2 * t - (t & (j - 1)), l = i + j;
        const bool up = (i & k2) == 0;
        const unsigned long long a = keys[i], b = keys[l];
        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
      }
      __syncthreads();
    }

  // ---- 2. local reference frame (shot_lrf.hpp getLocalRF)
  // lanes 0..5 own a covariance entry, lane 6 the weight sum: sequential double chains in neighbour order
  const int ia = lane == 0 || lane == 1 || lane == 2 ? 0 : (lane == 3 || lane == 4 ? 1 : 2);
  const int ib = lane == 0 ? 0 : (lane == 1 || lane == 3 ? 1 : 2);
  double chain = 0.0;
  int n_valid = 0;
  for (int b0 = 0; b0 < m; b0 += 64) {
    const int i = b0 + lane;
    bool ok = false;
    if (i < m) {
      const unsigned long long key = keys[i];
      const float4 p = pts[(unsigned)(key & 0xffffffffull)];
      ok = !(p.x == q.x && p.y == q.y && p.z == q.z);
      s_v[lane] = make_float4(p.x - q.x, p.y - q.y, p.z - q.z, __uint_as_float((unsigned)(key >> 32)));
    }
    const unsigned long long mask = ballot(ok);
    n_valid += __popcll(mask);
    __syncthreads();
    if (lane < 7) {
      const int bn = min(64, m - b0);
      for (int e = 0; e < bn; ++e)
        if ((mask >> e) & 1ull) {
          const float4 r = s_v[e];
          const double dist = radius - (double)sqrtf(r.w);
          const double va = ia == 0 ? (double)r.x : (ia == 1 ? (double)r.y : (double)r.z);
          const double vb = ib == 0 ? (double)r.x : (ib == 1 ? (double)r.y : (double)r.z);
          chain += lane == 6 ? dist : dist * (va * vb);
        }
    }
    __syncthreads();
  }
  if (n_valid < 5) { give_up(); return; }
  double cov[3][3], w[3], V[3][3];
  {
    const double sum = __shfl(chain, 6, 64);
    const double c00 = __shfl(chain, 0, 64) / sum, c01 = __shfl(chain, 1, 64) / sum, c02 = __shfl(chain, 2, 64) / sum;
    const double c11 = __shfl(chain, 3, 64) / sum, c12 = __shfl(chain, 4, 64) / sum, c22 = __shfl(chain, 5, 64) / sum;
    cov[0][0] = c00; cov[0][1] = cov[1][0] = c01; cov[0][2] = cov[2][0] = c02;
    cov[1][1] = c11; cov[1][2] = cov[2][1] = c12; cov[2][2] = c22;
  }
  shot_sym_eig3(cov, w, V);
  if (!isfinite(w[0]) || !isfinite(w[1]) || !isfinite(w[2])) { give_up(); return; }
  double v1[3] = {V[0][2], V[1][2], V[2][2]};   // largest eigenvalue: x axis
  double v3[3] = {V[0][0], V[1][0], V[2][0]};   // smallest: z axis
  {
    // disambiguation: majority of (neighbour - keypoint) . axis >= 0; on a tie the five neighbours
    // around the median of the valid list decide
    int plus_x = 0, plus_z = 0, five_x = 0, five_z = 0, base = 0;
    const int median = n_valid / 2;
    for (int b0 = 0; b0 < m; b0 += 64) {
      const int i = b0 + lane;
      bool ok = false;
      double dpx = 0.0, dpz = 0.0;
      if (i < m) {
        const float4 p = pts[(unsigned)(keys[i] & 0xffffffffull)];
        ok = !(p.x == q.x && p.y == q.y && p.z == q.z);
        const double vx = (double)(p.x - q.x), vy = (double)(p.y - q.y), vz = (double)(p.z - q.z);
        dpx = vx * v1[0] + vy * v1[1] + vz * v1[2];
        dpz = vx * v3[0] + vy * v3[1] + vz * v3[2];
      }
      const unsigned long long mask = ballot(ok);
      const int rank = base + __popcll(mask & ((1ull << lane) - 1ull));
      const bool mid = ok && rank >= median - 2 && rank <= median + 2;
      plus_x += __popcll(ballot(ok && dpx >= 0));
      plus_z += __popcll(ballot(ok && dpz >= 0));
      five_x += __popcll(ballot(mid && dpx > 0));
      five_z += __popcll(ballot(mid && dpz > 0));
      base += __popcll(mask);
    }
    int p = 2 * plus_x - n_valid;
    if (p == 0 ? five_x < 3 : p < 0) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
    p = 2 * plus_z - n_valid;
    if (p == 0 ? five_z < 3 : p < 0) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
  }
  float rf[9];
#pragma unroll
  for (int a = 0; a < 3; ++a) { rf[a] = (float)v1[a]; rf[6 + a] = (float)v3[a]; }
  rf[3] = rf[7] * rf[2] - rf[8] * rf[1];
  rf[4] = rf[8] * rf[0] - rf[6] * rf[2];
  rf[5] = rf[6] * rf[1] - rf[7] * rf[0];
  if (lane < 9) {
    float v = rf[0];
#pragma unroll
    for (int a = 1; a < 9; ++a) v = lane == a ? rf[a] : v;
    rf_out[(size_t)k * 9 + lane] = v;
  }

  // ---- 3. votes, applied in neighbour order; lane = (channel, volume)
  for (int b = lane; b < kShotDim; b += 64) s_hist[b] = 0.0f;
  const float4 lab_ref = shot_rgb2lab(__float_as_uint(q.w), lut);
  const int ch = lane >> 5, vol = lane & 31;
  float *hist = s_hist + (ch ? kShotColorOffset + vol * (kShotColorBins + 1) : vol * (kShotShapeBins + 1));
  __syncthreads();
  for (int b0 = 0; b0 < m; b0 += 64) {
    const int i = b0 + lane;
    if (i < m) {
      const unsigned long long key = keys[i];
      const unsigned oi = (unsigned)(key & 0xffffffffull);
      const float4 p = pts[oi];
      ShotVotes r;
      shot_neighbour_votes(p.x - q.x, p.y - q.y, p.z - q.z, __uint_as_float((unsigned)(key >> 32)), nrm[oi], lab[oi], lab_ref, rf,
                           radius, r);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 5; ++v) { s_vs[lane][c][v] = r.vs[c][v]; s_val[lane][c][v] = r.val[c][v]; }
    }
    __syncthreads();
    const int bn = min(64, m - b0);
    for (int e = 0; e < bn; ++e) {
#pragma unroll
      for (int v = 0; v < 5; ++v) {
        const unsigned p = s_vs[e][ch][v];
        if ((int)(p >> 8) == vol) hist[p & 0xffu] += s_val[e][ch][v];
      }
    }
    __syncthreads();
  }

  // ---- 4. normalizeHistogram: acc += shot[j] * shot[j] (float product, double sum, in order)
  double acc = 0.0;
  if (lane == 0)
    for (int j = 0; j < kShotDim; ++j) { const float s = s_hist[j]; acc += (double)(s * s); }
  acc = sqrt(__shfl(acc, 0, 64));
  const float nrmf = (float)acc;
  bool fin = true;
  for (int b = lane; b < kShotDim; b += 64) {
    const float v = s_hist[b] / nrmf;
    fin = fin && isfinite(v);
    out[b] = v;
  }
  const bool all_fin = __all(fin);
  if (lane == 0) valid[k] = all_fin ? 1 : 0;
}

__global__ void k_shot_compact(const float *__restrict__ in, const int *__restrict__ flags, const int *__restrict__ pos, int n, int dim,
                               float *__restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * dim) return;
  const int r = (int)(e / dim), cidx = (int)(e % dim);
  if (flags[r]) out[(size_t)pos[r] * dim + cidx] = in[e];
}

// sRGB_LUT[256] and sXYZ_LUT[4000] of SHOTColorEstimation (static tables filled on first use in
// PCL), computed with the host's powf like PCL does
static void shot_luts(float *lut)
{
  for (int i = 0; i < kLutRgb; ++i) {
    const float f = static_cast<float>(i) / 255.0f;
    if (f > 0.04045) lut[i] = powf((f + 0.055f) / 1.055f, 2.4f);
    else lut[i] = f / 12.92f;
  }
  for (int i = 0; i < kLutXyz; ++i) {
    const float f = static_cast<float>(i) / 4000.0f;
    if (f > 0.008856) lut[kLutRgb + i] = powf(f, 0.3333f);
    else lut[kLutRgb + i] = static_cast<float>((7.787 * f) + (16.0 / 116.0));
  }
}

mm3d_desc *compute_shot(Context *c, const mm3d_cloud *points, const mm3d_normals *normals, mm3d_cloud *keypoints, double radius)
{
  MM3D_REQUIRE(normals->n == points->n, "computeLocalDescriptors: normals and points differ in size");
  auto *res = new mm3d_desc();
  res->dim = kShotDim;
  res->type = MM3D_DESC_SHOT;
  const int nk = (int)keypoints->n;
  if (nk == 0) { res->n = 0; res->data = DevBuf<float>(c, 0); return res; }
  const float r2 = (float)(radius * radius);
  const Grid &g = cloud_grid(c, points, (float)(radius * 0.5));
  auto drop_all = [&]() {
    res->n = 0; res->data = DevBuf<float>(c, 0);
    keypoints->pts = DevBuf<float4>(c, 0); keypoints->n = 0; keypoints->grids.clear(); keypoints->host.clear();
    keypoints->reset_caches();
  };
  if (g.n == 0) { drop_all(); return res; }
  const int n = (int)points->n;
  DevBuf<float> lut(c, kLutRgb + kLutXyz);
  {
    float *h = (float *)c->pin((kLutRgb + kLutXyz) * sizeof(float));
    shot_luts(h);
    MM3D_HIP(hipMemcpyAsync(lut.get(), h, (kLutRgb + kLutXyz) * sizeof(float), hipMemcpyHostToDevice, c->stream));
  }
  DevBuf<float4> lab(c, (size_t)n);
  MM3D_LAUNCH(c, "shot_lab", n * 32.0, k_shot_lab, dim3(div_up((size_t)n, 256)), dim3(256), 0, (const float4 *)points->pts.get(), n,
              (const float *)lut.get(), lab.get());
  DevBuf<float> raw(c, (size_t)nk * kShotDim), rf(c, (size_t)nk * 9);
  DevBuf<int> valid(c, (size_t)nk + 1), overflow(c, (size_t)nk + 2);
  MM3D_HIP(hipMemsetAsync(valid.get(), 0, ((size_t)nk + 1) * sizeof(int), c->stream));
  MM3D_HIP(hipMemsetAsync(overflow.get(), 0, ((size_t)nk + 2) * sizeof(int), c->stream));
  MM3D_LAUNCH(c, "shot", nk * (200.0 * 48.0 * 3.0 + 5412.0), k_shot, dim3(nk), dim3(64), 0, (const float4 *)keypoints->pts.get(), nk, g.view(),
              (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), (const float4 *)lab.get(), (const float *)lut.get(),
              (float)radius, radius, r2, (const int *)nullptr, (unsigned long long *)nullptr, kShotCap, raw.get(), rf.get(), valid.get(),
              overflow.get());
  int *h = (int *)c->pin(64);
  MM3D_HIP(hipMemcpyAsync(h, overflow.get(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MM3D_HIP(hipMemcpyAsync(h + 1, overflow.get() + nk + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  if (h[0] > 0) {
    const int n_over = h[0];
    int cap = kShotCap;
    while (cap < h[1]) cap <<= 1;
    const double bytes = (double)n_over * cap * sizeof(unsigned long long);
    if (bytes > 8e9) throw Error(MM3D_EUNSUPPORTED, "SHOT: neighbourhoods too large for the scratch pass (reduce descriptor_radius)");
    DevBuf<unsigned long long> scratch(c, (size_t)n_over * cap);
    MM3D_LAUNCH(c, "shot", n_over * (cap * 48.0 * 3.0 + 5412.0), k_shot, dim3(n_over), dim3(64), 0, (const float4 *)keypoints->pts.get(), nk,
                g.view(), (const float4 *)points->pts.get(), (const float4 *)normals->nrm.get(), (const float4 *)lab.get(),
                (const float *)lut.get(), (float)radius, radius, r2, (const int *)(overflow.get() + 1), scratch.get(), cap, raw.get(),
                rf.get(), valid.get(), overflow.get());
    c->sync();
  }
  // prune invalid descriptors and the same keypoints (features.cpp:118-143)
  DevBuf<int> vpos(c, (size_t)nk + 1);
  exclusive_scan_int(c, valid.get(), vpos.get(), (size_t)nk + 1);
  MM3D_HIP(hipMemcpyAsync(h, vpos.get() + nk, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sync();
  const int nv = h[0];
  res->n = (size_t)nv;
  if (nv == nk) {
    res->data = std::move(raw);
    res->rf = std::move(rf);
  } else {
    res->data = DevBuf<float>(c, (size_t)nv * kShotDim);
    res->rf = DevBuf<float>(c, (size_t)nv * 9);
    DevBuf<float4> kp2(c, nv);
    if (nv) {
      MM3D_LAUNCH(c, "compact_rows", nk * 10824.0, k_shot_compact, dim3(div_up((size_t)nk * kShotDim, 256)), dim3(256), 0,
                  (const float *)raw.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, kShotDim, res->data.get());
      MM3D_LAUNCH(c, "compact_rows", nk * 72.0, k_shot_compact, dim3(div_up((size_t)nk * 9, 256)), dim3(256), 0, (const float *)rf.get(),
                  (const int *)valid.get(), (const int *)vpos.get(), nk, 9, res->rf.get());
      MM3D_LAUNCH(c, "compact_rows", nk * 32.0, k_shot_compact, dim3(div_up((size_t)nk * 4, 256)), dim3(256), 0,
                  (const float *)keypoints->pts.get(), (const int *)valid.get(), (const int *)vpos.get(), nk, 4, (float *)kp2.get());
    }
    c->sync();
    keypoints->pts = std::move(kp2);
    keypoints->n = (size_t)nv;
    keypoints->grids.clear();
    keypoints->host.clear();
    keypoints->reset_caches();
  }
  c->sync();
  return res;
}

}  // namespace mm3d
