// nn.hip -- ICP and transformScore: exact 1-NN of every (transformed) source point in the target
// grid, fused with the reduction the caller needs (K12/K13 in SURVEY 2.2).
//
// estimateTransformICP  R/src/matching.cpp:196-221 -> pcl::IterativeClosestPoint (point-to-point,
//                       TransformationEstimationSVD/Umeyama, DefaultConvergenceCriteria)
// transformScore        R/src/matching.cpp:259-268 -> TransformationValidationEuclidean
//
// Wave-cooperative search.  A wave owns one work item of the source's Hilbert order (<= 64 points
// forming a compact patch), so the cells its queries can touch form a small box of the target grid.  The wave
//   1. reads one distance-transform byte per lane (how many cells to the nearest occupied cell;
//      out of range -> that lane is done),
//   2. takes the bounding box of its lanes' cells, grows it by the largest radius any lane needs,
//   3. streams the box's rows (contiguous spans of the cell-sorted target) through LDS with
//      coalesced 16-byte loads, 512 points per tile,
//   4. every lane scans the tile out of LDS (same address for all lanes = broadcast reads; the tile is
//      kept per component, so four candidates are one 16-byte read and their distances pack),
//   5. a lane is finished when its best distance is within what the box provably covers; otherwise
//      the box grows once more to the radius that lane needs.
// Candidate traffic is therefore LDS traffic; HBM sees the source once (16 B/point) and each target
// span once per wave.  The ICP iteration stays two launches and no host round trip:
// icp_corr_reduce (this search + 17 double partial sums per block through wave shuffles) and
// icp_finalize (Umeyama by 3x3 Jacobi SVD, accumulate, PCL's convergence tests, all on the device).
// Algorithmic traffic: 12 B per source point per iteration.
#include <cfloat>
#include <cstddef>
#include <type_traits>

#include "device_util.hpp"
#include "linalg_shared.hpp"

namespace mm3d {

constexpr int kAcc = 17;     // sum p(3) | sum q(3) | sum q p^T (9, row = q) | sum d2 | count
#ifndef MM3D_NN_TILE
#define MM3D_NN_TILE 256
#endif
#ifndef MM3D_NN_WPE
#define MM3D_NN_WPE 4
#endif
#ifdef MM3D_NN_WPE
#define MM3D_NN_ATTR __attribute__((amdgpu_waves_per_eu(MM3D_NN_WPE, MM3D_NN_WPE)))
#else
#define MM3D_NN_ATTR
#endif
constexpr int kTile = MM3D_NN_TILE;   // staged target points per wave and tile (8 KiB of LDS)
#ifndef MM3D_NN_ROWS_PER_LANE
#define MM3D_NN_ROWS_PER_LANE 4
#endif
// MM3D_NN_PREFETCH=1: the next tile's gathers are issued into registers before this tile is scanned.  Measured and left off:
// 16 more VGPRs (112: four waves per SIMD instead of five for the ICP variant) for a latency that the other resident waves
// already cover -- headline 955 against 960 map-pairs/s, 64 x 50 k 10 250 against 10 520 (interleaved A/B, round 4).
#ifndef MM3D_NN_PREFETCH
#define MM3D_NN_PREFETCH 0
#endif
#ifndef MM3D_NN_TIGHT_BOX
#define MM3D_NN_TIGHT_BOX 1
#endif
#ifndef MM3D_NN_SHELL
#define MM3D_NN_SHELL 1
#endif
#ifndef MM3D_NN_LOWER_BOUND
#define MM3D_NN_LOWER_BOUND 1
#endif
#ifndef MM3D_NN_CORNERS
#define MM3D_NN_CORNERS 1
#endif
constexpr bool kPrefetch = MM3D_NN_PREFETCH != 0;
constexpr int kRowsPerLane = MM3D_NN_ROWS_PER_LANE;   // row headers a lane reads per chunk
constexpr int kRows = kWave * kRowsPerLane;           // rows of the box per chunk (power of two: the slot -> row search halves it)

#ifdef MM3D_NN_STATS
__device__ unsigned long long g_nn_stats[64];   // 0 waves, 1 passes, 2 row chunks, 3 staged points, 4 active lanes at pass, 5 rows, 6 max wave cycles, 7 sum wave cycles, 8.. log2 histogram of wave cycles
#define MM3D_STAT(i_, v_) do { if (MM3D_NN_STATS == 1 && lane == 0) atomicAdd(&g_nn_stats[i_], (unsigned long long)(v_)); } while (0)
#define MM3D_TICK(var_) const long long var_ = wall_clock64()
// (phase ticks are summed in registers and flushed once per wave: an atomic per chunk on one word slowed the kernel threefold)
#define MM3D_TOCK(i_, from_) do { stat_ticks[(i_) - 32] += wall_clock64() - (from_); } while (0)
#else
#define MM3D_STAT(i_, v_)
#define MM3D_TICK(var_)
#define MM3D_TOCK(i_, from_)
#endif

struct IcpState {
  float T[16];      // cumulative transform applied to the original source points (starts at the guess)
  float Tinc[16];
  double prev_mse;
  double rot_thresh, trans_thresh;
  int iters, done, converged, n_corr, max_iter;
  int scored;      // the score of the final transform has been taken (k_score_finalize)
};

// One (source cloud, target grid) search of a launch: blockIdx.y picks the job, so the searches of several map
// pairs that are ready at the same time share one launch (many small maps: a launch per pair leaves most of the
// chip idle, and only four launches run at a time).
struct NnJob {
  const float4 *src;          // source points in Hilbert order
  const int2 *items;          // their work items
  int n_items, nblocks;       // blocks this job uses of the launch's grid.x
  int split;                  // 1: one work item per block (partials per item), 0: four items per block
  GridView g;                 // target grid
  const float4 *tgt_ref;      // target points in reference order
  IcpState *st;               // ICP: the pair's state; score: T is read from its head (or from Tc)
  const float *Tc;            // score: the transform, when it is not the ICP state's
  double *partials;           // [nblocks][kAcc]
  double *out;                // score: {sum d2, count}
  int max_ring;
};

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, kWave));
  return v;
}
// LDS written by some lanes of a wave and read by others: order the accesses for the compiler;
// the hardware executes one wave's DS operations in order.
__device__ __forceinline__ void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODE 0: ICP (transform from the device state, accumulate Umeyama moments)
// MODE 1: transformScore (transform from Tc, accumulate sum d2 / count for d2 <= max_d2)
// SPLIT 1: one work item per wave, four items per block.
// SPLIT 4: one work item per BLOCK: its four waves hold the same 64 points and the same box, each stages and
//          scans a quarter of the box's candidates, and the four minima meet in LDS after every pass.  Same
//          result, a quarter of the time per item: for a source of a few hundred items (a 50 k point map) the
//          chip is mostly idle and the kernel's duration IS one wave's scan.
template <int MODE, int SPLIT>
__global__ void __launch_bounds__(256) MM3D_NN_ATTR
k_nn_wave(const NnJob *__restrict__ jobs, float max_d2, float rmax)
{
  const NnJob &job = jobs[blockIdx.y];
  if ((int)blockIdx.x >= job.nblocks) return;            // the grid is as wide as the batch's largest job
  const float4 *__restrict__ src = job.src;
  const int2 *__restrict__ items = job.items;
  const int n_items = job.n_items;
  const GridView g = job.g;
  const float4 *__restrict__ tgt_ref = job.tgt_ref;
  const IcpState *__restrict__ st = job.st;
  const float *__restrict__ Tc = job.Tc ? job.Tc : job.st->T;
  double *__restrict__ partials = job.partials;
  const int max_ring = job.max_ring;
  __shared__ float Ts[16];
  __shared__ double red[4][kAcc];
  // staged candidates, one array per component: four candidates' x (y, z, index) are ONE 16-byte
  // broadcast read, and the distance arithmetic of candidate pairs packs into v_pk_* instructions
  __shared__ __attribute__((aligned(16))) float s_cx[4][kTile], s_cy[4][kTile], s_cz[4][kTile];
  __shared__ __attribute__((aligned(16))) unsigned s_cw[4][kTile];
  __shared__ int s_off[4][kRows];
  __shared__ int s_beg[4][kRows];
  __shared__ unsigned long long s_merge[SPLIT == 4 ? 4 : 1][64];
  if (MODE == 0 && st->done) return;
  // the score of a pair's ICP result: once, in the first round after its ICP has finished
  if (MODE == 1 && st && (!st->done || st->scored)) return;
  if (threadIdx.x < 16) Ts[threadIdx.x] = (MODE == 0) ? st->T[threadIdx.x] : Tc[threadIdx.x];
  __syncthreads();
  const unsigned bid = xcd_remap(blockIdx.x, (unsigned)job.nblocks);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = SPLIT == 4 ? (int)bid : (int)bid * 4 + wave;   // one work item (<= 64 points of one coarse block)
  const int2 it = item < n_items ? items[item] : make_int2(0, 0);
  const int i = it.x + lane;
  const bool valid = lane < it.y;
  float3 p = make_float3(0.f, 0.f, 0.f);
  if (valid) {
    const float4 s = src[i];
    p = xform(Ts, s.x, s.y, s.z);
  }
  const int cx = cell_floor(p.x, g.minx, g.inv), cy = cell_floor(p.y, g.miny, g.inv), cz = cell_floor(p.z, g.minz, g.inv);
  bool active = valid;
  int need = max_ring;          // radius (in cells around the lane's own cell) the lane wants scanned
  float reach_cap = rmax;       // no target point of interest is farther: min(rmax, what the distance transform guarantees)
  if (active) {
    const bool inside = cx >= 0 && cx < g.dx && cy >= 0 && cy < g.dy && cz >= 0 && cz < g.dz;
    if (inside) {
      const int d0 = g.dt[((size_t)cz * g.dy + cy) * g.dx + cx];
      if (d0 > max_ring) active = false;          // nothing within range of this cell
      need = d0 > 1 ? d0 : 1;
      // an occupied cell d0 cells away holds a point within sqrt(3) (d0 + 1) cells of this one (its farthest corner): an upper
      // bound of the nearest-neighbour distance that needs no candidate (used when the corner filter below has dropped them)
      if (MM3D_NN_CORNERS && d0 <= max_ring) reach_cap = fminf(rmax, 1.7321f * (float)(d0 + 1) * g.cell * 1.0001f + 1e-5f);
#if MM3D_NN_LOWER_BOUND
      // The nearest occupied cell is d0 cells away along some axis, so no target point is nearer than (d0 - 1) cells plus the
      // way from this point to the nearest face of its own cell.  Where that already exceeds rmax the lane has no neighbour
      // in range and need not search: with 0.25 m cells and a range of 1 m that is every lane with d0 = 6 and nearly every
      // one with d0 = 5 -- a twelfth of the lanes of a headline ICP launch, and the ones whose boxes (11 and 13 cells wide)
      // were the largest of their waves (round 5; max_ring = ceil(rmax / cell) + 1 admitted them).
      if (active && d0 >= 2) {
        const float fx = p.x - (g.minx + (float)cx * g.cell), fy = p.y - (g.miny + (float)cy * g.cell), fz = p.z - (g.minz + (float)cz * g.cell);
        const float mface = fmaxf(fminf(fminf(fminf(fx, g.cell - fx), fminf(fy, g.cell - fy)), fminf(fz, g.cell - fz)), 0.0f);
        // (the slack scales with the cell and with how far the grid lies from the origin, like the corner filter's `edge`: a
        // point can sit outside its nominal cell faces by the rounding of minx + cx * cell, 3e-5 .. 6e-5 m a kilometre out)
        const float lb_edge = 1e-3f * g.cell + 1e-6f * (fabsf(g.minx) + fabsf(g.miny) + fabsf(g.minz) + (float)(g.dx + g.dy + g.dz) * g.cell);
        if ((float)(d0 - 1) * g.cell + mface * 0.999f - lb_edge >= rmax) active = false;
      }
#endif
    } else {
      // outside the grid: farther than rmax from its box means no neighbour in range
      const float ex = fmaxf(fmaxf(g.minx - p.x, p.x - (g.minx + g.dx * g.cell)), 0.0f);
      const float ey = fmaxf(fmaxf(g.miny - p.y, p.y - (g.miny + g.dy * g.cell)), 0.0f);
      const float ez = fmaxf(fmaxf(g.minz - p.z, p.z - (g.minz + g.dz * g.cell)), 0.0f);
      if (!(fmaxf(ex, fmaxf(ey, ez)) <= rmax)) active = false;
    }
  }
  // (d2 bits, original index) as one 64-bit key: d2 >= 0 so its bits order like the value, and the
  // low word breaks ties towards the lower original index, like the CPU path
  // The key starts at (max_d2, no index): a candidate beyond the correspondence distance is never a correspondence, so it
  // need not be found -- and must not send its group of four through the key-forming path below.  A candidate AT max_d2
  // still wins (any real index is below 0xffffffff), and "nothing in range yet" is the index word 0xffffffff.
  unsigned long long bkey = ((unsigned long long)__float_as_uint(max_d2) << 32) | 0xffffffffull;
  float best = INFINITY, bestd = INFINITY;

#ifdef MM3D_NN_STATS
  long long stat_ticks[3] = {0, 0, 0};
  const long long t_begin = wall_clock64();
  for (int e = 1; e <= 8; ++e) {          // what ring the lanes ask for before the first pass
    const int c_ = __popcll(ballot(active && (e < 8 ? need == e : need >= 8)));
    if (MM3D_NN_STATS == 1 && lane == 0 && c_) atomicAdd(&g_nn_stats[55 + e], (unsigned long long)c_);
  }
#endif
  bool have_old = false;                 // the previous pass's box (wave-uniform; SPLIT 4: the same in the four waves)
  int ox0 = 0, ox1 = -1, oy0 = 0, oy1 = -1, oz0 = 0, oz1 = -1;
  for (int pass = 0; pass < 64; ++pass) {
    if (!ballot(active)) break;
    MM3D_STAT(1, 1);
    MM3D_STAT(4, __popcll(ballot(active)));
    if (pass == 0) MM3D_STAT(0, 1);
    // box = bounding box of the active lanes' OWN boxes (a lane's cell grown by the radius that lane needs).  Until round 5 it
    // was the bounding box of the lanes' cells grown by the LARGEST radius any of them needs; the needs of a patch's lanes
    // differ (the distance transform changes by up to a cell per cell), and a lane with a small need at one end of the patch
    // does not have to be covered as if it had the largest.
    const int E = wave_max_i(active ? need : 0);
    MM3D_TICK(t_pass);
#if MM3D_NN_TIGHT_BOX
    const int bx0 = wave_min_i(active ? cx - need : 0x7fffffff), bx1 = wave_max_i(active ? cx + need : -0x7fffffff);
    const int by0 = wave_min_i(active ? cy - need : 0x7fffffff), by1 = wave_max_i(active ? cy + need : -0x7fffffff);
    const int bz0 = wave_min_i(active ? cz - need : 0x7fffffff), bz1 = wave_max_i(active ? cz + need : -0x7fffffff);
#else
    const int bx0 = wave_min_i(active ? cx : 0x7fffffff) - E, bx1 = wave_max_i(active ? cx : -0x7fffffff) + E;
    const int by0 = wave_min_i(active ? cy : 0x7fffffff) - E, by1 = wave_max_i(active ? cy : -0x7fffffff) + E;
    const int bz0 = wave_min_i(active ? cz : 0x7fffffff) - E, bz1 = wave_max_i(active ? cz : -0x7fffffff) + E;
#endif
    const int x0 = max(bx0, 0), x1 = min(bx1, g.dx - 1);
    const int y0 = max(by0, 0), y1 = min(by1, g.dy - 1);
    const int z0 = max(bz0, 0), z1 = min(bz1, g.dz - 1);
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int nrows = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
    // What this pass's box proves for a lane -- every target point nearer than `guard` has been staged -- is known before the
    // box is read (a face on the grid's own border proves everything beyond it).
    float guard = 0.0f;
    if (active) {
      const float gx0 = (bx0 > 0) ? p.x - (g.minx + (float)bx0 * g.cell) : INFINITY;
      const float gx1 = (bx1 < g.dx - 1) ? (g.minx + (float)(bx1 + 1) * g.cell) - p.x : INFINITY;
      const float gy0 = (by0 > 0) ? p.y - (g.miny + (float)by0 * g.cell) : INFINITY;
      const float gy1 = (by1 < g.dy - 1) ? (g.miny + (float)(by1 + 1) * g.cell) - p.y : INFINITY;
      const float gz0 = (bz0 > 0) ? p.z - (g.minz + (float)bz0 * g.cell) : INFINITY;
      const float gz1 = (bz1 < g.dz - 1) ? (g.minz + (float)(bz1 + 1) * g.cell) - p.z : INFINITY;
      guard = fminf(fminf(fminf(gx0, gx1), fminf(gy0, gy1)), fminf(gz0, gz1)) * 0.9999f - 1e-5f;
    }
#if MM3D_NN_CORNERS
    // Which candidates can matter is known too: none that is farther from a lane than the best that lane has (or, before it
    // has one, than the distance transform's bound).  A candidate farther than the LARGEST such bound of the active lanes from
    // the bounding box of their positions improves nobody's result and is dropped while its tile is staged: the corners of a
    // later pass's box, whose radius is that very bound rounded up to cells.  (The bound must not be what this pass PROVES,
    // `guard`: that drops more, a fifth of a headline ICP launch's candidates, but the next pass skips this pass's box on the
    // ground that its lanes have seen ALL of it -- measured: wrong scores.)
    const float far = active ? fminf(best < INFINITY ? sqrtf(best) : INFINITY, reach_cap) : 0.0f;
    const float keep_r = wave_max_f(far) * 1.0001f + 1e-5f, keep_r2 = keep_r * keep_r;
    const float plx = wave_min_f(active ? p.x : INFINITY), phx = wave_max_f(active ? p.x : -INFINITY);
    const float ply = wave_min_f(active ? p.y : INFINITY), phy = wave_max_f(active ? p.y : -INFINITY);
    const float plz = wave_min_f(active ? p.z : INFINITY), phz = wave_max_f(active ? p.z : -INFINITY);
    const float edge = 1e-3f * g.cell + 1e-6f * (fabsf(g.miny) + fabsf(g.minz) + (float)(g.dy + g.dz) * g.cell);
#endif
    // A later pass only looks at what the earlier ones have not shown its lanes: every lane that is still active scanned ALL
    // the candidates of the previous pass's box (every lane scans every staged candidate), and by induction of every box
    // before it.  So the part of the new box that lies inside the previous one is skipped: a row of the new box whose (y, z)
    // lies in the old box's range contributes the span LEFT of the old box, [x0, ox0 - 1], and -- as one of the `n_inner` extra
    // spans behind the rows -- the span RIGHT of it, [ox1 + 1, x1]; either may be empty.  (Round 5.  The second passes,
    // 0.5 per wave with the tight boxes, staged their first pass's candidates again: a sixth of all staged candidates.)
    const bool skip_old = MM3D_NN_SHELL && have_old && ox0 <= x1 && ox1 >= x0;
    const int iy0 = max(y0, oy0), iy1 = min(y1, oy1), iz0 = max(z0, oz0), iz1 = min(z1, oz1);
    const int iny = iy1 - iy0 + 1, inz = iz1 - iz0 + 1;
    const int n_inner = (skip_old && nrows > 0 && iny > 0 && inz > 0) ? iny * inz : 0;
    const int nspans = nrows + n_inner;
    for (int r0 = 0; r0 < nspans; r0 += kRows) {
      MM3D_TICK(t_hdr);
      // Span headers, FOUR per lane (spans r0 + 4 lane .. + 3): a box of up to 256 rows costs one header round trip and fuller
      // tiles instead of a header, a prefix scan and a ragged last tile per 64 rows (a pass has ~150 - 250 rows, a row ~3 points).
      // Worth 2 % where the searches are short and many (64 maps x 50 k points), nothing on the headline, whose step is bound
      // by instruction issue.  Exclusive scan of the span lengths.
      int hb[kRowsPerLane], hl[kRowsPerLane];
      {
        const int r = r0 + kRowsPerLane * lane;
        // (y, z) of span r: a row of the new box (r < nrows) or an inner row's right-hand span
        bool second = r >= nrows;
        int t = second ? r - nrows : r;
        int wy = second ? iny : ny;                    // rows per z layer of the group
        int zq = t / max(wy, 1), yr = t - zq * wy;
#pragma unroll
        for (int u = 0; u < kRowsPerLane; ++u) {
          hb[u] = 0; hl[u] = 0;
          if (!second && r + u == nrows) { second = true; wy = iny; zq = 0; yr = 0; }   // this lane's spans straddle the two groups
          if (r + u < nspans) {
            const int y = (second ? iy0 : y0) + yr, z = (second ? iz0 : z0) + zq;
            int xa = x0, xb = x1;
            if (second) xa = max(x0, ox1 + 1);
            else if (n_inner && y >= iy0 && y <= iy1 && z >= iz0 && z <= iz1) xb = min(x1, ox0 - 1);
#if MM3D_NN_CORNERS
            {
              // the same bound at cell granularity: of this row only the cells that reach into the ball around the lanes'
              // position box are read at all (`edge`: what the row's points may lie outside their cells' nominal faces)
              const float ylo = g.miny + (float)y * g.cell, zlo = g.minz + (float)z * g.cell;
              const float ey = fmaxf(fmaxf(fmaxf(ylo - phy, ply - (ylo + g.cell)), 0.0f) - edge, 0.0f);
              const float ez = fmaxf(fmaxf(fmaxf(zlo - phz, plz - (zlo + g.cell)), 0.0f) - edge, 0.0f);
              const float rem = keep_r2 - ey * ey - ez * ez;
              if (rem < 0.0f) {
                xb = xa - 1;
              } else {
                const float w = sqrtf(rem) * 1.0001f + edge;
                xa = max(xa, cell_floor(plx - w, g.minx, g.inv));     // (cell_floor is monotone: exact in x)
                xb = min(xb, cell_floor(phx + w, g.minx, g.inv));
              }
            }
#endif
            if (xa <= xb) {
              const int row = (z * g.dy + y) * g.dx;
              hb[u] = g.cell_start[row + xa];
              hl[u] = g.cell_start[row + xb + 1];
            }
          }
          if (++yr == wy) { yr = 0; ++zq; }
        }
      }
      int mine = 0;
#pragma unroll
      for (int u = 0; u < kRowsPerLane; ++u) { hl[u] -= hb[u]; mine += hl[u]; }
      int incl = mine;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
      }
      const int total = __shfl(incl, kWave - 1, kWave);
      MM3D_STAT(2, 1);
      MM3D_STAT(3, total);
      MM3D_STAT(5, min(nspans - r0, kRows));
      wave_lds_sync();                 // previous chunk's readers are done
      {
        int off = incl - mine;
#pragma unroll
        for (int u = 0; u < kRowsPerLane; ++u) {
          s_off[wave][kRowsPerLane * lane + u] = off;
          s_beg[wave][kRowsPerLane * lane + u] = hb[u];
          off += hl[u];
        }
      }
      wave_lds_sync();
      MM3D_TOCK(32, t_hdr);
      // SPLIT 4: this wave's quarter of the chunk's candidates (a multiple of four, so the padding stays at the end)
      const int share = SPLIT == 4 ? ((total + 15) >> 4) << 2 : total;
      const int t_first = SPLIT == 4 ? min(total, wave * share) : 0;
      const int t_last = SPLIT == 4 ? min(total, t_first + share) : total;
      // Tiles of kTile candidates: slot -> (row by binary search over the chunk's offsets) -> sorted target index
      // (all of a lane's gathers are issued before the first LDS store: one memory round trip per tile).
      constexpr int kPer = kTile / kWave;
      float4 stage[kPer];
      auto fetch_tile = [&](int t0, int cnt) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
          const int s = lane + u * kWave;
          const int slot = t0 + (s < cnt ? s : 0);
          int lo = 0;
#pragma unroll
          for (int step = kRows / 2; step > 0; step >>= 1)
            if (s_off[wave][lo + step] <= slot) lo += step;   // offsets are non-decreasing; empty rows collapse
          stage[u] = g.pts[s_beg[wave][lo] + (slot - s_off[wave][lo])];
        }
      };
      int t0 = t_first, cnt = min(kTile, t_last - t0);
      if (kPrefetch && cnt > 0) fetch_tile(t0, cnt);
      while (cnt > 0) {
        MM3D_TICK(t_stage);
        if (!kPrefetch) fetch_tile(t0, cnt);
#if MM3D_NN_CORNERS
        {
          int kept = 0;                                  // wave-uniform
#pragma unroll
          for (int u = 0; u < kPer; ++u) {
            const int s = lane + u * kWave;
            const float ex = fmaxf(fmaxf(plx - stage[u].x, stage[u].x - phx), 0.0f), ey = fmaxf(fmaxf(ply - stage[u].y, stage[u].y - phy), 0.0f);
            const float ez = fmaxf(fmaxf(plz - stage[u].z, stage[u].z - phz), 0.0f);
            const bool keep = s < cnt && ex * ex + ey * ey + ez * ez <= keep_r2;
            const unsigned long long m = ballot(keep);
            const int d = kept + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (keep) {
              s_cx[wave][d] = stage[u].x; s_cy[wave][d] = stage[u].y; s_cz[wave][d] = stage[u].z;
              s_cw[wave][d] = __float_as_uint(stage[u].w);
            }
            kept += __popcll(m);
          }
          MM3D_STAT(38, cnt - kept);
          cnt = kept;                                    // (the tile's next slot range was fixed above: t0n, cntn)
        }
#else
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
          const int s = lane + u * kWave;
          if (s < cnt) {
            s_cx[wave][s] = stage[u].x; s_cy[wave][s] = stage[u].y; s_cz[wave][s] = stage[u].z;
            s_cw[wave][s] = __float_as_uint(stage[u].w);
          }
        }
#endif
        // pad to a multiple of four with points at infinity (distance +inf never wins)
        if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) {
          s_cx[wave][cnt + lane] = INFINITY; s_cy[wave][cnt + lane] = INFINITY; s_cz[wave][cnt + lane] = INFINITY;
          s_cw[wave][cnt + lane] = 0x7fffffffu;
        }
        wave_lds_sync();
        const int t0n = t0 + kTile, cntn = min(kTile, t_last - t0n);
        if (kPrefetch && cntn > 0) fetch_tile(t0n, cntn);
        MM3D_TOCK(33, t_stage);
        MM3D_TICK(t_scan);
        if (active) {
          // two candidates per packed instruction; the sums are dist2()'s: ((dx*dx + dy*dy) + dz*dz)
          typedef float f2 __attribute__((ext_vector_type(2)));
          const f2 px2 = {p.x, p.x}, py2 = {p.y, p.y}, pz2 = {p.z, p.z};
          auto d2_pair = [&](float xa, float xb, float ya, float yb, float za, float zb) {
            const f2 dx = px2 - f2{xa, xb}, dy = py2 - f2{ya, yb}, dz = pz2 - f2{za, zb};
            f2 r = dx * dx;
            r += dy * dy;
            r += dz * dz;
            return r;
          };
          // (not unrolled: inside this tile loop the optimizer declines `#pragma unroll 2`, and two groups written out by
          // hand measured the same)
          for (int k = 0; k < cnt; k += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(&s_cx[wave][k]);
            const float4 Y = *reinterpret_cast<const float4 *>(&s_cy[wave][k]);
            const float4 Z = *reinterpret_cast<const float4 *>(&s_cz[wave][k]);
            const f2 da = d2_pair(X.x, X.y, Y.x, Y.y, Z.x, Z.y), db = d2_pair(X.z, X.w, Y.z, Y.w, Z.z, Z.w);
            if (MODE == 1) {
              // transformScore only needs the distance
              bestd = fminf(fminf(bestd, fminf(da.x, da.y)), fminf(db.x, db.y));
            } else {
              // The (distance, index) key is only formed where it can matter: if the nearest of these four candidates is
              // farther than what EVERY active lane already holds, no key of the group can win or tie (d2 >= 0: its bits
              // order like its value; the initial key holds max_d2).  Candidates arrive row by row, so a
              // wave's lanes stop improving together once the rows near their patch are behind them: about half of the
              // groups take this exit, and a group that does costs 3 instead of 22 instructions on top of the distances
              // (round 4: the step is bound by VALU instructions).
              const float m4 = fminf(fminf(da.x, da.y), fminf(db.x, db.y));
              if (ballot(__float_as_uint(m4) <= (unsigned)(bkey >> 32))) {
                const uint4 W = *reinterpret_cast<const uint4 *>(&s_cw[wave][k]);
                const unsigned long long k0 = ((unsigned long long)__float_as_uint(da.x) << 32) | W.x;
                const unsigned long long k1 = ((unsigned long long)__float_as_uint(da.y) << 32) | W.y;
                const unsigned long long k2 = ((unsigned long long)__float_as_uint(db.x) << 32) | W.z;
                const unsigned long long k3 = ((unsigned long long)__float_as_uint(db.y) << 32) | W.w;
                const unsigned long long a = k0 < k1 ? k0 : k1, b2 = k2 < k3 ? k2 : k3;
                const unsigned long long m = a < b2 ? a : b2;
                bkey = m < bkey ? m : bkey;
              }
            }
          }
        }
        wave_lds_sync();                 // the tile's readers are done before the next one is stored
        MM3D_TOCK(34, t_scan);
        t0 = t0n; cnt = cntn;
      }
    }
    if (SPLIT == 4) {      // the four quarters' minima (every wave then goes on with the same state)
      s_merge[wave][lane] = MODE == 1 ? (unsigned long long)__float_as_uint(bestd) : bkey;   // d2 >= 0: bits order like values
      __syncthreads();
      const unsigned long long m0 = s_merge[0][lane], m1 = s_merge[1 % SPLIT][lane], m2 = s_merge[2 % SPLIT][lane],
                               m3 = s_merge[3 % SPLIT][lane];
      const unsigned long long ma = m0 < m1 ? m0 : m1, mb = m2 < m3 ? m2 : m3, m = ma < mb ? ma : mb;
      __syncthreads();
      if (MODE == 1) bestd = __uint_as_float((unsigned)m);
      else bkey = m;
    }
    if (nrows > 0) { have_old = true; ox0 = x0; ox1 = x1; oy0 = y0; oy1 = y1; oz0 = z0; oz1 = z1; }
    // what the scanned box proves: every target point closer than `guard` to this lane has been seen
    best = MODE == 1 ? bestd : (((unsigned)bkey == 0xffffffffu) ? INFINITY : __uint_as_float((unsigned)(bkey >> 32)));
    if (active) {
      if (guard >= rmax || best <= guard * guard) {
        active = false;
      } else {
        const float reach = best < INFINITY ? fminf(sqrtf(best), reach_cap) : reach_cap;
        const int want = (int)ceilf(reach * g.inv * 1.001f + 0.01f);   // guard >= want*cell*0.9999 - 1e-5 >= reach
        // (at least one ring more than this lane had: best > guard^2 and guard >= need cells already make `want` that large;
        // with the common radius of rounds 1 - 4 it was E + 1, the wave's largest plus one)
        need = min(max(want, (MM3D_NN_TIGHT_BOX ? need : E) + 1), max_ring + pass + 1);
      }
    }
#ifdef MM3D_NN_STATS
    if (MM3D_NN_STATS == 1 && lane == 0) {     // per ring size: passes, their ticks, active lanes
      const int e = E < 7 ? E : 7;
      atomicAdd(&g_nn_stats[40 + e], 1ull);
      atomicAdd(&g_nn_stats[48 + e], (unsigned long long)(wall_clock64() - t_pass));
    }
#endif
  }

#ifdef MM3D_NN_STATS
  if (it.y > 0 && lane == 0) {
    const unsigned long long dt = (unsigned long long)(wall_clock64() - t_begin);   // 100 MHz ticks
    atomicMax(&g_nn_stats[6], dt);
    atomicAdd(&g_nn_stats[7], dt);
    for (int k_ = 0; k_ < 3; ++k_) atomicAdd(&g_nn_stats[32 + k_], (unsigned long long)stat_ticks[k_]);
    int b = 0;
    while ((dt >> b) > 1 && b < 30) ++b;
    atomicAdd(&g_nn_stats[8 + b], 1ull);
  }
#endif
  if (SPLIT == 4 && wave != 0) return;     // the four waves hold the same result
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
  if (valid && best <= max_d2) {   // false for INFINITY / NaN
    if (MODE == 0) {
      const float4 bq = tgt_ref[(unsigned)(bkey & 0xffffffffull)];
      const float bqx = bq.x, bqy = bq.y, bqz = bq.z;
      acc[0] = p.x; acc[1] = p.y; acc[2] = p.z;
      acc[3] = bqx; acc[4] = bqy; acc[5] = bqz;
      acc[6] = (double)bqx * p.x; acc[7] = (double)bqx * p.y; acc[8] = (double)bqx * p.z;
      acc[9] = (double)bqy * p.x; acc[10] = (double)bqy * p.y; acc[11] = (double)bqy * p.z;
      acc[12] = (double)bqz * p.x; acc[13] = (double)bqz * p.y; acc[14] = (double)bqz * p.z;
    }
    acc[15] = best;
    acc[16] = 1.0;
  }
  // A wave none of whose points found a neighbour in range adds seventeen zeros: it writes them without the seventeen
  // reductions (each six shuffle steps on a double).  With the headline's initial poses that is a good part of the waves.
  const bool any_corr = ballot(valid && best <= max_d2) != 0ull;       // wave-uniform
  if (SPLIT == 4) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) {
      const double v = ((MODE == 0 || k >= 15) && any_corr) ? wave_sum(acc[k]) : 0.0;
      if (lane == 0) partials[(size_t)bid * kAcc + k] = v;
    }
    return;
  }
#pragma unroll
  for (int k = (MODE == 0 ? 0 : 15); k < kAcc; ++k) {
    const double v = any_corr ? wave_sum(acc[k]) : 0.0;
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

// one block: reduce partials, Umeyama, accumulate, convergence (DefaultConvergenceCriteria)
// Partial sums of "block" b as the four-items-per-block kernel writes them.  The one-item-per-block kernel leaves
// one partial per item; adding four neighbours here, in the order that kernel's last step does, gives the same
// bits -- so which variant ran (a choice that depends on what else was ready at the time) never shows in a result.
__device__ __forceinline__ double nn_block_partial(const double *__restrict__ p, int b, int k, int split, int n_items)
{
  if (!split) return p[(size_t)b * kAcc + k];
  const int i = b * 4;
  double v = p[(size_t)i * kAcc + k];
  v += (i + 1 < n_items) ? p[(size_t)(i + 1) * kAcc + k] : 0.0;
  v += (i + 2 < n_items) ? p[(size_t)(i + 2) * kAcc + k] : 0.0;
  v += (i + 3 < n_items) ? p[(size_t)(i + 3) * kAcc + k] : 0.0;
  return v;
}

__global__ void __launch_bounds__(256) k_icp_finalize(const NnJob *__restrict__ jobs)
{
  __shared__ double red[4][kAcc];
  __shared__ double tot[kAcc];
  const double *__restrict__ partials = jobs[blockIdx.x].partials;
  const int split = jobs[blockIdx.x].split, n_items = jobs[blockIdx.x].n_items;
  const int nblocks = (n_items + 3) >> 2;              // in units of four items, whichever kernel wrote them
  IcpState *st = jobs[blockIdx.x].st;
  if (st->done) return;
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] += nn_block_partial(partials, b, k, split, n_items);
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
  svd3_shared(sigma, U, S, V);
  double Sd[3] = {1.0, 1.0, 1.0};
  if (det3_shared(sigma) < 0) Sd[2] = -1.0;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(fabs(S[i]) <= fabs(S[0]) * 1e-5)) ++rank;
  if (rank == 2) Sd[2] = (det3_shared(U) * det3_shared(V) > 0) ? 1.0 : -1.0;
  float Ti[16];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += U[r * 3 + k] * Sd[k] * V[c * 3 + k];
      Ti[c * 4 + r] = (float)a;
    }
  for (int r = 0; r < 3; ++r) {
    // t = dst_mean - R * src_mean (with the float R, like Eigen's float instantiation)
    const double a = mq[r] - ((double)Ti[0 * 4 + r] * mp[0] + (double)Ti[1 * 4 + r] * mp[1] + (double)Ti[2 * 4 + r] * mp[2]);
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

__global__ void __launch_bounds__(256) k_score_finalize(const NnJob *__restrict__ jobs)
{
  __shared__ double red[4][2];
  const double *__restrict__ partials = jobs[blockIdx.x].partials;
  const int split = jobs[blockIdx.x].split, n_items = jobs[blockIdx.x].n_items;
  const int nblocks = (n_items + 3) >> 2;
  double *out = jobs[blockIdx.x].out;
  IcpState *st = jobs[blockIdx.x].st;
  if (st && (!st->done || st->scored)) return;
  double s = 0.0, n = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    s += nn_block_partial(partials, b, 15, split, n_items);
    n += nn_block_partial(partials, b, 16, split, n_items);
  }
  s = wave_sum(s); n = wave_sum(n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s; red[wave][1] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    out[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    if (st) st->scored = 1;
  }
}

// the source runs in the cloud's Hilbert order, one compact work item (<= 64 points) per wave (grid.hip)
static const float4 *morton_source(Context *c, const mm3d_cloud *src, int &n)
{
  cloud_hilbert(c, src);
  n = (int)src->n_finite;
  return src->hil_pts.get();
}

static float nn_cell_for(double radius)
{
  float cell = (float)(radius * 0.25);
  if (!(cell > 1e-3f)) cell = 0.25f;
  return cell;
}

// One work item per block (k_nn_wave's SPLIT 4) when the source has too few items to keep the chip busy with one
// wave each: 256 CUs x 4 SIMDs take 1024 waves before any two share a SIMD.  Measured on MI355X, pairs/s with
// the split off / on: 16 x 100 k points (1.3 k items) 1765 / 1823, 4 x 200 k (2.5 k items) 245 / 257,
// 64 x 50 k (0.6 k items) 4135 / 4925, 16 x 500 k (7.8 k items) 706 / 669.
static bool nn_split_items(int n_items) { return n_items <= 4096; }

// launches k_nn_wave<MODE> over `count` jobs (device array), picking the one-item-per-block variant for small sources
template <int MODE>
static void launch_nn(Context *c, const char *name, double bytes, const NnJob *jobs_dev, int count, unsigned grid_x, bool split, float max_d2,
                      float rmax)
{
  if (split)
    MM3D_LAUNCH(c, name, bytes, (k_nn_wave<MODE, 4>), dim3(grid_x, count), dim3(256), 0, jobs_dev, max_d2, rmax);
  else
    MM3D_LAUNCH(c, name, bytes, (k_nn_wave<MODE, 1>), dim3(grid_x, count), dim3(256), 0, jobs_dev, max_d2, rmax);
}

// ICP from a guess and, if wanted, transformScore of the result -- the tail of every pair estimate -- for a BATCH
// of pairs in lockstep: one launch per step serves every pair of the batch (blockIdx.y = the pair), and the batch
// shares ONE host synchronisation per chunk of iterations.  A guess may already live on the device (SAC-IA's
// winning hypothesis), the score kernel reads the transform straight out of the ICP state, and the states and
// scores come back in one copy.  The score is launched speculatively after each chunk of iterations; a pair's
// score is only kept once its ICP has finished (it nearly always has: the reference's epsilon is loose).
void icp_score_batch(Context *c, IcpScoreJob *jobs, int n_jobs, bool run_icp, double max_corr_dist, int max_iterations, double eps,
                     bool want_score, double score_max_distance)
{
  static_assert(offsetof(IcpState, T) == 0, "the score kernel reads T at the head of the state");
  c->last_icp_iterations = 0;
  c->last_icp_converged = 0;
  const double score_radius = std::sqrt(score_max_distance > 0 ? score_max_distance : 0.0);
  // ICP search parameters
  const double max_dist_sqr = max_corr_dist * max_corr_dist;
  // (double)d2 > max_dist_sqr rejects: accept d2 <= largest float not above max_dist_sqr
  float max_d2 = (float)max_dist_sqr;
  if ((double)max_d2 > max_dist_sqr) max_d2 = std::nextafterf(max_d2, -INFINITY);
  const float rmax = (float)(max_corr_dist * 1.0001 + 1e-5);
  // score search parameters: max_range_ is compared with the SQUARED distance (PCL quirk), so the
  // search radius is sqrt(max_distance)
  float s_max_d2 = (float)score_max_distance;
  if ((double)s_max_d2 > score_max_distance) s_max_d2 = std::nextafterf(s_max_d2, -INFINITY);
  const float s_rmax = (float)(score_radius * 1.0001 + 1e-5);

  struct Live { int job; const float4 *sp; int ns, n_items; const Grid *tg, *sg; int max_ring, s_ring; };
  std::vector<Live> live;
  for (int j = 0; j < n_jobs; ++j) {
    IcpScoreJob &J = jobs[j];
    J.out.iterations = 0; J.out.converged = 0; J.out.n_corr = 0; J.out.score = DBL_MAX;
    int ns = 0;
    const float4 *sp = (J.src->n && J.tgt->n) ? morton_source(c, J.src, ns) : nullptr;
    const Grid *tg = (ns && run_icp) ? &cloud_grid(c, J.tgt, nn_cell_for(max_corr_dist)) : nullptr;
    const Grid *sg = (ns && want_score) ? &cloud_grid(c, J.tgt, nn_cell_for(score_radius)) : nullptr;
    if (ns == 0 || (tg && tg->n == 0) || (sg && sg->n == 0) || (!tg && !sg)) {
      // nothing to search: Identity * guess, and the score of an empty search
      if (J.guess_dev) {
        float *hT = (float *)c->pin(256);
        MM3D_HIP(hipMemcpyAsync(hT, J.guess_dev, 64, hipMemcpyDeviceToHost, c->stream));
        c->sync();
        memcpy(J.out.T, hT, sizeof(J.out.T));
      } else {
        memcpy(J.out.T, J.guess_host, sizeof(J.out.T));
      }
      continue;
    }
    Live L{j, sp, ns, J.src->n_wave_items, tg, sg, 0, 0};
    L.max_ring = tg ? (int)std::ceil(rmax / tg->cell) + 1 : 0;
    if (tg) grid_ensure_dt(c, *tg, L.max_ring);
    L.s_ring = sg ? (int)std::ceil(s_rmax / sg->cell) + 1 : 0;
    if (sg) grid_ensure_dt(c, *sg, L.s_ring);
    live.push_back(L);
  }
  const int B = (int)live.size();
  if (B == 0) return;

  // one work item per block while the whole batch has too few items to fill the chip with one wave each
  // (the finalize kernels add the partials up in one fixed order, so the choice never shows in a result)
  int total_items = 0;
  for (const Live &L : live) total_items += L.n_items;
  const bool split = nn_split_items(total_items);
  size_t part_total = 0;
  unsigned grid_x = 0;
  double icp_bytes = 0.0, score_bytes = 0.0;
  std::vector<unsigned> nb(B);
  for (int b = 0; b < B; ++b) {
    nb[b] = split ? (unsigned)live[b].n_items : div_up(live[b].n_items, 4);
    part_total += (size_t)nb[b] * kAcc;
    grid_x = std::max(grid_x, nb[b]);
    icp_bytes += live[b].ns * 12.0;
    score_bytes += live[b].ns * 12.0 + (live[b].sg ? live[b].sg->n * 12.0 : 0.0);
  }
  DevBuf<double> partials(c, part_total), s_partials(c, want_score ? part_total : 1);
  DevBuf<double> out(c, (size_t)2 * B);
  DevBuf<IcpState> st(c, B);
  DevBuf<NnJob> d_jobs(c, (size_t)2 * B);                 // [0, B): ICP, [B, 2B): score

  // host images, in the pinned arena: states | ICP jobs | score jobs | scores back
  const size_t st_bytes = sizeof(IcpState) * B, job_bytes = sizeof(NnJob) * 2 * B, out_bytes = 16 * (size_t)B;
  char *pinned = (char *)c->pin(st_bytes + job_bytes + out_bytes + 64);
  IcpState *hp = (IcpState *)pinned;
  NnJob *hj = (NnJob *)(pinned + ((st_bytes + 15) & ~(size_t)15));
  double *ho = (double *)((char *)hj + job_bytes);
  size_t off = 0;
  for (int b = 0; b < B; ++b) {
    const Live &L = live[b];
    const IcpScoreJob &J = jobs[L.job];
    IcpState h;
    memset(&h, 0, sizeof(h));
    if (!J.guess_dev) memcpy(h.T, J.guess_host, sizeof(h.T));
    h.prev_mse = DBL_MAX;
    h.rot_thresh = 1.0 - eps;
    h.trans_thresh = eps;
    h.max_iter = max_iterations;
    h.done = run_icp ? 0 : 1;
    hp[b] = h;
    NnJob q;
    memset(&q, 0, sizeof(q));
    q.src = L.sp;
    q.items = (const int2 *)J.src->wave_items.get();
    q.n_items = L.n_items;
    q.nblocks = (int)nb[b];
    q.split = split ? 1 : 0;
    q.tgt_ref = (const float4 *)J.tgt->pts.get();
    q.st = st.get() + b;
    q.Tc = nullptr;
    q.out = out.get() + 2 * b;
    if (L.tg) { q.g = L.tg->view(); q.max_ring = L.max_ring; }
    q.partials = partials.get() + off;
    hj[b] = q;
    if (L.sg) { q.g = L.sg->view(); q.max_ring = L.s_ring; }
    q.partials = s_partials.get() + (want_score ? off : 0);
    hj[B + b] = q;
    off += (size_t)nb[b] * kAcc;
  }
  MM3D_HIP(hipMemcpyAsync(st.get(), hp, st_bytes, hipMemcpyHostToDevice, c->stream));
  MM3D_HIP(hipMemcpyAsync(d_jobs.get(), hj, job_bytes, hipMemcpyHostToDevice, c->stream));
  for (int b = 0; b < B; ++b)
    if (jobs[live[b].job].guess_dev)
      MM3D_HIP(hipMemcpyAsync(st.get() + b, jobs[live[b].job].guess_dev, 64, hipMemcpyDeviceToDevice, c->stream));
  // iterations launched between two looks at the `done` flags: with the reference's loose epsilon 86 % of the pairs
  // converge in one iteration and 95 % in two.  A launch after `done` does nothing, but it is not free on a GPU
  // that runs sixteen streams: its blocks still queue for 20 KB of LDS and 128 registers behind the other streams'
  // kernels before they can find that out.  So the first look comes after ONE iteration (a batch of one or two
  // pairs is then usually finished), later ones after two.
  static const int first_chunk_knob = [] { const char *e = getenv("MM3D_ICP_FIRST_CHUNK"); return e ? atoi(e) : 0; }();   // A/B knob (1 or 2; 0: by batch size)
  const int min_chunk = first_chunk_knob > 0 ? std::min(first_chunk_knob, 2) : (B <= 2 ? 1 : 2);
  for (int round = 0;; ++round) {
    const int chunk = round == 0 ? min_chunk : 2;
    if (run_icp) {
      for (int k = 0; k < chunk; ++k) {
        launch_nn<0>(c, "icp_corr_reduce", icp_bytes, d_jobs.get(), B, grid_x, split, max_d2, rmax);
        MM3D_LAUNCH(c, "icp_finalize", part_total * 8.0, k_icp_finalize, dim3(B), dim3(256), 0, (const NnJob *)d_jobs.get());
      }
    }
    if (want_score) {
      launch_nn<1>(c, "score_nn_reduce", score_bytes, d_jobs.get() + B, B, grid_x, split, s_max_d2, s_rmax);
      MM3D_LAUNCH(c, "score_finalize", 0, k_score_finalize, dim3(B), dim3(256), 0, (const NnJob *)(d_jobs.get() + B));
      MM3D_HIP(hipMemcpyAsync(ho, out.get(), out_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    MM3D_HIP(hipMemcpyAsync(hp, st.get(), st_bytes, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    bool all_done = true;
    for (int b = 0; b < B; ++b) {
      IcpScoreJob &J = jobs[live[b].job];
      if (hp[b].done && !J.closed) {
        // first chunk after which this pair is finished: its state and score are final
        memcpy(J.out.T, hp[b].T, sizeof(J.out.T));
        J.out.iterations = hp[b].iters;
        J.out.converged = hp[b].converged;
        J.out.n_corr = hp[b].n_corr;
        if (want_score) J.out.score = ho[2 * b + 1] > 0.0 ? ho[2 * b] / ho[2 * b + 1] : DBL_MAX;
        J.closed = true;
      }
      if (!hp[b].done) all_done = false;
    }
    if (all_done) break;
  }
  const IcpScoreJob &last = jobs[live[B - 1].job];
  c->last_icp_iterations = last.out.iterations;
  c->last_icp_converged = last.out.converged;
}

PairTail icp_score(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float *guess_dev, const float guess_host[16],
                   bool run_icp, double max_corr_dist, int max_iterations, double eps, bool want_score, double score_max_distance)
{
  IcpScoreJob J;
  J.src = src; J.tgt = tgt; J.guess_dev = guess_dev;
  if (guess_host) memcpy(J.guess_host, guess_host, sizeof(J.guess_host));
  icp_score_batch(c, &J, 1, run_icp, max_corr_dist, max_iterations, eps, want_score, score_max_distance);
  return J.out;
}

IcpResult icp(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float guess[16],
              double max_corr_dist, int max_iterations, double eps)
{
  const PairTail t = icp_score(c, src, tgt, nullptr, guess, true, max_corr_dist, max_iterations, eps, false, 0.0);
  IcpResult res;
  memcpy(res.T, t.T, sizeof(res.T));
  res.iterations = t.iterations;
  res.converged = t.converged;
  return res;
}

double transform_score(Context *c, const mm3d_cloud *src, const mm3d_cloud *tgt, const float T[16], double max_distance)
{
  if (src->n == 0 || tgt->n == 0) return DBL_MAX;
  // max_range_ is compared with the SQUARED distance (PCL quirk): search radius sqrt(max_distance)
  const double radius = std::sqrt(max_distance > 0 ? max_distance : 0.0);
  const Grid &tg = cloud_grid(c, tgt, nn_cell_for(radius));
  int ns = 0;
  const float4 *sp = morton_source(c, src, ns);
  if (ns == 0 || tg.n == 0) return DBL_MAX;
  float max_d2 = (float)max_distance;
  if ((double)max_d2 > max_distance) max_d2 = std::nextafterf(max_d2, -INFINITY);
  const float rmax = (float)(radius * 1.0001 + 1e-5);
  const int max_ring = (int)std::ceil(rmax / tg.cell) + 1;
  grid_ensure_dt(c, tg, max_ring);
  const int n_items = src->n_wave_items;
  const bool split = nn_split_items(n_items);
  const unsigned nblocks = split ? (unsigned)n_items : div_up(n_items, 4);
  DevBuf<double> partials(c, (size_t)nblocks * kAcc);
  DevBuf<float> dT(c, 16);
  DevBuf<double> out(c, 2);
  DevBuf<NnJob> d_job(c, 1);
  char *pinned = (char *)c->pin(512 + sizeof(NnJob));
  float *hT = (float *)pinned;
  double *ho = (double *)(pinned + 128);
  NnJob *hj = (NnJob *)(pinned + 256);
  memcpy(hT, T, 64);
  NnJob q;
  memset(&q, 0, sizeof(q));
  q.src = sp;
  q.items = (const int2 *)src->wave_items.get();
  q.n_items = n_items;
  q.nblocks = (int)nblocks;
  q.split = split ? 1 : 0;
  q.g = tg.view();
  q.tgt_ref = (const float4 *)tgt->pts.get();
  q.Tc = dT.get();
  q.partials = partials.get();
  q.out = out.get();
  q.max_ring = max_ring;
  *hj = q;
  MM3D_HIP(hipMemcpyAsync(dT.get(), hT, 64, hipMemcpyHostToDevice, c->stream));
  MM3D_HIP(hipMemcpyAsync(d_job.get(), hj, sizeof(NnJob), hipMemcpyHostToDevice, c->stream));
  launch_nn<1>(c, "score_nn_reduce", ns * 12.0 + tg.n * 12.0, d_job.get(), 1, nblocks, split, max_d2, rmax);
  MM3D_LAUNCH(c, "score_finalize", 0, k_score_finalize, dim3(1), dim3(256), 0, (const NnJob *)d_job.get());
  MM3D_HIP(hipMemcpyAsync(ho, out.get(), 16, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  return ho[1] > 0.0 ? ho[0] / ho[1] : DBL_MAX;
}

#ifdef MM3D_NN_STATS
extern "C" void mm3d_debug_nn_stats(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_stats), sizeof(unsigned long long) * 64);
  if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stats), z, sizeof(z)); }
}
#endif

// Everything icp_score() would build lazily on its first use of these clouds: afterwards pair
// estimates only READ the clouds' caches, so one map can serve pairs on several contexts at once.
void prepare_pair_search(Context *c, const mm3d_cloud *points, double max_corr_dist, double score_max_distance)
{
  if (points->n == 0) return;
  int ns = 0;
  (void)morton_source(c, points, ns);
  const double radii[2] = {max_corr_dist, std::sqrt(score_max_distance > 0 ? score_max_distance : 0.0)};
  for (double radius : radii) {
    const Grid &g = cloud_grid(c, points, nn_cell_for(radius));
    if (g.n == 0) continue;
    const float rmax = (float)(radius * 1.0001 + 1e-5);
    grid_ensure_dt(c, g, (int)std::ceil(rmax / g.cell) + 1);
  }
}

}  // namespace mm3d
