// micro-benchmark of the SAC-IA rand() replay (host only): ns keypoints, H hypotheses
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
struct float4 { float x, y, z, w; };
struct GlibcRand {
  uint32_t ring[31]; int f = 3, b = 0;
  GlibcRand() { seed(1); }
  void seed(unsigned s) { if (s == 0) s = 1; int32_t r[31]; r[0] = (int32_t)s;
    for (int i = 1; i < 31; ++i) { long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773; long word = 16807 * lo - 2836 * hi; if (word < 0) word += 2147483647; r[i] = (int32_t)word; }
    for (int i = 0; i < 31; ++i) ring[i] = (uint32_t)r[i]; f = 3; b = 0; for (int i = 0; i < 310; ++i) (void)next(); }
  int next() { ring[f] += ring[b]; uint32_t res = ring[f] >> 1; f = (f + 1) % 31; b = (b + 1) % 31; return (int)res; }
};
static void draws_ref(GlibcRand &rnd, const std::vector<float4> &skp, int ns, float msd, int H)
{
  auto gri = [&](int n) { return (int)(n * (rnd.next() / (2147483647 + 1.0))); };
  int sample[3];
  for (int it = 0; it < H; ++it) {
    int cnt = 0, without = 0; const int max_without = 3 * ns;
    while (cnt < 3) {
      const int si = gri(ns); bool valid = true;
      for (int i = 0; i < cnt; ++i) { const float4 &a = skp[si], &b = skp[sample[i]]; const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz); if (si == sample[i] || dist < msd) { valid = false; break; } }
      if (valid) { sample[cnt++] = si; without = 0; } else ++without;
      if (without >= max_without) { msd *= 0.5f; without = 0; }
    }
    for (int i = 0; i < 3; ++i) (void)gri(10);
  }
}
// exact equivalent of "sqrtf(d2) < msd" without the square root: the smallest float t with sqrtf(t) >= msd; then sqrtf(d2) < msd <=> d2 < t
static float sq_threshold(float msd)
{
  if (!(msd > 0.0f)) return 0.0f;                       // sqrtf(d2) < msd never holds for msd <= 0 (d2 >= 0)
  float t = msd * msd;
  while (sqrtf(t) >= msd) t = nextafterf(t, 0.0f);      // go below
  while (sqrtf(t) < msd) t = nextafterf(t, INFINITY);   // first one at or above
  return t;
}
static void draws_fast(GlibcRand &rnd, const std::vector<float4> &skp, int ns, float msd, int H)
{
  uint32_t *ring = rnd.ring; int f = rnd.f, b = rnd.b;
  auto next = [&]() { ring[f] += ring[b]; const uint32_t res = ring[f] >> 1; if (++f == 31) f = 0; if (++b == 31) b = 0; return (int)res; };
  auto gri = [&](int n) { return (int)(n * (next() * (1.0 / 2147483648.0))); };
  float thr = sq_threshold(msd);
  int sample[3];
  for (int it = 0; it < H; ++it) {
    int cnt = 0, without = 0; const int max_without = 3 * ns;
    while (cnt < 3) {
      const int si = gri(ns); bool valid = true;
      const float4 a = skp[si];
      for (int i = 0; i < cnt; ++i) { const float4 &b2 = skp[sample[i]]; const float dx = a.x - b2.x, dy = a.y - b2.y, dz = a.z - b2.z;
        const float d2 = dx * dx + dy * dy + dz * dz; if (si == sample[i] || d2 < thr) { valid = false; break; } }
      if (valid) { sample[cnt++] = si; without = 0; } else ++without;
      if (without >= max_without) { msd *= 0.5f; thr = sq_threshold(msd); without = 0; }
    }
    for (int i = 0; i < 3; ++i) (void)gri(10);
  }
  rnd.f = f; rnd.b = b;
}
int main()
{
  for (int ns : {2400, 15700}) {
    std::vector<float4> kp(ns); GlibcRand g; for (auto &p : kp) { p.x = (g.next() % 60000) * 1e-3f; p.y = (g.next() % 60000) * 1e-3f; p.z = (g.next() % 3000) * 1e-3f; }
    GlibcRand r; auto t0 = std::chrono::steady_clock::now(); const int reps = 2000;
    for (int k = 0; k < reps; ++k) draws_ref(r, kp, ns, 0.5f, 500);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("ns %d: %.1f us per pair (state %u)\n", ns, us, r.ring[0]);
    GlibcRand r2; t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) draws_fast(r2, kp, ns, 0.5f, 500);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("ns %d: %.1f us per pair fast (state %u) %s\n", ns, us, r2.ring[0], (r2.ring[0] == r.ring[0] && r2.f == r.f) ? "same state" : "DIFFERENT");
  }
}
