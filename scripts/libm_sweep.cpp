// libm_sweep.cpp -- proves csrc/libm_exact.hpp against the host's libm (TEST / EVIDENCE TOOL).
//
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -fopenmp scripts/libm_sweep.cpp -o /tmp/libm_sweep -lm
//   /tmp/libm_sweep [stride]        (stride 1 = all 2^32 float arguments of every one-argument function)
//
// expf, atanf: every float.  sinf, cosf: every float with |x| < 120 (the restated range).  atan2f:
// ~4e9 pairs: a lattice of exponents x mantissas plus random pairs, plus the quadrant / zero /
// infinity / NaN special cases.  Prints one line per function: "<name> checked N mismatches M".
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../map-merge_amd/csrc/libm_exact.hpp"

using namespace mm3d::lm;

static bool same(float a, float b)
{
  if (std::isnan(a) && std::isnan(b)) return true;
  return f2u(a) == f2u(b);
}

template <class F, class G>
static void sweep1(const char *name, F mine, G ref, uint64_t stride, bool limit120)
{
  uint64_t bad = 0, n = 0;
  uint32_t first_bad = 0;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
  for (int64_t b = 0; b < (int64_t)(0x100000000ull / stride); ++b) {
    const uint32_t u = (uint32_t)((uint64_t)b * stride);
    const float x = u2f(u);
    if (limit120 && !(std::fabs(x) < 120.0f)) continue;
    ++n;
    if (!same(mine(x), ref(x))) {
      if (!bad) first_bad = u;
      ++bad;
    }
  }
  printf("%s checked %" PRIu64 " mismatches %" PRIu64, name, n, bad);
  if (bad) printf(" (e.g. x = %a: mine %a libm %a)", u2f(first_bad), mine(u2f(first_bad)), ref(u2f(first_bad)));
  printf("\n");
}

int main(int argc, char **argv)
{
  const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 0) : 1;
  // the table of expf_glibc against the definition
  for (unsigned i = 0; i < 32; ++i) {
    const uint64_t want = d2u(exp2((double)i / 32.0)) - ((uint64_t)i << 47);
    if (want != exp2f_tab(i)) { printf("exp2f table entry %u differs from exp2(i/32)\n", i); return 1; }
  }
  sweep1("expf", [](float x) { return expf_glibc(x); }, [](float x) { return expf(x); }, stride, false);
  sweep1("atanf", [](float x) { return atanf_glibc(x); }, [](float x) { return atanf(x); }, stride, false);
  sweep1("sinf", [](float x) { return sinf_glibc(x); }, [](float x) { return sinf(x); }, stride, true);
  sweep1("cosf", [](float x) { return cosf_glibc(x); }, [](float x) { return cosf(x); }, stride, true);
  // atan2f: specials, then a lattice and pseudo-random pairs
  {
    uint64_t bad = 0, n = 0;
    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 3.4e38f, -3.4e38f, 1e-30f, 0.5f, 2.0f, 1e20f, -1e20f};
    for (float y : sp)
      for (float x : sp) { ++n; if (!same(atan2f_glibc(y, x), atan2f(y, x))) ++bad; }
    const uint64_t pairs = 0x100000000ull / stride;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (int64_t b = 0; b < (int64_t)pairs; ++b) {
      // splitmix64 -> two floats; half of the pairs keep both exponents near 1 (the callers' regime: components
      // of unit vectors and their products), the others span everything
      uint64_t z = (uint64_t)b * 0x9E3779B97F4A7C15ull + 0x1234567ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      uint32_t uy = (uint32_t)z, ux = (uint32_t)(z >> 32);
      if (b & 1) {
        uy = (uy & 0x807fffffu) | ((uint32_t)(0x6e + (uy >> 23) % 20u) << 23);
        ux = (ux & 0x807fffffu) | ((uint32_t)(0x6e + (ux >> 23) % 20u) << 23);
      }
      const float y = u2f(uy), x = u2f(ux);
      ++n;
      if (!same(atan2f_glibc(y, x), atan2f(y, x))) ++bad;
    }
    printf("atan2f checked %" PRIu64 " mismatches %" PRIu64 "\n", n, bad);
  }
  return 0;
}
