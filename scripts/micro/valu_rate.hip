// VALU issue peak of the MI355X as THIS library's kernels see it: independent f32 instruction streams at 1 / 2 / 4 / 8
// waves per SIMD on all 256 CUs.  bench.py's VALU_PEAK_WINSTR_S (the denominator of every `valu_frac`) is what this prints
// for v_fma_f32 at full occupancy; the guide's table says 2 cycles per wave64 instruction on a SIMD-32
// (/opt/skills/guides/MI355X_MICROARCH.md "Per-instruction cycle constants").
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Output per (instruction, waves per SIMD): wave-instructions per second over the chip (HIP events, best of 5 after two
// full-length warm-up launches: the clocks ramp for milliseconds), the same as nanoseconds per wave-instruction per SIMD,
// and the s_memtime ticks one wave's own instruction takes (the counter follows the shader clock, which the power
// management moves with the load: compare rates by the wall clock, not by ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kChains = 16;      // independent accumulators per lane: no instruction waits for the one before it

// MODE 0 v_fma_f32, 1 v_add_f32, 2 v_pk_fma_f32 (two f32 per lane and instruction), 3 v_fma_f64, 4 v_add_f32 with a DPP quad broadcast,
// 5 v_add_u32, 6 v_and_b32, 7 v_cndmask_b32 (vcc), 8 v_cmp_lt_f32 + v_cndmask_b32 (two instructions), 9 v_mul_f32, 10 v_max_f32,
// 11 v_mbcnt_lo_u32_b32, 12 v_cvt_f32_i32, 13 v_rcp_f32, 14 v_exp_f32, 15 v_pk_add_f32, 16 v_pk_mul_f32, 17 v_sqrt_f32, 18 v_mul_lo_u32,
// 19 v_lshlrev_b32, 20 v_fma_f32 + v_and_b32 alternating (do an fp32 and an integer stream share a SIMD's cycles?)
template <int MODE>
__global__ void __launch_bounds__(1024)
k_rate(int iters, float *out, long long *cyc)
{
  const int l = threadIdx.x;
  float f[kChains]; v2f p[kChains]; double d[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c) { f[c] = 1.0f + (float)(l + c) * 1e-3f; p[c] = v2f{f[c], f[c] + 1.0f}; d[c] = (double)f[c]; }
  const float a = 1.0000001f, b = 1e-7f;
  const v2f pa = {a, a}, pb = {b, b};
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(b));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pa), "v"(pb));
      if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"((double)a), "v"((double)b));
      if (MODE == 4) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(f[c]) : "v"(f[(c + 1) % kChains]));
      if (MODE == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[c]) : "v"(b));
      if (MODE == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[c]) : "v"(a));
      if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[c]) : "v"(a));
      if (MODE == 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(f[c]) : "v"(a), "v"(b) : "vcc");
      if (MODE == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(a));
      if (MODE == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[c]) : "v"(a));
      if (MODE == 11) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(f[c]) : "v"(a));
      if (MODE == 12) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(f[c]));
      if (MODE == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
      if (MODE == 14) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
      if (MODE == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pb));
      if (MODE == 16) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pa));
      if (MODE == 17) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
      if (MODE == 18) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[c]) : "v"(a));
      if (MODE == 19) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(f[c]));
      if (MODE == 20) { if (c & 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[c]) : "v"(a)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(a), "v"(b)); }
    }
  }
  const long long t1 = clock64();
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < kChains; ++c) s += f[c] + p[c].x + p[c].y + (float)d[c];
  if (s == 12345.678f) out[0] = s;               // keeps the chains alive
  if ((l & 63) == 0 && blockIdx.x == 0 && l == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static int run(const char *name, int cus)
{
  float *out; long long *cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {
    if (MODE >= 5 && wps != 1 && wps != 4) continue;                 // the long table: one wave alone and the occupancy the kernels run at
    // wps waves on each of a CU's four SIMDs: blocks of 4 * min(wps, 4) waves, one or two blocks per CU
    const int waves_per_block = 4 * (wps < 4 ? wps : 4), blocks_per_cu = wps <= 4 ? 1 : wps / 4;
    const dim3 grid(cus * blocks_per_cu), block(64 * waves_per_block);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_rate<MODE>, grid, block, 0, 0, iters, out, cyc);      // warm-up (clocks, code)
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    long long c = 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rate<MODE>, grid, block, 0, 0, iters, out, cyc);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) { best = ms; CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost)); }
    }
    const double winstr = (double)cus * blocks_per_cu * waves_per_block * (double)iters * kChains * (MODE == 8 ? 2 : 1);
    const double rate = winstr / (best * 1e-3);
    printf("%-22s %d waves/SIMD: %8.3f ms  %.3e wave-instr/s chip = %.3f ns per wave-instruction per SIMD (2 cycles at 2.4 GHz = 0.833 ns);"
           " one wave's own instruction: %.2f ticks\n",
           name, wps, best, rate, 1e9 * (double)cus * 4.0 / rate, (double)c / ((double)iters * kChains));
  }
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s: %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  if (run<0>("v_fma_f32", cus)) return 1;
  if (run<1>("v_add_f32", cus)) return 1;
  if (run<2>("v_pk_fma_f32", cus)) return 1;
  if (run<3>("v_fma_f64", cus)) return 1;
  if (run<4>("v_add_f32_dpp quad", cus)) return 1;
  if (run<5>("v_add_u32", cus)) return 1;
  if (run<6>("v_and_b32", cus)) return 1;
  if (run<7>("v_cndmask_b32", cus)) return 1;
  if (run<8>("v_cmp+v_cndmask (x2)", cus)) return 1;
  if (run<9>("v_mul_f32", cus)) return 1;
  if (run<10>("v_max_f32", cus)) return 1;
  if (run<11>("v_mbcnt_lo", cus)) return 1;
  if (run<12>("v_cvt_f32_i32", cus)) return 1;
  if (run<13>("v_rcp_f32", cus)) return 1;
  if (run<14>("v_exp_f32", cus)) return 1;
  if (run<15>("v_pk_add_f32", cus)) return 1;
  if (run<16>("v_pk_mul_f32", cus)) return 1;
  if (run<17>("v_sqrt_f32", cus)) return 1;
  if (run<18>("v_mul_lo_u32", cus)) return 1;
  if (run<19>("v_lshlrev_b32", cus)) return 1;
  if (run<20>("v_fma_f32 | v_and_b32", cus)) return 1;
  return 0;
}
