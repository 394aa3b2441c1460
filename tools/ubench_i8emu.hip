// Micro-benchmark (round 5, DESIGN 9 "what comes next"): could the FP64 GEMMs of the encoder / rollout kernels run on the INT8
// matrix pipe as sliced fixed-point products (activations are GRU states / tanh outputs in (-1, 1), weights are constants)?
// What that needs from the hardware, measured here on gfx950:
//   1. v_mfma_i32_16x16x64_i8 issue cost (one and two waves per SIMD);
//   2. whether FP64 VALU / conversion / integer instructions of the SAME wave issue under a running i8 MFMA (they do not under
//      v_mfma_f64_16x16x4_f64: profiles/r2_ubench_trans.txt);
//   3. the same with the MFMAs in one wave and the VALU work in ANOTHER wave of the SIMD;
//   4. issue cost of the instructions the slicing / recombination is made of.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_i8emu.hip -o /tmp/ubench_i8emu && /tmp/ubench_i8emu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));

// KIND: probed VALU instruction; NM: i8 MFMAs per iteration (0 = none); F64: use the FP64 MFMA instead (reference)
template <int KIND, int NM, bool F64, int PER>
__device__ __forceinline__ void body(int iters, double* out, unsigned long long* clk, bool do_mfma, bool do_valu) {
  v4i acc[8];
  v4d accd[8];
  for (int i = 0; i < 8; ++i) {
    acc[i] = v4i{0, 0, 0, 0};
    accd[i] = v4d{0, 0, 0, 0};
  }
  const v4i a = {(int)threadIdx.x * 0x01010101, 0x01020304, 0x7f807f80, (int)threadIdx.x};
  const v4i b = {0x01010101, (int)threadIdx.x * 0x00010203, 0x10203040, 0x7f7f7f7f};
  const double da = threadIdx.x * 1e-3, db = 1.0 + threadIdx.x * 1e-6;
  double f[16];  // 16 independent registers per kind, each touched ONCE per iteration (throughput, not latency)
  int g[16];
  long long h[16];
  for (int i = 0; i < 16; ++i) {
    f[i] = 1.0 + da + i;
    g[i] = threadIdx.x + i;
    h[i] = threadIdx.x * 77 + i;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (do_mfma && i < NM) {
        if (F64)
          accd[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, accd[i], 0, 0, 0);
        else
          acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
          const int j = (2 * i + r) & 15;
          if (KIND == 1) f[j] = __builtin_fma(f[j], db, da);
          if (KIND == 2) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(f[j]) : "v"(g[j]));
          if (KIND == 3) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(g[j]) : "v"(f[j]));
          if (KIND == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(f[j]));
          if (KIND == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(g[j]) : "v"(g[(j + 1) & 15]), "v"(0x07020500));
          if (KIND == 6) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(h[j]) : "v"(g[j]), "v"(256) : "vcc");
          if (KIND == 7) asm volatile("v_add_u32 %0, %0, %0" : "+v"(g[j]));
          if (KIND == 8) asm volatile("v_lshlrev_b64 %0, 8, %0" : "+v"(h[j]));
          if (KIND == 9) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(f[j]));
          if (KIND == 10) asm volatile("v_bfe_i32 %0, %0, 7, 8" : "+v"(g[j]));
          if (KIND == 11) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[j]));
          if (KIND == 12) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[j]) : "v"(db));
          if (KIND == 13) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f[j]) : "v"(g[j]));
          if (KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(g[j]) : "v"(g[(j + 1) & 15]));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + accd[i][0] + accd[i][1] + accd[i][2] + accd[i][3];
  for (int i = 0; i < 16; ++i) s += f[i] + g[i] + (double)h[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[threadIdx.x >> 6] = t1 - t0;
}

// MODE 0: every wave runs MFMAs + VALU; MODE 1: wave 2w runs the MFMAs, wave 2w + 1 the VALU work (needs 512 threads: waves w and
// w + 4 share SIMD w % 4 -- so split by (wave >> 2))
template <int KIND, int NM, bool F64, int PER, int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters, unsigned long long* clk) {
  if (MODE == 0) {
    body<KIND, NM, F64, PER>(iters, out, clk, true, true);
  } else {
    const int w = threadIdx.x >> 6;
    if (w < 4)
      body<KIND, NM, F64, PER>(iters, out, clk, true, false);
    else
      body<KIND, NM, F64, PER>(iters, out, clk, false, true);
  }
}

static double g_last_clk24;  // the launch's elapsed time per iteration in clocks of a 2.4 GHz shader clock (cross-check of the ticks)
template <int KIND, int NM, bool F64, int PER, int MODE>
void launch(int threads, double* ticks_out, int nw) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount, iters = 20000;
  double* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, (size_t)blocks * 512 * 8);
  (void)hipMalloc(&clk, 64);
  (void)hipMemset(clk, 0, 64);
  // 100 KB of dynamic LDS per block: exactly ONE block per CU (a second block on block 0's CU would double its waves per SIMD)
  (void)hipFuncSetAttribute((const void*)k<KIND, NM, F64, PER, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NM, F64, PER, MODE>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, iters, clk);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NM, F64, PER, MODE>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, iters, clk);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  g_last_clk24 = ms * 1e-3 * 2.4e9 / iters;
  unsigned long long c[8];
  (void)hipMemcpy(c, clk, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < nw; ++i) ticks_out[i] = (double)c[i] / iters;
  (void)hipFree(out);
  (void)hipFree(clk);
}

template <int KIND>
void probe(const char* name) {
  double a1[8], a2[8], m1[8], m2[8], s[8], sf[8];
  launch<KIND, 0, false, 2, 0>(256, a1, 1);   // 16 VALU alone, one wave per SIMD
  launch<KIND, 0, false, 2, 0>(512, a2, 1);   // ... two waves per SIMD
  const double a2_wall = g_last_clk24;
  launch<KIND, 8, false, 2, 0>(256, m1, 1);   // 8 i8 MFMAs + 16 VALU interleaved in one wave
  launch<KIND, 8, false, 2, 0>(512, m2, 1);   // ... two such waves per SIMD
  const double m2_wall = g_last_clk24;
  launch<KIND, 8, false, 2, 1>(512, s, 8);    // wave A: 8 i8 MFMAs per iteration; wave B (same SIMD): 16 VALU per iteration
  launch<KIND, 8, true, 2, 1>(512, sf, 8);    // the same with FP64 MFMAs in wave A
  printf("%-16s alone %6.1f (1 w) %6.1f (2 w) | +8 i8 MFMA same wave %6.1f (1 w) %6.1f (2 w) | split waves: mfma %6.1f valu %6.1f | split, f64 MFMA: mfma %6.1f valu %6.1f | wall @2.4 GHz: alone 2 w %6.1f, +MFMA 2 w %6.1f\n",
         name, a1[0], a2[0], m1[0], m2[0], s[0], s[4], sf[0], sf[4], a2_wall, m2_wall);
}


// ---- phases: every wave alternates a burst of NM i8 MFMAs (8 accumulators in rotation) and a burst of NV independent v_fma_f64 --
// what one gate tile of the sliced encoder looks like from the SIMD.  ANTI: waves 4 .. 7 of the workgroup (the SIMDs' second waves)
// start with the VALU burst.
template <int NM, int NV, bool ANTI>
__global__ __launch_bounds__(512) void k_phase(double* out, int iters, unsigned long long* clk) {
  v4i acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4i{0, 0, 0, 0};
  const v4i a = {(int)threadIdx.x * 0x01010101, 0x01020304, 0x7f807f80, (int)threadIdx.x};
  const v4i b = {0x01010101, (int)threadIdx.x * 0x00010203, 0x10203040, 0x7f7f7f7f};
  const double da = threadIdx.x * 1e-3, db = 1.0 + threadIdx.x * 1e-6;
  double f[16];
  for (int i = 0; i < 16; ++i) f[i] = 1.0 + da + i;
  const bool second = ANTI && (threadIdx.x >> 8);
  auto valu = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) f[i & 15] = __builtin_fma(f[i & 15], db, da);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i & 7], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  if (second) valu();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    mfmas();
    valu();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
template <int NM, int NV, bool ANTI>
double run_phase(int threads) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount, iters = 2000;
  double* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, (size_t)blocks * 512 * 8);
  (void)hipMalloc(&clk, 64);
  (void)hipFuncSetAttribute((const void*)k_phase<NM, NV, ANTI>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((k_phase<NM, NV, ANTI>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, iters, clk);
  hipLaunchKernelGGL((k_phase<NM, NV, ANTI>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long c[8];
  (void)hipMemcpy(c, clk, 64, hipMemcpyDeviceToHost);
  (void)hipFree(out);
  (void)hipFree(clk);
  return (double)c[0] / iters;
}

// ---- one wave per SIMD, MFMAs and independent FP64 FMAs interleaved in program order: 1 MFMA, PER FMAs, ... (what a wave that works
// on TWO gate tiles in anti-phase would issue: one tile's MFMAs beside the other's recombination / gate math)
template <int PER>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, unsigned long long* clk) {
  v4i acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4i{0, 0, 0, 0};
  const v4i a = {(int)threadIdx.x * 0x01010101, 0x01020304, 0x7f807f80, (int)threadIdx.x};
  const v4i b = {0x01010101, (int)threadIdx.x * 0x00010203, 0x10203040, 0x7f7f7f7f};
  const double da = threadIdx.x * 1e-3, db = 1.0 + threadIdx.x * 1e-6;
  double f[16];
  for (int i = 0; i < 16; ++i) f[i] = 1.0 + da + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      acc[m & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < PER; ++r) f[(m * PER + r) & 15] = __builtin_fma(f[(m * PER + r) & 15], db, da);
    }
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int PER>
double run_mix() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount, iters = 2000;
  double* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  (void)hipMalloc(&clk, 64);
  (void)hipFuncSetAttribute((const void*)k_mix<PER>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((k_mix<PER>), dim3(blocks), dim3(256), 100 * 1024, 0, out, iters, clk);
  hipLaunchKernelGGL((k_mix<PER>), dim3(blocks), dim3(256), 100 * 1024, 0, out, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long c = 0;
  (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  (void)hipFree(out);
  (void)hipFree(clk);
  return (double)c / iters / 32;
}

int main() {
  printf("one wave per SIMD, 1 i8 MFMA + PER independent v_fma_f64 interleaved: ticks per MFMA slot  PER=0 %.1f | 1 %.1f | 2 %.1f | 3 %.1f | 4 %.1f | 5 %.1f | 6 %.1f\n", run_mix<0>(), run_mix<1>(), run_mix<2>(), run_mix<3>(), run_mix<4>(), run_mix<5>(), run_mix<6>());

  printf("phases (34 i8 MFMAs, then 128 v_fma_f64), ticks per iteration and wave: one wave per SIMD %.0f | two waves in phase %.0f | two waves, the second starting with its VALU burst %.0f\n",
         run_phase<34, 128, false>(256), run_phase<34, 128, false>(512), run_phase<34, 128, true>(512));
  printf("phases (34 i8 MFMAs, then 64 v_fma_f64): one wave %.0f | two in phase %.0f | two in anti-phase %.0f\n", run_phase<34, 64, false>(256), run_phase<34, 64, false>(512),
         run_phase<34, 64, true>(512));

  double t[8];
  launch<0, 8, false, 0, 0>(256, t, 1);
  printf("8 x v_mfma_i32_16x16x64_i8, one wave per SIMD: %.1f ticks per iteration (%.1f each)\n", t[0], t[0] / 8);
  launch<0, 8, false, 0, 0>(512, t, 1);
  printf("8 x v_mfma_i32_16x16x64_i8, two waves per SIMD: %.1f ticks per iteration (%.1f per MFMA and SIMD)\n", t[0], t[0] / 16);
  launch<0, 4, false, 0, 0>(256, t, 1);
  printf("4 x v_mfma_i32_16x16x64_i8 (4 accumulators), one wave per SIMD: %.1f ticks per iteration (%.1f each)\n", t[0], t[0] / 4);
  launch<0, 2, false, 0, 0>(256, t, 1);
  printf("2 x v_mfma_i32_16x16x64_i8 (2 accumulators), one wave per SIMD: %.1f ticks per iteration (%.1f each)\n", t[0], t[0] / 2);
  launch<0, 1, false, 0, 0>(256, t, 1);
  printf("1 x v_mfma_i32_16x16x64_i8 (dependent chain), one wave per SIMD: %.1f ticks per iteration\n", t[0]);
  launch<0, 8, true, 0, 0>(256, t, 1);
  printf("8 x v_mfma_f64_16x16x4_f64, one wave per SIMD: %.1f ticks per iteration (%.1f each)\n", t[0], t[0] / 8);
  printf("ticks per iteration (s_memtime shader clocks); every VALU row: 16 instructions per iteration\n");
  probe<1>("v_fma_f64");
  probe<12>("v_mul_f64");
  probe<2>("v_cvt_f64_i32");
  probe<13>("v_cvt_f64_u32");
  probe<3>("v_cvt_i32_f64");
  probe<4>("v_rndne_f64");
  probe<9>("v_ldexp_f64");
  probe<11>("v_rcp_f64");
  probe<5>("v_perm_b32");
  probe<6>("v_mad_i64_i32");
  probe<7>("v_add_u32");
  probe<14>("v_lshl_add_u32");
  probe<8>("v_lshlrev_b64");
  probe<10>("v_bfe_i32");
  return 0;
}
