// Micro-benchmark: do transcendental-pipe instructions (v_exp_f32, v_rcp_f32, v_rcp_f64, v_rsq_f64,
// v_sqrt_f64, v_ldexp_f64, v_cvt) issue UNDER a running v_mfma_f64_16x16x4_f64 of the same wave, where plain FP64 / FP32
// FMAs do not (profiles/r1_ubench_f64_mfma_valu.txt: 8 MFMAs + 16 FMAs = 512 + 72 cycles)?
// One wave per SIMD, 8 MFMAs per iteration (512 cycles alone) interleaved with 16 instructions of the probed kind.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_trans.hip -o /tmp/ubench_trans && /tmp/ubench_trans
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND, bool MFMA>
__global__ __launch_bounds__(256) void k(double* out, int iters, unsigned long long* clk) {
  v4d acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  double f[8];
  float g[8];
  for (int i = 0; i < 8; ++i) { f[i] = 1.0 + a + i; g[i] = (float)(1.0 + a + i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MFMA) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int j = (i + r) & 7;
        if (KIND == 1) f[j] = __builtin_fma(f[j], b, a);
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(g[j]));
        if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(g[j]));
        if (KIND == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[j]));
        if (KIND == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(f[j]));
        if (KIND == 6) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(f[j]));
        if (KIND == 7) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(g[j]) : "v"(f[j]));
        if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[j]) : "v"(b));
        if (KIND == 9) asm volatile("v_add_u32 %0, %0, %0" : "+v"(g[j]));
        if (KIND == 10) asm volatile("v_log_f32 %0, %0" : "+v"(g[j]));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i] + g[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount, iters = 20000;
  double* out;
  unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipMalloc(&clk, 8);
  double ticks[2];
  for (int m = 0; m < 2; ++m) {
    if (m == 0) hipLaunchKernelGGL((k<KIND, false>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    else hipLaunchKernelGGL((k<KIND, true>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    ticks[m] = (double)c / iters;
  }
  // s_memtime ticks are shader clocks here (the pure-MFMA loop reads 512.0 per 8 MFMAs)
  printf("%-22s 16 instr alone %6.1f ticks (%.1f each) | with 8 MFMA f64 %6.1f ticks -> added %6.1f (%.1f per instr; 0 = hidden under the MFMAs)\n",
         name, ticks[0], ticks[0] / 16.0, ticks[1], ticks[1] - 512.0, (ticks[1] - 512.0) / 16.0);
  hipFree(out);
  hipFree(clk);
}

int main() {
  run<1>("v_fma_f64");
  run<8>("v_mul_f64");
  run<2>("v_exp_f32");
  run<10>("v_log_f32");
  run<3>("v_rcp_f32");
  run<4>("v_rcp_f64");
  run<5>("v_rsq_f64");
  run<6>("v_ldexp_f64");
  run<7>("v_cvt_f32_f64");
  run<9>("v_add_u32");
  return 0;
}
