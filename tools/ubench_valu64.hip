// Micro-benchmark: issue cost of the FP64 VALU instructions the de Hoog / gate-math code is made of, on gfx950:
// v_fma_f64, v_mul_f64, v_add_f64, v_rcp_f64 (8 independent chains per lane), at 1, 2 and 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu64.hip -o tools/ubench_valu64.bin && tools/ubench_valu64.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
  double f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.0 + threadIdx.x * 1e-3 + i;
  const double b = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) f[i] = __builtin_fma(f[i], b, c);
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(b));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(c));
      if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[i]));
      if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(*(float*)&f[i]));
      if (OP == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(f[i]));
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int threads = 256 * waves_per_simd, blocks = p.multiProcessorCount, iters = 20000;
  double* out;
  (void)hipMalloc(&out, (size_t)blocks * threads * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, iters);  // warm clocks
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)waves_per_simd * iters * 8;
  printf("%-10s waves/SIMD %d: %8.3f ms  %.1f cycles per wave-instruction per SIMD @2.4 GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
  (void)hipFree(out);
}

int main() {
  const int ws[3] = {1, 2, 4};
  for (int w : ws) run<0>("v_fma_f64", w);
  for (int w : ws) run<1>("v_mul_f64", w);
  for (int w : ws) run<2>("v_add_f64", w);
  for (int w : ws) run<3>("v_rcp_f64", w);
  for (int w : ws) run<5>("v_rsq_f64", w);
  for (int w : ws) run<4>("v_rcp_f32", w);
  return 0;
}
