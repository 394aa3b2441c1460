// Micro-benchmark: issue rate of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4, 512 flop) vs v_mfma_f64_16x16x4_f64
// (2048 flop) on gfx950 -- decides whether a 4-sample-tile rollout kernel can lower the per-step latency at small K.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma4x4.hip -o tools/ubench_mfma4x4.bin && tools/ubench_mfma4x4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int SMALL>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  v4d acc[8];
  double acs[8];
  for (int i = 0; i < 8; ++i) { acc[i] = v4d{0, 0, 0, 0}; acs[i] = 0.0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (SMALL) acs[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acs[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acs[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SMALL>
void run(const char* name, int waves_per_simd) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * waves_per_simd, iters = 20000;
  double* out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SMALL>), dim3(blocks), dim3(256), 0, 0, out, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SMALL>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)blocks * 4 * iters * 8;
  const double flop = n_mfma * (SMALL ? 512.0 : 2048.0);
  // cycles per MFMA per SIMD at the nominal 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 / (n_mfma / (p.multiProcessorCount * 4.0));
  printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f TFLOP/s  ~%.1f cycles/MFMA/SIMD @2.4GHz\n", name, waves_per_simd, ms,
         flop / ms / 1e9, cyc);
  hipFree(out);
}

int main() {
  run<0>("v_mfma_f64_16x16x4_f64", 1);
  run<0>("v_mfma_f64_16x16x4_f64", 2);
  run<1>("v_mfma_f64_4x4x4_4b_f64", 1);
  run<1>("v_mfma_f64_4x4x4_4b_f64", 2);
  return 0;
}
