// Micro-benchmark: FP64 MFMA vs FP64/FP32 VALU issue rates and their overlap on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o ubench_f64 && ./ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// mode bits: 1 = MFMA f64 stream, 2 = FP64 FMA stream, 4 = FP32 FMA stream.
// wave_split: 0 -> every wave runs all selected streams interleaved; 1 -> even waves MFMA only, odd waves VALU only
template <int MODE, int SPLIT>
__global__ __launch_bounds__(512) void k(double* out, int iters, unsigned long long* clk) {
  const int wave = threadIdx.x >> 6;
  v4d acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  double f[8];
  float g[8];
  for (int i = 0; i < 8; ++i) { f[i] = a + i; g[i] = (float)(a + i); }
  const bool do_mfma = (MODE & 1) && (!SPLIT || (wave & 1) == 0);
  const bool do_f64 = (MODE & 2) && (!SPLIT || (wave & 1) == 1);
  const bool do_f32 = (MODE & 4) && (!SPLIT || (wave & 1) == 1);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (do_mfma) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (do_f64) {
#pragma unroll
        for (int r = 0; r < 2; ++r) f[(i + r) & 7] = __builtin_fma(f[(i + r) & 7], b, a);
      }
      if (do_f32) {
#pragma unroll
        for (int r = 0; r < 2; ++r) g[(i + r) & 7] = __builtin_fmaf(g[(i + r) & 7], 1.0001f, 0.5f);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i] + g[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int MODE, int SPLIT>
void run(const char* name, int threads, int blocks_per_cu, int iters) {
  int dev = 0;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, dev);
  const int blocks = p.multiProcessorCount * blocks_per_cu;
  double* out;
  unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * threads * 8);
  hipMalloc(&clk, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, SPLIT>), dim3(blocks), dim3(threads), 0, 0, out, iters / 10, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, SPLIT>), dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double waves = (double)blocks * threads / 64;
  const double mfma_waves = (MODE & 1) ? (SPLIT ? waves / 2 : waves) : 0;
  const double valu_waves = (MODE & 6) ? (SPLIT ? waves / 2 : waves) : 0;
  const double mfma_flop = mfma_waves * iters * 8.0 * 2048;
  const double f64_flop = (MODE & 2) ? valu_waves * iters * 16.0 * 64 * 2 : 0;
  const double f32_flop = (MODE & 4) ? valu_waves * iters * 16.0 * 64 * 2 : 0;
  printf("%-46s thr/blk %4d blk/CU %d : %8.3f ms | MFMA %7.2f TF | VALU f64 %7.2f TF | VALU f32 %7.2f TF | memtime ticks/iter %.1f\n", name,
         threads, blocks_per_cu, ms, mfma_flop / ms / 1e9, f64_flop / ms / 1e9, f32_flop / ms / 1e9, (double)c / iters);
  hipFree(out);
  hipFree(clk);
}

int main() {
  const int it = 20000;
  run<1, 0>("MFMA f64 only, 1 wave/SIMD", 256, 1, it);
  run<1, 0>("MFMA f64 only, 2 waves/SIMD", 512, 1, it);
  run<2, 0>("VALU f64 fma only, 1 wave/SIMD", 256, 1, it);
  run<2, 0>("VALU f64 fma only, 2 waves/SIMD", 512, 1, it);
  run<4, 0>("VALU f32 fma only, 2 waves/SIMD", 512, 1, it);
  run<3, 0>("MFMA + f64 VALU interleaved in one wave, 1/SIMD", 256, 1, it);
  run<3, 1>("MFMA wave + f64 VALU wave (2 waves/SIMD)", 512, 1, it);
  run<5, 0>("MFMA + f32 VALU interleaved in one wave, 1/SIMD", 256, 1, it);
  run<5, 1>("MFMA wave + f32 VALU wave (2 waves/SIMD)", 512, 1, it);
  return 0;
}
