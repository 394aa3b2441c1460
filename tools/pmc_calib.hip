// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with THIS project's access widths:
// streams a buffer of known size with 8 B/lane and 16 B/lane loads (and 8 B/lane stores).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double* __restrict__ p, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += __builtin_nontemporal_load(p + i);
  if (s == 12345.678) out[0] = s;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double2 v = p[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;
}
__global__ void write8(double* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main() {
  const size_t n = (size_t)1 << 28;  // 2^28 doubles = 2 GiB  (>> 256 MiB Infinity Cache)
  double *p, *out;
  if (hipMalloc(&p, n * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  hipMemset(p, 0, n * 8);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, p, n, out);
  hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2*)p, n / 2, out);
  hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, p, n);
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", n * 8);
  return 0;
}
