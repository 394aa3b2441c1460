// Attainable HBM READ bandwidth on this chip for a streaming kernel of the ILT kernel's size (917.5 MB read, ~26 MB written):
// grid-stride 16-B loads, a running sum per lane, one 8-B store per 280 B read.  tools only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm_read.hip -o tools/ubench_hbm_read.bin && tools/ubench_hbm_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d_t __attribute__((ext_vector_type(2)));
template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const v2d_t* __restrict__ in, double* __restrict__ out, size_t n16) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    v2d_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
  }
  for (; i < n16; i += stride) {
    const v2d_t v = in[i];
    acc += v.x + v.y;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const size_t bytes = 917504000;
  const size_t n16 = bytes / 16;
  v2d_t* in;
  double* out;
  if (hipMalloc(&in, bytes) != hipSuccess) return 1;
  const int kMaxGrid = 16384;
  if (hipMalloc(&out, (size_t)kMaxGrid * 256 * sizeof(double)) != hipSuccess) return 1;  // one double per thread of the largest grid
  hipMemset(in, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {1024, 2048, 4096, 8192, kMaxGrid}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((rd<8>), dim3(grid), dim3(256), 0, 0, in, out, n16);
    hipEventRecord(e0);
    const int R = 20;
    for (int rep = 0; rep < R; ++rep) hipLaunchKernelGGL((rd<8>), dim3(grid), dim3(256), 0, 0, in, out, n16);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %5d x 256, 8 x 16-B loads in flight per lane: %.4f ms per pass = %.0f GB/s\n", grid, ms / R, bytes / (ms / R) / 1e6);
  }
  return 0;
}
