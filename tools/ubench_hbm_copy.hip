// Attainable HBM READ + WRITE bandwidth for a stream of the Fourier ILT BACKWARD kernel's shape: two input arrays read, two
// output arrays written, N*d*S doubles each (891 MB each at N = 655 360, d = 5, S = 17), grid-stride 16-B accesses.  tools only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm_copy.hip -o tools/ubench_hbm_copy.bin && tools/ubench_hbm_copy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d_t __attribute__((ext_vector_type(2)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void cp2(const v2d_t* __restrict__ a, const v2d_t* __restrict__ b, v2d_t* __restrict__ x,
                                           v2d_t* __restrict__ y, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    v2d_t va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      va[u] = __builtin_nontemporal_load(a + i + u * stride);
      vb[u] = __builtin_nontemporal_load(b + i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const v2d_t s = va[u] + vb[u], d = va[u] - vb[u];
      if (NT) {
        __builtin_nontemporal_store(s, x + i + u * stride);
        __builtin_nontemporal_store(d, y + i + u * stride);
      } else {
        x[i + u * stride] = s;
        y[i + u * stride] = d;
      }
    }
  }
  for (; i < n16; i += stride) {
    x[i] = a[i] + b[i];
    y[i] = a[i] - b[i];
  }
}
int main() {
  const size_t bytes = (size_t)655360 * 5 * 17 * 8;  // one array
  const size_t n16 = bytes / 16;
  v2d_t *a, *b, *x, *y;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&x, bytes) != hipSuccess ||
      hipMalloc(&y, bytes) != hipSuccess)
    return 1;
  hipMemset(a, 0, bytes);
  hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int nt = 0; nt < 2; ++nt)
    for (int grid : {1024, 2048, 4096, 8192}) {
      auto launch = [&] {
        if (nt)
          hipLaunchKernelGGL((cp2<4, true>), dim3(grid), dim3(256), 0, 0, a, b, x, y, n16);
        else
          hipLaunchKernelGGL((cp2<4, false>), dim3(grid), dim3(256), 0, 0, a, b, x, y, n16);
      };
      for (int rep = 0; rep < 3; ++rep) launch();
      hipEventRecord(e0);
      const int R = 20;
      for (int rep = 0; rep < R; ++rep) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d x 256, %s stores: %.4f ms per pass = %.0f GB/s (2 arrays read + 2 written, %.0f MB)\n", grid,
             nt ? "nontemporal" : "plain", ms / R, 4.0 * bytes / (ms / R) / 1e6, 4.0 * bytes / 1e6);
    }
  return 0;
}
