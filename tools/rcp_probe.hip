// Accuracy of v_rcp_f64 (after 1 / 2 Newton steps, and after the one cubic step nlc_math.h uses) on gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* r3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  {
    const double e = fma(-d, r, 1.0);
    r3[i] = fma(r, fma(e, e, e), r);  // r (1 + e + e^2): 3 FMAs
  }
  r = fma(fma(-d, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-d, r, 1.0), r, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n), q(n);
  for (int i = 0; i < n; ++i) x[i] = 1.0 + (double)i / n + 1e-9 * (i % 7);  // [1, 2): denominators 1+e, 2+em
  double *dx, *d0, *d1, *d2, *d3;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(q.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    e0 = fmax(e0, (double)fabsl((a[i] - t) / t));
    e1 = fmax(e1, (double)fabsl((b[i] - t) / t));
    e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
    e3 = fmax(e3, (double)fabsl((q[i] - t) / t));
  }
  printf("max rel err: v_rcp_f64 %.3e (2^%.1f) | +1 Newton %.3e | +2 Newton %.3e | one cubic step %.3e (ulp = 1.1e-16)\n", e0,
         log2(e0), e1, e2, e3);
  return 0;
}
