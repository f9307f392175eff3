// Measures the accuracy of v_rcp_f64 / v_rsq_f64 (raw and after one
// Newton-Raphson step) on the device; used to justify agx_device.hpp's
// fast_rcp()/fast_rsqrt().  hipcc --offload-arch=gfx950 -O2 -o rcp_accuracy ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  double q0 = __builtin_amdgcn_rsq(v);
  // y' = y + y*(0.5 - 0.5*v*y*y)
  double h = 0.5 * q0;
  double q1 = fma(h, fma(-v * q0, q0, 1.0), q0);
  o[4 * i] = r0; o[4 * i + 1] = r1; o[4 * i + 2] = q0; o[4 * i + 3] = q1;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), o(4 * n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> e(-60.0, 60.0), m(1.0, 2.0);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), (int)e(g));
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, s0 = 0, s1 = 0;
  for (int i = 0; i < n; ++i) {
    double rr = 1.0 / x[i], qq = 1.0 / std::sqrt(x[i]);
    e0 = fmax(e0, fabs(o[4 * i] - rr) / rr); e1 = fmax(e1, fabs(o[4 * i + 1] - rr) / rr);
    s0 = fmax(s0, fabs(o[4 * i + 2] - qq) / qq); s1 = fmax(s1, fabs(o[4 * i + 3] - qq) / qq);
  }
  printf("rcp raw %.3e  rcp+1NR %.3e  rsq raw %.3e  rsq+1NR %.3e\n", e0, e1, s0, s1);
  return 0;
}
