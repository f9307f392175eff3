// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for this library's access
// width (8 B per lane, 512 B per wave instruction) on a known byte count, as
// MI355X_MICROARCH.md "HBM" asks for widths other than 16 B per lane.
//   read8 : streams N doubles in (1 GiB, past the 256 MiB Infinity Cache)
//   write8: streams N doubles out
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double* __restrict__ x, double* o, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  double s = 0.0;
  for (; i < n; i += stride) s += x[i];
  if (s == 12345.678) o[0] = s;
}
__global__ void write8(double* __restrict__ x, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] = (double)i;
}
int main() {
  const long n = 1L << 27;
  double *a, *o;
  hipMalloc(&a, n * 8); hipMalloc(&o, 8);
  hipMemset(a, 0, n * 8);
  for (int r = 0; r < 3; ++r) {
    write8<<<4096, 256>>>(a, n);
    read8<<<4096, 256>>>(a, o, n);
  }
  hipDeviceSynchronize();
  printf("bytes per launch: %ld\n", n * 8);
  return 0;
}
