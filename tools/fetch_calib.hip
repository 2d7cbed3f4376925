// dev tool: known-size streaming reads / writes with 8 B per lane (the access width of the search
// kernels) to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void read8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += in[i];
  if (s == 12345.678) out[0] = s;   // never true: keeps the loads
}
__global__ __launch_bounds__(256) void write8(double* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = 1.0;
}
int main() {
  const size_t n = (size_t)1 << 28;   // 2 GiB of doubles: far beyond the 256 MiB Infinity Cache
  double *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, 4096);
  hipMemset(a, 0, n * 8);
  for (int rep = 0; rep < 3; ++rep) {
    read8<<<4096, 256>>>(a, b, n);
    write8<<<4096, 256>>>(a, n);
  }
  hipDeviceSynchronize();
  printf("read8 / write8: %zu bytes per launch\n", n * 8);
  return 0;
}
