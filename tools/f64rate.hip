// dev tool: issue rate of v_max_f64 / v_add_f64 / v_fma_f64 on gfx950 (register-only loops)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double x, double y) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = a[i] + x;
      if (MODE == 1) a[i] = vmax(a[i], x + i);
      if (MODE == 2) a[i] = a[i] + vmax(y + i, x);
      if (MODE == 3) a[i] = __builtin_fma(a[i], x, y);
    }
  }
  double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int ops_per) {
  double* d; hipMalloc(&d, 1024 * 256 * 8 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4096, blocks = 256 * 8;
  k<MODE><<<blocks, 256>>>(d, iters, 1.0000001, 0.5);
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, iters, 1.0000001, 0.5); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * iters * 16 * ops_per;
  printf("%-28s %.3f ms  %.2f Tinstr-lanes/s\n", name, ms, ops / ms / 1e9);
}
int main() { run<0>("v_add_f64", 1); run<1>("v_max_f64 (+add for operand)", 2); run<2>("add+max", 2); run<3>("v_fma_f64", 1); return 0; }
