// Dev tool: do small device-to-host copies into PAGEABLE memory serialise the host threads of one process?
// T threads, each with its own stream: (tiny kernel, 4 KB copy back, synchronise) x N, pageable vs pinned target.
//   hipcc --offload-arch=gfx950 -O3 tools/d2h_contention.hip -o tools/d2h_contention.bin -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void touch(int* p) { p[threadIdx.x] = threadIdx.x; }

static double run(int n_threads, int iters, bool pinned, size_t bytes) {
  std::vector<std::thread> pool;
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < n_threads; ++t)
    pool.emplace_back([=] {
      hipStream_t st;
      hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      int* d;
      hipMalloc(&d, bytes);
      void* h;
      if (pinned) hipHostMalloc(&h, bytes, hipHostMallocDefault); else h = malloc(bytes);
      for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, d);
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
      }
      hipFree(d);
      if (pinned) hipHostFree(h); else free(h);
      hipStreamDestroy(st);
    });
  for (auto& th : pool) th.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  const int iters = 2000;
  run(1, 100, false, 4096);
  for (size_t bytes : {4096ul, 65536ul})
    for (int n : {1, 2, 6, 12})
      for (bool pinned : {false, true}) {
        const double s = run(n, iters, pinned, bytes);
        printf("%6zu B  threads %2d  %-8s  %7.1f us per (kernel + copy + sync) per thread, %8.0f rounds/s in total\n", bytes, n,
               pinned ? "pinned" : "pageable", 1e6 * s / iters, n * iters / s);
      }
  return 0;
}
