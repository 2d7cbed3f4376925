// Copy-number model fit (LCND, "CNgroup"): cn_model.py:124-204.
//
// For every candidate base depth b (one workgroup each) and every depth bin x (threads):
//   p(x | b) = max_n  N(x; b*n, dev_n) * bin_width          (calcCNGroupProb 179-204)
//   loglik(b) = sum_x log(p(x | b) + 1e-9) * histogram[x]   (fit 153-164)
// The host supplies the bin centres, candidate bases and per-CN deviations exactly as numpy's
// linspace / the reference's formulas produce them; exp / log come from the device math library,
// so likelihoods agree with scipy to a few ulp and the integer copy numbers are identical unless
// two candidate bases tie to that precision.
#include "gk_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCN = 16;

__global__ __launch_bounds__(kThreads) void cn_fit_kernel(const double* __restrict__ x, const double* __restrict__ density,
                                                          int bins, const double* __restrict__ bases,
                                                          const double* __restrict__ dev, int n_cn, int first_cn,
                                                          double space, double* __restrict__ loglik) {
  __shared__ double part[kThreads];
  const double base = bases[blockIdx.x];
  const double inv_sqrt_2pi_den = 2.5066282746310002;   // sqrt(2*pi), scipy's _norm_pdf_C
  double acc = 0.0;
  for (int i = threadIdx.x; i < bins; i += kThreads) {
    const double xi = x[i];
    double best = 0.0;
    for (int k = 0; k < n_cn; ++k) {
      const double loc = base * (double)(first_cn + k);
      const double z = (xi - loc) / dev[k];
      const double pdf = exp(-(z * z) / 2.0) / inv_sqrt_2pi_den / dev[k];
      const double v = pdf * space;
      best = k == 0 ? v : (v > best ? v : best);
    }
    acc += log(best + 1e-9) * density[i];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loglik[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(kThreads) void cn_assign_kernel(const double* __restrict__ x, int bins, double base,
                                                             const double* __restrict__ dev, int n_cn, int first_cn,
                                                             double space, int32_t* __restrict__ cn_of_bin) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= bins) return;
  double best = 0.0;
  int arg = 0;
  for (int k = 0; k < n_cn; ++k) {
    const double loc = base * (double)(first_cn + k);
    const double z = (x[i] - loc) / dev[k];
    const double pdf = exp(-(z * z) / 2.0) / 2.5066282746310002 / dev[k] * space;
    if (k == 0 || pdf > best) { best = pdf; arg = k; }   // first maximum, like numpy argmax
  }
  cn_of_bin[i] = arg;
}

}  // namespace

extern "C" {

// loglik_out[j] for bases[j]; cn_of_bin_out (may be null) = argmax CN row per bin at best_base.
int gk_cn_fit(gk_ctx* ctx, const double* x, const double* density, int32_t bins, const double* bases, int32_t n_bases,
              const double* dev, int32_t n_cn, int32_t first_cn, double space, double* loglik_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && x && density && bases && dev && loglik_out, "null pointer");
  GK_REQUIRE(bins > 0 && n_bases > 0 && n_cn > 0 && n_cn <= kMaxCN, "bad CN fit geometry");
  hipStream_t st = ctx->stream;
  double* d = nullptr;
  const size_t n_all = (size_t)2 * bins + n_bases + n_cn + n_bases;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d, n_all * sizeof(double)));
  double *d_x = d, *d_den = d + bins, *d_bases = d + 2 * bins, *d_dev = d_bases + n_bases, *d_ll = d_dev + n_cn;
  GK_HIP(hipMemcpyAsync(d_x, x, (size_t)bins * sizeof(double), hipMemcpyHostToDevice, st));
  GK_HIP(hipMemcpyAsync(d_den, density, (size_t)bins * sizeof(double), hipMemcpyHostToDevice, st));
  GK_HIP(hipMemcpyAsync(d_bases, bases, (size_t)n_bases * sizeof(double), hipMemcpyHostToDevice, st));
  GK_HIP(hipMemcpyAsync(d_dev, dev, (size_t)n_cn * sizeof(double), hipMemcpyHostToDevice, st));
  GK_KERNEL(cn_fit_kernel, dim3((unsigned)n_bases), dim3(kThreads), 0, st, d_x, d_den, bins, d_bases, d_dev,
                     n_cn, first_cn, space, d_ll);
  GK_HIP(hipGetLastError());
  GK_HIP(hipMemcpyAsync(loglik_out, d_ll, (size_t)n_bases * sizeof(double), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx, d);
  return GK_OK;
}

int gk_cn_assign(gk_ctx* ctx, const double* x, int32_t bins, double base, const double* dev, int32_t n_cn,
                 int32_t first_cn, double space, int32_t* cn_of_bin_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && x && dev && cn_of_bin_out && bins > 0 && n_cn > 0 && n_cn <= kMaxCN, "bad CN assign arguments");
  hipStream_t st = ctx->stream;
  double* d = nullptr;
  int32_t* d_cn = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d, (size_t)(bins + n_cn) * sizeof(double)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_cn, (size_t)bins * sizeof(int32_t)));
  GK_HIP(hipMemcpyAsync(d, x, (size_t)bins * sizeof(double), hipMemcpyHostToDevice, st));
  GK_HIP(hipMemcpyAsync(d + bins, dev, (size_t)n_cn * sizeof(double), hipMemcpyHostToDevice, st));
  GK_KERNEL(cn_assign_kernel, dim3((unsigned)((bins + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, d,
                     bins, base, d + bins, n_cn, first_cn, space, d_cn);
  GK_HIP(hipGetLastError());
  GK_HIP(hipMemcpyAsync(cn_of_bin_out, d_cn, (size_t)bins * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx, d);
  gk_pool_free(ctx, d_cn);
  return GK_OK;
}

}  // extern "C"
