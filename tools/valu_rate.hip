// Dev tool: issue rate of a few VALU instructions on gfx950 (cycles per wave64 instruction per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin && tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIter = 4096;

#define BODY8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

template <int kKind>
__global__ __launch_bounds__(256) void rate(double* out, double c, uint32_t sel) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t r = threadIdx.x, acc = 0;
  for (int i = 0; i < kIter; ++i) {
    if (kKind == 0) {
#define OP(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "s"(c));
      BODY8(OP)
#undef OP
    } else if (kKind == 1) {
#define OP(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "s"(c));
      BODY8(OP)
#undef OP
    } else if (kKind == 2) {
#define OP(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "s"(c));
      BODY8(OP)
#undef OP
    } else if (kKind == 3) {
#define OP(x) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(r)); acc += s; }
      BODY8(OP)
#undef OP
    } else if (kKind == 4) {
#define OP(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "s"(c));
      BODY8(OP)
#undef OP
    } else if (kKind == 5) {   // the masked pair of the compatibility kernel
#define OP(x) asm volatile("s_mov_b64 exec, %1\n v_mul_f64 %0, %0, %2\n s_not_b64 exec, exec\n v_mul_f64 %0, %0, %2\n s_mov_b64 exec, -1" : "+v"(x) : "s"((uint64_t)sel * 0x100000001ull), "s"(c) : "scc");
      BODY8(OP)
#undef OP
    } else if (kKind == 6) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(lo) : "v"(r)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 13) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(lo) : "v"(r)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 14) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(r)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 15) {   // VOP2 form: the mask is VCC
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(r) : "vcc"); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 16) {   // the select-by-mask factor: 2 x v_cndmask_b32 (SGPR-pair mask) + v_mul_f64
#define OP(x) { uint32_t hi, lo; asm volatile("v_cndmask_b32 %0, %3, %4, %7\n v_cndmask_b32 %1, %5, %6, %7\n v_mul_f64 %2, %2, %[f]" : "=&v"(hi), "=&v"(lo), "+v"(x) : "v"(0x3F50624D), "v"(0x3FEFF7CE), "v"(0xD2F1A9FC), "v"(0xD916872B), "s"((uint64_t)sel * 0x100000001ull), [f] "v"(c)); acc += hi ^ lo; }
      BODY8(OP)
#undef OP
    } else if (kKind == 7) {   // 32-bit ops on eight independent registers (the low halves of a0..a7)
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(lo) : "v"(r), "s"(sel)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 8) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_bfe_i32 %0, %0, %1, 1" : "+v"(lo) : "v"(r)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 9) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(r), "s"((uint64_t)sel * 0x100000001ull)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 10) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(r), "s"(sel)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 11) {
#define OP(x) { uint32_t lo = (uint32_t)__double2loint(x); asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(r)); x = __hiloint2double(__double2hiint(x), (int)lo); }
      BODY8(OP)
#undef OP
    } else if (kKind == 12) {   // the factor of the compatibility kernel: v_bfe_i32 + 2 x v_bfi_b32 + v_mul_f64
#define OP(x) { int32_t m; uint32_t hi, lo; asm volatile("v_bfe_i32 %0, %4, %5, 1\n v_bfi_b32 %1, %0, %6, %7\n v_bfi_b32 %2, %0, %8, %9\n v_mul_f64 %3, %3, %[f]" : "=&v"(m), "=&v"(hi), "=&v"(lo), "+v"(x) : "v"(r), "v"(sel), "s"(0x3FEFF7CE), "v"(0x3F50624D), "s"(0xD916872B), "v"(0xD2F1A9FC), [f] "v"(c)); acc += hi ^ lo; }
      BODY8(OP)
#undef OP
    } else if (kKind == 17) {   // what the compatibility kernel compiles to: s_xor_b64 vcc + 2 x v_cndmask_b32_e32 (VOP2) + v_mul_f64
#define OP(x) { uint32_t hi, lo; asm volatile("s_xor_b64 vcc, %7, %8\n v_cndmask_b32_e32 %0, %3, %4, vcc\n v_cndmask_b32_e32 %1, %5, %6, vcc\n v_mul_f64 %2, %2, %[f]" : "=&v"(hi), "=&v"(lo), "+v"(x) : "v"(0x3F50624D), "v"(0x3FEFF7CE), "v"(0xD2F1A9FC), "v"(0xD916872B), "s"((uint64_t)sel * 0x100000001ull), "s"((uint64_t)sel << 7), [f] "v"(c) : "vcc"); acc += hi ^ lo; }
      BODY8(OP)
#undef OP
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc + r;
}

template <int kKind>
void run(const char* name, int ops_per_body, int waves_per_simd = 8) {
  double* out;
  hipMalloc(&out, 256 * 4 * 8 * 256 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;   // N blocks of 4 waves per CU: N waves per SIMD
  rate<kKind><<<blocks, 256>>>(out, 1.0000001, 0x0F0F0F0F);
  hipEventRecord(e0);
  rate<kKind><<<blocks, 256>>>(out, 1.0000001, 0x0F0F0F0F);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 8 waves x kIter x ops instructions
  const double instr_per_simd = (double)waves_per_simd * kIter * ops_per_body;
  const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
  printf("%-28s %d waves/SIMD %8.3f ms  %6.2f cycles per wave64 instruction and SIMD (at 2.4 GHz) = %6.2f T lane-ops/s on 1024 SIMDs\n",
         name, waves_per_simd, ms, cyc, 1024 * 64 * 2.4e9 / cyc / 1e12);
  hipFree(out);
}

// shader clock under a VALU load: s_memtime ticks (core clock) against the constant 100 MHz wall clock
__global__ __launch_bounds__(256) void clock_probe(double* out, unsigned long long* ticks) {
  double a = threadIdx.x;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < (1 << 18); ++i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a));
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

int main() {
  {
    double* out; unsigned long long* t; unsigned long long h[2];
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double)); hipMalloc(&t, 16);
    clock_probe<<<256 * 8, 256>>>(out, t);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("shader clock under a full VALU load: %llu s_memtime ticks in %llu wall ticks of 10 ns = %.0f MHz if s_memtime counts core cycles\n",
           h[0], h[1], (double)h[0] / ((double)h[1] * 10e-9) / 1e6);
    hipFree(out); hipFree(t);
  }
  // guide (MI355X_MICROARCH.md, constants table): v_fma_f32 wave64 = 2 cycles on a SIMD-32 with >= 2 waves, 4 for one wave
  // alone; peak FP32 vector 157.3 TFLOP/s = 78.6 T lane-FMAs/s; f64 vector 78.6 TFLOP/s = 39.3 T lane-FMAs/s
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_mul_f64", 8, w);
    run<1>("v_add_f64", 8, w);
    run<2>("v_fma_f64", 8, w);
    run<4>("v_max_f64", 8, w);
    run<6>("v_mul_f32 (VOP2)", 8, w);
    run<13>("v_fma_f32 (VOP3)", 8, w);
    run<14>("v_and_b32 (VOP2)", 8, w);
    run<11>("v_add_u32", 8, w);
    run<7>("v_bfi_b32", 8, w);
    run<8>("v_bfe_i32", 8, w);
    run<9>("v_cndmask_b32 (VOP3, sgpr mask)", 8, w);
    run<10>("v_sad_u8", 8, w);
    run<12>("compat factor (bfe+2bfi+mul64)", 32, w);
    run<16>("select factor (2cndmask_e64+mul64)", 24, w);
    run<17>("select factor (vcc: 2cndmask_e32+mul64)", 24, w);
  }
  run<3>("v_readlane_b32", 8);
  run<5>("masked v_mul_f64 pair", 16);
  return 0;
}
