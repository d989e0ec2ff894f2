// Diagnostic (not product code): VERDICT r2 item 4 -- is a 5 x 52-bit-limb product on v_fma_f64 cheaper than the 8 x 32-bit
// product on v_mad_u64_u32 that fe_mul_gfx950.inc is made of?  Both instructions issue at the same ~4.1-4.3 cycles per
// wave-instruction per SIMD (profiles/r02_op_rates.txt), so the question is instruction count.
//
//   A  integer:  the 16-limb product a * b, product scanning, one v_mad_u64_u32 + one v_addc_co_u32 per partial product
//                (64 + 64; what the a * b part of the shipped multiply costs -- its reduction adds 32 + 32 + glue)
//   B  FP64:     the 25 partial products of 5 x 52-bit limbs split exactly into high and low halves the way Emmart, Zheng
//                and Weems do it (ARITH 2018): hi = fma_rz(a, b, 2^104), lo = fma_rz(a, b, (2^104 + 2^52) - hi); the
//                mantissas are integers, accumulated per column as raw 64-bit integers (the biases come off at the end)
// (Chaining the high halves inside the fma's addend -- h = fma(a, b, h) is exact while h stays in [2^104, 2^105) -- would
//  save the integer add of the high half, but five products of 52-bit limbs leave that binade; it needs limbs of 50 bits,
//  i.e. 6 limbs and 36 partial products for a 255-bit field: more instructions, not fewer.  profiles/r03_fma52_probe.txt.)
// Output: static VALU instruction counts (llvm-objdump over this binary, tools/ubench/fma52_count.sh) and measured time per
// product per lane at 1..3 waves per SIMD.  No Montgomery reduction in any variant: in radix 2^52 it is dearer than in 2^32
// (m = 1 mod 2^32 makes the 32-bit quotient digit a negation; mod 2^52 it is a multiplication) -- see profiles/r03_fma52_probe.txt.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/fma52_probe.hip -o tools/ubench/fma52_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// ---- A: 8 x 32 integer product (16 limbs out) ------------------------------------------------------------------
__device__ __forceinline__ void int_product(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&r)[16]) {
  uint64_t acc = 0;
  uint32_t hi = 0;
#pragma unroll
  for (int col = 0; col < 15; ++col) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = col - i;
      if (j < 0 || j > 7) continue;
      // acc(64) + a_i * b_j with the carry into a third word: v_mad_u64_u32 + v_addc_co_u32
      uint32_t c;
      asm("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi), "=s"(c) : "v"(a[i]), "v"(b[j]) : "vcc");
    }
    r[col] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)hi << 32);
    hi = 0;
  }
  r[15] = (uint32_t)acc;
}

// ---- B, C: 5 x 52 FP64 product (10 column sums of high parts and of low parts, raw integer bits) ------------------
__device__ __forceinline__ void set_f64_round_toward_zero() {
  // MODE.FP_ROUND[3:2] = 3: round toward zero for f64 / f16 (plain v_fma_f64 then truncates; hwreg id 1 = MODE)
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}
__device__ __forceinline__ void fp_product_emmart(const double (&a)[5], const double (&b)[5], uint64_t (&H)[9], uint64_t (&L)[9]) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
#pragma unroll
  for (int k = 0; k < 9; ++k) { H[k] = 0; L[k] = 0; }
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double hi = __builtin_fma(a[i], b[j], C1);
      const double lo = __builtin_fma(a[i], b[j], C2 - hi);
      H[i + j] += (uint64_t)__double_as_longlong(hi);
      L[i + j] += (uint64_t)__double_as_longlong(lo);
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(int iters, uint64_t* out, uint32_t* sink) {
  uint32_t a[8], b[8], r[16];
  double fa[5], fb[5];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = blockIdx.x * 40503u + i * 7 + 1; }
  for (int i = 0; i < 5; ++i) { fa[i] = (double)((uint64_t)(threadIdx.x * 2654435761u + i) << 19); fb[i] = (double)((uint64_t)(blockIdx.x * 40503u + 977u * i + 1) << 17); }
  if (MODE != 0) set_f64_round_toward_zero();
  uint64_t fold = 0;
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < iters; ++k) {
    if (MODE == 0) {
      int_product(a, b, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] ^= r[i + 8]; b[i] += r[i]; }        // dependent chain, as in a field operation
    } else {
      uint64_t H[9], L[9];
      fp_product_emmart(fa, fb, H, L);
#pragma unroll
      for (int i = 0; i < 9; ++i) fold += H[i] ^ (L[i] << 1);
#pragma unroll
      for (int i = 0; i < 5; ++i) { fa[i] = (double)((fold >> (3 * i)) & 0xFFFFFFFFFFFFFull); }
    }
  }
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) out[((size_t)blockIdx.x * 256 + threadIdx.x) / 64] = r1 - r0;
  uint32_t s = (uint32_t)fold;
  for (int i = 0; i < 8; ++i) s ^= a[i] + b[i];
  if (s == 0x12345678u) sink[0] = s;
}

template <int MODE> int run(const char* what) {
  const int iters = 2000;
  for (int wg = 1; wg <= 3; ++wg) {
    const int blocks = 256 * wg, waves = blocks * 4;
    uint64_t* d; uint32_t* s;
    CK(hipMalloc(&d, waves * 8)); CK(hipMalloc(&s, 4));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, 400, d, s);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d, s);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // wall time of the launch / (products per SIMD) : ns per wave-product per SIMD
    printf("%-44s %d waves/SIMD: %.1f ns per product per SIMD slot (kernel wall time %.3f ms)\n", what, wg, ms * 1e6 / iters / wg, ms);
    CK(hipFree(d)); CK(hipFree(s));
  }
  return 0;
}

int main() {
  if (run<0>("A  8x32 integer product (64 mad + 64 addc)")) return 1;
  if (run<1>("B  5x52 FP64 product, Emmart split")) return 1;
  return 0;
}
