// Diagnostic (not product code): what would a bucket addition cost in a carry-free 9 x 29-bit-limb representation with the
// table points stored natively in it?  (the lever left for k_accumulate after the FP64 route
// closed.)  Register data only, no memory traffic, 1..3 workgroups per CU, same harness as madd_probe.hip:
//   A  the shipped lazy mixed addition (8 x 32-bit limbs, fe_mul_gfx950.inc): 10 multiplications + 7 subtractions
//   B  the same addition over F29: 8 multiplications + 2 squarings (45 instead of 81 products), limb-wise subtractions with
//      borrow-proof multiples of m, 4 normalisations; the affine operand normalised (as a table in that form would hold it)
// Not bit-checked here (docs/experiments/f29 was, in round 1); the question is the instruction count and the time.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vdf_amd/csrc tools/ubench/f29_probe.hip -o tools/ubench/f29_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ec.cuh"
#include "f29_consts.h"
using namespace vdf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct F29 { uint32_t v[9]; };
static constexpr uint32_t MASK29 = 0x1FFFFFFFu;
__device__ __forceinline__ uint64_t shr29(uint64_t c) {
  const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
  return ((uint64_t)(hi >> 29) << 32) | __builtin_amdgcn_alignbit(hi, lo, 29);
}
__device__ __forceinline__ void f29_normalize(F29& a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { a.v[i + 1] += a.v[i] >> 29; a.v[i] &= MASK29; }
}
// Montgomery reduction of 17 column sums (radix 2^29, m = 1 mod 2^29, limbs 5..7 of m zero, limb 8 = 2^22)
__device__ __forceinline__ F29 f29_reduce(uint64_t (&c)[18]) {
  uint32_t m8;
  asm("s_mov_b32 %0, 0x400000" : "=s"(m8));
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t q = (0u - (uint32_t)c[i]) & MASK29;
    c[i + 1] += (uint64_t)q * F29_M[1];
    c[i + 2] += (uint64_t)q * F29_M[2];
    c[i + 3] += (uint64_t)q * F29_M[3];
    c[i + 4] += (uint64_t)q * F29_M[4];
    c[i + 8] += (uint64_t)q * m8;
    c[i + 1] += shr29(c[i] + q);
  }
  F29 r;
#pragma unroll
  for (int k = 9; k < 17; ++k) { c[k + 1] += shr29(c[k]); r.v[k - 9] = (uint32_t)c[k] & MASK29; }
  r.v[8] = (uint32_t)c[17];
  return r;
}
__device__ __forceinline__ F29 f29_mul(const F29& a, const F29& b) {       // a normalised, b limbs < 2^31.3
  uint64_t c[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) c[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a.v[j] * b.v[i];
  return f29_reduce(c);
}
__device__ __forceinline__ F29 f29_sqr(const F29& a) {                     // a normalised: 45 products
  uint64_t c[18];
  uint32_t d[9];
#pragma unroll
  for (int i = 0; i < 18; ++i) c[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = a.v[i] << 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    c[2 * i] += (uint64_t)a.v[i] * a.v[i];
#pragma unroll
    for (int j = i + 1; j < 9; ++j) c[i + j] += (uint64_t)a.v[i] * d[j];
  }
  return f29_reduce(c);
}
__device__ __forceinline__ F29 f29_sub(const F29& a, const F29& b, const uint32_t (&bias)[9]) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + bias[i] - b.v[i];
  return r;
}

struct XYZZ29 { F29 x, y, zz, zzz; };
__device__ __forceinline__ void madd29(XYZZ29& acc, const F29& bx, const F29& by) {
  const F29 U2 = f29_mul(bx, acc.zz), S2 = f29_mul(by, acc.zzz);
  F29 Pp = f29_sub(U2, acc.x, F29_B_30); f29_normalize(Pp);
  F29 Rr = f29_sub(S2, acc.y, F29_B_30); f29_normalize(Rr);
  const F29 PP = f29_sqr(Pp), PPP = f29_mul(Pp, PP), Qq = f29_mul(PP, acc.x), RR = f29_sqr(Rr);
  F29 X3;
#pragma unroll
  for (int i = 0; i < 9; ++i) X3.v[i] = RR.v[i] + F29_B_31[i] - PPP.v[i] - 2u * Qq.v[i];
  f29_normalize(X3);
  const F29 T = f29_sub(Qq, X3, F29_B_30);
  F29 Y3 = f29_sub(f29_mul(Rr, T), f29_mul(PPP, acc.y), F29_B_30);
  f29_normalize(Y3);
  acc.x = X3; acc.y = Y3;
  acc.zz = f29_mul(PP, acc.zz);
  acc.zzz = f29_mul(PPP, acc.zzz);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(int iters, uint64_t* out, uint32_t* sink) {
  uint32_t s = 0;
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    XYZZ<FpParams> acc; Affine<FpParams> b;
    for (int i = 0; i < 8; ++i) {
      acc.x.v[i] = threadIdx.x * 2654435761u + i; acc.y.v[i] = blockIdx.x * 40503u + i * 7 + 1;
      acc.zz.v[i] = threadIdx.x * 77u + i; acc.zzz.v[i] = blockIdx.x * 99u + i * 3;
      b.x.v[i] = threadIdx.x * 13u + i * 5 + 1; b.y.v[i] = threadIdx.x * 17u + i * 11 + 3;
    }
    acc.x.v[7] &= 0x3fffffffu; acc.y.v[7] &= 0x3fffffffu; acc.zz.v[7] &= 0x3fffffffu; acc.zzz.v[7] &= 0x3fffffffu;
    b.x.v[7] &= 0x3fffffffu; b.y.v[7] &= 0x3fffffffu;
    for (int k = 0; k < iters; ++k) {
      const Fe<FpParams> U2 = fe_mul_lazy(b.x, acc.zz), S2 = fe_mul_lazy(b.y, acc.zzz);
      const Fe<FpParams> Pp = fe_sub_lazy(U2, acc.x), Rr = fe_sub_lazy(S2, acc.y);
      const Fe<FpParams> PP = fe_mul_lazy(Pp, Pp), PPP = fe_mul_lazy(Pp, PP), Qq = fe_mul_lazy(acc.x, PP);
      const Fe<FpParams> X3 = fe_sub_lazy(fe_sub_lazy(fe_sub_lazy(fe_mul_lazy(Rr, Rr), PPP), Qq), Qq);
      const Fe<FpParams> Y3 = fe_sub_lazy(fe_mul_lazy(Rr, fe_sub_lazy(Qq, X3)), fe_mul_lazy(acc.y, PPP));
      acc.x = X3; acc.y = Y3; acc.zz = fe_mul_lazy(acc.zz, PP); acc.zzz = fe_mul_lazy(acc.zzz, PPP);
      b.x.v[0] += 2u;
    }
    s = acc.x.v[0] ^ acc.y.v[1] ^ acc.zz.v[2] ^ acc.zzz.v[3];
  } else {
    XYZZ29 acc; F29 bx, by;
    for (int i = 0; i < 9; ++i) {
      acc.x.v[i] = (threadIdx.x * 2654435761u + i) & MASK29; acc.y.v[i] = (blockIdx.x * 40503u + i * 7 + 1) & MASK29;
      acc.zz.v[i] = (threadIdx.x * 77u + i) & MASK29; acc.zzz.v[i] = (blockIdx.x * 99u + i * 3) & MASK29;
      bx.v[i] = (threadIdx.x * 13u + i * 5 + 1) & MASK29; by.v[i] = (threadIdx.x * 17u + i * 11 + 3) & MASK29;
    }
    acc.x.v[8] &= 0x3fffffu; acc.y.v[8] &= 0x3fffffu; acc.zz.v[8] &= 0x3fffffu; acc.zzz.v[8] &= 0x3fffffu; bx.v[8] &= 0x3fffffu; by.v[8] &= 0x3fffffu;
    for (int k = 0; k < iters; ++k) {
      madd29(acc, bx, by);
      bx.v[0] = (bx.v[0] + 2u) & MASK29;
    }
    s = acc.x.v[0] ^ acc.y.v[1] ^ acc.zz.v[2] ^ acc.zzz.v[3];
  }
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) out[((size_t)blockIdx.x * 256 + threadIdx.x) / 64] = r1 - r0;
  if (s == 0x12345678u) sink[0] = s;
}

template <int MODE> int run(const char* what) {
  const int iters = 400;
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
    printf("%-46s %d waves/SIMD: %.2f us per addition per SIMD slot (kernel wall time %.3f ms)\n", what, wg, ms * 1e3 / iters / wg, ms);
    CK(hipFree(d)); CK(hipFree(s));
  }
  return 0;
}

int main() {
  if (run<0>("A  8x32 mixed addition (10 M + 7 S, shipped)")) return 1;
  if (run<1>("B  9x29 mixed addition (8 M + 2 S' + 4 norm)")) return 1;
  return 0;
}
