// Diagnostic (not product code): cost of the lazy mixed addition (the body of k_accumulate) on register data,
// no memory traffic, at 1..3 workgroups per CU; compare with 10x the bare multiply of clock_probe.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vdf_amd/csrc tools/ubench/madd_probe.hip -o tools/ubench/madd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ec.cuh"
using namespace vdf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <class P>
__device__ __forceinline__ void madd_lazy(XYZZ<P>& acc, bool& have, const Affine<P>& b) {
  if (affine_is_identity(b)) return;
  if (!have) { acc = xyzz_from_affine(b); have = true; return; }
  const Fe<P> U2 = fe_mul_lazy(b.x, acc.zz);
  const Fe<P> S2 = fe_mul_lazy(b.y, acc.zzz);
  const Fe<P> Pp = fe_sub_lazy(U2, acc.x);
  const Fe<P> Rr = fe_sub_lazy(S2, acc.y);
  if (Pp.v[0] <= 2u && fe_is_zero(fe_canon(Pp))) {
    if (fe_is_zero(fe_canon(Rr))) acc = xyzz_dbl_affine(b);
    else have = false;
    return;
  }
  const Fe<P> PP = fe_mul_lazy(Pp, Pp);
  const Fe<P> PPP = fe_mul_lazy(Pp, PP);
  const Fe<P> Qq = fe_mul_lazy(acc.x, PP);
  const Fe<P> X3 = fe_sub_lazy(fe_sub_lazy(fe_sub_lazy(fe_mul_lazy(Rr, Rr), PPP), Qq), Qq);
  const Fe<P> Y3 = fe_sub_lazy(fe_mul_lazy(Rr, fe_sub_lazy(Qq, X3)), fe_mul_lazy(acc.y, PPP));
  acc.x = X3; acc.y = Y3;
  acc.zz = fe_mul_lazy(acc.zz, PP);
  acc.zzz = fe_mul_lazy(acc.zzz, PPP);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(int iters, uint64_t* out, uint32_t* sink) {
  XYZZ<FpParams> acc; Affine<FpParams> b;
  for (int i = 0; i < 8; ++i) {
    acc.x.v[i] = threadIdx.x * 2654435761u + i; acc.y.v[i] = blockIdx.x * 40503u + i * 7 + 1;
    acc.zz.v[i] = threadIdx.x * 77u + i; acc.zzz.v[i] = blockIdx.x * 99u + i * 3;
    b.x.v[i] = threadIdx.x * 13u + i * 5 + 1; b.y.v[i] = threadIdx.x * 17u + i * 11 + 3;
  }
  acc.x.v[7] &= 0x3fffffffu; acc.y.v[7] &= 0x3fffffffu; acc.zz.v[7] &= 0x3fffffffu; acc.zzz.v[7] &= 0x3fffffffu;
  b.x.v[7] &= 0x3fffffffu; b.y.v[7] &= 0x3fffffffu;
  bool have = true;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < iters; ++k) {
    if (MODE == 0) madd_lazy(acc, have, b);
    else {            // the same ten multiplications and seven subtractions without the exceptional-case branches
      const Fe<FpParams> U2 = fe_mul_lazy(b.x, acc.zz), S2 = fe_mul_lazy(b.y, acc.zzz);
      const Fe<FpParams> Pp = fe_sub_lazy(U2, acc.x), Rr = fe_sub_lazy(S2, acc.y);
      const Fe<FpParams> PP = fe_mul_lazy(Pp, Pp), PPP = fe_mul_lazy(Pp, PP), Qq = fe_mul_lazy(acc.x, PP);
      const Fe<FpParams> X3 = fe_sub_lazy(fe_sub_lazy(fe_sub_lazy(fe_mul_lazy(Rr, Rr), PPP), Qq), Qq);
      const Fe<FpParams> Y3 = fe_sub_lazy(fe_mul_lazy(Rr, fe_sub_lazy(Qq, X3)), fe_mul_lazy(acc.y, PPP));
      acc.x = X3; acc.y = Y3; acc.zz = fe_mul_lazy(acc.zz, PP); acc.zzz = fe_mul_lazy(acc.zzz, PPP);
    }
    b.x.v[0] += 2u;
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64;
    out[2 * w] = c1 - c0; out[2 * w + 1] = r1 - r0;
  }
  if (acc.x.v[0] == 0x12345678u && have) sink[0] = acc.y.v[1] + acc.zz.v[2] + acc.zzz.v[3];
}

template <int MODE> int run(const char* what) {
  const int iters = 400;
  for (int wg = 1; wg <= 3; ++wg) {
    const int blocks = 256 * wg, waves = blocks * 4;
    uint64_t* d; uint32_t* s;
    CK(hipMalloc(&d, waves * 16)); CK(hipMalloc(&s, 4));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, 400, d, s);      // warm-up (clock ramp)
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d, s);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(waves * 2);
    CK(hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost));
    double cyc = 0, ref = 0;
    for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; ref += h[2 * w + 1]; }
    printf("%-28s %d waves/SIMD: %.0f MHz, %.0f cycles per addition per wave = %.0f per SIMD slot, %.2f us per addition round\n",
           what, wg, cyc / ref * 100.0, cyc / waves / iters, cyc / waves / iters / wg, ref / waves / iters / 100.0);
    CK(hipFree(d)); CK(hipFree(s));
  }
  return 0;
}

// sustained load: the same kernel back to back for ~0.5 s, clock reported per launch
int sustained() {
  const int blocks = 768, waves = blocks * 4, iters = 2000;
  uint64_t* d; uint32_t* s;
  CK(hipMalloc(&d, waves * 16)); CK(hipMalloc(&s, 4));
  std::vector<uint64_t> h(waves * 2);
  for (int rep = 0; rep < 30; ++rep) {
    hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, 0, iters, d, s);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost));
    double cyc = 0, ref = 0;
    for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; ref += h[2 * w + 1]; }
    if (rep % 3 == 0) printf("sustained launch %2d: %.0f MHz, %.2f us per addition round (3 waves/SIMD)\n", rep, cyc / ref * 100.0, ref / waves / iters / 100.0);
  }
  return 0;
}

int main() {
  if (sustained()) return 1;
  if (run<0>("madd_lazy (with branches)")) return 1;
  if (run<1>("straight-line 10M + 7S")) return 1;
  return 0;
}
