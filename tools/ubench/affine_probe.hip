// Diagnostic (not product code): what a bucket addition costs in AFFINE coordinates with batched inversion, against the
// XYZZ mixed addition k_accumulate uses -- the measurement behind DESIGN.md 4.2 "Batched-affine accumulation".
//   MODE 0     XYZZ mixed addition, one accumulator per lane (10 multiplications, the body of k_accumulate)
//   MODE B > 0 B independent affine accumulators per lane; a round adds one point to each: B denominators, prefix
//              products (Montgomery's trick: 3 multiplications per element), ONE Fermat inversion per lane, B slopes
//              (3 multiplications per addition).  6 + I / B multiplications per addition, I = the inversion.
// Sharing the inversion across the 64 lanes of a wavefront does not change the count that matters: the wavefront issues
// one inversion's worth of instructions per round either way.  All state in registers; no memory traffic.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vdf_amd/csrc tools/ubench/affine_probe.hip -o affine_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "ec.cuh"
using namespace vdf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
using F = Fe<FpParams>;

__device__ __forceinline__ F seed_fe(uint32_t a, uint32_t b) {
  F r;
  for (int i = 0; i < 8; ++i) r.v[i] = a * 2654435761u + b * 40503u + i * 0x9E3779B9u;
  r.v[7] &= 0x3fffffffu;
  return r;
}

template <int B>
__global__ __launch_bounds__(256) void k_affine(int rounds, uint32_t* sink) {
  F ax[B], ay[B], pre[B];
  for (int j = 0; j < B; ++j) { ax[j] = seed_fe(threadIdx.x + 1, j); ay[j] = seed_fe(blockIdx.x + 7, j + 100); }
  F px = seed_fe(threadIdx.x + 3, 999), py = seed_fe(threadIdx.x + 5, 998);
  for (int r = 0; r < rounds; ++r) {
    // forward: denominators x2 - x1 and their prefix products (the incoming point differs per element by a cheap tweak)
    F run = fe_one<FpParams>();
#pragma unroll
    for (int j = 0; j < B; ++j) {
      F x2 = px; x2.v[0] += (uint32_t)j;
      pre[j] = run;
      run = fe_mul(run, fe_sub(x2, ax[j]));
    }
    F inv = fe_inv(run);                       // a^(m-2): 255 squarings + the multiplications of the exponent's set bits
    // backward: 1 / d_j, then the chord formulas
#pragma unroll
    for (int j = B - 1; j >= 0; --j) {
      F x2 = px; x2.v[0] += (uint32_t)j;
      const F d = fe_sub(x2, ax[j]);
      const F dinv = fe_mul(inv, pre[j]);
      inv = fe_mul(inv, d);
      const F lam = fe_mul(fe_sub(py, ay[j]), dinv);
      const F x3 = fe_sub(fe_sub(fe_sqr(lam), ax[j]), x2);
      ay[j] = fe_sub(fe_mul(lam, fe_sub(ax[j], x3)), ay[j]);
      ax[j] = x3;
    }
    px.v[0] += 17u;
  }
  uint32_t acc = 0;
  for (int j = 0; j < B; ++j) acc ^= ax[j].v[0] ^ ay[j].v[3];
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_xyzz(int iters, uint32_t* sink) {
  XYZZ<FpParams> acc;
  Affine<FpParams> b;
  acc.x = seed_fe(threadIdx.x, 1); acc.y = seed_fe(blockIdx.x, 2); acc.zz = seed_fe(threadIdx.x, 3); acc.zzz = seed_fe(blockIdx.x, 4);
  b.x = seed_fe(threadIdx.x, 5); b.y = seed_fe(threadIdx.x, 6);
  for (int k = 0; k < iters; ++k) {
    const F U2 = fe_mul_lazy(b.x, acc.zz), S2 = fe_mul_lazy(b.y, acc.zzz);
    const F Pp = fe_sub_lazy(U2, acc.x), Rr = fe_sub_lazy(S2, acc.y);
    const F PP = fe_mul_lazy(Pp, Pp), PPP = fe_mul_lazy(Pp, PP), Qq = fe_mul_lazy(acc.x, PP);
    const F X3 = fe_sub_lazy(fe_sub_lazy(fe_sub_lazy(fe_mul_lazy(Rr, Rr), PPP), Qq), Qq);
    const F Y3 = fe_sub_lazy(fe_mul_lazy(Rr, fe_sub_lazy(Qq, X3)), fe_mul_lazy(acc.y, PPP));
    acc.x = X3; acc.y = Y3; acc.zz = fe_mul_lazy(acc.zz, PP); acc.zzz = fe_mul_lazy(acc.zzz, PPP);
    b.x.v[0] += 2u;
  }
  if (acc.x.v[0] == 0x12345678u) sink[0] = acc.y.v[1];
}

template <int B> int run_affine(uint32_t* s, double xyzz_ns) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 64;
  hipLaunchKernelGGL(k_affine<B>, dim3(512), dim3(256), 0, 0, 2, s);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_affine<B>, dim3(512), dim3(256), 0, 0, rounds, s);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ns = ms * 1e6 / ((double)rounds * B);
  printf("affine, batch %2d per lane: %8.1f ns per addition per wavefront slot = %5.2f x the XYZZ mixed addition\n", B, ns, ns / xyzz_ns);
  return 0;
}

int main() {
  uint32_t* s; CK(hipMalloc(&s, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2048;
  hipLaunchKernelGGL(k_xyzz, dim3(512), dim3(256), 0, 0, 16, s);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_xyzz, dim3(512), dim3(256), 0, 0, iters, s);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double xyzz_ns = ms * 1e6 / iters;
  printf("grid: 512 workgroups x 256 lanes (2 wavefronts per SIMD), everything in registers\n");
  printf("XYZZ mixed addition       : %8.1f ns per addition per wavefront slot (10 multiplications)\n", xyzz_ns);
  if (run_affine<2>(s, xyzz_ns)) return 1;
  if (run_affine<4>(s, xyzz_ns)) return 1;
  if (run_affine<8>(s, xyzz_ns)) return 1;
  if (run_affine<12>(s, xyzz_ns)) return 1;
  printf("model: 6 + I / B multiplications per affine addition against 10; break-even B = I / 4\n");
  return 0;
}
