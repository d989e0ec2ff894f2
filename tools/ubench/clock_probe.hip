// Diagnostic (not product code): effective shader clock while every SIMD runs the Montgomery multiply at
// 1, 2 and 3 waves per SIMD, and the multiply's cost in shader cycles per wave.  s_memtime counts shader
// clocks, s_memrealtime the constant 100 MHz reference.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vdf_amd/csrc tools/ubench/clock_probe.hip -o tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fe.cuh"
using namespace vdf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_probe(int iters, uint64_t* out, uint32_t* sink) {
  Fe<FpParams> x, y;
  for (int i = 0; i < 8; ++i) { x.v[i] = threadIdx.x * 2654435761u + i; y.v[i] = blockIdx.x * 40503u + i * 7 + 1; }
  x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < iters; k += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) y = fe_mul_lazy(y, x);      // UNROLL x ~1.5 KB of straight-line code per trip
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64;
    out[2 * w] = c1 - c0; out[2 * w + 1] = r1 - r0;
  }
  if (y.v[0] == 0x12345678u) sink[0] = y.v[1];
}

template <int UNROLL> int run() {
  const int iters = 3840;
  printf("-- loop body = %d multiplications (~%.1f KB of code)\n", UNROLL, UNROLL * 1.5);
  for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {
    const int blocks = 256 * wg_per_cu, waves = blocks * 4;
    uint64_t* d; uint32_t* s;
    CK(hipMalloc(&d, waves * 16)); CK(hipMalloc(&s, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe<UNROLL>, dim3(blocks), dim3(256), 0, 0, UNROLL, d, s);     // warm-up
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe<UNROLL>, dim3(blocks), dim3(256), 0, 0, iters, d, s);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(waves * 2);
    CK(hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost));
    double cyc = 0, ref = 0;
    for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; ref += h[2 * w + 1]; }
    const double mhz = cyc / ref * 100.0;
    printf("%d WG/CU (%d waves/SIMD): kernel %.3f ms, shader clock %.0f MHz, %.0f shader cycles per multiply per wave, "
           "%.1f ns per multiply per wave\n", wg_per_cu, wg_per_cu, ms, mhz, cyc / waves / iters, ms * 1e6 / iters);
    CK(hipFree(d)); CK(hipFree(s));
  }
  return 0;
}

int main() {
  if (run<1>()) return 1;
  if (run<4>()) return 1;
  if (run<12>()) return 1;
  if (run<24>()) return 1;
  if (run<48>()) return 1;
  return 0;
}
