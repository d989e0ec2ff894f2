// Diagnostic (not product code): where does the workgroup dispatcher place the workgroups of a grid that
// exactly fills the register-limited slots (3 x 256-thread workgroups per CU at ~168 VGPRs)?  Records XCC / SE /
// CU and start / end times (100 MHz clock) per workgroup while each runs the same fixed amount of ALU work.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vdf_amd/csrc tools/ubench/dispatch_probe.hip -o tools/ubench/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#include "fe.cuh"
using namespace vdf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct Rec { uint32_t hw_id, xcc_id; uint64_t t0, t1; };

// VREGS pads the kernel's VGPR allocation (an array kept live) so that occupancy is register-limited
template <int VREGS>
__global__ __launch_bounds__(256) void k_work(int iters, Rec* out, uint32_t* sink) {
  uint32_t pad[VREGS + 1];
#pragma unroll
  for (int i = 0; i < VREGS; ++i) pad[i] = threadIdx.x * (i + 1);
  Fe<FpParams> x, y;
  for (int i = 0; i < 8; ++i) { x.v[i] = threadIdx.x * 2654435761u + i; y.v[i] = blockIdx.x * 40503u + i * 7 + 1; }
  x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < iters; ++k) {
    y = fe_mul_lazy(y, x);
#pragma unroll
    for (int i = 0; i < VREGS; ++i) asm volatile("" : "+v"(pad[i]));     // keep the padding registers allocated
  }
  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = Rec{hw, xcc, t0, t1};
  }
  uint32_t acc = y.v[0];
#pragma unroll
  for (int i = 0; i < VREGS; ++i) acc ^= pad[i];
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int VREGS> int run(int blocks) {
  Rec* d; uint32_t* s;
  CK(hipMalloc(&d, blocks * sizeof(Rec))); CK(hipMalloc(&s, 4));
  hipLaunchKernelGGL(k_work<VREGS>, dim3(blocks), dim3(256), 0, 0, 2000, d, s);
  hipLaunchKernelGGL(k_work<VREGS>, dim3(blocks), dim3(256), 0, 0, 2000, d, s);
  CK(hipDeviceSynchronize());
  std::vector<Rec> h(blocks);
  CK(hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
  uint64_t tmin = ~0ull, tmax = 0;
  std::map<uint32_t, int> per_cu;      // key: xcc | se | cu
  for (auto& r : h) {
    tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1);
    const uint32_t cu = (r.hw_id >> 8) & 0xf, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 0xf;
    per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  double avg = 0; uint64_t late = 0;
  for (auto& r : h) { avg += (double)(r.t1 - r.t0); if (r.t0 - tmin > 200) ++late; }
  printf("VGPR pad %3d, %4d workgroups: %zu distinct CUs; workgroups per CU histogram:", VREGS, blocks, per_cu.size());
  for (auto& kv : hist) printf("  %d WG x %d CUs", kv.first, kv.second);
  printf("\n    kernel span %.1f us, mean workgroup run time %.1f us, workgroups that started > 2 us after the first: %llu\n",
         (tmax - tmin) / 100.0, avg / blocks / 100.0, (unsigned long long)late);
  CK(hipFree(d)); CK(hipFree(s));
  return 0;
}

int main() {
  for (int blocks : {256, 512, 768, 769, 1024}) if (run<1>(blocks)) return 1;
  for (int blocks : {256, 512, 768, 769, 1024}) if (run<120>(blocks)) return 1;
  return 0;
}
