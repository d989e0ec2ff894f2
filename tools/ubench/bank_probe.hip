// Diagnostic (not product code): does the VGPR bank of v_mad_u64_u32 / v_addc operands change the issue rate on
// gfx950?  Four independent mad+addc chains (the Montgomery column pattern) with operands placed by hand.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/bank_probe.hip -o tools/ubench/bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// one column-like chain: acc pair ACC, carry HI, sources A and B
#define CHAIN(ACC, HI, A, B) \
  "v_mad_u64_u32 " ACC ", vcc, " A ", " B ", " ACC "\n\t" \
  "v_addc_co_u32_e32 " HI ", vcc, 0, " HI ", vcc\n\t"

template <int VARIANT>
__global__ __launch_bounds__(256) void k_bank(int iters, uint64_t* out) {
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < iters; ++k) {
#pragma unroll
   for (int rep = 0; rep < 16; ++rep) {          // 128 instructions per trip: the loop branch must not dominate
    if (VARIANT == 0) {   // serial chain as in the multiply: acc bank 0/1, a in bank 2, b in bank 3 (no conflict)
      asm volatile(CHAIN("v[20:21]", "v25", "v10", "v15") CHAIN("v[20:21]", "v25", "v14", "v11")
                   CHAIN("v[20:21]", "v25", "v18", "v19") CHAIN("v[20:21]", "v25", "v22", "v23")
                   ::: "vcc", "v20", "v21", "v25");
    } else if (VARIANT == 1) {   // serial chain, a and b and acc.lo all in bank 0
      asm volatile(CHAIN("v[20:21]", "v25", "v8", "v12") CHAIN("v[20:21]", "v25", "v16", "v24")
                   CHAIN("v[20:21]", "v25", "v28", "v32") CHAIN("v[20:21]", "v25", "v36", "v40")
                   ::: "vcc", "v20", "v21", "v25");
    } else if (VARIANT == 2) {   // serial chain, a in bank 0 (= acc.lo), b in bank 2
      asm volatile(CHAIN("v[20:21]", "v25", "v8", "v14") CHAIN("v[20:21]", "v25", "v16", "v18")
                   CHAIN("v[20:21]", "v25", "v28", "v22") CHAIN("v[20:21]", "v25", "v36", "v26")
                   ::: "vcc", "v20", "v21", "v25");
    } else if (VARIANT == 3) {   // serial chain, a and b both in bank 2 (acc in 0/1)
      asm volatile(CHAIN("v[20:21]", "v25", "v10", "v14") CHAIN("v[20:21]", "v25", "v18", "v22")
                   CHAIN("v[20:21]", "v25", "v26", "v30") CHAIN("v[20:21]", "v25", "v34", "v38")
                   ::: "vcc", "v20", "v21", "v25");
    } else if (VARIANT == 5) {   // TWO independent chains interleaved (two multiplications side by side)
      asm volatile(CHAIN("v[20:21]", "v25", "v10", "v15") CHAIN("v[30:31]", "v35", "v14", "v11")
                   CHAIN("v[20:21]", "v25", "v18", "v19") CHAIN("v[30:31]", "v35", "v22", "v23")
                   ::: "vcc", "v20", "v21", "v25", "v30", "v31", "v35");
    } else if (VARIANT == 6) {   // FOUR independent chains interleaved
      asm volatile(CHAIN("v[20:21]", "v25", "v10", "v15") CHAIN("v[30:31]", "v35", "v14", "v11")
                   CHAIN("v[40:41]", "v45", "v18", "v19") CHAIN("v[50:51]", "v55", "v22", "v23")
                   ::: "vcc", "v20", "v21", "v25", "v30", "v31", "v35", "v40", "v41", "v45", "v50", "v51", "v55");
    } else if (VARIANT == 7) {   // two chains, mads grouped then addcs cannot be (vcc): mad A, addc A, mad B, addc B = variant 5
      asm volatile("v_mad_u64_u32 v[20:21], s[20:21], v10, v15, v[20:21]\n\t"
                   "v_mad_u64_u32 v[30:31], s[22:23], v14, v11, v[30:31]\n\t"
                   "v_addc_co_u32_e64 v25, s[20:21], 0, v25, s[20:21]\n\t"
                   "v_addc_co_u32_e64 v35, s[22:23], 0, v35, s[22:23]\n\t"
                   "v_mad_u64_u32 v[20:21], s[20:21], v18, v19, v[20:21]\n\t"
                   "v_mad_u64_u32 v[30:31], s[22:23], v22, v23, v[30:31]\n\t"
                   "v_addc_co_u32_e64 v25, s[20:21], 0, v25, s[20:21]\n\t"
                   "v_addc_co_u32_e64 v35, s[22:23], 0, v35, s[22:23]\n\t"
                   ::: "v20", "v21", "v25", "v30", "v31", "v35", "s20", "s21", "s22", "s23");
    } else if (VARIANT == 4) {   // one source in an SGPR (the reduction terms)
      asm volatile(CHAIN("v[20:21]", "v25", "v10", "s10") CHAIN("v[20:21]", "v25", "v14", "s11")
                   CHAIN("v[20:21]", "v25", "v18", "s12") CHAIN("v[20:21]", "v25", "v22", "s13")
                   ::: "vcc", "v20", "v21", "v25");
    }
   }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[((size_t)blockIdx.x * 256 + threadIdx.x) / 64] = c1 - c0;
}

template <int V> int run(const char* what) {
  const int iters = 2000;
  for (int wg = 1; wg <= 4; ++wg) {
    const int blocks = 256 * wg, waves = blocks * 4;
    uint64_t* d; CK(hipMalloc(&d, waves * 8));
    hipLaunchKernelGGL(k_bank<V>, dim3(blocks), dim3(256), 0, 0, 10, d);
    hipLaunchKernelGGL(k_bank<V>, dim3(blocks), dim3(256), 0, 0, iters, d);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(waves);
    CK(hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost));
    double cyc = 0; for (auto v : h) cyc += v;
    printf("%-52s %d waves/SIMD: %.2f cycles per instruction per SIMD\n", what, wg, cyc / waves / iters / 128.0 / wg);
    CK(hipFree(d));
  }
  return 0;
}

int main() {
  if (run<0>("acc b0/1, a b2, b b3 (no conflict)")) return 1;
  if (run<4>("acc b0/1, a b2, b SGPR")) return 1;
  if (run<5>("two independent chains interleaved")) return 1;
  if (run<6>("four independent chains interleaved")) return 1;
  if (run<7>("two chains, carries in SGPR pairs (no vcc)")) return 1;
  return 0;
}
