// Diagnostic micro-benchmark (not product code): per-instruction issue rates of the
// integer / f64 VALU ops a 255-bit Montgomery multiply can be built from, on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates
// Prints wave-cycles per instruction for one wave per SIMD and for 4 waves per SIMD.
// SUPERSEDED by op_rates.hip: the per-wave cycle counters used here under-count at several waves per SIMD (the
// arbiter favours the oldest wave, so the waves of a SIMD finish at staggered times); op_rates.hip measures from
// kernel wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 8192;
constexpr int UNROLL = 16;   // independent chains per lane

template <int OP>
__global__ void k_rate(uint64_t* out, uint64_t* cycles, uint32_t seed) {
  uint32_t a = seed + threadIdx.x * 2654435761u, b = a * 40503u + 17u;
  uint64_t acc[UNROLL];
  double d[UNROLL];
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) { acc[j] = (uint64_t)(a + j) * 0x9E3779B97F4A7C15ull; d[j] = (double)(a + j) * 1.0000001; }
  double da = (double)a * 1e-9 + 1.0, db = (double)b * 1e-12;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      if (OP == 0) {          // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
      } else if (OP == 1) {   // v_mul_lo_u32
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else if (OP == 2) {   // v_mul_hi_u32
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else if (OP == 3) {   // v_fma_f64
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(da), "v"(db));
      } else if (OP == 4) {   // v_mad_u32_u24
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
        acc[j] = lo;
      } else if (OP == 5) {   // v_add_co_u32 + v_addc_co_u32 pair (64-bit add)
        uint32_t lo = (uint32_t)acc[j], hi = (uint32_t)(acc[j] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc[j] = ((uint64_t)hi << 32) | lo;
      } else if (OP == 6) {   // v_mul_hi_u32_u24
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else if (OP == 7) {   // v_add_u32 (full-rate reference)
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[j] = lo;
      } else if (OP == 8) {   // v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(da));
      } else if (OP == 9) {   // v_add_f64
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(db));
      } else if (OP == 10) {  // v_lshl_add_u64 (64-bit add in one op on gfx94x+)
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[j]) : "v"(acc[(j + 1) % UNROLL]));
      } else if (OP == 12) {  // mad + addc pair (the 96-bit column accumulate)
        uint32_t hi = (uint32_t)(acc[(j + 1) % UNROLL]);
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc[j]), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        d[j] = (double)hi;
      } else if (OP == 13) {  // v_addc_co_u32 alone (carry chain)
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(lo) : "v"(a) : "vcc");
        acc[j] = lo;
      } else if (OP == 14) {  // mad with SGPR carry-out (not vcc)
        asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "s20", "s21");
      } else if (OP == 15) {  // v_add_co_u32 alone
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(lo) : "v"(a) : "vcc");
        acc[j] = lo;
      } else if (OP == 16) {  // v_add3_u32
        uint32_t lo = (uint32_t)acc[j];
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
        acc[j] = lo;
      } else if (OP == 17) {  // v_mul_lo + v_mul_hi pair
        uint32_t lo = (uint32_t)acc[j], h2;
        asm volatile("v_mul_hi_u32 %1, %0, %2\n\tv_mul_lo_u32 %0, %0, %2" : "+v"(lo), "=&v"(h2) : "v"(a));
        acc[j] = lo + h2;
      } else if (OP == 11) {  // v_cvt_f64_u32 + back
        uint32_t lo = (uint32_t)acc[j];
        double t;
        asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(t) : "v"(lo));
        d[j] = t;
      }
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0; double ds = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) { s += acc[j]; ds += d[j]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (uint64_t)ds;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
int run(const char* name, int ops_per_inst) {
  // (a) one wave per SIMD: 256 CUs x 4 waves; (b) 4 waves per SIMD: 256 x 16 waves.
  for (int wpb : {4, 8, 16, 32}) {
    int blocks = 256, threads = wpb * 64;
    uint64_t *out, *cyc;
    CK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
    CK(hipMalloc(&cyc, sizeof(uint64_t) * blocks * wpb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<OP><<<blocks, threads>>>(out, cyc, 1u);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<OP><<<blocks, threads>>>(out, cyc, 2u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(blocks * wpb);
    CK(hipMemcpy(h.data(), cyc, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    double insts = (double)ITERS * UNROLL * ops_per_inst;
    // s_memtime ticks at 100 MHz-derived constant clock on some parts; also report wall-derived rate.
    double wave_insts_total = insts * blocks * wpb;
    double ginst = wave_insts_total / (ms * 1e-3) / 1e9;   // wave-instructions per ns across chip
    printf("%-22s waves/SIMD=%d  memtime-ticks/inst=%7.3f  wall=%8.3f ms  chip wave-inst/ns=%8.3f  => cyc/inst/SIMD@2.4GHz=%6.2f\n",
           name, wpb / 4, avg / insts, ms, ginst, 1024.0 * 2.4 / ginst);
    CK(hipFree(out)); CK(hipFree(cyc));
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  run<7>("v_add_u32", 1);
  run<0>("v_mad_u64_u32", 1);
  run<14>("v_mad_u64_u32 sgpr-cy", 1);
  run<12>("mad+addc pair", 2);
  run<13>("v_addc_co_u32", 1);
  run<15>("v_add_co_u32", 1);
  run<16>("v_add3_u32", 1);
  run<17>("mul_hi+mul_lo", 2);
  run<1>("v_mul_lo_u32", 1);
  run<2>("v_mul_hi_u32", 1);
  run<4>("v_mad_u32_u24", 1);
  run<6>("v_mul_hi_u32_u24", 1);
  run<5>("v_add_co+v_addc_co", 2);
  run<10>("v_lshl_add_u64", 1);
  run<3>("v_fma_f64", 1);
  run<8>("v_mul_f64", 1);
  run<9>("v_add_f64", 1);
  run<11>("v_cvt_f64_u32", 1);
  return 0;
}
