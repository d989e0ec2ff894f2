// Diagnostic (not product code): sustained VALU throughput on gfx950 by instruction, from KERNEL WALL TIME
// (HIP events) -- per-wave cycle counters mislead because the arbiter favours the oldest wave and waves of one
// SIMD finish at staggered times.  8 independent chains per lane, 128 instructions per loop trip.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/op_rates.hip -o tools/ubench/op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define R8(OPSTR) OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)
#define MAD64(i) "v_mad_u64_u32 %[a" #i "], vcc, %[x], %[y], %[a" #i "]\n\t"
#define MAD64S(i) "v_mad_u64_u32 %[a" #i "], vcc, %[x], s10, %[a" #i "]\n\t"
#define ADDC(i) "v_addc_co_u32_e32 %[l" #i "], vcc, 0, %[l" #i "], vcc\n\t"
#define ADDCO(i) "v_add_co_u32_e32 %[l" #i "], vcc, %[x], %[l" #i "]\n\t"
#define ADD32(i) "v_add_u32_e32 %[l" #i "], %[x], %[l" #i "]\n\t"
#define MULLO(i) "v_mul_lo_u32 %[l" #i "], %[x], %[l" #i "]\n\t"
#define MULHI(i) "v_mul_hi_u32 %[l" #i "], %[x], %[l" #i "]\n\t"
#define MAD24(i) "v_mad_u32_u24 %[l" #i "], %[x], %[y], %[l" #i "]\n\t"
#define MULHI24(i) "v_mul_hi_u32_u24 %[l" #i "], %[x], %[l" #i "]\n\t"
#define ADD3(i) "v_add3_u32 %[l" #i "], %[x], %[y], %[l" #i "]\n\t"
#define LSHLADD(i) "v_lshl_add_u32 %[l" #i "], %[x], 3, %[l" #i "]\n\t"
#define ALIGN(i) "v_alignbit_b32 %[l" #i "], %[x], %[l" #i "], 7\n\t"
#define CNDMASK(i) "v_cndmask_b32_e32 %[l" #i "], %[x], %[l" #i "], vcc\n\t"
#define CNDMASK64(i) "v_cndmask_b32_e64 %[l" #i "], %[x], %[l" #i "], s[10:11]\n\t"
#define DPPMOV(i) "v_mov_b32_dpp %[l" #i "], %[l" #i "] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define SUBB(i) "v_subb_co_u32_e32 %[l" #i "], vcc, %[x], %[l" #i "], vcc\n\t"
#define CMPCND(i) "v_cmp_lt_u32_e32 vcc, %[y], %[l" #i "]\n\tv_cndmask_b32_e32 %[l" #i "], %[x], %[l" #i "], vcc\n\t"
#define CMPCND64(i) "v_cmp_lt_u32_e64 s[10:11], %[y], %[l" #i "]\n\tv_cndmask_b32_e64 %[l" #i "], %[x], %[l" #i "], s[10:11]\n\t"
#define CMPONLY(i) "v_cmp_lt_u32_e32 vcc, %[y], %[l" #i "]\n\t"
#define FMA64(i) "v_fma_f64 %[a" #i "], %[a" #i "], %[a" #i "], %[a" #i "]\n\t"
#define DOT4(i) "v_dot4_u32_u8 %[l" #i "], %[x], %[y], %[l" #i "]\n\t"
#define PKMUL(i) "v_pk_mul_lo_u16 %[l" #i "], %[x], %[l" #i "]\n\t"
#define MOV(i) "v_mov_b32_e32 %[l" #i "], %[x]\n\t"
#define MADADDC(i) MAD64(i) ADDC(i)

#define OPERANDS \
  : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), \
    [l0] "+v"(l[0]), [l1] "+v"(l[1]), [l2] "+v"(l[2]), [l3] "+v"(l[3]), [l4] "+v"(l[4]), [l5] "+v"(l[5]), [l6] "+v"(l[6]), [l7] "+v"(l[7]) \
  : [x] "v"(x), [y] "v"(y) : "vcc", "s10", "s11"

template <int OP>
__global__ __launch_bounds__(256) void k_op(int trips, uint32_t* sink) {
  uint64_t a[8]; uint32_t l[8];
  uint32_t x = threadIdx.x * 2654435761u + 12345u, y = blockIdx.x * 40503u + 999u;
  for (int i = 0; i < 8; ++i) { a[i] = (uint64_t)(x + i) * 0x9E3779B97F4A7C15ull; l[i] = y + i; }
  asm volatile("s_mov_b64 s[10:11], 0x5555\n\ts_mov_b64 vcc, 0x3333" ::: "s10", "s11", "vcc");
  for (int k = 0; k < trips; ++k) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep) {     // 16 x 8 = 128 instructions (256 for the mad+addc pair)
      if (OP == 0) asm volatile(R8(MAD64) OPERANDS);
      else if (OP == 1) asm volatile(R8(ADDC) OPERANDS);
      else if (OP == 2) asm volatile(R8(ADDCO) OPERANDS);
      else if (OP == 3) asm volatile(R8(ADD32) OPERANDS);
      else if (OP == 4) asm volatile(R8(MULLO) OPERANDS);
      else if (OP == 5) asm volatile(R8(MULHI) OPERANDS);
      else if (OP == 6) asm volatile(R8(MAD24) OPERANDS);
      else if (OP == 7) asm volatile(R8(MULHI24) OPERANDS);
      else if (OP == 8) asm volatile(R8(ADD3) OPERANDS);
      else if (OP == 9) asm volatile(R8(LSHLADD) OPERANDS);
      else if (OP == 10) asm volatile(R8(ALIGN) OPERANDS);
      else if (OP == 11) asm volatile(R8(CNDMASK) OPERANDS);
      else if (OP == 12) asm volatile(R8(FMA64) OPERANDS);
      else if (OP == 13) asm volatile(R8(DOT4) OPERANDS);
      else if (OP == 14) asm volatile(R8(PKMUL) OPERANDS);
      else if (OP == 15) asm volatile(R8(MADADDC) OPERANDS);
      else if (OP == 16) asm volatile(R8(MAD64S) OPERANDS);
      else if (OP == 17) asm volatile(R8(MOV) OPERANDS);
      else if (OP == 21) asm volatile(R8(CMPCND) OPERANDS);
      else if (OP == 22) asm volatile(R8(CMPCND64) OPERANDS);
      else if (OP == 23) asm volatile(R8(CMPONLY) OPERANDS);
      else if (OP == 18) asm volatile(R8(CNDMASK64) OPERANDS);
      else if (OP == 19) asm volatile(R8(DPPMOV) OPERANDS);
      else if (OP == 20) asm volatile(R8(SUBB) OPERANDS);
    }
  }
  uint32_t acc = 0;
  for (int i = 0; i < 8; ++i) acc ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32) ^ l[i];
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int OP> int run(const char* name, int per_trip) {
  uint32_t* s; CK(hipMalloc(&s, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-22s", name);
  for (int wg : {1, 2, 4, 8}) {
    const int trips = 4000 / wg;
    hipLaunchKernelGGL(k_op<OP>, dim3(256 * wg), dim3(256), 0, 0, trips, s);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_op<OP>, dim3(256 * wg), dim3(256), 0, 0, trips, s);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)trips * per_trip * wg;
    printf("  %dw: %5.2f", wg, ms * 1e-3 * 2.375e9 / inst_per_simd);       // cycles at the sustained 2.375 GHz
  }
  printf("   cycles / wave-instruction / SIMD\n");
  CK(hipFree(s));
  return 0;
}

int main() {
  if (run<17>("v_mov_b32", 128)) return 1;
  if (run<3>("v_add_u32", 128)) return 1;
  if (run<2>("v_add_co_u32", 128)) return 1;
  if (run<1>("v_addc_co_u32", 128)) return 1;
  if (run<8>("v_add3_u32", 128)) return 1;
  if (run<9>("v_lshl_add_u32", 128)) return 1;
  if (run<10>("v_alignbit_b32", 128)) return 1;
  if (run<11>("v_cndmask_b32 (vcc)", 128)) return 1;
  if (run<18>("v_cndmask_b32 (sgpr pair)", 128)) return 1;
  if (run<21>("v_cmp vcc + cndmask vcc", 256)) return 1;
  if (run<22>("v_cmp sgpr + cndmask sgpr", 256)) return 1;
  if (run<23>("v_cmp_lt_u32 vcc", 128)) return 1;
  if (run<19>("v_mov_b32_dpp quad_perm", 128)) return 1;
  if (run<20>("v_subb_co_u32", 128)) return 1;
  if (run<6>("v_mad_u32_u24", 128)) return 1;
  if (run<7>("v_mul_hi_u32_u24", 128)) return 1;
  if (run<4>("v_mul_lo_u32", 128)) return 1;
  if (run<5>("v_mul_hi_u32", 128)) return 1;
  if (run<0>("v_mad_u64_u32", 128)) return 1;
  if (run<16>("v_mad_u64_u32 (sgpr)", 128)) return 1;
  if (run<15>("mad_u64 + addc pair", 256)) return 1;
  if (run<13>("v_dot4_u32_u8", 128)) return 1;
  if (run<14>("v_pk_mul_lo_u16", 128)) return 1;
  if (run<12>("v_fma_f64", 128)) return 1;
  return 0;
}
