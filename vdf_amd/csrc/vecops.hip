// Field-vector kernels of the Nova folding step for gfx950 (all HBM-streaming, one element per lane,
// 2 x dwordx4 per 32-byte element; a wave touches 2 KiB of contiguous memory per operand).
//
// Replaces, on the prove_step path (/root/reference/src/nova/proof.rs:342-349 -> nova-snark 0.8.0):
//   k_axpy            RelaxedR1CSWitness::fold   W1 + r*W2, E1 + r*T            
//   k_cross_term      NIFS::prove / commit_T     AZ1*BZ2 + AZ2*BZ1 - u1*CZ2 - CZ1 (K5, a14)
//   k_spmv            R1CSShape::multiply_vec    Az, Bz, Cz (CSR, coefficient dictionary) (K4, a13)
//   k_minroot_witness InverseMinRootCircuit::synthesize / inverse_round witness values
//                     (/root/reference/src/nova/proof.rs:107-126, :162-189)          (K7, a1/a2)
#include "internal.h"
#include "fe.cuh"

namespace vdf {

static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

template <class P>
__global__ __launch_bounds__(256) void k_axpy(const char* __restrict__ a, const char* __restrict__ r,
                                              const char* __restrict__ b, size_t n, char* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Fe<P> rr = fe_load<P>(r);
  Fe<P> x = fe_load<P>(a + i * 32);
  Fe<P> y = fe_load<P>(b + i * 32);
  fe_store<P>(out + i * 32, fe_add(x, fe_mul(rr, y)));
}

template <class P>
__global__ __launch_bounds__(256) void k_cross_term(const char* __restrict__ az1, const char* __restrict__ bz1,
                                                    const char* __restrict__ cz1, const char* __restrict__ az2,
                                                    const char* __restrict__ bz2, const char* __restrict__ cz2,
                                                    const char* __restrict__ u1, size_t n, char* __restrict__ T) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Fe<P> u = fe_load<P>(u1);
  Fe<P> a1 = fe_load<P>(az1 + i * 32), b1 = fe_load<P>(bz1 + i * 32), c1 = fe_load<P>(cz1 + i * 32);
  Fe<P> a2 = fe_load<P>(az2 + i * 32), b2 = fe_load<P>(bz2 + i * 32), c2 = fe_load<P>(cz2 + i * 32);
  Fe<P> t = fe_add(fe_mul(a1, b2), fe_mul(a2, b1));
  t = fe_sub(t, fe_mul(u, c2));
  t = fe_sub(t, c1);
  fe_store<P>(T + i * 32, t);
}

// Round j (0-based) of the inverse walk starts from forward state t-j and lands on t-j-1:
//   new_x = x_{t-j-1}, tmp1 = x_{t-j}^2, tmp2 = tmp1^2, new_y = y_{t-j-1}
// (new_x = y - (i-1) and new_y = x^5 - new_x of src/nova/proof.rs:167-189 are exactly the previous
// forward state, src/minroot.rs:329-344).  Thread t writes final_i = i0.
template <class P>
__global__ __launch_bounds__(256) void k_minroot_witness(const char* __restrict__ trace, const char* __restrict__ i0,
                                                         uint64_t t, char* __restrict__ W) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > t) return;
  if (j == t) {
    fe_store<P>(W + 4 * t * 32, fe_load<P>(i0));
    return;
  }
  const Fe<P> x = fe_load<P>(trace + (t - j) * 64);
  const Fe<P> nx = fe_load<P>(trace + (t - j - 1) * 64);
  const Fe<P> ny = fe_load<P>(trace + (t - j - 1) * 64 + 32);
  const Fe<P> t1 = fe_sqr(x);
  const Fe<P> t2 = fe_sqr(t1);
  char* o = W + j * 128;
  fe_store<P>(o, nx);
  fe_store<P>(o + 32, t1);
  fe_store<P>(o + 64, t2);
  fe_store<P>(o + 96, ny);
}

// CSR sparse mat-vec; coefficient index 0 means +1 and 1 means -1 (no multiply), anything else
// indexes the dictionary.  One row per lane: R1CS rows are short (1-4 entries) and uniform.
template <class P>
__global__ __launch_bounds__(256) void k_spmv(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col,
                                              const uint32_t* __restrict__ coef, const char* __restrict__ dict,
                                              const char* __restrict__ z, size_t rows, char* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const uint32_t lo = rowptr[r], hi = rowptr[r + 1];
  Fe<P> acc = fe_zero<P>();
  for (uint32_t k = lo; k < hi; ++k) {
    const Fe<P> v = fe_load<P>(z + (size_t)col[k] * 32);
    const uint32_t ci = coef[k];
    if (ci == 0) acc = fe_add(acc, v);
    else if (ci == 1) acc = fe_sub(acc, v);
    else acc = fe_add(acc, fe_mul(v, fe_load<P>(dict + (size_t)ci * 32)));
  }
  fe_store<P>(out + r * 32, acc);
}

template <class P>
__global__ __launch_bounds__(256) void k_mul(const char* __restrict__ a, const char* __restrict__ b, size_t n,
                                             char* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  fe_store<P>(out + i * 32, fe_mul(fe_load<P>(a + i * 32), fe_load<P>(b + i * 32)));
}

template <class P, int DIR>
__global__ __launch_bounds__(256) void k_mont(const char* __restrict__ a, size_t n, char* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fe<P> x = fe_load<P>(a + i * 32);
  fe_store<P>(out + i * 32, DIR ? fe_to_mont(x) : fe_from_mont(x));
}

template <class P>
__global__ __launch_bounds__(256) void k_mul_chain(const char* __restrict__ a, size_t n, int iters,
                                                   char* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fe<P> x = fe_load<P>(a + i * 32);
  Fe<P> y = x;
  for (int k = 0; k < iters; ++k) y = fe_mul(y, x);
  fe_store<P>(out + i * 32, y);
}

#define FIELD_DISPATCH(field, KERNEL, ...)                                                       \
  do {                                                                                           \
    if ((field) == VDF_FIELD_FP) hipLaunchKernelGGL((KERNEL<FpParams>), __VA_ARGS__);            \
    else if ((field) == VDF_FIELD_FQ) hipLaunchKernelGGL((KERNEL<FqParams>), __VA_ARGS__);       \
    else return Status{VDF_ERR_BAD_ARG, "unknown field"};                                        \
    VDF_TRY_HIP(hipGetLastError());                                                              \
  } while (0)

#define C(p) reinterpret_cast<const char*>(p)
#define M(p) reinterpret_cast<char*>(p)

Status vec_axpy(int field, const void* a, const void* r, const void* b, size_t n, void* out, hipStream_t s) {
  if (n == 0) return Status{};
  FIELD_DISPATCH(field, k_axpy, grid_for(n), dim3(256), 0, s, C(a), C(r), C(b), n, M(out));
  return Status{};
}

Status vec_cross_term(int field, const void* az1, const void* bz1, const void* cz1, const void* az2, const void* bz2,
                      const void* cz2, const void* u1, size_t n, void* T, hipStream_t s) {
  if (n == 0) return Status{};
  FIELD_DISPATCH(field, k_cross_term, grid_for(n), dim3(256), 0, s, C(az1), C(bz1), C(cz1), C(az2), C(bz2), C(cz2),
                 C(u1), n, M(T));
  return Status{};
}

Status vec_minroot_witness(int field, const void* trace_xy, const void* i0, uint64_t t, void* W, hipStream_t s) {
  FIELD_DISPATCH(field, k_minroot_witness, grid_for(t + 1), dim3(256), 0, s, C(trace_xy), C(i0), t, M(W));
  return Status{};
}

Status vec_spmv(int field, const uint32_t* rowptr, const uint32_t* col, const uint32_t* coef, const void* dict,
                const void* z, size_t rows, void* out, hipStream_t s) {
  if (rows == 0) return Status{};
  FIELD_DISPATCH(field, k_spmv, grid_for(rows), dim3(256), 0, s, rowptr, col, coef, C(dict), C(z), rows, M(out));
  return Status{};
}

Status vec_mul(int field, const void* a, const void* b, size_t n, void* out, hipStream_t s) {
  if (n == 0) return Status{};
  FIELD_DISPATCH(field, k_mul, grid_for(n), dim3(256), 0, s, C(a), C(b), n, M(out));
  return Status{};
}

Status vec_to_mont(int field, const void* a, size_t n, void* out, hipStream_t s) {
  if (n == 0) return Status{};
  if (field == VDF_FIELD_FP) hipLaunchKernelGGL((k_mont<FpParams, 1>), grid_for(n), dim3(256), 0, s, C(a), n, M(out));
  else if (field == VDF_FIELD_FQ) hipLaunchKernelGGL((k_mont<FqParams, 1>), grid_for(n), dim3(256), 0, s, C(a), n, M(out));
  else return Status{VDF_ERR_BAD_ARG, "unknown field"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status vec_from_mont(int field, const void* a, size_t n, void* out, hipStream_t s) {
  if (n == 0) return Status{};
  if (field == VDF_FIELD_FP) hipLaunchKernelGGL((k_mont<FpParams, 0>), grid_for(n), dim3(256), 0, s, C(a), n, M(out));
  else if (field == VDF_FIELD_FQ) hipLaunchKernelGGL((k_mont<FqParams, 0>), grid_for(n), dim3(256), 0, s, C(a), n, M(out));
  else return Status{VDF_ERR_BAD_ARG, "unknown field"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status vec_mul_chain(int field, const void* a, size_t n, int iters, void* out, hipStream_t s) {
  if (n == 0) return Status{};
  FIELD_DISPATCH(field, k_mul_chain, grid_for(n), dim3(256), 0, s, C(a), n, iters, M(out));
  return Status{};
}

}  // namespace vdf
