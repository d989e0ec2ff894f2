// Field-vector kernels of the Nova folding step for gfx950 (all HBM-streaming, one element per lane,
// 2 x dwordx4 per 32-byte element; a wave touches 2 KiB of contiguous memory per operand).
//
// Replaces, on the prove_step path (/root/reference/src/nova/proof.rs:342-349 -> nova-snark 0.8.0):
//   k_axpy            RelaxedR1CSWitness::fold   W1 + r*W2, E1 + r*T            
//   k_cross_term      NIFS::prove / commit_T     AZ1*BZ2 + AZ2*BZ1 - u1*CZ2 - CZ1 (K5, a14)
//   k_spmv            R1CSShape::multiply_vec    Az, Bz, Cz (CSR, coefficient dictionary) (K4, a13)
//   k_minroot_witness InverseMinRootCircuit::synthesize / inverse_round witness values
//                     (/root/reference/src/nova/proof.rs:107-126, :162-189)          (K7, a1/a2)
#include <cstring>
#include "internal.h"
#include "fe.cuh"

namespace vdf {

static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

template <class P>
__global__ __launch_bounds__(256) void k_axpy(const char* __restrict__ a, const char* __restrict__ r,
                                              const char* __restrict__ b, size_t n, char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
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
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
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
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
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
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const uint32_t lo = rowptr[r], hi = rowptr[r + 1];
  if (hi - lo > VDF_LONG_ROW) return;                  // summed by a wavefront already (k_spmv_long)
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

// ---- fused NIFS step kernels: one launch per stage of a fold, small operands by value ---------------------
struct FeVal { uint32_t v[8]; };                      // a field element as a kernel argument
struct StepConsts { FeVal z_in[3], i0, u, X[6]; };    // everything of a fresh z that is not a round value
template <class P> __device__ __forceinline__ Fe<P> fe_from_val(const FeVal& a) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = a.v[i];
  return r;
}

// z = [ z_in(3) | per round new_x, tmp1, tmp2, new_y (4t) | final_i | u | X(6) ]: the whole fresh column vector
// of the exposed-IO step circuit; round values as in k_minroot_witness, the rest from kernel arguments.
// `packed` (optional): the same witness without the new_x values, [ z_in(3) | per round tmp1, tmp2, new_y (3t) |
// final_i ] -- new_x of a round is an affine image of another witness value, so a commitment needs no term for it
// (vdf_minroot_step_z_packed).
template <class P>
__global__ __launch_bounds__(256) void k_step_z(const char* __restrict__ trace, StepConsts k, uint64_t t,
                                                char* __restrict__ z, char* __restrict__ packed) {
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > t) return;
  if (j == t) {
    for (int i = 0; i < 3; ++i) fe_store<P>(z + i * 32, fe_from_val<P>(k.z_in[i]));
    char* tail = z + (3 + 4 * t) * 32;
    fe_store<P>(tail, fe_from_val<P>(k.i0));
    fe_store<P>(tail + 32, fe_from_val<P>(k.u));
    for (int i = 0; i < 6; ++i) fe_store<P>(tail + 64 + i * 32, fe_from_val<P>(k.X[i]));
    if (packed) {
      for (int i = 0; i < 3; ++i) fe_store<P>(packed + i * 32, fe_from_val<P>(k.z_in[i]));
      fe_store<P>(packed + (3 + 3 * t) * 32, fe_from_val<P>(k.i0));
    }
    return;
  }
  const Fe<P> x = fe_load<P>(trace + (t - j) * 64);
  const Fe<P> nx = fe_load<P>(trace + (t - j - 1) * 64);
  const Fe<P> ny = fe_load<P>(trace + (t - j - 1) * 64 + 32);
  const Fe<P> t1 = fe_sqr(x);
  const Fe<P> t2 = fe_sqr(t1);
  char* o = z + 96 + j * 128;
  fe_store<P>(o, nx);
  fe_store<P>(o + 32, t1);
  fe_store<P>(o + 64, t2);
  fe_store<P>(o + 96, ny);
  if (packed) {
    char* q = packed + 96 + j * 96;
    fe_store<P>(q, t1);
    fe_store<P>(q + 32, t2);
    fe_store<P>(q + 64, ny);
  }
}

// The variables InverseMinRootCircuit::synthesize allocates inside an augmented circuit (no z_in, u, X around them):
// per round [new_x,] tmp1, tmp2, new_y, then final_i.  per = 4 is the reference's allocation (src/nova/proof.rs:167-181),
// per = 3 the bound form without new_x (oracle/nova.py InverseMinRootCircuit).
template <class P>
__global__ __launch_bounds__(256) void k_step_segment(const char* __restrict__ trace, FeVal i0, uint64_t t, int per,
                                                      char* __restrict__ out, char* __restrict__ packed, FeVal i_in) {
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > t) return;
  if (j == t) {
    fe_store<P>(out + (size_t)per * t * 32, fe_from_val<P>(i0));
    if (packed) {                    // [3t] final_i, [3t + 1] y of the first round, [3t + 2] i of the first round, [3t + 3] one
      char* q = packed + (size_t)3 * t * 32;
      fe_store<P>(q, fe_from_val<P>(i0));
      fe_store<P>(q + 32, fe_load<P>(trace + t * 64 + 32));
      fe_store<P>(q + 64, fe_from_val<P>(i_in));
      fe_store<P>(q + 96, fe_one<P>());
    }
    return;
  }
  const Fe<P> x = fe_load<P>(trace + (t - j) * 64);
  const Fe<P> t1 = fe_sqr(x);
  const Fe<P> t2 = fe_sqr(t1);
  const Fe<P> ny = fe_load<P>(trace + (t - j - 1) * 64 + 32);
  char* o = out + j * (size_t)per * 32;
  if (per == 4) { fe_store<P>(o, fe_load<P>(trace + (t - j - 1) * 64)); o += 32; }
  fe_store<P>(o, t1);
  fe_store<P>(o + 32, t2);
  fe_store<P>(o + 64, ny);
  if (packed) {
    char* q = packed + j * 96;
    fe_store<P>(q, t1);
    fe_store<P>(q + 32, t2);
    fe_store<P>(q + 64, ny);
  }
}

struct Csr3 { const uint32_t* rowptr[3]; const uint32_t* col[3]; const uint32_t* coef[3]; };

// One term of a sparse row: +-z[col] or coefficient * z[col] (dictionary index 0 = +1, 1 = -1).
template <class P>
__device__ __forceinline__ Fe<P> spmv_term(const Fe<P>& v, uint32_t ci, const char* __restrict__ dict) {
  if (ci == 0) return v;
  if (ci == 1) return fe_neg(v);
  return fe_mul(v, fe_load<P>(dict + (size_t)ci * 32));
}

// Rows of more than VDF_LONG_ROW entries, one wavefront each: lanes stride over the entries, then a butterfly sum.
template <class P>
__global__ __launch_bounds__(256) void k_spmv_long(Csr3 m, const char* __restrict__ dict, const char* __restrict__ z,
                                                   const uint32_t* __restrict__ long_rows, size_t n_long, char* __restrict__ o0,
                                                   char* __restrict__ o1, char* __restrict__ o2) {
  __builtin_amdgcn_s_setprio(3);
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_long) return;                                          // whole wavefronts leave together
  const uint32_t lane = threadIdx.x & 63, packed = long_rows[w], k = packed >> 30, r = packed & 0x3FFFFFFFu;
  const uint32_t* rowptr = k == 0 ? m.rowptr[0] : (k == 1 ? m.rowptr[1] : m.rowptr[2]);
  const uint32_t* col = k == 0 ? m.col[0] : (k == 1 ? m.col[1] : m.col[2]);
  const uint32_t* coef = k == 0 ? m.coef[0] : (k == 1 ? m.coef[1] : m.coef[2]);
  char* out = k == 0 ? o0 : (k == 1 ? o1 : o2);
  const uint32_t lo = rowptr[r], hi = rowptr[r + 1];
  Fe<P> acc = fe_zero<P>();
  for (uint32_t e = lo + lane; e < hi; e += 64)
    acc = fe_add(acc, spmv_term<P>(fe_load<P>(z + (size_t)col[e] * 32), coef[e], dict));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Fe<P> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = __shfl_xor(acc.v[i], off, 64);
    acc = fe_add(acc, o);
  }
  if (lane == 0) fe_store<P>(out + (size_t)r * 32, acc);
}

// multiply_vec(z2) and the cross term in one pass over the rows:
//   (a2, b2, c2) = (A z2, B z2, C z2)[row];  T[row] = a1*b2 + a2*b1 - u1*c2 - c1      (u2 = 1)
// The loads are issued in three waves instead of nine dependent steps: all six row pointers, then the first column /
// coefficient of each matrix, then the three z values (R1CS rows are short: the first entry is usually the only one);
// the remaining entries of a row follow in a plain loop.
template <class P>
__global__ __launch_bounds__(256) void k_nifs_cross(Csr3 m, const char* __restrict__ dict, const char* __restrict__ z2,
                                                    const char* __restrict__ az1, const char* __restrict__ bz1,
                                                    const char* __restrict__ cz1, FeVal u1, size_t rows,
                                                    size_t skip_begin, size_t skip_len,
                                                    char* __restrict__ az2, char* __restrict__ bz2,
                                                    char* __restrict__ cz2, char* __restrict__ T) {
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
  size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  if (r >= skip_begin) r += skip_len;  // `rows` rows of the matrix, leaving out [skip_begin, skip_begin + skip_len)
  uint32_t lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { lo[k] = m.rowptr[k][r]; hi[k] = m.rowptr[k][r + 1]; }
  const Fe<P> a1 = fe_load<P>(az1 + r * 32), b1 = fe_load<P>(bz1 + r * 32), c1 = fe_load<P>(cz1 + r * 32);
  // Loads in three waves instead of one dependent round trip per entry: the row pointers; then column and coefficient of
  // the first PRE[k] entries of each matrix (a MinRoot row is 1 + 1 + at most 4 entries: src/nova/proof.rs:219-227); then
  // all their z values at once.  Entries past PRE[k] (rows of the augmented circuit, at most VDF_LONG_ROW) follow in a loop.
  constexpr int PRE[3] = {2, 2, 4};
  uint32_t cc[3][4], kk[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < PRE[k]; ++j) {
      const uint32_t e = lo[k] + j < hi[k] ? lo[k] + j : lo[k];       // (arrays are nnz + 1 long: lo is always readable)
      cc[k][j] = m.col[k][e];
      kk[k][j] = m.coef[k][e];
    }
  Fe<P> v[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < PRE[k]; ++j) v[k][j] = fe_load<P>(z2 + (size_t)(lo[k] + j < hi[k] ? cc[k][j] : 0u) * 32);
  Fe<P> acc[3];
  char* const outs[3] = {az2, bz2, cz2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (hi[k] - lo[k] > VDF_LONG_ROW) { acc[k] = fe_load<P>(outs[k] + r * 32); continue; }      // k_spmv_long ran first
    acc[k] = hi[k] > lo[k] ? spmv_term<P>(v[k][0], kk[k][0], dict) : fe_zero<P>();
#pragma unroll
    for (int j = 1; j < PRE[k]; ++j)
      if (lo[k] + j < hi[k]) acc[k] = fe_add(acc[k], spmv_term<P>(v[k][j], kk[k][j], dict));
    for (uint32_t e = lo[k] + PRE[k]; e < hi[k]; ++e)
      acc[k] = fe_add(acc[k], spmv_term<P>(fe_load<P>(z2 + (size_t)m.col[k][e] * 32), m.coef[k][e], dict));
  }
  fe_store<P>(az2 + r * 32, acc[0]);
  fe_store<P>(bz2 + r * 32, acc[1]);
  fe_store<P>(cz2 + r * 32, acc[2]);
  Fe<P> t = fe_add(fe_mul(a1, acc[1]), fe_mul(acc[0], b1));
  t = fe_sub(t, fe_mul(fe_from_val<P>(u1), acc[2]));
  t = fe_sub(t, c1);
  fe_store<P>(T + r * 32, t);
}

// The same with LPR lanes per row (4 or 8): lane l of a row's group takes the entries l, l + LPR, ... of each of the three
// sparse rows, so that all gathers of a row are in flight at once instead of one dependent round trip per entry, and a
// butterfly over the group's lanes (log2 LPR steps) adds the terms up; the group's first lane finishes the cross term.
// What it buys is LATENCY: the ~10^4 rows of an augmented circuit that a step waits for take 12 us instead of 26-48 (a lane
// that walks 8 entries of 3 matrices in turn pays 24 round trips); for the 2 x 10^5 uniform rows of the MinRoot rounds
// (1 + 1 + 4 entries) four lanes per row issue the four gathers of C together.
// (LEAVE_LONG: a row with a matrix of more than VDF_LONG_ROW entries is left to the wavefront k_nifs_cross_f gives it)
template <class P, int LPR, bool LEAVE_LONG>
__device__ __forceinline__ void nifs_cross_group(const Csr3& m, const char* __restrict__ dict, const char* __restrict__ z2,
                                                 const char* __restrict__ az1, const char* __restrict__ bz1,
                                                 const char* __restrict__ cz1, const FeVal& u1, size_t rows, size_t skip_begin,
                                                 size_t skip_len, char* __restrict__ az2, char* __restrict__ bz2,
                                                 char* __restrict__ cz2, char* __restrict__ T, uint32_t block) {
  constexpr uint32_t RPB = 256 / LPR;                                // rows per workgroup
  const uint32_t l = threadIdx.x & (LPR - 1);
  size_t r = (size_t)block * RPB + threadIdx.x / LPR;
  bool live = r < rows;
  if (!live) r = rows ? rows - 1 : 0;                                // keeps the group's lanes together for the butterfly
  if (r >= skip_begin) r += skip_len;
  uint32_t lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { lo[k] = m.rowptr[k][r]; hi[k] = m.rowptr[k][r + 1]; }
  if (LEAVE_LONG && (hi[0] - lo[0] > VDF_LONG_ROW || hi[1] - lo[1] > VDF_LONG_ROW || hi[2] - lo[2] > VDF_LONG_ROW)) {
    live = false;                                                    // (all lanes of the group agree: the row is theirs)
    hi[0] = lo[0]; hi[1] = lo[1]; hi[2] = lo[2];
  }
  Fe<P> a1, b1, c1;
  if (l == 0) { a1 = fe_load<P>(az1 + r * 32); b1 = fe_load<P>(bz1 + r * 32); c1 = fe_load<P>(cz1 + r * 32); }
  char* const outs[3] = {az2, bz2, cz2};
  Fe<P> acc[3], v[3];
  uint32_t cc[3], kk[3];
  bool has[3], longrow[3];
  // every lane's first entry of each matrix: column and coefficient of all three issued together, then the three gathers
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    longrow[k] = hi[k] - lo[k] > VDF_LONG_ROW;
    has[k] = !longrow[k] && lo[k] + l < hi[k];
    const uint32_t e = has[k] ? lo[k] + l : lo[k];                   // (arrays are nnz + 1 long: lo is always readable)
    cc[k] = m.col[k][e];
    kk[k] = m.coef[k][e];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = fe_load<P>(z2 + (size_t)(has[k] ? cc[k] : 0u) * 32);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    acc[k] = has[k] ? spmv_term<P>(v[k], kk[k], dict) : fe_zero<P>();
    if (longrow[k]) { if (l == 0) acc[k] = fe_load<P>(outs[k] + r * 32); continue; }       // k_spmv_long ran first
    for (uint32_t e = lo[k] + l + LPR; e < hi[k]; e += LPR)
      acc[k] = fe_add(acc[k], spmv_term<P>(fe_load<P>(z2 + (size_t)m.col[k][e] * 32), m.coef[k][e], dict));
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) {
      Fe<P> o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.v[i] = __shfl_xor(acc[k].v[i], off, 64);
      acc[k] = fe_add(acc[k], o);
    }
  }
  if (l != 0 || !live) return;
  fe_store<P>(az2 + r * 32, acc[0]);
  fe_store<P>(bz2 + r * 32, acc[1]);
  fe_store<P>(cz2 + r * 32, acc[2]);
  Fe<P> t = fe_add(fe_mul(a1, acc[1]), fe_mul(acc[0], b1));
  t = fe_sub(t, fe_mul(fe_from_val<P>(u1), acc[2]));
  t = fe_sub(t, c1);
  fe_store<P>(T + r * 32, t);
}

template <class P, int LPR>
__global__ __launch_bounds__(256) void k_nifs_cross_w(Csr3 m, const char* __restrict__ dict, const char* __restrict__ z2,
                                                      const char* __restrict__ az1, const char* __restrict__ bz1,
                                                      const char* __restrict__ cz1, FeVal u1, size_t rows,
                                                      size_t skip_begin, size_t skip_len,
                                                      char* __restrict__ az2, char* __restrict__ bz2,
                                                      char* __restrict__ cz2, char* __restrict__ T) {
  __builtin_amdgcn_s_setprio(3);
  nifs_cross_group<P, LPR, false>(m, dict, z2, az1, bz1, cz1, u1, rows, skip_begin, skip_len, az2, bz2, cz2, T, blockIdx.x);
}

// The cross term of an augmented circuit in ONE launch instead of k_spmv_long followed by k_nifs_cross_w: the first
// `long_wgs` workgroups give every row that has a matrix of more than VDF_LONG_ROW entries (the S-box rows of the
// circuit's hash: sums of a few hundred terms) a wavefront of its own, the others run the eight-lane groups over all rows
// and leave those rows alone.  A step waits for this kernel with the device otherwise idle on its queue, so what counts is
// its depth in dependent memory round trips, not its work: a long row is the list entry, the six row pointers, then per
// pass of 128 entries the column / coefficient indices of all three matrices, then their z and dictionary values -- every
// load of a pass is in flight before the first product (two passes cover the longest rows of the built-in circuits).
template <class P>
__global__ __launch_bounds__(256) void k_nifs_cross_f(Csr3 m, const char* __restrict__ dict, const char* __restrict__ z2,
                                                      const char* __restrict__ az1, const char* __restrict__ bz1,
                                                      const char* __restrict__ cz1, FeVal u1, size_t rows,
                                                      size_t skip_begin, size_t skip_len,
                                                      const uint32_t* __restrict__ long_rowlist, size_t n_long_rows,
                                                      uint32_t long_wgs, char* __restrict__ az2, char* __restrict__ bz2,
                                                      char* __restrict__ cz2, char* __restrict__ T) {
  __builtin_amdgcn_s_setprio(3);
  if (blockIdx.x >= long_wgs) {
    nifs_cross_group<P, 8, true>(m, dict, z2, az1, bz1, cz1, u1, rows, skip_begin, skip_len, az2, bz2, cz2, T,
                                 blockIdx.x - long_wgs);
    return;
  }
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_long_rows) return;                                     // whole wavefronts leave together
  const uint32_t lane = threadIdx.x & 63;
  const size_t r = long_rowlist[w];
  uint32_t lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { lo[k] = m.rowptr[k][r]; hi[k] = m.rowptr[k][r + 1]; }
  Fe<P> a1, b1, c1;
  if (lane == 0) { a1 = fe_load<P>(az1 + r * 32); b1 = fe_load<P>(bz1 + r * 32); c1 = fe_load<P>(cz1 + r * 32); }
  uint32_t longest = hi[0] - lo[0];
  if (hi[1] - lo[1] > longest) longest = hi[1] - lo[1];
  if (hi[2] - lo[2] > longest) longest = hi[2] - lo[2];
  Fe<P> acc[3] = {fe_zero<P>(), fe_zero<P>(), fe_zero<P>()};
  for (uint32_t base = 0; base < longest; base += 128) {            // (the same trip count for every lane of the wavefront)
    uint32_t cc[3][2], kk[3][2];
    bool has[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t e = lo[k] + base + lane + 64 * j;
        has[k][j] = e < hi[k];
        cc[k][j] = m.col[k][has[k][j] ? e : lo[k]];                   // (arrays are nnz + 1 long: lo is always readable)
        kk[k][j] = m.coef[k][has[k][j] ? e : lo[k]];
      }
    Fe<P> v[3][2], d[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[k][j] = fe_load<P>(z2 + (size_t)(has[k][j] ? cc[k][j] : 0u) * 32);
        d[k][j] = fe_load<P>(dict + (size_t)(has[k][j] ? kk[k][j] : 0u) * 32);
      }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!has[k][j]) continue;
        const Fe<P> term = kk[k][j] == 0 ? v[k][j] : (kk[k][j] == 1 ? fe_neg(v[k][j]) : fe_mul(v[k][j], d[k][j]));
        acc[k] = fe_add(acc[k], term);
      }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      Fe<P> o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.v[i] = __shfl_xor(acc[k].v[i], off, 64);
      acc[k] = fe_add(acc[k], o);
    }
  }
  if (lane != 0) return;
  fe_store<P>(az2 + r * 32, acc[0]);
  fe_store<P>(bz2 + r * 32, acc[1]);
  fe_store<P>(cz2 + r * 32, acc[2]);
  Fe<P> t = fe_add(fe_mul(a1, acc[1]), fe_mul(acc[0], b1));
  t = fe_sub(t, fe_mul(fe_from_val<P>(u1), acc[2]));
  t = fe_sub(t, c1);
  fe_store<P>(T + r * 32, t);
}

// The cross term over the rows of the built-in MinRoot step circuits WITHOUT the sparse matrices: the 3t + 1 constraints
// InverseMinRootCircuit::synthesize records (src/nova/proof.rs:107-133, :219-227) are a fixed stencil over the round's own
// variables, so A z2, B z2, C z2 of a row are copies of witness values (and one four-term sum): no row pointers, no
// column / coefficient loads, no dependent gathers.  With S = seg_begin (first round variable), the step circuit's input
// z_in = (x, y, i) in the three variables before S, `one` the constant's column, PER = 4 (the reference's rounds: new_x,
// tmp1, tmp2, new_y) or 3 (bound form: tmp1, tmp2, new_y), round j and y_j = new_y_(j-1) (y_0 = z_in.y):
//   row 3j     x_j * x_j   = tmp1_j             x_j = new_x_(j-1)  (PER = 4; x_0 = z_in.x)
//   row 3j + 1 tmp1 * tmp1 = tmp2                   = y_(j-1) - i + j * one  (PER = 3, j > 0)
//   row 3j + 2 tmp2 * x_j  = new_y_j + y_j - i + (j + 1) * one
//   row 3t     final_i * one = i - t * one
// One row per lane (every running / fresh vector is read and written as contiguous 32-byte elements, like k_fold_many);
// the witness values a row needs sit in the 128 (96) bytes of its round and the one before.  Exact for ANY z2: the
// constant's coefficient is multiplied by z2[one] (1 in a fresh instance).  The host checks the stencil against the
// shape's triples before it uses this kernel (libvdf_nova.so public_params); other circuits keep k_nifs_cross.
// A z2, B z2, C z2 of stencil row i (0 .. 3t) from the witness: the part both stencil kernels share
template <class P, int PER>
__device__ __forceinline__ void minroot_stencil_row(const char* __restrict__ z2, size_t S, size_t one_col, uint64_t t, uint64_t i,
                                                    Fe<P>& a2, Fe<P>& b2, Fe<P>& c2) {
  const Fe<P> onev = fe_load<P>(z2 + one_col * 32);
  const Fe<P> i_in = fe_load<P>(z2 + (S - 1) * 32);
  // k * z2[one]: the constant's value is ONE in every fresh instance (the same for all lanes: a uniform branch), and then
  // the small-integer conversion is all it takes -- no Montgomery product (fe.cuh fe_from_small: ~50 instructions)
  const bool unit = fe_eq(onev, fe_one<P>());
  auto times_one = [&](uint64_t k) -> Fe<P> {
    const Fe<P> km = k < (1u << 30) ? fe_from_small<P>((uint32_t)k) : fe_from_u64<P>(k);
    return unit ? km : fe_mul(km, onev);
  };
  // The three kinds of row differ in WHERE their values sit, not in what is done with them: the addresses are selected
  // first and every load of the row is issued before the first use (a wavefront holds all three kinds: as branches they
  // would be three serial rounds of loads).  pa / pb / pc = the sources of A z2, B z2 and the first term of C z2.
  const bool last = i == 3 * t;
  const uint64_t j = last ? t - 1 : i / 3;
  const uint32_t role = last ? 3u : (uint32_t)(i - 3 * j);
  const char* rd = z2 + (S + (size_t)PER * j) * 32;                // this round's variables
  const char* t1p = rd + (PER - 3) * 32;                            // tmp1, tmp2, new_y
  const char* yp = j ? rd - 32 : z2 + (S - 2) * 32;                 // y_j = new_y_(j-1): the element right before this round
  // x_j: a variable (PER = 4: new_x_(j-1), or z_in.x), or y_(j-1) - i + j * one (PER = 3, j > 0; y_(-1) does not exist)
  const char* px = PER == 4 ? (j ? rd - (size_t)PER * 32 : z2 + (S - 3) * 32)
                            : (j == 0 ? z2 + (S - 3) * 32 : (j > 1 ? rd - (size_t)PER * 32 - 32 : z2 + (S - 2) * 32));
  const char* pa = role == 0 ? px : (role == 1 ? t1p : (role == 2 ? t1p + 32 : z2 + (S + (size_t)PER * t) * 32));
  const char* pb = role == 1 ? t1p : (role == 3 ? z2 + one_col * 32 : px);
  const char* pc = role == 3 ? z2 + (S - 1) * 32 : t1p + role * 32;
  a2 = fe_load<P>(pa); b2 = fe_load<P>(pb); c2 = fe_load<P>(pc);
  const Fe<P> y = fe_load<P>(yp);
  if (PER == 3 && j > 0) {                                          // the loaded value is y_(j-1): turn it into x_j
    const Fe<P> xa = fe_add(fe_sub(role == 0 ? a2 : b2, i_in), times_one(j));
    if (role == 0) { a2 = xa; b2 = xa; } else if (role == 2) b2 = xa;
  }
  if (role == 2) c2 = fe_add(fe_sub(fe_add(c2, y), i_in), times_one(j + 1));
  if (role == 3) c2 = fe_sub(c2, times_one(t));
}

template <class P, int PER>
__global__ __launch_bounds__(256) void k_nifs_cross_minroot(const char* __restrict__ z2, size_t S, size_t one_col, uint64_t t,
                                                            size_t row0, const char* __restrict__ az1, const char* __restrict__ bz1,
                                                            const char* __restrict__ cz1, FeVal u1, char* __restrict__ az2,
                                                            char* __restrict__ bz2, char* __restrict__ cz2, char* __restrict__ T) {
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > 3 * t) return;
  const size_t r = row0 + i;
  const Fe<P> a1 = fe_load<P>(az1 + r * 32), b1 = fe_load<P>(bz1 + r * 32), c1 = fe_load<P>(cz1 + r * 32);
  Fe<P> a2, b2, c2;
  minroot_stencil_row<P, PER>(z2, S, one_col, t, i, a2, b2, c2);
  fe_store<P>(az2 + r * 32, a2);
  fe_store<P>(bz2 + r * 32, b2);
  fe_store<P>(cz2 + r * 32, c2);
  Fe<P> tt = fe_add(fe_mul(a1, b2), fe_mul(a2, b1));
  tt = fe_sub(tt, fe_mul(fe_from_val<P>(u1), c2));
  tt = fe_sub(tt, c1);
  fe_store<P>(T + r * 32, tt);
}

// y * r for a scalar r below 2^128 given as a PLAIN integer (four limbs), y and the result in Montgomery form: the fold
// challenges of NIFS are 128-bit (CHAL_BITS), so  y R * r  is a 383-bit integer whose residue is the Montgomery form of y r
// -- no Montgomery reduction at all.  With m = 2^254 + c (c = the low four limbs of the modulus, below 2^126):
//   P = y r = Ph 2^254 + Pl,  Pl < 2^254,  Ph <= 2^128   (y < m, r < 2^128),   P = Pl - c Ph  (mod m),   c Ph < 2^254 < m:
// both terms are canonical, one fe_sub finishes.  32 + 16 limb products instead of the 96 of a Montgomery multiplication.
template <class P>
__device__ __forceinline__ Fe<P> fe_mul_u128(const Fe<P>& y, const uint32_t r[4]) {
  uint32_t p[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t t = (uint64_t)y.v[j] * r[i] + p[i + j] + carry;        // <= 2^64 - 1: no overflow
      p[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    p[i + 8] = carry;
  }
  uint32_t ph[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) ph[k] = (p[7 + k] >> 30) | (p[8 + k] << 2);
  ph[4] = p[11] >> 30;                                                       // 0 or 1 (Ph <= 2^128)
  Fe<P> lo;
#pragma unroll
  for (int i = 0; i < 7; ++i) lo.v[i] = p[i];
  lo.v[7] = p[7] & 0x3FFFFFFFu;
  uint32_t d[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                              // c = MOD[0..3]
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint64_t t = (uint64_t)ph[j] * P::MOD[i] + d[i + j] + carry;
      d[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    d[i + 5] = carry;
  }
  Fe<P> hi;
#pragma unroll
  for (int i = 0; i < 8; ++i) hi.v[i] = d[i];                               // (d[8] = 0: c Ph < 2^254)
  return fe_sub(lo, hi);
}

// The stencil rows WITH the previous step's fold of those rows applied on the way: the rows' cross term needs
// the running A z, B z, C z folded with the previous fresh ones -- which this very kernel wrote a step ago into az2 / bz2 / cz2
// and is about to overwrite.  So each lane reads its row of the running vectors AND of the previous fresh ones, folds
// (X1 <- X1 + r X2prev; E1 <- E1 + r Tprev when e1 is given), stores the folded row, and goes on as k_nifs_cross_minroot does
// with the folded values.  One pass over the row's 7 (8) streams instead of a 5-vector k_fold_many over ALL rows in front of
// it on the step's longest dependent path; the fold of z and of the ~10^4 other rows runs elsewhere, off that path.
// U128: r is a plain integer below 2^128 (every NIFS challenge: fe_mul_u128, no Montgomery reduction), else Montgomery form.
template <class P, int PER, bool U128>
__global__ __launch_bounds__(256) void k_nifs_cross_minroot_fold(const char* __restrict__ z2, size_t S, size_t one_col, uint64_t t,
                                                                 size_t row0, FeVal rv, char* __restrict__ az1, char* __restrict__ bz1,
                                                                 char* __restrict__ cz1, char* __restrict__ e1, const char* tprev,
                                                                 FeVal u1, char* az2, char* bz2, char* cz2, char* T) {
  __builtin_amdgcn_s_setprio(3);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > 3 * t) return;
  const size_t r = row0 + i;
  Fe<P> a1 = fe_load<P>(az1 + r * 32), b1 = fe_load<P>(bz1 + r * 32), c1 = fe_load<P>(cz1 + r * 32);
  const Fe<P> ap = fe_load<P>(az2 + r * 32), bp = fe_load<P>(bz2 + r * 32), cp = fe_load<P>(cz2 + r * 32);
  Fe<P> a2, b2, c2;
  minroot_stencil_row<P, PER>(z2, S, one_col, t, i, a2, b2, c2);
  const uint32_t rr[4] = {rv.v[0], rv.v[1], rv.v[2], rv.v[3]};
  const Fe<P> rm = fe_from_val<P>(rv);
  auto times_r = [&](const Fe<P>& y) -> Fe<P> { return U128 ? fe_mul_u128<P>(y, rr) : fe_mul(rm, y); };
  if (e1) {
    const Fe<P> e = fe_load<P>(e1 + r * 32), tp = fe_load<P>(tprev + r * 32);
    fe_store<P>(e1 + r * 32, fe_add(e, times_r(tp)));
  }
  a1 = fe_add(a1, times_r(ap));
  b1 = fe_add(b1, times_r(bp));
  c1 = fe_add(c1, times_r(cp));
  fe_store<P>(az1 + r * 32, a1);
  fe_store<P>(bz1 + r * 32, b1);
  fe_store<P>(cz1 + r * 32, c1);
  fe_store<P>(az2 + r * 32, a2);
  fe_store<P>(bz2 + r * 32, b2);
  fe_store<P>(cz2 + r * 32, c2);
  Fe<P> tt = fe_add(fe_mul(a1, b2), fe_mul(a2, b1));
  tt = fe_sub(tt, fe_mul(fe_from_val<P>(u1), c2));
  tt = fe_sub(tt, c1);
  fe_store<P>(T + r * 32, tt);
}

// acc_k <- acc_k + r * add_k for up to 8 vectors in one launch (the witness fold W, E and the running Az, Bz, Cz)
struct FoldArgs { char* acc[8]; const char* add[8]; uint32_t blk_end[8]; uint64_t n[8]; int k; };
template <class P>
__global__ __launch_bounds__(256) void k_fold_many(FoldArgs a, FeVal rv) {
  __builtin_amdgcn_s_setprio(3);     // light kernel: do not starve behind a co-running k_accumulate
  int seg = 0;
  while (seg < a.k - 1 && blockIdx.x >= a.blk_end[seg]) ++seg;
  const uint32_t blk0 = seg ? a.blk_end[seg - 1] : 0u;
  const size_t i = (size_t)(blockIdx.x - blk0) * 256 + threadIdx.x;
  if (i >= a.n[seg]) return;
  const Fe<P> r = fe_from_val<P>(rv);
  const Fe<P> x = fe_load<P>(a.acc[seg] + i * 32);
  const Fe<P> y = fe_load<P>(a.add[seg] + i * 32);
  fe_store<P>(a.acc[seg] + i * 32, fe_add(x, fe_mul(r, y)));
}
// the same with r below 2^128, as a plain integer in rv.v[0..3] (every fold challenge: fe_mul_u128)
template <class P>
__global__ __launch_bounds__(256) void k_fold_many_u128(FoldArgs a, FeVal rv) {
  __builtin_amdgcn_s_setprio(3);
  int seg = 0;
  while (seg < a.k - 1 && blockIdx.x >= a.blk_end[seg]) ++seg;
  const uint32_t blk0 = seg ? a.blk_end[seg - 1] : 0u;
  const size_t i = (size_t)(blockIdx.x - blk0) * 256 + threadIdx.x;
  if (i >= a.n[seg]) return;
  const uint32_t r[4] = {rv.v[0], rv.v[1], rv.v[2], rv.v[3]};
  const Fe<P> x = fe_load<P>(a.acc[seg] + i * 32);
  const Fe<P> y = fe_load<P>(a.add[seg] + i * 32);
  fe_store<P>(a.acc[seg] + i * 32, fe_add(x, fe_mul_u128<P>(y, r)));
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

// Box fingerprint (bench.py): every SIMD runs `iters` dependent Montgomery products; s_memtime counts shader clocks,
// s_memrealtime the constant 100 MHz reference, so their ratio over all wavefronts is the clock the device sustained
// under the integer load the MSM puts on it.  out[0] += shader cycles, out[1] += reference ticks (one add per wavefront).
__global__ __launch_bounds__(256) void k_clock_probe(int iters, unsigned long long* __restrict__ out) {
  Fe<FpParams> x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x.v[i] = threadIdx.x * 2654435761u + i; y.v[i] = blockIdx.x * 40503u + i * 7 + 1; }
  x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < iters; ++k) y = fe_mul_lazy(y, x);
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, (unsigned long long)(c1 - c0));
    atomicAdd(out + 1, (unsigned long long)(r1 - r0));
  }
  if (y.v[0] == 0x12345678u && y.v[5] == 0x9abcdef0u) out[2] = y.v[1];          // keeps the chain alive
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
  KTimer kt(s, "k_axpy", 96.0 * n);
  FIELD_DISPATCH(field, k_axpy, grid_for(n), dim3(256), 0, s, C(a), C(r), C(b), n, M(out));
  return Status{};
}

Status vec_cross_term(int field, const void* az1, const void* bz1, const void* cz1, const void* az2, const void* bz2,
                      const void* cz2, const void* u1, size_t n, void* T, hipStream_t s) {
  if (n == 0) return Status{};
  KTimer kt(s, "k_cross_term", 224.0 * n);
  FIELD_DISPATCH(field, k_cross_term, grid_for(n), dim3(256), 0, s, C(az1), C(bz1), C(cz1), C(az2), C(bz2), C(cz2),
                 C(u1), n, M(T));
  return Status{};
}

Status vec_minroot_witness(int field, const void* trace_xy, const void* i0, uint64_t t, void* W, hipStream_t s) {
  KTimer kt(s, "k_minroot_witness", 192.0 * t);
  FIELD_DISPATCH(field, k_minroot_witness, grid_for(t + 1), dim3(256), 0, s, C(trace_xy), C(i0), t, M(W));
  return Status{};
}

Status vec_spmv_long(int field, const uint32_t* const rowptr[3], const uint32_t* const col[3], const uint32_t* const coef[3],
                     const void* dict, const void* z, const uint32_t* long_rows, size_t n_long, void* const out[3], hipStream_t s) {
  if (n_long == 0) return Status{};
  Csr3 m;
  for (int k = 0; k < 3; ++k) { m.rowptr[k] = rowptr[k]; m.col[k] = col[k]; m.coef[k] = coef[k]; }
  KTimer kt(s, "k_spmv_long", 0.0);                    // priced with the cross term that reads its results
  FIELD_DISPATCH(field, k_spmv_long, dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, s, m, C(dict), C(z), long_rows, n_long,
                 M(out[0]), M(out[1]), M(out[2]));
  return Status{};
}

Status vec_spmv(int field, const uint32_t* rowptr, const uint32_t* col, const uint32_t* coef, const void* dict,
                const void* z, size_t rows, void* out, hipStream_t s) {
  if (rows == 0) return Status{};
  FIELD_DISPATCH(field, k_spmv, grid_for(rows), dim3(256), 0, s, rowptr, col, coef, C(dict), C(z), rows, M(out));
  return Status{};
}

static FeVal to_val(const vdf_fe* p) { FeVal v; std::memcpy(v.v, p, 32); return v; }

Status vec_step_z(int field, const void* trace_xy, uint64_t t, const vdf_fe z_in[3], const vdf_fe* i0, const vdf_fe* u,
                  const vdf_fe X[6], void* z, void* packed, hipStream_t s) {
  StepConsts k;
  for (int i = 0; i < 3; ++i) k.z_in[i] = to_val(&z_in[i]);
  k.i0 = to_val(i0);
  k.u = to_val(u);
  for (int i = 0; i < 6; ++i) k.X[i] = to_val(&X[i]);
  FIELD_DISPATCH(field, k_step_z, grid_for(t + 1), dim3(256), 0, s, C(trace_xy), k, t, M(z), M(packed));
  return Status{};
}

Status vec_step_segment(int field, const void* trace_xy, uint64_t t, const vdf_fe* i0, int per, void* out, void* packed,
                        const vdf_fe* i_in, hipStream_t s) {
  KTimer kt(s, "k_step_segment", (64.0 + 32.0 * per) * t);    // trace (x, y) read + `per` variables written per round
  FIELD_DISPATCH(field, k_step_segment, grid_for(t + 1), dim3(256), 0, s, C(trace_xy), to_val(i0), t, per, M(out), M(packed),
                 to_val(i_in ? i_in : i0));
  return Status{};
}

// lanes per row: 8 for the ~10^4 rows a step waits for (latency), 4 for long runs of short rows (the MinRoot rounds: all
// gathers of a row at once), 1 = the lane-per-row kernel.  vdf_hip_tuning.nifs_lanes = 1 | 4 | 8 forces one (tuning / A-B measurements).
int nifs_cross_lanes(size_t rows) {
  const int forced = tuning().nifs_lanes;
  return forced == 1 || forced == 4 || forced == 8 ? forced : (rows <= (1u << 15) ? 8 : 1);
}

// With `long_rowlist` (the distinct rows that have a matrix of more than VDF_LONG_ROW entries) the launch sums those rows
// itself (k_nifs_cross_f; the caller checked nifs_cross_lanes(rows) == 8); without, vec_spmv_long ran before it.
Status vec_nifs_cross(int field, const uint32_t* const rowptr[3], const uint32_t* const col[3],
                      const uint32_t* const coef[3], const void* dict, const void* z2, const void* az1, const void* bz1,
                      const void* cz1, const vdf_fe* u1, size_t rows, size_t skip_begin, size_t skip_len,
                      const uint32_t* long_rowlist, size_t n_long_rows, void* az2, void* bz2, void* cz2, void* T,
                      double alg_bytes, hipStream_t s) {
  if (rows == 0) return Status{};
  Csr3 m;
  for (int k = 0; k < 3; ++k) { m.rowptr[k] = rowptr[k]; m.col[k] = col[k]; m.coef[k] = coef[k]; }
  const int lpr = nifs_cross_lanes(rows);
  if (long_rowlist && n_long_rows) {
    if (lpr != 8) return Status{VDF_ERR_BAD_ARG, "the fused cross term runs eight lanes per row"};
    KTimer kt(s, "k_nifs_cross_f", alg_bytes);
    const uint32_t long_wgs = (uint32_t)((n_long_rows + 3) / 4);
    const dim3 grid((unsigned)((rows + 31) / 32) + long_wgs);
    FIELD_DISPATCH(field, k_nifs_cross_f, grid, dim3(256), 0, s, m, C(dict), C(z2), C(az1), C(bz1), C(cz1), to_val(u1), rows,
                   skip_begin, skip_len, long_rowlist, n_long_rows, long_wgs, M(az2), M(bz2), M(cz2), M(T));
    return Status{};
  }
  KTimer kt(s, lpr == 1 ? "k_nifs_cross" : (lpr == 4 ? "k_nifs_cross_w4" : "k_nifs_cross_w8"), alg_bytes);
  if (lpr == 1) {
    FIELD_DISPATCH(field, k_nifs_cross, grid_for(rows), dim3(256), 0, s, m, C(dict), C(z2), C(az1), C(bz1), C(cz1),
                   to_val(u1), rows, skip_begin, skip_len, M(az2), M(bz2), M(cz2), M(T));
  } else if (lpr == 4) {
    const dim3 grid((unsigned)((rows + 63) / 64));
    if (field == VDF_FIELD_FP) hipLaunchKernelGGL((k_nifs_cross_w<FpParams, 4>), grid, dim3(256), 0, s, m, C(dict), C(z2), C(az1), C(bz1), C(cz1),
                                                  to_val(u1), rows, skip_begin, skip_len, M(az2), M(bz2), M(cz2), M(T));
    else if (field == VDF_FIELD_FQ) hipLaunchKernelGGL((k_nifs_cross_w<FqParams, 4>), grid, dim3(256), 0, s, m, C(dict), C(z2), C(az1), C(bz1), C(cz1),
                                                       to_val(u1), rows, skip_begin, skip_len, M(az2), M(bz2), M(cz2), M(T));
    else return Status{VDF_ERR_BAD_ARG, "unknown field"};
    VDF_TRY_HIP(hipGetLastError());
  } else {
    const dim3 grid((unsigned)((rows + 31) / 32));
    if (field == VDF_FIELD_FP) hipLaunchKernelGGL((k_nifs_cross_w<FpParams, 8>), grid, dim3(256), 0, s, m, C(dict), C(z2), C(az1), C(bz1), C(cz1),
                                                  to_val(u1), rows, skip_begin, skip_len, M(az2), M(bz2), M(cz2), M(T));
    else if (field == VDF_FIELD_FQ) hipLaunchKernelGGL((k_nifs_cross_w<FqParams, 8>), grid, dim3(256), 0, s, m, C(dict), C(z2), C(az1), C(bz1), C(cz1),
                                                       to_val(u1), rows, skip_begin, skip_len, M(az2), M(bz2), M(cz2), M(T));
    else return Status{VDF_ERR_BAD_ARG, "unknown field"};
    VDF_TRY_HIP(hipGetLastError());
  }
  return Status{};
}

Status vec_nifs_cross_minroot(int field, int per, uint64_t t, size_t seg_begin, size_t one_col, size_t row0, const void* z2,
                              const void* az1, const void* bz1, const void* cz1, const vdf_fe* u1, void* az2, void* bz2, void* cz2,
                              void* T, hipStream_t s) {
  if (per != 3 && per != 4) return Status{VDF_ERR_BAD_ARG, "variables per round must be 3 or 4"};
  if (t == 0 || seg_begin < 3) return Status{VDF_ERR_BAD_ARG, "bad segment"};
  const size_t rows = 3 * (size_t)t + 1;
  // algorithmic bytes: 3 running + 3 fresh + T elements per row, and the round's variables once (SURVEY 8d counts 224 B per
  // row for the cross term alone and the sparse products apart; here the products are the witness copies themselves)
  KTimer kt(s, "k_nifs_cross_minroot", (double)rows * 7 * 32 + (double)per * t * 32);
  const dim3 grid = grid_for(rows);
#define LAUNCH_MR(PP, PERV) hipLaunchKernelGGL((k_nifs_cross_minroot<PP, PERV>), grid, dim3(256), 0, s, C(z2), seg_begin, one_col, t, row0, \
                                               C(az1), C(bz1), C(cz1), to_val(u1), M(az2), M(bz2), M(cz2), M(T))
  if (field == VDF_FIELD_FP) { if (per == 4) LAUNCH_MR(FpParams, 4); else LAUNCH_MR(FpParams, 3); }
  else if (field == VDF_FIELD_FQ) { if (per == 4) LAUNCH_MR(FqParams, 4); else LAUNCH_MR(FqParams, 3); }
  else return Status{VDF_ERR_BAD_ARG, "unknown field"};
#undef LAUNCH_MR
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

// r: Montgomery form (host); sent as a plain integer when it is below 2^128 and tuning().fold_u128 allows (as vec_fold_many does)
template <class P> static bool plain_if_u128(const vdf_fe* r, FeVal& out) {
  Fe<P> t; memcpy(t.v, r, 32); t = fe_from_mont(t); memcpy(out.v, t.v, 32);
  return (out.v[4] | out.v[5] | out.v[6] | out.v[7]) == 0;
}
Status vec_nifs_cross_minroot_fold(int field, int per, uint64_t t, size_t seg_begin, size_t one_col, size_t row0, const void* z2,
                                   const vdf_fe* r, void* az1, void* bz1, void* cz1, void* e1, const void* tprev, const vdf_fe* u1,
                                   void* az2, void* bz2, void* cz2, void* T, hipStream_t s) {
  if (per != 3 && per != 4) return Status{VDF_ERR_BAD_ARG, "variables per round must be 3 or 4"};
  if (t == 0 || seg_begin < 3) return Status{VDF_ERR_BAD_ARG, "bad segment"};
  if (field != VDF_FIELD_FP && field != VDF_FIELD_FQ) return Status{VDF_ERR_BAD_ARG, "unknown field"};
  const size_t rows = 3 * (size_t)t + 1;
  // algorithmic bytes: the row of 3 running vectors read and written, 3 previous fresh read, 3 fresh + T written (E and the
  // previous T: read, read, written), and the round's variables once
  KTimer kt(s, "k_nifs_cross_minroot_fold", (double)rows * (13 + (e1 ? 3 : 0)) * 32 + (double)per * t * 32);
  const dim3 grid = grid_for(rows);
  FeVal plain{};
  const bool small = tuning().fold_u128 && (field == VDF_FIELD_FP ? plain_if_u128<FpParams>(r, plain) : plain_if_u128<FqParams>(r, plain));
  const FeVal rv = small ? plain : to_val(r);
#define LAUNCH_MRF(PP, PERV, U) hipLaunchKernelGGL((k_nifs_cross_minroot_fold<PP, PERV, U>), grid, dim3(256), 0, s, C(z2), seg_begin, one_col, t, row0, rv, \
                                                  M(az1), M(bz1), M(cz1), M(e1), C(tprev), to_val(u1), M(az2), M(bz2), M(cz2), M(T))
#define LAUNCH_MRF2(PP, PERV) do { if (small) LAUNCH_MRF(PP, PERV, true); else LAUNCH_MRF(PP, PERV, false); } while (0)
  if (field == VDF_FIELD_FP) { if (per == 4) LAUNCH_MRF2(FpParams, 4); else LAUNCH_MRF2(FpParams, 3); }
  else { if (per == 4) LAUNCH_MRF2(FqParams, 4); else LAUNCH_MRF2(FqParams, 3); }
#undef LAUNCH_MRF2
#undef LAUNCH_MRF
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status vec_fold_many(int field, const vdf_fe* r, int k, void* const acc[], const void* const add[], const size_t n[],
                     hipStream_t s) {
  if (k <= 0) return Status{};
  if (k > 8) return Status{VDF_ERR_BAD_ARG, "at most 8 vectors per fold"};
  FoldArgs a{};
  a.k = k;
  uint64_t blocks = 0;
  for (int i = 0; i < k; ++i) {
    a.acc[i] = M(acc[i]); a.add[i] = C(add[i]); a.n[i] = n[i];
    blocks += (n[i] + 255) / 256;
    if (blocks >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "fold too large"};
    a.blk_end[i] = (uint32_t)blocks;
  }
  if (blocks == 0) return Status{};
  double elems = 0;
  for (int i = 0; i < k; ++i) elems += (double)n[i];
  KTimer kt(s, "k_fold_many", 96.0 * elems);
  // a challenge below 2^128 (every NIFS fold): its plain value travels instead, and the product needs no Montgomery reduction
  if (tuning().fold_u128) {
    FeVal plain{};
    bool small = false;
    if (field == VDF_FIELD_FP) { Fe<FpParams> t; memcpy(t.v, r, 32); t = fe_from_mont(t); memcpy(plain.v, t.v, 32); }
    else if (field == VDF_FIELD_FQ) { Fe<FqParams> t; memcpy(t.v, r, 32); t = fe_from_mont(t); memcpy(plain.v, t.v, 32); }
    else return Status{VDF_ERR_BAD_ARG, "unknown field"};
    small = (plain.v[4] | plain.v[5] | plain.v[6] | plain.v[7]) == 0;
    if (small) {
      FIELD_DISPATCH(field, k_fold_many_u128, dim3((unsigned)blocks), dim3(256), 0, s, a, plain);
      return Status{};
    }
  }
  FIELD_DISPATCH(field, k_fold_many, dim3((unsigned)blocks), dim3(256), 0, s, a, to_val(r));
  return Status{};
}

// flag |= 1 when any of the n 32-byte elements is not all-zero (a residual vector checked where it lies)
__global__ __launch_bounds__(256) void k_any_nonzero(const uint4* __restrict__ v, size_t n16, uint32_t* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t any = 0;
  for (size_t k = i; k < n16; k += (size_t)gridDim.x * 256) { const uint4 w = v[k]; any |= w.x | w.y | w.z | w.w; }
  if (__any(any != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
Status vec_any_nonzero(const void* v, size_t n, uint32_t* d_flag, hipStream_t s) {
  VDF_TRY_HIP(hipMemsetAsync(d_flag, 0, 4, s));
  if (n == 0) return Status{};
  const size_t n16 = n * 2;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_any_nonzero, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(v), n16, d_flag);
  VDF_TRY_HIP(hipGetLastError());
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

Status vec_clock_probe(int iters, int workgroups, unsigned long long* d_out3, hipStream_t s) {
  hipLaunchKernelGGL(k_clock_probe, dim3(workgroups), dim3(256), 0, s, iters, d_out3);
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

}  // namespace vdf
