// Kernels of the compression SNARK (protocol "vdf-spartan-v3", restated by the test oracle, spartan.py): a Spartan-style
// argument for the folded relaxed R1CS instance with inner-product-argument openings -- the work behind
// `NovaVDFProof::compress` / verification of the compressed proof (/root/reference/src/nova/proof.rs:360-368, :383;
// nova-snark 0.8.0 `CompressedSNARK`).
//
// Everything here is a streaming pass or a reduction over vectors of field elements: HBM-bound, no MFMA.
// Multilinear tables are MSB-first (index i = sum x_j 2^(k-j)): binding a variable folds the upper half of a table
// onto its lower half, so every pass reads two coalesced halves.
//   k_eq_table        out[i] = prod_j (bit_j(i) ? hi_j : lo_j)   (eq(r, .) with lo = 1 - r, hi = r; the inner-
//                     product argument's generator coefficients with lo = x^-1, hi = x)
//   k_fold_halves     v[i] <- c_lo v[i] + c_hi v[i + h] for up to 8 vectors (sum-check binding: c = (1 - r, r);
//                     inner-product argument: (x, x^-1) / (x^-1, x))
//   k_reduce / k_reduce_final   sums over i < h of a per-kind term (dot product, the two sum-check round
//                     polynomials at their evaluation points, the argument's cross terms); per-workgroup LDS tree,
//                     then one workgroup over the partials
//   k_spmvt / k_spmvt_heavy     M(y) = sum_x eq[x] (A + rho B + rho^2 C)[x, y]: the three matrices merged in column-major
//                     order; a thread per column, a workgroup per heavy column (the constant column has ~t entries)
//   k_ipa_scalars     the two scalar vectors whose MSMs over the ORIGINAL generators are a round's L and R
//   k_scale_pattern   s[t] *= ((t mod n_j) >= n_j / 2) ? x : x^-1
#include <cstring>
#include "internal.h"
#include "fe.cuh"

namespace vdf {

struct FeArg { uint32_t v[8]; };
template <class P> __device__ __forceinline__ Fe<P> arg_fe(const FeArg& a) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = a.v[i];
  return r;
}
static FeArg to_arg(const vdf_fe* p) { FeArg v; std::memcpy(v.v, p, 32); return v; }
static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

#define SNARK_DISPATCH(field, KERNEL, ...)                                                       \
  do {                                                                                           \
    if ((field) == VDF_FIELD_FP) hipLaunchKernelGGL((KERNEL<FpParams>), __VA_ARGS__);            \
    else if ((field) == VDF_FIELD_FQ) hipLaunchKernelGGL((KERNEL<FqParams>), __VA_ARGS__);       \
    else return Status{VDF_ERR_BAD_ARG, "unknown field"};                                        \
    VDF_TRY_HIP(hipGetLastError());                                                              \
  } while (0)

// ---- tensor-product tables ---------------------------------------------------------------------------
struct PairArgs { FeArg lo[24], hi[24]; int k; };

template <class P>
__global__ __launch_bounds__(256) void k_eq_table(PairArgs a, size_t n, char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fe<P> acc = fe_one<P>();
  for (int j = 0; j < a.k; ++j) {
    const bool bit = (i >> (a.k - 1 - j)) & 1;               // variable j is bit k-1-j of the index (x_1 = MSB)
    acc = fe_mul(acc, arg_fe<P>(bit ? a.hi[j] : a.lo[j]));
  }
  fe_store<P>(out + i * 32, acc);
}

Status snark_pair_table(int field, const vdf_fe* lo, const vdf_fe* hi, int k, void* out, hipStream_t s) {
  if (k < 0 || k > 24) return Status{VDF_ERR_BAD_LENGTH, "at most 24 variables"};
  PairArgs a{};
  a.k = k;
  for (int j = 0; j < k; ++j) { a.lo[j] = to_arg(&lo[j]); a.hi[j] = to_arg(&hi[j]); }
  const size_t n = (size_t)1 << k;
  KTimer kt(s, "k_eq_table", 32.0 * n);                          // one element written per index; the factors are kernel arguments
  SNARK_DISPATCH(field, k_eq_table, grid_for(n), dim3(256), 0, s, a, n, reinterpret_cast<char*>(out));
  return Status{};
}

// out[i] = (pair table over the top k bits of i) * pattern[low log_m bits of i]: the coefficient of generator i in the
// verifier's check of an inner-product argument that stopped at a vector of 2^log_m elements
struct PatternArgs { FeArg p[16]; int log_m; };
template <class P>
__global__ __launch_bounds__(256) void k_eq_table_pattern(PairArgs a, PatternArgs pat, size_t n, char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t hi_idx = i >> pat.log_m;
  Fe<P> acc = arg_fe<P>(pat.p[i & (((size_t)1 << pat.log_m) - 1)]);
  for (int j = 0; j < a.k; ++j) {
    const bool bit = (hi_idx >> (a.k - 1 - j)) & 1;
    acc = fe_mul(acc, arg_fe<P>(bit ? a.hi[j] : a.lo[j]));
  }
  fe_store<P>(out + i * 32, acc);
}

Status snark_pair_table_pattern(int field, const vdf_fe* lo, const vdf_fe* hi, int k, const vdf_fe* pattern, int log_m, void* out,
                                hipStream_t s) {
  if (k < 0 || log_m < 0 || log_m > 4 || k + log_m > 24) return Status{VDF_ERR_BAD_LENGTH, "at most 24 variables, pattern of at most 16"};
  PairArgs a{};
  a.k = k;
  for (int j = 0; j < k; ++j) { a.lo[j] = to_arg(&lo[j]); a.hi[j] = to_arg(&hi[j]); }
  PatternArgs pa{};
  pa.log_m = log_m;
  for (int j = 0; j < (1 << log_m); ++j) pa.p[j] = to_arg(&pattern[j]);
  const size_t n = (size_t)1 << (k + log_m);
  KTimer kt(s, "k_eq_table_pattern", 32.0 * n);
  SNARK_DISPATCH(field, k_eq_table_pattern, grid_for(n), dim3(256), 0, s, a, pa, n, reinterpret_cast<char*>(out));
  return Status{};
}

// ---- folding the two halves of up to 8 vectors ----------------------------------------------------------
struct FoldHalvesArgs { char* v[8]; FeArg c_lo[8], c_hi[8]; int k; };

template <class P>
__global__ __launch_bounds__(256) void k_fold_halves(FoldHalvesArgs a, size_t h) {
  __builtin_amdgcn_s_setprio(3);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= h) return;
  const int t = blockIdx.y;
  char* v = a.v[t];
  const Fe<P> lo = fe_load<P>(v + i * 32), hi = fe_load<P>(v + (h + i) * 32);
  fe_store<P>(v + i * 32, fe_add(fe_mul(arg_fe<P>(a.c_lo[t]), lo), fe_mul(arg_fe<P>(a.c_hi[t]), hi)));
}

Status snark_fold_halves(int field, int k, void* const v[], const vdf_fe c_lo[], const vdf_fe c_hi[], size_t n, hipStream_t s) {
  if (k <= 0) return Status{};
  if (k > 8) return Status{VDF_ERR_BAD_ARG, "at most 8 vectors"};
  if (n < 2 || (n & (n - 1))) return Status{VDF_ERR_BAD_LENGTH, "length must be a power of two >= 2"};
  FoldHalvesArgs a{};
  a.k = k;
  for (int t = 0; t < k; ++t) { a.v[t] = reinterpret_cast<char*>(v[t]); a.c_lo[t] = to_arg(&c_lo[t]); a.c_hi[t] = to_arg(&c_hi[t]); }
  const size_t h = n / 2;
  dim3 grid((unsigned)((h + 255) / 256), (unsigned)k);
  KTimer kt(s, "k_fold_halves", 96.0 * h * k);                   // two halves read, the lower one written, per vector
  SNARK_DISPATCH(field, k_fold_halves, grid, dim3(256), 0, s, a, h);
  return Status{};
}

// ---- reductions ------------------------------------------------------------------------------------------
// kind 0: dot product            sum a[i] b[i], i < n                                   (1 value)
// kind 1: quadratic round        g(0), g(2) of sum (p_lo + t dp)(q_lo + t dq), i < h     (2 values)
// kind 2: cubic R1CS round       g(0), g(2), g(3) of sum eq_t (a_t b_t - u c_t - e_t)    (3 values)
// kind 3: argument cross terms   sum a[i] b[h + i],  sum a[h + i] b[i], i < h            (2 values)
struct ReduceArgs { const char* t[5]; FeArg u; size_t n; };
static constexpr int REDUCE_BLOCKS = 512;

template <class P, int KIND>
__device__ __forceinline__ void reduce_term(const ReduceArgs& a, size_t i, size_t h, Fe<P> acc[3]) {
  if (KIND == 0) {
    acc[0] = fe_add(acc[0], fe_mul(fe_load<P>(a.t[0] + i * 32), fe_load<P>(a.t[1] + i * 32)));
  } else if (KIND == 1) {
    const Fe<P> p0 = fe_load<P>(a.t[0] + i * 32), p1 = fe_load<P>(a.t[0] + (h + i) * 32);
    const Fe<P> q0 = fe_load<P>(a.t[1] + i * 32), q1 = fe_load<P>(a.t[1] + (h + i) * 32);
    acc[0] = fe_add(acc[0], fe_mul(p0, q0));
    const Fe<P> p2 = fe_sub(fe_dbl(p1), p0), q2 = fe_sub(fe_dbl(q1), q0);        // lo + 2 (hi - lo)
    acc[1] = fe_add(acc[1], fe_mul(p2, q2));
  } else if (KIND == 2) {
    const Fe<P> u = arg_fe<P>(a.u);
    Fe<P> lo[5], d[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      lo[k] = fe_load<P>(a.t[k] + i * 32);
      d[k] = fe_sub(fe_load<P>(a.t[k] + (h + i) * 32), lo[k]);
    }
    // t = 0
    acc[0] = fe_add(acc[0], fe_mul(lo[0], fe_sub(fe_sub(fe_mul(lo[1], lo[2]), fe_mul(u, lo[3])), lo[4])));
    // t = 2, then t = 3
    Fe<P> v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = fe_add(lo[k], fe_dbl(d[k]));
    acc[1] = fe_add(acc[1], fe_mul(v[0], fe_sub(fe_sub(fe_mul(v[1], v[2]), fe_mul(u, v[3])), v[4])));
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = fe_add(v[k], d[k]);
    acc[2] = fe_add(acc[2], fe_mul(v[0], fe_sub(fe_sub(fe_mul(v[1], v[2]), fe_mul(u, v[3])), v[4])));
  } else {
    acc[0] = fe_add(acc[0], fe_mul(fe_load<P>(a.t[0] + i * 32), fe_load<P>(a.t[1] + (h + i) * 32)));
    acc[1] = fe_add(acc[1], fe_mul(fe_load<P>(a.t[0] + (h + i) * 32), fe_load<P>(a.t[1] + i * 32)));
  }
}

template <int KIND> struct ReduceOuts { static constexpr int N = KIND == 0 ? 1 : KIND == 2 ? 3 : 2; };

template <class P, int NOUT>
__device__ __forceinline__ void block_tree(Fe<P> acc[3], char* lds) {
  // lds: 256 x NOUT field elements
  for (int k = 0; k < NOUT; ++k) fe_store<P>(lds + ((size_t)k * 256 + threadIdx.x) * 32, acc[k]);
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if ((int)threadIdx.x < stride)
      for (int k = 0; k < NOUT; ++k) {
        char* p = lds + ((size_t)k * 256 + threadIdx.x) * 32;
        fe_store<P>(p, fe_add(fe_load<P>(p), fe_load<P>(p + (size_t)stride * 32)));
      }
    __syncthreads();
  }
}

template <class P, int KIND>
__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a, char* __restrict__ partials) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) char lds[256 * 3 * 32];
  constexpr int NOUT = ReduceOuts<KIND>::N;
  const size_t h = KIND == 0 ? a.n : a.n / 2;
  Fe<P> acc[3] = {fe_zero<P>(), fe_zero<P>(), fe_zero<P>()};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < h; i += (size_t)gridDim.x * 256) reduce_term<P, KIND>(a, i, h, acc);
  block_tree<P, NOUT>(acc, lds);
  if (threadIdx.x < NOUT) fe_store<P>(partials + ((size_t)blockIdx.x * NOUT + threadIdx.x) * 32, fe_load<P>(lds + (size_t)threadIdx.x * 256 * 32));
}

template <class P, int NOUT>
__global__ __launch_bounds__(256) void k_reduce_final(const char* __restrict__ partials, int nblocks, char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) char lds[256 * 3 * 32];
  Fe<P> acc[3] = {fe_zero<P>(), fe_zero<P>(), fe_zero<P>()};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int k = 0; k < NOUT; ++k) acc[k] = fe_add(acc[k], fe_load<P>(partials + ((size_t)b * NOUT + k) * 32));
  block_tree<P, NOUT>(acc, lds);
  if (threadIdx.x < NOUT) fe_store<P>(out + (size_t)threadIdx.x * 32, fe_load<P>(lds + (size_t)threadIdx.x * 256 * 32));
}

template <class P, int KIND>
static Status reduce_launch(const ReduceArgs& a, void* scratch, void* out, hipStream_t s) {
  constexpr int NOUT = ReduceOuts<KIND>::N;
  const size_t h = KIND == 0 ? a.n : a.n / 2;
  int blocks = (int)((h + 255) / 256);
  if (blocks > REDUCE_BLOCKS) blocks = REDUCE_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_reduce<P, KIND>), dim3(blocks), dim3(256), 0, s, a, reinterpret_cast<char*>(scratch));
  hipLaunchKernelGGL((k_reduce_final<P, NOUT>), dim3(1), dim3(256), 0, s, reinterpret_cast<const char*>(scratch), blocks,
                     reinterpret_cast<char*>(out));
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

size_t snark_reduce_scratch_bytes() { return (size_t)REDUCE_BLOCKS * 3 * 32; }

Status snark_reduce(int field, int kind, const void* const tables[], const vdf_fe* u, size_t n, void* scratch, void* out,
                    hipStream_t s) {
  if (kind < 0 || kind > 3) return Status{VDF_ERR_BAD_ARG, "unknown reduction"};
  if (kind != 0 && (n < 2 || (n & (n - 1)))) return Status{VDF_ERR_BAD_LENGTH, "length must be a power of two >= 2"};
  ReduceArgs a{};
  const int ntab = kind == 2 ? 5 : 2;
  for (int k = 0; k < ntab; ++k) a.t[k] = reinterpret_cast<const char*>(tables[k]);
  if (u) a.u = to_arg(u);
  a.n = n;
  // bytes read: kind 0 two vectors of n; kind 1 / 3 two vectors of n (both halves); kind 2 five tables of n
  KTimer kt(s, kind == 0 ? "k_reduce(dot)" : kind == 1 ? "k_reduce(quad round)" : kind == 2 ? "k_reduce(cubic round)" : "k_reduce(ipa cross)",
            32.0 * n * ntab);
#define RL(P, K) reduce_launch<P, K>(a, scratch, out, s)
  if (field == VDF_FIELD_FP) return kind == 0 ? RL(FpParams, 0) : kind == 1 ? RL(FpParams, 1) : kind == 2 ? RL(FpParams, 2) : RL(FpParams, 3);
  if (field == VDF_FIELD_FQ) return kind == 0 ? RL(FqParams, 0) : kind == 1 ? RL(FqParams, 1) : kind == 2 ? RL(FqParams, 2) : RL(FqParams, 3);
#undef RL
  return Status{VDF_ERR_BAD_ARG, "unknown field"};
}

// ---- transposed sparse product -----------------------------------------------------------------------------
// Column-major merge of the three matrices: entry = (row, coefficient index | matrix << 30).
static constexpr uint32_t SPMVT_HEAVY = 64;      // columns with more entries than this get a workgroup of their own

template <class P>
__device__ __forceinline__ Fe<P> spmvt_entry(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ cm,
                                             const char* __restrict__ dict, const char* __restrict__ eq, const Fe<P> pw[3],
                                             uint32_t k) {
  const uint32_t c = cm[k], ci = c & 0x3FFFFFFFu, mat = c >> 30;
  Fe<P> v = fe_load<P>(eq + (size_t)rows[k] * 32);
  if (ci == 1) v = fe_neg(v);                                   // dictionary index 0 = +1, 1 = -1
  else if (ci > 1) v = fe_mul(v, fe_load<P>(dict + (size_t)ci * 32));
  return mat == 0 ? v : fe_mul(v, pw[mat]);
}

template <class P>
__global__ __launch_bounds__(256) void k_spmvt(const uint32_t* __restrict__ colptr, const uint32_t* __restrict__ rows,
                                               const uint32_t* __restrict__ cm, const char* __restrict__ dict,
                                               const char* __restrict__ eq, FeArg rho, size_t ncols, char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  const uint32_t lo = colptr[c], hi = colptr[c + 1];
  if (hi - lo > SPMVT_HEAVY) return;                            // k_spmvt_heavy writes this column
  Fe<P> pw[3];
  pw[0] = fe_one<P>(); pw[1] = arg_fe<P>(rho); pw[2] = fe_mul(pw[1], pw[1]);
  Fe<P> acc = fe_zero<P>();
  for (uint32_t k = lo; k < hi; ++k) acc = fe_add(acc, spmvt_entry<P>(rows, cm, dict, eq, pw, k));
  fe_store<P>(out + c * 32, acc);
}

template <class P>
__global__ __launch_bounds__(256) void k_spmvt_heavy(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ colptr,
                                                     const uint32_t* __restrict__ rows, const uint32_t* __restrict__ cm,
                                                     const char* __restrict__ dict, const char* __restrict__ eq, FeArg rho,
                                                     char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) char lds[256 * 3 * 32];
  const uint32_t c = heavy[blockIdx.x];
  const uint32_t lo = colptr[c], hi = colptr[c + 1];
  Fe<P> pw[3];
  pw[0] = fe_one<P>(); pw[1] = arg_fe<P>(rho); pw[2] = fe_mul(pw[1], pw[1]);
  Fe<P> acc[3] = {fe_zero<P>(), fe_zero<P>(), fe_zero<P>()};
  for (uint32_t k = lo + threadIdx.x; k < hi; k += 256) acc[0] = fe_add(acc[0], spmvt_entry<P>(rows, cm, dict, eq, pw, k));
  block_tree<P, 1>(acc, lds);
  if (threadIdx.x == 0) fe_store<P>(out + (size_t)c * 32, fe_load<P>(lds));
}

// a heavy column shared by SPMVT_PARTS workgroups (the constant column of the step circuit has ~3t entries: one
// workgroup walking them was 0.5 ms): each sums a contiguous chunk into a partial, k_spmvt_heavy_sum adds the partials
static constexpr uint32_t SPMVT_PARTS = 64;
template <class P>
__global__ __launch_bounds__(256) void k_spmvt_heavy_part(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ colptr,
                                                          const uint32_t* __restrict__ rows, const uint32_t* __restrict__ cm,
                                                          const char* __restrict__ dict, const char* __restrict__ eq, FeArg rho,
                                                          char* __restrict__ partials) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) char lds[256 * 3 * 32];
  const uint32_t item = blockIdx.x / SPMVT_PARTS, part = blockIdx.x % SPMVT_PARTS;
  const uint32_t c = heavy[item];
  const uint32_t lo = colptr[c], hi = colptr[c + 1];
  const uint32_t chunk = (hi - lo + SPMVT_PARTS - 1) / SPMVT_PARTS;
  const uint32_t k0 = lo + part * chunk, k1 = (k0 + chunk < hi) ? k0 + chunk : hi;
  Fe<P> pw[3];
  pw[0] = fe_one<P>(); pw[1] = arg_fe<P>(rho); pw[2] = fe_mul(pw[1], pw[1]);
  Fe<P> acc[3] = {fe_zero<P>(), fe_zero<P>(), fe_zero<P>()};
  for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) acc[0] = fe_add(acc[0], spmvt_entry<P>(rows, cm, dict, eq, pw, k));
  block_tree<P, 1>(acc, lds);
  if (threadIdx.x == 0) fe_store<P>(partials + (size_t)blockIdx.x * 32, fe_load<P>(lds));
}
template <class P>
__global__ __launch_bounds__(64) void k_spmvt_heavy_sum(const uint32_t* __restrict__ heavy, const char* __restrict__ partials,
                                                        char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) char lds[64 * 32];
  fe_store<P>(lds + (size_t)threadIdx.x * 32, fe_load<P>(partials + ((size_t)blockIdx.x * SPMVT_PARTS + threadIdx.x) * 32));
  __syncthreads();
  for (int stride = 32; stride >= 1; stride >>= 1) {
    if ((int)threadIdx.x < stride) {
      char* p = lds + (size_t)threadIdx.x * 32;
      fe_store<P>(p, fe_add(fe_load<P>(p), fe_load<P>(p + (size_t)stride * 32)));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) fe_store<P>(out + (size_t)heavy[blockIdx.x] * 32, fe_load<P>(lds));
}

// scratch: snark_reduce_scratch_bytes() bytes (the context's reduction scratch), or nullptr
Status snark_spmvt(int field, const uint32_t* colptr, const uint32_t* rows, const uint32_t* cm, const uint32_t* heavy,
                   size_t nheavy, size_t nbig, const void* dict, const void* eq, const vdf_fe* rho, size_t ncols, void* out,
                   void* scratch, hipStream_t s) {
  if (ncols == 0) return Status{};
  const FeArg r = to_arg(rho);
  SNARK_DISPATCH(field, k_spmvt, grid_for(ncols), dim3(256), 0, s, colptr, rows, cm, reinterpret_cast<const char*>(dict),
                 reinterpret_cast<const char*>(eq), r, ncols, reinterpret_cast<char*>(out));
  if (!nheavy) return Status{};
  // the list is sorted longest first: the first `nbig` columns (thousands of entries) are shared by SPMVT_PARTS workgroups
  // each, as far as the scratch reaches; the others get one workgroup each
  size_t shared = 0;
  if (scratch) {
    shared = nbig < nheavy ? nbig : nheavy;
    const size_t room = snark_reduce_scratch_bytes() / (SPMVT_PARTS * 32);
    if (shared > room) shared = room;
  }
  if (shared) {
    SNARK_DISPATCH(field, k_spmvt_heavy_part, dim3((unsigned)(shared * SPMVT_PARTS)), dim3(256), 0, s, heavy, colptr, rows, cm,
                   reinterpret_cast<const char*>(dict), reinterpret_cast<const char*>(eq), r, reinterpret_cast<char*>(scratch));
    SNARK_DISPATCH(field, k_spmvt_heavy_sum, dim3((unsigned)shared), dim3(64), 0, s, heavy, reinterpret_cast<const char*>(scratch),
                   reinterpret_cast<char*>(out));
  }
  if (nheavy > shared)
    SNARK_DISPATCH(field, k_spmvt_heavy, dim3((unsigned)(nheavy - shared)), dim3(256), 0, s, heavy + shared, colptr, rows, cm,
                   reinterpret_cast<const char*>(dict), reinterpret_cast<const char*>(eq), r, reinterpret_cast<char*>(out));
  return Status{};
}

// ---- inner-product argument helpers ------------------------------------------------------------------------
// Folded generators are never materialised: G^(j)_i = sum over the original indices t = i (mod n_j) of s[t] G_t, so a
// round's L = <a_lo, G_hi> and R = <a_hi, G_lo> are MSMs over the ORIGINAL generators with these scalars:
//   sL[t] = s[t] a[(t mod n_j) - h]  if (t mod n_j) >= h  else 0        sR[t] = s[t] a[(t mod n_j) + h]  if (t mod n_j) < h  else 0
template <class P>
__global__ __launch_bounds__(256) void k_ipa_scalars(const char* __restrict__ a, const char* __restrict__ sv, size_t n,
                                                     size_t nj, char* __restrict__ sL, char* __restrict__ sR) {
  __builtin_amdgcn_s_setprio(3);
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const size_t r = t & (nj - 1), h = nj >> 1;
  const Fe<P> sc = fe_load<P>(sv + t * 32);
  const Fe<P> z = fe_zero<P>();
  if (r >= h) {
    fe_store<P>(sL + t * 32, fe_mul(sc, fe_load<P>(a + (r - h) * 32)));
    fe_store<P>(sR + t * 32, z);
  } else {
    fe_store<P>(sL + t * 32, z);
    fe_store<P>(sR + t * 32, fe_mul(sc, fe_load<P>(a + (r + h) * 32)));
  }
}

template <class P>
__global__ __launch_bounds__(256) void k_scale_pattern(char* __restrict__ sv, size_t n, size_t nj, FeArg x_lo, FeArg x_hi) {
  __builtin_amdgcn_s_setprio(3);
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const bool upper = (t & (nj - 1)) >= (nj >> 1);
  fe_store<P>(sv + t * 32, fe_mul(fe_load<P>(sv + t * 32), arg_fe<P>(upper ? x_hi : x_lo)));
}

Status snark_ipa_scalars(int field, const void* a, const void* sv, size_t n, size_t nj, void* sL, void* sR, hipStream_t s) {
  if (n == 0 || (n & (n - 1)) || nj < 2 || (nj & (nj - 1)) || nj > n) return Status{VDF_ERR_BAD_LENGTH, "lengths must be powers of two, 2 <= n_j <= n"};
  KTimer kt(s, "k_ipa_scalars", 96.0 * n + 32.0 * nj);           // s read, sL and sR written, a read once
  SNARK_DISPATCH(field, k_ipa_scalars, grid_for(n), dim3(256), 0, s, reinterpret_cast<const char*>(a),
                 reinterpret_cast<const char*>(sv), n, nj, reinterpret_cast<char*>(sL), reinterpret_cast<char*>(sR));
  return Status{};
}

Status snark_scale_pattern(int field, void* sv, size_t n, size_t nj, const vdf_fe* x_lo, const vdf_fe* x_hi, hipStream_t s) {
  if (n == 0 || (n & (n - 1)) || nj < 2 || (nj & (nj - 1)) || nj > n) return Status{VDF_ERR_BAD_LENGTH, "lengths must be powers of two, 2 <= n_j <= n"};
  KTimer kt(s, "k_scale_pattern", 64.0 * n);
  SNARK_DISPATCH(field, k_scale_pattern, grid_for(n), dim3(256), 0, s, reinterpret_cast<char*>(sv), n, nj, to_arg(x_lo), to_arg(x_hi));
  return Status{};
}

}  // namespace vdf
