// Direct-sum MSM over a digit table, for the SMALL fixed-base commitments on a prover's critical path.
//
// The commitments a Nova step waits for (the ~10^4-term witness of an augmented circuit and the rows of a cross term
// that depend on it; /root/reference/src/nova/proof.rs:342-349 -> RecursiveSNARK::prove_step) are latency problems:
// the bucket method spends 50 us adding points and 300 us reducing buckets in chains of dependent additions
// (profiles/r02_prove_step_timeline.txt).  With the generators fixed and 288 GB of HBM, the buckets can be dropped
// altogether:
//
//   digit table   D[slot][j][d-1] = [d * 2^(c j)] G_slot      d = 1 .. 2^(c-1), j = 0 .. W-1     (affine, 64 B)
//   scalar        k = sum_j e_j 2^(c j),  e_j = window_j(k + H) - 2^(c-1)  in [-2^(c-1), 2^(c-1)),  H = sum_j 2^(c j + c - 1)
//   MSM           sum_i sum_j sign(e_ij) D[i][j][|e_ij| - 1]                       -- a plain sum of gathered points
//
// c = 10: 26 gathers per scalar from an 852 KB region per generator (8.6 GB for the 10^4 generators of a secondary
// circuit).  k_direct_sum is one round of one wavefront per SIMD: a lane adds its share of the (window, scalar) entries
// (~8 mixed additions for 2 x 10^4 scalars, the gather of the next one in flight), the 64 accumulators of a wavefront
// and then the 4 wavefronts of a workgroup are summed through LDS with the quad-cooperative group law, one point per
// workgroup; k_direct_final sums a group's workgroup points.  Depth: ~8 lane additions + ~20 quad additions, instead
// of ~80 quad operations behind a sort.
//
// Only groups whose generators lie inside the table's ranges take this path (abi.hip msm_core); everything else is
// the bucket method of msm.hip.  The result is the same group element (its Jacobian representative differs).
#include <cstdlib>
#include <cstring>
#include "internal.h"
#include "ec.cuh"
#include "ecq.cuh"

namespace vdf {

int direct_windows(int c) {
  int W = (256 + c - 1) / c;
  const int top = 255 - c * (W - 1);           // scalar bits that reach the top window (scalars < 2^255)
  if (top > c - 2) ++W;                        // top digit + 2^(c-1) + carry must stay below 2^c
  return W;
}

// ------------------------------------------------------------------------------------------
// table construction
// ------------------------------------------------------------------------------------------
// d = 1 of every window: thread per slot, c doublings per window, one inversion each
template <class P>
__global__ __launch_bounds__(256) void k_digit_base(const char* __restrict__ pts, uint32_t nslots, int c, int W,
                                                    char* __restrict__ D) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nslots) return;
  Affine<P> a = affine_load<P>(pts + (size_t)s * 64);
  for (int j = 0; j < W; ++j) {
    affine_store<P>(D + ((((size_t)s * W + j) << (c - 1))) * 64, a);
    if (j + 1 == W) break;
    XYZZ<P> r = xyzz_from_affine(a);
    for (int k = 0; k < c; ++k) r = xyzz_dbl(r);
    a = xyzz_to_affine(r);
  }
}

// d = 2 .. 2^(c-1): thread per (slot, window).  From P_1..P_m to P_{m+1}..P_{2m} as P_k + P_m (k = m: the doubling), the
// m slope denominators inverted together (prefix products in `scratch`, one field inversion per level).
template <class P>
__global__ __launch_bounds__(256) void k_digit_fill(char* __restrict__ D, uint32_t first_pair, uint32_t npairs, int c,
                                                    char* __restrict__ scratch) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= npairs) return;
  const uint32_t M = 1u << (c - 1);
  char* base = D + (((size_t)first_pair + t) << (c - 1)) * 64;
  char* pre = scratch + (size_t)t * (M / 2) * 32;
  for (uint32_t m = 1; m < M; m <<= 1) {
    const Affine<P> Pm = affine_load<P>(base + (size_t)(m - 1) * 64);
    Fe<P> run = fe_one<P>();
    for (uint32_t k = 1; k <= m; ++k) {
      const Fe<P> den = k < m ? fe_sub(fe_load<P>(base + (size_t)(k - 1) * 64), Pm.x) : fe_dbl(Pm.y);
      fe_store<P>(pre + (size_t)(k - 1) * 32, run);
      run = fe_mul(run, den);
    }
    Fe<P> inv = fe_inv(run);
    for (uint32_t k = m; k >= 1; --k) {
      const Affine<P> Pk = affine_load<P>(base + (size_t)(k - 1) * 64);
      const Fe<P> den = k < m ? fe_sub(Pk.x, Pm.x) : fe_dbl(Pm.y);
      const Fe<P> ik = fe_mul(inv, fe_load<P>(pre + (size_t)(k - 1) * 32));
      inv = fe_mul(inv, den);
      Fe<P> lam;
      if (k < m) lam = fe_mul(fe_sub(Pk.y, Pm.y), ik);
      else { const Fe<P> xx = fe_sqr(Pm.x); lam = fe_mul(fe_add(fe_dbl(xx), xx), ik); }
      Affine<P> o;
      o.x = fe_sub(fe_sub(fe_sqr(lam), Pk.x), Pm.x);
      o.y = fe_sub(fe_mul(lam, fe_sub(Pk.x, o.x)), Pk.y);
      affine_store<P>(base + (size_t)(m + k - 1) * 64, o);
    }
  }
}

Status digits_build(int curve, const void* d_pts, size_t first, size_t nslots, size_t slot0, int c, void* d_digits,
                    hipStream_t stream) {
  if (nslots == 0) return Status{};
  const int W = direct_windows(c);
  const uint32_t M = 1u << (c - 1);
  char* D = reinterpret_cast<char*>(d_digits) + ((slot0 * W) << (c - 1)) * 64;
  const char* pts = reinterpret_cast<const char*>(d_pts) + first * 64;
  // the fill keeps M/2 prefix products per thread: chunks of (slot, window) pairs under 256 MiB of scratch
  const size_t per_thread = (size_t)(M / 2) * 32;
  size_t chunk = ((size_t)256 << 20) / per_thread;
  if (chunk > nslots * W) chunk = nslots * W;
  if (chunk < 256) chunk = 256;
  void* scratch = nullptr;
  VDF_TRY_HIP(hipMalloc(&scratch, chunk * per_thread));
  const dim3 gb((unsigned)((nslots + 255) / 256));
  if (curve == VDF_CURVE_PALLAS) hipLaunchKernelGGL((k_digit_base<FpParams>), gb, dim3(256), 0, stream, pts, (uint32_t)nslots, c, W, D);
  else hipLaunchKernelGGL((k_digit_base<FqParams>), gb, dim3(256), 0, stream, pts, (uint32_t)nslots, c, W, D);
  for (size_t p0 = 0; p0 < nslots * W; p0 += chunk) {
    const size_t np = (p0 + chunk <= nslots * W) ? chunk : nslots * W - p0;
    const dim3 g((unsigned)((np + 255) / 256));
    if (curve == VDF_CURVE_PALLAS)
      hipLaunchKernelGGL((k_digit_fill<FpParams>), g, dim3(256), 0, stream, D, (uint32_t)p0, (uint32_t)np, c, reinterpret_cast<char*>(scratch));
    else
      hipLaunchKernelGGL((k_digit_fill<FqParams>), g, dim3(256), 0, stream, D, (uint32_t)p0, (uint32_t)np, c, reinterpret_cast<char*>(scratch));
  }
  hipError_t e = hipStreamSynchronize(stream);
  (void)hipFree(scratch);
  VDF_TRY_HIP(e);
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

// ------------------------------------------------------------------------------------------
// the sum
// ------------------------------------------------------------------------------------------
struct DirectArgs {
  const uint32_t* scalars[MSM_MAX_GROUPS];
  uint32_t n[MSM_MAX_GROUPS], slot0[MSM_MAX_GROUPS];
  uint32_t wave_end[MSM_MAX_GROUPS];   // group g owns the wavefronts [wave_end[g-1], wave_end[g]) of k_direct_sum
  uint32_t wg_end[MSM_MAX_GROUPS];     // ... and the workgroup points [wg_end[g-1], wg_end[g]) that k_direct_final reads
  uint32_t half[9];                    // H = sum_j 2^(c j + c - 1)
  uint32_t per_lane;                   // entries (scalar, window) per lane
  int groups;
};

// The entries of a group are its (window, scalar) pairs, window-major: entry e = j * n + s.  A group owns L lanes
// (whole wavefronts); lane x takes the entries x, x + L, x + 2L, ... (per_lane of them at most): consecutive lanes read
// consecutive scalars, and the launch is ONE round of one wavefront per SIMD -- the additions are VALU-issue bound, so
// a second wavefront on a SIMD only doubles the time of both.
// Wave priority of the direct sum: 2 -- above a bucket accumulation (0), below the sort and bucket-reduction kernels of an
// MSM pipeline (3).  A prover's direct sums are its main queue's, but the longest dependent path of a step runs through the
// side queue that commits the early rows of the cross term (sort, accumulation, bucket reduction): at equal priority the
// direct sum's one wavefront per SIMD takes half of the issue slots those latency-bound kernels need.  Measured r3, one box:
// 0.942 ms per step at 3, 0.906 at 2, 0.910 at 1, 0.902 at 0 (vdf_hip_tuning.direct_priority, a kernel argument).
// fused = 0 (vdf_hip_tuning.direct_fused, or no arrival counters): the launch ends with the workgroup points and
// k_direct_final adds them up.
template <class P, class SP>
__global__ __launch_bounds__(256) void k_direct_sum(DirectArgs a, int is_mont, int c, int W, const char* __restrict__ D,
                                                    char* __restrict__ partials, uint32_t* __restrict__ arrived, char* __restrict__ out,
                                                    int wave_prio, int fused) {
  if (wave_prio >= 3) __builtin_amdgcn_s_setprio(3); else if (wave_prio == 2) __builtin_amdgcn_s_setprio(2); else if (wave_prio == 1) __builtin_amdgcn_s_setprio(1);
  __shared__ uint32_t limbs[9 * 256];                              // k + H, one private column per thread
  __shared__ __align__(16) char pts[256 * 128];
  __shared__ uint32_t ticket;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;     // a wavefront serves one group
  int g = 0;
  while (g < a.groups - 1 && wave >= a.wave_end[g]) ++g;
  const uint32_t w0 = g ? a.wave_end[g - 1] : 0u;
  const bool live_wave = wave < a.wave_end[a.groups - 1];
  XYZZ<P> acc = xyzz_identity<P>();
  bool have = false, flip = false;            // ec.cuh xyzz_madd_lazy: the mixed addition of the bucket loop, sign-tracked
  if (live_wave) {
    const uint32_t n = a.n[g];
    const uint32_t E = n * (uint32_t)W;                            // entries of the group (at most 2^17 x 33)
    const uint32_t L = (a.wave_end[g] - w0) * 64u;                 // its lanes
    const uint32_t x = (wave - w0) * 64u + lane;
    uint32_t* col = limbs + threadIdx.x;
    const uint32_t mask = (1u << c) - 1u, mid = 1u << (c - 1);
    const char* Dg = D + ((((size_t)a.slot0[g]) * W) << (c - 1)) * 64;
    const uint32_t* sc = a.scalars[g];
    // Three entries in flight: the scalar of entry i + 2 is being read while the table point of entry i + 1 (whose address
    // needed ITS scalar) is being gathered and entry i is added -- two dependent memory round trips behind one addition.
    auto scalar_of = [&](uint32_t e) -> Fe<SP> { return fe_load<SP>(sc + (size_t)(e % n) * 8); };
    // entry -> signed digit and the address of its table point (nullptr: digit 0)
    auto locate = [&](uint32_t e, Fe<SP> k, bool& neg) -> const char* {
      const uint32_t j = e / n, s = e - j * n;
      if (is_mont) k = fe_from_mont(k);
      uint64_t carry = 0;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        carry += (uint64_t)k.v[l] + a.half[l];
        col[l * 256] = (uint32_t)carry;
        carry >>= 32;
      }
      col[8 * 256] = (uint32_t)carry + a.half[8];
      const uint32_t bit = j * (uint32_t)c, l = bit >> 5, sh = bit & 31u;
      const uint32_t lo = col[l * 256];
      const uint32_t hi = (l + 1 < 9) ? col[(l + 1) * 256] : 0u;
      const int d = (int)((uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & mask) - (int)mid;
      if (d == 0) return nullptr;
      neg = d < 0;
      const uint32_t mag = (uint32_t)(neg ? -d : d);
      return Dg + (((((size_t)s * W) + j) << (c - 1)) + (mag - 1)) * 64;
    };
    const uint32_t count = x < E ? (E - x + L - 1) / L : 0u;       // entries of this lane (at most per_lane)
    uint32_t e = x;
    bool neg = false, negn = false;
    Fe<SP> k1 = fe_zero<SP>();
    const char* ptr = nullptr;
    Affine<P> pt;
    if (count) {
      const Fe<SP> k0 = scalar_of(e);
      if (count > 1) k1 = scalar_of(e + L);
      ptr = locate(e, k0, neg);
      if (ptr) pt = affine_load<P>(ptr);
    }
    for (uint32_t i = 0; i < count; ++i) {
      Fe<SP> k2 = fe_zero<SP>();
      if (i + 2 < count) k2 = scalar_of(e + 2 * L);
      const char* ptrn = (i + 1 < count) ? locate(e + L, k1, negn) : nullptr;
      Affine<P> ptn;
      if (ptrn) ptn = affine_load<P>(ptrn);
      if (ptr) {
        if (neg != (have && flip)) pt.y = fe_neg_nz(pt.y);
        xyzz_madd_lazy<P>(acc, have, flip, pt);
      }
      e += L; ptr = ptrn; neg = negn; pt = ptn; k1 = k2;
    }
  }
  xyzz_store<P>(pts + (size_t)threadIdx.x * 128, xyzz_lazy_resolve<P>(acc, have, flip));
  __syncthreads();
  // the wavefront's 64 points: each of its 16 quads adds four, then a butterfly over the quads
  const uint32_t qd = lane >> 2, wv = threadIdx.x >> 6;
  const char* mine = pts + (size_t)((threadIdx.x & ~63u) + 4u * qd) * 128;
  QPoint<P> r = qpoint_load_lazy<P>(mine);
#pragma unroll 1
  for (int m = 1; m < 4; ++m) r = qpoint_add<P>(r, qpoint_load_lazy<P>(mine + (size_t)m * 128));
  r = qpoint_wave_sum(r);
  __syncthreads();                                                 // every wavefront has read its points: reuse the buffer
  if (qd == 0) qpoint_store<P>(pts + (size_t)wv * 128, r);
  __syncthreads();
  // the four wavefronts of a workgroup serve one group (wave_end is a multiple of 4): one point per workgroup
  if (wv == 0 && qd == 0) {
    QPoint<P> t = qpoint_load<P>(pts);
#pragma unroll 1
    for (int m = 1; m < 4; ++m) t = qpoint_add<P>(t, qpoint_load<P>(pts + (size_t)m * 128));
    if (live_wave) qpoint_store<P>(partials + (size_t)blockIdx.x * 128, t);
    __threadfence();                                               // the point is visible device-wide before the ticket is taken
  }
  if (!fused) return;                                              // (uniform: a kernel argument) k_direct_final follows
  // The group's LAST workgroup to get here adds up the group's workgroup points (what a second launch, k_direct_final,
  // did before: one launch and its start-up less on a path the prover waits on -- the pattern of msm.hip's giant buckets).
  // The counter is left at zero for the next call.
  if (threadIdx.x == 0) ticket = live_wave ? atomicAdd(&arrived[g], 1u) : 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t p0 = g ? a.wg_end[g - 1] : 0u, p1 = a.wg_end[g];
  if (ticket != p1 - p0 - 1u) return;
  __threadfence();
  if (threadIdx.x == 0) arrived[g] = 0u;
  const uint32_t tq = threadIdx.x >> 2;
  QPoint<P> acc2 = qpoint_identity<P>();
  for (uint32_t p = p0 + tq; p < p1; p += 64) acc2 = qpoint_add<P>(acc2, qpoint_load<P>(partials + (size_t)p * 128));
  acc2 = qpoint_wave_sum(acc2);
  __syncthreads();                                                 // (the buffer's earlier readers are all past their loads)
  if ((tq & 15u) == 0) qpoint_store<P>(pts + (size_t)wv * 128, acc2);
  __syncthreads();
  if (tq != 0) return;
  QPoint<P> v = qpoint_load<P>(pts);
#pragma unroll 1
  for (int m = 1; m < 4; ++m) v = qpoint_add<P>(v, qpoint_load<P>(pts + (size_t)m * 128));
  const Jac<P> jj = xyzz_to_jac(qpoint_gather(v));
  if (threadIdx.x == 0) {
    fe_store<P>(out + (size_t)g * 96, jj.x);
    fe_store<P>(out + (size_t)g * 96 + 32, jj.y);
    fe_store<P>(out + (size_t)g * 96 + 64, jj.z);
  }
}

// one workgroup (one wavefront per SIMD) per group: 64 quads stride over the group's workgroup points, a butterfly per
// wavefront, LDS, the last three additions
template <class P>
__global__ __launch_bounds__(256) void k_direct_final(DirectArgs a, const char* __restrict__ partials, char* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) char lds[4 * 128];
  const int g = (int)blockIdx.x;
  const uint32_t p0 = g ? a.wg_end[g - 1] : 0u, p1 = a.wg_end[g];
  const uint32_t t = threadIdx.x >> 2, wv = threadIdx.x >> 6;
  QPoint<P> acc = qpoint_identity<P>();
  for (uint32_t p = p0 + t; p < p1; p += 64) acc = qpoint_add<P>(acc, qpoint_load<P>(partials + (size_t)p * 128));
  acc = qpoint_wave_sum(acc);
  if ((t & 15u) == 0) qpoint_store<P>(lds + (size_t)wv * 128, acc);
  __syncthreads();
  if (t != 0) return;
  QPoint<P> v = qpoint_load<P>(lds);
#pragma unroll 1
  for (int m = 1; m < 4; ++m) v = qpoint_add<P>(v, qpoint_load<P>(lds + (size_t)m * 128));
  const Jac<P> j = xyzz_to_jac(qpoint_gather(v));
  if (threadIdx.x == 0) {
    fe_store<P>(out + (size_t)g * 96, j.x);
    fe_store<P>(out + (size_t)g * 96 + 32, j.y);
    fe_store<P>(out + (size_t)g * 96 + 64, j.z);
  }
}

// Lanes of the launch: one wavefront per SIMD (num_cus x 4), shared out by entry count; at least MIN_PER_LANE entries per
// lane, so a short vector does not pay the reduction of wavefronts that have nothing to add.
static constexpr uint32_t MIN_PER_LANE = 4;
struct DirectGeom { uint32_t per_lane, waves[MSM_MAX_GROUPS], total_wgs; };
static DirectGeom direct_geom(int groups, const size_t* n, int W, int num_cus) {
  DirectGeom d{};
  uint64_t E = 0;
  for (int g = 0; g < groups; ++g) E += (uint64_t)n[g] * W;
  const uint64_t lanes = (uint64_t)num_cus * 256;
  uint64_t per = (E + lanes - 1) / lanes;
  if (per < MIN_PER_LANE) per = MIN_PER_LANE;
  // whole workgroups per group: the rounding can push the launch past one round; a longer slice brings it back
  for (;;) {
    uint64_t wgs = 0;
    for (int g = 0; g < groups; ++g) wgs += ((uint64_t)n[g] * W + per * 256 - 1) / (per * 256);
    if (wgs <= (uint64_t)num_cus || per > (1u << 20)) break;
    ++per;
  }
  d.per_lane = (uint32_t)per;
  for (int g = 0; g < groups; ++g) {
    const uint32_t wgs = (uint32_t)(((uint64_t)n[g] * W + per * 256 - 1) / (per * 256));
    d.waves[g] = wgs * 4;
    d.total_wgs += wgs;
  }
  return d;
}

size_t direct_ws_bytes(int groups, const size_t* n, int c, int num_cus) {
  return ((size_t)direct_geom(groups, n, direct_windows(c), num_cus).total_wgs + 1) * 128;
}

template <class P, class SP>
static Status direct_run_t(int groups, const size_t* n, const size_t* slot0, const void* const* d_scalars, bool is_mont, int c,
                           int num_cus, const void* d_digits, void* ws, void* d_out, uint32_t* arrived, hipStream_t st) {
  DirectArgs a{};
  a.groups = groups;
  const int W = direct_windows(c);
  const DirectGeom geo = direct_geom(groups, n, W, num_cus);
  a.per_lane = geo.per_lane;
  uint32_t waves = 0;
  for (int g = 0; g < groups; ++g) {
    a.scalars[g] = reinterpret_cast<const uint32_t*>(d_scalars[g]);
    a.n[g] = (uint32_t)n[g];
    a.slot0[g] = (uint32_t)slot0[g];
    waves += geo.waves[g];
    a.wave_end[g] = waves;
    a.wg_end[g] = waves / 4;
  }
  for (int l = 0; l < 9; ++l) a.half[l] = 0;
  for (int j = 0; j < W; ++j) { const int bit = c * j + c - 1; a.half[bit >> 5] |= 1u << (bit & 31); }
  double nsum = 0;
  for (int g = 0; g < groups; ++g) nsum += (double)n[g];
  const int fused = (arrived && tuning().direct_fused) ? 1 : 0;
  bool empty_group = false;
  for (int g = 0; g < groups; ++g) empty_group |= geo.waves[g] == 0;
  if (waves) {
    KTimer kt(st, "k_direct_sum", 96.0 * nsum);         // a commitment's algorithmic bytes: 96 B per (base, scalar) pair
    hipLaunchKernelGGL((k_direct_sum<P, SP>), dim3(waves / 4), dim3(256), 0, st, a, is_mont ? 1 : 0, c, W,
                       reinterpret_cast<const char*>(d_digits), reinterpret_cast<char*>(ws), arrived, reinterpret_cast<char*>(d_out),
                       tuning().direct_priority, fused);
  }
  if (empty_group || !fused) {     // a group without scalars has no workgroup to write its identity: the second launch does
    KTimer kt(st, "k_direct_final", 0.0);
    hipLaunchKernelGGL((k_direct_final<P>), dim3(groups), dim3(256), 0, st, a, reinterpret_cast<const char*>(ws),
                       reinterpret_cast<char*>(d_out));
  }
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status msm_direct_run(int curve, int groups, const size_t* n, const size_t* slot0, const void* const* d_scalars, bool is_mont,
                      int c, int num_cus, const void* d_digits, void* ws, void* d_out, uint32_t* arrived, hipStream_t stream) {
  if (curve == VDF_CURVE_PALLAS)
    return direct_run_t<FpParams, FqParams>(groups, n, slot0, d_scalars, is_mont, c, num_cus, d_digits, ws, d_out, arrived, stream);
  if (curve == VDF_CURVE_VESTA)
    return direct_run_t<FqParams, FpParams>(groups, n, slot0, d_scalars, is_mont, c, num_cus, d_digits, ws, d_out, arrived, stream);
  return Status{VDF_ERR_BAD_ARG, "unknown curve"};
}

}  // namespace vdf
