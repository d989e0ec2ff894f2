// Pippenger multi-scalar multiplication over Pallas / Vesta for gfx950.
//
// Replaces pasta-msm 0.1.1's `mult_pippenger_{pallas,vesta}` (Cargo.toml:18), reached by
// every Pedersen commitment of `RecursiveSNARK::prove_step`
// (/root/reference/src/nova/proof.rs:342-349).
//
// Pipeline (all on the device, one stream, no host round trip):
//   k_digits      scalar -> signed c-bit digits (Booth-style carry), one u32 per (window, point)
//   k_hist        per (window, point-chunk) workgroup: bucket histogram in LDS (2^(c-1) u32 counters)
//   k_scan_chunks per bucket: exclusive scan over the chunks that feed it
//   k_scan_keys   exclusive scan over buckets -> bucket start offsets
//   k_scatter     per (window, chunk) workgroup: LDS offset table + ds atomics -> entries sorted by bucket
//   k_accumulate  every thread owns a fixed-length slice of the SORTED entry list and runs a
//                 sequential segmented reduce over it with an XYZZ accumulator in registers
//                 (mixed addition of gathered affine points); perfectly balanced for any scalar
//                 distribution.  Runs that start in the slice go to the bucket array; a run that
//                 continues from the previous slice goes to a per-thread "head" slot.
//   k_fixup       per bucket: add the heads of the slices it spans; buckets spanning many slices are
//                 queued and reduced by a whole wavefront (k_fixup_heavy: strided partial sums, then
//                 a 6-step wavefront butterfly of XYZZ additions).
//   k_reduce1/2   sum_b b*B_b per bucket set (running sums per segment + scalar offset, LDS tree)
//   k_final       Horner over bucket sets, XYZZ -> Jacobian
//
// With a fixed-base table (vdf_bases_precompute) window w = j*sets + s reads table j
// (2^(c*sets*j) * P_i) and feeds bucket set s, so the Horner tail is (sets-1)*c doublings; sets = 1
// removes it entirely.  The table trades HBM capacity (288 GB) for the serial tail.
#include <cstdlib>
#include "internal.h"
#include "ec.cuh"
#include "ecq.cuh"

namespace vdf {

static constexpr uint32_t SIGN_BIT = 0x80000000u;
static constexpr int HEAVY_SPAN = 24;       // slices per bucket above which a wavefront takes over
static constexpr int RED_SEG = 2;           // buckets per thread in k_reduce1 (serial depth 2*RED_SEG)

struct WsLayout {
  size_t dig, counts, bcount, bstart, sorted, bucket_acc, heads, heavy, partials, wsum, total;
  uint32_t red_threads_per_set, red_block, red_blocks_per_set;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static WsLayout ws_layout(const MsmPlan& p) {
  WsLayout w{};
  size_t off = 0;
  const size_t nkeys = (size_t)p.sets * p.nbk;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  w.dig = take((size_t)p.windows * p.n * 4);
  w.counts = take((size_t)p.windows * p.K * p.nbk * 4);
  w.bcount = take(nkeys * 4);
  w.bstart = take((nkeys + 1) * 4);
  w.sorted = take(((size_t)p.windows * p.n + 64 * (size_t)p.L + 64) * 4);
  w.bucket_acc = take(nkeys * 128);
  w.heads = take((size_t)p.nthreads * 128);
  w.heavy = take((nkeys + 4) * 4);
  uint32_t tps = p.nbk / RED_SEG;
  if (tps == 0) tps = 1;
  w.red_threads_per_set = tps;                 // logical threads (quads): one per RED_SEG buckets
  w.red_block = tps < 64 ? tps : 64;           // quads per workgroup (256 lanes)
  w.red_blocks_per_set = tps / w.red_block;
  w.partials = take((size_t)p.sets * w.red_blocks_per_set * 128);
  w.wsum = take((size_t)p.sets * 128);
  w.total = off;
  return w;
}

int msm_auto_window(size_t n) {
  int lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) ++lg;
  int c = lg - 4;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  return c;
}

MsmPlan msm_make_plan(size_t n, int c, int sets, int tables, int num_cus) {
  MsmPlan p;
  p.n = (uint32_t)n;
  p.c = c;
  p.windows = (256 + c - 1) / c;
  if (sets <= 0 || tables <= 0) { sets = p.windows; tables = 1; }
  p.sets = sets;
  p.tables = tables;
  p.nbk = 1u << (c - 1);
  // sort chunks: aim at ~2 workgroups per CU over all windows, 4096 <= chunk <= 65536 points
  size_t target_blocks = (size_t)num_cus * 2;
  size_t chunk = ((size_t)n * p.windows + target_blocks - 1) / target_blocks;
  if (chunk < 4096) chunk = 4096;
  if (chunk > 65536) chunk = 65536;
  p.chunk = (uint32_t)chunk;
  p.K = (uint32_t)((n + chunk - 1) / chunk);
  if (p.K == 0) p.K = 1;
  // accumulate slices: ~4 waves per SIMD worth of threads (k_accumulate is resident at 3 per SIMD; measured on
  // MI355X, L = 32..64 is the flat optimum at 2^18..2^20: shorter slices multiply the slice heads k_fixup must
  // add, longer ones leave a thin last round), 32 <= L <= 64, multiple of 4.
  size_t ne = (size_t)n * p.windows;
  size_t want_threads = (size_t)num_cus * 4 * 4 * 64;
  size_t L = (ne + want_threads - 1) / want_threads;
  L = (L + 3) / 4 * 4;
  if (L < 32) L = 32;
  if (L > 64) L = 64;
  if (const char* ov = std::getenv("VDF_MSM_L")) { long v = std::atol(ov); if (v >= 4 && v <= 4096) L = (size_t)(v + 3) / 4 * 4; }   // tuning override
  p.L = (uint32_t)L;
  p.nthreads = (uint32_t)((ne + L - 1) / L);
  if (p.nthreads == 0) p.nthreads = 1;
  p.ws_bytes = ws_layout(p).total;
  return p;
}

// ------------------------------------------------------------------------------------------
// digits
// ------------------------------------------------------------------------------------------
template <class SP>
__global__ __launch_bounds__(256) void k_digits(const uint32_t* __restrict__ scalars, uint32_t n, int is_mont, int c,
                                                int windows, uint32_t* __restrict__ dig) {
  __shared__ uint32_t limbs[9 * 256];     // SoA: limb l of thread t at [l*256 + t]; limb 8 = 0
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    Fe<SP> s = fe_load<SP>(scalars + (size_t)i * 8);
    if (is_mont) s = fe_from_mont(s);
#pragma unroll
    for (int l = 0; l < 8; ++l) limbs[l * 256 + threadIdx.x] = s.v[l];
    limbs[8 * 256 + threadIdx.x] = 0;
  }
  __syncthreads();
  if (i >= n) return;
  const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
  uint32_t carry = 0;
  for (int w = 0; w < windows; ++w) {
    const int bit = w * c;
    const int l = bit >> 5, sh = bit & 31;
    uint32_t lo = (l < 8) ? limbs[l * 256 + threadIdx.x] : 0u;
    uint32_t hi = (l + 1 < 9) ? limbs[(l + 1) * 256 + threadIdx.x] : 0u;
    uint64_t two = ((uint64_t)hi << 32) | lo;
    uint32_t raw = ((uint32_t)(two >> sh) & mask) + carry;
    uint32_t out;
    if (raw > half) { out = ((1u << c) - raw) | SIGN_BIT; carry = 1; }
    else { out = raw; carry = 0; }
    dig[(size_t)w * n + i] = out;
  }
}

// ------------------------------------------------------------------------------------------
// counting sort by bucket (LDS histograms)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_hist(const uint32_t* __restrict__ dig, uint32_t n, uint32_t nbk,
                                               uint32_t chunk, uint32_t K, uint32_t* __restrict__ counts) {
  extern __shared__ uint32_t lds[];
  const uint32_t w = blockIdx.x / K, k = blockIdx.x % K;
  for (uint32_t b = threadIdx.x; b < nbk; b += blockDim.x) lds[b] = 0;
  __syncthreads();
  const uint32_t lo = k * chunk;
  const uint32_t hi = (lo + chunk < n) ? lo + chunk : n;
  const uint32_t* d = dig + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint32_t mag = d[i] & ~SIGN_BIT;
    if (mag) atomicAdd(&lds[mag - 1], 1u);
  }
  __syncthreads();
  uint32_t* out = counts + (size_t)blockIdx.x * nbk;
  for (uint32_t b = threadIdx.x; b < nbk; b += blockDim.x) out[b] = lds[b];
}

// one thread per bucket key (set s, bucket b): exclusive scan over its feeding chunks
// (counts[chunk][bucket]: coalesced across the threads of a wave; eight chunks are loaded before
// their dependent stores so the loop is not one L2 round trip per chunk)
__global__ __launch_bounds__(256) void k_scan_chunks(uint32_t* __restrict__ counts, uint32_t nbk, uint32_t K, int sets,
                                                     int tables, uint32_t* __restrict__ bcount) {
  const uint32_t key = blockIdx.x * 256 + threadIdx.x;
  if (key >= (uint32_t)sets * nbk) return;
  const uint32_t s = key / nbk, b = key % nbk;
  uint32_t run = 0;
  for (int j = 0; j < tables; ++j) {
    const uint32_t w = (uint32_t)j * sets + s;
    uint32_t* base = counts + (size_t)w * K * nbk + b;
    uint32_t k = 0;
    for (; k + 8 <= K; k += 8) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(k + u) * nbk];
#pragma unroll
      for (int u = 0; u < 8; ++u) { base[(size_t)(k + u) * nbk] = run; run += v[u]; }
    }
    for (; k < K; ++k) {
      uint32_t v = base[(size_t)k * nbk];
      base[(size_t)k * nbk] = run;
      run += v;
    }
  }
  bcount[key] = run;
}

// single workgroup exclusive scan: bstart[0..nkeys], bstart[nkeys] = total entries.
// Each thread owns a contiguous run of keys; loads are issued eight at a time so the run is a few
// L2 round trips instead of one per key.
__global__ __launch_bounds__(1024) void k_scan_keys(const uint32_t* __restrict__ bcount, uint32_t nkeys,
                                                    uint32_t* __restrict__ bstart) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (nkeys + 1023) / 1024;
  const uint32_t lo = threadIdx.x * per;
  const uint32_t hi = (lo + per < nkeys) ? lo + per : nkeys;
  uint32_t sum = 0;
  uint32_t i = lo;
  for (; i + 8 <= hi; i += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = bcount[i + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; i < hi; ++i) sum += bcount[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = (threadIdx.x >= d) ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  i = lo;
  for (; i + 8 <= hi; i += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = bcount[i + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) { bstart[i + u] = run; run += v[u]; }
  }
  for (; i < hi; ++i) { bstart[i] = run; run += bcount[i]; }
  if (threadIdx.x == 1023) bstart[nkeys] = part[1023];
}

__global__ __launch_bounds__(1024) void k_scatter(const uint32_t* __restrict__ dig, uint32_t n, uint32_t nbk,
                                                  uint32_t chunk, uint32_t K, int sets, uint32_t tstride, uint32_t L,
                                                  const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ bstart, uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lds[];
  // XCD-aware block -> chunk map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an
  // XCD, and with it an L2).  The chunks of one window write neighbouring positions of every bucket's
  // region, so giving one XCD a run of CONSECUTIVE chunks lets its L2 merge their 4-byte scatter writes
  // into whole lines before they leave for HBM.  Placement only affects speed, never correctness.
  uint32_t bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t w = bid / K, k = bid % K;
  const uint32_t s = w % (uint32_t)sets, j = w / (uint32_t)sets;
  const uint32_t* cnt = counts + (size_t)bid * nbk;
  const uint32_t* bs = bstart + (size_t)s * nbk;
  for (uint32_t b = threadIdx.x; b < nbk; b += blockDim.x) lds[b] = cnt[b] + bs[b];
  __syncthreads();
  const uint32_t lo = k * chunk;
  const uint32_t hi = (lo + chunk < n) ? lo + chunk : n;
  const uint32_t* d = dig + (size_t)w * n;
  const uint32_t src_base = j * tstride;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint32_t v = d[i];
    uint32_t mag = v & ~SIGN_BIT;
    if (mag) {
      uint32_t pos = atomicAdd(&lds[mag - 1], 1u);
      // lane-interleaved layout: the slice of accumulate-thread t = pos / L lives at stride 64 inside
      // its wave's 64*L block, so a wave reads entry k of all its lanes as one 256-byte line
      const uint32_t wv = pos / (64u * L), within = pos % (64u * L);
      sorted[(size_t)wv * 64u * L + (within % L) * 64u + within / L] = (src_base + i) | (v & SIGN_BIT);
    }
  }
}

// ------------------------------------------------------------------------------------------
// bucket accumulation: sequential segmented reduce over fixed-length slices of the sorted list
// ------------------------------------------------------------------------------------------
// Mixed addition acc += b in the lazy domain (fe.cuh): coordinates of acc in [0, 2m + eps), b canonical.
// `have` says whether acc holds a point yet (the identity has no lazy encoding).
template <class P>
__device__ __forceinline__ void madd_lazy(XYZZ<P>& acc, bool& have, const Affine<P>& b) {
  if (affine_is_identity(b)) return;
  if (!have) { acc = xyzz_from_affine(b); have = true; return; }
  const Fe<P> U2 = fe_mul_lazy(b.x, acc.zz);
  const Fe<P> S2 = fe_mul_lazy(b.y, acc.zzz);
  const Fe<P> Pp = fe_sub_lazy(U2, acc.x);
  const Fe<P> Rr = fe_sub_lazy(S2, acc.y);
  // P == 0 (mod m) means P in {0, m, 2m}; m == 1 (mod 2^32), so the low limb is 0, 1 or 2: cheap filter
  if (Pp.v[0] <= 2u && fe_is_zero(fe_canon(Pp))) {
    if (fe_is_zero(fe_canon(Rr))) acc = xyzz_dbl_affine(b);        // same point: double (canonical output)
    else have = false;                                              // opposite points: identity
    return;
  }
  const Fe<P> PP = fe_mul_lazy(Pp, Pp);
  const Fe<P> PPP = fe_mul_lazy(Pp, PP);
  const Fe<P> Qq = fe_mul_lazy(acc.x, PP);
  const Fe<P> X3 = fe_sub_lazy(fe_sub_lazy(fe_sub_lazy(fe_mul_lazy(Rr, Rr), PPP), Qq), Qq);
  const Fe<P> Y3 = fe_sub_lazy(fe_mul_lazy(Rr, fe_sub_lazy(Qq, X3)), fe_mul_lazy(acc.y, PPP));
  acc.x = X3;
  acc.y = Y3;
  acc.zz = fe_mul_lazy(acc.zz, PP);
  acc.zzz = fe_mul_lazy(acc.zzz, PPP);
}

template <class P>
__device__ __forceinline__ void flush_lazy(const XYZZ<P>& acc, bool have, char* dst) {
  XYZZ<P> o;
  if (have) { o.x = fe_canon(acc.x); o.y = fe_canon(acc.y); o.zz = fe_canon(acc.zz); o.zzz = fe_canon(acc.zzz); }
  else o = xyzz_identity<P>();
  xyzz_store<P>(dst, o);
}

template <class P>
__global__ __launch_bounds__(256) void k_accumulate(const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ bstart, uint32_t nkeys,
                                                    const char* __restrict__ points, char* __restrict__ bucket_acc,
                                                    char* __restrict__ heads, uint32_t L, uint32_t nthreads) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nthreads) return;
  const uint32_t ne = bstart[nkeys];
  const uint64_t lo64 = (uint64_t)t * L;
  if (lo64 >= ne) return;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = (lo64 + L < ne) ? lo + L : ne;
  // bucket containing position lo: bstart[g] <= lo < bstart[g+1]
  uint32_t a = 0, b = nkeys;
  while (b - a > 1) {
    uint32_t mid = (a + b) >> 1;
    if (bstart[mid] <= lo) a = mid; else b = mid;
  }
  uint32_t g = a;
  uint32_t next = bstart[g + 1];
  bool is_head = bstart[g] < lo;
  XYZZ<P> acc = xyzz_identity<P>();
  bool have = false;
  const uint32_t* mine = sorted + (size_t)(t >> 6) * 64u * L + (t & 63u);   // entry k of this slice: mine[k * 64]
  uint32_t e = mine[0];
  Affine<P> pt = affine_load<P>(points + (size_t)(e & ~SIGN_BIT) * 64);
  for (uint32_t pos = lo; pos < hi; ++pos) {
    // prefetch the next entry's point while this one is added
    const uint32_t pn = (pos + 1 < hi) ? pos + 1 : pos;
    const uint32_t en = mine[(size_t)(pn - lo) * 64u];
    Affine<P> ptn = affine_load<P>(points + (size_t)(en & ~SIGN_BIT) * 64);
    if (pos >= next) {
      flush_lazy<P>(acc, have, is_head ? heads + (size_t)t * 128 : bucket_acc + (size_t)g * 128);
      is_head = false;
      have = false;
      ++g;
      if (bstart[g + 1] <= pos) {   // empty buckets in between: binary search
        uint32_t x = g, y = nkeys;
        while (y - x > 1) {
          uint32_t mid = (x + y) >> 1;
          if (bstart[mid] <= pos) x = mid; else y = mid;
        }
        g = x;
      }
      next = bstart[g + 1];
    }
    if (e & SIGN_BIT) pt.y = fe_neg(pt.y);
    madd_lazy<P>(acc, have, pt);
    e = en;
    pt = ptn;
  }
  flush_lazy<P>(acc, have, is_head ? heads + (size_t)t * 128 : bucket_acc + (size_t)g * 128);
}

// ------------------------------------------------------------------------------------------
// Tail of the pipeline.  Every kernel below is a short chain of dependent point additions run by few
// waves, so they use the quad-cooperative group law of ecq.cuh: four lanes per point, four
// multiplication stages per addition instead of fourteen serial multiplications.
// "Logical thread" = quad = (global lane index) / 4.
// ------------------------------------------------------------------------------------------
// One quad per bucket: add the heads of the slices the bucket spans (table mode: ~8 per bucket).
template <class P>
__global__ __launch_bounds__(256) void k_fixup(const uint32_t* __restrict__ bstart, uint32_t nkeys, uint32_t L,
                                               char* __restrict__ bucket_acc, const char* __restrict__ heads,
                                               uint32_t* __restrict__ heavy) {
  const uint32_t g = (blockIdx.x * 256 + threadIdx.x) >> 2;
  if (g >= nkeys) return;                                          // quad-uniform from here on
  const uint32_t s = bstart[g], e = bstart[g + 1];
  if (e <= s) return;
  const uint32_t tf = s / L, tl = (e - 1) / L;
  if (tl == tf) return;
  if (tl - tf > (uint32_t)HEAVY_SPAN) {
    if ((threadIdx.x & 3u) == 0) {
      uint32_t slot = atomicAdd(&heavy[0], 1u);
      heavy[1 + slot] = g;
    }
    return;
  }
  QPoint<P> acc = qpoint_load<P>(bucket_acc + (size_t)g * 128);
  QPoint<P> nxt = qpoint_load<P>(heads + (size_t)(tf + 1) * 128);
  for (uint32_t t = tf + 1; t <= tl; ++t) {
    const QPoint<P> cur = nxt;
    if (t < tl) nxt = qpoint_load<P>(heads + (size_t)(t + 1) * 128);   // next head in flight during the addition
    acc = qpoint_add<P>(acc, cur);
  }
  qpoint_store<P>(bucket_acc + (size_t)g * 128, acc);
}

// One wavefront (16 quads) per queued heavy bucket: quads stride over the bucket's heads, then a
// 4-step butterfly of quad additions across the wavefront.
template <class P>
__global__ __launch_bounds__(64) void k_fixup_heavy(const uint32_t* __restrict__ bstart, uint32_t L,
                                                    char* __restrict__ bucket_acc, const char* __restrict__ heads,
                                                    const uint32_t* __restrict__ heavy) {
  const uint32_t count = heavy[0];
  const uint32_t quad = threadIdx.x >> 2;
  for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    const uint32_t g = heavy[1 + item];
    const uint32_t s = bstart[g], e = bstart[g + 1];
    const uint32_t tf = s / L, tl = (e - 1) / L;
    QPoint<P> acc = qpoint_identity<P>();
    for (uint32_t t = tf + 1 + quad; t <= tl; t += 16) acc = qpoint_add<P>(acc, qpoint_load<P>(heads + (size_t)t * 128));
    acc = qpoint_wave_sum(acc);
    if (quad == 0) {
      QPoint<P> base = qpoint_load<P>(bucket_acc + (size_t)g * 128);
      qpoint_store<P>(bucket_acc + (size_t)g * 128, qpoint_add<P>(base, acc));
    }
  }
}

// ------------------------------------------------------------------------------------------
// bucket reduction  sum_{b} (b+1) * B[b]  per set
// ------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void k_reduce1(const char* __restrict__ bucket_acc, uint32_t nbk,
                                                 uint32_t threads_per_set, uint32_t blocks_per_set,
                                                 char* __restrict__ partials) {
  extern __shared__ __align__(16) char lds_raw[];
  const uint32_t nlog = blockDim.x >> 2;                          // logical threads (quads) per block
  const uint32_t lt = threadIdx.x >> 2;
  const uint32_t set = blockIdx.x / blocks_per_set;
  const uint32_t blk = blockIdx.x % blocks_per_set;
  const uint32_t seg = blk * nlog + lt;                            // segment index within the set
  const uint32_t nseg = (nbk < (uint32_t)RED_SEG) ? nbk : (uint32_t)RED_SEG;
  const uint32_t base = seg * nseg;                                // first bucket of the segment
  QPoint<P> run = qpoint_identity<P>(), tot = qpoint_identity<P>();
  if (seg < threads_per_set) {
    const char* bp = bucket_acc + ((size_t)set * nbk + base) * 128;
    for (int l = (int)nseg - 1; l >= 0; --l) {
      run = qpoint_add<P>(run, qpoint_load<P>(bp + (size_t)l * 128));
      tot = qpoint_add<P>(tot, run);                               // tot = sum (l+1) * B[base+l]
    }
    if (base) {                                                    // + base * run (double-and-add, base < 2^16)
      QPoint<P> m = qpoint_identity<P>();
      for (int bit = 31 - __builtin_clz(base); bit >= 0; --bit) {
        m = qpoint_dbl<P>(m);
        if ((base >> bit) & 1u) m = qpoint_add<P>(m, run);
      }
      tot = qpoint_add<P>(tot, m);
    }
  }
  // workgroup tree reduction through LDS (each lane moves its own 32-byte coordinate)
  qpoint_store<P>(lds_raw + (size_t)lt * 128, tot);
  __syncthreads();
  for (uint32_t stride = nlog >> 1; stride >= 1; stride >>= 1) {
    if (lt < stride) {
      tot = qpoint_add<P>(tot, qpoint_load<P>(lds_raw + (size_t)(lt + stride) * 128));
      qpoint_store<P>(lds_raw + (size_t)lt * 128, tot);
    }
    __syncthreads();
  }
  if (lt == 0) qpoint_store<P>(partials + (size_t)blockIdx.x * 128, tot);
}

// one workgroup of 64 quads per set: strided partial sums, then an LDS tree
template <class P>
__global__ __launch_bounds__(256) void k_reduce2(const char* __restrict__ partials, uint32_t blocks_per_set,
                                                 char* __restrict__ wsum) {
  __shared__ __align__(16) char lds_raw[64 * 128];
  const uint32_t set = blockIdx.x;
  const uint32_t lt = threadIdx.x >> 2;
  QPoint<P> acc = qpoint_identity<P>();
  for (uint32_t i = lt; i < blocks_per_set; i += 64)
    acc = qpoint_add<P>(acc, qpoint_load<P>(partials + ((size_t)set * blocks_per_set + i) * 128));
  qpoint_store<P>(lds_raw + (size_t)lt * 128, acc);
  __syncthreads();
  for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
    if (lt < stride) {
      acc = qpoint_add<P>(acc, qpoint_load<P>(lds_raw + (size_t)(lt + stride) * 128));
      qpoint_store<P>(lds_raw + (size_t)lt * 128, acc);
    }
    __syncthreads();
  }
  if (lt == 0) qpoint_store<P>(wsum + (size_t)set * 128, acc);
}

// Horner over bucket sets (one quad), XYZZ -> Jacobian
template <class P>
__global__ __launch_bounds__(64) void k_final(const char* __restrict__ wsum, int sets, int c, char* __restrict__ out_jac) {
  if (blockIdx.x != 0 || threadIdx.x >= 4) return;
  QPoint<P> acc = qpoint_identity<P>();
  for (int s = sets - 1; s >= 0; --s) {
    if (s != sets - 1)
      for (int k = 0; k < c; ++k) acc = qpoint_dbl<P>(acc);
    acc = qpoint_add<P>(acc, qpoint_load<P>(wsum + (size_t)s * 128));
  }
  const Jac<P> j = xyzz_to_jac(qpoint_gather(acc));
  if (threadIdx.x == 0) {
    fe_store<P>(out_jac, j.x);
    fe_store<P>(out_jac + 32, j.y);
    fe_store<P>(out_jac + 64, j.z);
  }
}

// ------------------------------------------------------------------------------------------
// generators
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class P>
__global__ __launch_bounds__(256) void k_bases_generate(uint64_t seed, uint64_t start, uint32_t n, char* __restrict__ pts) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = splitmix64(seed * 0xD1342543DE82EF95ull + (start + i)) | 1ull;
  Affine<P> g;
#pragma unroll
  for (int l = 0; l < 8; ++l) { g.x.v[l] = P::GEN_X[l]; g.y.v[l] = P::GEN_Y[l]; }
  XYZZ<P> r = xyzz_mul_u64(g, k);
  affine_store<P>(pts + (size_t)i * 64, xyzz_to_affine(r));
}

// table[j][i] = 2^(shift*j) * P_i
template <class P>
__global__ __launch_bounds__(256) void k_precompute(const char* __restrict__ pts, uint32_t n, int shift, int tables,
                                                    char* __restrict__ table) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Affine<P> a = affine_load<P>(pts + (size_t)i * 64);
  affine_store<P>(table + (size_t)i * 64, a);
  XYZZ<P> r = xyzz_from_affine(a);
  for (int j = 1; j < tables; ++j) {
    for (int k = 0; k < shift; ++k) r = xyzz_dbl(r);
    Affine<P> o = xyzz_to_affine(r);
    affine_store<P>(table + ((size_t)j * n + i) * 64, o);
    r = xyzz_from_affine(o);          // keep zz = 1 so the next doublings stay cheap
  }
}

// ------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------
template <class P, class SP>
static Status msm_run_t(const MsmPlan& p, const void* d_points, const void* d_scalars, bool is_mont, void* ws,
                        void* d_out, hipStream_t st, hipEvent_t* ev) {
  const WsLayout w = ws_layout(p);
  char* base = reinterpret_cast<char*>(ws);
  uint32_t* dig = reinterpret_cast<uint32_t*>(base + w.dig);
  uint32_t* counts = reinterpret_cast<uint32_t*>(base + w.counts);
  uint32_t* bcount = reinterpret_cast<uint32_t*>(base + w.bcount);
  uint32_t* bstart = reinterpret_cast<uint32_t*>(base + w.bstart);
  uint32_t* sorted = reinterpret_cast<uint32_t*>(base + w.sorted);
  char* bucket_acc = base + w.bucket_acc;
  char* heads = base + w.heads;
  uint32_t* heavy = reinterpret_cast<uint32_t*>(base + w.heavy);
  char* partials = base + w.partials;
  char* wsum = base + w.wsum;
  const uint32_t nkeys = (uint32_t)p.sets * p.nbk;
  const size_t lds_sort = (size_t)p.nbk * 4;

  if (lds_sort > 64 * 1024) {
    VDF_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
    VDF_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
  }
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[0], st));
  VDF_TRY_HIP(hipMemsetAsync(bucket_acc, 0, (size_t)nkeys * 128, st));
  VDF_TRY_HIP(hipMemsetAsync(heavy, 0, 4, st));
  hipLaunchKernelGGL((k_digits<SP>), dim3((p.n + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<const uint32_t*>(d_scalars), p.n, is_mont ? 1 : 0, p.c, p.windows, dig);
  hipLaunchKernelGGL(k_hist, dim3(p.windows * p.K), dim3(1024), lds_sort, st, dig, p.n, p.nbk, p.chunk, p.K, counts);
  hipLaunchKernelGGL(k_scan_chunks, dim3((nkeys + 255) / 256), dim3(256), 0, st, counts, p.nbk, p.K, p.sets, p.tables,
                     bcount);
  hipLaunchKernelGGL(k_scan_keys, dim3(1), dim3(1024), 0, st, bcount, nkeys, bstart);
  hipLaunchKernelGGL(k_scatter, dim3(p.windows * p.K), dim3(1024), lds_sort, st, dig, p.n, p.nbk, p.chunk, p.K, p.sets,
                     p.tstride, p.L, counts, bstart, sorted);
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[1], st));
  hipLaunchKernelGGL((k_accumulate<P>), dim3((p.nthreads + 255) / 256), dim3(256), 0, st, sorted, bstart, nkeys,
                     reinterpret_cast<const char*>(d_points), bucket_acc, heads, p.L, p.nthreads);
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[2], st));
  hipLaunchKernelGGL((k_fixup<P>), dim3((nkeys * 4 + 255) / 256), dim3(256), 0, st, bstart, nkeys, p.L, bucket_acc, heads,
                     heavy);
  hipLaunchKernelGGL((k_fixup_heavy<P>), dim3(1024), dim3(64), 0, st, bstart, p.L, bucket_acc, heads, heavy);
  hipLaunchKernelGGL((k_reduce1<P>), dim3(p.sets * w.red_blocks_per_set), dim3(w.red_block * 4),
                     (size_t)w.red_block * 128, st, bucket_acc, p.nbk, w.red_threads_per_set, w.red_blocks_per_set,
                     partials);
  hipLaunchKernelGGL((k_reduce2<P>), dim3(p.sets), dim3(256), 0, st, partials, w.red_blocks_per_set, wsum);
  hipLaunchKernelGGL((k_final<P>), dim3(1), dim3(64), 0, st, wsum, p.sets, p.c, reinterpret_cast<char*>(d_out));
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[3], st));
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status msm_run(int curve, const MsmPlan& plan, const void* d_points, const void* d_scalars, bool is_mont, void* ws,
               void* d_out, hipStream_t stream, hipEvent_t* ev) {
  // Pallas: coordinates in Fp, scalars in Fq.  Vesta: coordinates in Fq, scalars in Fp.
  if (curve == VDF_CURVE_PALLAS)
    return msm_run_t<FpParams, FqParams>(plan, d_points, d_scalars, is_mont, ws, d_out, stream, ev);
  if (curve == VDF_CURVE_VESTA)
    return msm_run_t<FqParams, FpParams>(plan, d_points, d_scalars, is_mont, ws, d_out, stream, ev);
  return Status{VDF_ERR_BAD_ARG, "unknown curve"};
}

// sum of n Jacobian points: one wavefront = 16 quads striding over the inputs, butterfly reduce
template <class P>
__global__ __launch_bounds__(64) void k_point_sum(const char* __restrict__ pts, uint32_t n, char* __restrict__ out) {
  const uint32_t quad = threadIdx.x >> 2;
  QPoint<P> acc = qpoint_identity<P>();
  for (uint32_t i = quad; i < n; i += 16) {
    Jac<P> j;
    j.x = fe_load<P>(pts + (size_t)i * 96);
    j.y = fe_load<P>(pts + (size_t)i * 96 + 32);
    j.z = fe_load<P>(pts + (size_t)i * 96 + 64);
    acc = qpoint_add<P>(acc, qpoint_scatter(jac_to_xyzz(j)));
  }
  acc = qpoint_wave_sum(acc);
  const Jac<P> j = xyzz_to_jac(qpoint_gather(acc));
  if (threadIdx.x == 0) {
    fe_store<P>(out, j.x);
    fe_store<P>(out + 32, j.y);
    fe_store<P>(out + 64, j.z);
  }
}

Status point_sum(int curve, const void* d_jac, size_t n, void* d_out, hipStream_t stream) {
  if (curve == VDF_CURVE_PALLAS)
    hipLaunchKernelGGL((k_point_sum<FpParams>), dim3(1), dim3(64), 0, stream, reinterpret_cast<const char*>(d_jac),
                       (uint32_t)n, reinterpret_cast<char*>(d_out));
  else if (curve == VDF_CURVE_VESTA)
    hipLaunchKernelGGL((k_point_sum<FqParams>), dim3(1), dim3(64), 0, stream, reinterpret_cast<const char*>(d_jac),
                       (uint32_t)n, reinterpret_cast<char*>(d_out));
  else
    return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status bases_generate(int curve, uint64_t seed, size_t start, size_t n, void* d_pts, hipStream_t stream) {
  if (n == 0) return Status{};
  dim3 grid((unsigned)((n + 255) / 256));
  if (curve == VDF_CURVE_PALLAS)
    hipLaunchKernelGGL((k_bases_generate<FpParams>), grid, dim3(256), 0, stream, seed, (uint64_t)start, (uint32_t)n,
                       reinterpret_cast<char*>(d_pts));
  else if (curve == VDF_CURVE_VESTA)
    hipLaunchKernelGGL((k_bases_generate<FqParams>), grid, dim3(256), 0, stream, seed, (uint64_t)start, (uint32_t)n,
                       reinterpret_cast<char*>(d_pts));
  else
    return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status bases_precompute(int curve, const void* d_pts, size_t n, int c, int sets, int tables, void* d_table,
                        hipStream_t stream) {
  if (n == 0) return Status{};
  dim3 grid((unsigned)((n + 255) / 256));
  const int shift = c * sets;
  if (curve == VDF_CURVE_PALLAS)
    hipLaunchKernelGGL((k_precompute<FpParams>), grid, dim3(256), 0, stream, reinterpret_cast<const char*>(d_pts),
                       (uint32_t)n, shift, tables, reinterpret_cast<char*>(d_table));
  else if (curve == VDF_CURVE_VESTA)
    hipLaunchKernelGGL((k_precompute<FqParams>), grid, dim3(256), 0, stream, reinterpret_cast<const char*>(d_pts),
                       (uint32_t)n, shift, tables, reinterpret_cast<char*>(d_table));
  else
    return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

}  // namespace vdf
