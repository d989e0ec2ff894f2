// Pippenger multi-scalar multiplication over Pallas / Vesta for gfx950.
//
// Replaces pasta-msm 0.1.1's `mult_pippenger_{pallas,vesta}` (Cargo.toml:18), reached by
// every Pedersen commitment of `RecursiveSNARK::prove_step`
// (/root/reference/src/nova/proof.rs:342-349).
//
// A call handles a batch of 1..4 MSMs ("groups") over one generator table; to everything after pass A a group is
// just more bucket sets.  Pipeline (all on the device, one stream, no host round trip, 10 launches):
//   Two-level counting sort of the (window, point) entries by bucket, b = (p << fb) | f:
//   k_part<hist> / k_part_scan / k_scan_keys / k_part_staged (k_part<scatter>)   pass A: every workgroup turns a chunk of one
//                 group's scalars into signed c-bit digits (Booth-style carry) on the fly and partitions the
//                 entries by group, bucket set and the high bucket bits p (about one partition per CU, cursors
//                 in LDS); each (workgroup, partition) run is contiguous; a record is a 4-byte payload (point | table |
//                 sign) and a 2-byte fine key in two arrays, and a tile of records is laid out partition by partition
//                 in LDS first so that it leaves as whole pieces (k_part_staged; round 5)
//   k_fine_staged (k_fine)   pass B: one 1024-thread workgroup per partition: histogram of the low bits f in LDS, scan
//                 (= the partition's bucket starts), scatter into the sorted list tile by tile through LDS; it also zeroes the
//                 accumulator of every empty bucket (no memset) and tells each k_accumulate thread its first bucket
//   k_accumulate  every thread owns a fixed-length slice of the SORTED entry list and runs a
//                 sequential segmented reduce over it with an XYZZ accumulator in registers
//                 (mixed addition of gathered affine points, lazy domain); perfectly balanced for any scalar
//                 distribution.  Runs that start in the slice go to the bucket array; a run that
//                 continues from the previous slice goes to a per-thread "head" slot.  The grid exactly fills
//                 the resident workgroup slots in one round (msm_make_plan).
//   k_fixup       per bucket (one quad): add the heads of the slices it spans; buckets spanning many slices are
//                 queued and reduced by a whole wavefront (k_fixup_heavy: strided partial sums, then
//                 a butterfly of quad additions).
//   k_red_sums/weights/combine   sum_b (b+1)*B_b per bucket set: row and column sums of the bucket matrix, small
//                 multiplications of those, one workgroup per set (k_reduce1/2: the first version, for tiny sets)
//   k_final       per group: Horner over its bucket sets, XYZZ -> Jacobian
// The tail kernels use the quad-cooperative group law of ecq.cuh.  msm_run with an external bucket array stops
// after the fix-up; msm_tail then reduces the groups of an MSM job (vdf_msm_job_*) together.
//
// With a fixed-base table (vdf_bases_precompute) window w = j*sets + s reads table j
// (2^(c*sets*j) * P_i) and feeds bucket set s, so the Horner tail is (sets-1)*c doublings; sets = 1
// removes it entirely.  The table trades HBM capacity (288 GB) for the serial tail.
#include <cstdlib>
#include <cstring>
#include "internal.h"
#include "ec.cuh"
#include "ecq.cuh"

namespace vdf {

static constexpr uint32_t SIGN_BIT = 0x80000000u;

// Every kernel of the pipeline except k_accumulate is short or latency-bound; k_accumulate is a long pure-ALU
// grid that keeps every SIMD's issue port busy.  When two MSMs run on two streams (prove_step commits W and T
// side by side) the light kernels of one would starve behind the other's accumulate waves, which the
// oldest-first arbiter favours; a raised wave priority lets them through.
// Wave priority of the pipeline's light kernels (sort, fix-up, bucket reduction): above the bucket accumulation (0); a
// context lowers its own (vdf_ctx_set_light_priority), the process-wide ceiling is vdf_hip_tuning.light_priority -- both
// are folded into the `wave_prio` argument on the host (msm_run).
__device__ __forceinline__ void raise_wave_priority(int wave_prio = 3) {
  if (wave_prio >= 3) __builtin_amdgcn_s_setprio(3);
  else if (wave_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (wave_prio == 1) __builtin_amdgcn_s_setprio(1);
}
static constexpr int ACC_WG_PER_CU = 3;     // resident k_accumulate workgroups per CU (VGPR budget)
static constexpr int ACC_WG_FILL = 2;       // ... of which one round fills this many below 2^22 entries and in batches: a third fewer slices means a
                                            // third fewer slice heads for k_fixup to add (large single MSMs fill all three: msm_make_plan)
// The fix-up is a chain of dependent additions per bucket, so its duration is the LONGEST chain of the launch (a quad
// addition is ~3 us): a bucket spanning more than heavy_span slices goes to a wavefront (chain span/16 + 5), one
// spanning more than GIANT_SPAN to several wavefronts, GIANT_CHUNK heads each.  heavy_span adapts to the launch
// (k_fixup): twice the average span + 4, never below HEAVY_MIN -- a 0/1-heavy witness of a few thousand scalars puts
// hundreds of slice heads into ONE bucket while the average bucket spans one or two slices.
static constexpr uint32_t HEAVY_MIN = 6;
static constexpr uint32_t GIANT_SPAN = 64;
static constexpr uint32_t GIANT_CHUNK = 32;  // heads per wavefront of a shared bucket (two per quad), while the parts last
static constexpr uint32_t GIANT_PARTS = 256; // wavefronts per shared bucket at most
static constexpr uint32_t MAX_GIANTS = 192;  // shared buckets per launch; the surplus is reduced by one wavefront each
__device__ __forceinline__ uint32_t giant_parts(uint32_t span) {
  const uint32_t p = (span + GIANT_CHUNK - 1) / GIANT_CHUNK;
  return p > GIANT_PARTS ? GIANT_PARTS : p;
}

// Slice length of k_accumulate, computed on the device from the ACTUAL entry count (zero digits are skipped, so
// scalars with few non-zero windows produce far fewer entries than n * windows): one round over `slots` threads,
// at least 8 entries per slice.  `fixed` != 0 (tuning override) wins.
__device__ __forceinline__ uint32_t slice_len(uint32_t ne, uint32_t slots, uint32_t fixed) {
  if (fixed) return fixed;
  const uint32_t L = (ne + slots - 1) / slots;
  return L < 8u ? 8u : L;
}
static constexpr uint32_t RED_QUADS = 8192;  // k_reduce1 quads aimed for: enough to fill the chip, few enough that the
                                             // per-quad offset multiplication (~30 point ops) stays a small share

struct WsLayout {
  size_t countsA, pcount, pstart, recs, bcount, bstart, tstart, sorted, bucket_acc, heads, heavy, giant, partials, wsum, total;
  uint32_t red_seg, red_threads_per_set, red_block, red_blocks_per_set;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct RedGeom { uint32_t seg, threads_per_set, block, blocks_per_set; };
static RedGeom red_geom(size_t nkeys, uint32_t nbk) {
  RedGeom r;
  uint32_t seg = 2;                            // buckets per quad in k_reduce1 (serial depth 2*seg), power of two
  const uint32_t quads = tuning().reduction_quads ? (uint32_t)tuning().reduction_quads : RED_QUADS;
  while (seg < 64 && nkeys / seg > quads) seg <<= 1;
  if (seg > nbk) seg = nbk;
  r.seg = seg;
  r.threads_per_set = nbk / seg;               // logical threads (quads): one per seg buckets
  r.block = r.threads_per_set < 64 ? r.threads_per_set : 64;   // quads per workgroup (256 lanes)
  r.blocks_per_set = r.threads_per_set / r.block;
  return r;
}

// Bucket reduction as a matrix (k_red_sums / k_red_weights / k_red_combine).  The nbk = 2^m buckets of a set form a
// matrix of 2^H rows and 2^L columns (bucket b = h 2^L + l);  sum_b (b+1) B_b = sum_l (l+1) C_l + 2^L sum_h h R_h  with
// the column sums C_l and row sums R_h: two additions per bucket and then only 2^H + 2^L small multiplications,
// where one segment of buckets per quad (k_reduce1) multiplies every quad's running sum by its offset.
struct MatGeom {
  uint32_t Lb, nrows, ncols, nsums;      // columns = 2^Lb
  uint32_t Q, sums_per_wg, wgs_per_set;  // phase 1: Q quads (a power of two <= 64) per sum
  uint32_t bA, bB;                       // phase 2 workgroups per set: column part, row part
  size_t sums_bytes, scratch_bytes;      // per call: sums of all sets, then the phase-2 partials
};
static MatGeom mat_geom(size_t gsets, uint32_t nbk) {
  MatGeom g;
  uint32_t m = 0;
  while ((1u << m) < nbk) ++m;
  g.Lb = (m + 1) / 2;
  g.ncols = 1u << g.Lb;
  g.nrows = nbk >> g.Lb;
  g.nsums = g.nrows + g.ncols;
  uint32_t q = g.nrows / 2;                                  // at least two points per quad in a column sum
  if (q < 1) q = 1;
  if (q > 64) q = 64;
  g.Q = q;
  g.sums_per_wg = 64 / g.Q;
  g.wgs_per_set = (g.nsums + g.sums_per_wg - 1) / g.sums_per_wg;
  g.bA = (g.ncols + 63) / 64;
  g.bB = (g.nrows + 63) / 64;
  g.sums_bytes = align_up(gsets * g.nsums * 128, 256);
  g.scratch_bytes = g.sums_bytes + align_up(gsets * (g.bA + g.bB) * 128, 256);
  return g;
}
static bool use_matrix_reduction(uint32_t nbk) {
  return tuning().reduction != 0 && nbk >= 16;         // (tuning: 0 = segments)
}

static WsLayout ws_layout(const MsmPlan& p) {
  WsLayout w{};
  size_t off = 0;
  const size_t nkeys = (size_t)p.gsets * p.nbk;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  w.countsA = take((size_t)p.nblkA * p.bins * 4);
  w.pcount = take((size_t)p.bins * 4);
  w.pstart = take(((size_t)p.bins + 1) * 4);
  w.recs = take((size_t)p.windows * p.n * 8);
  w.bcount = take(nkeys * 4);
  w.bstart = take((nkeys + 1) * 4);
  w.tstart = take(((size_t)p.nthreads + 1) * 4);
  w.sorted = take(((size_t)p.windows * p.n + 64) * 4);
  w.bucket_acc = take(nkeys * 128);
  w.heads = take((size_t)p.nthreads * 128);
  w.heavy = take((nkeys + 4 + 2 * MAX_GIANTS) * 4);        // count, queued buckets, giant count, giant arrival counters, giant queue
  w.giant = take((size_t)MAX_GIANTS * GIANT_PARTS * 128);     // partial sums of giant buckets
  const RedGeom rg = red_geom(nkeys, p.nbk);
  w.red_seg = rg.seg; w.red_threads_per_set = rg.threads_per_set; w.red_block = rg.block;
  w.red_blocks_per_set = rg.blocks_per_set;
  {
    const size_t seg_bytes = (size_t)p.gsets * w.red_blocks_per_set * 128;
    const size_t mat_bytes = mat_geom(p.gsets, p.nbk).scratch_bytes;
    w.partials = take(seg_bytes > mat_bytes ? seg_bytes : mat_bytes);
  }
  w.wsum = take((size_t)p.gsets * 128);
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

// The two-level sort needs  bucket sets x 2^(bucket bits - 10)  partitions (pass B sorts at most 10 bits) and has at
// most 8192 (pass A keeps a cursor per partition in LDS): large windows with many bucket sets do not fit.
bool msm_plan_feasible(int groups, int c, int sets) {
  const int windows = (256 + c - 1) / c;
  if (sets <= 0) sets = windows;
  const int over = c - 1 - 10;
  return ((uint64_t)groups * sets << (over > 0 ? over : 0)) <= 8192u;
}

MsmPlan msm_make_plan(int groups, const size_t* gn, const size_t* goff, int c, int sets, int tables, int num_cus, int acc_fill, int scalar_bits) {
  MsmPlan p;
  p.groups = groups;
  size_t n = 0, nmax = 0;
  for (int g = 0; g < groups; ++g) {
    p.gn[g] = (uint32_t)gn[g];
    p.goff[g] = (uint32_t)goff[g];
    n += gn[g];
    if (gn[g] > nmax) nmax = gn[g];
  }
  p.n = (uint32_t)n;
  p.c = c;
  p.windows = (scalar_bits + c - 1) / c;       // 256: field elements; 132: the half-scalars of the endomorphism (129 bits + the digits' carry)
  p.signed_scalars = scalar_bits != 256;
  if (sets <= 0 || tables <= 0) { sets = p.windows; tables = 1; }
  p.sets = sets;
  p.gsets = groups * sets;
  p.tables = tables;
  p.nbk = 1u << (c - 1);
  // two-level sort geometry: pass A splits on the high pb bucket bits, pass B on the low fb.  One pass-B
  // workgroup sorts one partition, so: about one partition per CU (measured optimum at 2^18 and 2^20; more
  // partitions shorten pass A's contiguous runs), at most 8192 (pass A keeps a cursor per partition in LDS),
  // fine bits <= 10 (one thread per fine bucket in k_fine)
  p.pb = 0;
  while (p.pb < c - 1 && ((uint32_t)p.gsets << p.pb) < (uint32_t)num_cus) ++p.pb;
  if (tuning().part_bits >= 0 && tuning().part_bits <= c - 1) p.pb = tuning().part_bits;      // tuning override
  // windows of 18 bits and more WITHOUT the staged sort (sort_staged = 0): the partitions take the surplus bits (up to 8192 of
  // them: pass A's cursors are 32 KB of LDS), pass B keeps 2^8 fine buckets, and pass A's chunks grow so that a (workgroup,
  // partition) run stays a few lines long (window 20 at 2^22 points: sort 1.78 -> 1.16 ms; profiles/r05_window_sweep.txt)
  // (with the STAGED sort, sort_staged = 1, both passes write whole pieces whatever the bucket count, and pass A's staging
  // needs few partitions to have several records per partition and tile: 512 partitions, up to 2^10 fine buckets)
  const int fine_max = c >= 18 && tuning().part_bits < 0 && !tuning().sort_staged ? 8 : 10;
  while (c - 1 - p.pb > fine_max) ++p.pb;
  while (p.pb > 0 && ((uint32_t)p.gsets << p.pb) > 8192u) --p.pb;
  p.fb = c - 1 - p.pb;
  p.bins = (uint32_t)p.gsets << p.pb;
  size_t chA = (n + 511) / 512;                       // ~2 pass-A workgroups per CU
  chA = (chA + 255) / 256 * 256;
  if (chA < 256) chA = 256;
  const size_t chA_max = p.bins > 512 ? 16384 : 4096;
  if (chA > chA_max) chA = chA_max;
  p.chA = (uint32_t)chA;
  uint32_t blk = 0;
  for (int g = 0; g < groups; ++g) {
    blk += (uint32_t)((gn[g] + chA - 1) / chA);
    p.gblk_end[g] = blk;
  }
  p.nblkA = blk ? blk : 1;
  // accumulate slices.  k_accumulate is resident at ACC_WG_PER_CU workgroups per CU (register-limited) and
  // ALU-issue-bound: measured on MI355X, a grid that exactly fills the resident slots takes L x 14.1 us, and a
  // single workgroup more costs a whole extra round.  So: one round, every slot used, L = ceil(entries / slots)
  // (any L: entries are read one dword at a time), at least 8 so that slice heads stay few.
  size_t ne = (size_t)n * p.windows;
  int acc_wg = acc_fill ? acc_fill : tuning().accumulate_fill;         // the context's choice, else the process-wide one
  // automatic: three for a single MSM of 2^22 (scalar, window) entries or more, two below and for batches.  Three wavefronts per SIMD hide the gathers and the
  // dependent issue better than two (k_accumulate alone at 2^20 points over a table: 1.043-1.056 against 1.076-1.079 ms; without a
  // table, 2n points [P | phi(P)] and nine bucket sets: 2.47 -> 2.33 ms per call); below that the extra slice heads cost more than
  // the occupancy gives (table-less 2^16: 0.79 -> 0.84 ms).  MSMs in flight measure the same either way; a prover's side queues
  // set three themselves (profiles/r05_fill_tableless.txt).
  if (acc_wg == 0) acc_wg = (groups == 1 && ne >= ((size_t)1 << 22)) ? ACC_WG_PER_CU : ACC_WG_FILL;   // (batches: two, as measured in a prover)
  if (acc_wg < 1 || acc_wg > ACC_WG_PER_CU) acc_wg = ACC_WG_FILL;
  const size_t slots = (size_t)num_cus * acc_wg * 256;
  size_t L = (ne + slots - 1) / slots;           // upper bound: the kernels shorten it to the actual entry count
  if (L < 8) L = 8;
  p.L = (uint32_t)L;
  p.Lfixed = 0;
  p.slots = (uint32_t)slots;
  p.nthreads = (uint32_t)slots;
  if (tuning().slice_len >= 1 && tuning().slice_len <= 65536) {          // tuning override: a fixed slice length
    const long v = tuning().slice_len;
    p.L = p.Lfixed = (uint32_t)v; p.nthreads = (uint32_t)((ne + v - 1) / v); if (p.nthreads < 1) p.nthreads = 1;
  }
  p.ws_bytes = ws_layout(p).total;
  return p;
}

// ------------------------------------------------------------------------------------------
// signed digits
// ------------------------------------------------------------------------------------------
// The scalar's limbs sit in LDS (SoA, one private column per thread: no barrier needed) because the
// windows index them dynamically.  digit(w) = signed c-bit digit of window w with the running carry.
struct DigitIter {
  const uint32_t* col;      // &limbs[threadIdx.x], stride 256
  uint32_t carry;
  __device__ __forceinline__ uint32_t next(int w, int c) {          // returns magnitude | sign<<31
    const int bit = w * c;
    const int l = bit >> 5, sh = bit & 31;
    const uint32_t lo = (l < 8) ? col[l * 256] : 0u;
    const uint32_t hi = (l + 1 < 9) ? col[(l + 1) * 256] : 0u;
    const uint64_t two = ((uint64_t)hi << 32) | lo;
    const uint32_t raw = ((uint32_t)(two >> sh) & ((1u << c) - 1u)) + carry;
    if (raw > (1u << (c - 1))) { carry = 1; return ((1u << c) - raw) | SIGN_BIT; }
    carry = 0;
    return raw;
  }
};
template <class SP>
__device__ __forceinline__ void stage_scalar(uint32_t* limbs, Fe<SP> s, int is_mont) {
  if (is_mont) s = fe_from_mont(s);
#pragma unroll
  for (int l = 0; l < 8; ++l) limbs[l * 256 + threadIdx.x] = s.v[l];
  limbs[8 * 256 + threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------
// pass A: partition the entries by the high bucket bits
// ------------------------------------------------------------------------------------------
struct PartGroups {
  const uint32_t* scalars[MSM_MAX_GROUPS];
  uint32_t n[MSM_MAX_GROUPS], off[MSM_MAX_GROUPS], blk_end[MSM_MAX_GROUPS];
  int groups;
  int signed_scalars;      // the scalars are sign-and-magnitude words (bit 255 = sign; k_glv_split): the sign flips every digit
};

template <class SP, bool SCATTER>
__global__ __launch_bounds__(256) void k_part(PartGroups pg, int is_mont, int c, int windows, int sets, int pb, int fb,
                                              uint32_t bins, uint32_t chA, uint32_t tstride,
                                              uint32_t* __restrict__ countsA, const uint32_t* __restrict__ pstart,
                                              uint32_t* __restrict__ recsP, uint16_t* __restrict__ recsK, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ uint32_t limbs[9 * 256];
  extern __shared__ uint32_t cur[];                   // bins counters (histogram) or cursors (scatter)
  uint32_t* mine = countsA + (size_t)blockIdx.x * bins;
  for (uint32_t b = threadIdx.x; b < bins; b += 256) cur[b] = SCATTER ? pstart[b] + mine[b] : 0u;
  __syncthreads();
  int g = 0;
  while (g < pg.groups - 1 && blockIdx.x >= pg.blk_end[g]) ++g;
  const uint32_t blk0 = g ? pg.blk_end[g - 1] : 0u;
  const uint32_t* __restrict__ scalars = pg.scalars[g];
  const uint32_t n = pg.n[g];
  const uint32_t lo = (blockIdx.x - blk0) * chA;
  const uint32_t hi = (lo + chA < n) ? lo + chA : n;
  const uint32_t fmask = (1u << fb) - 1u;
  const uint32_t set0 = (uint32_t)g * (uint32_t)sets;        // a group is `sets` bucket sets of its own
  const uint32_t pbase = pg.off[g];
  uint32_t i = lo + threadIdx.x;
  Fe<SP> nxt = fe_zero<SP>();
  if (i < hi) nxt = fe_load<SP>(scalars + (size_t)i * 8);
  for (; i < hi; i += 256) {
    const Fe<SP> curs = nxt;
    if (i + 256 < hi) nxt = fe_load<SP>(scalars + (size_t)(i + 256) * 8);   // next scalar in flight
    Fe<SP> mags = curs;
    uint32_t ssign = 0;
    if (pg.signed_scalars) { ssign = mags.v[7] & SIGN_BIT; mags.v[7] &= ~SIGN_BIT; }
    stage_scalar<SP>(limbs, mags, is_mont);
    DigitIter it{limbs + threadIdx.x, 0u};
    for (int w = 0; w < windows; ++w) {
      const uint32_t d = it.next(w, c) ^ ssign;
      const uint32_t mag = d & ~SIGN_BIT;
      if (!mag) continue;
      const uint32_t s = (uint32_t)w % (uint32_t)sets, j = (uint32_t)w / (uint32_t)sets;
      const uint32_t bin = ((set0 + s) << pb) | ((mag - 1) >> fb);
      const uint32_t pos = atomicAdd(&cur[bin], 1u);
      if (SCATTER) { recsP[pos] = (j * tstride + pbase + i) | (d & SIGN_BIT); recsK[pos] = (uint16_t)((mag - 1) & fmask); }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += 256) mine[b] = cur[b];
  }
}

// Pass A's scatter STAGED through LDS (vdf_hip_tuning.sort_staged; up to 512 partitions, up to 16 windows).  k_part<true>
// writes every 8-byte record on its own, and a (workgroup, partition) run -- contiguous in memory -- fills over the whole
// life of the workgroup: with 2048 partitions and 64 workgroups per XCD that is 16 MB of half-written lines against 4 MB of
// L2, and the lines go to HBM several times (2^24 points, window 20: 2.3 ms for 1.75 GB).  Here a workgroup takes its chunk
// in tiles of 256 scalars: the tile's records (up to 4096) are ranked by partition in LDS, laid out partition by partition,
// and copied out by consecutive lanes, so a partition's share of a tile (8 records on average at 512 partitions) leaves as
// one 64-byte piece.  Same cursors, same result layout as k_part<true>; the order inside a run differs, which nothing reads.
static constexpr uint32_t PART_STAGE_BINS = 512;
template <class SP>
__global__ __launch_bounds__(256) void k_part_staged(PartGroups pg, int is_mont, int c, int windows, int sets, int pb, int fb,
                                                     uint32_t bins, uint32_t chA, uint32_t tstride,
                                                     const uint32_t* __restrict__ countsA, const uint32_t* __restrict__ pstart,
                                                     uint32_t* __restrict__ recsP, uint16_t* __restrict__ recsK, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ uint32_t limbs[9 * 256];
  __shared__ uint32_t cur[PART_STAGE_BINS];                        // global cursors: where each partition's next record goes
  __shared__ uint32_t tcnt[PART_STAGE_BINS];                       // the tile's records per partition
  __shared__ uint32_t toff[PART_STAGE_BINS + 1];                   // ... their exclusive scan (the tile's layout in `stage`)
  __shared__ uint32_t tpos[PART_STAGE_BINS];                       // ... and the cursors of the placement pass
  __shared__ uint32_t stage[256 * 16];                             // payloads; 8 bytes of LDS per record: three workgroups per CU
  __shared__ uint32_t skb[256 * 16];                               // fine key | partition << 16
  const uint32_t* mine = countsA + (size_t)blockIdx.x * bins;
  for (uint32_t b = threadIdx.x; b < bins; b += 256) { cur[b] = pstart[b] + mine[b]; tcnt[b] = 0; }
  int g = 0;
  while (g < pg.groups - 1 && blockIdx.x >= pg.blk_end[g]) ++g;
  const uint32_t blk0 = g ? pg.blk_end[g - 1] : 0u;
  const uint32_t* __restrict__ scalars = pg.scalars[g];
  const uint32_t n = pg.n[g];
  const uint32_t lo = (blockIdx.x - blk0) * chA;
  const uint32_t hi = (lo + chA < n) ? lo + chA : n;
  const uint32_t fmask = (1u << fb) - 1u;
  const uint32_t set0 = (uint32_t)g * (uint32_t)sets;
  const uint32_t pbase = pg.off[g];
  const uint32_t per_lane = (bins + 63) / 64;
  Fe<SP> nxt = fe_zero<SP>();
  if (lo + threadIdx.x < hi) nxt = fe_load<SP>(scalars + (size_t)(lo + threadIdx.x) * 8);
  __syncthreads();
  for (uint32_t base = lo; base < hi; base += 256) {
    const uint32_t i = base + threadIdx.x;
    Fe<SP> mags = nxt;
    if (i + 256 < hi) nxt = fe_load<SP>(scalars + (size_t)(i + 256) * 8);    // the next tile's scalar in flight across this tile's barriers
    uint32_t ssign = 0;
    // pass 1: count the tile's records per partition (the digits are extracted twice -- two LDS reads and a few instructions
    // each -- rather than kept: sixteen records per lane in registers would be indexed dynamically, i.e. live in scratch)
    if (i < hi) {
      if (pg.signed_scalars) { ssign = mags.v[7] & SIGN_BIT; mags.v[7] &= ~SIGN_BIT; }
      stage_scalar<SP>(limbs, mags, is_mont);
      DigitIter it{limbs + threadIdx.x, 0u};
      for (int w = 0; w < windows; ++w) {
        const uint32_t mag = it.next(w, c) & ~SIGN_BIT;
        if (!mag) continue;
        atomicAdd(&tcnt[((set0 + (uint32_t)w % (uint32_t)sets) << pb) | ((mag - 1) >> fb)], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                        // exclusive scan of the tile's partition counts: wavefront 0
      uint32_t run = 0;
      const uint32_t b0 = threadIdx.x * per_lane;
      for (uint32_t k = 0; k < per_lane; ++k) if (b0 + k < bins) run += tcnt[b0 + k];
      uint32_t incl = run;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)threadIdx.x >= d) incl += o; }
      uint32_t off = incl - run;
      for (uint32_t k = 0; k < per_lane; ++k) if (b0 + k < bins) { toff[b0 + k] = off; tpos[b0 + k] = off; off += tcnt[b0 + k]; }
      if (threadIdx.x == 63) toff[bins] = incl;                    // the tile's record count
    }
    __syncthreads();
    if (i < hi) {                                                  // pass 2: place the records, partition by partition
      DigitIter it{limbs + threadIdx.x, 0u};
      for (int w = 0; w < windows; ++w) {
        const uint32_t d = it.next(w, c) ^ ssign;
        const uint32_t mag = d & ~SIGN_BIT;
        if (!mag) continue;
        const uint32_t s = (uint32_t)w % (uint32_t)sets, j = (uint32_t)w / (uint32_t)sets;
        const uint32_t bin = ((set0 + s) << pb) | ((mag - 1) >> fb);
        const uint32_t p = atomicAdd(&tpos[bin], 1u);
        stage[p] = (j * tstride + pbase + i) | (d & SIGN_BIT);
        skb[p] = ((mag - 1) & fmask) | (bin << 16);
      }
    }
    __syncthreads();
    const uint32_t tile_n = toff[bins];
    for (uint32_t q = threadIdx.x; q < tile_n; q += 256) {
      const uint32_t kb = skb[q], b = kb >> 16;
      const uint32_t pos = cur[b] + (q - toff[b]);
      recsP[pos] = stage[q];
      recsK[pos] = (uint16_t)kb;
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += 256) { cur[b] += tcnt[b]; tcnt[b] = 0; }
    __syncthreads();
  }
}

// per bin: exclusive scan over the pass-A workgroups.  One 256-thread workgroup per bin scans 256 counts at
// a time through LDS (a thread per bin walking hundreds of counts would be one L2 round trip each).
__global__ __launch_bounds__(256) void k_part_scan(uint32_t* __restrict__ countsA, uint32_t bins, uint32_t nblk,
                                                   uint32_t* __restrict__ pcount, uint32_t* __restrict__ heavy,
                                                   uint32_t* __restrict__ giant_done, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ uint32_t sc[256];
  const uint32_t b = blockIdx.x;
  uint32_t carry = 0;
  for (uint32_t k0 = 0; k0 < nblk; k0 += 256) {
    const uint32_t k = k0 + threadIdx.x;
    const uint32_t v = (k < nblk) ? countsA[(size_t)k * bins + b] : 0u;
    sc[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
      const uint32_t t = (threadIdx.x >= d) ? sc[threadIdx.x - d] : 0u;
      __syncthreads();
      sc[threadIdx.x] += t;
      __syncthreads();
    }
    if (k < nblk) countsA[(size_t)k * bins + b] = carry + sc[threadIdx.x] - v;
    const uint32_t tot = sc[255];
    __syncthreads();
    carry += tot;
  }
  if (threadIdx.x == 0) pcount[b] = carry;
  if (b == 0 && threadIdx.x == 0) { heavy[0] = 0; giant_done[-1] = 0; }   // both queues of this run start empty (heavy_layout)
  if (b == 0 && threadIdx.x < MAX_GIANTS) giant_done[threadIdx.x] = 0;
}

// single workgroup exclusive scan: out[0..n], out[n] = total.  Each thread owns a contiguous run; loads are
// issued eight at a time so the run is a few L2 round trips instead of one per element.
__global__ __launch_bounds__(1024) void k_scan_keys(const uint32_t* __restrict__ bcount, uint32_t nkeys,
                                                    uint32_t* __restrict__ bstart, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ uint32_t part[1024];
  const uint32_t per = (nkeys + 1023) / 1024;
  const uint32_t lo = threadIdx.x * per < nkeys ? threadIdx.x * per : nkeys;
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

// ------------------------------------------------------------------------------------------
// pass B: sort every partition by the low bucket bits.
// ------------------------------------------------------------------------------------------
// One 1024-thread workgroup per partition does all of pass B for it: histogram of the fine bucket bits in LDS,
// exclusive scan (which yields the partition's bucket starts: a partition's buckets are contiguous inside the
// partition's own region, so there is no scan over all keys), then the scatter with the scanned counters as
// cursors.  The second read of the records comes from L2 / MALL.  The scan also zeroes the accumulator of every
// empty bucket (k_accumulate writes each non-empty one exactly once), so the pipeline needs no memset.
// atomicAdd(&h[key], 1) for every active lane; when the whole wavefront holds ONE key (a hot bucket: scalars with a
// repeated digit, 0/1 witnesses) one lane adds the lane count instead of 64 serialised atomics on one address
__device__ __forceinline__ uint32_t lds_take_slot(uint32_t* h, uint32_t key) {
  const uint32_t first = __builtin_amdgcn_readfirstlane(key);
  const uint64_t act = __ballot(1);
  if (__ballot(key == first) == act) {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    uint32_t base = 0;
    if (rank == 0) base = atomicAdd(&h[first], (uint32_t)__popcll(act));
    return (uint32_t)__builtin_amdgcn_readfirstlane(base) + rank;
  }
  return atomicAdd(&h[key], 1u);
}

__global__ __launch_bounds__(1024) void k_fine(const uint32_t* __restrict__ recsP, const uint16_t* __restrict__ recsK, const uint32_t* __restrict__ pstart,
                                               uint32_t bins, uint32_t nf, uint32_t* __restrict__ bstart,
                                               uint32_t* __restrict__ sorted, char* __restrict__ bucket_acc,
                                               uint32_t slots, uint32_t Lfixed, uint32_t* __restrict__ tstart, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ uint32_t h[1024];
  __shared__ uint32_t sc[1024];
  const uint32_t bin = blockIdx.x, f = threadIdx.x;
  const uint32_t lo = pstart[bin], hi = pstart[bin + 1];
  const uint32_t L = slice_len(pstart[bins], slots, Lfixed);
  h[f] = 0;
  __syncthreads();
  uint32_t i = lo + f;
  for (; i + 3 * 1024 < hi; i += 4 * 1024) {              // four independent loads in flight per lane
    uint32_t k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) k[u] = recsK[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 4; ++u) lds_take_slot(h, k[u]);
  }
  for (; i < hi; i += 1024) lds_take_slot(h, (uint32_t)recsK[i]);
  __syncthreads();
  const uint32_t cnt = (f < nf) ? h[f] : 0u;
  sc[f] = cnt;
  __syncthreads();
  for (uint32_t d = 1; d < nf; d <<= 1) {
    const uint32_t t = (f >= d) ? sc[f - d] : 0u;
    __syncthreads();
    sc[f] += t;
    __syncthreads();
  }
  if (f < nf) {
    const uint32_t start = lo + sc[f] - cnt;
    bstart[(size_t)bin * nf + f] = start;
    h[f] = start;                                          // cursor
    if (cnt == 0) {
      uint4* z = reinterpret_cast<uint4*>(bucket_acc + ((size_t)bin * nf + f) * 128);
#pragma unroll
      for (int q = 0; q < 8; ++q) z[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (bin == bins - 1 && f == nf - 1) bstart[(size_t)bins * nf] = pstart[bins];
    // k_accumulate thread t starts at sorted position t * L: tell it which bucket that is (replaces a 16-step
    // binary search over bstart, a chain of dependent loads at the start of every thread).  Buckets of up to 32
    // slices are written by their own thread; hot buckets (thousands of slices) by the whole workgroup below.
    if ((uint64_t)cnt <= 32ull * L)
      for (uint32_t t = (start + L - 1) / L; (uint64_t)t * L < (uint64_t)start + cnt; ++t) tstart[t] = bin * nf + f;
  }
  __syncthreads();
  for (uint32_t ff = 0; ff < nf; ++ff) {
    const uint32_t c = sc[ff] - (ff ? sc[ff - 1] : 0u);
    if ((uint64_t)c <= 32ull * L) continue;
    const uint32_t st = h[ff];                              // cursors still hold the bucket starts here
    for (uint32_t t = (st + L - 1) / L + f; (uint64_t)t * L < (uint64_t)st + c; t += 1024) tstart[t] = bin * nf + ff;
  }
  __syncthreads();
  i = lo + f;
  for (; i + 3 * 1024 < hi; i += 4 * 1024) {
    uint32_t r[4], kk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { r[u] = recsP[i + u * 1024]; kk[u] = recsK[i + u * 1024]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t pos = lds_take_slot(h, kk[u]);
      sorted[pos] = r[u];
    }
  }
  for (; i < hi; i += 1024) {
    const uint32_t pos = lds_take_slot(h, (uint32_t)recsK[i]);
    sorted[pos] = recsP[i];
  }
}

// Pass B with its scatter STAGED through LDS (vdf_hip_tuning.sort_staged).  k_fine's second phase writes every entry on its
// own -- 64 lanes, 64 four-byte writes to ~60 different lines per instruction, one L2 transaction each: at 2^24 points the
// 2 x 10^8 entries are 0.8 ms of L2 transactions alone.  Here a workgroup takes its partition in tiles of 4096 records: a
// tile's entries are ranked by fine bucket in LDS (a tile histogram, one wavefront's scan), laid out bucket by bucket in an
// LDS buffer, and copied out by consecutive lanes -- a run of a tile's entries for one bucket (16 on average at 2^8 fine
// buckets) leaves as one or two whole lines.  Phase 1 (histogram, bucket starts, empty buckets, slice starts) is k_fine's.
template <int PER>                                      // records per thread and tile: 4 (4096 per tile) up to 2^8 fine buckets, 8 beyond
__global__ __launch_bounds__(1024) void k_fine_staged(const uint32_t* __restrict__ recsP, const uint16_t* __restrict__ recsK, const uint32_t* __restrict__ pstart,
                                                      uint32_t bins, uint32_t nf, uint32_t* __restrict__ bstart,
                                                      uint32_t* __restrict__ sorted, char* __restrict__ bucket_acc,
                                                      uint32_t slots, uint32_t Lfixed, uint32_t* __restrict__ tstart, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ uint32_t h[1024];
  __shared__ uint32_t sc[1024];
  __shared__ uint32_t tcnt[1024];
  constexpr uint32_t FINE_TILE = 1024u * PER;
  __shared__ uint32_t stage[FINE_TILE];
  __shared__ uint16_t sbin[FINE_TILE];
  const uint32_t bin = blockIdx.x, f = threadIdx.x;
  const uint32_t lo = pstart[bin], hi = pstart[bin + 1];
  const uint32_t L = slice_len(pstart[bins], slots, Lfixed);
  h[f] = 0;
  __syncthreads();
  uint32_t i = lo + f;
  for (; i + 3 * 1024 < hi; i += 4 * 1024) {
    uint32_t k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) k[u] = recsK[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 4; ++u) lds_take_slot(h, k[u]);
  }
  for (; i < hi; i += 1024) lds_take_slot(h, (uint32_t)recsK[i]);
  __syncthreads();
  const uint32_t cnt = (f < nf) ? h[f] : 0u;
  sc[f] = cnt;
  __syncthreads();
  for (uint32_t d = 1; d < nf; d <<= 1) {
    const uint32_t t = (f >= d) ? sc[f - d] : 0u;
    __syncthreads();
    sc[f] += t;
    __syncthreads();
  }
  if (f < nf) {
    const uint32_t start = lo + sc[f] - cnt;
    bstart[(size_t)bin * nf + f] = start;
    h[f] = start;                                          // cursor: the bucket's next free position
    if (cnt == 0) {
      uint4* z = reinterpret_cast<uint4*>(bucket_acc + ((size_t)bin * nf + f) * 128);
#pragma unroll
      for (int q = 0; q < 8; ++q) z[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (bin == bins - 1 && f == nf - 1) bstart[(size_t)bins * nf] = pstart[bins];
    if ((uint64_t)cnt <= 32ull * L)
      for (uint32_t t = (start + L - 1) / L; (uint64_t)t * L < (uint64_t)start + cnt; ++t) tstart[t] = bin * nf + f;
  }
  __syncthreads();
  for (uint32_t ff = 0; ff < nf; ++ff) {
    const uint32_t c = sc[ff] - (ff ? sc[ff - 1] : 0u);
    if ((uint64_t)c <= 32ull * L) continue;
    const uint32_t st = h[ff];
    for (uint32_t t = (st + L - 1) / L + f; (uint64_t)t * L < (uint64_t)st + c; t += 1024) tstart[t] = bin * nf + ff;
  }
  __syncthreads();
  // ---- phase 2: tile by tile through LDS
  const uint32_t per_lane = (nf + 63) / 64;                 // fine buckets each lane of wavefront 0 scans (nf <= 1024: at most 16)
  for (uint32_t base = lo; base < hi; base += FINE_TILE) {
    const uint32_t tile_n = hi - base < FINE_TILE ? hi - base : FINE_TILE;
    tcnt[f] = 0;
    __syncthreads();
    uint32_t key[PER], pay[PER], rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const uint32_t s = (uint32_t)u * 1024 + f;
      key[u] = 0xFFFFFFFFu;
      if (s < tile_n) {
        key[u] = recsK[base + s]; pay[u] = recsP[base + s];
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) if (key[u] != 0xFFFFFFFFu) rank[u] = lds_take_slot(tcnt, key[u]);
    __syncthreads();
    // exclusive scan of the tile histogram by wavefront 0: a lane's run of per_lane buckets, then a shuffle scan over the lanes
    if (f < 64) {
      uint32_t run = 0;
      const uint32_t b0 = f * per_lane;
      for (uint32_t k = 0; k < per_lane; ++k) if (b0 + k < nf) run += tcnt[b0 + k];
      uint32_t incl = run;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)f >= d) incl += o; }
      uint32_t off = incl - run;
      for (uint32_t k = 0; k < per_lane; ++k) if (b0 + k < nf) { sc[b0 + k] = off; off += tcnt[b0 + k]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) if (key[u] != 0xFFFFFFFFu) {
      const uint32_t p = sc[key[u]] + rank[u];
      stage[p] = pay[u];
      sbin[p] = (uint16_t)key[u];
    }
    __syncthreads();
    for (uint32_t s = f; s < tile_n; s += 1024) {
      const uint32_t b = sbin[s];
      sorted[h[b] + (s - sc[b])] = stage[s];
    }
    __syncthreads();
    if (f < nf) h[f] += tcnt[f];
    __syncthreads();
  }
}

// The mixed addition of the bucket loop is ec.cuh's xyzz_madd_lazy (sign-tracked, product pair for Y3).
template <class P>
__device__ __forceinline__ void flush_lazy(const XYZZ<P>& acc, bool have, bool flip, char* dst) {
  // stored as is, in the lazy domain: the tail kernels canonicalise on load (qpoint_load_lazy), each lane its
  // own coordinate, which takes ~140 instructions out of a path most iterations execute for a few lanes
  xyzz_store<P>(dst, xyzz_lazy_resolve<P>(acc, have, flip));
}

template <class P>
__global__ __launch_bounds__(256) void k_accumulate(const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ bstart, uint32_t nkeys,
                                                    const uint32_t* __restrict__ tstart,
                                                    const char* __restrict__ points, char* __restrict__ bucket_acc,
                                                    char* __restrict__ heads, uint32_t slots, uint32_t Lfixed,
                                                    uint32_t nthreads) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nthreads) return;
  const uint32_t ne = bstart[nkeys];
  const uint32_t L = slice_len(ne, slots, Lfixed);
  const uint64_t lo64 = (uint64_t)t * L;
  if (lo64 >= ne) return;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = (lo64 + L < ne) ? lo + L : ne;
  uint32_t g = tstart[t];                    // bucket containing position lo (written by k_fine)
  uint32_t next = bstart[g + 1];
  bool is_head = bstart[g] < lo;
  XYZZ<P> acc = xyzz_identity<P>();
  bool have = false, flip = false;            // flip: the accumulator holds the NEGATED partial sum (ec.cuh xyzz_madd_lazy)
  // software pipeline: while entry `pos` is added, the point of entry pos+1 is in flight.  Entry indices arrive
  // four at a time as one aligned dwordx4 (every sorted line is then fetched from HBM once; one dword per
  // iteration re-fetched lines the L1 had already dropped: +20 % FETCH_SIZE); the chunk for positions 4k..4k+3
  // is requested while position 4k-2 is processed.  The slice may start and end inside a chunk: the list is
  // padded, and positions outside [lo, hi) are never used.
  const uint4* chunks = reinterpret_cast<const uint4*>(sorted);
  uint4 cq = chunks[lo >> 2];                                   // chunk holding `pos`
  uint4 nq = chunks[(lo >> 2) + 1];                             // the following chunk
  auto pick = [](const uint4& q, uint32_t k) { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; };
  uint32_t e = pick(cq, lo & 3u);
  Affine<P> pt = affine_load<P>(points + (size_t)(e & ~SIGN_BIT) * 64);
  for (uint32_t pos = lo; pos < hi; ++pos) {
    const uint32_t pn = pos + 1;
    if ((pn & 3u) == 0) { cq = nq; nq = chunks[(pn >> 2) + 1]; }   // crossed into the next chunk: rotate, prefetch
    const uint32_t en = (pn < hi) ? pick(cq, pn & 3u) : e;
    Affine<P> ptn = affine_load<P>(points + (size_t)(en & ~SIGN_BIT) * 64);
    if (pos >= next) {
      flush_lazy<P>(acc, have, flip, is_head ? heads + (size_t)t * 128 : bucket_acc + (size_t)g * 128);
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
    if (!affine_is_identity(pt)) {
      if (((e & SIGN_BIT) != 0) != (have && flip)) pt.y = fe_neg_nz(pt.y);   // digit sign XOR the accumulator's pending sign (y != 0 on these curves)
      xyzz_madd_lazy<P>(acc, have, flip, pt);
    }
    e = en;
    pt = ptn;
  }
  flush_lazy<P>(acc, have, flip, is_head ? heads + (size_t)t * 128 : bucket_acc + (size_t)g * 128);
}

// ------------------------------------------------------------------------------------------
// Tail of the pipeline.  Every kernel below is a short chain of dependent point additions run by few
// waves, so they use the quad-cooperative group law of ecq.cuh: four lanes per point, four
// multiplication stages per addition instead of fourteen serial multiplications.
// "Logical thread" = quad = (global lane index) / 4.
// ------------------------------------------------------------------------------------------
// The queues k_fixup fills for k_fixup_heavy, in one allocation of nkeys + 4 + 2 * MAX_GIANTS words.
struct HeavyLayout { uint32_t* count; uint32_t* items; uint32_t* giant_count; uint32_t* done; uint32_t* giants; };
__host__ __device__ __forceinline__ HeavyLayout heavy_layout(uint32_t* heavy, uint32_t nkeys) {
  HeavyLayout q;
  q.count = heavy;                         // plain heavy buckets queued
  q.items = heavy + 1;                     // ... and which (at most nkeys)
  q.giant_count = heavy + 1 + nkeys + 1;   // giant buckets seen (may exceed MAX_GIANTS: the surplus is queued as plain)
  q.done = heavy + 1 + nkeys + 2;          // arrival counter per giant
  q.giants = q.done + MAX_GIANTS;          // the giants, numbered by their position
  return q;
}

struct FixupTune { uint32_t heavy_min, giant_span; };
static FixupTune fixup_tune() {                                     // tuning overrides
  FixupTune t{HEAVY_MIN, GIANT_SPAN};
  if (tuning().heavy_min >= 1 && tuning().heavy_min <= 4096) t.heavy_min = (uint32_t)tuning().heavy_min;
  if (tuning().giant_span >= 16 && tuning().giant_span <= (1 << 20)) t.giant_span = (uint32_t)tuning().giant_span;
  return t;
}

// One quad per bucket: add the heads of the slices the bucket spans.
template <class P>
__global__ __launch_bounds__(256) void k_fixup(const uint32_t* __restrict__ bstart, uint32_t nkeys, uint32_t slots,
                                               uint32_t Lfixed, char* __restrict__ bucket_acc,
                                               const char* __restrict__ heads, uint32_t* __restrict__ heavy,
                                               uint32_t heavy_min, uint32_t giant_span, int wave_prio) {
  raise_wave_priority(wave_prio);
  const uint32_t g = (blockIdx.x * 256 + threadIdx.x) >> 2;
  if (g >= nkeys) return;                                          // quad-uniform from here on
  const uint32_t ne = bstart[nkeys];
  const uint32_t L = slice_len(ne, slots, Lfixed);
  const uint32_t s = bstart[g], e = bstart[g + 1];
  if (e <= s) return;
  const uint32_t tf = s / L, tl = (e - 1) / L;
  if (tl == tf) return;
  uint32_t heavy_span = 2u * ((ne / nkeys + L - 1) / L) + 4u;
  if (heavy_span < heavy_min) heavy_span = heavy_min;
  if (tl - tf > heavy_span) {
    if ((threadIdx.x & 3u) == 0) {
      const HeavyLayout q = heavy_layout(heavy, nkeys);
      bool queued = false;
      if (tl - tf > giant_span) {                                  // its own queue: the position is the giant's number
        const uint32_t idx = atomicAdd(q.giant_count, 1u);
        if (idx < MAX_GIANTS) { q.giants[idx] = g; queued = true; }
      }
      if (!queued) q.items[atomicAdd(q.count, 1u)] = g;
    }
    return;
  }
  QPoint<P> acc = qpoint_load_lazy<P>(bucket_acc + (size_t)g * 128);
  QPoint<P> nxt = qpoint_load_lazy<P>(heads + (size_t)(tf + 1) * 128);
  for (uint32_t t = tf + 1; t <= tl; ++t) {
    const QPoint<P> cur = nxt;
    if (t < tl) nxt = qpoint_load_lazy<P>(heads + (size_t)(t + 1) * 128);   // next head in flight during the addition
    acc = qpoint_add<P>(acc, cur);
  }
  qpoint_store<P>(bucket_acc + (size_t)g * 128, acc);
}

// The same with ONE LANE per bucket (vdf_hip_tuning.fixup_serial): the heads are added by ec.cuh's xyzz_add_lazy, ~3,300
// instructions per addition where a quad spends 4 x ~2,070 -- on a device whose limit is instruction issue, a bucket's few heads
// cost 2.4 times less this way; what it gives up is latency (one lane's addition is 14 dependent products, a quad's is 4
// stages), so buckets spanning more than `serial_span` slices still go to the wavefront queue.
template <class P>
__global__ __launch_bounds__(256) void k_fixup_serial(const uint32_t* __restrict__ bstart, uint32_t nkeys, uint32_t slots,
                                                      uint32_t Lfixed, char* __restrict__ bucket_acc,
                                                      const char* __restrict__ heads, uint32_t* __restrict__ heavy,
                                                      uint32_t heavy_min, uint32_t giant_span, int wave_prio) {
  raise_wave_priority(wave_prio);
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nkeys) return;
  const uint32_t ne = bstart[nkeys];
  const uint32_t L = slice_len(ne, slots, Lfixed);
  const uint32_t s = bstart[g], e = bstart[g + 1];
  if (e <= s) return;
  const uint32_t tf = s / L, tl = (e - 1) / L;
  if (tl == tf) return;
  uint32_t heavy_span = 2u * ((ne / nkeys + L - 1) / L) + 4u;
  if (heavy_span < heavy_min) heavy_span = heavy_min;
  if (tl - tf > heavy_span) {
    const HeavyLayout q = heavy_layout(heavy, nkeys);
    bool queued = false;
    if (tl - tf > giant_span) {
      const uint32_t idx = atomicAdd(q.giant_count, 1u);
      if (idx < MAX_GIANTS) { q.giants[idx] = g; queued = true; }
    }
    if (!queued) q.items[atomicAdd(q.count, 1u)] = g;
    return;
  }
  XYZZ<P> acc = xyzz_load<P>(bucket_acc + (size_t)g * 128);          // lazy coordinates, identity = all-zero zz
  bool have = !fe_is_zero(acc.zz), flip = false;
  if (have) { acc.x = fe_canon(acc.x); acc.y = fe_canon(acc.y); acc.zz = fe_canon(acc.zz); acc.zzz = fe_canon(acc.zzz); }
  XYZZ<P> nxt = xyzz_load<P>(heads + (size_t)(tf + 1) * 128);
  for (uint32_t t = tf + 1; t <= tl; ++t) {
    XYZZ<P> cur = nxt;
    if (t < tl) nxt = xyzz_load<P>(heads + (size_t)(t + 1) * 128);    // next head in flight during the addition
    if (fe_is_zero(cur.zz)) continue;                                  // a head whose points cancelled
    cur.x = fe_canon(cur.x); cur.y = fe_canon(cur.y); cur.zz = fe_canon(cur.zz); cur.zzz = fe_canon(cur.zzz);
    if (have && flip) cur.y = fe_neg_nz(cur.y);
    xyzz_add_lazy<P>(acc, have, flip, cur);
  }
  xyzz_store<P>(bucket_acc + (size_t)g * 128, xyzz_lazy_resolve<P>(acc, have, flip));
}

// One wavefront (16 quads) per queued heavy bucket: quads stride over the bucket's heads, then a 4-step butterfly of
// quad additions across the wavefront.  A GIANT bucket (more than GIANT_SPAN heads: a hot digit shared by most
// scalars) is shared by GIANT_PARTS wavefronts (the grid holds 16 such groups, so 16 giant buckets proceed side by side): each sums a contiguous chunk of the heads into
// a scratch slot, and the last one to arrive (a counter per giant bucket) adds the slots to the bucket.  k_fixup
// queues giants apart, so a giant's number is its position there and no wavefront walks a queue: with few buckets
// and many points every bucket is heavy, and a walk would be thousands of dependent loads.
template <class P>
__global__ __launch_bounds__(64) void k_fixup_heavy(const uint32_t* __restrict__ bstart, uint32_t nkeys, uint32_t slots,
                                                    uint32_t Lfixed, char* __restrict__ bucket_acc,
                                                    const char* __restrict__ heads, uint32_t* __restrict__ heavy,
                                                    char* __restrict__ giant, int wave_prio) {
  raise_wave_priority(wave_prio);
  const HeavyLayout q = heavy_layout(heavy, nkeys);
  const uint32_t count = *q.count;
  uint32_t ngiant = *q.giant_count;
  if (ngiant > MAX_GIANTS) ngiant = MAX_GIANTS;
  if (count == 0 && ngiant == 0) return;
  const uint32_t L = slice_len(bstart[nkeys], slots, Lfixed);
  const uint32_t quad = threadIdx.x >> 2;
  // giants: GIANT_PARTS wavefronts each, several giants side by side
  const uint32_t part = blockIdx.x % GIANT_PARTS, ngroups = gridDim.x / GIANT_PARTS;
  if (blockIdx.x < ngroups * GIANT_PARTS)
    for (uint32_t my = blockIdx.x / GIANT_PARTS; my < ngiant; my += ngroups) {
      const uint32_t g = q.giants[my];
      const uint32_t s = bstart[g], e = bstart[g + 1];
      const uint32_t tf = s / L, tl = (e - 1) / L, span = tl - tf;  // heads tf+1 .. tl
      const uint32_t parts = giant_parts(span);
      if (part >= parts) continue;
      const uint32_t chunk = (span + parts - 1) / parts;
      const uint32_t t0 = tf + 1 + part * chunk;
      const uint32_t t1 = (t0 + chunk - 1 < tl) ? t0 + chunk - 1 : tl;
      QPoint<P> acc = qpoint_identity<P>();
      for (uint32_t t = t0 + quad; t <= t1; t += 16) acc = qpoint_add<P>(acc, qpoint_load_lazy<P>(heads + (size_t)t * 128));
      acc = qpoint_wave_sum(acc);
      char* slot = giant + ((size_t)my * GIANT_PARTS + part) * 128;
      if (quad == 0) qpoint_store<P>(slot, acc);
      __threadfence();
      uint32_t arrived = 0;
      if (threadIdx.x == 0) arrived = atomicAdd(&q.done[my], 1u);
      arrived = (uint32_t)__builtin_amdgcn_readfirstlane(arrived);
      if (arrived != parts - 1) continue;
      __threadfence();
      QPoint<P> tot = qpoint_identity<P>();
      for (uint32_t p = quad; p < parts; p += 16) tot = qpoint_add<P>(tot, qpoint_load<P>(giant + ((size_t)my * GIANT_PARTS + p) * 128));
      tot = qpoint_wave_sum(tot);
      if (quad == 0) {
        QPoint<P> base = qpoint_load_lazy<P>(bucket_acc + (size_t)g * 128);
        qpoint_store<P>(bucket_acc + (size_t)g * 128, qpoint_add<P>(base, tot));
      }
    }
  // plain heavy buckets: one wavefront each
  for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    const uint32_t g = q.items[item];
    const uint32_t s = bstart[g], e = bstart[g + 1];
    const uint32_t tf = s / L, tl = (e - 1) / L;
    QPoint<P> acc = qpoint_identity<P>();
    for (uint32_t t = tf + 1 + quad; t <= tl; t += 16) acc = qpoint_add<P>(acc, qpoint_load_lazy<P>(heads + (size_t)t * 128));
    acc = qpoint_wave_sum(acc);
    if (quad == 0) {
      QPoint<P> base = qpoint_load_lazy<P>(bucket_acc + (size_t)g * 128);
      qpoint_store<P>(bucket_acc + (size_t)g * 128, qpoint_add<P>(base, acc));
    }
  }
}

// ------------------------------------------------------------------------------------------
// bucket reduction  sum_{b} (b+1) * B[b]  per set
// ------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void k_reduce1(const char* __restrict__ bucket_acc, uint32_t nbk, uint32_t nseg,
                                                 uint32_t threads_per_set, uint32_t blocks_per_set,
                                                 char* __restrict__ partials, int wave_prio) {
  raise_wave_priority(wave_prio);
  extern __shared__ __align__(16) char lds_raw[];
  const uint32_t nlog = blockDim.x >> 2;                          // logical threads (quads) per block
  const uint32_t lt = threadIdx.x >> 2;
  const uint32_t set = blockIdx.x / blocks_per_set;
  const uint32_t blk = blockIdx.x % blocks_per_set;
  const uint32_t seg = blk * nlog + lt;                            // segment index within the set
  const uint32_t base = seg * nseg;                                // first bucket of the segment
  QPoint<P> run = qpoint_identity<P>(), tot = qpoint_identity<P>();
  if (seg < threads_per_set) {
    const char* bp = bucket_acc + ((size_t)set * nbk + base) * 128;
    for (int l = (int)nseg - 1; l >= 0; --l) {
      run = qpoint_add<P>(run, qpoint_load_lazy<P>(bp + (size_t)l * 128));
      tot = qpoint_add<P>(tot, run);                               // tot = sum (l+1) * B[base+l]
    }
    if (base) {                                                    // + base * run (double-and-add, base < 2^19)
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
                                                 char* __restrict__ wsum, int wave_prio) {
  raise_wave_priority(wave_prio);
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

// ---- bucket reduction as a matrix (mat_geom) ----
// phase 1: every row sum and column sum of every set; Q quads share a sum (strided), LDS tree within the Q quads
template <class P>
__global__ __launch_bounds__(256) void k_red_sums(const char* __restrict__ bucket_acc, uint32_t nbk, uint32_t Lb, uint32_t Q,
                                                  uint32_t wgs_per_set, char* __restrict__ sums, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ __align__(16) char lds_raw[64 * 128];
  const uint32_t ncols = 1u << Lb, nrows = nbk >> Lb, nsums = nrows + ncols;
  const uint32_t set = blockIdx.x / wgs_per_set, blk = blockIdx.x % wgs_per_set;
  const uint32_t lt = threadIdx.x >> 2, sub = lt / Q, qi = lt % Q;
  const uint32_t sidx = blk * (64 / Q) + sub;                      // which sum of the set
  const char* bp = bucket_acc + (size_t)set * nbk * 128;
  QPoint<P> acc = qpoint_identity<P>();
  if (sidx < nrows) {
    for (uint32_t l = qi; l < ncols; l += Q) acc = qpoint_add<P>(acc, qpoint_load_lazy<P>(bp + ((size_t)sidx * ncols + l) * 128));
  } else if (sidx < nsums) {
    const uint32_t l = sidx - nrows;
    for (uint32_t h = qi; h < nrows; h += Q) acc = qpoint_add<P>(acc, qpoint_load_lazy<P>(bp + ((size_t)h * ncols + l) * 128));
  }
  qpoint_store<P>(lds_raw + (size_t)lt * 128, acc);
  __syncthreads();
  for (uint32_t stride = Q >> 1; stride >= 1; stride >>= 1) {
    if (qi < stride) {
      acc = qpoint_add<P>(acc, qpoint_load<P>(lds_raw + (size_t)(lt + stride) * 128));
      qpoint_store<P>(lds_raw + (size_t)lt * 128, acc);
    }
    __syncthreads();
  }
  if (qi == 0 && sidx < nsums) qpoint_store<P>(sums + ((size_t)set * nsums + sidx) * 128, acc);
}

// phase 2: one quad per sum: (l+1) C_l or h R_h by double-and-add, then an LDS tree; a workgroup holds sums of one
// kind of one set, so the two kinds come out as separate partials (the rows still lack their common factor 2^Lb)
template <class P>
__global__ __launch_bounds__(256) void k_red_weights(const char* __restrict__ sums, uint32_t nbk, uint32_t Lb, uint32_t bA,
                                                     uint32_t bB, char* __restrict__ partials, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ __align__(16) char lds_raw[64 * 128];
  const uint32_t ncols = 1u << Lb, nrows = nbk >> Lb, nsums = nrows + ncols;
  const uint32_t set = blockIdx.x / (bA + bB), blk = blockIdx.x % (bA + bB);
  const uint32_t lt = threadIdx.x >> 2;
  const bool col = blk < bA;
  const uint32_t idx = (col ? blk : blk - bA) * 64 + lt;           // column l or row h
  const uint32_t weight = col ? idx + 1 : idx;
  QPoint<P> tot = qpoint_identity<P>();
  if (idx < (col ? ncols : nrows) && weight) {
    // signed double-and-add (non-adjacent form, read off 3w and w): at most one addition per two doublings, and the
    // slowest quad of the workgroup sets the pace
    const QPoint<P> pt = qpoint_load<P>(sums + ((size_t)set * nsums + (col ? nrows + idx : idx)) * 128);
    const QPoint<P> mpt = qpoint_neg<P>(pt);
    const uint32_t h3 = 3u * weight;
    for (int bit = 31 - __builtin_clz(h3); bit >= 1; --bit) {
      tot = qpoint_dbl<P>(tot);
      const uint32_t hb = (h3 >> bit) & 1u, wb = (weight >> bit) & 1u;
      if (hb != wb) tot = qpoint_add<P>(tot, hb ? pt : mpt);
    }
  }
  qpoint_store<P>(lds_raw + (size_t)lt * 128, tot);
  __syncthreads();
  for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
    if (lt < stride) {
      tot = qpoint_add<P>(tot, qpoint_load<P>(lds_raw + (size_t)(lt + stride) * 128));
      qpoint_store<P>(lds_raw + (size_t)lt * 128, tot);
    }
    __syncthreads();
  }
  if (lt == 0) qpoint_store<P>(partials + ((size_t)set * (bA + bB) + blk) * 128, tot);
}

// phase 3: per set, the column partials + 2^Lb * the row partials (quads 0..31 / 32..63, a tree in each half)
// out_jac != nullptr (one bucket set per group): the set's sum is the group's result and leaves as a Jacobian point,
// k_final's work for that case
template <class P>
__global__ __launch_bounds__(256) void k_red_combine(const char* __restrict__ partials, uint32_t Lb, uint32_t bA, uint32_t bB,
                                                     char* __restrict__ wsum, char* __restrict__ out_jac, int wave_prio) {
  raise_wave_priority(wave_prio);
  __shared__ __align__(16) char lds_raw[64 * 128];
  const uint32_t set = blockIdx.x;
  const uint32_t lt = threadIdx.x >> 2, half = lt >> 5, qi = lt & 31u;
  const char* base = partials + (size_t)set * (bA + bB) * 128 + (half ? (size_t)bA * 128 : 0);
  const uint32_t count = half ? bB : bA;
  QPoint<P> acc = qpoint_identity<P>();
  for (uint32_t i = qi; i < count; i += 32) acc = qpoint_add<P>(acc, qpoint_load<P>(base + (size_t)i * 128));
  qpoint_store<P>(lds_raw + (size_t)lt * 128, acc);
  __syncthreads();
  for (uint32_t stride = 16; stride >= 1; stride >>= 1) {
    if (qi < stride) {
      acc = qpoint_add<P>(acc, qpoint_load<P>(lds_raw + (size_t)(lt + stride) * 128));
      qpoint_store<P>(lds_raw + (size_t)lt * 128, acc);
    }
    __syncthreads();
  }
  if (lt == 32) {
    for (uint32_t k = 0; k < Lb; ++k) acc = qpoint_dbl<P>(acc);
    qpoint_store<P>(lds_raw + (size_t)32 * 128, acc);
  }
  __syncthreads();
  if (lt == 0) {
    const QPoint<P> res = qpoint_add<P>(acc, qpoint_load<P>(lds_raw + (size_t)32 * 128));
    if (!out_jac) {
      qpoint_store<P>(wsum + (size_t)set * 128, res);
    } else {
      const Jac<P> j = xyzz_to_jac(qpoint_gather(res));
      if (threadIdx.x == 0) {
        fe_store<P>(out_jac + (size_t)set * 96, j.x);
        fe_store<P>(out_jac + (size_t)set * 96 + 32, j.y);
        fe_store<P>(out_jac + (size_t)set * 96 + 64, j.z);
      }
    }
  }
}

// Horner over a group's bucket sets (one quad per group, one workgroup each), XYZZ -> Jacobian
template <class P>
__global__ __launch_bounds__(64) void k_final(const char* __restrict__ wsum_all, int sets, int c, char* __restrict__ out_all, int wave_prio) {
  raise_wave_priority(wave_prio);
  if (threadIdx.x >= 4) return;
  const char* wsum = wsum_all + (size_t)blockIdx.x * sets * 128;
  char* out_jac = out_all + (size_t)blockIdx.x * 96;
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

// Square root in a Pasta field (2-adicity 32) by Tonelli-Shanks; ok = false when a is a non-residue.
template <class P>
__device__ bool fe_sqrt(const Fe<P>& a, Fe<P>& root) {
  if (fe_is_zero(a)) { root = a; return true; }
  uint32_t e[8];
  Fe<P> z;
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = P::TS_EXP[i]; z.v[i] = P::TS_Z[i]; }
  const Fe<P> w = fe_pow(a, e);                 // a^((T-1)/2)
  Fe<P> x = fe_mul(a, w);                       // a^((T+1)/2)
  Fe<P> b = fe_mul(x, w);                       // a^T: order divides 2^32
  const Fe<P> one = fe_one<P>();
  int v = 32;
  while (!fe_eq(b, one)) {
    int k = 0;
    Fe<P> t = b;
    while (!fe_eq(t, one)) { t = fe_sqr(t); ++k; if (k == v) return false; }   // order 2^v: non-residue
    Fe<P> zz = z;
    for (int i = 0; i < v - k - 1; ++i) zz = fe_sqr(zz);
    x = fe_mul(x, zz);
    z = fe_sqr(zz);
    b = fe_mul(b, z);
    v = k;
  }
  root = x;
  return true;
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

// Generator family 1, "try and increment" (SURVEY.md 8d, config 2): per index a xoshiro256** stream seeded with
// four splitmix64 words of (seed, index); a candidate x is 256 stream bits reduced mod m; it is accepted when
// x^3 + 5 is a square, and y is the root whose canonical value is even.  Discrete logarithms are unknown, as for
// generators derived from a hash.
template <class P>
__global__ __launch_bounds__(256) void k_bases_generate_tai(uint64_t seed, uint64_t start, uint32_t n, char* __restrict__ pts) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t st[4];
  uint64_t sm = seed * 0xD1342543DE82EF95ull + (start + i) * 0x9E3779B97F4A7C15ull;
  for (int k = 0; k < 4; ++k) { st[k] = splitmix64(sm); sm += 0x9E3779B97F4A7C15ull; }
  Fe<P> five;
#pragma unroll
  for (int l = 0; l < 8; ++l) five.v[l] = P::FIVE[l];
  for (;;) {
    Fe<P> x;
    for (int k = 0; k < 4; ++k) {                                  // xoshiro256** (Blackman, Vigna; public domain)
      const uint64_t r = rotl64(st[1] * 5, 7) * 9, t = st[1] << 17;
      st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3]; st[2] ^= t; st[3] = rotl64(st[3], 45);
      x.v[2 * k] = (uint32_t)r; x.v[2 * k + 1] = (uint32_t)(r >> 32);
    }
    for (int k = 0; k < 3; ++k) {                                  // 2^256 < 4m: at most three subtractions of m
      if (fe_is_canonical(x)) break;
      uint32_t borrow = 0;
      for (int l = 0; l < 8; ++l) {
        const uint64_t d = (uint64_t)x.v[l] - P::MOD[l] - borrow;
        x.v[l] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
      }
    }
    const Fe<P> xm = fe_to_mont(x);
    const Fe<P> rhs = fe_add(fe_mul(fe_sqr(xm), xm), five);
    Fe<P> y;
    if (!fe_sqrt(rhs, y)) continue;
    if (fe_from_mont(y).v[0] & 1u) y = fe_neg(y);
    Affine<P> a; a.x = xm; a.y = y;
    affine_store<P>(pts + (size_t)i * 64, a);
    return;
  }
}

// ---- generator family 2: derived from a label, the way nova-snark derives its CommitGens (label -> SHAKE256 stream ->
// curve points).  Its own constants are not in /root/reference, so the encoding is this build's:
//   block = "vdf-gens-v1" | curve u8 | len u8 | label[len <= 64] | index LE64 | counter LE32       (one SHAKE256 block)
//   x = first 64 output bytes as a little-endian integer mod m;  accepted if x^3 + 5 is a square;  y = the even root;
//   counter = 0, 1, ... until accepted.  Restated in oracle/pasta.py label_base.
__device__ __forceinline__ void keccak_f1600(uint64_t s[25]) {
  const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int round = 0; round < 24; ++round) {
    uint64_t C[5], D[5], B[25];
#pragma unroll
    for (int x = 0; x < 5; ++x) C[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
#pragma unroll
    for (int x = 0; x < 5; ++x) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
#pragma unroll
    for (int i = 0; i < 25; ++i) s[i] ^= D[i % 5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int y = 0; y < 5; ++y) {
        const int i = x + 5 * y;
        B[y + 5 * ((2 * x + 3 * y) % 5)] = ROT[i] ? rotl64(s[i], ROT[i]) : s[i];
      }
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) s[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y]);
    s[0] ^= RC[round];
  }
}

struct GenLabel { uint8_t bytes[64]; uint32_t len; };

template <class P>
__global__ __launch_bounds__(256) void k_bases_generate_label(GenLabel label, uint32_t curve, uint64_t start, uint32_t n,
                                                              char* __restrict__ pts) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fe<P> five, r2;
#pragma unroll
  for (int l = 0; l < 8; ++l) { five.v[l] = P::FIVE[l]; r2.v[l] = P::R2[l]; }
  uint8_t msg[136];
  const char head[12] = "vdf-gens-v1";
  for (uint32_t ctr = 0;; ++ctr) {
    for (int k = 0; k < 136; ++k) msg[k] = 0;
    int pos = 0;
    for (int k = 0; k < 11; ++k) msg[pos++] = (uint8_t)head[k];
    msg[pos++] = (uint8_t)curve;
    msg[pos++] = (uint8_t)label.len;
    for (uint32_t k = 0; k < label.len; ++k) msg[pos++] = label.bytes[k];
    const uint64_t idx = start + i;
    for (int k = 0; k < 8; ++k) msg[pos++] = (uint8_t)(idx >> (8 * k));
    for (int k = 0; k < 4; ++k) msg[pos++] = (uint8_t)(ctr >> (8 * k));
    msg[pos] ^= 0x1F;                                   // SHAKE domain bits + first padding bit
    msg[135] ^= 0x80;
    uint64_t st[25];
    for (int w = 0; w < 25; ++w) st[w] = 0;
    for (int w = 0; w < 17; ++w) {
      uint64_t v = 0;
      for (int k = 0; k < 8; ++k) v |= (uint64_t)msg[8 * w + k] << (8 * k);
      st[w] = v;
    }
    keccak_f1600(st);
    // 64 output bytes = lo | hi (256 bits each); x = lo + hi * 2^256 mod m
    Fe<P> lo, hi;
    for (int k = 0; k < 4; ++k) {
      lo.v[2 * k] = (uint32_t)st[k]; lo.v[2 * k + 1] = (uint32_t)(st[k] >> 32);
      hi.v[2 * k] = (uint32_t)st[4 + k]; hi.v[2 * k + 1] = (uint32_t)(st[4 + k] >> 32);
    }
    for (int half = 0; half < 2; ++half) {
      Fe<P>& x = half ? hi : lo;
      for (int k = 0; k < 3; ++k) {                     // 2^256 < 4m: at most three subtractions of m
        if (fe_is_canonical(x)) break;
        uint32_t borrow = 0;
        for (int l = 0; l < 8; ++l) {
          const uint64_t d = (uint64_t)x.v[l] - P::MOD[l] - borrow;
          x.v[l] = (uint32_t)d;
          borrow = (uint32_t)(d >> 63);
        }
      }
    }
    const Fe<P> xm = fe_add(fe_to_mont(lo), fe_mul(fe_to_mont(hi), r2));      // Montgomery form of lo + hi * R
    const Fe<P> rhs = fe_add(fe_mul(fe_sqr(xm), xm), five);
    Fe<P> y;
    if (fe_is_zero(xm) || !fe_sqrt(rhs, y)) continue;
    if (fe_from_mont(y).v[0] & 1u) y = fe_neg(y);
    Affine<P> a; a.x = xm; a.y = y;
    affine_store<P>(pts + (size_t)i * 64, a);
    return;
  }
}

// flags[0] |= 1: a coordinate is not a canonical residue; |= 2: a canonical point that is neither the identity
// (0, 0) nor on y^2 = x^3 + 5.  flags[1] = smallest offending index.
template <class P>
__global__ __launch_bounds__(256) void k_validate_points(const char* __restrict__ pts, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Affine<P> a = affine_load<P>(pts + (size_t)i * 64);
  uint32_t bad = 0;
  if (!fe_is_canonical(a.x) || !fe_is_canonical(a.y)) bad = 1;
  else if (!affine_is_identity(a)) {
    Fe<P> five;
#pragma unroll
    for (int l = 0; l < 8; ++l) five.v[l] = P::FIVE[l];
    if (!fe_eq(fe_sqr(a.y), fe_add(fe_mul(fe_sqr(a.x), a.x), five))) bad = 2;
  }
  if (bad) { atomicOr(&flags[0], bad); atomicMin(&flags[1], i); }
}

// ---- the curves' endomorphism for the table-less path (vdf_hip_tuning.glv) --------------------------------------------
// y^2 = x^3 + 5 has phi(x, y) = (zeta x, y) = [lambda] (x, y) with zeta, lambda primitive cube roots of unity in the base and
// scalar field.  k = k1 + lambda k2 with |k1|, |k2| < 2^129 (pasta_constants.h: the lattice and its rounding constants), so
// sum k_i P_i = sum k1_i P_i + k2_i phi(P_i): twice the points, HALF the scalar length -- the same number of bucket additions,
// 9 bucket sets instead of 16 and a Horner chain of 128 doublings instead of 240 (the table-less path's tail was 42 % of
// its time, all of it that chain).
// out[i] = P_i, out[n + i] = phi(P_i)
template <class P>
__global__ __launch_bounds__(256) void k_glv_points(const char* __restrict__ pts, uint32_t n, char* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Affine<P> a = affine_load<P>(pts + (size_t)i * 64);
  affine_store<P>(out + (size_t)i * 64, a);
  Affine<P> b = a;
  Fe<P> zeta;
#pragma unroll
  for (int l = 0; l < 8; ++l) zeta.v[l] = P::GLV_ZETA[l];
  b.x = fe_mul(a.x, zeta);                                  // (0, 0), the identity, stays (0, 0)
  affine_store<P>(out + ((size_t)n + i) * 64, b);
}
// round(a * b / 2^382) for two 256-bit integers, as five limbs
__device__ __forceinline__ void mul_shift_382(const uint32_t a[8], const uint32_t b[8], uint32_t c[5]) {
  uint32_t p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t t = (uint64_t)a[i] * b[j] + p[i + j] + carry;
      p[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    p[i + 8] = carry;
  }
  // + 2^381: the quotient is ROUNDED, not truncated -- the half-scalars then stay below (|a1| + |a2|) / 2 < 2^127, so that the
  // eighth 16-bit window's digit never carries (a carry would put half of all scalars into one bucket of a ninth window)
  uint64_t cy = (uint64_t)p[11] + (1u << 29);
  p[11] = (uint32_t)cy; cy >>= 32;
#pragma unroll
  for (int i = 12; i < 16; ++i) { cy += p[i]; p[i] = (uint32_t)cy; cy >>= 32; }
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = (p[11 + j] >> 30) | (p[12 + j] << 2);
  c[4] = p[15] >> 30;
}
// acc (8 limbs, mod 2^256) += or -= c (5 limbs) * m (5 limbs)
template <bool SUB>
__device__ __forceinline__ void muladd_5x5(uint32_t acc[8], const uint32_t c[5], const uint32_t m[5]) {
  uint32_t p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (i + j < 8) {
        const uint64_t t = (uint64_t)c[i] * m[j] + p[i + j] + carry;
        p[i + j] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
      }
    }
    if (i + 5 < 8) p[i + 5] = carry;
  }
  uint64_t cy = 0;                                            // carry / borrow
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (SUB) { const uint64_t d = (uint64_t)acc[i] - p[i] - cy; acc[i] = (uint32_t)d; cy = (d >> 63) & 1u; }
    else { const uint64_t d = (uint64_t)acc[i] + p[i] + cy; acc[i] = (uint32_t)d; cy = d >> 32; }
  }
}
// a two's-complement 256-bit integer of magnitude below 2^130 -> magnitude | sign << 255
__device__ __forceinline__ void to_sign_magnitude(uint32_t v[8]) {
  const uint32_t neg = v[7] >> 31;
  if (neg) {
    uint64_t cy = 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint64_t d = (uint64_t)(~v[i]) + cy; v[i] = (uint32_t)d; cy = d >> 32; }
  }
  v[7] = (v[7] & 0x7FFFFFFFu) | (neg << 31);
}
// SP = the scalar field.  out[i] = k1_i, out[n + i] = k2_i as sign-and-magnitude words
template <class SP>
__global__ __launch_bounds__(256) void k_glv_split(const uint32_t* __restrict__ scalars, uint32_t n, int is_mont, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fe<SP> k = fe_load<SP>(scalars + (size_t)i * 8);
  if (is_mont) k = fe_from_mont(k);
  uint32_t g1[8], g2[8], a1[5], a2[5], nb1[5], b2[5], c1[5], c2[5];
#pragma unroll
  for (int l = 0; l < 8; ++l) { g1[l] = SP::GLV_G1[l]; g2[l] = SP::GLV_G2[l]; }
#pragma unroll
  for (int l = 0; l < 5; ++l) { a1[l] = SP::GLV_A1[l]; a2[l] = SP::GLV_A2[l]; nb1[l] = SP::GLV_NB1[l]; b2[l] = SP::GLV_B2[l]; }
  mul_shift_382(k.v, g1, c1);
  mul_shift_382(k.v, g2, c2);
  uint32_t k1[8], k2[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) { k1[l] = k.v[l]; k2[l] = 0; }
  muladd_5x5<true>(k1, c1, a1);
  muladd_5x5<true>(k1, c2, a2);
  muladd_5x5<false>(k2, c1, nb1);
  muladd_5x5<true>(k2, c2, b2);
  to_sign_magnitude(k1);
  to_sign_magnitude(k2);
#pragma unroll
  for (int l = 0; l < 8; ++l) { out[(size_t)i * 8 + l] = k1[l]; out[((size_t)n + i) * 8 + l] = k2[l]; }
}
Status glv_points(int curve, const void* d_pts, size_t n, void* d_out, hipStream_t stream) {
  const dim3 grid((unsigned)((n + 255) / 256));
  if (curve == VDF_CURVE_PALLAS) hipLaunchKernelGGL((k_glv_points<FpParams>), grid, dim3(256), 0, stream, reinterpret_cast<const char*>(d_pts), (uint32_t)n, reinterpret_cast<char*>(d_out));
  else if (curve == VDF_CURVE_VESTA) hipLaunchKernelGGL((k_glv_points<FqParams>), grid, dim3(256), 0, stream, reinterpret_cast<const char*>(d_pts), (uint32_t)n, reinterpret_cast<char*>(d_out));
  else return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}
Status glv_split(int curve, const void* d_scalars, size_t n, bool is_mont, void* d_out, hipStream_t stream) {
  const dim3 grid((unsigned)((n + 255) / 256));
  KTimer kt(stream, "k_glv_split", 96.0 * n);
  if (curve == VDF_CURVE_PALLAS) hipLaunchKernelGGL((k_glv_split<FqParams>), grid, dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(d_scalars), (uint32_t)n, is_mont ? 1 : 0, reinterpret_cast<uint32_t*>(d_out));
  else if (curve == VDF_CURVE_VESTA) hipLaunchKernelGGL((k_glv_split<FpParams>), grid, dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(d_scalars), (uint32_t)n, is_mont ? 1 : 0, reinterpret_cast<uint32_t*>(d_out));
  else return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
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
template <class P>
static Status msm_tail_t(int c, int sets, int groups, uint32_t nbk, const char* bucket_acc, char* partials, char* wsum,
                         void* d_out, hipStream_t st, int prio) {
  const uint32_t gsets = (uint32_t)(groups * sets);
  if (use_matrix_reduction(nbk)) {
    const MatGeom g = mat_geom(gsets, nbk);
    char* sums = partials;
    char* parts = partials + g.sums_bytes;
    hipLaunchKernelGGL((k_red_sums<P>), dim3(gsets * g.wgs_per_set), dim3(256), 0, st, bucket_acc, nbk, g.Lb, g.Q, g.wgs_per_set, sums, prio);
    hipLaunchKernelGGL((k_red_weights<P>), dim3(gsets * (g.bA + g.bB)), dim3(256), 0, st, sums, nbk, g.Lb, g.bA, g.bB, parts, prio);
    char* direct = sets == 1 ? reinterpret_cast<char*>(d_out) : nullptr;
    hipLaunchKernelGGL((k_red_combine<P>), dim3(gsets), dim3(256), 0, st, parts, g.Lb, g.bA, g.bB, wsum, direct, prio);
    if (direct) {
      VDF_TRY_HIP(hipGetLastError());
      return Status{};
    }
  } else {
    const RedGeom rg = red_geom((size_t)gsets * nbk, nbk);
    hipLaunchKernelGGL((k_reduce1<P>), dim3(gsets * rg.blocks_per_set), dim3(rg.block * 4), (size_t)rg.block * 128, st,
                       bucket_acc, nbk, rg.seg, rg.threads_per_set, rg.blocks_per_set, partials, prio);
    hipLaunchKernelGGL((k_reduce2<P>), dim3(gsets), dim3(256), 0, st, partials, rg.blocks_per_set, wsum, prio);
  }
  hipLaunchKernelGGL((k_final<P>), dim3(groups), dim3(64), 0, st, wsum, sets, c, reinterpret_cast<char*>(d_out), prio);
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

// ext_bucket_acc != nullptr: bucket accumulators live there (a job's shared array) and the run stops after the
// fix-up; the reduction of all groups follows in msm_tail.
template <class P, class SP>
static Status msm_run_t(const MsmPlan& p, const void* d_points, const void* const* d_scalars, bool is_mont, void* ws,
                        void* d_out, hipStream_t st, hipEvent_t* ev, char* ext_bucket_acc, hipEvent_t acc_gate, int prio) {
  const WsLayout w = ws_layout(p);
  char* base = reinterpret_cast<char*>(ws);
  uint32_t* countsA = reinterpret_cast<uint32_t*>(base + w.countsA);
  uint32_t* pcount = reinterpret_cast<uint32_t*>(base + w.pcount);
  uint32_t* pstart = reinterpret_cast<uint32_t*>(base + w.pstart);
  // records of pass A, split: 4-byte payloads (point | table | sign), then 2-byte fine keys -- 6 bytes per entry over the
  // passes where the packed 8-byte record moved 8 (written once, read twice: 18 instead of 28 bytes per entry with the output)
  uint32_t* recsP = reinterpret_cast<uint32_t*>(base + w.recs);
  uint16_t* recsK = reinterpret_cast<uint16_t*>(base + w.recs + (size_t)p.windows * p.n * 4);
  uint32_t* bstart = reinterpret_cast<uint32_t*>(base + w.bstart);
  uint32_t* sorted = reinterpret_cast<uint32_t*>(base + w.sorted);
  char* bucket_acc = ext_bucket_acc ? ext_bucket_acc : base + w.bucket_acc;
  char* heads = base + w.heads;
  uint32_t* heavy = reinterpret_cast<uint32_t*>(base + w.heavy);
  char* partials = base + w.partials;
  char* wsum = base + w.wsum;
  const uint32_t nkeys = (uint32_t)p.gsets * p.nbk;
  const uint32_t nf = 1u << p.fb;
  PartGroups pg{};
  pg.groups = p.groups;
  pg.signed_scalars = p.signed_scalars ? 1 : 0;
  for (int g = 0; g < p.groups; ++g) {
    pg.scalars[g] = reinterpret_cast<const uint32_t*>(d_scalars[g]);
    pg.n[g] = p.gn[g]; pg.off[g] = p.goff[g]; pg.blk_end[g] = p.gblk_end[g];
  }
  const size_t lds_bins = (size_t)p.bins * 4;

  if (ev) VDF_TRY_HIP(hipEventRecord(ev[0], st));
  // pass A
  {
  KTimer kt(st, "msm_sort(5 launches)", 0.0);
  hipLaunchKernelGGL((k_part<SP, false>), dim3(p.nblkA), dim3(256), lds_bins, st, pg, is_mont ? 1 : 0, p.c, p.windows,
                     p.sets, p.pb, p.fb, p.bins, p.chA, p.tstride, countsA, pstart, recsP, recsK, prio);
  hipLaunchKernelGGL(k_part_scan, dim3(p.bins), dim3(256), 0, st, countsA, p.bins, p.nblkA, pcount, heavy, heavy + 1 + nkeys + 2, prio);
  hipLaunchKernelGGL(k_scan_keys, dim3(1), dim3(1024), 0, st, pcount, p.bins, pstart, prio);
  if (tuning().sort_staged && p.bins <= PART_STAGE_BINS && p.windows <= 16)
    hipLaunchKernelGGL((k_part_staged<SP>), dim3(p.nblkA), dim3(256), 0, st, pg, is_mont ? 1 : 0, p.c, p.windows,
                       p.sets, p.pb, p.fb, p.bins, p.chA, p.tstride, countsA, pstart, recsP, recsK, prio);
  else
    hipLaunchKernelGGL((k_part<SP, true>), dim3(p.nblkA), dim3(256), lds_bins, st, pg, is_mont ? 1 : 0, p.c, p.windows,
                       p.sets, p.pb, p.fb, p.bins, p.chA, p.tstride, countsA, pstart, recsP, recsK, prio);
  // pass B
  if (tuning().sort_staged && nf <= 256)
    hipLaunchKernelGGL(k_fine_staged<4>, dim3(p.bins), dim3(1024), 0, st, recsP, recsK, pstart, p.bins, nf, bstart, sorted, bucket_acc, p.slots, p.Lfixed,
                       reinterpret_cast<uint32_t*>(base + w.tstart), prio);
  else if (tuning().sort_staged)
    hipLaunchKernelGGL(k_fine_staged<8>, dim3(p.bins), dim3(1024), 0, st, recsP, recsK, pstart, p.bins, nf, bstart, sorted, bucket_acc, p.slots, p.Lfixed,
                       reinterpret_cast<uint32_t*>(base + w.tstart), prio);
  else
    hipLaunchKernelGGL(k_fine, dim3(p.bins), dim3(1024), 0, st, recsP, recsK, pstart, p.bins, nf, bstart, sorted, bucket_acc, p.slots, p.Lfixed,
                       reinterpret_cast<uint32_t*>(base + w.tstart), prio);
  }
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[1], st));
  // the sort is light; the accumulation fills every SIMD: a caller that knows of latency-critical work on another queue
  // holds it back until that work's mark (vdf_ctx_gate_accumulate)
  if (acc_gate) VDF_TRY_HIP(hipStreamWaitEvent(st, acc_gate, 0));
  {
  KTimer kt(st, "k_accumulate", 96.0 * p.n);            // the pipeline's algorithmic bytes: 96 B per (base, scalar) pair, SURVEY.md 8d
  // (tuning) vdf_hip_tuning.accumulate_lds: unused dynamic LDS per workgroup, which caps how many accumulate workgroups a CU
  // takes (160 KB per CU: 54 KB -> two) whatever their register count allows -- the dispatcher then cannot pack three onto
  // one CU and one onto another when other queues hold slots (a CU with three takes 1.5 x as long: the launch's tail)
  const unsigned acc_lds = (unsigned)(tuning().accumulate_lds >= 0 && tuning().accumulate_lds <= 65536 ? tuning().accumulate_lds : 0);
  hipLaunchKernelGGL((k_accumulate<P>), dim3((p.nthreads + 255) / 256), dim3(256), acc_lds, st, sorted, bstart, nkeys,
                     reinterpret_cast<const uint32_t*>(base + w.tstart), reinterpret_cast<const char*>(d_points), bucket_acc, heads, p.slots, p.Lfixed, p.nthreads);
  }
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[2], st));
  const FixupTune tune = fixup_tune();
  KTimer kt_tail(st, ext_bucket_acc ? "msm_fixup(2 launches)" : "msm_tail(fixup+reduce)", 0.0);
  if (tuning().fixup_serial)
    hipLaunchKernelGGL((k_fixup_serial<P>), dim3((nkeys + 255) / 256), dim3(256), 0, st, bstart, nkeys, p.slots, p.Lfixed, bucket_acc, heads,
                       heavy, tune.heavy_min, tune.giant_span, prio);
  else
    hipLaunchKernelGGL((k_fixup<P>), dim3((nkeys * 4 + 255) / 256), dim3(256), 0, st, bstart, nkeys, p.slots, p.Lfixed, bucket_acc, heads,
                       heavy, tune.heavy_min, tune.giant_span, prio);
  hipLaunchKernelGGL((k_fixup_heavy<P>), dim3(16 * GIANT_PARTS), dim3(64), 0, st, bstart, nkeys, p.slots, p.Lfixed, bucket_acc, heads, heavy,
                     base + w.giant, prio);
  if (!ext_bucket_acc) VDF_TRY(msm_tail_t<P>(p.c, p.sets, p.groups, p.nbk, bucket_acc, partials, wsum, d_out, st, prio));
  if (ev) VDF_TRY_HIP(hipEventRecord(ev[3], st));
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status msm_run(int curve, const MsmPlan& plan, const void* d_points, const void* const* d_scalars, bool is_mont, void* ws,
               void* d_out, hipStream_t stream, hipEvent_t* ev, void* ext_bucket_acc, hipEvent_t acc_gate, int prio) {
  // Pallas: coordinates in Fp, scalars in Fq.  Vesta: coordinates in Fq, scalars in Fp.
  char* ext = reinterpret_cast<char*>(ext_bucket_acc);
  prio = std::min(prio, (int)tuning().light_priority);     // the process-wide ceiling holds on EVERY path that gets here (jobs too)
  if (curve == VDF_CURVE_PALLAS)
    return msm_run_t<FpParams, FqParams>(plan, d_points, d_scalars, is_mont, ws, d_out, stream, ev, ext, acc_gate, prio);
  if (curve == VDF_CURVE_VESTA)
    return msm_run_t<FqParams, FpParams>(plan, d_points, d_scalars, is_mont, ws, d_out, stream, ev, ext, acc_gate, prio);
  return Status{VDF_ERR_BAD_ARG, "unknown curve"};
}

// Stand-alone bucket reduction over `groups` x `sets` bucket sets laid out back to back (a job's shared array).
// tail_ws: msm_tail_ws_bytes() bytes = the bucket array followed by the reduction scratch.
static size_t tail_scratch_bytes(size_t gsets, uint32_t nbk) {
  const RedGeom rg = red_geom(gsets * nbk, nbk);
  const size_t seg = align_up(gsets * rg.blocks_per_set * 128, 256), mat = mat_geom(gsets, nbk).scratch_bytes;
  return seg > mat ? seg : mat;
}
size_t msm_tail_ws_bytes(int groups, int sets, uint32_t nbk) {
  const size_t gsets = (size_t)groups * sets;
  return align_up(gsets * nbk * 128, 256) + tail_scratch_bytes(gsets, nbk) + align_up(gsets * 128, 256);
}
Status msm_tail(int curve, int c, int sets, int groups, uint32_t nbk, void* tail_ws, void* d_out, hipStream_t stream, int prio) {
  const size_t gsets = (size_t)groups * sets;
  prio = std::min(prio, (int)tuning().light_priority);
  char* bucket_acc = reinterpret_cast<char*>(tail_ws);
  char* partials = bucket_acc + align_up(gsets * nbk * 128, 256);
  char* wsum = partials + tail_scratch_bytes(gsets, nbk);
  if (curve == VDF_CURVE_PALLAS) return msm_tail_t<FpParams>(c, sets, groups, nbk, bucket_acc, partials, wsum, d_out, stream, prio);
  if (curve == VDF_CURVE_VESTA) return msm_tail_t<FqParams>(c, sets, groups, nbk, bucket_acc, partials, wsum, d_out, stream, prio);
  return Status{VDF_ERR_BAD_ARG, "unknown curve"};
}

// sum of n Jacobian points: one wavefront = 16 quads striding over the inputs, butterfly reduce
template <class P>
__global__ __launch_bounds__(64) void k_point_sum(const char* __restrict__ pts, uint32_t n, char* __restrict__ out, int wave_prio) {
  raise_wave_priority(wave_prio);
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
  const int prio = std::min(3, (int)tuning().light_priority);
  if (curve == VDF_CURVE_PALLAS)
    hipLaunchKernelGGL((k_point_sum<FpParams>), dim3(1), dim3(64), 0, stream, reinterpret_cast<const char*>(d_jac),
                       (uint32_t)n, reinterpret_cast<char*>(d_out), prio);
  else if (curve == VDF_CURVE_VESTA)
    hipLaunchKernelGGL((k_point_sum<FqParams>), dim3(1), dim3(64), 0, stream, reinterpret_cast<const char*>(d_jac),
                       (uint32_t)n, reinterpret_cast<char*>(d_out), prio);
  else
    return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status bases_generate(int curve, int family, uint64_t seed, size_t start, size_t n, void* d_pts, hipStream_t stream) {
  if (n == 0) return Status{};
  dim3 grid((unsigned)((n + 255) / 256));
  if (family == VDF_GENS_TRY_AND_INCREMENT) {
    if (curve == VDF_CURVE_PALLAS)
      hipLaunchKernelGGL((k_bases_generate_tai<FpParams>), grid, dim3(256), 0, stream, seed, (uint64_t)start, (uint32_t)n,
                         reinterpret_cast<char*>(d_pts));
    else if (curve == VDF_CURVE_VESTA)
      hipLaunchKernelGGL((k_bases_generate_tai<FqParams>), grid, dim3(256), 0, stream, seed, (uint64_t)start, (uint32_t)n,
                         reinterpret_cast<char*>(d_pts));
    else
      return Status{VDF_ERR_BAD_ARG, "unknown curve"};
    VDF_TRY_HIP(hipGetLastError());
    return Status{};
  }
  if (family != VDF_GENS_KNOWN_DLOG) return Status{VDF_ERR_BAD_ARG, "unknown generator family"};
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

Status bases_generate_label(int curve, const uint8_t* label, size_t len, size_t start, size_t n, void* d_pts, hipStream_t stream) {
  if (n == 0) return Status{};
  if (len > 64 || (len && !label)) return Status{VDF_ERR_BAD_LENGTH, "label of at most 64 bytes"};
  GenLabel gl;
  std::memset(&gl, 0, sizeof(gl));
  if (len) std::memcpy(gl.bytes, label, len);
  gl.len = (uint32_t)len;
  dim3 grid((unsigned)((n + 255) / 256));
  if (curve == VDF_CURVE_PALLAS)
    hipLaunchKernelGGL((k_bases_generate_label<FpParams>), grid, dim3(256), 0, stream, gl, (uint32_t)curve, (uint64_t)start, (uint32_t)n,
                       reinterpret_cast<char*>(d_pts));
  else if (curve == VDF_CURVE_VESTA)
    hipLaunchKernelGGL((k_bases_generate_label<FqParams>), grid, dim3(256), 0, stream, gl, (uint32_t)curve, (uint64_t)start, (uint32_t)n,
                       reinterpret_cast<char*>(d_pts));
  else
    return Status{VDF_ERR_BAD_ARG, "unknown curve"};
  VDF_TRY_HIP(hipGetLastError());
  return Status{};
}

Status bases_validate(int curve, const void* d_pts, size_t n, uint32_t* d_flags, hipStream_t stream) {
  if (n == 0) return Status{};
  dim3 grid((unsigned)((n + 255) / 256));
  if (curve == VDF_CURVE_PALLAS)
    hipLaunchKernelGGL((k_validate_points<FpParams>), grid, dim3(256), 0, stream, reinterpret_cast<const char*>(d_pts), (uint32_t)n, d_flags);
  else if (curve == VDF_CURVE_VESTA)
    hipLaunchKernelGGL((k_validate_points<FqParams>), grid, dim3(256), 0, stream, reinterpret_cast<const char*>(d_pts), (uint32_t)n, d_flags);
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
