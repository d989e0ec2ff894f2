// Internal declarations shared by the translation units of libvdf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/vdf_hip.h"

namespace vdf {
// Per-launch timing for the roofline report (vdf_ctx_set_kernel_timing): a launch site constructs a KTimer around its
// launch; when the calling thread has a sink (set by the ABI's guard while the context's kernel timing is on) two HIP
// events bracket the launch on its stream.  Off by default: no events, no cost.
struct KRecord { const char* name; double bytes; hipEvent_t e0, e1; };
struct KSink {
  std::vector<KRecord> rec;
  std::vector<hipEvent_t> pool;
};
extern thread_local KSink* tl_ksink;
struct KTimer {
  hipStream_t s; KSink* k; KRecord r;
  KTimer(hipStream_t st, const char* name, double bytes) : s(st), k(tl_ksink) {
    if (!k) return;
    r.name = name; r.bytes = bytes; r.e0 = r.e1 = nullptr;
    for (hipEvent_t* e : {&r.e0, &r.e1}) {
      if (!k->pool.empty()) { *e = k->pool.back(); k->pool.pop_back(); }
      else if (hipEventCreate(e) != hipSuccess) { (void)hipGetLastError(); *e = nullptr; }
    }
    if (r.e0 && r.e1) (void)hipEventRecord(r.e0, s); else k = nullptr;
  }
  ~KTimer() {
    if (!k) return;
    (void)hipEventRecord(r.e1, s);
    k->rec.push_back(r);
  }
  KTimer(const KTimer&) = delete;
  KTimer& operator=(const KTimer&) = delete;
};
}  // namespace vdf

namespace vdf { const vdf_hip_tuning& tuning(); }     // process-wide tuning (abi.hip), environment overrides applied once

// A caller context's stream and the two streams opened right behind it (abi.hip: the hardware maps queues to its pipes in the
// order they were created, and a prover whose three queues are neighbours runs 10-14 % faster than one whose queues are not)
struct vdf_queue_family {
  int device = 0;
  static constexpr int N = 3;
  hipStream_t s[N] = {};
  bool used[N] = {};
  ~vdf_queue_family();
};

struct vdf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int pool_slot = -1;                // >= 0: the stream belongs to the device's pool of hardware queues (vdf_ctx_create_pooled)
  std::shared_ptr<vdf_queue_family> family;   // the stream is family->s[family_idx] (vdf_ctx_create: 0; vdf_ctx_create_pooled_near: 1, 2)
  int family_idx = -1;
  bool foreign_stream = false;       // vdf_ctx_set_stream put a caller's stream in place of the context's own
  bool async = false;
  int msm_window = 0;          // 0 = automatic
  std::mutex mu;
  std::string err;
  // MSM workspace (grown on demand, reused across calls)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  void* d_out = nullptr;       // 128 B result slot
  static constexpr size_t SMALL_POOL_BYTES = 16 * 1024;
  void* small_pool = nullptr;  // staging for tiny host operands (challenge scalars, result points)
  static constexpr size_t PINNED_OUT_BYTES = 256;
  void* h_out = nullptr;       // pinned host memory mapped into the device: result slot kernels write directly
  void* h_out_dev = nullptr;   // its device-side address
  int num_cus = 256;
  // stage timing (bench.py roofline leg)
  bool timing = false;
  bool ktiming = false;              // per-launch events (vdf_ctx_set_kernel_timing)
  vdf::KSink ksink;
  struct TimedCall { hipEvent_t ev[4]; };
  std::vector<TimedCall> timed;      // events of calls not yet queried
  std::vector<hipEvent_t> ev_pool;   // recycled events
  hipEvent_t wait_ev = nullptr;      // vdf_ctx_wait
  hipEvent_t marks[VDF_MARK_SLOTS] = {};                          // vdf_ctx_mark (0..3 the caller's, 4..15 libvdf_nova.so's)
  // MSM jobs (vdf_msm_job_*): one side stream and two events per vector, created on first use
  hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t side_go[4] = {nullptr, nullptr, nullptr, nullptr}, side_done[4] = {nullptr, nullptr, nullptr, nullptr};
  bool job_open = false;
  void* reduce_scratch = nullptr;    // per-workgroup partial sums of vdf_reduce
  int acc_fill = 0;                  // accumulate workgroups per CU this context's MSMs fill (vdf_ctx_set_accumulate_fill; 0 = process-wide)
  int light_prio = 3;                // wave priority of this context's sort / bucket-reduction kernels (vdf_ctx_set_light_priority)
  hipEvent_t acc_gate = nullptr;     // one-shot: the next bucket-method MSM's accumulation waits for this event (vdf_ctx_gate_accumulate)
  void* glv_scalars = nullptr;       // 2n half-scalars of the endomorphism's split (table-less MSMs over a whole generator set)
  size_t glv_bytes = 0;
  void* glv_pts = nullptr;           // [P | phi(P)] of an EPHEMERAL generator set (the uncached shims): scratch, reused call after call
  size_t glv_pts_bytes = 0;
  uint32_t* direct_arrived = nullptr;  // MSM_MAX_GROUPS counters of the direct sum's last-arriver step (zero between calls)
};

struct vdf_bases {
  vdf_ctx* ctx = nullptr;
  int curve = 0;
  size_t n = 0;
  void* d_pts = nullptr;       // n affine points, 64 B each
  void* d_pts2 = nullptr;      // [P | phi(P)], 2n points: made by the first table-less MSM over the whole set (abi.hip msm_core)
  std::mutex glv_mu;
  bool ephemeral = false;      // made for one call (abi.hip shim): nothing derived from it is worth keeping
  // fixed-base table: tables x n affine points; table j holds 2^(c*sets*j) * P_i
  int tbl_c = 0, tbl_sets = 0, tbl_tables = 0;
  void* d_table = nullptr;
  // digit table (vdf_bases_precompute_digits, msm_direct.hip): every multiple d * 2^(c j) of the generators in up to 4
  // index ranges; range r holds slots [dg_slot0[r], dg_slot0[r] + dg_count[r])
  int dg_c = 0, dg_ranges = 0;
  size_t dg_begin[4] = {0, 0, 0, 0}, dg_count[4] = {0, 0, 0, 0}, dg_slot0[4] = {0, 0, 0, 0};
  void* d_digits = nullptr;
  size_t dg_bytes = 0;
};

struct vdf_shape {
  vdf_ctx* ctx = nullptr;
  int field = 0;
  size_t num_cons = 0, num_cols = 0;
  size_t nnz[3] = {0, 0, 0};
  uint32_t* d_rowptr[3] = {nullptr, nullptr, nullptr};   // num_cons + 1
  uint32_t* d_col[3] = {nullptr, nullptr, nullptr};      // nnz
  uint32_t* d_coef[3] = {nullptr, nullptr, nullptr};     // nnz, index into dictionary
  void* d_dict = nullptr;                                 // dictionary of field elements
  size_t dict_len = 0;
  // the three matrices merged in column-major order (vdf_spmv3_t): entry = (row, coefficient index | matrix << 30)
  uint32_t* d_t_colptr = nullptr;                         // num_cols + 1
  uint32_t* d_t_row = nullptr;                            // nnz[0] + nnz[1] + nnz[2]
  uint32_t* d_t_cm = nullptr;
  uint32_t* d_t_heavy = nullptr;                          // columns with more than 64 entries
  size_t t_nheavy = 0, t_nbig = 0;                        // ... sorted longest first; the first t_nbig have more than 4096
  // rows with more than VDF_LONG_ROW entries, as row | matrix << 30: one wavefront each (vec_spmv_long) instead of one lane
  uint32_t* d_long = nullptr;
  size_t n_long = 0;
  // the distinct rows among them, ascending: one wavefront each in the fused cross term (k_nifs_cross_f)
  uint32_t* d_long_rowlist = nullptr;
  size_t n_long_rowlist = 0;
  std::vector<uint64_t> h_nnz_prefix;                     // entries of A + B + C in the rows above r (pricing of a row range)
  std::vector<uint32_t> h_long_rows;                      // ... their row numbers, ascending (vdf_nifs_cross_term_rows)
};
// A row of more than this many entries is summed by a whole wavefront ahead of the lane-per-row kernels, which then
// only read the result: the rows of an augmented circuit that pack 255 bits or carry a Poseidon state of ~60 terms would
// otherwise serialise a launch behind one lane (1 ms instead of 20 us at t = 2^16).
#define VDF_LONG_ROW 8

namespace vdf {

struct Status {
  int code = VDF_OK;
  std::string msg;
  bool ok() const { return code == VDF_OK; }
};

inline Status hip_status(hipError_t e, const char* what) {
  Status s;
  if (e != hipSuccess) {
    s.code = (e == hipErrorOutOfMemory) ? VDF_ERR_OOM : VDF_ERR_DEVICE;
    s.msg = std::string(what) + ": " + hipGetErrorString(e);
  }
  return s;
}

#define VDF_TRY_HIP(expr)                                   \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) return ::vdf::hip_status(e__, #expr); \
  } while (0)
#define VDF_TRY(expr)                 \
  do {                                \
    ::vdf::Status s__ = (expr);       \
    if (!s__.ok()) return s__;        \
  } while (0)

// ---- msm.hip --------------------------------------------------------------------------
constexpr int MSM_MAX_GROUPS = 4;
// A plan covers a BATCH of 1..4 independent MSMs ("groups") over the same generator table: each group has its
// own scalar vector, length and generator offset and yields its own point.  The groups share every launch
// (one sort, one accumulate grid balanced over all entries, one latency-bound tail); to everything after pass A
// a group is simply `sets` more bucket sets.
struct MsmPlan {
  int groups = 1;
  uint32_t gn[MSM_MAX_GROUPS] = {0, 0, 0, 0};       // points per group
  uint32_t goff[MSM_MAX_GROUPS] = {0, 0, 0, 0};     // generator offset per group (points)
  uint32_t gblk_end[MSM_MAX_GROUPS] = {0, 0, 0, 0}; // pass-A workgroups: group g owns [gblk_end[g-1], gblk_end[g])
  uint32_t n = 0;        // points, all groups
  int c = 0;             // window bits
  int windows = 0;       // ceil(256 / c)  (ceil(132 / c) for the endomorphism's half-scalars)
  bool signed_scalars = false;   // the scalars are sign-and-magnitude words (k_glv_split)
  int sets = 0;          // bucket sets per group (Horner length); windows = sets * tables
  int gsets = 0;         // groups * sets
  int tables = 0;
  uint32_t nbk = 0;      // buckets per set = 2^(c-1)   (digit magnitudes 1..2^(c-1))
  // two-level counting sort: a bucket index b = (p << fb) | f; pass A partitions by p, pass B sorts by f
  int pb = 0, fb = 0;    // partition bits / fine bits, pb + fb = c - 1
  uint32_t bins = 0;     // gsets << pb
  uint32_t chA = 0;      // points per pass-A workgroup
  uint32_t nblkA = 0;    // pass-A workgroups
  uint32_t L = 0;        // sorted entries per accumulate thread: upper bound (the kernels use the actual entry count)
  uint32_t Lfixed = 0;   // tuning override: a fixed slice length (0 = computed on the device)
  uint32_t slots = 0;    // resident k_accumulate thread slots (workgroups per CU x 256 x CUs)
  uint32_t nthreads = 0; // accumulate threads
  uint32_t tstride = 0;  // points per fixed-base table (tables > 1)
  size_t ws_bytes = 0;
};
MsmPlan msm_make_plan(int groups, const size_t* n, const size_t* offsets, int c, int sets, int tables, int num_cus, int acc_fill = 0,
                      int scalar_bits = 256);
Status glv_points(int curve, const void* d_pts, size_t n, void* d_out, hipStream_t stream);      // [P | phi(P)]: 2n points
Status glv_split(int curve, const void* d_scalars, size_t n, bool is_mont, void* d_out, hipStream_t stream);   // [k1 | k2]: 2n sign-and-magnitude words
inline MsmPlan msm_make_plan(size_t n, int c, int sets, int tables, int num_cus) {
  const size_t zero = 0;
  return msm_make_plan(1, &n, &zero, c, sets, tables, num_cus);
}
int msm_auto_window(size_t n);
bool msm_plan_feasible(int groups, int c, int sets);       // sets <= 0: one bucket set per window (no table)
// d_points: table (tables*tstride affine) or plain bases (tables == 1).  d_scalars[g]: gn[g] x 32 B device.
// d_out: groups x 96 B device (Jacobian).  Enqueues on `stream`; no synchronisation.
// ev (optional): 4 events recorded at start / before accumulate / after accumulate / end.
// ext_bucket_acc (optional): the bucket accumulators of this plan's key range inside a job's shared array; the
// run then stops after the fix-up and msm_tail reduces all groups of the job at once.
Status msm_run(int curve, const MsmPlan& plan, const void* d_points, const void* const* d_scalars, bool is_mont,
               void* ws, void* d_out, hipStream_t stream, hipEvent_t* ev = nullptr, void* ext_bucket_acc = nullptr,
               hipEvent_t acc_gate = nullptr, int prio = 3);
size_t msm_tail_ws_bytes(int groups, int sets, uint32_t nbk);
Status msm_tail(int curve, int c, int sets, int groups, uint32_t nbk, void* tail_ws, void* d_out, hipStream_t stream, int prio = 3);
Status bases_generate(int curve, int family, uint64_t seed, size_t start, size_t n, void* d_pts, hipStream_t stream);
// msm_direct.hip: the digit table of generators [first, first + nslots) into slots [slot0, ...) of d_digits, and the
// direct sum over it (groups x 96 B Jacobian to d_out; ws: direct_ws_bytes)
int direct_windows(int c);
Status digits_build(int curve, const void* d_pts, size_t first, size_t nslots, size_t slot0, int c, void* d_digits, hipStream_t stream);
size_t direct_ws_bytes(int groups, const size_t* n, int c, int num_cus);
Status msm_direct_run(int curve, int groups, const size_t* n, const size_t* slot0, const void* const* d_scalars, bool is_mont,
                      int c, int num_cus, const void* d_digits, void* ws, void* d_out, uint32_t* arrived, hipStream_t stream);
Status bases_generate_label(int curve, const uint8_t* label, size_t len, size_t start, size_t n, void* d_pts, hipStream_t stream);
Status point_sum(int curve, const void* d_jac, size_t n, void* d_out, hipStream_t stream);
Status bases_validate(int curve, const void* d_pts, size_t n, uint32_t* d_flags, hipStream_t stream);   // d_flags: 2 words
Status bases_precompute(int curve, const void* d_pts, size_t n, int c, int sets, int tables, void* d_table,
                        hipStream_t stream);

// ---- vecops.hip ------------------------------------------------------------------------
Status vec_axpy(int field, const void* a, const void* r, const void* b, size_t n, void* out, hipStream_t s);
Status vec_cross_term(int field, const void* az1, const void* bz1, const void* cz1, const void* az2,
                      const void* bz2, const void* cz2, const void* u1, size_t n, void* T, hipStream_t s);
Status vec_minroot_witness(int field, const void* trace_xy, const void* i0, uint64_t t, void* W, hipStream_t s);
Status vec_spmv_long(int field, const uint32_t* const rowptr[3], const uint32_t* const col[3], const uint32_t* const coef[3],
                     const void* dict, const void* z, const uint32_t* long_rows, size_t n_long, void* const out[3], hipStream_t s);
Status vec_spmv(int field, const uint32_t* rowptr, const uint32_t* col, const uint32_t* coef, const void* dict,
                const void* z, size_t rows, void* out, hipStream_t s);
// fused step kernels; vdf_fe* arguments are HOST pointers whose values travel as kernel arguments
Status vec_step_segment(int field, const void* trace_xy, uint64_t t, const vdf_fe* i0, int per, void* out, void* packed,
                        const vdf_fe* i_in, hipStream_t s);
Status vec_step_z(int field, const void* trace_xy, uint64_t t, const vdf_fe z_in[3], const vdf_fe* i0, const vdf_fe* u,
                  const vdf_fe X[6], void* z, void* packed, hipStream_t s);
Status vec_nifs_cross(int field, const uint32_t* const rowptr[3], const uint32_t* const col[3],
                      const uint32_t* const coef[3], const void* dict, const void* z2, const void* az1, const void* bz1,
                      const void* cz1, const vdf_fe* u1, size_t rows, size_t skip_begin, size_t skip_len,
                      const uint32_t* long_rowlist, size_t n_long_rows, void* az2, void* bz2, void* cz2, void* T,
                      double alg_bytes, hipStream_t s);
int nifs_cross_lanes(size_t rows);                       // lanes per row vec_nifs_cross picks for a launch over `rows` rows
Status vec_nifs_cross_minroot(int field, int per, uint64_t t, size_t seg_begin, size_t one_col, size_t row0, const void* z2,
                              const void* az1, const void* bz1, const void* cz1, const vdf_fe* u1, void* az2, void* bz2, void* cz2,
                              void* T, hipStream_t s);
Status vec_nifs_cross_minroot_fold(int field, int per, uint64_t t, size_t seg_begin, size_t one_col, size_t row0, const void* z2,
                                   const vdf_fe* r, void* az1, void* bz1, void* cz1, void* e1, const void* tprev, const vdf_fe* u1,
                                   void* az2, void* bz2, void* cz2, void* T, hipStream_t s);
Status vec_fold_many(int field, const vdf_fe* r, int k, void* const acc[], const void* const add[], const size_t n[],
                     hipStream_t s);
Status vec_mul(int field, const void* a, const void* b, size_t n, void* out, hipStream_t s);
Status vec_any_nonzero(const void* v, size_t n, uint32_t* d_flag, hipStream_t s);     // *d_flag = 1 iff some element is non-zero
Status vec_to_mont(int field, const void* a, size_t n, void* out, hipStream_t s);
Status vec_from_mont(int field, const void* a, size_t n, void* out, hipStream_t s);
Status vec_mul_chain(int field, const void* a, size_t n, int iters, void* out, hipStream_t s);
Status vec_clock_probe(int iters, int workgroups, unsigned long long* d_out3, hipStream_t s);   // d_out3 zeroed by the caller

// ---- snark.hip -------------------------------------------------------------------------
// vdf_fe* arguments are HOST pointers whose values travel as kernel arguments; void* are device vectors
Status snark_pair_table(int field, const vdf_fe* lo, const vdf_fe* hi, int k, void* out, hipStream_t s);
Status snark_pair_table_pattern(int field, const vdf_fe* lo, const vdf_fe* hi, int k, const vdf_fe* pattern, int log_m, void* out,
                                hipStream_t s);
Status snark_fold_halves(int field, int k, void* const v[], const vdf_fe c_lo[], const vdf_fe c_hi[], size_t n, hipStream_t s);
size_t snark_reduce_scratch_bytes();
Status snark_reduce(int field, int kind, const void* const tables[], const vdf_fe* u, size_t n, void* scratch, void* out,
                    hipStream_t s);
Status snark_spmvt(int field, const uint32_t* colptr, const uint32_t* rows, const uint32_t* cm, const uint32_t* heavy,
                   size_t nheavy, size_t nbig, const void* dict, const void* eq, const vdf_fe* rho, size_t ncols, void* out, void* scratch,
                   hipStream_t s);
Status snark_ipa_scalars(int field, const void* a, const void* sv, size_t n, size_t nj, void* sL, void* sR, hipStream_t s);
Status snark_scale_pattern(int field, void* sv, size_t n, size_t nj, const vdf_fe* x_lo, const vdf_fe* x_hi, hipStream_t s);

}  // namespace vdf
