// extern "C" boundary of libvdf_hip.so (include/vdf_hip.h).  Plain pointers and sizes only; no
// exception leaves this file.  There is no CPU back-end: without a GPU vdf_ctx_create fails.
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <map>
#include <array>
#include <new>
#include <memory>
#include <atomic>
#include <deque>
#include <mutex>
#include <initializer_list>
#include <utility>
#include "internal.h"
#include "fe.cuh"

using vdf::Status;

namespace {

thread_local std::string g_create_err;
std::mutex g_default_mu;
vdf_ctx* g_default_ctx = nullptr;

// pinned, device-mapped host allocations handed out by vdf_host_alloc: kernels use them in place
std::mutex g_host_mu;
std::vector<std::pair<const char*, size_t>> g_host_allocs;
bool ptr_is_mapped_host(const void* p) {
  std::lock_guard<std::mutex> lock(g_host_mu);
  for (auto& a : g_host_allocs)
    if ((const char*)p >= a.first && (const char*)p < a.first + a.second) return true;
  return false;
}

bool ptr_is_device(const void* p) {
  if (!p) return false;
  if (ptr_is_mapped_host(p)) return true;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Host buffers are staged through temporaries; device buffers are used in place.
struct Staging {
  vdf_ctx* ctx;
  std::vector<void*> temps;
  struct Out { void* host; void* dev; size_t bytes; };
  std::vector<Out> outs;
  Out pinned_out{nullptr, nullptr, 0};
  bool any_host = false;
  size_t small_used = 0;     // bytes taken from the context's small staging pool (calls hold ctx->mu)
  explicit Staging(vdf_ctx* c) : ctx(c) {}
  ~Staging() { for (void* t : temps) (void)hipFree(t); }
  Status temp(size_t bytes, void** t) {
    const size_t need = (bytes + 255) / 256 * 256;
    if (need <= 1024 && ctx->small_pool && small_used + need <= vdf_ctx::SMALL_POOL_BYTES) {
      *t = reinterpret_cast<char*>(ctx->small_pool) + small_used;
      small_used += need;
      return Status{};
    }
    VDF_TRY_HIP(hipMalloc(t, bytes));
    temps.push_back(*t);
    return Status{};
  }
  Status in(const void* p, size_t bytes, const void** dev) {
    if (bytes == 0) { *dev = p; return Status{}; }
    if (!p) return Status{VDF_ERR_BAD_ARG, "null input pointer"};
    if (ptr_is_device(p)) { *dev = p; return Status{}; }
    any_host = true;
    void* t = nullptr;
    VDF_TRY(temp(bytes, &t));
    VDF_TRY_HIP(hipMemcpyAsync(t, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = t;
    return Status{};
  }
  Status out(void* p, size_t bytes, void** dev) {
    if (bytes == 0) { *dev = p; return Status{}; }
    if (!p) return Status{VDF_ERR_BAD_ARG, "null output pointer"};
    if (ptr_is_device(p)) { *dev = p; return Status{}; }
    any_host = true;
    // tiny results (an MSM's point): the kernel stores into pinned, device-mapped host memory, so the call ends
    // with one stream synchronisation and a host-side copy instead of a device-to-host transfer of its own
    if (bytes <= vdf_ctx::PINNED_OUT_BYTES && ctx->h_out && !pinned_out.host) {
      pinned_out = {p, ctx->h_out, bytes};
      *dev = ctx->h_out_dev;
      return Status{};
    }
    void* t = nullptr;
    VDF_TRY(temp(bytes, &t));
    outs.push_back({p, t, bytes});
    *dev = t;
    return Status{};
  }
  Status finish() {
    for (auto& o : outs) VDF_TRY_HIP(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (any_host || !ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    if (pinned_out.host) std::memcpy(pinned_out.host, pinned_out.dev, pinned_out.bytes);
    return Status{};
  }
};

int fail(vdf_ctx* ctx, const Status& s) {
  if (ctx) ctx->err = s.msg; else g_create_err = s.msg;
  return s.code;
}


// ---- process-wide tuning (include/vdf_hip.h vdf_hip_tuning) ---------------------------------------------------------
// Readers (every launch, on any thread) take the current snapshot through one atomic pointer; a set publishes a NEW immutable
// snapshot and never touches an old one (they are kept: a few dozen bytes per call), so a launch racing with
// vdf_hip_tuning_set sees either the old values or the new ones, never a mixture, and holds no lock.
std::once_flag g_tune_once;
std::mutex g_tune_mu;                                   // writers only
std::deque<vdf_hip_tuning> g_tune_snapshots;            // (a deque: push_back moves no element)
std::atomic<const vdf_hip_tuning*> g_tune{nullptr};
vdf_hip_tuning tuning_defaults() {
  vdf_hip_tuning t{};
  t.struct_size = (uint32_t)sizeof(vdf_hip_tuning);
  t.msm_direct = 1; t.direct_priority = 2; t.direct_fused = 1; t.light_priority = 3; t.accumulate_fill = 0;
  t.accumulate_lds = 0; t.slice_len = 0; t.part_bits = -1; t.reduction = 1; t.reduction_quads = 0; t.heavy_min = 0;
  t.giant_span = 0; t.nifs_lanes = 0; t.shim_cache = 0; t.nifs_fused = 1; t.fold_u128 = 1;
  t.fixup_serial = 1; t.sort_staged = 1; t.glv = 1;
  return t;
}
bool tuning_valid(const vdf_hip_tuning& t) {
  auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  return in(t.msm_direct, 0, 1) && in(t.direct_priority, 0, 3) && in(t.direct_fused, 0, 1) && in(t.light_priority, 0, 3) &&
         in(t.accumulate_fill, 0, 3) && in(t.accumulate_lds, 0, 65536) && in(t.slice_len, 0, 65536) && in(t.part_bits, -1, 19) &&
         in(t.reduction, 0, 1) && (t.reduction_quads == 0 || in(t.reduction_quads, 64, 65536)) &&
         (t.heavy_min == 0 || in(t.heavy_min, 1, 4096)) && (t.giant_span == 0 || in(t.giant_span, 16, 1 << 20)) &&
         (t.nifs_lanes == 0 || t.nifs_lanes == 1 || t.nifs_lanes == 4 || t.nifs_lanes == 8) && in(t.shim_cache, 0, 64) &&
         in(t.nifs_fused, 0, 1) && in(t.fold_u128, 0, 1) && in(t.fixup_serial, 0, 1) && in(t.sort_staged, 0, 1) && in(t.glv, 0, 1);
}
void tuning_publish(const vdf_hip_tuning& t) {           // caller holds g_tune_mu (or is the once-initialiser)
  g_tune_snapshots.push_back(t);
  g_tune_snapshots.back().struct_size = (uint32_t)sizeof(vdf_hip_tuning);
  g_tune.store(&g_tune_snapshots.back(), std::memory_order_release);
}
// the environment overrides of earlier rounds, read once: the only place this library looks at the environment for tuning
void tuning_from_env() {
  vdf_hip_tuning t = tuning_defaults();
  const struct { const char* name; int32_t* field; } vars[] = {
      {"VDF_MSM_DIRECT", &t.msm_direct}, {"VDF_MSM_DIRECT_PRIO", &t.direct_priority}, {"VDF_MSM_DIRECT_FUSED", &t.direct_fused},
      {"VDF_MSM_LIGHT_PRIO", &t.light_priority}, {"VDF_MSM_ACC_WG", &t.accumulate_fill}, {"VDF_MSM_ACC_LDS", &t.accumulate_lds},
      {"VDF_MSM_L", &t.slice_len}, {"VDF_MSM_PB", &t.part_bits}, {"VDF_MSM_RED", &t.reduction},
      {"VDF_MSM_RED_QUADS", &t.reduction_quads}, {"VDF_MSM_HEAVY_MIN", &t.heavy_min}, {"VDF_MSM_GIANT_SPAN", &t.giant_span},
      {"VDF_NIFS_LANES", &t.nifs_lanes}, {"VDF_SHIM_CACHE", &t.shim_cache}, {"VDF_NIFS_FUSED", &t.nifs_fused}, {"VDF_FOLD_U128", &t.fold_u128},
      {"VDF_MSM_FIXUP_SERIAL", &t.fixup_serial}, {"VDF_MSM_SORT_STAGED", &t.sort_staged}, {"VDF_MSM_GLV", &t.glv}};
  for (const auto& v : vars) {
    const char* e = std::getenv(v.name);
    if (!e || !*e) continue;
    const int32_t old = *v.field;
    *v.field = (int32_t)std::atol(e);
    if (!tuning_valid(t)) *v.field = old;               // an out-of-range override is ignored, as before
  }
  std::lock_guard<std::mutex> lock(g_tune_mu);
  tuning_publish(t);
}
}  // namespace
namespace vdf {
thread_local KSink* tl_ksink = nullptr;
const vdf_hip_tuning& tuning() {
  std::call_once(g_tune_once, tuning_from_env);
  return *g_tune.load(std::memory_order_acquire);
}
}  // namespace vdf
extern "C" int vdf_hip_tuning_get(vdf_hip_tuning* out) {
  if (!out) return VDF_ERR_BAD_ARG;
  *out = vdf::tuning();
  return VDF_OK;
}
extern "C" int vdf_hip_tuning_set(const vdf_hip_tuning* in) {
  if (!in || in->struct_size != sizeof(vdf_hip_tuning) || !tuning_valid(*in)) return VDF_ERR_BAD_ARG;
  (void)vdf::tuning();
  std::lock_guard<std::mutex> lock(g_tune_mu);
  tuning_publish(*in);
  return VDF_OK;
}
namespace {
struct SinkScope {            // launches made under this call record into the context's sink (nested calls restore the outer one)
  vdf::KSink* prev;
  explicit SinkScope(vdf_ctx* c) : prev(vdf::tl_ksink) { vdf::tl_ksink = c->ktiming ? &c->ksink : nullptr; }
  ~SinkScope() { vdf::tl_ksink = prev; }
};

template <class F>
int guarded(vdf_ctx* ctx, F&& body) {
  if (!ctx) { g_create_err = "null context"; return VDF_ERR_BAD_ARG; }
  try {
    std::lock_guard<std::mutex> lock(ctx->mu);
    SinkScope sink_scope(ctx);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, vdf::hip_status(e, "hipSetDevice"));
    Status s = body();
    if (!s.ok()) return fail(ctx, s);
    return VDF_OK;
  } catch (const std::bad_alloc&) {
    ctx->err = "host allocation failed";
    return VDF_ERR_OOM;
  } catch (const std::exception& ex) {
    ctx->err = ex.what();
    return VDF_ERR_DEVICE;
  } catch (...) {
    ctx->err = "unknown failure";
    return VDF_ERR_DEVICE;
  }
}

Status ensure_ws(vdf_ctx* ctx, size_t bytes) {
  if (ctx->ws_bytes >= bytes) return Status{};
  if (ctx->job_open) return Status{VDF_ERR_BAD_ARG, "the MSM workspace cannot grow while a job is open on this context"};
  if (ctx->ws) { VDF_TRY_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
  VDF_TRY_HIP(hipMalloc(&ctx->ws, bytes));
  ctx->ws_bytes = bytes;
  return Status{};
}

// the direct sum is for commitments a host waits on; beyond this many scalars per call the bucket method's throughput wins
constexpr size_t DIRECT_MAX_SCALARS = (size_t)1 << 17;
bool direct_enabled() { return vdf::tuning().msm_direct != 0; }       // (tuning: 0 = bucket method only)

size_t field_of_curve_scalar(int curve) { return curve == VDF_CURVE_PALLAS ? VDF_FIELD_FQ : VDF_FIELD_FP; }

constexpr size_t GLV_MIN_POINTS = 1u << 12;      // below this the split's launches cost what the shorter chain saves
#ifndef VDF_GLV_MAX_LOG2
#define VDF_GLV_MAX_LOG2 21          /* (an A/B build may move it: make ab AB_FLAGS=-DVDF_GLV_MAX_LOG2=23) */
#endif
constexpr size_t GLV_MAX_POINTS = (size_t)1 << VDF_GLV_MAX_LOG2;      // measured (profiles/r05_glv_tableless.txt): 2^14 1.25 -> 0.75 ms, 2^18 1.81 -> 1.29, 2^20 2.86 -> 2.61;
                                                 // at 2^22 the doubled point array slows the gathers by what the chain saves (7.11 -> 7.22)
Status msm_core(vdf_ctx* ctx, const vdf_bases* bases, int groups, const size_t* offset, const vdf_fe* const* scalars,
                const size_t* n, int is_mont, vdf_jac* out) {
  if (!bases || !out || !offset || !scalars || !n) return Status{VDF_ERR_BAD_ARG, "null bases/out/arrays"};
  if (groups < 1 || groups > vdf::MSM_MAX_GROUPS) return Status{VDF_ERR_BAD_ARG, "1..4 MSMs per batch"};
  if (ctx->job_open) return Status{VDF_ERR_BAD_ARG, "an MSM job is open on this context: its workspace is in use"};
  // generators may be shared by several contexts of one device (e.g. a second stream for overlap)
  if (bases->ctx != ctx && bases->ctx->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "bases live on another device"};
  size_t ntot = 0, nmax = 0;
  for (int g = 0; g < groups; ++g) {
    if (offset[g] > bases->n || n[g] > bases->n - offset[g]) return Status{VDF_ERR_BAD_LENGTH, "offset + n exceeds the generator table"};
    ntot += n[g];
    if (n[g] > nmax) nmax = n[g];
  }
  if (ntot >= (1ull << 27)) return Status{VDF_ERR_BAD_LENGTH, "n too large (max 2^27 - 1 points per call)"};
  Staging st(ctx);
  void* d_out = nullptr;
  VDF_TRY(st.out(out, groups * sizeof(vdf_jac), &d_out));
  if (ntot == 0) {
    VDF_TRY_HIP(hipMemsetAsync(d_out, 0, groups * sizeof(vdf_jac), ctx->stream));
    return st.finish();
  }
  const void* d_scalars[vdf::MSM_MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
  for (int g = 0; g < groups; ++g) VDF_TRY(st.in(scalars[g], n[g] * sizeof(vdf_fe), &d_scalars[g]));
  // small commitments over generators that have a digit table: a plain sum of gathered multiples (msm_direct.hip)
  if (bases->d_digits && ctx->msm_window == 0 && ntot <= DIRECT_MAX_SCALARS && direct_enabled()) {
    size_t slot0[vdf::MSM_MAX_GROUPS] = {0, 0, 0, 0};
    bool inside = true;
    for (int g = 0; g < groups && inside; ++g) {
      int r = 0;
      while (r < bases->dg_ranges && !(offset[g] >= bases->dg_begin[r] && offset[g] + n[g] <= bases->dg_begin[r] + bases->dg_count[r])) ++r;
      if (r == bases->dg_ranges) inside = n[g] == 0;
      else slot0[g] = bases->dg_slot0[r] + (offset[g] - bases->dg_begin[r]);
    }
    if (inside) {
      VDF_TRY(ensure_ws(ctx, vdf::direct_ws_bytes(groups, n, bases->dg_c, ctx->num_cus)));
      hipEvent_t* dev = nullptr;
      vdf_ctx::TimedCall dtc;
      int have = 0;                                       // events taken so far: they go back to the pool if the call fails
      struct Giveback { vdf_ctx* c; vdf_ctx::TimedCall* t; int* have; bool armed; ~Giveback() { if (armed) for (int i = 0; i < *have; ++i) c->ev_pool.push_back(t->ev[i]); } }
          giveback{ctx, &dtc, &have, true};
      if (ctx->timing) {
        for (int i = 0; i < 4; ++i) {
          if (!ctx->ev_pool.empty()) { dtc.ev[i] = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
          else VDF_TRY_HIP(hipEventCreate(&dtc.ev[i]));
          have = i + 1;
        }
        dev = dtc.ev;
        VDF_TRY_HIP(hipEventRecord(dev[0], ctx->stream));
        VDF_TRY_HIP(hipEventRecord(dev[1], ctx->stream));
      }
      if (!ctx->direct_arrived) {
        VDF_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->direct_arrived), 64));
        VDF_TRY_HIP(hipMemsetAsync(ctx->direct_arrived, 0, 64, ctx->stream));
      }
      VDF_TRY(vdf::msm_direct_run(bases->curve, groups, n, slot0, d_scalars, is_mont != 0, bases->dg_c, ctx->num_cus, bases->d_digits, ctx->ws, d_out,
                                  ctx->direct_arrived, ctx->stream));
      if (dev) {
        VDF_TRY_HIP(hipEventRecord(dev[2], ctx->stream));
        VDF_TRY_HIP(hipEventRecord(dev[3], ctx->stream));
        ctx->timed.push_back(dtc);
      }
      giveback.armed = false;
      return st.finish();
    }
  }
  vdf::MsmPlan plan;
  const char* pts;
  if (bases->d_table && (ctx->msm_window == 0 || ctx->msm_window == bases->tbl_c)) {
    if (!vdf::msm_plan_feasible(groups, bases->tbl_c, bases->tbl_sets))
      return Status{VDF_ERR_BAD_ARG, "a batch this wide does not fit the sort under this table's window and bucket sets: fewer MSMs per call"};
    plan = vdf::msm_make_plan(groups, n, offset, bases->tbl_c, bases->tbl_sets, bases->tbl_tables, ctx->num_cus, ctx->acc_fill);
    plan.tstride = (uint32_t)bases->n;
    pts = reinterpret_cast<const char*>(bases->d_table);
  } else if (groups == 1 && offset[0] == 0 && n[0] == bases->n && n[0] >= GLV_MIN_POINTS && n[0] <= GLV_MAX_POINTS && ctx->msm_window == 0 &&
             vdf::tuning().glv) {
    // No table, the whole generator set: the endomorphism (msm.hip k_glv_*).  2n points [P | phi(P)] kept with the generators
    // (made on first use), 2n half-scalars of 129 bits + sign in this context's scratch: 9 bucket sets instead of 16 and a
    // Horner chain of 128 doublings instead of 240 behind the same number of bucket additions.
    vdf_bases* mb = const_cast<vdf_bases*>(bases);
    const void* pts2 = nullptr;
    if (bases->ephemeral) {
      // a generator set made for ONE call (the uncached shims: upload, MSM, free): [P | phi(P)] goes into this context's scratch,
      // made on the call's own stream -- no allocation, no synchronisation, nothing to free with the set (an allocation and a
      // hipFree more per shim call showed as occasional 25 ms stalls of later calls: the driver's deferred unmapping)
      const size_t pbytes = 2 * bases->n * sizeof(vdf_affine);
      if (ctx->glv_pts_bytes < pbytes) {
        VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->glv_pts) (void)hipFree(ctx->glv_pts);
        ctx->glv_pts = nullptr; ctx->glv_pts_bytes = 0;
        VDF_TRY_HIP(hipMalloc(&ctx->glv_pts, pbytes));
        ctx->glv_pts_bytes = pbytes;
      }
      VDF_TRY(vdf::glv_points(bases->curve, bases->d_pts, bases->n, ctx->glv_pts, ctx->stream));
      pts2 = ctx->glv_pts;
    } else {
      std::lock_guard<std::mutex> lock(mb->glv_mu);
      if (!mb->d_pts2) {
        void* p2 = nullptr;
        VDF_TRY_HIP(hipMalloc(&p2, 2 * bases->n * sizeof(vdf_affine)));
        Status sg = vdf::glv_points(bases->curve, bases->d_pts, bases->n, p2, ctx->stream);
        if (sg.code == VDF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) sg = Status{VDF_ERR_DEVICE, "glv points"};
        if (sg.code != VDF_OK) { (void)hipFree(p2); return sg; }
        mb->d_pts2 = p2;
      }
      pts2 = mb->d_pts2;
    }
    const size_t need = 2 * n[0] * sizeof(vdf_fe);
    if (ctx->glv_bytes < need) {
      VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->glv_scalars) (void)hipFree(ctx->glv_scalars);
      ctx->glv_scalars = nullptr; ctx->glv_bytes = 0;
      VDF_TRY_HIP(hipMalloc(&ctx->glv_scalars, need));
      ctx->glv_bytes = need;
    }
    VDF_TRY(vdf::glv_split(bases->curve, d_scalars[0], n[0], is_mont != 0, ctx->glv_scalars, ctx->stream));
    d_scalars[0] = ctx->glv_scalars;
    is_mont = 0;
    const size_t n2[1] = {2 * n[0]}, off2[1] = {0};
    // the window: with the rounded split the half-scalars are below 2^127, so 16-bit windows use eight FULL windows (the top
    // digit has 15 significant bits and never carries); a ninth bucket set exists for scalars that hit the rounding's last unit
    // and is empty otherwise.  Windows that leave the top digit a few bits wide (17: 10 of 17; 15: 9 of 15) put an eighth of all
    // entries into one sort partition (measured: the sort 0.4 -> 2.2 ms at 2^20).  Smaller sets: 13 (ten windows, 10 of 13 on top).
    const int ca = vdf::msm_auto_window(n2[0]);
    const int c = ca >= 15 ? 16 : 13;
    plan = vdf::msm_make_plan(1, n2, off2, c, 0, 0, ctx->num_cus, ctx->acc_fill, 132);
    plan.tstride = 0;
    pts = reinterpret_cast<const char*>(pts2);
    ntot = n2[0];
  } else {
    int c = ctx->msm_window ? ctx->msm_window : vdf::msm_auto_window(nmax);
    while (c > 4 && !vdf::msm_plan_feasible(groups, c, 0)) --c;     // a table-less window is a tuning knob: lowered to what fits
    plan = vdf::msm_make_plan(groups, n, offset, c, 0, 0, ctx->num_cus, ctx->acc_fill);
    plan.tstride = 0;
    pts = reinterpret_cast<const char*>(bases->d_pts);
  }
  if ((size_t)plan.tstride * plan.tables + bases->n >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "table index exceeds 31 bits"};
  if ((uint64_t)ntot * plan.windows >= 0xFFF00000ull) return Status{VDF_ERR_BAD_LENGTH, "n * windows exceeds 32-bit entry positions"};
  VDF_TRY(ensure_ws(ctx, plan.ws_bytes));
  hipEvent_t* ev = nullptr;
  vdf_ctx::TimedCall tc;
  if (ctx->timing) {
    for (int i = 0; i < 4; ++i) {
      if (!ctx->ev_pool.empty()) { tc.ev[i] = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
      else VDF_TRY_HIP(hipEventCreate(&tc.ev[i]));
    }
    ev = tc.ev;
  }
  hipEvent_t gate = ctx->acc_gate;
  ctx->acc_gate = nullptr;                                      // one-shot
  VDF_TRY(vdf::msm_run(bases->curve, plan, pts, d_scalars, is_mont != 0, ctx->ws, d_out, ctx->stream, ev, nullptr, gate, std::min(ctx->light_prio, (int)vdf::tuning().light_priority)));
  if (ev) ctx->timed.push_back(tc);
  return st.finish();
}

vdf_ctx* default_ctx() {
  std::lock_guard<std::mutex> lock(g_default_mu);
  if (!g_default_ctx) {
    int dev = 0;
    if (vdf_ctx_create(&dev, 1, &g_default_ctx) != VDF_OK) g_default_ctx = nullptr;
  }
  return g_default_ctx;
}

// Generator cache of the shims (vdf_shim_set_cache).  nova-snark commits under the same CommitGens for the life of the
// process; the upstream signature gives no handle to keep them resident, so the shim recognises a generator array by
// (curve, address, length) and a content hash of all its points, keeps it in HBM, and from the second call on with
// its fixed-base table.  The address is a key only: it is dereferenced in the call that passes it, never later.
struct ShimEntry { int curve; const void* ptr; size_t n; uint64_t fp; vdf_bases* bases; uint64_t last_use; bool table; };
std::mutex g_shim_mu;
std::vector<ShimEntry> g_shim;
uint64_t g_shim_clock = 0;

// Content hash of the WHOLE array (every coordinate word), so a generator rewritten in place anywhere is a different
// set: chunks hashed on up to 8 threads (a 2^19-point array is 32 MiB; one pass costs ~0.3 ms that way), combined in
// chunk order.  64-bit multiply-xorshift per word; a collision needs a deliberate adversary, which a process's own
// generator arrays are not.
uint64_t shim_fingerprint(const vdf_affine* points, size_t n) {
  auto hash_range = [points](size_t lo, size_t hi) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)lo;
    const uint64_t* w = reinterpret_cast<const uint64_t*>(points + lo);
    for (size_t k = 0, e = (hi - lo) * 8; k < e; ++k) { h = (h ^ w[k]) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    return h;
  };
  const size_t nth = n >= (1u << 16) ? 8 : 1, per = (n + nth - 1) / nth;
  std::vector<uint64_t> part(nth, 0);
  std::vector<std::thread> th;
  for (size_t t = 1; t < nth; ++t)
    th.emplace_back([&, t] { const size_t lo = t * per, hi = lo + per < n ? lo + per : n; if (lo < hi) part[t] = hash_range(lo, hi); });
  part[0] = hash_range(0, per < n ? per : n);
  for (auto& t : th) t.join();
  uint64_t h = 0x100000001b3ull ^ (uint64_t)n;
  for (uint64_t v : part) { h = (h ^ v) * 0x100000001b3ull; h ^= h >> 31; }
  return h;
}

// The upstream signature returns nothing, and an all-zero `out` is the identity -- a valid-looking commitment.  A failed
// shim call therefore must not return: it says why on stderr and aborts (a caller that wants a status uses vdf_msm).
[[noreturn]] void shim_die(const char* what, const std::string& why) {
  std::fprintf(stderr, "libvdf_hip: mult_pippenger shim failed (%s): %s\n", what, why.c_str());
  std::abort();
}

int shim_cache_capacity() {          // caller holds g_shim_mu
  // ONE knob: vdf_hip_tuning.shim_cache, read on every call (vdf_hip_tuning_set takes effect on the next call, as the header
  // says: nothing is latched); vdf_shim_set_cache publishes a tuning snapshot with the new value, so tuning_get reports it
  return (int)vdf::tuning().shim_cache;
}

void shim(int curve, vdf_jac* out, const vdf_affine* points, size_t n, const vdf_fe* scalars, bool is_mont) {
  if (!out) shim_die("arguments", "null output pointer");
  std::memset(out, 0, sizeof(*out));
  vdf_ctx* ctx = default_ctx();
  if (!ctx) shim_die("default context", g_create_err);
  // With the cache on, a call holds the cache lock from lookup to the end of its MSM: an entry must not be evicted (and
  // its generators freed) by another thread while this one computes with it.  Nothing is lost -- the shims share one
  // default context, whose calls are serialised anyway.
  std::unique_lock<std::mutex> lock(g_shim_mu);
  const int cap = shim_cache_capacity();
  if (cap <= 0 || !points || !n) {
    lock.unlock();
    vdf_bases* b = nullptr;
    if (vdf_bases_upload(ctx, curve, points, n, &b) != VDF_OK) shim_die("generator upload", ctx->err);
    b->ephemeral = true;                               // one call's generators: scratch, not a second allocation, for [P | phi(P)]
    if (vdf_msm(ctx, b, 0, scalars, n, is_mont ? 1 : 0, out) != VDF_OK) shim_die("MSM", ctx->err);
    vdf_bases_free(b);
    return;
  }
  const uint64_t fp = shim_fingerprint(points, n);
  ShimEntry* hit = nullptr;
  for (ShimEntry& e : g_shim)
    if (e.curve == curve && e.ptr == (const void*)points && e.n == n && e.fp == fp) { hit = &e; break; }
  if (!hit) {
    // a stale entry for this address (same array, new contents) goes; then the least recently used
    for (size_t i = 0; i < g_shim.size();)
      if (g_shim[i].curve == curve && g_shim[i].ptr == (const void*)points && g_shim[i].n == n) { vdf_bases_free(g_shim[i].bases); g_shim.erase(g_shim.begin() + i); }
      else ++i;
    while ((int)g_shim.size() >= cap) {
      size_t lru = 0;
      for (size_t i = 1; i < g_shim.size(); ++i) if (g_shim[i].last_use < g_shim[lru].last_use) lru = i;
      vdf_bases_free(g_shim[lru].bases);
      g_shim.erase(g_shim.begin() + lru);
    }
    vdf_bases* b = nullptr;
    if (vdf_bases_upload(ctx, curve, points, n, &b) != VDF_OK) shim_die("generator upload", ctx->err);
    g_shim.push_back(ShimEntry{curve, (const void*)points, n, fp, b, 0, false});
    hit = &g_shim.back();
  } else if (!hit->table) {
    if (n >= 1024) (void)vdf_bases_precompute(ctx, hit->bases, 16, 1);   // the second call for a set pays for its table;
    hit->table = true;                                                   // failure (memory) leaves the plain path
  }
  hit->last_use = ++g_shim_clock;
  if (vdf_msm(ctx, hit->bases, 0, scalars, n, is_mont ? 1 : 0, out) != VDF_OK) shim_die("MSM", ctx->err);
}

template <class P>
void build_dict_consts(std::array<uint32_t, 8>& one, std::array<uint32_t, 8>& minus_one) {
  vdf::Fe<P> o = vdf::fe_one<P>();
  vdf::Fe<P> m = vdf::fe_neg(o);
  for (int i = 0; i < 8; ++i) { one[i] = o.v[i]; minus_one[i] = m.v[i]; }
}

}  // namespace

// ---- the process's budget of hardware queues (include/vdf_hip.h vdf_ctx_create_pooled) -----------------------------------------
// The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (16 asked for below; 4 by default); past that, streams share a queue and a kernel
// waits behind another stream's (two provers + a compression opened 11 streams: the two-chain rate fell below one chain's,
// profiles/r04_box_spread.txt).  Contexts a library makes for its own queues take their stream from this per-device pool:
// a new stream while the device's total (pooled + caller-owned contexts) is below the budget, else they SHARE the least used
// pooled stream of their role (a side queue -- a lookahead, a compression's second opening -- with another side queue, before
// a latency-critical one).  Sharing only serialises; every cross-context order in this library is an event recorded before it
// is waited for, so two contexts on one stream cannot deadlock.
namespace {
struct PoolStream { hipStream_t s = nullptr; int users = 0; int role = 0; };
struct DevicePool { std::vector<PoolStream> streams; int owned = 0; };
std::mutex g_pool_mu;
std::map<int, DevicePool> g_pool;
int hw_queue_budget() {
  static const int b = [] { const char* e = std::getenv("GPU_MAX_HW_QUEUES"); const int v = e ? std::atoi(e) : 4; return v < 2 ? 2 : (v > 64 ? 64 : v); }();
  return b;
}
// a stream for a pooled context of `role` on `device`; *slot = its index in the pool
hipError_t pool_take(int device, int role, hipStream_t* out, int* slot) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  DevicePool& dp = g_pool[device];
  int live = dp.owned;
  for (const PoolStream& ps : dp.streams) if (ps.users) ++live;
  // one queue is left to the host's own streams (torch's, a caller's copies) while pooled contexts can still share
  if (live < hw_queue_budget() - 1) {
    int idx = -1;
    for (size_t i = 0; i < dp.streams.size(); ++i) if (!dp.streams[i].users) { idx = (int)i; break; }
    if (idx < 0) { dp.streams.emplace_back(); idx = (int)dp.streams.size() - 1; }
    PoolStream& ps = dp.streams[idx];
    if (!ps.s) { const hipError_t e = hipStreamCreateWithFlags(&ps.s, hipStreamNonBlocking); if (e != hipSuccess) return e; }
    ps.users = 1; ps.role = role;
    *out = ps.s; *slot = idx;
    return hipSuccess;
  }
  int best = -1;
  for (int pass = 0; pass < 2 && best < 0; ++pass)                      // same role first (side with side), then any pooled stream
    for (size_t i = 0; i < dp.streams.size(); ++i) {
      const PoolStream& ps = dp.streams[i];
      if (!ps.users || (pass == 0 && ps.role != role)) continue;
      if (best < 0 || ps.users < dp.streams[best].users) best = (int)i;
    }
  if (best < 0) {                                                        // nothing pooled yet (the callers hold the whole budget): one stream over it
    dp.streams.emplace_back();
    best = (int)dp.streams.size() - 1;
    const hipError_t e = hipStreamCreateWithFlags(&dp.streams[best].s, hipStreamNonBlocking);
    if (e != hipSuccess) { dp.streams.pop_back(); return e; }
    dp.streams[best].role = role;
  }
  dp.streams[best].users += 1;
  *out = dp.streams[best].s; *slot = best;
  return hipSuccess;
}
void pool_release(int device, int slot) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  DevicePool& dp = g_pool[device];
  if (slot < 0 || slot >= (int)dp.streams.size() || dp.streams[slot].users <= 0) return;
  if (--dp.streams[slot].users == 0 && dp.streams[slot].s) { (void)hipStreamDestroy(dp.streams[slot].s); dp.streams[slot].s = nullptr; }
}
void pool_count_owned(int device, int delta) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  g_pool[device].owned += delta;
}
}  // namespace
vdf_queue_family::~vdf_queue_family() {
  int n = 0;
  for (hipStream_t& q : s) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); q = nullptr; ++n; }
  if (n) pool_count_owned(device, -n);
}

extern "C" {

int vdf_ctx_device(vdf_ctx* ctx) { return ctx ? ctx->device : 0; }

const char* vdf_version(void) { return "vdf_hip gfx950 r1 (" __DATE__ ")"; }

static int ctx_create_impl(const int* device_ids, int n_devices, int role, vdf_ctx** out);
int vdf_ctx_create_pooled_near(vdf_ctx* parent, int role, vdf_ctx** out);
int vdf_ctx_create(const int* device_ids, int n_devices, vdf_ctx** out) { return ctx_create_impl(device_ids, n_devices, 0, out); }
int vdf_ctx_create_pooled(const int* device_ids, int n_devices, int role, vdf_ctx** out) {
  if (role != VDF_QUEUE_CRITICAL && role != VDF_QUEUE_SIDE) { g_create_err = "vdf_ctx_create_pooled: unknown role"; return VDF_ERR_BAD_ARG; }
  return ctx_create_impl(device_ids, n_devices, role, out);
}
int vdf_ctx_create_pooled_near(vdf_ctx* parent, int role, vdf_ctx** out) {
  if (!parent || !out) { g_create_err = "vdf_ctx_create_pooled_near: null argument"; return VDF_ERR_BAD_ARG; }
  if (role != VDF_QUEUE_CRITICAL && role != VDF_QUEUE_SIDE) { g_create_err = "vdf_ctx_create_pooled_near: unknown role"; return VDF_ERR_BAD_ARG; }
  const int dev = parent->device;
  std::shared_ptr<vdf_queue_family> fam;
  int idx = -1;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    const int want = role == VDF_QUEUE_SIDE ? 1 : 2;   // the neighbour right behind the parent's stream, and the one after it (other
                                                       // assignments and larger families measured no better: profiles/r05_family_slots.txt)
    if (parent->family && !parent->foreign_stream && parent->family->s[want] && !parent->family->used[want]) { fam = parent->family; idx = want; fam->used[idx] = true; }
  }
  if (idx < 0) return vdf_ctx_create_pooled(&dev, 1, role, out);    // taken (a second prover on this context) or no family: the pool
  const int rc = ctx_create_impl(&dev, 1, -1, out);                  // role -1: no stream of its own
  if (rc != VDF_OK) { std::lock_guard<std::mutex> lock(g_pool_mu); fam->used[idx] = false; return rc; }
  (*out)->family = fam; (*out)->family_idx = idx; (*out)->stream = fam->s[idx];
  return VDF_OK;
}
int vdf_ctx_queue_info(vdf_ctx* ctx, int* pooled, int* sharers, int* device_streams) {
  if (!ctx) return VDF_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  DevicePool& dp = g_pool[ctx->device];
  if (pooled) *pooled = (ctx->pool_slot >= 0 || ctx->family_idx > 0) ? 1 : 0;
  if (sharers) *sharers = ctx->pool_slot >= 0 ? dp.streams[ctx->pool_slot].users : 1;
  if (device_streams) { int live = dp.owned; for (const PoolStream& ps : dp.streams) if (ps.users) ++live; *device_streams = live; }
  return VDF_OK;
}
static int ctx_create_impl(const int* device_ids, int n_devices, int role, vdf_ctx** out) {
  if (!out) { g_create_err = "null out"; return VDF_ERR_BAD_ARG; }
  *out = nullptr;
  if (n_devices != 1 || !device_ids) {
    g_create_err = "vdf_ctx_create: exactly one device per context (one process per GPU); there is no CPU back-end";
    return n_devices == 0 ? VDF_ERR_NO_DEVICE : VDF_ERR_BAD_ARG;
  }
  // a prover keeps several queues busy at once (vdf_nova.h): the runtime's default of 4 hardware queues makes streams share
  // one, and kernels then wait behind another stream's; honoured only when this is the process's first HIP call
  // -- and set at most once per process (setenv is not safe against concurrent readers of the environment: a second prover thread making its
  // contexts, the runtime's own threads).  A host that makes its first HIP call elsewhere sets the variable itself.
  static std::once_flag hwq_once;
  std::call_once(hwq_once, [] { setenv("GPU_MAX_HW_QUEUES", "16", 0); });
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    g_create_err = "vdf_ctx_create: no HIP device visible (the HIP path is mandatory; no CPU fallback exists)";
    return VDF_ERR_NO_DEVICE;
  }
  if (device_ids[0] < 0 || device_ids[0] >= count) { g_create_err = "device id out of range"; return VDF_ERR_BAD_ARG; }
  vdf_ctx* c = new (std::nothrow) vdf_ctx();
  if (!c) { g_create_err = "host allocation failed"; return VDF_ERR_OOM; }
  c->device = device_ids[0];
  e = hipSetDevice(c->device);
  if (e == hipSuccess) {
    if (role == 0) {
      // the context's stream and two more right behind it: neighbours in the hardware's queue order, kept for the queues a
      // prover opens beside this one (vdf_ctx_create_pooled_near; profiles/r05_single_chain_vs_padding.txt)
      auto fam = std::make_shared<vdf_queue_family>();
      fam->device = c->device;
      int made = 0;
      for (int k = 0; k < vdf_queue_family::N && e == hipSuccess; ++k) { e = hipStreamCreateWithFlags(&fam->s[k], hipStreamNonBlocking); if (e == hipSuccess) ++made; }
      pool_count_owned(c->device, made);
      if (e == hipSuccess) { fam->used[0] = true; c->family = fam; c->family_idx = 0; c->stream = fam->s[0]; }
    } else if (role > 0) e = pool_take(c->device, role, &c->stream, &c->pool_slot);
  }
  if (e == hipSuccess) e = hipMalloc(&c->d_out, 256);
  if (e == hipSuccess) e = hipMalloc(&c->small_pool, vdf_ctx::SMALL_POOL_BYTES);
  if (e == hipSuccess) {
    // optional fast path; without it small results take the ordinary staged copy
    if (hipHostMalloc(&c->h_out, vdf_ctx::PINNED_OUT_BYTES, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&c->h_out_dev, c->h_out, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (c->h_out) (void)hipHostFree(c->h_out);
      c->h_out = c->h_out_dev = nullptr;
    }
  }
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, c->device);
  if (e != hipSuccess) {
    g_create_err = std::string("vdf_ctx_create: ") + hipGetErrorString(e);
    if (c->pool_slot >= 0) pool_release(c->device, c->pool_slot);
    c->family.reset();
    delete c;
    return VDF_ERR_DEVICE;
  }
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  *out = c;
  return VDF_OK;
}

void vdf_ctx_destroy(vdf_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->small_pool) (void)hipFree(ctx->small_pool);
  if (ctx->h_out) (void)hipHostFree(ctx->h_out);
  if (ctx->reduce_scratch) (void)hipFree(ctx->reduce_scratch);
  if (ctx->direct_arrived) (void)hipFree(ctx->direct_arrived);
  if (ctx->glv_scalars) (void)hipFree(ctx->glv_scalars);
  if (ctx->glv_pts) (void)hipFree(ctx->glv_pts);
  for (auto& tc : ctx->timed) for (int i = 0; i < 4; ++i) (void)hipEventDestroy(tc.ev[i]);
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
  for (hipEvent_t& m : ctx->marks) if (m) (void)hipEventDestroy(m);
  for (int g = 0; g < 4; ++g) {
    if (ctx->side_go[g]) (void)hipEventDestroy(ctx->side_go[g]);
    if (ctx->side_done[g]) (void)hipEventDestroy(ctx->side_done[g]);
    if (ctx->side[g]) { (void)hipStreamSynchronize(ctx->side[g]); (void)hipStreamDestroy(ctx->side[g]); }
  }
  if (ctx->own_stream && ctx->stream) { (void)hipStreamDestroy(ctx->stream); pool_count_owned(ctx->device, -1); }
  if (ctx->pool_slot >= 0) pool_release(ctx->device, ctx->pool_slot);
  if (ctx->family) { std::lock_guard<std::mutex> lock(g_pool_mu); ctx->family->used[ctx->family_idx] = false; }
  ctx->family.reset();                               // the family's streams go with its last member
  delete ctx;
}

int vdf_ctx_set_stream(vdf_ctx* ctx, void* hip_stream) {
  return guarded(ctx, [&]() -> Status {
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    if (!hip_stream) {                                   // back to the context's own stream (a foreign one was set before)
      if (ctx->family && ctx->family_idx >= 0) { ctx->stream = ctx->family->s[ctx->family_idx]; ctx->foreign_stream = false; return Status{}; }
      if (ctx->foreign_stream) return Status{VDF_ERR_BAD_ARG, "this context has no stream of its own to return to"};
      return Status{};
    }
    if (ctx->own_stream && ctx->stream) { (void)hipStreamDestroy(ctx->stream); pool_count_owned(ctx->device, -1); }
    if (ctx->pool_slot >= 0) { pool_release(ctx->device, ctx->pool_slot); ctx->pool_slot = -1; }
    // (a family member keeps its place: vdf_ctx_set_stream(ctx, NULL) returns to it; its neighbours are not handed out meanwhile)
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    ctx->foreign_stream = true;
    return Status{};
  });
}

void* vdf_ctx_get_stream(vdf_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

int vdf_ctx_set_async(vdf_ctx* ctx, int async) {
  return guarded(ctx, [&]() -> Status { ctx->async = async != 0; return Status{}; });
}

int vdf_ctx_get_async(vdf_ctx* ctx, int* async) {
  return guarded(ctx, [&]() -> Status {
    if (!async) return Status{VDF_ERR_BAD_ARG, "null output"};
    *async = ctx->async ? 1 : 0;
    return Status{};
  });
}

int vdf_ctx_sync(vdf_ctx* ctx) {
  return guarded(ctx, [&]() -> Status {
    // poll first: a prover waits ~1 ms for a commitment, and the interrupt-driven wake-up of a blocking
    // synchronise costs 10-20 us of it; after a few milliseconds fall back to the blocking call
    for (int spin = 0; spin < 4000; ++spin) {
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipSuccess) return Status{};
      if (q != hipErrorNotReady) return vdf::hip_status(q, "hipStreamQuery");
    }
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_ctx_set_msm_window(vdf_ctx* ctx, int window_bits) {
  return guarded(ctx, [&]() -> Status {
    if (window_bits != 0 && (window_bits < 4 || window_bits > 20)) return Status{VDF_ERR_BAD_ARG, "window_bits must be 0 or 4..20"};
    ctx->msm_window = window_bits;
    return Status{};
  });
}

const char* vdf_last_error(vdf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// ---- bases --------------------------------------------------------------------------------
int vdf_bases_upload(vdf_ctx* ctx, int curve, const vdf_affine* bases, size_t n, vdf_bases** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    *out = nullptr;
    if (curve != VDF_CURVE_PALLAS && curve != VDF_CURVE_VESTA) return Status{VDF_ERR_BAD_ARG, "unknown curve"};
    if (n && !bases) return Status{VDF_ERR_BAD_ARG, "null bases"};
    vdf_bases* b = new vdf_bases();
    b->ctx = ctx; b->curve = curve; b->n = n;
    if (n) {
      hipError_t e = hipMalloc(&b->d_pts, n * sizeof(vdf_affine));
      if (e != hipSuccess) { delete b; return vdf::hip_status(e, "hipMalloc(bases)"); }
      e = hipMemcpyAsync(b->d_pts, bases, n * sizeof(vdf_affine), hipMemcpyDefault, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) { (void)hipFree(b->d_pts); delete b; return vdf::hip_status(e, "upload bases"); }
    }
    *out = b;
    return Status{};
  });
}

int vdf_bases_validate(vdf_ctx* ctx, const vdf_bases* bases, size_t* first_bad) {
  return guarded(ctx, [&]() -> Status {
    if (!bases || (bases->ctx != ctx && bases->ctx->device != ctx->device)) return Status{VDF_ERR_BAD_ARG, "bad bases handle"};
    if (first_bad) *first_bad = 0;
    if (bases->n == 0) return Status{};
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(ctx->small_pool);
    const uint32_t init[2] = {0u, 0xFFFFFFFFu};
    VDF_TRY_HIP(hipMemcpyAsync(d_flags, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    VDF_TRY(vdf::bases_validate(bases->curve, bases->d_pts, bases->n, d_flags, ctx->stream));
    uint32_t res[2];
    VDF_TRY_HIP(hipMemcpyAsync(res, d_flags, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    if (res[0] == 0) return Status{};
    if (first_bad) *first_bad = res[1];
    if (res[0] & 1u) return Status{VDF_ERR_NONCANONICAL, "a generator coordinate is not a canonical field element"};
    return Status{VDF_ERR_BAD_ARG, "a generator is neither the identity nor a point of the curve"};
  });
}

int vdf_bases_generate(vdf_ctx* ctx, int curve, uint64_t seed, size_t n, vdf_bases** out) {
  return vdf_bases_generate_range(ctx, curve, seed, 0, n, out);
}

int vdf_bases_generate_range(vdf_ctx* ctx, int curve, uint64_t seed, size_t start, size_t n, vdf_bases** out) {
  return vdf_bases_generate_family(ctx, curve, VDF_GENS_KNOWN_DLOG, seed, start, n, out);
}

int vdf_bases_generate_family(vdf_ctx* ctx, int curve, int family, uint64_t seed, size_t start, size_t n, vdf_bases** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    *out = nullptr;
    if (curve != VDF_CURVE_PALLAS && curve != VDF_CURVE_VESTA) return Status{VDF_ERR_BAD_ARG, "unknown curve"};
    if (n >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "too many bases"};
    vdf_bases* b = new vdf_bases();
    b->ctx = ctx; b->curve = curve; b->n = n;
    if (n) {
      hipError_t e = hipMalloc(&b->d_pts, n * sizeof(vdf_affine));
      if (e != hipSuccess) { delete b; return vdf::hip_status(e, "hipMalloc(bases)"); }
      Status s = vdf::bases_generate(curve, family, seed, start, n, b->d_pts, ctx->stream);
      if (s.ok()) s = vdf::hip_status(hipStreamSynchronize(ctx->stream), "bases_generate");
      if (!s.ok()) { (void)hipFree(b->d_pts); delete b; return s; }
    }
    *out = b;
    return Status{};
  });
}

int vdf_bases_generate_label(vdf_ctx* ctx, int curve, const uint8_t* label, size_t label_len, size_t start, size_t n, vdf_bases** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    *out = nullptr;
    if (curve != VDF_CURVE_PALLAS && curve != VDF_CURVE_VESTA) return Status{VDF_ERR_BAD_ARG, "unknown curve"};
    if (n >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "too many bases"};
    if (label_len > 64 || (label_len && !label)) return Status{VDF_ERR_BAD_LENGTH, "label of at most 64 bytes"};
    vdf_bases* b = new vdf_bases();
    b->ctx = ctx; b->curve = curve; b->n = n;
    if (n) {
      hipError_t e = hipMalloc(&b->d_pts, n * sizeof(vdf_affine));
      if (e != hipSuccess) { delete b; return vdf::hip_status(e, "hipMalloc(bases)"); }
      Status s = vdf::bases_generate_label(curve, label, label_len, start, n, b->d_pts, ctx->stream);
      if (s.ok()) s = vdf::hip_status(hipStreamSynchronize(ctx->stream), "bases_generate_label");
      if (!s.ok()) { (void)hipFree(b->d_pts); delete b; return s; }
    }
    *out = b;
    return Status{};
  });
}

int vdf_bases_precompute(vdf_ctx* ctx, vdf_bases* bases, int window_bits, int sets) {
  return guarded(ctx, [&]() -> Status {
    if (!bases || bases->ctx != ctx) return Status{VDF_ERR_BAD_ARG, "bad bases handle"};
    // measured (profiles/r05_window_sweep.txt): 16 below 2^19 generators, 17 up to 2^21, 20 from 2^22 on (13 windows
    // instead of 15: the accumulation's -13 % outweighs a bucket reduction over 2^19 buckets only there; 18 and 19 leave a top
    // window of 2 / 7 significant bits, i.e. a handful of hot buckets for uniform scalars)
    if (window_bits == 0) window_bits = bases->n >= ((size_t)1 << 22) ? 20 : (bases->n >= ((size_t)1 << 19) ? 17 : 16);
    if (window_bits < 4 || window_bits > 20) return Status{VDF_ERR_BAD_ARG, "window_bits must be 0 (recommended) or 4..20"};
    const int windows = (256 + window_bits - 1) / window_bits;
    if (sets <= 0) sets = 1;
    if (sets > windows) sets = windows;
    while (windows % sets) ++sets;                 // windows = sets * tables exactly
    const int tables = windows / sets;
    if (bases->d_table) { (void)hipFree(bases->d_table); bases->d_table = nullptr; }
    bases->tbl_c = bases->tbl_sets = bases->tbl_tables = 0;
    if (bases->n == 0 || tables == 1) return Status{};
    VDF_TRY_HIP(hipMalloc(&bases->d_table, (size_t)tables * bases->n * sizeof(vdf_affine)));
    Status s = vdf::bases_precompute(bases->curve, bases->d_pts, bases->n, window_bits, sets, tables, bases->d_table, ctx->stream);
    if (s.ok()) s = vdf::hip_status(hipStreamSynchronize(ctx->stream), "bases_precompute");
    if (!s.ok()) { (void)hipFree(bases->d_table); bases->d_table = nullptr; return s; }
    bases->tbl_c = window_bits; bases->tbl_sets = sets; bases->tbl_tables = tables;
    return Status{};
  });
}

int vdf_bases_precompute_digits(vdf_ctx* ctx, vdf_bases* bases, int window_bits, int ranges, const size_t begin[], const size_t count[]) {
  return guarded(ctx, [&]() -> Status {
    if (!bases || bases->ctx != ctx) return Status{VDF_ERR_BAD_ARG, "bad bases handle"};
    if (bases->d_digits) { (void)hipFree(bases->d_digits); bases->d_digits = nullptr; }
    bases->dg_c = bases->dg_ranges = 0;
    if (ranges == 0) return Status{};
    if (ranges < 0 || ranges > 4 || !begin || !count) return Status{VDF_ERR_BAD_ARG, "1..4 generator ranges"};
    if (window_bits == 0) window_bits = 10;
    if (window_bits < 4 || window_bits > 12) return Status{VDF_ERR_BAD_ARG, "digit window must be 0 (recommended) or 4..12"};
    size_t slots = 0;
    for (int r = 0; r < ranges; ++r) {
      if (begin[r] > bases->n || count[r] > bases->n - begin[r]) return Status{VDF_ERR_BAD_LENGTH, "range exceeds the generator table"};
      for (int q = 0; q < r; ++q)
        if (begin[r] < begin[q] + count[q] && begin[q] < begin[r] + count[r]) return Status{VDF_ERR_BAD_ARG, "generator ranges overlap"};
      slots += count[r];
    }
    if (slots == 0) return Status{};
    const int W = vdf::direct_windows(window_bits);
    if (slots * W >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "too many generators for a digit table"};
    const size_t bytes = ((slots * W) << (window_bits - 1)) * sizeof(vdf_affine);
    size_t free_b = 0, total_b = 0;
    VDF_TRY_HIP(hipMemGetInfo(&free_b, &total_b));
    if (bytes + ((size_t)1 << 30) > free_b) return Status{VDF_ERR_OOM, "the digit table does not fit the free device memory"};
    if (hipMalloc(&bases->d_digits, bytes) != hipSuccess) {
      (void)hipGetLastError();
      bases->d_digits = nullptr;
      return Status{VDF_ERR_OOM, "the digit table could not be allocated"};
    }
    size_t slot0 = 0;
    Status s{};
    for (int r = 0; r < ranges && s.ok(); ++r) {
      s = vdf::digits_build(bases->curve, bases->d_pts, begin[r], count[r], slot0, window_bits, bases->d_digits, ctx->stream);
      bases->dg_begin[r] = begin[r]; bases->dg_count[r] = count[r]; bases->dg_slot0[r] = slot0;
      slot0 += count[r];
    }
    if (!s.ok()) { (void)hipFree(bases->d_digits); bases->d_digits = nullptr; return s; }
    bases->dg_c = window_bits; bases->dg_ranges = ranges; bases->dg_bytes = bytes;
    return Status{};
  });
}

int vdf_bases_digit_window(const vdf_bases* bases) { return bases && bases->d_digits ? bases->dg_c : 0; }
size_t vdf_bases_digit_table_bytes(const vdf_bases* bases) { return bases && bases->d_digits ? bases->dg_bytes : 0; }

int vdf_bases_window(const vdf_bases* bases) { return bases && bases->d_table ? bases->tbl_c : 0; }
size_t vdf_bases_table_bytes(const vdf_bases* bases) { return bases && bases->d_table ? (size_t)bases->tbl_tables * bases->n * 64 : 0; }
size_t vdf_digit_table_bytes(int window_bits, size_t generators) {
  if (window_bits < 4 || window_bits > 12) return 0;
  return generators * (size_t)vdf::direct_windows(window_bits) * ((size_t)64 << (window_bits - 1));
}

int vdf_bases_download(vdf_ctx* ctx, const vdf_bases* bases, size_t offset, size_t n, vdf_affine* out) {
  return guarded(ctx, [&]() -> Status {
    if (!bases || bases->ctx != ctx || (n && !out)) return Status{VDF_ERR_BAD_ARG, "bad arguments"};
    if (offset > bases->n || n > bases->n - offset) return Status{VDF_ERR_BAD_LENGTH, "range exceeds the generator table"};
    if (n == 0) return Status{};
    VDF_TRY_HIP(hipMemcpyAsync(out, reinterpret_cast<const char*>(bases->d_pts) + offset * 64, n * 64, hipMemcpyDefault, ctx->stream));
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

size_t vdf_bases_len(const vdf_bases* bases) { return bases ? bases->n : 0; }
const void* vdf_bases_device_ptr(const vdf_bases* bases) { return bases ? bases->d_pts : nullptr; }

void vdf_bases_free(vdf_bases* bases) {
  if (!bases) return;
  if (bases->ctx) {
    std::lock_guard<std::mutex> lock(bases->ctx->mu);
    (void)hipSetDevice(bases->ctx->device);
    (void)hipStreamSynchronize(bases->ctx->stream);
    if (bases->d_pts) (void)hipFree(bases->d_pts);
    if (bases->d_table) (void)hipFree(bases->d_table);
    if (bases->d_digits) (void)hipFree(bases->d_digits);
    if (bases->d_pts2) (void)hipFree(bases->d_pts2);
  }
  delete bases;
}

// ---- MSM ------------------------------------------------------------------------------------
int vdf_msm(vdf_ctx* ctx, const vdf_bases* bases, size_t offset, const vdf_fe* scalars, size_t n, int is_mont, vdf_jac* out) {
  return guarded(ctx, [&]() -> Status { return msm_core(ctx, bases, 1, &offset, &scalars, &n, is_mont, out); });
}

int vdf_msm_batch(vdf_ctx* ctx, const vdf_bases* bases, int k, const size_t offset[], const vdf_fe* const scalars[],
                  const size_t n[], int is_mont, vdf_jac out[]) {
  return guarded(ctx, [&]() -> Status { return msm_core(ctx, bases, k, offset, scalars, n, is_mont, out); });
}

// ---- MSM jobs: vectors pushed one at a time, one shared tail ------------------------------------------------
struct vdf_msm_job {
  vdf_ctx* ctx = nullptr;
  const vdf_bases* bases = nullptr;
  int k = 0, is_mont = 0;
  vdf::MsmPlan plan[vdf::MSM_MAX_GROUPS];
  size_t ws_off[vdf::MSM_MAX_GROUPS] = {0, 0, 0, 0};
  size_t tail_off = 0;
  bool pushed[vdf::MSM_MAX_GROUPS] = {false, false, false, false};
  const char* pts = nullptr;
};

int vdf_msm_job_begin(vdf_ctx* ctx, const vdf_bases* bases, int k, const size_t offset[], const size_t n[], int is_mont,
                      vdf_msm_job** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    *out = nullptr;
    if (!bases || !offset || !n) return Status{VDF_ERR_BAD_ARG, "null argument"};
    if (k < 1 || k > vdf::MSM_MAX_GROUPS) return Status{VDF_ERR_BAD_ARG, "1..4 vectors per job"};
    if (bases->ctx != ctx && bases->ctx->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "bases live on another device"};
    if (ctx->job_open) return Status{VDF_ERR_BAD_ARG, "another MSM job is open on this context"};
    if (!bases->d_table || bases->tbl_sets != 1)
      return Status{VDF_ERR_BAD_ARG, "MSM jobs need a fixed-base table with one bucket set (vdf_bases_precompute(c, 1))"};
    std::unique_ptr<vdf_msm_job> job(new vdf_msm_job());
    job->ctx = ctx; job->bases = bases; job->k = k; job->is_mont = is_mont;
    job->pts = reinterpret_cast<const char*>(bases->d_table);
    size_t off = 0;
    for (int g = 0; g < k; ++g) {
      if (offset[g] > bases->n || n[g] > bases->n - offset[g]) return Status{VDF_ERR_BAD_LENGTH, "offset + n exceeds the generator table"};
      if (n[g] == 0 || n[g] >= (1ull << 27)) return Status{VDF_ERR_BAD_LENGTH, "every vector of a job needs 1 .. 2^27 - 1 elements"};
      job->plan[g] = vdf::msm_make_plan(1, &n[g], &offset[g], bases->tbl_c, 1, bases->tbl_tables, ctx->num_cus, ctx->acc_fill);
      job->plan[g].tstride = (uint32_t)bases->n;
      if ((uint64_t)n[g] * job->plan[g].windows >= 0xFFF00000ull) return Status{VDF_ERR_BAD_LENGTH, "n * windows exceeds 32-bit entry positions"};
      job->ws_off[g] = off;
      off += (job->plan[g].ws_bytes + 255) / 256 * 256;
    }
    if ((size_t)bases->n * (bases->tbl_tables + 1) >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "table index exceeds 31 bits"};
    job->tail_off = off;
    off += vdf::msm_tail_ws_bytes(k, 1, job->plan[0].nbk);
    VDF_TRY(ensure_ws(ctx, off));
    for (int g = 0; g < k; ++g) {
      if (!ctx->side[g]) VDF_TRY_HIP(hipStreamCreateWithFlags(&ctx->side[g], hipStreamNonBlocking));
      if (!ctx->side_go[g]) VDF_TRY_HIP(hipEventCreateWithFlags(&ctx->side_go[g], hipEventDisableTiming));
      if (!ctx->side_done[g]) VDF_TRY_HIP(hipEventCreateWithFlags(&ctx->side_done[g], hipEventDisableTiming));
    }
    ctx->job_open = true;
    *out = job.release();
    return Status{};
  });
}

int vdf_msm_job_push(vdf_msm_job* job, int g, const vdf_fe* scalars) {
  if (!job) return VDF_ERR_BAD_ARG;
  vdf_ctx* ctx = job->ctx;
  return guarded(ctx, [&]() -> Status {
    if (g < 0 || g >= job->k) return Status{VDF_ERR_BAD_ARG, "vector index out of range"};
    if (job->pushed[g]) return Status{VDF_ERR_BAD_ARG, "vector already pushed"};
    if (!ptr_is_device(scalars)) return Status{VDF_ERR_BAD_ARG, "job vectors live in device memory"};
    // the vector's pipeline starts after everything enqueued on the context so far (its producer), on a stream of
    // its own: the context's stream stays free for the caller's next kernels
    VDF_TRY_HIP(hipEventRecord(ctx->side_go[g], ctx->stream));
    VDF_TRY_HIP(hipStreamWaitEvent(ctx->side[g], ctx->side_go[g], 0));
    const void* sc[vdf::MSM_MAX_GROUPS] = {scalars, nullptr, nullptr, nullptr};
    char* ws = reinterpret_cast<char*>(ctx->ws);
    char* buckets = ws + job->tail_off + (size_t)g * job->plan[g].nbk * 128;
    VDF_TRY(vdf::msm_run(job->bases->curve, job->plan[g], job->pts, sc, job->is_mont != 0, ws + job->ws_off[g], nullptr,
                         ctx->side[g], nullptr, buckets));
    VDF_TRY_HIP(hipEventRecord(ctx->side_done[g], ctx->side[g]));
    job->pushed[g] = true;
    return Status{};
  });
}

int vdf_msm_job_finish(vdf_msm_job* job, vdf_jac out[]) {
  if (!job) return VDF_ERR_BAD_ARG;
  vdf_ctx* ctx = job->ctx;
  const int rc = guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    for (int g = 0; g < job->k; ++g) if (!job->pushed[g]) return Status{VDF_ERR_BAD_ARG, "not every vector was pushed"};
    Staging st(ctx);
    void* d_out = nullptr;
    VDF_TRY(st.out(out, job->k * sizeof(vdf_jac), &d_out));
    for (int g = 0; g < job->k; ++g) VDF_TRY_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_done[g], 0));
    VDF_TRY(vdf::msm_tail(job->bases->curve, job->plan[0].c, 1, job->k, job->plan[0].nbk,
                          reinterpret_cast<char*>(ctx->ws) + job->tail_off, d_out, ctx->stream));
    return st.finish();
  });
  // the job ends here whatever happened; an abandoned pipeline must not outlive the workspace it writes
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (rc != VDF_OK) for (int g = 0; g < job->k; ++g) if (ctx->side[g]) (void)hipStreamSynchronize(ctx->side[g]);
    ctx->job_open = false;
  }
  delete job;
  return rc;
}

int vdf_point_sum(vdf_ctx* ctx, int curve, const vdf_jac* points, size_t n, vdf_jac* out) {
  return guarded(ctx, [&]() -> Status {
    if (n >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "too many points"};
    Staging st(ctx);
    const void* dp; void* dout;
    VDF_TRY(st.in(points, n * sizeof(vdf_jac), &dp));
    VDF_TRY(st.out(out, sizeof(vdf_jac), &dout));
    VDF_TRY(vdf::point_sum(curve, dp, n, dout, ctx->stream));
    return st.finish();
  });
}

// ---- point-chunk-sharded MSM across GPUs -------------------------------------------------------------
int vdf_msm_sharded(vdf_ctx* ctx, const vdf_bases* shard_bases, size_t offset, const vdf_fe* shard_scalars, size_t n, int is_mont,
                    int rank, int world, vdf_allgather_fn gather, void* user, int flags, vdf_jac* partial, vdf_jac* gathered,
                    vdf_jac* out) {
  if (!ctx) return VDF_ERR_BAD_ARG;
  if (world < 1 || rank < 0 || rank >= world) { ctx->err = "rank / world out of range"; return VDF_ERR_BAD_ARG; }
  const bool always = (flags & VDF_SHARDED_ALWAYS_GATHER) != 0;
  if (world == 1 && !always)                               // one rank: its partial is the result
    return vdf_msm(ctx, shard_bases, offset, shard_scalars, n, is_mont, out);
  if (!gather || !partial || !gathered) { ctx->err = "sharded MSM needs a collective and its two buffers"; return VDF_ERR_BAD_ARG; }
  if (!ptr_is_device(partial) || !ptr_is_device(gathered)) {
    ctx->err = "the collective's buffers live in device memory";
    return VDF_ERR_BAD_ARG;
  }
  int rc = vdf_msm(ctx, shard_bases, offset, shard_scalars, n, is_mont, partial);
  if (rc != VDF_OK) return rc;
  // the host's collective, ordered on this context's stream after the partial; no lock held: it may block on other ranks
  if (gather(user, partial, gathered, sizeof(vdf_jac), ctx->stream) != 0) {
    ctx->err = "the all-gather supplied by the host failed";
    return VDF_ERR_DEVICE;
  }
  return vdf_point_sum(ctx, shard_bases->curve, gathered, (size_t)world, out);
}

// One process driving several GPUs: a context, a generator shard and a scalar slice per device; the partials land in
// pinned host memory and are summed on the first context's device.
int vdf_msm_multi(vdf_ctx* const ctxs[], const vdf_bases* const bases[], const size_t offsets[], const vdf_fe* const scalars[],
                  const size_t n[], int k, int is_mont, vdf_jac* out) {
  if (!ctxs || !bases || !scalars || !n || !out || k < 1 || k > 64 || !ctxs[0]) return VDF_ERR_BAD_ARG;
  vdf_ctx* c0 = ctxs[0];
  vdf_jac* h = nullptr;
  int rc = vdf_host_alloc(c0, (size_t)k * sizeof(vdf_jac), reinterpret_cast<void**>(&h));
  if (rc != VDF_OK) return rc;
  int was_async[64] = {};
  bool switched[64] = {};                        // only contexts whose mode was read AND changed are restored below
  for (int i = 0; i < k && rc == VDF_OK; ++i) {
    if (!ctxs[i] || !bases[i] || bases[i]->curve != bases[0]->curve) { c0->err = "bad context / generator shard"; rc = VDF_ERR_BAD_ARG; break; }
    rc = vdf_ctx_get_async(ctxs[i], &was_async[i]);
    if (rc == VDF_OK) { rc = vdf_ctx_set_async(ctxs[i], 1); switched[i] = rc == VDF_OK; }
    // every device works at once: enqueue all partials (results into pinned memory), then wait for each
    if (rc == VDF_OK) rc = vdf_msm(ctxs[i], bases[i], offsets ? offsets[i] : 0, scalars[i], n[i], is_mont, &h[i]);
    if (rc != VDF_OK && ctxs[i] != c0) c0->err = ctxs[i]->err;
  }
  for (int i = 0; i < k; ++i) {
    if (!ctxs[i] || !switched[i]) continue;
    const int r2 = vdf_ctx_sync(ctxs[i]);
    if (rc == VDF_OK && r2 != VDF_OK) { rc = r2; c0->err = ctxs[i]->err; }
    (void)vdf_ctx_set_async(ctxs[i], was_async[i]);
  }
  if (rc == VDF_OK) rc = vdf_point_sum(c0, bases[0]->curve, h, (size_t)k, out);
  if (rc == VDF_OK) rc = vdf_ctx_sync(c0);
  (void)vdf_host_free(c0, h);
  return rc;
}

// ---- per-launch timing (the roofline report of bench.py) ----------------------------------------------------------
namespace {
std::mutex g_base_mu;
hipEvent_t g_base_ev[64] = {};          // one time origin per device, shared by every context on it
}
int vdf_ctx_set_kernel_timing(vdf_ctx* ctx, int enable) {
  return guarded(ctx, [&]() -> Status {
    if (enable && ctx->device >= 0 && ctx->device < 64) {
      std::lock_guard<std::mutex> l(g_base_mu);
      if (!g_base_ev[ctx->device]) {
        hipEvent_t e = nullptr;
        VDF_TRY_HIP(hipEventCreate(&e));
        VDF_TRY_HIP(hipEventRecord(e, ctx->stream));
        VDF_TRY_HIP(hipEventSynchronize(e));
        g_base_ev[ctx->device] = e;
      }
    }
    ctx->ktiming = enable != 0;
    return Status{};
  });
}
int vdf_ctx_kernel_events(vdf_ctx* ctx, vdf_kernel_event* out, size_t cap, size_t* n) {
  return guarded(ctx, [&]() -> Status {
    if (!n) return Status{VDF_ERR_BAD_ARG, "null count"};
    auto& rec = ctx->ksink.rec;
    *n = rec.size();
    if (!out) return Status{};                                  // a query of the count only: nothing is drained
    if (cap < rec.size()) return Status{VDF_ERR_BAD_LENGTH, "more launches recorded than the buffer holds"};
    hipEvent_t base = (ctx->device >= 0 && ctx->device < 64) ? g_base_ev[ctx->device] : nullptr;
    if (!base && !rec.empty()) return Status{VDF_ERR_BAD_ARG, "kernel timing was never enabled on this device"};
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < rec.size(); ++i) {
      float a = 0, b = 0;
      // a record's stream may be another than the context's (the side streams of an MSM job): wait for its end event
      VDF_TRY_HIP(hipEventSynchronize(rec[i].e1));
      VDF_TRY_HIP(hipEventElapsedTime(&a, base, rec[i].e0));
      VDF_TRY_HIP(hipEventElapsedTime(&b, base, rec[i].e1));
      std::memset(out[i].name, 0, sizeof(out[i].name));
      std::strncpy(out[i].name, rec[i].name, sizeof(out[i].name) - 1);
      out[i].bytes = rec[i].bytes; out[i].start_ms = a; out[i].end_ms = b;
      ctx->ksink.pool.push_back(rec[i].e0); ctx->ksink.pool.push_back(rec[i].e1);
    }
    rec.clear();
    return Status{};
  });
}

int vdf_ctx_set_timing(vdf_ctx* ctx, int enable) {
  return guarded(ctx, [&]() -> Status { ctx->timing = enable != 0; return Status{}; });
}

int vdf_msm_timing(vdf_ctx* ctx, float ms[4], int* calls) {
  return guarded(ctx, [&]() -> Status {
    if (!ms || !calls) return Status{VDF_ERR_BAD_ARG, "null output"};
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) ms[i] = 0.f;
    *calls = (int)ctx->timed.size();
    for (auto& tc : ctx->timed) {
      float a = 0, b = 0, c = 0, d = 0;
      VDF_TRY_HIP(hipEventElapsedTime(&a, tc.ev[0], tc.ev[1]));
      VDF_TRY_HIP(hipEventElapsedTime(&b, tc.ev[1], tc.ev[2]));
      VDF_TRY_HIP(hipEventElapsedTime(&c, tc.ev[2], tc.ev[3]));
      VDF_TRY_HIP(hipEventElapsedTime(&d, tc.ev[0], tc.ev[3]));
      ms[0] += a; ms[1] += b; ms[2] += c; ms[3] += d;
      for (int i = 0; i < 4; ++i) ctx->ev_pool.push_back(tc.ev[i]);
    }
    ctx->timed.clear();
    return Status{};
  });
}

int vdf_shim_set_cache(int entries) {
  if (entries < 0 || entries > 64) return VDF_ERR_BAD_ARG;
  {
    (void)vdf::tuning();                               // (environment overrides applied before the first publication)
    std::lock_guard<std::mutex> tl(g_tune_mu);
    vdf_hip_tuning t = vdf::tuning();
    t.shim_cache = entries;
    tuning_publish(t);
  }
  std::lock_guard<std::mutex> lock(g_shim_mu);
  while ((int)g_shim.size() > entries) { vdf_bases_free(g_shim.back().bases); g_shim.pop_back(); }
  return VDF_OK;
}

void mult_pippenger_pallas(vdf_jac* out, const vdf_affine* points, size_t npoints, const vdf_fe* scalars, bool is_mont) {
  shim(VDF_CURVE_PALLAS, out, points, npoints, scalars, is_mont);
}
void mult_pippenger_vesta(vdf_jac* out, const vdf_affine* points, size_t npoints, const vdf_fe* scalars, bool is_mont) {
  shim(VDF_CURVE_VESTA, out, points, npoints, scalars, is_mont);
}

// ---- R1CS shape -------------------------------------------------------------------------------
int vdf_shape_create(vdf_ctx* ctx, int field, size_t num_cons, size_t num_cols, const uint32_t* const rows[3],
                     const uint32_t* const cols[3], const vdf_fe* const vals[3], const size_t nnz[3], vdf_shape** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out || !rows || !cols || !vals || !nnz) return Status{VDF_ERR_BAD_ARG, "null argument"};
    *out = nullptr;
    if (field != VDF_FIELD_FP && field != VDF_FIELD_FQ) return Status{VDF_ERR_BAD_ARG, "unknown field"};
    if (num_cons >= (1ull << 31) || num_cols >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "shape too large"};
    std::array<uint32_t, 8> one, minus_one;
    if (field == VDF_FIELD_FP) build_dict_consts<FpParams>(one, minus_one); else build_dict_consts<FqParams>(one, minus_one);
    std::map<std::array<uint32_t, 8>, uint32_t> dict_idx;
    std::vector<std::array<uint32_t, 8>> dict;
    dict.push_back(one); dict_idx[one] = 0;
    dict.push_back(minus_one); dict_idx[minus_one] = 1;
    std::vector<uint32_t> rowptr[3], col[3], coef[3];
    for (int k = 0; k < 3; ++k) {
      const size_t z = nnz[k];
      if (z && (!rows[k] || !cols[k] || !vals[k])) return Status{VDF_ERR_BAD_ARG, "null matrix arrays"};
      if (z >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "too many non-zeros"};
      rowptr[k].assign(num_cons + 1, 0);
      for (size_t i = 0; i < z; ++i) {
        if (rows[k][i] >= num_cons || cols[k][i] >= num_cols) return Status{VDF_ERR_BAD_LENGTH, "matrix entry out of range"};
        rowptr[k][rows[k][i] + 1]++;
      }
      for (size_t r = 0; r < num_cons; ++r) rowptr[k][r + 1] += rowptr[k][r];
      col[k].resize(z); coef[k].resize(z);
      std::vector<uint32_t> fill(rowptr[k].begin(), rowptr[k].end() - 1);
      for (size_t i = 0; i < z; ++i) {
        std::array<uint32_t, 8> v;
        std::memcpy(v.data(), &vals[k][i], 32);
        auto it = dict_idx.find(v);
        uint32_t ci;
        if (it == dict_idx.end()) { ci = (uint32_t)dict.size(); dict.push_back(v); dict_idx[v] = ci; } else ci = it->second;
        uint32_t p = fill[rows[k][i]]++;
        col[k][p] = cols[k][i];
        coef[k][p] = ci;
      }
    }
    vdf_shape* s = new vdf_shape();
    s->ctx = ctx; s->field = field; s->num_cons = num_cons; s->num_cols = num_cols; s->dict_len = dict.size();
    auto cleanup = [&]() {
      for (int k = 0; k < 3; ++k) { (void)hipFree(s->d_rowptr[k]); (void)hipFree(s->d_col[k]); (void)hipFree(s->d_coef[k]); }
      (void)hipFree(s->d_dict);
      (void)hipFree(s->d_t_colptr); (void)hipFree(s->d_t_row); (void)hipFree(s->d_t_cm); (void)hipFree(s->d_t_heavy);
      (void)hipFree(s->d_long);
      (void)hipFree(s->d_long_rowlist);
      delete s;
    };
    hipError_t e = hipMalloc(&s->d_dict, dict.size() * 32);
    if (e == hipSuccess) e = hipMemcpy(s->d_dict, dict.data(), dict.size() * 32, hipMemcpyHostToDevice);
    for (int k = 0; k < 3 && e == hipSuccess; ++k) {
      s->nnz[k] = nnz[k];
      if (k == 0) s->h_nnz_prefix.assign(num_cons + 1, 0);
      for (size_t r = 0; r <= num_cons; ++r) s->h_nnz_prefix[r] += rowptr[k][r];       // entries of A, B, C above row r
      e = hipMalloc(reinterpret_cast<void**>(&s->d_rowptr[k]), (num_cons + 1) * 4);
      if (e == hipSuccess) e = hipMemcpy(s->d_rowptr[k], rowptr[k].data(), (num_cons + 1) * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_col[k]), (nnz[k] + 1) * 4);
      if (e == hipSuccess && nnz[k]) e = hipMemcpy(s->d_col[k], col[k].data(), nnz[k] * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_coef[k]), (nnz[k] + 1) * 4);
      if (e == hipSuccess && nnz[k]) e = hipMemcpy(s->d_coef[k], coef[k].data(), nnz[k] * 4, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
      // merged column-major form for vdf_spmv3_t
      std::vector<uint32_t> tcol(num_cols + 1, 0);
      for (int k = 0; k < 3; ++k) for (size_t i = 0; i < nnz[k]; ++i) tcol[cols[k][i] + 1]++;
      for (size_t c = 0; c < num_cols; ++c) tcol[c + 1] += tcol[c];
      const size_t nnz3 = tcol[num_cols];
      std::vector<uint32_t> trow(nnz3 + 1), tcm(nnz3 + 1), fill(tcol.begin(), tcol.end() - 1), heavy;
      for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < nnz[k]; ++i) {
          std::array<uint32_t, 8> v;
          std::memcpy(v.data(), &vals[k][i], 32);
          const uint32_t ci = dict_idx[v];
          if (ci >= (1u << 30)) { e = hipErrorInvalidValue; break; }
          const uint32_t p = fill[cols[k][i]]++;
          trow[p] = rows[k][i];
          tcm[p] = ci | ((uint32_t)k << 30);
        }
      for (size_t c = 0; c < num_cols; ++c) if (tcol[c + 1] - tcol[c] > 64) heavy.push_back((uint32_t)c);
      // longest first: the few columns of thousands of entries (the constant's, the step counter's) are shared by
      // several workgroups each, the many of a few hundred (bits used all over an augmented circuit) get one
      std::sort(heavy.begin(), heavy.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t la = tcol[a + 1] - tcol[a], lb = tcol[b + 1] - tcol[b];
        return la != lb ? la > lb : a < b;
      });
      s->t_nheavy = heavy.size();
      s->t_nbig = 0;
      while (s->t_nbig < heavy.size() && tcol[heavy[s->t_nbig] + 1] - tcol[heavy[s->t_nbig]] > 4096) ++s->t_nbig;
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_t_colptr), (num_cols + 1) * 4);
      if (e == hipSuccess) e = hipMemcpy(s->d_t_colptr, tcol.data(), (num_cols + 1) * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_t_row), (nnz3 + 1) * 4);
      if (e == hipSuccess) e = hipMemcpy(s->d_t_row, trow.data(), (nnz3 + 1) * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_t_cm), (nnz3 + 1) * 4);
      if (e == hipSuccess) e = hipMemcpy(s->d_t_cm, tcm.data(), (nnz3 + 1) * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_t_heavy), (heavy.size() + 1) * 4);
      if (e == hipSuccess && !heavy.empty()) e = hipMemcpy(s->d_t_heavy, heavy.data(), heavy.size() * 4, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
      std::vector<uint32_t> lng;
      for (int k = 0; k < 3; ++k)
        for (size_t r = 0; r < num_cons; ++r)
          if (rowptr[k][r + 1] - rowptr[k][r] > VDF_LONG_ROW) lng.push_back((uint32_t)r | ((uint32_t)k << 30));
      if (num_cons >= (1u << 30) && !lng.empty()) e = hipErrorInvalidValue;
      s->n_long = lng.size();
      for (uint32_t v : lng) s->h_long_rows.push_back(v & 0x3FFFFFFFu);
      std::sort(s->h_long_rows.begin(), s->h_long_rows.end());
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_long), (lng.size() + 1) * 4);
      if (e == hipSuccess && !lng.empty()) e = hipMemcpy(s->d_long, lng.data(), lng.size() * 4, hipMemcpyHostToDevice);
      std::vector<uint32_t> distinct(s->h_long_rows);
      distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
      s->n_long_rowlist = distinct.size();
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_long_rowlist), (distinct.size() + 1) * 4);
      if (e == hipSuccess && !distinct.empty())
        e = hipMemcpy(s->d_long_rowlist, distinct.data(), distinct.size() * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { cleanup(); return vdf::hip_status(e, "vdf_shape_create upload"); }
    *out = s;
    return Status{};
  });
}

void vdf_shape_free(vdf_shape* shape) {
  if (!shape) return;
  if (shape->ctx) {
    std::lock_guard<std::mutex> lock(shape->ctx->mu);
    (void)hipSetDevice(shape->ctx->device);
    (void)hipStreamSynchronize(shape->ctx->stream);
    for (int k = 0; k < 3; ++k) { (void)hipFree(shape->d_rowptr[k]); (void)hipFree(shape->d_col[k]); (void)hipFree(shape->d_coef[k]); }
    (void)hipFree(shape->d_dict);
    (void)hipFree(shape->d_t_colptr); (void)hipFree(shape->d_t_row); (void)hipFree(shape->d_t_cm); (void)hipFree(shape->d_t_heavy);
    (void)hipFree(shape->d_long);
    (void)hipFree(shape->d_long_rowlist);
  }
  delete shape;
}

int vdf_spmv3(vdf_ctx* ctx, const vdf_shape* shape, const vdf_fe* z, vdf_fe* Az, vdf_fe* Bz, vdf_fe* Cz) {
  return guarded(ctx, [&]() -> Status {
    // the shape is read-only device data: any context of its device may run over it
    if (!shape || !shape->ctx || shape->ctx->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "bad shape handle"};
    Staging st(ctx);
    const void* dz; void* o[3];
    VDF_TRY(st.in(z, shape->num_cols * 32, &dz));
    VDF_TRY(st.out(Az, shape->num_cons * 32, &o[0]));
    VDF_TRY(st.out(Bz, shape->num_cons * 32, &o[1]));
    VDF_TRY(st.out(Cz, shape->num_cons * 32, &o[2]));
    VDF_TRY(vdf::vec_spmv_long(shape->field, shape->d_rowptr, shape->d_col, shape->d_coef, shape->d_dict, dz, shape->d_long,
                               shape->n_long, o, ctx->stream));
    for (int k = 0; k < 3; ++k)
      VDF_TRY(vdf::vec_spmv(shape->field, shape->d_rowptr[k], shape->d_col[k], shape->d_coef[k], shape->d_dict, dz,
                            shape->num_cons, o[k], ctx->stream));
    return st.finish();
  });
}

// ---- vector ops -------------------------------------------------------------------------------
int vdf_cross_term(vdf_ctx* ctx, int field, const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1, const vdf_fe* Az2,
                   const vdf_fe* Bz2, const vdf_fe* Cz2, const vdf_fe* u1, size_t n, vdf_fe* T) {
  return guarded(ctx, [&]() -> Status {
    Staging st(ctx);
    const void* in[7]; void* dT;
    const vdf_fe* src[6] = {Az1, Bz1, Cz1, Az2, Bz2, Cz2};
    for (int k = 0; k < 6; ++k) VDF_TRY(st.in(src[k], n * 32, &in[k]));
    VDF_TRY(st.in(u1, 32, &in[6]));
    VDF_TRY(st.out(T, n * 32, &dT));
    VDF_TRY(vdf::vec_cross_term(field, in[0], in[1], in[2], in[3], in[4], in[5], in[6], n, dT, ctx->stream));
    return st.finish();
  });
}

int vdf_vec_is_zero(vdf_ctx* ctx, const vdf_fe* v, size_t n, int* is_zero) {
  return guarded(ctx, [&]() -> Status {
    if (!is_zero) return Status{VDF_ERR_BAD_ARG, "null out"};
    Staging st(ctx);
    const void* dv;
    VDF_TRY(st.in(v, n * 32, &dv));
    if (!ctx->reduce_scratch) VDF_TRY_HIP(hipMalloc(&ctx->reduce_scratch, vdf::snark_reduce_scratch_bytes()));
    uint32_t* d_flag = reinterpret_cast<uint32_t*>(ctx->reduce_scratch);
    VDF_TRY(vdf::vec_any_nonzero(dv, n, d_flag, ctx->stream));
    uint32_t flag = 1;
    VDF_TRY_HIP(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));                 // the answer is the point of the call
    *is_zero = flag == 0;
    return st.finish();
  });
}

int vdf_axpy(vdf_ctx* ctx, int field, const vdf_fe* a, const vdf_fe* r, const vdf_fe* b, size_t n, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    Staging st(ctx);
    const void *da, *dr, *db; void* dout;
    VDF_TRY(st.in(a, n * 32, &da));
    VDF_TRY(st.in(r, 32, &dr));
    VDF_TRY(st.in(b, n * 32, &db));
    VDF_TRY(st.out(out, n * 32, &dout));
    VDF_TRY(vdf::vec_axpy(field, da, dr, db, n, dout, ctx->stream));
    return st.finish();
  });
}

int vdf_minroot_witness(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, const vdf_fe* i0, uint64_t t, vdf_fe* W_segment) {
  return guarded(ctx, [&]() -> Status {
    if (t >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "t too large"};
    Staging st(ctx);
    const void *dt, *di; void* dw;
    VDF_TRY(st.in(trace_xy, (t + 1) * 64, &dt));
    VDF_TRY(st.in(i0, 32, &di));
    VDF_TRY(st.out(W_segment, (4 * t + 1) * 32, &dw));
    VDF_TRY(vdf::vec_minroot_witness(field, dt, di, t, dw, ctx->stream));
    return st.finish();
  });
}

// ---- fused step operations ------------------------------------------------------------------------
static int step_z_impl(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe z_in[3], const vdf_fe* i0,
                       const vdf_fe* u, const vdf_fe X[6], vdf_fe* z, vdf_fe* packed) {
  return guarded(ctx, [&]() -> Status {
    if (t == 0 || t >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "t out of range"};
    if (!z_in || !i0 || !u || !X) return Status{VDF_ERR_BAD_ARG, "null scalar operand"};
    if (ptr_is_device(z_in) || ptr_is_device(i0) || ptr_is_device(u) || ptr_is_device(X))
      return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
    if (!ptr_is_device(trace_xy) || !ptr_is_device(z) || (packed && !ptr_is_device(packed)))
      return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
    VDF_TRY(vdf::vec_step_z(field, trace_xy, t, z_in, i0, u, X, z, packed, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_minroot_step_z(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe z_in[3], const vdf_fe* i0,
                       const vdf_fe* u, const vdf_fe X[6], vdf_fe* z) {
  return step_z_impl(ctx, field, trace_xy, t, z_in, i0, u, X, z, nullptr);
}

int vdf_minroot_step_z_packed(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe z_in[3],
                              const vdf_fe* i0, const vdf_fe* u, const vdf_fe X[6], vdf_fe* z, vdf_fe* w_packed) {
  if (!w_packed) return ctx ? (ctx->err = "null w_packed", VDF_ERR_BAD_ARG) : VDF_ERR_BAD_ARG;
  return step_z_impl(ctx, field, trace_xy, t, z_in, i0, u, X, z, w_packed);
}

int vdf_minroot_step_segment(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe* i0, int vars_per_round,
                             vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    if (t == 0 || t >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "t out of range"};
    if (vars_per_round != 3 && vars_per_round != 4) return Status{VDF_ERR_BAD_ARG, "vars_per_round is 3 (bound) or 4 (reference)"};
    if (!i0 || ptr_is_device(i0)) return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
    if (!ptr_is_device(trace_xy) || !ptr_is_device(out))
      return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
    VDF_TRY(vdf::vec_step_segment(field, trace_xy, t, i0, vars_per_round, out, nullptr, nullptr, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_minroot_step_segment_packed(vdf_ctx* ctx, int field, const vdf_fe* trace_xy, uint64_t t, const vdf_fe* i0, const vdf_fe* i_in,
                                    vdf_fe* out, vdf_fe* packed) {
  return guarded(ctx, [&]() -> Status {
    if (t == 0 || t >= (1ull << 31)) return Status{VDF_ERR_BAD_LENGTH, "t out of range"};
    if (!i0 || !i_in || ptr_is_device(i0) || ptr_is_device(i_in)) return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
    if (!ptr_is_device(trace_xy) || !ptr_is_device(out) || !ptr_is_device(packed))
      return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
    VDF_TRY(vdf::vec_step_segment(field, trace_xy, t, i0, 4, out, packed, i_in, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

static Status nifs_cross_impl(vdf_ctx* ctx, const vdf_shape* shape, size_t row_begin, size_t row_count, int part, const vdf_fe* z2,
                              const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2,
                              vdf_fe* Cz2, vdf_fe* T) {
  // the shape is read-only device data: any context of its device may run over it (the early rows run on a second one)
  if (!shape || !shape->ctx || shape->ctx->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "bad shape handle"};
  if (!u1 || ptr_is_device(u1)) return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
  const void* vec[8] = {z2, Az1, Bz1, Cz1, Az2, Bz2, Cz2, T};
  for (const void* v : vec)
    if (!ptr_is_device(v)) return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
  if (part != VDF_ROWS_ALL && part != VDF_ROWS_INSIDE && part != VDF_ROWS_OUTSIDE) return Status{VDF_ERR_BAD_ARG, "unknown row selection"};
  if (part != VDF_ROWS_ALL) {
    if (row_begin > shape->num_cons || row_count > shape->num_cons - row_begin) return Status{VDF_ERR_BAD_LENGTH, "row range outside the shape"};
    // the range is meant for the uniform rows of a step circuit: a row summed by a wavefront (k_spmv_long) has no place in it
    const auto it = std::lower_bound(shape->h_long_rows.begin(), shape->h_long_rows.end(), (uint32_t)row_begin);
    if (row_count && it != shape->h_long_rows.end() && *it < row_begin + row_count)
      return Status{VDF_ERR_BAD_ARG, "the row range holds a row of more than VDF_LONG_ROW entries"};
  }
  size_t rows = shape->num_cons, skip_begin = shape->num_cons, skip_len = 0;
  if (part == VDF_ROWS_INSIDE) { rows = row_count; skip_begin = 0; skip_len = row_begin; }
  if (part == VDF_ROWS_OUTSIDE) { rows = shape->num_cons - row_count; skip_begin = row_begin; skip_len = row_count; }
  // the rows of more than VDF_LONG_ROW entries: summed inside the cross term's own launch when it runs eight lanes per row
  // (an augmented circuit, ~10^4 rows: one launch to wait for instead of two), else by a launch of their own before it
  const bool long_inside = part != VDF_ROWS_INSIDE && shape->n_long_rowlist > 0 && vdf::tuning().nifs_fused &&
                           vdf::nifs_cross_lanes(rows) == 8;
  if (part != VDF_ROWS_INSIDE && !long_inside) {
    void* const outs[3] = {Az2, Bz2, Cz2};
    VDF_TRY(vdf::vec_spmv_long(shape->field, shape->d_rowptr, shape->d_col, shape->d_coef, shape->d_dict, z2, shape->d_long,
                               shape->n_long, outs, ctx->stream));
  }
  // algorithmic bytes of the launch: per entry of A, B, C the column and coefficient index (8 B), per
  // row three row pointers and three results (3 x 36 B) and the seven vectors of the cross term (7 x 32 B); the gathered
  // elements of z are not counted (they are re-reads of a vector the launch already reads once through the cache)
  double alg_bytes = 0;
  if (shape->h_nnz_prefix.size() == shape->num_cons + 1) {
    const auto& pf = shape->h_nnz_prefix;
    const double nnz_all = (double)pf[shape->num_cons];
    const double nnz_range = (part == VDF_ROWS_ALL) ? 0.0 : (double)(pf[row_begin + row_count] - pf[row_begin]);
    const double nnz_run = part == VDF_ROWS_INSIDE ? nnz_range : nnz_all - nnz_range;
    alg_bytes = nnz_run * 8.0 + (double)rows * (3 * 36.0 + 7 * 32.0);
  }
  VDF_TRY(vdf::vec_nifs_cross(shape->field, shape->d_rowptr, shape->d_col, shape->d_coef, shape->d_dict, z2, Az1, Bz1, Cz1,
                              u1, rows, skip_begin, skip_len, long_inside ? shape->d_long_rowlist : nullptr,
                              long_inside ? shape->n_long_rowlist : 0, Az2, Bz2, Cz2, T, alg_bytes, ctx->stream));
  if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
  return Status{};
}

int vdf_nifs_cross_term(vdf_ctx* ctx, const vdf_shape* shape, const vdf_fe* z2, const vdf_fe* Az1, const vdf_fe* Bz1,
                        const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2, vdf_fe* Cz2, vdf_fe* T) {
  return guarded(ctx, [&]() -> Status { return nifs_cross_impl(ctx, shape, 0, 0, VDF_ROWS_ALL, z2, Az1, Bz1, Cz1, u1, Az2, Bz2, Cz2, T); });
}

int vdf_nifs_cross_term_rows(vdf_ctx* ctx, const vdf_shape* shape, size_t row_begin, size_t row_count, int part, const vdf_fe* z2,
                             const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2,
                             vdf_fe* Cz2, vdf_fe* T) {
  return guarded(ctx, [&]() -> Status { return nifs_cross_impl(ctx, shape, row_begin, row_count, part, z2, Az1, Bz1, Cz1, u1, Az2, Bz2, Cz2, T); });
}

int vdf_nifs_cross_term_minroot(vdf_ctx* ctx, int field, int vars_per_round, uint64_t t, size_t seg_begin, size_t one_col, size_t row_begin,
                                const vdf_fe* z2, const vdf_fe* Az1, const vdf_fe* Bz1, const vdf_fe* Cz1, const vdf_fe* u1, vdf_fe* Az2,
                                vdf_fe* Bz2, vdf_fe* Cz2, vdf_fe* T) {
  return guarded(ctx, [&]() -> Status {
    if (!z2 || !Az1 || !Bz1 || !Cz1 || !u1 || !Az2 || !Bz2 || !Cz2 || !T) return Status{VDF_ERR_BAD_ARG, "null argument"};
    if (ptr_is_device(u1)) return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
    for (const void* v : {(const void*)z2, (const void*)Az1, (const void*)Bz1, (const void*)Cz1, (const void*)Az2, (const void*)Bz2,
                          (const void*)Cz2, (const void*)T})
      if (!ptr_is_device(v)) return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
    if (t == 0 || t > (1ull << 26)) return Status{VDF_ERR_BAD_LENGTH, "t out of range"};
    if (seg_begin < 3 || one_col < seg_begin + (size_t)vars_per_round * t + 1) return Status{VDF_ERR_BAD_ARG, "the constant's column lies behind the rounds"};
    VDF_TRY(vdf::vec_nifs_cross_minroot(field, vars_per_round, t, seg_begin, one_col, row_begin, z2, Az1, Bz1, Cz1, u1, Az2, Bz2, Cz2, T, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_nifs_cross_term_minroot_fold(vdf_ctx* ctx, int field, int vars_per_round, uint64_t t, size_t seg_begin, size_t one_col,
                                     size_t row_begin, const vdf_fe* z2, const vdf_fe* r, vdf_fe* Az1, vdf_fe* Bz1, vdf_fe* Cz1,
                                     vdf_fe* E1, const vdf_fe* T_prev, const vdf_fe* u1, vdf_fe* Az2, vdf_fe* Bz2, vdf_fe* Cz2,
                                     vdf_fe* T) {
  return guarded(ctx, [&]() -> Status {
    if (!z2 || !r || !Az1 || !Bz1 || !Cz1 || !u1 || !Az2 || !Bz2 || !Cz2 || !T) return Status{VDF_ERR_BAD_ARG, "null argument"};
    if (E1 && !T_prev) return Status{VDF_ERR_BAD_ARG, "E1 without T_prev"};
    if (ptr_is_device(u1) || ptr_is_device(r)) return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
    for (const void* v : {(const void*)z2, (const void*)Az1, (const void*)Bz1, (const void*)Cz1, (const void*)Az2, (const void*)Bz2,
                          (const void*)Cz2, (const void*)T})
      if (!ptr_is_device(v)) return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
    if (E1 && (!ptr_is_device(E1) || !ptr_is_device(T_prev))) return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
    if (t == 0 || t > (1ull << 26)) return Status{VDF_ERR_BAD_LENGTH, "t out of range"};
    if (seg_begin < 3 || one_col < seg_begin + (size_t)vars_per_round * t + 1) return Status{VDF_ERR_BAD_ARG, "the constant's column lies behind the rounds"};
    VDF_TRY(vdf::vec_nifs_cross_minroot_fold(field, vars_per_round, t, seg_begin, one_col, row_begin, z2, r, Az1, Bz1, Cz1, E1, E1 ? T_prev : nullptr,
                                             u1, Az2, Bz2, Cz2, T, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_fold_many(vdf_ctx* ctx, int field, const vdf_fe* r, int k, vdf_fe* const acc[], const vdf_fe* const add[],
                  const size_t n[]) {
  return guarded(ctx, [&]() -> Status {
    if (k < 0 || k > 8) return Status{VDF_ERR_BAD_ARG, "k must be 0..8"};
    if (k == 0) return Status{};
    if (!r || !acc || !add || !n) return Status{VDF_ERR_BAD_ARG, "null argument"};
    if (ptr_is_device(r)) return Status{VDF_ERR_BAD_ARG, "scalar operands of fused calls live in host memory"};
    void* a[8]; const void* b[8];
    for (int i = 0; i < k; ++i) {
      if (n[i] && (!ptr_is_device(acc[i]) || !ptr_is_device(add[i])))
        return Status{VDF_ERR_BAD_ARG, "vector operands of fused calls live in device memory"};
      a[i] = acc[i]; b[i] = add[i];
    }
    VDF_TRY(vdf::vec_fold_many(field, r, k, a, b, n, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_ctx_wait(vdf_ctx* ctx, vdf_ctx* other) {
  if (!other || other == ctx) return ctx ? VDF_OK : VDF_ERR_BAD_ARG;
  return guarded(ctx, [&]() -> Status {
    if (other->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "contexts live on different devices"};
    if (!ctx->wait_ev) VDF_TRY_HIP(hipEventCreateWithFlags(&ctx->wait_ev, hipEventDisableTiming));
    VDF_TRY_HIP(hipEventRecord(ctx->wait_ev, other->stream));
    VDF_TRY_HIP(hipStreamWaitEvent(ctx->stream, ctx->wait_ev, 0));
    return Status{};
  });
}

int vdf_ctx_mark(vdf_ctx* ctx, int slot) {
  return guarded(ctx, [&]() -> Status {
    if (slot < 0 || slot >= VDF_MARK_SLOTS) return Status{VDF_ERR_BAD_ARG, "mark slot must be 0..15"};
    if (!ctx->marks[slot]) VDF_TRY_HIP(hipEventCreateWithFlags(&ctx->marks[slot], hipEventDisableTiming));
    VDF_TRY_HIP(hipEventRecord(ctx->marks[slot], ctx->stream));
    return Status{};
  });
}

int vdf_ctx_wait_mark(vdf_ctx* ctx, vdf_ctx* other, int slot) {
  if (!other) return VDF_ERR_BAD_ARG;
  return guarded(ctx, [&]() -> Status {
    if (slot < 0 || slot >= VDF_MARK_SLOTS) return Status{VDF_ERR_BAD_ARG, "mark slot must be 0..15"};
    if (other->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "contexts live on different devices"};
    if (!other->marks[slot]) return Status{VDF_ERR_BAD_ARG, "no mark was set in this slot"};
    if (other == ctx) return Status{};
    VDF_TRY_HIP(hipStreamWaitEvent(ctx->stream, other->marks[slot], 0));
    return Status{};
  });
}

int vdf_ctx_set_accumulate_fill(vdf_ctx* ctx, int workgroups_per_cu) {
  return guarded(ctx, [&]() -> Status {
    if (workgroups_per_cu < 0 || workgroups_per_cu > 3) return Status{VDF_ERR_BAD_ARG, "accumulate fill must be 0 (process-wide) or 1..3"};
    ctx->acc_fill = workgroups_per_cu;
    return Status{};
  });
}

int vdf_ctx_set_light_priority(vdf_ctx* ctx, int priority) {
  return guarded(ctx, [&]() -> Status {
    if (priority < 0 || priority > 3) return Status{VDF_ERR_BAD_ARG, "wave priority is 0..3"};
    ctx->light_prio = priority;
    return Status{};
  });
}

int vdf_ctx_gate_accumulate(vdf_ctx* ctx, vdf_ctx* other, int slot) {
  if (!other) return VDF_ERR_BAD_ARG;
  return guarded(ctx, [&]() -> Status {
    if (slot < 0 || slot >= VDF_MARK_SLOTS) return Status{VDF_ERR_BAD_ARG, "mark slot must be 0..15"};
    if (other->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "contexts live on different devices"};
    if (!other->marks[slot]) return Status{VDF_ERR_BAD_ARG, "no mark was set in this slot"};
    ctx->acc_gate = other == ctx ? nullptr : other->marks[slot];
    return Status{};
  });
}

int vdf_ctx_sync_mark(vdf_ctx* ctx, int slot) {
  return guarded(ctx, [&]() -> Status {
    if (slot < 0 || slot >= VDF_MARK_SLOTS) return Status{VDF_ERR_BAD_ARG, "mark slot must be 0..15"};
    if (!ctx->marks[slot]) return Status{VDF_ERR_BAD_ARG, "no mark was set in this slot"};
    for (int spin = 0; spin < 4000; ++spin) {                    // poll first, as vdf_ctx_sync does
      hipError_t q = hipEventQuery(ctx->marks[slot]);
      if (q == hipSuccess) return Status{};
      if (q != hipErrorNotReady) return vdf::hip_status(q, "hipEventQuery");
    }
    VDF_TRY_HIP(hipEventSynchronize(ctx->marks[slot]));
    return Status{};
  });
}

// ---- compression SNARK building blocks ---------------------------------------------------------------------
namespace {
bool all_device(std::initializer_list<const void*> ps) {
  for (const void* p : ps) if (!ptr_is_device(p)) return false;
  return true;
}
const char* kDevVec = "vector operands of the SNARK building blocks live in device memory";
const char* kHostScalar = "scalar operands of the SNARK building blocks live in host memory";
}  // namespace

int vdf_pair_table(vdf_ctx* ctx, int field, const vdf_fe* lo, const vdf_fe* hi, int k, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    if (k < 0 || k > 24) return Status{VDF_ERR_BAD_LENGTH, "0..24 variables"};
    if (k && (!lo || !hi || ptr_is_device(lo) || ptr_is_device(hi))) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    if (!ptr_is_device(out)) return Status{VDF_ERR_BAD_ARG, kDevVec};
    VDF_TRY(vdf::snark_pair_table(field, lo, hi, k, out, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_pair_table_pattern(vdf_ctx* ctx, int field, const vdf_fe* lo, const vdf_fe* hi, int k, const vdf_fe* pattern, int log_m,
                           vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    if (k < 0 || log_m < 0 || log_m > 4 || k + log_m > 24) return Status{VDF_ERR_BAD_LENGTH, "0..24 variables in all, pattern of 1..16"};
    if (k && (!lo || !hi || ptr_is_device(lo) || ptr_is_device(hi))) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    if (!pattern || ptr_is_device(pattern)) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    if (!ptr_is_device(out)) return Status{VDF_ERR_BAD_ARG, kDevVec};
    VDF_TRY(vdf::snark_pair_table_pattern(field, lo, hi, k, pattern, log_m, out, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_fold_halves(vdf_ctx* ctx, int field, int k, vdf_fe* const v[], const vdf_fe c_lo[], const vdf_fe c_hi[], size_t n) {
  return guarded(ctx, [&]() -> Status {
    if (k < 0 || k > 8) return Status{VDF_ERR_BAD_ARG, "k must be 0..8"};
    if (k == 0) return Status{};
    if (!v || !c_lo || !c_hi || ptr_is_device(c_lo) || ptr_is_device(c_hi)) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    void* dv[8];
    for (int i = 0; i < k; ++i) { if (!ptr_is_device(v[i])) return Status{VDF_ERR_BAD_ARG, kDevVec}; dv[i] = v[i]; }
    VDF_TRY(vdf::snark_fold_halves(field, k, dv, c_lo, c_hi, n, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_reduce(vdf_ctx* ctx, int field, int kind, const vdf_fe* const tables[], const vdf_fe* u, size_t n, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    if (kind < 0 || kind > 3 || !tables || !out) return Status{VDF_ERR_BAD_ARG, "bad argument"};
    if (kind == 2 && (!u || ptr_is_device(u))) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    const int ntab = kind == 2 ? 5 : 2, nout = kind == 0 ? 1 : kind == 2 ? 3 : 2;
    const void* t[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < ntab; ++i) { if (n && !ptr_is_device(tables[i])) return Status{VDF_ERR_BAD_ARG, kDevVec}; t[i] = tables[i]; }
    if (!ctx->reduce_scratch) VDF_TRY_HIP(hipMalloc(&ctx->reduce_scratch, vdf::snark_reduce_scratch_bytes()));
    Staging st(ctx);
    void* d_out = nullptr;
    VDF_TRY(st.out(out, (size_t)nout * sizeof(vdf_fe), &d_out));
    VDF_TRY(vdf::snark_reduce(field, kind, t, u, n, ctx->reduce_scratch, d_out, ctx->stream));
    return st.finish();
  });
}

int vdf_spmv3_t(vdf_ctx* ctx, const vdf_shape* shape, const vdf_fe* eq, const vdf_fe* rho, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    // the shape is read-only device data: any context of its device may run over it
    if (!shape || !shape->ctx || shape->ctx->device != ctx->device) return Status{VDF_ERR_BAD_ARG, "bad shape handle"};
    if (!rho || ptr_is_device(rho)) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    if (!all_device({eq, out})) return Status{VDF_ERR_BAD_ARG, kDevVec};
    if (!ctx->reduce_scratch) VDF_TRY_HIP(hipMalloc(&ctx->reduce_scratch, vdf::snark_reduce_scratch_bytes()));
    // algorithmic bytes of M(y) = sum_x eq[x] (A + rho B + rho^2 C)[x, y]: per entry a row index, a packed coefficient index and
    // the gathered eq value; per column a pointer and the result (SURVEY 8d's SpMV formula, transposed)
    const double nnz3 = (double)(shape->nnz[0] + shape->nnz[1] + shape->nnz[2]);
    vdf::KTimer kt(ctx->stream, "k_spmvt(+heavy cols)", nnz3 * (4 + 4 + 32) + (double)shape->num_cols * (4 + 32));
    VDF_TRY(vdf::snark_spmvt(shape->field, shape->d_t_colptr, shape->d_t_row, shape->d_t_cm, shape->d_t_heavy, shape->t_nheavy,
                             shape->t_nbig, shape->d_dict, eq, rho, shape->num_cols, out, ctx->reduce_scratch, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_ipa_scalars(vdf_ctx* ctx, int field, const vdf_fe* a, const vdf_fe* s, size_t n, size_t nj, vdf_fe* sL, vdf_fe* sR) {
  return guarded(ctx, [&]() -> Status {
    if (!all_device({a, s, sL, sR})) return Status{VDF_ERR_BAD_ARG, kDevVec};
    VDF_TRY(vdf::snark_ipa_scalars(field, a, s, n, nj, sL, sR, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_scale_pattern(vdf_ctx* ctx, int field, vdf_fe* s, size_t n, size_t nj, const vdf_fe* x_lo, const vdf_fe* x_hi) {
  return guarded(ctx, [&]() -> Status {
    if (!x_lo || !x_hi || ptr_is_device(x_lo) || ptr_is_device(x_hi)) return Status{VDF_ERR_BAD_ARG, kHostScalar};
    if (!ptr_is_device(s)) return Status{VDF_ERR_BAD_ARG, kDevVec};
    VDF_TRY(vdf::snark_scale_pattern(field, s, n, nj, x_lo, x_hi, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_fe_mul(vdf_ctx* ctx, int field, const vdf_fe* a, const vdf_fe* b, size_t n, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    Staging st(ctx);
    const void *da, *db; void* dout;
    VDF_TRY(st.in(a, n * 32, &da));
    VDF_TRY(st.in(b, n * 32, &db));
    VDF_TRY(st.out(out, n * 32, &dout));
    VDF_TRY(vdf::vec_mul(field, da, db, n, dout, ctx->stream));
    return st.finish();
  });
}

int vdf_fe_to_mont(vdf_ctx* ctx, int field, const vdf_fe* a, size_t n, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    Staging st(ctx);
    const void* da; void* dout;
    VDF_TRY(st.in(a, n * 32, &da));
    VDF_TRY(st.out(out, n * 32, &dout));
    VDF_TRY(vdf::vec_to_mont(field, da, n, dout, ctx->stream));
    return st.finish();
  });
}

int vdf_fe_from_mont(vdf_ctx* ctx, int field, const vdf_fe* a, size_t n, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    Staging st(ctx);
    const void* da; void* dout;
    VDF_TRY(st.in(a, n * 32, &da));
    VDF_TRY(st.out(out, n * 32, &dout));
    VDF_TRY(vdf::vec_from_mont(field, da, n, dout, ctx->stream));
    return st.finish();
  });
}

int vdf_fe_mul_chain(vdf_ctx* ctx, int field, const vdf_fe* a, size_t n, int iters, vdf_fe* out) {
  return guarded(ctx, [&]() -> Status {
    Staging st(ctx);
    const void* da; void* dout;
    VDF_TRY(st.in(a, n * 32, &da));
    VDF_TRY(st.out(out, n * 32, &dout));
    VDF_TRY(vdf::vec_mul_chain(field, da, n, iters, dout, ctx->stream));
    return st.finish();
  });
}

int vdf_ctx_clock_probe(vdf_ctx* ctx, int iters, double* shader_mhz, double* kernel_ms) {
  return guarded(ctx, [&]() -> Status {
    if (iters < 1 || iters > (1 << 22)) return Status{VDF_ERR_BAD_ARG, "iters out of range"};
    unsigned long long* d = reinterpret_cast<unsigned long long*>(ctx->small_pool);
    if (!d) return Status{VDF_ERR_DEVICE, "no staging pool"};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    VDF_TRY_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e0); return Status{VDF_ERR_DEVICE, "hipEventCreate"}; }
    Status st{};
    unsigned long long h[3] = {0, 0, 0};
    float ms = 0.f;
    hipError_t e = hipMemsetAsync(d, 0, 24, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) st = vdf::vec_clock_probe(iters, 2 * ctx->num_cus, d, ctx->stream);     // two wavefronts per SIMD
    if (e == hipSuccess && st.code == VDF_OK) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess && st.code == VDF_OK) e = hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && st.code == VDF_OK) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && st.code == VDF_OK) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (st.code != VDF_OK) return st;
    VDF_TRY_HIP(e);
    if (shader_mhz) *shader_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    if (kernel_ms) *kernel_ms = ms;
    return Status{};
  });
}

// ---- device memory helpers ----------------------------------------------------------------------
int vdf_dev_alloc(vdf_ctx* ctx, size_t bytes, void** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    *out = nullptr;
    VDF_TRY_HIP(hipMalloc(out, bytes ? bytes : 1));
    return Status{};
  });
}

int vdf_dev_mem_info(vdf_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  return guarded(ctx, [&]() -> Status {
    size_t f = 0, t = 0;
    VDF_TRY_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return Status{};
  });
}

int vdf_dev_free(vdf_ctx* ctx, void* p) {
  return guarded(ctx, [&]() -> Status {
    if (!p) return Status{};
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    VDF_TRY_HIP(hipFree(p));
    return Status{};
  });
}

int vdf_host_alloc(vdf_ctx* ctx, size_t bytes, void** out) {
  return guarded(ctx, [&]() -> Status {
    if (!out) return Status{VDF_ERR_BAD_ARG, "null out"};
    *out = nullptr;
    void* h = nullptr;
    VDF_TRY_HIP(hipHostMalloc(&h, bytes ? bytes : 1, hipHostMallocMapped));
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || d != h) {      // unified addressing is what "in place" relies on
      (void)hipGetLastError();
      (void)hipHostFree(h);
      return Status{VDF_ERR_DEVICE, "pinned host memory is not addressable in place by the device"};
    }
    std::lock_guard<std::mutex> lock(g_host_mu);
    g_host_allocs.emplace_back((const char*)h, bytes ? bytes : 1);
    *out = h;
    return Status{};
  });
}

int vdf_host_free(vdf_ctx* ctx, void* p) {
  return guarded(ctx, [&]() -> Status {
    if (!p) return Status{};
    {
      std::lock_guard<std::mutex> lock(g_host_mu);
      bool found = false;
      for (size_t i = 0; i < g_host_allocs.size(); ++i)
        if (g_host_allocs[i].first == (const char*)p) { g_host_allocs.erase(g_host_allocs.begin() + i); found = true; break; }
      if (!found) return Status{VDF_ERR_BAD_ARG, "not a vdf_host_alloc pointer"};
    }
    VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    VDF_TRY_HIP(hipHostFree(p));
    return Status{};
  });
}

int vdf_dev_memcpy(vdf_ctx* ctx, void* dst, const void* src, size_t bytes) {
  return guarded(ctx, [&]() -> Status {
    if (bytes == 0) return Status{};
    if (!dst || !src) return Status{VDF_ERR_BAD_ARG, "null pointer"};
    const bool both_dev = ptr_is_device(dst) && ptr_is_device(src);
    VDF_TRY_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    if (!(both_dev && ctx->async)) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

int vdf_dev_memset(vdf_ctx* ctx, void* dst, int value, size_t bytes) {
  return guarded(ctx, [&]() -> Status {
    if (bytes == 0) return Status{};
    if (!dst) return Status{VDF_ERR_BAD_ARG, "null pointer"};
    VDF_TRY_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    if (!ctx->async) VDF_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return Status{};
  });
}

}  // extern "C"
