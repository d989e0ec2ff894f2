// libvdf_nova.so, part 3: NovaVDFProof::compress and verification of the compressed proof: one Spartan-style argument
// per side of the curve cycle (SS1 / SS2 of src/nova/proof.rs:32-33), after the last secondary instance is folded.
#include <thread>
#include "nova_internal.hpp"

using namespace vdfnova;

// =============================================================================================================
// Compression SNARK: NovaVDFProof::compress / verification of the compressed proof (src/nova/proof.rs:360-368, :383).
// Protocol "vdf-spartan-v3", restated line by line by the test oracle (spartan.py: prove / verify); every pass over a vector is
// a call through include/vdf_hip.h, the host keeps the transcript, O(log n) field work and O(log n) point work.
// =============================================================================================================
namespace {

struct Transcript {
  uint8_t state[32];
  explicit Transcript(const char* label) {
    Shake256 h;
    h.absorb("vdf-spartan-v3|", 15);
    h.absorb(label, strlen(label));
    h.squeeze(state, 32);
  }
  void absorb(const char* label, const void* data, size_t n) {
    Shake256 h;
    h.absorb(state, 32);
    h.absorb(label, strlen(label));
    h.absorb(":", 1);
    h.absorb(data, n);
    h.squeeze(state, 32);
  }
  void absorb_fe(const char* label, const Fe* v, size_t k, const Field& F) {
    std::vector<uint8_t> b(k * 32);
    for (size_t i = 0; i < k; ++i) { const Fe c = from_mont(v[i], F); memcpy(&b[i * 32], c.l, 32); }
    absorb(label, b.data(), b.size());
  }
  void absorb_pt(const char* label, const Aff* p, size_t k, const Field& F) {      // F: the points' coordinate field
    std::vector<uint8_t> b(k * 64, 0);
    for (size_t i = 0; i < k; ++i)
      if (!p[i].is_id()) {
        const Fe x = from_mont(p[i].x, F), y = from_mont(p[i].y, F);
        memcpy(&b[i * 64], x.l, 32); memcpy(&b[i * 64 + 32], y.l, 32);
      }
    absorb(label, b.data(), b.size());
  }
  // 128-bit challenge: raw = the integer, return = its Montgomery form in F
  Fe challenge(const char* label, const Field& F, uint64_t raw[4]) {
    Shake256 h;
    h.absorb(state, 32);
    h.absorb(label, strlen(label));
    h.absorb("?", 1);
    uint8_t out[48];
    h.squeeze(out, 48);
    memcpy(state, out + 16, 32);
    raw[0] = raw[1] = raw[2] = raw[3] = 0;
    memcpy(raw, out, 16);
    Fe r;
    memcpy(r.l, raw, 32);
    return to_mont(r, F);
  }
};

// The halving of an inner-product argument stops at IPA_STOP elements: the prover sends that vector instead of four more
// rounds (each a pair of MSMs over all generators for the prover; the verifier's one MSM is the same either way).
constexpr size_t IPA_STOP = 16;
inline size_t ipa_final(size_t n) { return n < IPA_STOP ? n : IPA_STOP; }
inline size_t ipa_rounds(size_t n) { size_t k = 0; for (size_t m = n; m > IPA_STOP; m >>= 1) ++k; return k; }
struct Ipa { std::vector<Aff> L, R; std::vector<Fe> a; };
struct Spartan {
  std::vector<std::array<Fe, 3>> outer;
  Fe claims[4];
  std::vector<std::array<Fe, 2>> inner;
  Fe w_eval;
  Ipa ipaW, ipaE;
};

size_t pow2_at_least(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
int log2_exact(size_t n) { int k = 0; while (((size_t)1 << k) < n) ++k; return k; }

struct Layout { size_t M, NW, Z; int s, l1; };
Layout layout_of(const Side& pp_) {
  const Side* pp = &pp_;
  Layout l;
  l.M = pow2_at_least(pp->num_cons); l.NW = pow2_at_least(pp->num_vars); l.Z = 2 * l.NW;
  l.s = log2_exact(l.M); l.l1 = log2_exact(l.Z);
  return l;
}

// device buffers released on scope exit
struct DevBufs {
  vdf_ctx* ctx;
  Arena* arena;                    // the side's block (kept by the parameter set), or none: every vector its own allocation
  size_t reserved = 0, used = 0;
  std::vector<void*> v;
  explicit DevBufs(vdf_ctx* c, Arena* a = nullptr) : ctx(c), arena(a) {}
  ~DevBufs() { for (void* p : v) vdf_dev_free(ctx, p); }
  // all vectors of a call at once: one allocation the first time (or when a larger call comes), one memset every time
  int reserve(size_t elems) {
    if (!arena) return VDF_OK;
    const size_t bytes = elems * 32;
    if (arena->cap < bytes) {
      if (arena->p) { vdf_ctx_sync(ctx); vdf_dev_free(ctx, arena->p); arena->p = nullptr; arena->cap = 0; }
      int rc = vdf_dev_alloc(ctx, bytes, &arena->p);
      if (rc != VDF_OK) return rc;
      arena->cap = bytes;
    }
    reserved = bytes; used = 0;
    return vdf_dev_memset(ctx, arena->p, 0, bytes);
  }
  int zeros(size_t elems, void** out) {
    if (arena && used + elems * 32 <= reserved) { *out = static_cast<char*>(arena->p) + used; used += elems * 32; return VDF_OK; }
    int rc = vdf_dev_alloc(ctx, elems * 32, out);
    if (rc != VDF_OK) return rc;
    v.push_back(*out);
    return vdf_dev_memset(ctx, *out, 0, elems * 32);
  }
};

Fe small(uint64_t k, const Field& F) { return from_u64(k, F); }
// value at r of the polynomial through (0, y0), (1, y1), (2, y2)[, (3, y3)]
Fe interpolate(const Fe* y, int npts, const Fe& r, const Field& F) {
  Fe acc = zero();
  for (int i = 0; i < npts; ++i) {
    Fe num = one(F), den = one(F);
    for (int j = 0; j < npts; ++j)
      if (i != j) {
        num = mul(num, sub(r, small(j, F), F), F);
        den = mul(den, sub(small(i, F), small(j, F), F), F);
      }
    acc = add(acc, mul(y[i], mul(num, inverse(den, F), F), F), F);
  }
  return acc;
}

void instance_bytes(Transcript& tr, const Side& sd, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X) {
  const Field& F = *sd.F;
  tr.absorb("shape", sd.digest, 32);
  const Aff pts[2] = {cW, cE};
  tr.absorb_pt("inst", pts, 2, *sd.Fb);
  Fe v[1 + NUM_IO];
  v[0] = u;
  for (int j = 0; j < NUM_IO; ++j) v[1 + j] = X[j];
  tr.absorb_fe("inst", v, 1 + NUM_IO, F);
}

// lo = 1 - r, hi = r table of eq(r, .) on the device
int eq_table_dev(const Side& sd, const std::vector<Fe>& r, void* out) {
  vdf_ctx* ctx = sd.ctx;
  const Field& F = *sd.F;
  std::vector<Fe> lo(r.size());
  for (size_t j = 0; j < r.size(); ++j) lo[j] = sub(one(F), r[j], F);
  HIPCALL(ctx, vdf_pair_table(ctx, sd.field, (const vdf_fe*)lo.data(), (const vdf_fe*)r.data(), (int)r.size(), (vdf_fe*)out));
  return VDF_OK;
}

// M(y) in the padded layout (W at [0, NW), u at NW, X after it) from eq(r_x, .)
int m_vector_dev(const Side& sd, const Layout& L, const void* d_eq_rx, const Fe& rho, void* d_cols, void* d_mvec) {
  const Side* pp = &sd;
  vdf_ctx* ctx = sd.ctx;
  HIPCALL(ctx, vdf_spmv3_t(ctx, pp->shape, (const vdf_fe*)d_eq_rx, (const vdf_fe*)&rho, (vdf_fe*)d_cols));
  HIPCALL(ctx, vdf_dev_memset(ctx, d_mvec, 0, L.Z * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_mvec, d_cols, pp->num_vars * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)d_mvec + L.NW * 32, (const char*)d_cols + pp->num_vars * 32, (1 + NUM_IO) * 32));
  return VDF_OK;
}

Pt pt_mul_fe(const Pt& p, const Fe& k_mont, const Field& Fscalar, const Field& Fb) {       // k in Montgomery form of the scalar field
  const Fe k = from_mont(k_mont, Fscalar);
  return pt_mul(p, k.l, 255, Fb);
}

// Fixed-base multiples of one point (the Q of an inner-product argument is multiplied by two fresh scalars per
// round): 4-bit windows, d * 16^w * Q for d = 1..15, so a product is 64 additions and no doubling.
struct FixedBase {
  std::vector<Pt> tab;            // [w * 15 + (d - 1)]
  const Field& Fb;
  FixedBase(const Pt& q, const Field& fb) : tab(64 * 15), Fb(fb) {
    Pt base = q;
    for (int w = 0; w < 64; ++w) {
      Pt acc = base;
      for (int d = 1; d <= 15; ++d) { tab[w * 15 + d - 1] = acc; acc = pt_add(acc, base, Fb); }
      base = acc;                  // 16 * base
    }
  }
  Pt mul(const Fe& k_mont, const Field& Fscalar) const {
    const Fe k = from_mont(k_mont, Fscalar);
    Pt r = pt_identity();
    for (int w = 0; w < 64; ++w) {
      const unsigned d = (unsigned)(k.l[w / 16] >> (4 * (w % 16))) & 15u;
      if (d) r = pt_add(r, tab[w * 15 + d - 1], Fb);
    }
    return r;
  }
};

// One opening <a, b> = v under P = <a, G[0..n)>: a, b, s are device vectors of length n that the argument consumes
// (s = all ones), sL / sR scratch of the same length.
struct IpaJob {
  const char* label;
  size_t n;
  void *d_a, *d_b, *d_s, *d_sL, *d_sR;
  Fe v;
  Aff P;
  Ipa* out;
  // state
  size_t nj = 0;
  uint64_t q_raw[4];               // the 128-bit challenge Q = q_raw * gen_u comes from
  Pt Qp;
  std::unique_ptr<FixedBase> Qtab;
  // Q and its table of 960 multiples (~0.15 ms of host work): wanted when the first round's points come back, not before
  void make_q(const Aff& gen_u, const Field& Fb) {
    Qp = pt_mul(pt_from_aff(gen_u, Fb), q_raw, 128, Fb);
    Qtab.reset(new FixedBase(Qp, Fb));
  }
  Fe* cross = nullptr;             // two elements in pinned, device-mapped memory: the reduction writes them in place and they are
                                   // read after the round's one synchronisation (behind the MSM), not after one of their own
};

// The same two openings, same transcript, on TWO queues half a round apart.  A round's batched MSM is sort (0.26 ms at
// 2^19 + 2^18 generators), bucket accumulation (0.91), fix-up and bucket reduction (0.26): only the middle one fills the
// device, and in lockstep the device idles through the other two and through the host's turn, 15 times.  Here opening W
// runs on the side's queue and opening E on a second one, E's accumulation gated behind W's whole MSM
// (vdf_ctx_gate_accumulate): E's sort runs under W's accumulation, E's accumulation under W's host turn and W's next sort,
// E's reduction under W's next accumulation.  The transcript sees what it saw before -- round by round W's L, R and
// challenge, then E's -- because only LAUNCHES move: W's next round is enqueued as soon as W's challenge is drawn, before
// E's points of this round are read.  The bytes of the proof do not change (tests/test_gpu_compress.py compares them with
// the oracle's lockstep prover).
constexpr int IPA_MARK = 8;                  // (a library slot, vdf_hip.h: 0..3 on the caller's context are the caller's; prove_step holds 4..7)
static int ipa_prove_two_queues(const Side& sd, Transcript& tr, IpaJob* jobs, vdf_jac* h_lr) {
  const Side* pp = &sd;
  vdf_ctx* cq[2] = {sd.ctx, sd.ctx_b};
  const Field& F = *sd.F;
  const Field& Fb = *sd.Fb;
  uint64_t raw[4];
  // whatever way this function is left, nothing of it is still on either queue: the caller frees the pinned result slots
  struct Drain { vdf_ctx* a; vdf_ctx* b; ~Drain() { (void)vdf_ctx_sync(a); (void)vdf_ctx_sync(b); } } drain{cq[0], cq[1]};
  HIPCALL(cq[1], vdf_ctx_set_async(cq[1], 1));
  HIPCALL(cq[1], vdf_ctx_wait(cq[1], cq[0]));                          // the E opening's vectors were made on the first queue
  bool w_marked = false;
  auto enqueue = [&](int q) -> int {                                   // this round's L and R of opening q, on its queue
    IpaJob& jb = jobs[q];
    vdf_ctx* ctx = cq[q];
    const vdf_fe* ab[2] = {(const vdf_fe*)jb.d_a, (const vdf_fe*)jb.d_b};
    HIPCALL(ctx, vdf_reduce(ctx, sd.field, VDF_REDUCE_IPA_CROSS, ab, nullptr, jb.nj, (vdf_fe*)jb.cross));        // pinned: no wait here
    HIPCALL(ctx, vdf_ipa_scalars(ctx, sd.field, (const vdf_fe*)jb.d_a, (const vdf_fe*)jb.d_s, jb.n, jb.nj, (vdf_fe*)jb.d_sL,
                                 (vdf_fe*)jb.d_sR));
    if (q == 1 && w_marked) HIPCALL(ctx, vdf_ctx_gate_accumulate(ctx, cq[0], IPA_MARK));
    const size_t off[2] = {0, 0}, len[2] = {jb.n, jb.n};
    const vdf_fe* sc[2] = {(const vdf_fe*)jb.d_sL, (const vdf_fe*)jb.d_sR};
    HIPCALL(ctx, vdf_msm_batch(ctx, pp->gens, 2, off, sc, len, 1, h_lr + 2 * q));
    if (q == 0) { HIPCALL(ctx, vdf_ctx_mark(ctx, IPA_MARK)); w_marked = true; }
    return VDF_OK;
  };
  auto finish = [&](int q) -> int {                                    // the round's points, challenge and fold of opening q
    IpaJob& jb = jobs[q];
    vdf_ctx* ctx = cq[q];
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    Aff l0, r0;
    jac_to_aff2(h_lr[2 * q], h_lr[2 * q + 1], Fb, &l0, &r0);
    const Aff Lp = pt_to_aff(pt_add(pt_from_aff(l0, Fb), jb.Qtab->mul(jb.cross[0], F), Fb), Fb);
    const Aff Rp = pt_to_aff(pt_add(pt_from_aff(r0, Fb), jb.Qtab->mul(jb.cross[1], F), Fb), Fb);
    const Aff lr[2] = {Lp, Rp};
    tr.absorb_pt(jb.label, lr, 2, Fb);
    const Fe x = tr.challenge(jb.label, F, raw);
    const Fe xi = inverse(x, F);
    vdf_fe* vecs[2] = {(vdf_fe*)jb.d_a, (vdf_fe*)jb.d_b};
    const Fe c_lo[2] = {x, xi}, c_hi[2] = {xi, x};
    HIPCALL(ctx, vdf_fold_halves(ctx, sd.field, 2, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, jb.nj));
    HIPCALL(ctx, vdf_scale_pattern(ctx, sd.field, (vdf_fe*)jb.d_s, jb.n, jb.nj, (const vdf_fe*)&xi, (const vdf_fe*)&x));
    jb.out->L.push_back(Lp); jb.out->R.push_back(Rp);
    jb.nj >>= 1;
    return VDF_OK;
  };
  for (int q = 0; q < 2; ++q)
    if (jobs[q].nj > IPA_STOP) { int rc = enqueue(q); if (rc != VDF_OK) return rc; }
  for (int q = 0; q < 2; ++q) jobs[q].make_q(pp->gen_u, Fb);             // under the first round's MSMs
  for (;;) {
    const bool active[2] = {jobs[0].nj > IPA_STOP, jobs[1].nj > IPA_STOP};       // in THIS round (the lockstep loop's `act`)
    if (!active[0] && !active[1]) break;
    w_marked = false;
    for (int q = 0; q < 2; ++q) {
      if (!active[q]) continue;
      { int rc = finish(q); if (rc != VDF_OK) return rc; }
      if (jobs[q].nj > IPA_STOP) { int rc = enqueue(q); if (rc != VDF_OK) return rc; }
    }
  }
  for (int q = 0; q < 2; ++q) {
    HIPCALL(cq[q], vdf_ctx_sync(cq[q]));
    jobs[q].out->a.resize(jobs[q].nj);
    HIPCALL(cq[q], vdf_dev_memcpy(cq[q], jobs[q].out->a.data(), jobs[q].d_a, jobs[q].nj * 32));
  }
  return VDF_OK;
}

// Several inner-product arguments in lockstep (the test oracle's ipa_prove_many): statements and values are absorbed job
// by job; then every round ALL still-active jobs put their L and R into ONE batched MSM (up to four groups: one sort, one
// accumulate grid, one bucket reduction) before any challenge of the round is drawn, and job by job absorb them, draw
// their challenge and fold.  h_lr: 2 * jobs pinned result slots.
int ipa_prove_many(const Side& sd, Transcript& tr, IpaJob* jobs, int njobs, vdf_jac* h_lr) {
  const Side* pp = &sd;
  vdf_ctx* ctx = sd.ctx;
  const Field& F = *sd.F;
  const Field& Fb = *sd.Fb;
  uint64_t raw[4];
  if (njobs < 1 || njobs > 2) return fail(VDF_ERR_BAD_ARG, "one or two openings at a time (four MSMs per batch)");
  for (int q = 0; q < njobs; ++q) {
    IpaJob& jb = jobs[q];
    tr.absorb_pt(jb.label, &jb.P, 1, Fb);
    tr.absorb_fe(jb.label, &jb.v, 1, F);
    tr.challenge(jb.label, F, jb.q_raw);
    jb.nj = jb.n;
    jb.out->L.clear(); jb.out->R.clear();
  }
  if (njobs == 2 && sd.ctx_b) return ipa_prove_two_queues(sd, tr, jobs, h_lr);
  for (int q = 0; q < njobs; ++q) jobs[q].make_q(pp->gen_u, Fb);
  for (;;) {
    IpaJob* act[2];
    int na = 0;
    for (int q = 0; q < njobs; ++q) if (jobs[q].nj > IPA_STOP) act[na++] = &jobs[q];
    if (!na) break;
    size_t off[4] = {0, 0, 0, 0}, len[4];
    const vdf_fe* sc[4];
    for (int q = 0; q < na; ++q) {
      IpaJob& jb = *act[q];
      const vdf_fe* ab[2] = {(const vdf_fe*)jb.d_a, (const vdf_fe*)jb.d_b};
      HIPCALL(ctx, vdf_reduce(ctx, sd.field, VDF_REDUCE_IPA_CROSS, ab, nullptr, jb.nj, (vdf_fe*)jb.cross));      // pinned: no wait here
      HIPCALL(ctx, vdf_ipa_scalars(ctx, sd.field, (const vdf_fe*)jb.d_a, (const vdf_fe*)jb.d_s, jb.n, jb.nj, (vdf_fe*)jb.d_sL,
                                   (vdf_fe*)jb.d_sR));
      sc[2 * q] = (const vdf_fe*)jb.d_sL; sc[2 * q + 1] = (const vdf_fe*)jb.d_sR;
      len[2 * q] = len[2 * q + 1] = jb.n;
    }
    HIPCALL(ctx, vdf_msm_batch(ctx, pp->gens, 2 * na, off, sc, len, 1, h_lr));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    for (int q = 0; q < na; ++q) {
      IpaJob& jb = *act[q];
      Aff l0, r0;
      jac_to_aff2(h_lr[2 * q], h_lr[2 * q + 1], Fb, &l0, &r0);
      const Aff Lp = pt_to_aff(pt_add(pt_from_aff(l0, Fb), jb.Qtab->mul(jb.cross[0], F), Fb), Fb);
      const Aff Rp = pt_to_aff(pt_add(pt_from_aff(r0, Fb), jb.Qtab->mul(jb.cross[1], F), Fb), Fb);
      const Aff lr[2] = {Lp, Rp};
      tr.absorb_pt(jb.label, lr, 2, Fb);
      const Fe x = tr.challenge(jb.label, F, raw);
      const Fe xi = inverse(x, F);
      vdf_fe* vecs[2] = {(vdf_fe*)jb.d_a, (vdf_fe*)jb.d_b};
      const Fe c_lo[2] = {x, xi}, c_hi[2] = {xi, x};
      HIPCALL(ctx, vdf_fold_halves(ctx, sd.field, 2, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, jb.nj));
      HIPCALL(ctx, vdf_scale_pattern(ctx, sd.field, (vdf_fe*)jb.d_s, jb.n, jb.nj, (const vdf_fe*)&xi, (const vdf_fe*)&x));
      jb.out->L.push_back(Lp); jb.out->R.push_back(Rp);
      jb.nj >>= 1;
    }
  }
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  for (int q = 0; q < njobs; ++q) {
    jobs[q].out->a.resize(jobs[q].nj);
    HIPCALL(ctx, vdf_dev_memcpy(ctx, jobs[q].out->a.data(), jobs[q].d_a, jobs[q].nj * 32));
  }
  return VDF_OK;
}

// b is eq(rb, .): its fold is a closed form; the coefficient vector of the folded generators is a tensor-product table
struct IpaCheck {
  const char* label;
  size_t n;
  const std::vector<Fe>* rb;       // one entry per variable of the full vector
  Fe v;
  Aff P;
  const Ipa* proof;
  // state
  size_t k = 0, m = 0, idx = 0;
  int log_m = 0;
  Pt Qp, acc;
  Fe bfin;
  std::vector<Fe> xs, xis;
};

// The transcript order of ipa_prove_many; then, per opening, two device MSMs: sum_j x_j^2 L_j + x_j^-2 R_j (2k
// multiplications by full-size scalars: on the host they were the whole cost of verification, 0.25 ms each) and the
// folded generators against the sent vector.
int ipa_verify_many(const Side& sd, Transcript& tr, IpaCheck* jobs, int njobs, void* d_s, bool* ok) {
  const Side* pp = &sd;
  vdf_ctx* ctx = sd.ctx;
  const Field& F = *sd.F;
  const Field& Fb = *sd.Fb;
  *ok = false;
  uint64_t raw[4];
  for (int q = 0; q < njobs; ++q) {
    IpaCheck& c = jobs[q];
    c.k = c.proof->L.size();
    c.m = ipa_final(c.n);
    int log_n = 0;
    while (((size_t)1 << c.log_m) < c.m) ++c.log_m;
    while (((size_t)1 << log_n) < c.n) ++log_n;
    if (((size_t)1 << log_n) != c.n || c.k != ipa_rounds(c.n) || c.proof->R.size() != c.k || c.proof->a.size() != c.m ||
        c.rb->size() != (size_t)log_n)
      return VDF_OK;
  }
  for (int q = 0; q < njobs; ++q) {
    IpaCheck& c = jobs[q];
    tr.absorb_pt(c.label, &c.P, 1, Fb);
    tr.absorb_fe(c.label, &c.v, 1, F);
    tr.challenge(c.label, F, raw);
    c.Qp = pt_mul(pt_from_aff(pp->gen_u, Fb), raw, 128, Fb);
    c.acc = pt_add(pt_from_aff(c.P, Fb), pt_mul_fe(c.Qp, c.v, F, Fb), Fb);
    c.bfin = one(F);
    c.xs.resize(c.k); c.xis.resize(c.k);
  }
  for (;;) {
    bool any = false;
    for (int q = 0; q < njobs; ++q) {
      IpaCheck& c = jobs[q];
      if (c.idx >= c.k) continue;
      any = true;
      const size_t j = c.idx++;
      const Aff lr[2] = {c.proof->L[j], c.proof->R[j]};
      tr.absorb_pt(c.label, lr, 2, Fb);
      const Fe x = tr.challenge(c.label, F, raw);
      if (x.is_zero()) return VDF_OK;
      const Fe xi = inverse(x, F);
      const Fe& r = (*c.rb)[j];
      c.bfin = mul(c.bfin, add(mul(sub(one(F), r, F), xi, F), mul(r, x, F), F), F);
      c.xs[j] = x; c.xis[j] = xi;
    }
    if (!any) break;
  }
  bool all = true;
  for (int q = 0; q < njobs; ++q) {
    IpaCheck& c = jobs[q];
    const size_t k = c.k, m = c.m;
    if (k) {
      std::vector<Aff> lr_pts(2 * k);
      std::vector<Fe> lr_sc(2 * k);
      for (size_t j = 0; j < k; ++j) {
        lr_pts[2 * j] = c.proof->L[j]; lr_sc[2 * j] = sqr(c.xs[j], F);
        lr_pts[2 * j + 1] = c.proof->R[j]; lr_sc[2 * j + 1] = sqr(c.xis[j], F);
      }
      vdf_bases* lrb = nullptr;
      HIPCALL(ctx, vdf_bases_upload(ctx, sd.curve, (const vdf_affine*)lr_pts.data(), 2 * k, &lrb));
      vdf_jac jlr;
      const int rc = vdf_msm(ctx, lrb, 0, (const vdf_fe*)lr_sc.data(), 2 * k, 1, &jlr);
      const std::string err = rc == VDF_OK ? "" : vdf_last_error(ctx);
      vdf_bases_free(lrb);
      if (rc != VDF_OK) return fail(rc, "vdf_msm (L, R): " + err);
      HIPCALL(ctx, vdf_ctx_sync(ctx));
      c.acc = pt_add(c.acc, pt_from_aff(jac_to_aff(jlr, Fb), Fb), Fb);
    }
    // coefficients of the original generators in sum_i a_i G'_i, G'_i the folded generators: the table of the performed
    // rounds over the top index bits times the sent vector over the low ones; b folded in closed form: bfin (the
    // rounds) times eq over the remaining variables
    HIPCALL(ctx, vdf_pair_table_pattern(ctx, sd.field, (const vdf_fe*)c.xis.data(), (const vdf_fe*)c.xs.data(), (int)k,
                                        (const vdf_fe*)c.proof->a.data(), c.log_m, (vdf_fe*)d_s));
    vdf_jac jg;
    HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)d_s, c.n, 1, &jg));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    Fe ab = zero();
    for (size_t i = 0; i < m; ++i) {
      Fe bi = c.bfin;
      for (int j = 0; j < c.log_m; ++j) {
        const Fe& r = (*c.rb)[k + j];
        bi = mul(bi, ((i >> (c.log_m - 1 - j)) & 1) ? r : sub(one(F), r, F), F);
      }
      ab = add(ab, mul(c.proof->a[i], bi, F), F);
    }
    const Pt rhs = pt_add(pt_from_aff(jac_to_aff(jg, Fb), Fb), pt_mul_fe(c.Qp, ab, F, Fb), Fb);
    const Aff a1 = pt_to_aff(c.acc, Fb), a2 = pt_to_aff(rhs, Fb);
    all = all && memcmp(&a1, &a2, sizeof(Aff)) == 0;
  }
  *ok = all;
  return VDF_OK;
}

// tables of this many entries and fewer finish their sum-check on the host (a power of two; at least 2)
constexpr size_t SUMCHECK_HOST_TAIL = 512;

int spartan_prove(const Side& sd, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const void* d_z, const void* d_E,
                  Spartan* out) {
  const Side* pp = &sd;
  vdf_ctx* ctx = sd.ctx;
  const Field& F = *sd.F;
  const Layout L = layout_of(sd);
  if (L.l1 > 24 || L.s > 24) return fail(VDF_ERR_BAD_LENGTH, "shape too large for the compression SNARK (2^24 entries)");
  if (L.NW > pp->num_gens) return fail(VDF_ERR_BAD_LENGTH, "not enough generators");
  const size_t nv = pp->num_vars, nc = pp->num_cons;
  DevBufs bufs(ctx, sd.arena);
  { int rc = bufs.reserve(5 * L.M + 2 * L.Z + 4 * L.NW + pp->ncols); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  void *d_eq, *d_az, *d_bz, *d_cz, *d_e, *d_cols, *d_mvec, *d_zpad, *d_w, *d_s, *d_sL, *d_sR;
  for (void** p : {&d_eq, &d_az, &d_bz, &d_cz, &d_e}) { int rc = bufs.zeros(L.M, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  for (void** p : {&d_mvec, &d_zpad}) { int rc = bufs.zeros(L.Z, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  for (void** p : {&d_w, &d_s, &d_sL, &d_sR}) { int rc = bufs.zeros(L.NW, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = bufs.zeros(pp->ncols, &d_cols); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  vdf_jac* h_lr = nullptr;                        // pinned: four result points of a round's batched MSM, then 2 x 2 cross terms
  HIPCALL(ctx, vdf_host_alloc(ctx, 4 * sizeof(vdf_jac) + 4 * sizeof(Fe), (void**)&h_lr));
  struct HostFree { vdf_ctx* c; void* p; ~HostFree() { vdf_host_free(c, p); } } hf{ctx, h_lr};
  Fe* h_cross = reinterpret_cast<Fe*>(h_lr + 4);

  Transcript tr("compress");
  instance_bytes(tr, sd, cW, cE, u, X);
  HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)d_z, (vdf_fe*)d_az, (vdf_fe*)d_bz, (vdf_fe*)d_cz));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_e, d_E, nc * 32));
  uint64_t raw[4];
  std::vector<Fe> tau(L.s);
  for (int j = 0; j < L.s; ++j) tau[j] = tr.challenge("tau", F, raw);
  { int rc = eq_table_dev(sd, tau, d_eq); if (rc != VDF_OK) return rc; }
  // ---- outer sum-check -----------------------------------------------------------------------------------
  std::vector<Fe> rx;
  out->outer.clear();
  {
    const vdf_fe* tabs[5] = {(const vdf_fe*)d_eq, (const vdf_fe*)d_az, (const vdf_fe*)d_bz, (const vdf_fe*)d_cz, (const vdf_fe*)d_e};
    vdf_fe* vecs[5] = {(vdf_fe*)d_eq, (vdf_fe*)d_az, (vdf_fe*)d_bz, (vdf_fe*)d_cz, (vdf_fe*)d_e};
    size_t n = L.M;
    for (; n > SUMCHECK_HOST_TAIL; n >>= 1) {
      std::array<Fe, 3> ev;
      HIPCALL(ctx, vdf_reduce(ctx, sd.field, VDF_REDUCE_R1CS_ROUND, tabs, (const vdf_fe*)&u, n, (vdf_fe*)ev.data()));
      tr.absorb_fe("outer", ev.data(), 3, F);
      const Fe r = tr.challenge("outer", F, raw);
      const Fe omr = sub(one(F), r, F);
      const Fe c_lo[5] = {omr, omr, omr, omr, omr}, c_hi[5] = {r, r, r, r, r};
      HIPCALL(ctx, vdf_fold_halves(ctx, sd.field, 5, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, n));
      out->outer.push_back(ev);
      rx.push_back(r);
    }
    // The last rounds on the HOST: a round over a table of a few hundred entries is two launches, a synchronisation and the
    // transcript -- ~0.25 ms of latency for microseconds of arithmetic -- nine times per sum-check.  The tables come down once
    // (5 x 512 elements) and the same sums, the same challenges and the same folds follow in host arithmetic (exact: the
    // values are the kernels', snark.hip reduce_term<2> / k_fold_halves).
    std::vector<Fe> hv[5];
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    for (int k = 0; k < 5; ++k) { hv[k].resize(n); HIPCALL(ctx, vdf_dev_memcpy(ctx, hv[k].data(), vecs[k], n * 32)); }
    for (; n > 1; n >>= 1) {
      const size_t h = n / 2;
      std::array<Fe, 3> ev = {zero(), zero(), zero()};
      for (size_t i = 0; i < h; ++i) {
        Fe lo[5], d[5], v[5];
        for (int k = 0; k < 5; ++k) { lo[k] = hv[k][i]; d[k] = sub(hv[k][h + i], lo[k], F); }
        auto term = [&](const Fe* w) { return mul(w[0], sub(sub(mul(w[1], w[2], F), mul(u, w[3], F), F), w[4], F), F); };
        ev[0] = add(ev[0], term(lo), F);
        for (int k = 0; k < 5; ++k) v[k] = add(lo[k], add(d[k], d[k], F), F);
        ev[1] = add(ev[1], term(v), F);
        for (int k = 0; k < 5; ++k) v[k] = add(v[k], d[k], F);
        ev[2] = add(ev[2], term(v), F);
      }
      tr.absorb_fe("outer", ev.data(), 3, F);
      const Fe r = tr.challenge("outer", F, raw);
      const Fe omr = sub(one(F), r, F);
      for (int k = 0; k < 5; ++k)
        for (size_t i = 0; i < h; ++i) hv[k][i] = add(mul(omr, hv[k][i], F), mul(r, hv[k][h + i], F), F);
      out->outer.push_back(ev);
      rx.push_back(r);
    }
    out->claims[0] = hv[1][0]; out->claims[1] = hv[2][0]; out->claims[2] = hv[3][0]; out->claims[3] = hv[4][0];
  }
  tr.absorb_fe("claims", out->claims, 4, F);
  const Fe rho = tr.challenge("rho", F, raw);
  // ---- inner sum-check -----------------------------------------------------------------------------------
  void* d_eq_rx = d_az;                                       // the outer tables are spent: reuse one as eq(r_x, .)
  { int rc = eq_table_dev(sd, rx, d_eq_rx); if (rc != VDF_OK) return rc; }
  { int rc = m_vector_dev(sd, L, d_eq_rx, rho, d_cols, d_mvec); if (rc != VDF_OK) return rc; }
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_zpad, d_z, nv * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)d_zpad + L.NW * 32, (const char*)d_z + nv * 32, (1 + NUM_IO) * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_w, d_z, nv * 32));
  std::vector<Fe> ry;
  out->inner.clear();
  {
    const vdf_fe* tabs[2] = {(const vdf_fe*)d_mvec, (const vdf_fe*)d_zpad};
    vdf_fe* vecs[2] = {(vdf_fe*)d_mvec, (vdf_fe*)d_zpad};
    size_t n = L.Z;
    for (; n > SUMCHECK_HOST_TAIL; n >>= 1) {
      std::array<Fe, 2> ev;
      HIPCALL(ctx, vdf_reduce(ctx, sd.field, VDF_REDUCE_QUADRATIC_ROUND, tabs, nullptr, n, (vdf_fe*)ev.data()));
      tr.absorb_fe("inner", ev.data(), 2, F);
      const Fe r = tr.challenge("inner", F, raw);
      const Fe omr = sub(one(F), r, F);
      const Fe c_lo[2] = {omr, omr}, c_hi[2] = {r, r};
      HIPCALL(ctx, vdf_fold_halves(ctx, sd.field, 2, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, n));
      out->inner.push_back(ev);
      ry.push_back(r);
    }
    std::vector<Fe> hv[2];                                      // the last rounds on the host (as in the outer sum-check)
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    for (int k = 0; k < 2; ++k) { hv[k].resize(n); HIPCALL(ctx, vdf_dev_memcpy(ctx, hv[k].data(), vecs[k], n * 32)); }
    for (; n > 1; n >>= 1) {
      const size_t h = n / 2;
      std::array<Fe, 2> ev = {zero(), zero()};
      for (size_t i = 0; i < h; ++i) {
        const Fe p0 = hv[0][i], p1 = hv[0][h + i], q0 = hv[1][i], q1 = hv[1][h + i];
        ev[0] = add(ev[0], mul(p0, q0, F), F);
        ev[1] = add(ev[1], mul(sub(add(p1, p1, F), p0, F), sub(add(q1, q1, F), q0, F), F), F);
      }
      tr.absorb_fe("inner", ev.data(), 2, F);
      const Fe r = tr.challenge("inner", F, raw);
      const Fe omr = sub(one(F), r, F);
      for (int k = 0; k < 2; ++k)
        for (size_t i = 0; i < h; ++i) hv[k][i] = add(mul(omr, hv[k][i], F), mul(r, hv[k][h + i], F), F);
      out->inner.push_back(ev);
      ry.push_back(r);
    }
  }
  // ---- openings ----------------------------------------------------------------------------------------------
  void* d_eq_ry = d_mvec;                                     // spent: reuse for eq(r_y[1:], .) (NW entries)
  { std::vector<Fe> rest(ry.begin() + 1, ry.end()); int rc = eq_table_dev(sd, rest, d_eq_ry); if (rc != VDF_OK) return rc; }
  {
    const vdf_fe* tabs[2] = {(const vdf_fe*)d_w, (const vdf_fe*)d_eq_ry};
    HIPCALL(ctx, vdf_reduce(ctx, sd.field, VDF_REDUCE_DOT, tabs, nullptr, L.NW, (vdf_fe*)&out->w_eval));
  }
  tr.absorb_fe("weval", &out->w_eval, 1, F);
  // the two openings advance in lockstep (one four-group MSM per round).  W: a = W padded, b = eq(r_y[1:], .);
  // E: a = E padded to M (fresh copy: d_e was folded by the sum-check), b = eq(r_x, .).  Each needs its own coefficient
  // and scalar vectors: the E opening's live in buffers the sum-checks are done with.
  std::vector<Fe> ones_lo(24, one(F));
  HIPCALL(ctx, vdf_pair_table(ctx, sd.field, (const vdf_fe*)ones_lo.data(), (const vdf_fe*)ones_lo.data(), L.l1 - 1, (vdf_fe*)d_s));
  HIPCALL(ctx, vdf_dev_memset(ctx, d_e, 0, L.M * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_e, d_E, nc * 32));
  void *d_sE = d_eq, *d_sLE = d_bz, *d_sRE = d_cz;            // M entries each, spent by the outer sum-check (d_az holds eq(r_x, .))
  HIPCALL(ctx, vdf_pair_table(ctx, sd.field, (const vdf_fe*)ones_lo.data(), (const vdf_fe*)ones_lo.data(), L.s, (vdf_fe*)d_sE));
  IpaJob jobs[2];
  jobs[0].label = "ipaW"; jobs[0].n = L.NW; jobs[0].d_a = d_w; jobs[0].d_b = d_eq_ry; jobs[0].d_s = d_s; jobs[0].d_sL = d_sL;
  jobs[0].d_sR = d_sR; jobs[0].v = out->w_eval; jobs[0].P = cW; jobs[0].out = &out->ipaW;
  jobs[1].label = "ipaE"; jobs[1].n = L.M; jobs[1].d_a = d_e; jobs[1].d_b = d_eq_rx; jobs[1].d_s = d_sE; jobs[1].d_sL = d_sLE;
  jobs[1].d_sR = d_sRE; jobs[1].v = out->claims[3]; jobs[1].P = cE; jobs[1].out = &out->ipaE;
  jobs[0].cross = h_cross; jobs[1].cross = h_cross + 2;
  return ipa_prove_many(sd, tr, jobs, 2, h_lr);
}

int spartan_verify(const Side& sd, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const Spartan& pf, bool* ok) {
  const Side* pp = &sd;
  vdf_ctx* ctx = sd.ctx;
  const Field& F = *sd.F;
  const Layout L = layout_of(sd);
  *ok = false;
  if ((int)pf.outer.size() != L.s || (int)pf.inner.size() != L.l1) return VDF_OK;
  Transcript tr("compress");
  instance_bytes(tr, sd, cW, cE, u, X);
  uint64_t raw[4];
  std::vector<Fe> tau(L.s), rx, ry;
  for (int j = 0; j < L.s; ++j) tau[j] = tr.challenge("tau", F, raw);
  // outer
  Fe claim = zero();
  for (const auto& ev : pf.outer) {
    const Fe y[4] = {ev[0], sub(claim, ev[0], F), ev[1], ev[2]};
    tr.absorb_fe("outer", ev.data(), 3, F);
    const Fe r = tr.challenge("outer", F, raw);
    claim = interpolate(y, 4, r, F);
    rx.push_back(r);
  }
  Fe eq_tau = one(F);
  for (int j = 0; j < L.s; ++j)
    eq_tau = mul(eq_tau, add(mul(tau[j], rx[j], F), mul(sub(one(F), tau[j], F), sub(one(F), rx[j], F), F), F), F);
  const Fe a = pf.claims[0], b = pf.claims[1], c = pf.claims[2], e = pf.claims[3];
  if (claim != mul(eq_tau, sub(sub(mul(a, b, F), mul(u, c, F), F), e, F), F)) return VDF_OK;
  tr.absorb_fe("claims", pf.claims, 4, F);
  const Fe rho = tr.challenge("rho", F, raw);
  // inner
  claim = add(a, add(mul(rho, b, F), mul(sqr(rho, F), c, F), F), F);
  for (const auto& ev : pf.inner) {
    const Fe y[3] = {ev[0], sub(claim, ev[0], F), ev[1]};
    tr.absorb_fe("inner", ev.data(), 2, F);
    const Fe r = tr.challenge("inner", F, raw);
    claim = interpolate(y, 3, r, F);
    ry.push_back(r);
  }
  // M(r_y) on the device: eq(r_x, .) -> transposed product -> dot with eq(r_y, .)
  DevBufs bufs(ctx, sd.arena);
  { int rc = bufs.reserve(L.M + pp->ncols + 2 * L.Z + L.NW); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  void *d_eq_rx, *d_cols, *d_mvec, *d_eq_ry, *d_s;
  { int rc = bufs.zeros(L.M, &d_eq_rx); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = bufs.zeros(pp->ncols, &d_cols); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  for (void** p : {&d_mvec, &d_eq_ry}) { int rc = bufs.zeros(L.Z, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = bufs.zeros(L.NW, &d_s); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = eq_table_dev(sd, rx, d_eq_rx); if (rc != VDF_OK) return rc; }
  { int rc = m_vector_dev(sd, L, d_eq_rx, rho, d_cols, d_mvec); if (rc != VDF_OK) return rc; }
  { int rc = eq_table_dev(sd, ry, d_eq_ry); if (rc != VDF_OK) return rc; }
  Fe m_ry;
  {
    const vdf_fe* tabs[2] = {(const vdf_fe*)d_mvec, (const vdf_fe*)d_eq_ry};
    HIPCALL(ctx, vdf_reduce(ctx, sd.field, VDF_REDUCE_DOT, tabs, nullptr, L.Z, (vdf_fe*)&m_ry));
  }
  // z(r_y) = (1 - r_y[0]) W~(rest) + r_y[0] * (u, X)~(rest); index i of the public half has bits MSB-first over rest
  std::vector<Fe> rest(ry.begin() + 1, ry.end());
  const int k = (int)rest.size();
  Fe pub = zero();
  for (int i = 0; i < 1 + NUM_IO; ++i) {
    Fe w = one(F);
    for (int j = 0; j < k; ++j) w = mul(w, ((i >> (k - 1 - j)) & 1) ? rest[j] : sub(one(F), rest[j], F), F);
    pub = add(pub, mul(w, i == 0 ? u : X[i - 1], F), F);
  }
  const Fe z_ry = add(mul(sub(one(F), ry[0], F), pf.w_eval, F), mul(ry[0], pub, F), F);
  if (claim != mul(m_ry, z_ry, F)) return VDF_OK;
  tr.absorb_fe("weval", &pf.w_eval, 1, F);
  IpaCheck checks[2];
  checks[0].label = "ipaW"; checks[0].n = L.NW; checks[0].rb = &rest; checks[0].v = pf.w_eval; checks[0].P = cW; checks[0].proof = &pf.ipaW;
  checks[1].label = "ipaE"; checks[1].n = L.M; checks[1].rb = &rx; checks[1].v = e; checks[1].P = cE; checks[1].proof = &pf.ipaE;
  return ipa_verify_many(sd, tr, checks, 2, d_s, ok);
}

}  // namespace

struct vdf_snark {          // NovaVDFProof::Compressed, src/nova/proof.rs:54 = nova-snark CompressedSNARK
  Inst r_U1, r_U2, l_u2;    // running primary, running secondary BEFORE the last fold, the last secondary instance
  Aff T2;                   // cross-term commitment of that last fold
  Spartan sp[2];            // sp[0]: r_U1 is satisfiable; sp[1]: fold(r_U2, l_u2) is
  std::vector<Fe> zi1;      // z_i of the primary side (arity of its step circuit)
  Fe zi2[1];
  uint64_t t = 0;           // the public parameters it was made under (wire header)
  uint8_t digest[32];
};

namespace {

// instance fold on the host: U + r u with the cross-term commitment T (two 128-bit scalar multiplications)
Inst fold_instance(const Side& sd, const Inst& U, const Inst& u, const Aff& T, const uint64_t r[4]) {
  const Field& F = *sd.F;
  const Field& Fb = *sd.Fb;
  const Fe rf = int_to_fe(r, F);
  Inst o;
  o.comm_W = pt_to_aff(pt_add(pt_from_aff(U.comm_W, Fb), pt_mul(pt_from_aff(u.comm_W, Fb), r, 128, Fb), Fb), Fb);
  o.comm_E = pt_to_aff(pt_add(pt_from_aff(U.comm_E, Fb), pt_mul(pt_from_aff(T, Fb), r, 128, Fb), Fb), Fb);
  o.u = add(U.u, rf, F);
  for (int k = 0; k < NUM_IO; ++k) o.X[k] = add(U.X[k], mul(rf, u.X[k], F), F);
  return o;
}

void fold_challenge(const vdf_pp* pp, const Inst& U2, const Inst& l2, const Aff& T2, uint64_t r[4]) {
  const Field& F2 = *pp->s[SECONDARY].F;
  uint64_t ux[2][4];
  for (int k = 0; k < 2; ++k) fe_to_int(l2.X[k], F2, ux[k]);
  hash_challenge(pp->s[PRIMARY].field, pp->params[PRIMARY], to_relaxed(U2, F2), l2.comm_W, ux, T2, r, pp->ro);
}

size_t spartan_flat_size(const Spartan& p) {
  return 32 * (3 * p.outer.size() + 4 + 2 * p.inner.size() + 1 + p.ipaW.a.size() + p.ipaE.a.size()) + 128 * (p.ipaW.L.size() + p.ipaE.L.size());
}
size_t spartan_wire_size(const Layout& L) {
  return 32 * (3 * (size_t)L.s + 4 + 2 * (size_t)L.l1 + 1 + ipa_final(L.NW) + ipa_final(L.M)) + 64 * (ipa_rounds(L.NW) + ipa_rounds(L.M));
}
void spartan_resize(Spartan& p, const Layout& L) {
  p.outer.resize(L.s); p.inner.resize(L.l1);
  p.ipaW.L.resize(ipa_rounds(L.NW)); p.ipaW.R.resize(ipa_rounds(L.NW)); p.ipaW.a.resize(ipa_final(L.NW));
  p.ipaE.L.resize(ipa_rounds(L.M)); p.ipaE.R.resize(ipa_rounds(L.M)); p.ipaE.a.resize(ipa_final(L.M));
}

// the statement part of a compressed proof on the wire: instances with 32-byte points, then both z_i
// without the primary z_i (32 bytes per element of the step circuit's arity)
constexpr size_t STATEMENT_WIRE_FIXED = 5 * 32 * 2 + 3 * 32 + 32 + 32;
inline size_t statement_wire(size_t arity) { return STATEMENT_WIRE_FIXED + 32 * arity; }
}  // namespace

extern "C" {

int vdf_nova_compress(const vdf_proof* p, vdf_pp* pp, vdf_snark** out) {
  return nova_guard([&]() -> int {
    if (!p || !pp || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (p->pp != pp) return fail(VDF_ERR_BAD_ARG, "proof was made under other public parameters");
    if (p->i == 0) return fail(VDF_ERR_BAD_LENGTH, "nothing to compress");
    { int rc = finalize_l2(p); if (rc != VDF_OK) return rc; }
    vdf_ctx* ctx = pp->ctx;
    const Side& S1 = pp->s[PRIMARY];
    const Side& S2 = pp->s[SECONDARY];
    int was_async = 0;
    HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
    struct Restore { vdf_ctx* c; int a; ~Restore() { vdf_ctx_sync(c); vdf_ctx_set_async(c, a); } } restore{ctx, was_async};
    std::unique_ptr<vdf_snark> s(new vdf_snark());
    s->t = pp->t;
    memcpy(s->digest, pp->digest, 32);
    s->r_U1 = p->r[PRIMARY].inst; s->r_U2 = p->r[SECONDARY].inst; s->l_u2 = p->l2;
    s->zi1 = p->zi[PRIMARY];
    s->zi2[0] = p->zi[SECONDARY][0];
    // the last secondary instance is folded into the running one (NIFS, as a prove_step would): into scratch, the proof is
    // left as it is
    vdf_proof* q = const_cast<vdf_proof*>(p);
    SideState& s2 = q->r[SECONDARY];
    DevBufs bufs(ctx);
    void *d_fz, *d_fE;
    { int rc = bufs.zeros(S2.ncols, &d_fz); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
    { int rc = bufs.zeros(S2.num_cons, &d_fE); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
    HIPCALL(ctx, vdf_nifs_cross_term(ctx, S2.shape, (const vdf_fe*)p->d_l2z, (const vdf_fe*)s2.d_abc[0], (const vdf_fe*)s2.d_abc[1],
                                     (const vdf_fe*)s2.d_abc[2], (const vdf_fe*)&s2.inst.u, (vdf_fe*)s2.d_abc2[0], (vdf_fe*)s2.d_abc2[1],
                                     (vdf_fe*)s2.d_abc2[2], (vdf_fe*)s2.d_T));
    vdf_jac jt;
    HIPCALL(ctx, vdf_msm(ctx, S2.gens, 0, (const vdf_fe*)s2.d_T, S2.num_cons, 1, &jt));
    s->T2 = jac_to_aff(jt, *S2.Fb);
    uint64_t r[4];
    fold_challenge(pp, s->r_U2, s->l_u2, s->T2, r);
    const Inst f2 = fold_instance(S2, s->r_U2, s->l_u2, s->T2, r);
    const Fe rf = int_to_fe(r, *S2.F);
    HIPCALL(ctx, vdf_axpy(ctx, S2.field, (const vdf_fe*)s2.d_z, (const vdf_fe*)&rf, (const vdf_fe*)p->d_l2z, S2.ncols, (vdf_fe*)d_fz));
    HIPCALL(ctx, vdf_axpy(ctx, S2.field, (const vdf_fe*)s2.d_E, (const vdf_fe*)&rf, (const vdf_fe*)s2.d_T, S2.num_cons, (vdf_fe*)d_fE));
    // The two arguments are independent (a transcript each): the secondary side's runs on a second queue of the device, driven
    // by a second host thread, beside the primary's -- its small latency-bound MSMs and its host work (transcript, point
    // arithmetic between rounds) disappear under the primary side's 2^19-generator MSMs.  (nova-snark's CompressedSNARK::prove
    // runs the two provers in parallel as well.)
    std::lock_guard<std::mutex> aux_lock(pp->aux_mu);
    if (!pp->aux_ctx) {
      const int dev = vdf_ctx_device(ctx);
      if (vdf_ctx_create_pooled(&dev, 1, VDF_QUEUE_SIDE, &pp->aux_ctx) != VDF_OK) return fail(VDF_ERR_DEVICE, std::string("compress: second context: ") + vdf_last_error(nullptr));
    }
    if (!pp->aux_ctx2 && pp->tune.compress_queues) {
      const int dev = vdf_ctx_device(ctx);
      if (vdf_ctx_create_pooled(&dev, 1, VDF_QUEUE_SIDE, &pp->aux_ctx2) != VDF_OK) return fail(VDF_ERR_DEVICE, std::string("compress: third context: ") + vdf_last_error(nullptr));
    }
    vdf_ctx* cb = pp->aux_ctx;
    HIPCALL(cb, vdf_ctx_set_async(cb, 1));
    HIPCALL(cb, vdf_ctx_wait(cb, ctx));                                // the folded secondary witness was made on the first queue
    Side S2b = S2;
    S2b.ctx = cb;
    int rc2 = VDF_OK;
    std::string err2;
    std::thread side2([&] {
      try { rc2 = spartan_prove(S2b, f2.comm_W, f2.comm_E, f2.u, f2.X, d_fz, d_fE, &s->sp[1]); }
      catch (const std::exception& ex) { rc2 = VDF_ERR_DEVICE; err2 = ex.what(); }
      if (rc2 != VDF_OK && err2.empty()) err2 = vdf_nova_last_error();  // (the message is per thread: carried over by hand)
      (void)vdf_ctx_sync(cb);
    });
    int rc = VDF_OK;
    Side S1b = S1;
    S1b.ctx_b = pp->tune.compress_queues ? pp->aux_ctx2 : nullptr;   // the primary side's second opening on a queue of its own
    try { rc = spartan_prove(S1b, s->r_U1.comm_W, s->r_U1.comm_E, s->r_U1.u, s->r_U1.X, p->r[PRIMARY].d_z, p->r[PRIMARY].d_E, &s->sp[0]); }
    catch (...) { side2.join(); throw; }
    side2.join();
    if (rc != VDF_OK) return rc;
    if (rc2 != VDF_OK) return fail(rc2, "compress, secondary side: " + err2);
    *out = s.release();
    return VDF_OK;
  });
}

void vdf_nova_snark_free(vdf_snark* s) { delete s; }

// verification of the compressed proof (src/nova/proof.rs:383): the two output hashes, the last fold of the instances,
// then one argument per side
int vdf_nova_verify_compressed(const vdf_snark* s, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok) {
  return nova_guard([&]() -> int {
    if (!s || !pp || !z0 || !zi || !ok) return fail(VDF_ERR_BAD_ARG, "null argument");
    *ok = 0;
    if (s->t != pp->t || memcmp(s->digest, pp->digest, 32) != 0) return fail(VDF_ERR_BAD_ARG, "proof was made under other public parameters");
    if (num_steps == 0) return VDF_OK;
    const Side& S1 = pp->s[PRIMARY];
    const Side& S2 = pp->s[SECONDARY];
    const Field& F1 = *S1.F;
    const Field& F2 = *S2.F;
    if (s->zi1.size() != pp->arity) return VDF_OK;
    const std::vector<Fe> z0p((const Fe*)z0, (const Fe*)z0 + pp->arity), z0s(1, zero()), zi1 = s->zi1, zi2(s->zi2, s->zi2 + 1);
    uint64_t hv[4];
    hash_state(S1.field, pp->params[PRIMARY], from_u64(num_steps, F1), z0p, zi1, to_relaxed(s->r_U2, F2), hv, pp->ro);
    if (int_to_fe(hv, F2) != s->l_u2.X[0]) return VDF_OK;
    hash_state(S2.field, pp->params[SECONDARY], from_u64(num_steps, F2), z0s, zi2, to_relaxed(s->r_U1, F1), hv, pp->ro);
    if (int_to_fe(hv, F2) != s->l_u2.X[1]) return VDF_OK;
    uint64_t r[4];
    fold_challenge(pp, s->r_U2, s->l_u2, s->T2, r);
    const Inst f2 = fold_instance(S2, s->r_U2, s->l_u2, s->T2, r);
    vdf_ctx* ctx = pp->ctx;
    int was_async = 0;
    HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
    struct Restore { vdf_ctx* c; int a; ~Restore() { vdf_ctx_sync(c); vdf_ctx_set_async(c, a); } } restore{ctx, was_async};
    bool good = false;
    int rc = spartan_verify(S1, s->r_U1.comm_W, s->r_U1.comm_E, s->r_U1.u, s->r_U1.X, s->sp[0], &good);
    if (rc != VDF_OK) return rc;
    if (!good) return VDF_OK;
    rc = spartan_verify(S2, f2.comm_W, f2.comm_E, f2.u, f2.X, s->sp[1], &good);
    if (rc != VDF_OK) return rc;
    if (!good) return VDF_OK;
    *ok = (memcmp(s->zi1.data(), zi, 32 * pp->arity) == 0 && s->zi2[0].is_zero()) ? 1 : 0;       // src/nova/proof.rs:386
    return VDF_OK;
  });
}

// flat canonical encoding of the two arguments (little-endian, non-Montgomery), primary then secondary; per argument:
// outer rounds (3 each), 4 claims, inner rounds (2 each), w, then per opening (L, R) per round as affine (x, y) and the
// final vector (at most 16 elements)
size_t vdf_nova_snark_size(const vdf_snark* s) { return s ? spartan_flat_size(s->sp[0]) + spartan_flat_size(s->sp[1]) : 0; }

int vdf_nova_snark_bytes(const vdf_snark* s, uint8_t* out, size_t cap) {
  return nova_guard([&]() -> int {
    if (!s || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
    if (cap < vdf_nova_snark_size(s)) return fail(VDF_ERR_BAD_LENGTH, "buffer too small");
    uint8_t* o = out;
    for (int side = 0; side < 2; ++side) {
      const Field& F = field(side_field(side));
      const Field& Fb = field(side_field(1 - side));
      const Spartan& sp = s->sp[side];
      auto put = [&](const Fe& v, const Field& f) { const Fe c = from_mont(v, f); memcpy(o, c.l, 32); o += 32; };
      auto put_pt = [&](const Aff& a) { if (a.is_id()) { memset(o, 0, 64); o += 64; } else { put(a.x, Fb); put(a.y, Fb); } };
      for (const auto& ev : sp.outer) for (const Fe& v : ev) put(v, F);
      for (const Fe& v : sp.claims) put(v, F);
      for (const auto& ev : sp.inner) for (const Fe& v : ev) put(v, F);
      put(sp.w_eval, F);
      for (const Ipa* ip : {&sp.ipaW, &sp.ipaE}) {
        for (size_t j = 0; j < ip->L.size(); ++j) { put_pt(ip->L[j]); put_pt(ip->R[j]); }
        for (const Fe& v : ip->a) put(v, F);
      }
    }
    return VDF_OK;
  });
}

// replaces both arguments by the given encoding (the tests use it to tamper): field elements must be canonical and
// every point the identity or on its curve
int vdf_nova_snark_set_bytes(vdf_snark* s, const uint8_t* in, size_t len) {
  return nova_guard([&]() -> int {
    if (!s || !in) return fail(VDF_ERR_BAD_ARG, "null argument");
    if (len != vdf_nova_snark_size(s)) return fail(VDF_ERR_BAD_LENGTH, "encoding has the wrong length for this shape");
    const uint8_t* i = in;
    bool canonical = true, on_curve = true;
    Spartan tmp[2] = {s->sp[0], s->sp[1]};
    for (int side = 0; side < 2; ++side) {
      const Field& F = field(side_field(side));
      const Field& Fb = field(side_field(1 - side));
      Spartan& sp = tmp[side];
      auto get = [&](Fe& v, const Field& f) { Fe c; memcpy(c.l, i, 32); i += 32; if (geq(c.l, f.m)) canonical = false; v = to_mont(c, f); };
      auto get_pt = [&](Aff& a) {
        get(a.x, Fb); get(a.y, Fb);
        if (!a.is_id() && sqr(a.y, Fb) != add(mul(sqr(a.x, Fb), a.x, Fb), from_u64(5, Fb), Fb)) on_curve = false;
      };
      for (auto& ev : sp.outer) for (Fe& v : ev) get(v, F);
      for (Fe& v : sp.claims) get(v, F);
      for (auto& ev : sp.inner) for (Fe& v : ev) get(v, F);
      get(sp.w_eval, F);
      for (Ipa* ip : {&sp.ipaW, &sp.ipaE}) {
        for (size_t j = 0; j < ip->L.size(); ++j) { get_pt(ip->L[j]); get_pt(ip->R[j]); }
        for (Fe& v : ip->a) get(v, F);
      }
    }
    if (!canonical) return fail(VDF_ERR_NONCANONICAL, "a field element of the encoding is not canonical");
    if (!on_curve) return fail(VDF_ERR_NONCANONICAL, "a point of the encoding is not on its curve");
    s->sp[0] = tmp[0]; s->sp[1] = tmp[1];
    return VDF_OK;
  });
}

// ---- the whole compressed proof as one byte string ("VDFSNK03", layout in include/vdf_nova.h) -----------------------
size_t vdf_nova_snark_serialized_size(const vdf_snark* s) {
  if (!s) return 0;
  size_t n = 8 + 8 + 32 + statement_wire(s->zi1.size());
  for (const Spartan& p : s->sp)
    n += 32 * (3 * p.outer.size() + 4 + 2 * p.inner.size() + 1 + p.ipaW.a.size() + p.ipaE.a.size()) + 64 * (p.ipaW.L.size() + p.ipaE.L.size());
  return n;
}

int vdf_nova_snark_serialize(const vdf_snark* s, uint8_t* out, size_t cap) {
  return nova_guard([&]() -> int {
    if (!s || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
    if (cap < vdf_nova_snark_serialized_size(s)) return fail(VDF_ERR_BAD_LENGTH, "buffer too small");
    Side sd[2];
    for (int k = 0; k < 2; ++k) { sd[k].F = &field(side_field(k)); sd[k].Fb = &field(side_field(1 - k)); }
    uint8_t* o = out;
    memcpy(o, WIRE_MAGIC_SNARK, 8); o += 8;
    memcpy(o, &s->t, 8); o += 8;
    memcpy(o, s->digest, 32); o += 32;
    o = put_inst(o, s->r_U1, sd[0], true);
    o = put_inst(o, s->r_U2, sd[1], true);
    o = put_inst(o, s->l_u2, sd[1], false);
    pt_compress(s->T2, *sd[1].Fb, o); o += 32;
    for (const Fe& v : s->zi1) o = wire_put_fe(o, v, *sd[0].F);
    o = wire_put_fe(o, s->zi2[0], *sd[1].F);
    for (int side = 0; side < 2; ++side) {
      const Field& F = *sd[side].F;
      const Field& Fb = *sd[side].Fb;
      const Spartan& sp = s->sp[side];
      for (const auto& ev : sp.outer) for (const Fe& v : ev) o = wire_put_fe(o, v, F);
      for (const Fe& v : sp.claims) o = wire_put_fe(o, v, F);
      for (const auto& ev : sp.inner) for (const Fe& v : ev) o = wire_put_fe(o, v, F);
      o = wire_put_fe(o, sp.w_eval, F);
      for (const Ipa* ip : {&sp.ipaW, &sp.ipaE}) {
        for (size_t j = 0; j < ip->L.size(); ++j) { pt_compress(ip->L[j], Fb, o); pt_compress(ip->R[j], Fb, o + 32); o += 64; }
        for (const Fe& v : ip->a) o = wire_put_fe(o, v, F);
      }
    }
    return VDF_OK;
  });
}

// A verifier that never saw the prover's objects: bytes -> vdf_snark
int vdf_nova_snark_deserialize(vdf_pp* pp, const uint8_t* in, size_t len, vdf_snark** out) {
  return nova_guard([&]() -> int {
    if (!pp || !in || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (len < 48 + statement_wire(pp->arity)) return fail(VDF_ERR_BAD_LENGTH, "encoding is shorter than its header");
    if (memcmp(in, WIRE_MAGIC_SNARK, 8) != 0) return fail(VDF_ERR_BAD_ARG, "not this kind of encoding (magic)");
    uint64_t t;
    memcpy(&t, in + 8, 8);
    if (t != pp->t || memcmp(in + 16, pp->digest, 32) != 0) return fail(VDF_ERR_BAD_ARG, "encoding was made under other public parameters");
    const Layout L[2] = {layout_of(pp->s[0]), layout_of(pp->s[1])};
    if (len != 48 + statement_wire(pp->arity) + spartan_wire_size(L[0]) + spartan_wire_size(L[1]))
      return fail(VDF_ERR_BAD_LENGTH, "encoding has the wrong length for this shape");
    std::unique_ptr<vdf_snark> s(new vdf_snark());
    s->t = t;
    memcpy(s->digest, pp->digest, 32);
    const uint8_t* i = in + 48;
    bool canonical = true, on_curve = true;
    i = get_inst(i, &s->r_U1, pp->s[0], true, &canonical, &on_curve);
    i = get_inst(i, &s->r_U2, pp->s[1], true, &canonical, &on_curve);
    i = get_inst(i, &s->l_u2, pp->s[1], false, &canonical, &on_curve);
    on_curve &= pt_decompress(i, *pp->s[1].Fb, &s->T2); i += 32;
    s->zi1.resize(pp->arity);
    for (size_t k = 0; k < pp->arity; ++k, i += 32) canonical &= wire_get_fe(i, *pp->s[0].F, &s->zi1[k]);
    canonical &= wire_get_fe(i, *pp->s[1].F, &s->zi2[0]); i += 32;
    for (int side = 0; side < 2; ++side) {
      const Field& F = *pp->s[side].F;
      const Field& Fb = *pp->s[side].Fb;
      Spartan& p = s->sp[side];
      spartan_resize(p, L[side]);
      auto get = [&](Fe& v) { canonical &= wire_get_fe(i, F, &v); i += 32; };
      for (auto& ev : p.outer) for (Fe& v : ev) get(v);
      for (Fe& v : p.claims) get(v);
      for (auto& ev : p.inner) for (Fe& v : ev) get(v);
      get(p.w_eval);
      for (Ipa* ip : {&p.ipaW, &p.ipaE}) {
        for (size_t j = 0; j < ip->L.size(); ++j) {
          on_curve &= pt_decompress(i, Fb, &ip->L[j]);
          on_curve &= pt_decompress(i + 32, Fb, &ip->R[j]);
          i += 64;
        }
        for (Fe& v : ip->a) get(v);
      }
    }
    if (!canonical) return fail(VDF_ERR_NONCANONICAL, "a field element of the encoding is not canonical");
    if (!on_curve) return fail(VDF_ERR_NONCANONICAL, "a point of the encoding does not decode to a curve point");
    *out = s.release();
    return VDF_OK;
  });
}

}  // extern "C"
