// libvdf_nova.so, part 2: the reference's Nova proof surface (src/nova/proof.rs) on top of the kernel ABI
// (include/vdf_hip.h): public parameters, circuits, prove_step / prove_recursively, verify.  See include/vdf_nova.h
// for the stage this implements.
#include "nova_internal.hpp"

using namespace vdfnova;

namespace vdfnova {
static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace vdfnova

namespace {


void absorb_fe(Shake256& h, const Fe& a, const Field& F) { Fe c = from_mont(a, F); h.absorb(c.l, 32); }
void absorb_aff(Shake256& h, const Aff& p) {
  const Field& F = field_fp();    // Pallas coordinates live in Fp
  absorb_fe(h, p.x, F);
  absorb_fe(h, p.y, F);
}

// Builds the COO triples of the wrapped step circuit; same layout as the test oracle (step_circuit_shape)
// (constraint order of src/nova/proof.rs:176-178, :219-227, :128-133, then the six IO-binding rows).
struct Coo { std::vector<uint32_t> rows, cols; std::vector<Fe> vals; };
void build_shape(uint64_t t, Coo m[3], size_t* num_cons, size_t* num_vars) {
  const Field& F = field(PRIMARY_FIELD);
  const Fe ONE = one(F), MINUS_ONE = neg(ONE, F);
  const uint32_t nv = (uint32_t)(3 + 4 * t + 1), one_col = nv, io = nv + 1;
  auto push = [](Coo& c, uint32_t r, uint32_t col, const Fe& v) { c.rows.push_back(r); c.cols.push_back(col); c.vals.push_back(v); };
  uint32_t xv = 0, yv = 1, iv = 2, row = 0;
  for (uint64_t j = 0; j < t; ++j) {
    const uint32_t base = (uint32_t)(3 + 4 * j), new_x = base, tmp1 = base + 1, tmp2 = base + 2, new_y = base + 3;
    push(m[0], row, xv, ONE); push(m[1], row, xv, ONE); push(m[2], row, tmp1, ONE); ++row;          // x*x = tmp1
    push(m[0], row, tmp1, ONE); push(m[1], row, tmp1, ONE); push(m[2], row, tmp2, ONE); ++row;      // tmp1*tmp1 = tmp2
    push(m[0], row, tmp2, ONE); push(m[1], row, xv, ONE);                                           // tmp2*x = new_y + y - i + 1
    push(m[2], row, new_y, ONE); push(m[2], row, yv, ONE); push(m[2], row, iv, MINUS_ONE);
    push(m[2], row, one_col, from_u64(j + 1, F)); ++row;                                            // i = z_in.i - j
    xv = new_x; yv = new_y;
  }
  const uint32_t final_i = (uint32_t)(3 + 4 * t);
  push(m[0], row, final_i, ONE); push(m[1], row, one_col, ONE);                                     // final_i*1 = i - t
  push(m[2], row, iv, ONE); push(m[2], row, one_col, neg(from_u64(t, F), F)); ++row;
  const uint32_t outs[6] = {0, 1, 2, xv, yv, final_i};
  for (int k = 0; k < 6; ++k) {
    push(m[0], row, outs[k], ONE); push(m[1], row, one_col, ONE); push(m[2], row, io + k, ONE); ++row;
  }
  *num_cons = row;
  *num_vars = nv;
}




// a[i] + b[i] for affine points with one shared inversion (Montgomery's trick); identities and equal abscissae take
// the general formulas
void batch_add_affine(const std::vector<Aff>& a, const std::vector<Aff>& b, std::vector<Aff>* out) {
  const Field& F = field_fp();
  const size_t n = a.size();
  out->resize(n);
  std::vector<Fe> d(n), pre(n);
  std::vector<char> special(n, 0);
  Fe run = one(F);
  for (size_t i = 0; i < n; ++i) {
    d[i] = sub(b[i].x, a[i].x, F);
    if (a[i].is_id() || b[i].is_id() || d[i].is_zero()) { special[i] = 1; d[i] = one(F); }
    pre[i] = run;
    run = mul(run, d[i], F);
  }
  Fe inv = inverse(run, F);
  for (size_t i = n; i-- > 0;) {
    const Fe di = mul(inv, pre[i], F);           // 1 / d[i]
    inv = mul(inv, d[i], F);
    if (special[i]) { (*out)[i] = pt_to_aff(pt_add(pt_from_aff(a[i], F), pt_from_aff(b[i], F), F), F); continue; }
    const Fe lam = mul(sub(b[i].y, a[i].y, F), di, F);
    const Fe x3 = sub(sub(sqr(lam, F), a[i].x, F), b[i].x, F);
    (*out)[i].x = x3;
    (*out)[i].y = sub(mul(lam, sub(a[i].x, x3, F), F), a[i].y, F);
  }
}

// generators of the packed witness [z_in(3) | tmp1, tmp2, new_y per round | final_i] and the two fixed points of the
// correction (nova_internal.hpp)
int setup_packed_generators(vdf_pp* pp) {
  vdf_ctx* ctx = pp->ctx;
  const size_t t = pp->t, nv = pp->num_vars;
  std::vector<Aff> G(nv);
  HIPCALL(ctx, vdf_bases_download(ctx, pp->gens, 0, nv, (vdf_affine*)G.data()));
  pp->num_w = 3 * t + 4;
  std::vector<Aff> Gw(pp->num_w), a, b, sum, Gx(t);
  for (size_t j = 0; j < t; ++j) Gx[j] = G[3 + 4 * j];
  a.push_back(G[1]); b.push_back(G[3]);                                     // y_0 = z_in.y carries round 0's new_x
  for (size_t j = 0; j + 1 < t; ++j) { a.push_back(G[6 + 4 * j]); b.push_back(G[3 + 4 * (j + 1)]); }
  batch_add_affine(a, b, &sum);
  Gw[0] = G[0]; Gw[1] = sum[0]; Gw[2] = G[2];
  for (size_t j = 0; j < t; ++j) {
    Gw[3 + 3 * j] = G[4 + 4 * j];
    Gw[4 + 3 * j] = G[5 + 4 * j];
    Gw[5 + 3 * j] = (j + 1 < t) ? sum[1 + j] : G[6 + 4 * j];
  }
  Gw[3 + 3 * t] = G[3 + 4 * t];
  HIPCALL(ctx, vdf_bases_upload(ctx, PRIMARY_CURVE, (const vdf_affine*)Gw.data(), pp->num_w, &pp->gens_w));
  HIPCALL(ctx, vdf_bases_precompute(ctx, pp->gens_w, 16, 1));
  // S0 = sum_j G_{3+4j}, S1 = sum_j j G_{3+4j}: two MSMs with small scalars
  vdf_bases* bx = nullptr;
  HIPCALL(ctx, vdf_bases_upload(ctx, PRIMARY_CURVE, (const vdf_affine*)Gx.data(), t, &bx));
  std::vector<Fe> ones(t), idx(t);
  for (size_t j = 0; j < t; ++j) { ones[j] = Fe{{1, 0, 0, 0}}; idx[j] = Fe{{(uint64_t)j, 0, 0, 0}}; }
  vdf_jac j0, j1;
  int rc = vdf_msm(ctx, bx, 0, (const vdf_fe*)ones.data(), t, 0, &j0);
  if (rc == VDF_OK) rc = vdf_msm(ctx, bx, 0, (const vdf_fe*)idx.data(), t, 0, &j1);
  if (rc == VDF_OK) rc = vdf_ctx_sync(ctx);
  const std::string err = rc == VDF_OK ? "" : vdf_last_error(ctx);
  vdf_bases_free(bx);
  if (rc != VDF_OK) return fail(rc, "packed generators: " + err);
  const Field& Fb = field_fp();
  pp->S0 = jac_to_aff(j0, Fb);
  pp->S1 = jac_to_aff(j1, Fb);
  const uint64_t tk[4] = {t, 0, 0, 0};
  pp->tS0 = pt_to_aff(pt_mul(pt_from_aff(pp->S0, Fb), tk, 64, Fb), Fb);
  return VDF_OK;
}

}  // namespace

namespace vdfnova {
// r = SHAKE256(digest | U1 | u2 | comm_T) squeezed to 128 bits
Fe challenge(const vdf_pp* pp, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const Aff& cw2, const Fe* X2,
             const Aff& cT, uint64_t r_raw[4]) {
  const Field& F = field(PRIMARY_FIELD);
  Shake256 h;
  h.absorb("vdf-nova-fold-v1", 16);
  h.absorb(pp->digest, 32);
  absorb_aff(h, cW); absorb_aff(h, cE); absorb_fe(h, u, F);
  for (int k = 0; k < NUM_IO; ++k) absorb_fe(h, X[k], F);
  absorb_aff(h, cw2);
  for (int k = 0; k < NUM_IO; ++k) absorb_fe(h, X2[k], F);
  absorb_aff(h, cT);
  r_raw[0] = r_raw[1] = r_raw[2] = r_raw[3] = 0;
  h.squeeze(r_raw, 16);
  Fe r;
  memcpy(r.l, r_raw, 32);
  return to_mont(r, F);
}

Aff fold_commitment(const Aff& a, const uint64_t r_raw[4], const Aff& b) {      // a + r*b on Pallas
  const Field& F = field_fp();
  Pt rb = pt_mul(pt_from_aff(b, F), r_raw, 128, F);
  return pt_to_aff(pt_add(pt_from_aff(a, F), rb, F), F);
}

bool fold_replay(const vdf_pp* pp, const std::vector<StepRecord>& steps, Fe* r_out, Aff* cW, Aff* cE, Fe* u, Fe X[NUM_IO]) {
  const Field& F = field(PRIMARY_FIELD);
  *cW = steps[0].comm_w;
  cE->x = cE->y = zero();
  *u = one(F);
  for (int j = 0; j < NUM_IO; ++j) X[j] = steps[0].X[j];
  if (r_out) r_out[0] = zero();
  for (size_t k = 1; k < steps.size(); ++k) {
    const StepRecord& s = steps[k];
    uint64_t r_raw[4];
    const Fe r = challenge(pp, *cW, *cE, *u, X, s.comm_w, s.X, s.comm_T, r_raw);
    if (r_out) r_out[k] = r;
    else if (r != s.r) return false;
    *cW = fold_commitment(*cW, r_raw, s.comm_w);
    *cE = fold_commitment(*cE, r_raw, s.comm_T);
    *u = add(*u, r, F);
    for (int j = 0; j < NUM_IO; ++j) X[j] = add(X[j], mul(r, s.X[j], F), F);
  }
  return true;
}

int alloc_proof_buffers(vdf_proof* p) {
  vdf_pp* pp = p->pp;
  vdf_ctx* ctx = pp->ctx;
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->ncols * 32, &p->d_z1));
  for (int k = 0; k < vdf_proof::RING; ++k) HIPCALL(ctx, vdf_dev_alloc(ctx, pp->ncols * 32, &p->d_z2s[k]));
  for (int k = 0; k < vdf_proof::RING; ++k) HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_w * 32, &p->d_wps[k]));
  p->slot = 0;
  p->d_z2 = p->d_z2s[0];
  for (int k = 0; k < vdf_proof::DEPTH; ++k) {
    const int dev = vdf_ctx_device(ctx);
    if (vdf_ctx_create(&dev, 1, &p->ctx2[k]) != VDF_OK)
      return fail(VDF_ERR_DEVICE, std::string("lookahead context: ") + vdf_last_error(nullptr));
    HIPCALL(p->ctx2[k], vdf_ctx_set_async(p->ctx2[k], 1));
  }
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_cons * 32, &p->d_E));
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_cons * 32, &p->d_T));
  for (int k = 0; k < 6; ++k) HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_cons * 32, &p->d_abc[k]));
  HIPCALL(ctx, vdf_host_alloc(ctx, (vdf_proof::RING + 1) * sizeof(vdf_jac), (void**)&p->h_comm));
  HIPCALL(ctx, vdf_dev_memset(ctx, p->d_E, 0, pp->num_cons * 32));
  return VDF_OK;
}
}  // namespace vdfnova

void vdf_proof::join() const {
  if (!pending.valid) return;
  vdf_proof* self = const_cast<vdf_proof*>(this);
  self->comm_W = fold_commitment(pending.cW0, pending.r, pending.cw);
  self->comm_E = fold_commitment(pending.cE0, pending.r, pending.cT);
  pending.valid = false;
}

extern "C" {

const char* vdf_nova_last_error(void) { return vdfnova::g_err.c_str(); }

// ---- public parameters -------------------------------------------------------------------------------
int vdf_nova_public_params(vdf_ctx* ctx, uint64_t t, vdf_pp** out) {
  if (!ctx || !out || t == 0 || t > (1ull << 24)) return fail(VDF_ERR_BAD_ARG, "bad argument");
  *out = nullptr;
  vdf_pp* pp = new vdf_pp();
  pp->ctx = ctx;
  pp->t = t;
  Coo m[3];
  build_shape(t, m, &pp->num_cons, &pp->num_vars);
  pp->ncols = pp->num_vars + 1 + NUM_IO;
  pp->nnz3 = m[0].rows.size() + m[1].rows.size() + m[2].rows.size();
  const uint32_t* rows[3] = {m[0].rows.data(), m[1].rows.data(), m[2].rows.data()};
  const uint32_t* cols[3] = {m[0].cols.data(), m[1].cols.data(), m[2].cols.data()};
  const vdf_fe* vals[3] = {(const vdf_fe*)m[0].vals.data(), (const vdf_fe*)m[1].vals.data(), (const vdf_fe*)m[2].vals.data()};
  const size_t nnz[3] = {m[0].rows.size(), m[1].rows.size(), m[2].rows.size()};
  int rc = vdf_shape_create(ctx, PRIMARY_FIELD, pp->num_cons, pp->ncols, rows, cols, vals, nnz, &pp->shape);
  if (rc != VDF_OK) { std::string e = vdf_last_error(ctx); delete pp; return fail(rc, "vdf_shape_create: " + e); }
  size_t need = pp->num_vars > pp->num_cons ? pp->num_vars : pp->num_cons;
  size_t g = 1;
  while (g < need) g <<= 1;                                        // next_pow2(max(vars, cons)), SURVEY.md App. C
  pp->num_gens = g;
  rc = vdf_bases_generate_family(ctx, PRIMARY_CURVE, GENS_FAMILY, GENS_SEED, 0, g, &pp->gens);
  if (rc == VDF_OK) rc = vdf_bases_precompute(ctx, pp->gens, 16, 1);
  if (rc == VDF_OK) {
    vdf_bases* ub = nullptr;
    rc = vdf_bases_generate_family(ctx, PRIMARY_CURVE, GENS_FAMILY, GENS_SEED, g, 1, &ub);
    if (rc == VDF_OK) rc = vdf_bases_download(ctx, ub, 0, 1, (vdf_affine*)&pp->gen_u);
    if (ub) vdf_bases_free(ub);
  }
  if (rc == VDF_OK) rc = setup_packed_generators(pp);
  if (rc == VDF_OK) rc = vdf_dev_alloc(ctx, pp->num_cons * 32, &pp->d_zero);
  if (rc == VDF_OK) rc = vdf_dev_memset(ctx, pp->d_zero, 0, pp->num_cons * 32);
  if (rc != VDF_OK) { std::string e = vdf_last_error(ctx); vdf_nova_pp_free(pp); return fail(rc, "generator setup: " + e); }
  // shape digest: sizes, every COO triple in canonical form, generator family
  Shake256 h;
  h.absorb("vdf-nova-shape-v1", 17);
  uint64_t hdr[6] = {t, (uint64_t)pp->num_cons, (uint64_t)pp->num_vars, (uint64_t)NUM_IO, GENS_SEED, (uint64_t)GENS_FAMILY};
  h.absorb(hdr, sizeof(hdr));
  const Field& F = field(PRIMARY_FIELD);
  for (int k = 0; k < 3; ++k)
    for (size_t e = 0; e < m[k].rows.size(); ++e) {
      uint32_t rc2[2] = {m[k].rows[e], m[k].cols[e]};
      h.absorb(rc2, 8);
      absorb_fe(h, m[k].vals[e], F);
    }
  h.squeeze(pp->digest, 32);
  *out = pp;
  return VDF_OK;
}
void vdf_nova_pp_free(vdf_pp* pp) {
  if (!pp) return;
  if (pp->d_zero) vdf_dev_free(pp->ctx, pp->d_zero);
  if (pp->shape) vdf_shape_free(pp->shape);
  if (pp->gens) vdf_bases_free(pp->gens);
  if (pp->gens_w) vdf_bases_free(pp->gens_w);
  delete pp;
}
int vdf_nova_pp_sizes(const vdf_pp* pp, uint64_t* num_cons, uint64_t* num_vars, uint64_t* num_io, uint64_t* nnz3,
                      uint64_t* num_gens) {
  if (!pp) return fail(VDF_ERR_BAD_ARG, "null pp");
  if (num_cons) *num_cons = pp->num_cons;
  if (num_vars) *num_vars = pp->num_vars;
  if (num_io) *num_io = NUM_IO;
  if (nnz3) *nnz3 = pp->nnz3;
  if (num_gens) *num_gens = pp->num_gens;
  return VDF_OK;
}

// ---- circuits ----------------------------------------------------------------------------------------
int vdf_nova_eval_and_make_circuits(int mode, uint64_t t, size_t num_steps, const vdf_state* initial_state,
                                    vdf_fe z0_primary[3], vdf_circuits** out) {
  if (!valid_mode(mode) || !initial_state || !z0_primary || !out || t == 0) return fail(VDF_ERR_BAD_ARG, "bad argument");
  if (num_steps == 0) return fail(VDF_ERR_BAD_ARG, "num_steps must be > 0 (assert!, src/nova/proof.rs:268)");
  vdf_circuits* cs = new vdf_circuits();
  St state = load_state(initial_state);
  for (size_t s = 0; s < num_steps; ++s) {                          // :274-279
    Circuit c;
    c.t = t;
    c.input = state;                                                 // previous_state, :285-291
    c.trace_xy.resize(2 * (t + 1));
    vdf_state res;
    vdf_state in;
    store_state(&in, state);
    vdf_minroot_eval(PRIMARY_FIELD, mode, &in, t, &res, (vdf_fe*)c.trace_xy.data());
    c.result = load_state(&res);
    state = c.result;
    cs->v.push_back(std::move(c));
  }
  memcpy(z0_primary, &state, 96);                                    // z0 = final state, :278-281
  std::vector<Circuit> rev(cs->v.rbegin(), cs->v.rend());            // circuits.reverse(), :294
  cs->v.swap(rev);
  *out = cs;
  return VDF_OK;
}
int vdf_nova_circuits_upload(vdf_ctx* ctx, vdf_circuits* c) {
  if (!ctx || !c) return fail(VDF_ERR_BAD_ARG, "null argument");
  c->ctx = ctx;
  for (auto& k : c->v) {
    if (k.d_trace) continue;
    HIPCALL(ctx, vdf_dev_alloc(ctx, k.trace_xy.size() * 32, &k.d_trace));
    HIPCALL(ctx, vdf_dev_memcpy(ctx, k.d_trace, k.trace_xy.data(), k.trace_xy.size() * 32));
  }
  return VDF_OK;
}
size_t vdf_nova_circuits_len(const vdf_circuits* c) { return c ? c->v.size() : 0; }
int vdf_nova_circuit_states(const vdf_circuits* c, size_t k, vdf_state* result, vdf_state* input) {
  if (!c || k >= c->v.size()) return fail(VDF_ERR_BAD_LENGTH, "circuit index out of range");
  if (result) store_state(result, c->v[k].result);
  if (input) store_state(input, c->v[k].input);
  return VDF_OK;
}
void vdf_nova_circuits_free(vdf_circuits* c) {
  if (!c) return;
  if (c->ctx) for (auto& k : c->v) if (k.d_trace) vdf_dev_free(c->ctx, k.d_trace);
  delete c;
}

// ---- prove_step ----------------------------------------------------------------------------------------
static int prove_step_impl(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3],
                           vdf_proof** fresh);

int vdf_nova_prove_step(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3]) {
  vdf_proof* fresh = nullptr;                     // a proof object this call created (the `None` case)
  const int rc = prove_step_impl(pp, proof, circuits, k, z0, &fresh);
  if (rc != VDF_OK && fresh) vdf_nova_proof_free(fresh);      // never leak a half-built proof
  return rc;
}

static int prove_step_impl(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3],
                           vdf_proof** fresh) {
  if (!pp || !proof || !circuits || !z0) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (k >= circuits->v.size()) return fail(VDF_ERR_BAD_LENGTH, "circuit index out of range");
  const Circuit& c = circuits->v[k];
  if (c.t != pp->t) return fail(VDF_ERR_BAD_LENGTH, "circuit t differs from the public parameters");
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  vdf_proof* p = *proof;
  const bool first = (p == nullptr);
  if (first) {
    p = new vdf_proof();
    *fresh = p;
    p->pp = pp;
    memcpy(p->zi, z0, 96);
    int rc = alloc_proof_buffers(p);
    if (rc != VDF_OK) return rc;
  }
  // StepCircuit::output's debug assertion: z_i must be the circuit's result (src/nova/proof.rs:147-149)
  if (memcmp(p->zi, &c.result, 96) != 0)
    return fail(VDF_ERR_BAD_ARG, "z_i does not match the circuit's result state");
  const double t0 = now_ms();
  const size_t nc = pp->num_cons;
  // The step is enqueued asynchronously: every call below is stream-ordered, and the host waits only for the
  // two commitments the transcript needs.  The caller's synchronisation mode is restored on the way out.
  int was_async = 0;
  HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
  HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
  struct Restore { vdf_ctx* c; int a; ~Restore() { if (!a) { vdf_ctx_sync(c); vdf_ctx_set_async(c, 0); } } } restore{ctx, was_async};
  // --- fresh z2 = [z_in | per-round new_x, tmp1, tmp2, new_y | final_i | 1 | X2] in one launch, and its
  // commitment: both depend on the trace only.  Steady state: an earlier step already enqueued them on a lookahead
  // context; otherwise they are enqueued here, the same way.
  constexpr int D = vdf_proof::DEPTH, R = vdf_proof::RING;
  Fe X2[NUM_IO] = {c.result.x, c.result.y, c.result.i, c.input.x, c.input.y, c.input.i};
  enum { MARK_Z = 0, MARK_W = 1 };                // marks on a lookahead context: z2 written / its commitment landed
  bool touched[D] = {};                           // lookahead contexts that were given work by this call
  // enqueues z2 of step j into ring slot s on that step's lookahead context, makes the first context wait for z2
  // (not for the commitment: vdf_ctx_wait covers what is enqueued so far), then the commitment
  auto enqueue_fresh = [&](size_t j, int s, bool cold) -> int {
    const Circuit& cc = circuits->v[j];
    vdf_ctx* q = p->ctx2[j % D];
    if (cold) HIPCALL(q, vdf_ctx_wait(q, ctx));   // outside the steady state the ring slot may still be read by a fold
    const void* d_trace = cc.d_trace;
    if (!d_trace) {                               // trace not resident: staged through the proof's own buffer
      void*& stage = p->d_traces[j % D];
      if (!stage) HIPCALL(ctx, vdf_dev_alloc(ctx, (pp->t + 1) * 64, &stage));
      HIPCALL(q, vdf_dev_memcpy(q, stage, cc.trace_xy.data(), (pp->t + 1) * 64));
      d_trace = stage;
    }
    const Fe Xf[NUM_IO] = {cc.result.x, cc.result.y, cc.result.i, cc.input.x, cc.input.y, cc.input.i};
    const Fe u2 = one(F);
    HIPCALL(q, vdf_minroot_step_z_packed(q, PRIMARY_FIELD, (const vdf_fe*)d_trace, pp->t, (const vdf_fe*)Xf,
                                         (const vdf_fe*)&cc.input.i, (const vdf_fe*)&u2, (const vdf_fe*)Xf, (vdf_fe*)p->d_z2s[s],
                                         (vdf_fe*)p->d_wps[s]));
    HIPCALL(q, vdf_ctx_mark(q, MARK_Z));
    HIPCALL(ctx, vdf_ctx_wait(ctx, q));
    // the commitment over the packed witness and the merged generators: 3t + 4 terms instead of 4t + 4
    HIPCALL(q, vdf_msm(q, pp->gens_w, 0, (const vdf_fe*)p->d_wps[s], pp->num_w, 1, &p->h_comm[s]));
    HIPCALL(q, vdf_ctx_mark(q, MARK_W));
    touched[j % D] = true;
    vdf_proof::Ahead a;
    a.result = cc.result; a.input = cc.input; a.slot = s;
    p->ahead.push_back(a);
    return VDF_OK;
  };
  const bool hit = !p->ahead.empty() && p->ahead_circuits == circuits && p->ahead_k == k &&
                   memcmp(&p->ahead[0].result, &c.result, sizeof(St)) == 0 && memcmp(&p->ahead[0].input, &c.input, sizeof(St)) == 0;
  if (!hit) {
    if (!p->ahead.empty()) for (vdf_ctx* q : p->ctx2) HIPCALL(q, vdf_ctx_sync(q));       // lookaheads nobody came for
    p->ahead.clear();
    p->ahead_circuits = circuits;
    p->ahead_k = k;
    int rc = enqueue_fresh(k, first ? 0 : (p->slot + 1) % R, !first);
    if (rc != VDF_OK) return rc;
  }
  const int slot = p->ahead[0].slot;
  vdf_ctx* cq = p->ctx2[k % D];
  p->slot = slot;
  p->d_z2 = p->d_z2s[slot];
  const double t1 = now_ms();
  vdf_jac* jw = &p->h_comm[slot];
  vdf_jac* jt = &p->h_comm[R];
  Aff comm_w;
  StepRecord rec;
  for (int j = 0; j < NUM_IO; ++j) rec.X[j] = X2[j];
  double t2 = t1, t3 = t1, t4 = t1, t5 = t1, t6 = t1;
  // keeps the fresh work of the next D steps enqueued (called once this step's commitment has landed: its
  // context is free again)
  auto look_ahead = [&]() -> int {
    p->ahead.erase(p->ahead.begin());
    p->ahead_k = k + 1;
    while (p->ahead.size() < (size_t)D) {
      const size_t j = p->ahead_k + p->ahead.size();
      if (j >= circuits->v.size() || circuits->v[j].t != pp->t) break;
      const int last = p->ahead.empty() ? slot : p->ahead.back().slot;
      int rc = enqueue_fresh(j, (last + 1) % R, !hit);
      if (rc != VDF_OK) return rc;
    }
    return VDF_OK;
  };
  // commitment of the fresh witness = (MSM over the packed witness) - ((i_0 - 1) S0 - S1), i_0 = the step's counter
  auto fresh_commitment = [&](const vdf_jac& j) -> Aff {
    const Field& Fb = field_fp();
    if (p->c_valid && sub(p->c_i0, from_u64(pp->t, F), F) == c.result.i) {
      Aff m = pp->tS0;
      m.y = neg(m.y, Fb);
      p->c_pt = pt_add(p->c_pt, pt_from_aff(m, Fb), Fb);                     // consecutive steps: C -= t S0
    } else {
      const Fe i0m1 = from_mont(sub(c.result.i, one(F), F), F);              // i_0 - 1 as an integer
      Aff s1 = pp->S1;
      s1.y = neg(s1.y, Fb);
      p->c_pt = pt_add(pt_mul(pt_from_aff(pp->S0, Fb), i0m1.l, 255, Fb), pt_from_aff(s1, Fb), Fb);
    }
    p->c_i0 = c.result.i;
    p->c_valid = true;
    Fe X, Y, Z;
    memcpy(X.l, j.x.l, 32); memcpy(Y.l, j.y.l, 32); memcpy(Z.l, j.z.l, 32);
    Pt w;
    w.x = X; w.y = Y; w.zz = sqr(Z, Fb); w.zzz = mul(w.zz, Z, Fb);
    Pt cneg = p->c_pt;
    cneg.y = neg(cneg.y, Fb);
    return pt_to_aff(pt_add(w, cneg, Fb), Fb);
  };
  if (first) {
    // running := fresh as a relaxed instance (E = 0, u = 1); the `None` case of prove_step.  A z, B z, C z of
    // the running instance are computed here once and folded from then on (they are linear in z).
    HIPCALL(ctx, vdf_dev_memcpy(ctx, p->d_z1, p->d_z2, pp->ncols * 32));
    HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)p->d_z1, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]));
    HIPCALL(cq, vdf_ctx_sync(cq));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    t2 = now_ms();
    comm_w = fresh_commitment(*jw);
    p->comm_W = comm_w;
    p->comm_E.x = p->comm_E.y = zero();
    p->u = one(F);
    for (int j = 0; j < NUM_IO; ++j) p->X[j] = X2[j];
    rec.comm_T.x = rec.comm_T.y = zero();
    rec.r = zero();
    int rc = look_ahead();
    if (rc != VDF_OK) return rc;
    t6 = now_ms();
  } else {
    // --- NIFS.prove, the critical path of the chain: multiply_vec(z2) + cross term (one
    // launch), the commitment to T into pinned host memory, the challenge, the fold.  While the GPU works the host
    // finishes the previous step's instance fold.
    HIPCALL(ctx, vdf_nifs_cross_term(ctx, pp->shape, (const vdf_fe*)p->d_z2, (const vdf_fe*)p->d_abc[0], (const vdf_fe*)p->d_abc[1],
                                     (const vdf_fe*)p->d_abc[2], (const vdf_fe*)&p->u, (vdf_fe*)p->d_abc[3], (vdf_fe*)p->d_abc[4],
                                     (vdf_fe*)p->d_abc[5], (vdf_fe*)p->d_T));
    t2 = now_ms();
    HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_T, nc, 1, jt));
    t3 = now_ms();
    // this step's fresh commitment has been in flight since an earlier step; once it has landed its context is
    // free for a later step's
    HIPCALL(cq, vdf_ctx_sync_mark(cq, MARK_W));
    {
      int rc = look_ahead();
      if (rc != VDF_OK) return rc;
    }
    t4 = now_ms();
    comm_w = fresh_commitment(*jw);                              // host point work while the GPU commits to T
    p->join();                                                   // the previous step's instance fold
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    t5 = now_ms();
    const Aff comm_T = jac_to_aff(*jt, field_fp());
    uint64_t r_raw[4];
    const Fe r = challenge(pp, p->comm_W, p->comm_E, p->u, p->X, comm_w, X2, comm_T, r_raw);
    // witness fold on the device, one launch: z1 += r*z2 (W, and with it u and X), E += r*T, and the running
    // A z, B z, C z += r * (A z2, B z2, C z2)
    {
      vdf_fe* acc[5] = {(vdf_fe*)p->d_z1, (vdf_fe*)p->d_E, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]};
      const vdf_fe* add[5] = {(const vdf_fe*)p->d_z2, (const vdf_fe*)p->d_T, (const vdf_fe*)p->d_abc[3],
                              (const vdf_fe*)p->d_abc[4], (const vdf_fe*)p->d_abc[5]};
      const size_t len[5] = {pp->ncols, nc, nc, nc, nc};
      HIPCALL(ctx, vdf_fold_many(ctx, PRIMARY_FIELD, (const vdf_fe*)&r, 5, acc, add, len));
    }
    t6 = now_ms();
    // instance fold on the host (O(1)): u and X now, the two commitments deferred (vdf_proof::join)
    p->pending.cW0 = p->comm_W; p->pending.cE0 = p->comm_E;
    p->pending.cw = comm_w; p->pending.cT = comm_T;
    memcpy(p->pending.r, r_raw, 32);
    p->pending.valid = true;
    p->u = add(p->u, r, F);
    for (int j = 0; j < NUM_IO; ++j) p->X[j] = add(p->X[j], mul(r, X2[j], F), F);
    rec.comm_T = comm_T;
    rec.r = r;
  }
  // nothing in flight reads the circuits' memory once this call returns (a lookahead's z2 is long written)
  for (int j = 0; j < D; ++j) if (touched[j]) HIPCALL(p->ctx2[j], vdf_ctx_sync_mark(p->ctx2[j], MARK_Z));
  rec.comm_w = comm_w;
  p->steps.push_back(rec);
  p->i += 1;
  p->zi[0] = c.input.x; p->zi[1] = c.input.y; p->zi[2] = c.input.i;   // c1.output(zi), src/nova/proof.rs:142-152
  const double t7 = now_ms();
  // fresh witness (only without lookahead) | commitment-of-T launch | cross-term launch | wait for the fresh
  // commitment + lookahead launch | host fold of the previous step + wait for T | transcript + fold launch |
  // bookkeeping | total
  p->ms[0] = t1 - t0; p->ms[1] = t3 - t2; p->ms[2] = t2 - t1; p->ms[3] = t4 - t3;
  p->ms[4] = t5 - t4; p->ms[5] = t6 - t5; p->ms[6] = t7 - t6; p->ms[7] = t7 - t0;
  *proof = p;
  *fresh = nullptr;                               // handed over to the caller
  return VDF_OK;
}

int vdf_nova_prove_recursively(vdf_pp* pp, const vdf_circuits* circuits, uint64_t num_iters_per_step, const vdf_fe z0[3],
                               vdf_proof** out) {
  if (!pp || !circuits || !out || !z0) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (num_iters_per_step != pp->t) return fail(VDF_ERR_BAD_LENGTH, "num_iters_per_step differs from the public parameters");
  if (circuits->v.empty()) return fail(VDF_ERR_BAD_LENGTH, "no circuits (recursive_snark.unwrap(), src/nova/proof.rs:357)");
  vdf_proof* p = nullptr;
  for (size_t k = 0; k < circuits->v.size(); ++k) {                   // :318-355
    int rc = vdf_nova_prove_step(pp, &p, circuits, k, z0);
    if (rc != VDF_OK) { vdf_nova_proof_free(p); *out = nullptr; return rc; }
  }
  *out = p;
  return VDF_OK;
}

void vdf_nova_proof_free(vdf_proof* p) {
  if (!p) return;
  p->join();
  vdf_ctx* ctx = p->pp ? p->pp->ctx : nullptr;
  if (ctx) {
    for (vdf_ctx* q : p->ctx2) if (q) vdf_ctx_sync(q);                   // lookaheads may still be in flight
    vdf_ctx_sync(ctx);
    void* bufs[] = {p->d_z1, p->d_E, p->d_T, p->d_abc[0], p->d_abc[1], p->d_abc[2], p->d_abc[3], p->d_abc[4], p->d_abc[5]};
    for (void* b : bufs) if (b) vdf_dev_free(ctx, b);
    for (void* b : p->d_z2s) if (b) vdf_dev_free(ctx, b);
    for (void* b : p->d_wps) if (b) vdf_dev_free(ctx, b);
    for (void* b : p->d_traces) if (b) vdf_dev_free(ctx, b);
    for (vdf_ctx* q : p->ctx2) if (q) vdf_ctx_destroy(q);
    if (p->h_comm) vdf_host_free(ctx, p->h_comm);
  }
  delete p;
}
size_t vdf_nova_proof_num_steps(const vdf_proof* p) { return p ? p->i : 0; }

int vdf_nova_proof_instance(const vdf_proof* p, vdf_affine* comm_W, vdf_affine* comm_E, vdf_fe* u, vdf_fe X[6]) {
  if (!p) return fail(VDF_ERR_BAD_ARG, "null proof");
  p->join();
  if (comm_W) memcpy(comm_W, &p->comm_W, 64);
  if (comm_E) memcpy(comm_E, &p->comm_E, 64);
  if (u) memcpy(u, &p->u, 32);
  if (X) memcpy(X, p->X, 32 * NUM_IO);
  return VDF_OK;
}
int vdf_nova_proof_witness_ptrs(const vdf_proof* p, const void** d_W, const void** d_E) {
  if (!p) return fail(VDF_ERR_BAD_ARG, "null proof");
  HIPCALL(p->pp->ctx, vdf_ctx_sync(p->pp->ctx));      // a step may have returned with its fold still in flight
  if (d_W) *d_W = p->d_z1;
  if (d_E) *d_E = p->d_E;
  return VDF_OK;
}
int vdf_nova_proof_step_record(const vdf_proof* p, size_t k, vdf_affine* comm_w, vdf_affine* comm_T, vdf_fe* r, vdf_fe X[6]) {
  if (!p || k >= p->steps.size()) return fail(VDF_ERR_BAD_LENGTH, "step index out of range");
  const StepRecord& s = p->steps[k];
  if (comm_w) memcpy(comm_w, &s.comm_w, 64);
  if (comm_T) memcpy(comm_T, &s.comm_T, 64);
  if (r) memcpy(r, &s.r, 32);
  if (X) memcpy(X, s.X, 32 * NUM_IO);
  return VDF_OK;
}
int vdf_nova_last_step_ms(const vdf_proof* p, double ms[8]) {
  if (!p || !ms) return fail(VDF_ERR_BAD_ARG, "null argument");
  memcpy(ms, p->ms, sizeof(p->ms));
  return VDF_OK;
}

// ---- verify --------------------------------------------------------------------------------------------
int vdf_nova_verify(const vdf_proof* p, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok) {
  if (!p || !pp || !z0 || !zi || !ok) return fail(VDF_ERR_BAD_ARG, "null argument");
  *ok = 0;
  if (p->pp != pp) return fail(VDF_ERR_BAD_ARG, "proof was made under other public parameters");
  p->join();
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  if (num_steps == 0 || p->steps.size() != num_steps || p->i != num_steps) return VDF_OK;   // NovaError::ProofVerifyError
  // (1) public-IO chain: X_0.z_in = z0, X_k.z_out = X_{k+1}.z_in; the last z_out is the verified z_i
  if (memcmp(p->steps[0].X, z0, 96) != 0) return VDF_OK;
  for (size_t k = 0; k + 1 < num_steps; ++k)
    if (memcmp(&p->steps[k].X[3], &p->steps[k + 1].X[0], 96) != 0) return VDF_OK;
  // every step must move the counter by exactly t in the inverse direction
  const Fe tfe = from_u64(pp->t, F);
  for (size_t k = 0; k < num_steps; ++k)
    if (sub(p->steps[k].X[2], p->steps[k].X[5], F) != tfe) return VDF_OK;
  // (2) replay the folds of the instances
  Aff cW, cE;
  Fe u, X[NUM_IO];
  if (!fold_replay(pp, p->steps, nullptr, &cW, &cE, &u, X)) return VDF_OK;
  if (memcmp(&cW, &p->comm_W, 64) || memcmp(&cE, &p->comm_E, 64) || u != p->u || memcmp(X, p->X, sizeof(X))) return VDF_OK;
  // (3) the running witness opens the folded instance: commitments and relaxed satisfiability
  const size_t nv = pp->num_vars, nc = pp->num_cons;
  vdf_jac j1, j2;
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_z1, nv, 1, &j1));
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_E, nc, 1, &j2));
  Aff a1 = jac_to_aff(j1, field_fp()), a2 = jac_to_aff(j2, field_fp());
  if (memcmp(&a1, &p->comm_W, 64) || memcmp(&a2, &p->comm_E, 64)) return VDF_OK;
  // z = (W, u, X) must carry the instance's u and X
  std::vector<Fe> tail(1 + NUM_IO);
  HIPCALL(ctx, vdf_dev_memcpy(ctx, tail.data(), (const char*)p->d_z1 + nv * 32, tail.size() * 32));
  if (tail[0] != p->u || memcmp(&tail[1], p->X, 32 * NUM_IO)) return VDF_OK;
  HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)p->d_z1, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]));
  // residual Az*Bz - u*Cz - E through the cross-term kernel with Az2 = Bz1 = 0
  HIPCALL(ctx, vdf_cross_term(ctx, PRIMARY_FIELD, (const vdf_fe*)p->d_abc[0], (const vdf_fe*)pp->d_zero, (const vdf_fe*)p->d_E,
                              (const vdf_fe*)pp->d_zero, (const vdf_fe*)p->d_abc[1], (const vdf_fe*)p->d_abc[2],
                              (const vdf_fe*)&p->u, nc, (vdf_fe*)p->d_T));
  std::vector<uint64_t> res(nc * 4);
  HIPCALL(ctx, vdf_dev_memcpy(ctx, res.data(), p->d_T, nc * 32));
  uint64_t any = 0;
  for (uint64_t w : res) any |= w;
  if (any) return VDF_OK;
  // Ok(zi_primary == zi_primary_verified), src/nova/proof.rs:386
  *ok = memcmp(&p->steps[num_steps - 1].X[3], zi, 96) == 0 ? 1 : 0;
  return VDF_OK;
}

}  // extern "C"
